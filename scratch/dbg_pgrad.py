# compare the pair-form stage B with the patch-form kernel (DPGP_GRAD_PATCH=1) on random small problems
import os, sys, subprocess, numpy as np
def run(shape, patch):
    import torch
    from dp_gp_lvm_amd import ops
    n, d, m, q = shape
    rng = np.random.default_rng(m)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device='cuda:0')
    y = rng.standard_normal((n, d)); z = rng.standard_normal((m, q)) * 2.0; mu = rng.standard_normal((n, q)) * 1.5
    s = np.exp(0.3 * rng.standard_normal((n, q))); gamma = np.exp(0.3 * rng.standard_normal((d, q))) * 1.5
    alpha = np.exp(0.2 * rng.standard_normal(d))
    mp = 16 * ((m + 15) // 16)
    g = rng.standard_normal((d, mp, mp)); g = g + g.transpose(0, 2, 1); g[:, m:, :] = 0; g[:, :, m:] = 0
    wk = np.zeros((d, mp, mp)); gv = np.zeros((d, mp))
    out = ops.elbo_grad_psi(t(y), t(z), t(mu), t(s), t(gamma), t(alpha), t(g), t(wk), t(gv), prec='mixed')
    return [o.cpu().numpy() for o in out]
if __name__ == '__main__':
    if len(sys.argv) > 1:
        shape = tuple(int(v) for v in sys.argv[1].split(','))
        res = run(shape, False)
        np.savez(sys.argv[2], *res)
    else:
        for shape in [(40, 6, 12, 3), (60, 10, 15, 4), (33, 5, 17, 5), (200, 6, 70, 4), (300, 16, 100, 10)]:
            arg = ','.join(map(str, shape))
            subprocess.check_call([sys.executable, __file__, arg, '/tmp/pg_new.npz'], env=dict(os.environ))
            subprocess.check_call([sys.executable, __file__, arg, '/tmp/pg_old.npz'], env=dict(os.environ, DPGP_GRAD_PATCH='1'))
            a, b = np.load('/tmp/pg_new.npz'), np.load('/tmp/pg_old.npz')
            print(shape, [float(np.abs(a[k] - b[k]).max() / max(np.abs(b[k]).max(), 1e-300)) for k in a.files], [bool(np.isnan(a[k]).any()) for k in a.files])
