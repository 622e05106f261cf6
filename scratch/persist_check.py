# scratch: the persistent one-workgroup-per-matrix Cholesky (DPGP_POTRF_PERSISTENT=1) against NumPy, then its rate at B = 256, M = 512
# (renamed from test_persist.py: not a pytest module — it runs GPU code at import)
def main():
    import os, sys, time
    os.environ['DPGP_POTRF_PERSISTENT'] = sys.argv[1] if len(sys.argv) > 1 else '1'
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from dp_gp_lvm_amd import ops
    dev = torch.device('cuda', 0)
    for b, m in ((3, 256), (5, 512), (2, 300), (2, 200), (3, 384), (2, 640)):
        rng = np.random.default_rng(m)
        a = rng.standard_normal((b, m, m + 3)); a = a @ a.transpose(0, 2, 1) + 0.5 * m * np.eye(m)
        l_ref = np.linalg.cholesky(a)
        l, info = ops.potrf_batched(torch.as_tensor(a, dtype=torch.float64, device=dev))
        torch.cuda.synchronize()
        l = l.cpu().numpy()
        err = np.abs(l - l_ref).max() / np.abs(l_ref).max()
        print('B %d M %d: max rel err %.2e, upper max %.1e, info %s' % (b, m, err, np.abs(np.triu(l, 1)).max(), info.tolist()), flush=True)
    bad = np.eye(512)[None].repeat(2, axis=0); bad[1, 150, 150] = -1.0
    _, info = ops.potrf_batched(torch.as_tensor(bad, dtype=torch.float64, device=dev))
    print('info for a non-PD matrix (expect [0, 151]):', info.tolist(), flush=True)
    b, m = 256, 512
    g = torch.Generator(device='cpu').manual_seed(1)
    a0 = torch.randn((b, m, m), generator=g, dtype=torch.float64).to(dev)
    spd = a0 @ a0.transpose(1, 2) + m * torch.eye(m, dtype=torch.float64, device=dev)
    work = spd.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for i in range(8):
        work.copy_(spd)
        e0.record(); l, info = ops.potrf_batched(work); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    tc = []
    for i in range(5):
        e0.record(); w2 = spd.clone(); e1.record(); e1.synchronize(); tc.append(e0.elapsed_time(e1))
    t = min(ts[2:]) - min(tc[1:])
    err = float((l @ l.transpose(1, 2) - spd).abs().max() / spd.abs().max())
    print('B 256 M 512: %.3f ms (operator incl. its copy %.3f, copy %.3f) -> %.1f TFLOP/s = %.3f of 78.6; resid %.1e' % (t, min(ts[2:]), min(tc[1:]), b * m ** 3 / 3 / t / 1e9, b * m ** 3 / 3 / t / 1e9 / 78.6, err), flush=True)


if __name__ == '__main__':
    main()
