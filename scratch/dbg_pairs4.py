import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
dev = torch.device('cuda', 0)
n, d, m, q = CONFIGS[3]
sl = np.arange(d // 2, d // 2 + 16)
ps = make_problem(3, d_slice=sl)
t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a), dtype=dt, device=dev)
def run(rows, dsel, dt, algo='auto', qmask=None):
    z, mu, s = ps['z'].copy(), ps['mu'][rows].copy(), ps['s'][rows].copy()
    g = ps['gamma'][dsel].copy()
    if qmask is not None:
        g = g * qmask
    return ops.psi2(t(z, dt), t(mu, dt), t(s, dt), t(g, dt), t(ps['alpha'][dsel], dt), algo=algo).double().cpu().numpy()
for rows in ([753], [752, 753], [753, 754], list(range(740, 760))):
    for dsel in ([14], [13, 14]):
        ref = run(rows, dsel, torch.float64)
        got = run(rows, dsel, torch.float32)
        old = run(rows, dsel, torch.float32, 'patch_f16')
        i = (len(dsel) - 1, 51, 51)
        print('rows', rows[:3], len(rows), 'dims', dsel, 'psi2[.,51,51]: ref %.9e pairs %.9e (rel %.2e) patch %.9e (rel %.2e)' % (
            ref[i], got[i], got[i] / ref[i] - 1, old[i], old[i] / ref[i] - 1), flush=True)
for k in range(q):
    mask = np.ones(q); mask[k] = 1e-6
    ref = run([753], [14], torch.float64, qmask=mask); got = run([753], [14], torch.float32, qmask=mask)
    print('latent dim', k, 'switched off: rel err %.2e' % (got[0, 51, 51] / ref[0, 51, 51] - 1), flush=True)
