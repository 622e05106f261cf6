import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
dev = torch.device('cuda', 0)
for cfg in (3, 5):
    n, d, m, q = CONFIGS[cfg]
    sl = np.arange(d // 2, d // 2 + 16)
    ps = make_problem(cfg, d_slice=sl)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a), dtype=dt, device=dev)
    args = lambda dt: [t(ps[k], dt) for k in ('z', 'mu', 's', 'gamma', 'alpha')]
    ref = ops.psi2(*args(torch.float64)).cpu().numpy()
    for algo in ('auto', 'patch_f16', 'mfma_f32', 'plain'):
        got = ops.psi2(*args(torch.float32), algo=algo).double().cpu().numpy()
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-4 * np.abs(ref).max())
        i = np.unravel_index(np.argmax(rel), rel.shape)
        zc = ps['z'] - ps['z'].mean(0)
        print('cfg', cfg, algo, 'max rel %.2e at %s (ref %.4e, max %.3e); |z_m - c|^2 = %.1f %.1f; rms rel %.2e; fro rel %.2e' % (
            rel.max(), i, ref[i], ref.max(), (zc[i[1]] ** 2).sum(), (zc[i[2]] ** 2).sum(), np.sqrt((rel ** 2).mean()),
            np.linalg.norm(got - ref) / np.linalg.norm(ref)), flush=True)
