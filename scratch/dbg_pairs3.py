import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
dev = torch.device('cuda', 0)
cfg = 3
n, d, m, q = CONFIGS[cfg]
sl = np.arange(d // 2, d // 2 + 16)
ps = make_problem(cfg, d_slice=sl)
t = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a), dtype=dt, device=dev)
args = lambda dt, nn=n: [t(ps['z'], dt), t(ps['mu'][:nn], dt), t(ps['s'][:nn], dt), t(ps['gamma'], dt), t(ps['alpha'], dt)]
for nn in (2000, 1000, 754, 753, 700):
    ref = ops.psi2(*args(torch.float64, nn)).cpu().numpy()
    got = ops.psi2(*args(torch.float32, nn)).double().cpu().numpy()
    rel = (got - ref) / np.maximum(np.abs(ref), 1e-4 * np.abs(ref).max())
    bad = np.argwhere(np.abs(rel) > 5e-5)
    print('N', nn, 'ns env', os.environ.get('DPGP_PSI2_NS'), 'bad entries', len(bad), [(tuple(int(x) for x in b), '%.2e' % rel[tuple(b)]) for b in bad[:12]], flush=True)
