# scratch: per-workgroup timeline of the fused psi2 dispatch (profile build, DPGP_LIBRARY=scratch/libdpgp_hip_prof0.so)
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops
from dp_gp_lvm_amd.utils.synthetic import make_problem
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = make_problem(cfg)
dev = torch.device('cuda', 0)
t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=dev)
args = [t(p[k]) for k in ('y', 'z', 'mu', 's', 'gamma', 'alpha', 'beta')]
n, d = p['y'].shape; m, q = p['z'].shape
w = ops.ElboWorkspace(d, n, m, q, 'mixed', dev)
for _ in range(3): ops.elbo_fhat(*args, prec='mixed', workspace=w)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ['DPGP_LIBRARY'])
nwg = min(8192, d * 3 * 8 + d)
buf = (ctypes.c_longlong * (3 * nwg))()
lib.dpgp_debug_psi2_wg(buf, nwg)
a = np.array(buf[:], dtype=np.int64).reshape(nwg, 3)
ids = np.nonzero(a[:, 1] > 0)[0]
a = a[ids]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
print('cfg %d: %d psi2 workgroups stamped; first start 0, last start %.1f, makespan %.1f us; duration mean %.1f min %.1f max %.1f' %
      (cfg, len(a), st.max(), en.max(), (en - st).mean(), (en - st).min(), (en - st).max()))
step = max(1, int(en.max()) // 16)
for lo in range(0, int(en.max()) + 1, step):
    print('   t=%5d us: psi2 workgroups running %d' % (lo, int(((st <= lo) & (en > lo)).sum())))
order = np.argsort(st)
print('   start time of the k-th psi2 workgroup: ' + ' '.join('%d:%.0f' % (k, st[order[k]]) for k in range(0, len(a), max(1, len(a) // 16))))
