import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
from dp_gp_lvm_amd import ops, _lib
from dp_gp_lvm_amd.utils.synthetic import make_problem
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda')
p = make_problem(cfg)
t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=dev)
args = [t(p[k]) for k in ('y', 'z', 'mu', 's', 'gamma', 'alpha', 'beta')]
n, d = p['y'].shape; m, q = p['z'].shape
for prec in ('mixed', 'f32'):
    w = ops.ElboWorkspace(d, n, m, q, prec, dev)
    for _ in range(3):
        ops.elbo_fhat(*args, prec=prec, workspace=w)
    torch.cuda.synchronize()
    st = (ctypes.c_longlong * 16)()
    _lib.lib().dpgp_debug_stamps(st)
    s = np.array(st[:4], dtype=np.float64)
    names = ['assemble', 'potrf B', 'logdet']
    print('   accumulated over 3 evals (us): diag', st[4] / 100.0, 'panel(w1)', st[5] / 100.0, 'trailing(w1)', st[6] / 100.0)
    print(prec, 'cfg', cfg, 'chain_b total us', (s[3] - s[0]) / 100.0, {nm: round((s[i + 1] - s[i]) / 100.0, 1) for i, nm in enumerate(names)})
