import numpy as np, torch, sys
sys.path.insert(0, '.')
from dp_gp_lvm_amd import ops
from dp_gp_lvm_amd.utils.synthetic import make_problem
from oracle.c_oracle import COracle
dev = torch.device('cuda')
p = make_problem(2, d_slice=[0, 21, 42, 63])
c = COracle(True)
tref, info, parts = c.fhat_terms(p['y'], p['z'], p['mu'], p['s'], p['gamma'], p['alpha'], p['beta'], return_parts=True, nthreads=8)
for dt in (torch.float32, torch.float64):
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dt, device=dev)
    for algo in ('auto', 'plain'):
        p2 = ops.psi2(t(p['z']), t(p['mu']), t(p['s']), t(p['gamma']), t(p['alpha']), algo=algo).double().cpu().numpy()
        err = np.abs(p2 - parts['psi_2'])
        print(dt, algo, 'psi2 max abs err', err.max(), 'rel', (err / np.abs(parts['psi_2']).max()).max(), 'nan', np.isnan(p2).sum())
        if err.max() > 1e-3:
            d, i, j = np.unravel_index(np.argmax(err), err.shape); print('  worst at', d, i, j, p2[d, i, j], parts['psi_2'][d, i, j])
            bad = err[0] > 1e-4
            print('  bad tiles (16x16) d=0:'); print((bad.reshape(8, 16, 8, 16).sum(axis=(1, 3))))
t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=dev)
for prec in ('f64', 'mixed', 'f32'):
    w = ops.ElboWorkspace(4, 2000, 128, 10, prec, dev)
    terms, sums, inf = ops.elbo_fhat(t(p['y']), t(p['z']), t(p['mu']), t(p['s']), t(p['gamma']), t(p['alpha']), t(p['beta']), prec=prec, workspace=w)
    torch.cuda.synchronize()
    print(prec, inf.tolist(), (terms.cpu().numpy() / tref - 1))
