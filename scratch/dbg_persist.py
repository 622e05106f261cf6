import os, sys
os.environ['DPGP_POTRF_PERSISTENT'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops
dev = torch.device('cuda', 0)
m = 256
rng = np.random.default_rng(m)
a = rng.standard_normal((1, m, m + 3)); a = a @ a.transpose(0, 2, 1) + 0.5 * m * np.eye(m)
l_ref = np.linalg.cholesky(a)[0]
l, info = ops.potrf_batched(torch.as_tensor(a, dtype=torch.float64, device=dev))
l = l.cpu().numpy()[0]
sc = np.abs(l_ref).max()
for I in range(m // 16):
    print(' '.join('%8.1e' % (np.abs(l[16*I:16*I+16, 16*J:16*J+16] - l_ref[16*I:16*I+16, 16*J:16*J+16]).max() / sc) for J in range(I + 1)))
A = a[0]; L00 = l_ref[:16, :16]
blk = l[128:144, 0:16]; ref = l_ref[128:144, 0:16]; A_I0 = A[128:144, 0:16]
cands = {'expected A L^-T': A_I0 @ np.linalg.inv(L00).T, 'A L^-1': A_I0 @ np.linalg.inv(L00), 'A untouched': A_I0, '(L^-1 A^T) untransposed': (np.linalg.inv(L00) @ A_I0.T), 'A L^T': A_I0 @ L00.T, 'A L': A_I0 @ L00, 'A diag^-1': A_I0 / np.diag(L00)[None, :]}
for k_, v in cands.items(): print('%-28s max abs diff %.3e' % (k_, np.abs(blk - v).max()))
print('got row 0  ', blk[0, :6]); print('expect row0', ref[0, :6])
Li = np.linalg.inv(L00)
# implied inverse tile from the panel the kernel produced: P = A Linv^T  ->  Linv_impl^T = A^-1 P  (A_I0 16x16 generic)
Limp = np.linalg.solve(A_I0, blk).T
np.set_printoptions(linewidth=200, precision=4, suppress=True)
print('true Linv[:5,:5]\n', Li[:5, :5]); print('implied Linv[:5,:5]\n', Limp[:5, :5])
print('ratio implied/true (lower 6x6)\n', (Limp[:6, :6] / np.where(np.abs(Li[:6, :6]) > 1e-14, Li[:6, :6], np.nan)))
