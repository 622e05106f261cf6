# scratch: the over-T model's gradients at the reference test's shape (N=200, D=22, M=75, Q=10, T=20; dpgplvm_unitttests.py:460-576)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm, dp_gp_lvm_t
prec = sys.argv[1] if len(sys.argv) > 1 else 'f64'
which = sys.argv[2] if len(sys.argv) > 2 else 't'
n, d, m, q, t = 200, 22, 75, 10, 20
rng = np.random.default_rng(3)
y = rng.standard_normal((n, d)); y = (y - y.mean(0)) / y.std(0)
np.random.seed(1)
mdl = (dp_gp_lvm_t if which == 't' else dp_gp_lvm)(y, num_latent_dims=q, num_inducing_points=m, truncation_level=t, device='cuda:0', precision=prec)
for _ in range(3): mdl.gradients()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): mdl.gradients()
torch.cuda.synchronize()
print('%s %s gradients %.3f ms' % (which, prec, (time.perf_counter() - t0) / 20 * 1e3))
