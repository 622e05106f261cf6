# scratch: one optimise() iteration of the over-T model against the over-D model (the reference's only performance assertion:
# test/unittests/dpgplvm_unitttests.py:460-576, N=200, D=22, M=75, Q=10, T=20, 5000 Adam iterations each) and objectives at init
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm, dp_gp_lvm_t
shapes = [(200, 22, 75, 10, 20), (2000, 512, 128, 10, 8), (2000, 64, 128, 10, 8)]
for n, d, m, q, t in shapes:
    rng = np.random.default_rng(3)
    y = rng.standard_normal((n, d)); y = (y - y.mean(0)) / y.std(0)
    for prec in ('f64', 'mixed'):
        out = {}
        for name, fac in (('over_d', dp_gp_lvm), ('over_t', dp_gp_lvm_t)):
            np.random.seed(1)
            kw = dict(seed=1) if fac is dp_gp_lvm_t else {}
            mdl = fac(y, num_latent_dims=q, num_inducing_points=m, truncation_level=t, device='cuda:0', precision=prec, **kw)
            obj = float(mdl.objective)
            mdl.optimise(3)
            torch.cuda.synchronize(); t0 = time.perf_counter(); mdl.optimise(10); torch.cuda.synchronize()
            out[name] = (obj, (time.perf_counter() - t0) / 10 * 1e3)
        print('N %d D %d M %d Q %d T %d %-5s: objective at init over-D %.9g over-T %.9g (rel %.1e); ms per optimise() iteration over-D %.3f over-T %.3f'
              % (n, d, m, q, t, prec, out['over_d'][0], out['over_t'][0], abs(out['over_d'][0] - out['over_t'][0]) / abs(out['over_d'][0]),
                 out['over_d'][1], out['over_t'][1]), flush=True)
