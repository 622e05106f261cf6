# scratch: time one objective evaluation vs evaluation + gradients at a bench configuration
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, d, m, q = CONFIGS[cfg]
p = make_problem(cfg)
dev = torch.device('cuda', 0)
mdl = dp_gp_lvm(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=p['phi'].shape[1],
                alpha_prior_params=np.array([p['s1'], p['s2']]), device=dev, precision='mixed',
                initial_values=dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']),
                                    gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'],
                                    gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2']))
for name, fn, reps in (('objective', mdl.evaluate_, 50), ('objective + gradients', mdl.gradients, 5)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print('cfg %d %-24s %.3f ms' % (cfg, name, (time.perf_counter() - t0) / reps * 1e3))
g = mdl.gradients()
print({k: float(v.abs().max()) for k, v in g.items()})
