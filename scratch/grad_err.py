# scratch: mixed-precision gradients against the fp64 path at a bench configuration (max abs difference / max abs value)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n, d, m, q = CONFIGS[cfg]
p = make_problem(cfg)
dev = torch.device('cuda', 0)
res = {}
for prec in ('f64', 'mixed'):
    mdl = dp_gp_lvm(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=p['phi'].shape[1],
                    alpha_prior_params=np.array([p['s1'], p['s2']]), device=dev, precision=prec,
                    initial_values=dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']),
                                        gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'],
                                        gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2']))
    res[prec] = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in mdl.gradients().items()}
for k in res['f64']:
    a, b = res['f64'][k], res['mixed'][k]
    print('%-14s max|g| %.3e  max err / max|g| %.2e   rms err / rms g %.2e' % (k, np.abs(a).max(), np.abs(a - b).max() / np.abs(a).max(),
          np.sqrt(((a - b) ** 2).mean()) / np.sqrt((a ** 2).mean())))
