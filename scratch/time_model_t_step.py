# scratch: the over-T model at a bench configuration: gradients() eager / from the HIP graph, one optimise() iteration
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm_t
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
prec = sys.argv[2] if len(sys.argv) > 2 else 'mixed'
n, d, m, q = CONFIGS[cfg]
p = make_problem(cfg)
init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']), gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'],
            beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
mdl = dp_gp_lvm_t(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=p['phi'].shape[1],
                  alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', precision=prec, initial_values=init)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print('cfg %d %s: gradients eager %.3f ms, from the graph %.3f ms, optimise() iteration %.3f ms'
      % (cfg, prec, timed(lambda: mdl.gradients()), timed(lambda: mdl.gradients(graph=True)), timed(lambda: mdl.optimise(1), 5)))
