# evaluation time by the number of output dims on one GPU (the per-GPU shares of the strong-scaling run): config 3 sliced to D columns
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
n, dfull, m, q = CONFIGS[3]; p = make_problem(3); t = p['phi'].shape[1]
base = None
for d in ([int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (512, 256, 128, 64)):
    init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi'][:d]), gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
    model = dp_gp_lvm(p['y'][:, :d], num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', initial_values=init, precision='mixed')
    for _ in range(20): model.evaluate_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): model.evaluate_()
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 200
    base = base or ms
    print('D = %3d: %.4f ms per evaluation; %d GPUs would give <= %.2fx' % (d, ms, dfull // d, base / ms))
