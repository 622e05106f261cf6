# config-3 Adam runs: pure mixed (pair-tile stage B) against the training configuration (fp64 forward, patch-form stage B)
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
c = 3; n, d, m, q = CONFIGS[c]; p = make_problem(c); t = p['phi'].shape[1]
init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']), gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
kw = dict(num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', initial_values=init)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
res = {}
for name, a in (('mixed', dict(precision='mixed')), ('f64+mixed', dict(precision='f64', backward_precision='mixed'))):
    model = dp_gp_lvm(p['y'], **a, **kw)
    traj = []
    t0 = time.time()
    try:
        model.optimise(iters, 0.01, callback=lambda it: traj.append(float(model.objective)) if it % 20 == 19 else None)
        status = 'ok'
    except FloatingPointError as e:
        status = 'raised: ' + str(e)[:60]
    torch.cuda.synchronize()
    res[name] = traj
    print(name, status, 'ms/iter %.2f' % (1e3 * (time.time() - t0) / max(len(traj) * 20, 1)), ['%.1f' % v for v in traj])
a, b = res['mixed'], res['f64+mixed']
k = min(len(a), len(b))
print('relative difference of the trajectories:', ['%.2e' % (abs(x - y) / abs(y)) for x, y in zip(a[:k], b[:k])])
