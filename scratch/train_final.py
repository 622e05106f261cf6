# final check of the round-2 claims at config 3: (a) training configuration, 400 Adam iterations without a flag;
# (b) mixed mode: optimise() raises once the conditioning guard fires (pair-tile stage B in use before that)
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
c = 3; n, d, m, q = CONFIGS[c]; p = make_problem(c); t = p['phi'].shape[1]
init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']), gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
kw = dict(num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', initial_values=init)
model = dp_gp_lvm(p['y'], precision='f64', backward_precision='mixed', **kw)
o0 = float(model.objective); torch.cuda.synchronize(); t0 = time.time()
forms = []
model.optimise(400, 0.01, callback=lambda it: forms.append(model.last_stage_b_form))
torch.cuda.synchronize(); dt = time.time() - t0
sw = next((i for i, f in enumerate(forms) if f == 'mixed_patch'), None)
print('stage B form: pair-tile for the first %s iterations, patch form afterwards (%d of 400)' % (sw if sw is not None else 400, sum(f == 'mixed_patch' for f in forms)))
print('training configuration: 400 iterations, %.2f ms per iteration, objective %.1f -> %.1f' % (1e3 * dt / 400, o0, float(model.objective)), flush=True)
model = dp_gp_lvm(p['y'], precision='mixed', **kw)
done = [0]
torch.cuda.synchronize(); t0 = time.time()
try:
    model.optimise(400, 0.01, callback=lambda it: done.__setitem__(0, it + 1))
    print('mixed: no flag in 400 iterations')
except FloatingPointError as e:
    torch.cuda.synchronize()
    print('mixed: raised at iteration %d after %.2f ms per iteration (%s...)' % (done[0], 1e3 * (time.time() - t0) / max(done[0], 1), str(e)[:70]))
