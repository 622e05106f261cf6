import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
n, dfull, m, q = CONFIGS[3]; p = make_problem(3); t = p['phi'].shape[1]
for d in [int(v) for v in sys.argv[1].split(',')]:
    init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi'][:d]), gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
    model = dp_gp_lvm(p['y'][:, :d], num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', initial_values=init, precision='mixed')
    tm, info = model.per_dimension_terms
    torch.cuda.synchronize()
    print(d, 'info abs max', int(info.abs().max()), 'nonzero', int((info != 0).sum()), 'objective', float(model.objective), 'terms finite', bool(torch.isfinite(tm).all()))
print('--- with a poisoned allocator cache')
for d in [int(v) for v in sys.argv[1].split(',')]:
    junk = [torch.full((sz,), 1.2345e300, dtype=torch.float64, device='cuda:0') for sz in (1 << 27, 1 << 20, 1 << 14, 1 << 10, 256, 64)]
    del junk
    init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi'][:d]), gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
    model = dp_gp_lvm(p['y'][:, :d], num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', initial_values=init, precision='mixed')
    tm, info = model.per_dimension_terms
    bad = (info != 0).nonzero().flatten()
    print(d, 'info nonzero', int(bad.numel()), bad[:8].tolist(), 'terms finite', bool(torch.isfinite(tm).all()), 'objective', float(model.objective))
