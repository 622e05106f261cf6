# scratch: find the first Adam iteration with a non-finite gradient / objective at a BASELINE configuration
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
cfg = int(sys.argv[1]); prec = sys.argv[2]; iters = int(sys.argv[3])
n, d, m, q = CONFIGS[cfg]
p = make_problem(cfg)
mdl = dp_gp_lvm(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=p['phi'].shape[1],
                alpha_prior_params=np.array([p['s1'], p['s2']]), device=torch.device('cuda', 0), precision=prec,
                initial_values=dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']),
                                    gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'],
                                    gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2']))
params = mdl.raw
opt = torch.optim.Adam(list(params.values()), lr=0.01)
for it in range(iters):
    g = mdl.gradients()
    bad = [k for k, v in g.items() if not bool(torch.isfinite(v).all())]
    terms, info = mdl.per_dimension_terms
    if bad or int(info.abs().max()) != 0 or it % 20 == 0:
        s = torch.nn.functional.softplus(params['x_var'])
        print('iter %d obj %.4f info max %d bad %s | min S %.3e max|mu| %.3f gamma atoms [%.3e, %.3e] beta atoms [%.3e, %.3e] alpha [%.3e,%.3e] max|g| %s' % (
            it, float(mdl.objective), int(info.abs().max()), bad, float(s.min()), float(params['x_mean'].abs().max()),
            float(mdl.dp_atoms[0].min()), float(mdl.dp_atoms[0].max()), float(mdl.dp_atoms[2].min()), float(mdl.dp_atoms[2].max()),
            float(mdl.dp_atoms[1].min()), float(mdl.dp_atoms[1].max()),
            {k: '%.2e' % float(v.abs().max()) for k, v in g.items() if k in ('x_mean', 'x_var', 'x_u', 'beta_atoms')}), flush=True)
    if bad or int(info.abs().max()) != 0:
        d_bad = torch.nonzero(info).flatten()[:8].tolist()
        print('first failing output dims', d_bad, 'info', info[d_bad].tolist() if d_bad else None)
        print('their beta', mdl.noise_precision.flatten()[d_bad].tolist() if d_bad else None)
        break
    for k, p_ in params.items():
        key = k
        p_.grad = g[key].reshape(p_.shape).clone()
    opt.step()
