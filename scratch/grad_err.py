# scratch: accuracy and time of the mixed-precision gradients (pair-tile stage B) against the all-fp64 gradients at a bench configuration
# usage: [DPGP_LIBRARY=...] python scratch/grad_err.py <config> [D]
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, d, m, q = CONFIGS[cfg]
p = make_problem(cfg)
dsub = int(sys.argv[2]) if len(sys.argv) > 2 else d
y = np.ascontiguousarray(p['y'][:, :dsub])
dev = torch.device('cuda', 0)
def build(prec, bprec=None):
    return dp_gp_lvm(y, num_latent_dims=q, num_inducing_points=m, truncation_level=p['phi'].shape[1],
                     alpha_prior_params=np.array([p['s1'], p['s2']]), device=dev, precision=prec, backward_precision=bprec,
                     initial_values=dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi'][:dsub]),
                                         gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'],
                                         gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2']))
ref = {k: v.cpu().numpy() for k, v in build('f64', 'f64').gradients().items()} if m <= 128 else None
for name, (prec, bprec) in {'mixed': ('mixed', None), 'mixed + fast stage B': ('mixed', 'mixed_fast'),
                            'f64 fwd + mixed stage B': ('f64', 'mixed')}.items():
    mdl = build(prec, bprec)
    for _ in range(2): g = mdl.gradients()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g = mdl.gradients()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    line = '%-24s %.3f ms ' % (name, ms)
    if ref is not None:
        line += ' '.join('%s %.1e' % (k, np.abs(v.cpu().numpy() - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-300)) for k, v in g.items())
    print('cfg %d D %d: %s' % (cfg, dsub, line), flush=True)
