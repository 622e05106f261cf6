# host-side cost of one gradients() call (config 3, mixed): wall time with and without a sync per call, and a cProfile of the host side
import sys, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0, '/root/repo')
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
c = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, d, m, q = CONFIGS[c]; p = make_problem(c); t = p['phi'].shape[1]
init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']), gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
model = dp_gp_lvm(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', initial_values=init, precision='mixed')
for _ in range(3): model.gradients()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): model.gradients()
t_issue = (time.perf_counter() - t0) / 20
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(20):
    model.gradients(); torch.cuda.synchronize()
t_sync = (time.perf_counter() - t0) / 20
print('per call: host issue %.2f ms, back-to-back %.2f ms, with a sync per call %.2f ms' % (1e3 * t_issue, 1e3 * t_all, 1e3 * t_sync))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): model.gradients()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)

# the optimiser side of one training iteration
params = [v for v in model.raw.values()] if hasattr(model, 'raw') else []
g = model.gradients(); torch.cuda.synchronize()
for kwargs in (dict(), dict(foreach=True), dict(fused=True)):
    try:
        ps = [torch.nn.Parameter(v.detach().clone()) for v in g.values()]
        opt = torch.optim.Adam(ps, lr=0.01, **kwargs)
        for p_, v in zip(ps, g.values()): p_.grad = v.clone()
        for _ in range(3): opt.step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            for p_, v in zip(ps, g.values()): p_.grad = v.reshape(p_.shape).clone()
            opt.step()
        torch.cuda.synchronize()
        print('Adam', kwargs, '%.2f ms per step' % (1e3 * (time.perf_counter() - t0) / 20))
    except Exception as e:
        print('Adam', kwargs, 'failed:', str(e)[:80])
torch.cuda.synchronize(); t0 = time.perf_counter()
model.optimise(30, 0.01)
torch.cuda.synchronize()
print('optimise(): %.2f ms per iteration' % (1e3 * (time.perf_counter() - t0) / 30))
