# scratch: cost of the sharded code path (pack -> all_reduce -> finalize) with a 1-rank RCCL group vs the single-GPU tail
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
n, d, m, q = CONFIGS[cfg]
p = make_problem(cfg)
def build(group):
    return dp_gp_lvm(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=p['phi'].shape[1],
                     alpha_prior_params=np.array([p['s1'], p['s2']]), device=dev, precision='mixed', process_group=group,
                     initial_values=dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']),
                                         gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'],
                                         gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2']))
for name, g in (('single', None), ('1-rank RCCL group', dist.group.WORLD)):
    mdl = build(g)
    for _ in range(20): mdl.evaluate_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): mdl.evaluate_()
    torch.cuda.synchronize()
    print('%-20s cfg %d: %.1f us per evaluation, objective %.9f' % (name, cfg, (time.perf_counter() - t0) / 200 * 1e6, float(mdl.evaluate_()[0])))
dist.destroy_process_group()
