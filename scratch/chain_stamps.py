# phase times of chain_b_kernel (workgroup 0) from the diagnostic build (DPGP_PROFILE_CHAIN): DPGP_LIBRARY=scratch/libdpgp_hip_stamps.so
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo')
from dp_gp_lvm_amd import _lib
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
c = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, d, m, q = CONFIGS[c]; p = make_problem(c); t = p['phi'].shape[1]
init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']), gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
model = dp_gp_lvm(p['y'], num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([p['s1'], p['s2']]), device='cuda:0', initial_values=init, precision='mixed')
for _ in range(5): float(model.objective)
torch.cuda.synchronize()
out = (ctypes.c_longlong * 128)()
l = ctypes.CDLL(_lib.LIB_PATH)
l.dpgp_debug_stamps(out)
s = list(out)
print('stamps (10 ns units):', s[:8])
print('assemble %.1f us, factorisation %.1f us, sums %.1f us' % ((s[1] - s[0]) * 0.01, (s[2] - s[1]) * 0.01, (s[3] - s[2]) * 0.01))
print('shader cycles per step k, panel (wave 0):', s[16:24])
print('shader cycles per step k, update (wave 1):', s[32:40])
print('   of which inside run() (summed over all evaluations / 5):', [x // 5 for x in s[48:56]])
print('shader clock during the factorisation: %.0f MHz (s_memtime ticks / wall clock)' % ((s[10] - s[9]) / ((s[2] - s[1]) * 0.01)))
