# Psi2 operator alone (fp32 matrix-core paths) by algorithm at a BASELINE config, D sliced: ms per call and the exp rate
# usage: python scratch/time_psi2_algo.py <config> <D> [algos]
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from dp_gp_lvm_amd import ops
from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
cfg = int(sys.argv[1]); n, dfull, m, q = CONFIGS[cfg]
d = int(sys.argv[2]) if len(sys.argv) > 2 else dfull
p = make_problem(cfg, d_slice=np.arange(d))
t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device='cuda:0')
args = (t(p['z']), t(p['mu']), t(p['s']), t(p['gamma']), t(p['alpha']))
ref = None
for algo in (sys.argv[3].split(',') if len(sys.argv) > 3 else ('auto', 'patch_f16')):
    out = ops.psi2(*args, algo=algo)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 5
    for _ in range(reps): out = ops.psi2(*args, algo=algo)
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / reps
    exps = d * n * m * (m + 1) / 2
    if ref is None: ref = out
    err = float((out - ref).abs().max() / ref.abs().max())
    print('config %d D=%d algo %-9s: %.3f ms per call, %.3f of the v_exp_f32 rate, max|diff| / max = %.2e' % (cfg, d, algo, ms, exps / (ms * 1e-3) / 1.966e13, err))
