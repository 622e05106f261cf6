# scratch: the two kernel-level workloads BASELINE.json names, for rocprofv3 (kernel trace / PMC passes of scratch/prof_r03.sh):
# gram build (K_uu of config 3, fp64; one large fp32 gram) and the batched fp64 Cholesky (M = 128, B = 512; M = 512, B = 256)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops
from dp_gp_lvm_amd.utils.synthetic import make_problem
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device('cuda', 0)
p = make_problem(3)
t64 = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64, device=dev)
z, g, al, be = t64(p['z']), t64(p['gamma']), t64(p['alpha']), t64(p['beta'])
rng = np.random.default_rng(7)
x = torch.as_tensor(rng.standard_normal((4096, z.shape[1])), dtype=torch.float32, device=dev)
g32, a32, b32 = g[:16].float().contiguous(), al[:16].float().contiguous(), be[:16].float().contiguous()
mats = {}
for bb, mm in ((512, 128), (256, 512)):
    a0 = torch.as_tensor(rng.standard_normal((bb, mm, mm)), dtype=torch.float64, device=dev)
    mats[(bb, mm)] = a0 @ a0.transpose(1, 2) + mm * torch.eye(mm, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for _ in range(reps):
    ops.ard_rbf_gram(z, None, g, al, be, include_jitter=True)
    ops.ard_rbf_gram(x, None, g32, a32, b32)
    for k, spd in mats.items():
        ops.potrf_batched(spd)
torch.cuda.synchronize()
print('done')
