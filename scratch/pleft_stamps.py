# phase cycles of the left-looking persistent Cholesky (workgroup 0) from the stamped build: scratch/build_variant.sh stamps2
# potrf_persist.hip -DPP_STAMPS, DPGP_LIBRARY=scratch/libdpgp_hip_stamps2.so
import os, sys, ctypes
os.environ['DPGP_POTRF_PERSISTENT'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops, _lib
dev = torch.device('cuda', 0)
b, m = 256, 512
g = torch.Generator(device='cpu').manual_seed(1)
a0 = torch.randn((b, m, m), generator=g, dtype=torch.float64).to(dev)
spd = a0 @ a0.transpose(1, 2) + m * torch.eye(m, dtype=torch.float64, device=dev)
reps = 4
for _ in range(reps): l, info = ops.potrf_batched(spd)
torch.cuda.synchronize()
out = (ctypes.c_longlong * 64)()
ctypes.CDLL(_lib.LIB_PATH).dpgp_debug_persist_stamps(out)
s = list(out)
for k in range(4):
    if k < 3:
        print('k %d: diagonal block update + zeros %d, potrf_lds %d, write-back + inverses %d, blocks below %d (of which their updates %d)' % (
            k, s[8 * k + 1] - s[8 * k], s[8 * k + 2] - s[8 * k + 1], s[8 * k + 3] - s[8 * k + 2], s[8 * k + 4] - s[8 * k + 3], s[40 + k] // reps))
    else:
        print('k %d: diagonal block update %d, potrf_lds %d, write-back %d' % (k, s[8 * k + 1] - s[8 * k], s[8 * k + 2] - s[8 * k + 1], s[63] - s[8 * k + 2]))
print('total cycles %d' % (s[63] - s[0]))
