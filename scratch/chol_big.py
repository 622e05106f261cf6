# scratch: the multi-workgroup batched Cholesky at config 4's shape (B = 256, M = 512, fp64) for rocprofv3 --kernel-trace
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops
b, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
dev = torch.device('cuda', 0)
g = torch.Generator(device='cpu').manual_seed(1)
a0 = torch.randn((b, m, m), generator=g, dtype=torch.float64).to(dev)
spd = a0 @ a0.transpose(1, 2) + m * torch.eye(m, dtype=torch.float64, device=dev)
for _ in range(6):
    l, info = ops.potrf_batched(spd)
torch.cuda.synchronize()
err = float((l @ l.transpose(1, 2) - spd).abs().max() / spd.abs().max())
print('B %d M %d: max |L L^T - A| / max |A| = %.2e, info max %d' % (b, m, err, int(info.abs().max())))
