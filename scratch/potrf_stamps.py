# phase times of potrf_batched_lds_kernel (workgroup 0) from the diagnostic build: DPGP_LIBRARY=scratch/libdpgp_hip_stamps.so
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo')
from dp_gp_lvm_amd import _lib, ops
b, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 128)
rng = np.random.default_rng(0)
a = rng.standard_normal((b, m, m)); a = a @ a.transpose(0, 2, 1) + m * np.eye(m)
ad = torch.as_tensor(a, device='cuda:0')
for _ in range(3): l, info = ops.potrf_batched(ad)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): l, info = ops.potrf_batched(ad)
e1.record(); torch.cuda.synchronize()
print('operator call: %.1f us (incl. the copy of the input)' % (e0.elapsed_time(e1) * 1e3 / 20))
out = (ctypes.c_longlong * 128)()
ctypes.CDLL(_lib.LIB_PATH).dpgp_debug_stamps(out)
s = list(out)
print('workgroup 0: load %.1f us, factorisation %.1f us, store %.1f us' % ((s[61] - s[60]) * 0.01, (s[62] - s[61]) * 0.01, (s[63] - s[62]) * 0.01))
