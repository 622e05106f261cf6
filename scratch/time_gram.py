# scratch: gram build rates (kernel time by torch events around a batch of calls; bytes = output only)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops
from dp_gp_lvm_amd.utils.synthetic import make_problem
dev = torch.device('cuda', 0)
p = make_problem(3)
t64 = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float64, device=dev)
z, g, al, be = t64(p['z']), t64(p['gamma']), t64(p['alpha']), t64(p['beta'])
rng = np.random.default_rng(7)
x = torch.as_tensor(rng.standard_normal((4096, z.shape[1])), dtype=torch.float32, device=dev)
g32, a32, b32 = g[:16].float().contiguous(), al[:16].float().contiguous(), be[:16].float().contiguous()
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timed(lambda: ops.ard_rbf_gram(x, None, g32, a32, b32))
print('gram [16,4096,4096] fp32: %.1f us -> %.2f TB/s' % (ms * 1e3, 4.0 * 16 * 4096 * 4096 / ms / 1e9))
ms = timed(lambda: ops.ard_rbf_gram(z, None, g, al, be, include_jitter=True))
print('gram K_uu [512,128,128] fp64: %.1f us -> %.2f TB/s' % (ms * 1e3, 8.0 * 512 * 128 * 128 / ms / 1e9))
k = ops.ard_rbf_gram(x[:300], None, g32, a32, b32).cpu().numpy()
xs = x[:300].cpu().numpy().astype(np.float64); gg = g32.cpu().numpy().astype(np.float64)
ref = a32.cpu().numpy()[:, None, None] * np.exp(-0.5 * np.einsum('bq,ijq->bij', gg, (xs[:, None, :] - xs[None, :, :]) ** 2))
print('max rel err vs numpy %.2e' % (np.abs(k - ref).max() / np.abs(ref).max()))
