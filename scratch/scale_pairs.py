"""scratch: psi2 (stand-alone fp32 operator, pair-tile kernel) time vs N and B at M=128, Q=10."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops
dev = torch.device('cuda', 0)
rng = np.random.default_rng(0)
def run(b, n, m, q, algo='auto'):
    T = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    z, mu = T(rng.standard_normal((m, q))), T(rng.standard_normal((n, q)))
    s = T(np.exp(0.3 * rng.standard_normal((n, q))))
    g, al = T(np.exp(0.3 * rng.standard_normal((b, q)))), T(np.ones((b, 1)))
    for _ in range(3): ops.psi2(z, mu, s, g, al, algo=algo)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.psi2(z, mu, s, g, al, algo=algo)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3
for algo in ('auto', 'patch_f16'):
    for b, n in [(512, 500), (512, 1000), (512, 2000), (512, 4000), (512, 8000), (64, 2000), (128, 2000), (256, 2000), (1024, 2000)]:
        ms = run(b, n, 128, 10, algo)
        ex = b * n * 128 * 129 / 2
        print('%s B %4d N %5d: %.3f ms  (%.2f Texp/s)' % (algo, b, n, ms, ex / ms / 1e9), flush=True)
