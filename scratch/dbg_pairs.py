"""scratch: accuracy of the pair-tile psi2 kernel on assorted shapes against the oracle (and the patch kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dp_gp_lvm_amd import ops
from oracle import dpgp_oracle as orc
dev = torch.device('cuda', 0)
T = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a), dtype=dt, device=dev)
for shape in [(3, 37, 5, 1), (2, 300, 33, 3), (5, 257, 64, 10), (2, 500, 100, 20), (1, 130, 130, 30), (3, 1000, 128, 10), (2, 2000, 128, 10), (1, 700, 200, 12)]:
    b, n, m, q = shape
    rng = np.random.default_rng(sum(shape))
    z, mu = rng.standard_normal((m, q)), rng.standard_normal((n, q))
    s = np.exp(0.5 * rng.standard_normal((n, q)))
    gam, al = np.exp(0.3 * rng.standard_normal((b, q))), np.exp(0.3 * rng.standard_normal((b, 1)))
    ref = orc.psi2(z, mu, s, gam, al)
    args = [T(a) for a in (z, mu, s, gam, al)]
    for algo in ('auto', 'patch_f16', 'mfma_f32'):
        got = ops.psi2(*args, algo=algo).double().cpu().numpy()
        err = np.abs(got - ref)
        rel = err / np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
        i = np.unravel_index(np.nanargmax(rel), rel.shape)
        print(shape, algo, 'max rel err %.2e (at %s: got %.6e ref %.6e)  nan %d  max|ref| %.3e' % (np.nanmax(rel), i, got[i], ref[i], int(np.isnan(got).sum()), np.abs(ref).max()), flush=True)
