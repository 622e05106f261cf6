"""CPU emulation of the f16 hi/lo split exponent GEMM for psi2: error of the exponent (log2 units) vs fp64."""
import numpy as np, sys
sys.path.insert(0, '.')
from dp_gp_lvm_amd.utils.synthetic import make_problem
LOG2E = 1.4426950408889634
def split2(x):
    h = x.astype(np.float16)
    l = (x - h.astype(np.float32)).astype(np.float16)
    return h, l
def split3(x):
    h = x.astype(np.float16); r = x - h.astype(np.float32)
    m = r.astype(np.float16); l = (r - m.astype(np.float32)).astype(np.float16)
    return h, m, l
for cfg in (2, 5):
    p = make_problem(cfg); n, d, m_, q = p['y'].shape[0], p['y'].shape[1], p['z'].shape[0], p['z'].shape[1]
    rng = np.random.default_rng(0)
    for dd in (0, d // 2):
        g = p['gamma'][dd]
        zc = p['z'].mean(axis=0); z = p['z'] - zc; mu = p['mu'] - zc; s = p['s']
        den = 2 * g * s + 1; w = g / den
        X = -0.5 * LOG2E * w                       # [N,Q]
        P = LOG2E * np.sum(0.5 * w * mu**2 - 0.25 * np.log(den), axis=1)[:, None] - LOG2E * 0.25 * np.einsum('nq,nmq->nm', w, (z[None] - 2 * mu[:, None]) ** 2)
        ns = rng.integers(0, n, 4000); ms = rng.integers(0, m_, 4000); mps = rng.integers(0, m_, 4000)
        exact = P[ns, ms] + P[ns, mps] + np.einsum('kq,kq,kq->k', X[ns], z[ms], z[mps])
        # f32 baseline (what the current kernel does)
        X32, z32, P32 = X.astype(np.float32), z.astype(np.float32), P.astype(np.float32)
        f32 = (P32[ns, ms] + P32[ns, mps]).astype(np.float32)
        A32 = (X32[ns] * z32[ms]).astype(np.float32)
        acc = np.zeros(len(ns), np.float32)
        for k in range(q): acc = (acc + A32[:, k] * z32[mps][:, k]).astype(np.float32)
        f32 = (acc + P32[ns, ms] + P32[ns, mps]).astype(np.float32)
        # f16 split: A = X*z (f32) split 2-way, B = z split 2-way, P split 3-way; f32 accumulate
        Ah, Al = split2(A32); Bh, Bl = split2(z32[mps])
        acc = np.zeros(len(ns), np.float32)
        for k in range(q):
            for a_, b_ in ((Ah, Bh), (Ah, Bl), (Al, Bh)):
                acc = (acc + a_[:, k].astype(np.float32) * b_[:, k].astype(np.float32)).astype(np.float32)
        for parts in (split3(P32[ns, ms]), split3(P32[ns, mps])):
            for pp in parts: acc = (acc + pp.astype(np.float32)).astype(np.float32)
        # 2-way split of P for comparison
        acc2 = np.zeros(len(ns), np.float32)
        for k in range(q):
            for a_, b_ in ((Ah, Bh), (Ah, Bl), (Al, Bh)):
                acc2 = (acc2 + a_[:, k].astype(np.float32) * b_[:, k].astype(np.float32)).astype(np.float32)
        for parts in (split2(P32[ns, ms]), split2(P32[ns, mps])):
            for pp in parts: acc2 = (acc2 + pp.astype(np.float32)).astype(np.float32)
        sel = exact > -60          # terms that matter (2^-60 relative to O(1))
        print('cfg', cfg, 'd', dd, 'exponent range', exact.min().round(1), exact.max().round(1), 'significant', sel.sum(),
              '| abs err f32 max %.2e rms %.2e' % (np.abs(f32 - exact)[sel].max(), np.sqrt(np.mean((f32 - exact)[sel] ** 2))),
              '| f16 split (P 3-way) max %.2e rms %.2e' % (np.abs(acc - exact)[sel].max(), np.sqrt(np.mean((acc - exact)[sel] ** 2))),
              '| (P 2-way) max %.2e rms %.2e' % (np.abs(acc2 - exact)[sel].max(), np.sqrt(np.mean((acc2 - exact)[sel] ** 2))),
              '| max |P| %.1f' % np.abs(P).max())
