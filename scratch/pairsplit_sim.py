"""CPU emulation of the pair-tile psi2 exponent GEMM (psi2_pairs.hip) with fewer f16 products per term: error of the psi2 ENTRIES
(after the sum over the observations) against fp64.  Variants: 3 = a_h f_h + a_h f_l + a_l f_h (the kernel today);
2r = drop a_l f_h (row-side residual: varies with n);  2p = drop a_h f_l (pair-side residual: the same for every n)."""
import numpy as np, sys
sys.path.insert(0, '.')
from dp_gp_lvm_amd.utils.synthetic import make_problem
LOG2E = 1.4426950408889634
f16 = lambda x: x.astype(np.float16).astype(np.float32)
def split(x):
    x = x.astype(np.float32); h = f16(x); return h, f16(x - h)
def run(p, dd, npairs, rng, label):
    z = p['z'] - p['z'].mean(axis=0); mu = p['mu'] - p['z'].mean(axis=0); s = p['s']; g = p['gamma'][dd]
    n, q = mu.shape; m = z.shape[0]
    den = 2 * g * s + 1; w = g / den
    a = -0.25 * w * LOG2E; b = w * mu * LOG2E; c = -np.sum(w * mu**2 * LOG2E + 0.5 * np.log2(den), axis=1)
    ms = rng.integers(0, m, npairs); mps = rng.integers(0, m, npairs)
    sp = z[ms] + z[mps]                                           # [P,Q]
    E = a @ (sp**2).T + b @ sp.T + c[:, None]                     # [N,P] exact
    ref = np.exp2(E).sum(axis=0)
    ah, al = split(64 * a); bh, bl = split(b); ch, cl = split(c)
    f1h, f1l = split(sp**2 / 64); f2h, f2l = split(sp)
    def gemm(x, y): return (x.astype(np.float32) @ y.astype(np.float32).T)      # fp32 accumulate (order differs: fine)
    base = gemm(ah, f1h) + gemm(bh, f2h) + ch[:, None] + cl[:, None]
    v3 = base + gemm(ah, f1l) + gemm(al, f1h) + gemm(bh, f2l) + gemm(bl, f2h)
    v2r = base + gemm(ah, f1l) + gemm(bh, f2l)
    v2p = base + gemm(al, f1h) + gemm(bl, f2h)
    v2r_quad = base + gemm(ah, f1l) + gemm(bh, f2l) + gemm(bl, f2h)              # drop a_l f_h on the quadratic features only
    out = []
    for name, v in (('3', v3), ('2r', v2r), ('2p', v2p), ('2r-quad', v2r_quad)):
        got = np.exp2(v.astype(np.float64)).sum(axis=0)
        rel = np.abs(got - ref) / ref.max()
        out.append('%s: max %.1e rms %.1e' % (name, rel.max(), np.sqrt(np.mean(rel**2))))
    print(label, 'd', dd, '| entry error / max entry |', ' | '.join(out))
rng = np.random.default_rng(0)
for cfg in (4, 5):
    p = make_problem(cfg)
    for dd in (0, p['y'].shape[1] // 2): run(p, dd, 300, rng, 'config %d' % cfg)
for shape in ((100, 3, 30, 20), (50, 3, 25, 17), (300, 3, 40, 30)):
    p = make_problem(shape=shape, seed=3)
    run(p, 0, 300, rng, 'shape %s' % (shape,))
