"""
Synthetic DP-GP-LVM problems of the BASELINE.json shapes (SURVEY.md §8d recipe).  NumPy only, fp64.

There is no network for the reference's data sets, so the benchmark, the parity tests and the golden-vector
generator all draw their inputs from here: post-initialisation parameter VALUES of the same shapes and
distributions that ``dp_gp_lvm(...)`` construction yields in the reference (src/models/dp_gp_lvm.py:62-102,
src/models/dirichlet_process.py:39-59), perturbed so that the per-output hyper-parameters differ.
"""

import numpy as np

# (N, D, M, Q) of BASELINE.json `configs`; index 0 is the reference's own CPU-runnable plumbing shape.
CONFIGS = {
    1: (100, 12, 20, 4),
    2: (2000, 64, 128, 10),
    3: (2000, 512, 128, 10),
    4: (10000, 256, 512, 20),
    5: (1965, 560, 64, 20),
    # the shape test/synthetic_data_test.py really uses (SURVEY.md §0 discrepancy 1)
    6: (100, 20, 25, 10),
}
DEFAULT_TRUNCATION = 8          # src/utils/constants.py:121


def _softplus(x):
    return np.logaddexp(0.0, x)


def make_problem(cfg=None, shape=None, truncation_level=DEFAULT_TRUNCATION, seed=None, d_slice=None):
    """
    Draw one problem.  ``cfg`` selects CONFIGS[cfg] (seed 1000+cfg); or pass ``shape=(N,D,M,Q)`` and ``seed``.
    Returns a dict of fp64 arrays:
      y [N,D] column-standardised, mu [N,Q], s [N,Q] (diagonal of q(X) covariance), z [M,Q], phi [D,T],
      gamma_atoms [T,Q], alpha_atoms [T,1], beta_atoms [T,1], g1,g2 [T-1], w1,w2 (scalars), s1,s2 (prior),
      and the mixed per-output gamma [D,Q], alpha [D,1], beta [D,1].
    ``d_slice`` (a slice or index array) keeps only those output dims of y/phi/gamma/alpha/beta — the draw order
    is unchanged, so rank r of a D-sharded run sees exactly columns d_slice of the single-GPU problem.
    """
    if shape is None:
        shape = CONFIGS[cfg]
        seed = 1000 + cfg if seed is None else seed
    n, d, m, q = shape
    t = min(truncation_level, d, n)
    rng = np.random.default_rng(seed)
    y = rng.standard_normal((n, d))
    y = (y - y.mean(axis=0)) / y.std(axis=0)                       # dp_gp_lvm.py:30-32 assumes this
    mu = rng.standard_normal((n, q))
    mu = mu / np.mean(mu.std(axis=0, ddof=1))                      # what the PCA init yields (utils/expressions.py:73)
    s = 1.0 * np.exp(0.25 * rng.standard_normal((n, q)))          # init 1.0 (dp_gp_lvm.py:67), perturbed
    z = mu[rng.permutation(n)[:m]] + 0.01 * rng.standard_normal((m, q))       # dp_gp_lvm.py:72-73
    logits = rng.standard_normal((d, t))
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    phi = e / e.sum(axis=1, keepdims=True)                         # dirichlet_process.py:40-42
    gamma_atoms = np.exp(0.3 * rng.standard_normal((t, q)))       # init 1.0 (constants.py:97-99), perturbed
    alpha_atoms = np.exp(0.3 * rng.standard_normal((t, 1)))
    beta_atoms = np.exp(0.3 * rng.standard_normal((t, 1)))
    g1 = _softplus(rng.standard_normal(max(t - 1, 0)))            # dirichlet_process.py:54-55
    g2 = _softplus(rng.standard_normal(max(t - 1, 0)))
    p = dict(y=y, mu=mu, s=s, z=z, phi=phi, gamma_atoms=gamma_atoms, alpha_atoms=alpha_atoms,
             beta_atoms=beta_atoms, g1=g1, g2=g2, w1=1.0, w2=1.0, s1=1.0, s2=1.0,
             gamma=phi @ gamma_atoms, alpha=phi @ alpha_atoms, beta=phi @ beta_atoms)
    if d_slice is not None:
        for k in ('phi', 'gamma', 'alpha', 'beta'):
            p[k] = np.ascontiguousarray(p[k][d_slice])
        p['y'] = np.ascontiguousarray(p['y'][:, d_slice])
    return p
