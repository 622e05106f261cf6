"""
TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT PATH.

CPU restatement (NumPy, fp64) of the DP-GP-LVM variational ELBO inner loop of AndrewRLawrence/dp_gp_lvm.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this module,
and there only as the checker.  ``dp_gp_lvm_amd`` never imports it and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``oracle/gen_golden.py`` (container-only) executes the reference's own, unmodified source
(`src/kernels/rbf_kernel.py`, `src/models/dp_gp_lvm.py`, `src/models/dirichlet_process.py`, ...) and the reference's
own pure-NumPy known-answer functions (`test/unittests/kernel_unittests.py:14-147`,
`bgplvm_unittests.py:17-52,123-135`, `dp_unittests.py:13-131`) and checks every function below against both before
it writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks this module against those fixtures.
Caveat: TensorFlow 1.15 is not installable here, so the reference source ran on an eager NumPy stand-in for the
``tf.*`` ops (``oracle/standin``) — LAPACK instead of Eigen for cholesky / triangular solves.

Every function cites the reference lines (relative to /root/reference) that it restates.  Shapes: N observations,
D observed dims (= kernel batch B), M inducing points, Q latent dims, T truncation level.
``s`` is always the DIAGONAL of q(X)'s covariance, shape [N, Q] (the reference passes [N,Q,Q] and only ever takes
its diagonal: rbf_kernel.py:151,182; gp_expressions.py:20).
"""

import numpy as np
import scipy.linalg as sla
import scipy.special as sp

GP_DEFAULT_JITTER = 1.0e-8   # src/utils/constants.py:96
LOG_2PI = float(np.log(2.0 * np.pi))


def _hyp(gamma, alpha, beta=None):
    gamma = np.atleast_2d(np.asarray(gamma, dtype=np.float64))          # [B,Q]
    alpha = np.asarray(alpha, dtype=np.float64).reshape(-1)              # [B]
    beta = None if beta is None else np.asarray(beta, dtype=np.float64).reshape(-1)
    return gamma, alpha, beta


# ---------------------------------------------------------------------------------------------------------------
# Kernel operators: src/kernels/rbf_kernel.py
# ---------------------------------------------------------------------------------------------------------------

def ard_rbf_gram(x0, x1, gamma, alpha, beta, include_noise=False, include_jitter=False, jitter=GP_DEFAULT_JITTER):
    """rbf_kernel.py:58-93.  K[b,i,j] = alpha_b exp(-1/2 sum_q gamma_bq (x0_iq - x1_jq)^2); noise beta^-1 I and
    jitter I are added ONLY when x1 is None (rbf_kernel.py:80,86), even if the caller passes the same array twice."""
    gamma, alpha, beta = _hyp(gamma, alpha, beta)
    x0 = np.asarray(x0, dtype=np.float64)
    xa = np.sqrt(gamma)[:, None, :] * x0[None]                                    # [B,N0,Q]  (:70-71)
    xb = xa if x1 is None else np.sqrt(gamma)[:, None, :] * np.asarray(x1, dtype=np.float64)[None]
    aa = -0.5 * np.sum(xa * xa, axis=-1)                                           # (:74)
    bb = -0.5 * np.sum(xb * xb, axis=-1)                                           # (:75)
    k = alpha[:, None, None] * np.exp(aa[:, :, None] + bb[:, None, :] + xa @ np.swapaxes(xb, 1, 2))   # (:77-78)
    if x1 is None:
        eye = np.eye(x0.shape[0])
        if include_noise:
            k = k + (1.0 / beta)[:, None, None] * eye                              # (:80-85)
        if include_jitter:
            k = k + jitter * eye                                                   # (:86-91)
    return k


def ard_rbf_diag(n, alpha, beta, include_noise=False, include_jitter=False, jitter=GP_DEFAULT_JITTER):
    """rbf_kernel.py:96-116 -> [B,N]; the values of input_0 are never read, only its row count."""
    _, alpha, beta = _hyp(np.zeros((1, 1)), alpha, beta)
    k = alpha[:, None] * np.ones((1, int(n)))
    if include_noise:
        k = k + (1.0 / beta)[:, None]
    if include_jitter:
        k = k + jitter
    return k


def psi0(n, alpha):
    """rbf_kernel.py:119-132 -> [B,1] = alpha * N."""
    return np.asarray(alpha, dtype=np.float64).reshape(-1, 1) * float(n)


def psi1(z, mu, s, gamma, alpha):
    """rbf_kernel.py:135-161 -> [B,N,M].
    log psi1[b,n,m] = log alpha_b - 1/2 sum_q ( gamma_bq (mu_nq - z_mq)^2 / (gamma_bq s_nq + 1) + log(gamma_bq s_nq + 1) )."""
    gamma, alpha, _ = _hyp(gamma, alpha)
    z, mu, s = (np.asarray(a, dtype=np.float64) for a in (z, mu, s))
    out = np.empty((gamma.shape[0], mu.shape[0], z.shape[0]))
    sqd = np.square(mu[:, None, :] - z[None, :, :])                                # [N,M,Q]   (:156)
    for b in range(gamma.shape[0]):
        den = gamma[b][None, :] * s + 1.0                                          # [N,Q]     (:155)
        e = np.einsum('nmq,nq->nm', sqd, gamma[b][None, :] / den) + np.sum(np.log(den), axis=-1)[:, None]
        out[b] = np.exp(np.log(alpha[b]) - 0.5 * e)                                # (:158-161)
    return out


def psi2(z, mu, s, gamma, alpha, chunk=64):
    """rbf_kernel.py:164-199 -> [B,M,M], the literal formula, streamed over n-chunks (the reference materialises
    [B,N,M,M,Q]; the sum over n at :199 is additive so chunking changes nothing but memory).
    log psi2[b,n,m,m'] = 2 log alpha_b - sum_q ( 1/2 log(2 gamma s + 1) + gamma (z_m - z_m')^2 / 4
                                               + gamma (mu_n - (z_m+z_m')/2)^2 / (2 gamma s + 1) )."""
    gamma, alpha, _ = _hyp(gamma, alpha)
    z, mu, s = (np.asarray(a, dtype=np.float64) for a in (z, mu, s))
    b_, n_, m_ = gamma.shape[0], mu.shape[0], z.shape[0]
    zbar = 0.5 * (z[:, None, :] + z[None, :, :])                                   # [M,M,Q]   (:189)
    zdif2 = np.square(z[:, None, :] - z[None, :, :])                               # [M,M,Q]   (:191)
    out = np.zeros((b_, m_, m_))
    for b in range(b_):
        t1 = 0.25 * np.einsum('q,ijq->ij', gamma[b], zdif2)                        # (:191)
        for n0 in range(0, n_, chunk):
            den = 2.0 * gamma[b][None, :] * s[n0:n0 + chunk] + 1.0                 # [c,Q]     (:193)
            num = np.square(mu[n0:n0 + chunk, None, None, :] - zbar[None])         # [c,M,M,Q] (:194)
            e = np.einsum('cijq,cq->cij', num, gamma[b][None, :] / den)
            lp = 2.0 * np.log(alpha[b]) - (0.5 * np.sum(np.log(den), axis=-1)[:, None, None] + t1[None] + e)
            out[b] += np.sum(np.exp(lp), axis=0)                                   # (:196-199)
    return out


def psi1T_y(z, mu, s, gamma, alpha, y):
    """Psi1_d^T y_d -> [D,M]: the only way Psi1 enters the objective (dp_gp_lvm.py:132-145, via c = L_A^-1 L^-1 Psi1^T)."""
    p1 = psi1(z, mu, s, gamma, alpha)                                              # [D,N,M]
    return np.einsum('dnm,nd->dm', p1, np.asarray(y, dtype=np.float64))


# ---------------------------------------------------------------------------------------------------------------
# Log-density terms: src/models/expressions/gp_expressions.py, src/distributions/*.py
# ---------------------------------------------------------------------------------------------------------------

def kl_qx(mu, s):
    """gp_expressions.py:10-24: KL(q(X) || N(0,I)) = 1/2 ( sum mu^2 + sum (s - log s) - N Q )."""
    mu, s = np.asarray(mu, dtype=np.float64), np.asarray(s, dtype=np.float64)
    return 0.5 * (np.sum(mu * mu) + np.sum(s - np.log(s)) - mu.shape[0] * mu.shape[1])


def log_normal_log_pdf(x):
    """log_normal.py:24-39 with mean 0, var 1: -log x - 1/2 (log 2 pi + log^2 x)."""
    lx = np.log(np.asarray(x, dtype=np.float64))
    return -lx - 0.5 * (LOG_2PI + lx * lx)


def hyperprior(gamma_atoms, alpha_atoms, beta_atoms):
    """dp_gp_lvm.py:96-98: sum of log-normal log-pdfs over the DP ATOMS (not the mixed per-d values)."""
    return float(sum(np.sum(log_normal_log_pdf(a)) for a in (gamma_atoms, alpha_atoms, beta_atoms)))


def beta_entropy(a, b):
    """distributions/beta.py:8-19."""
    t = a + b
    return sp.gammaln(a) + sp.gammaln(b) - sp.gammaln(t) - (a - 1.0) * sp.digamma(a) - (b - 1.0) * sp.digamma(b) \
        + (t - 2.0) * sp.digamma(t)


def gamma_entropy(a, b):
    """distributions/gamma.py:8-17 (shape a, rate b)."""
    return a - np.log(b) + sp.gammaln(a) + (1.0 - a) * sp.digamma(a)


def multinomial_entropy(p):
    """distributions/multinomial.py:8-16."""
    return -np.sum(p * np.log(p), axis=-1)


def dp_objective(phi, g1, g2, w1, w2, s1, s2):
    """dirichlet_process.py:64-88: negative ELBO of the truncated stick-breaking DP.
    phi [D,T]; g1,g2 [T-1] (q(V) Beta params); w1,w2 scalars (q(alpha) Gamma params); s1,s2 prior Gamma params."""
    phi = np.asarray(phi, dtype=np.float64)
    g1 = np.asarray(g1, dtype=np.float64).reshape(-1)
    g2 = np.asarray(g2, dtype=np.float64).reshape(-1)
    w1, w2, s1, s2 = float(w1), float(w2), float(s1), float(s2)
    t = phi.shape[1]
    dg12 = sp.digamma(g1 + g2)
    tail = (np.cumsum(phi[:, ::-1], axis=1)[:, ::-1] - phi)[:, :-1]               # exclusive reverse cumsum (:65)
    ev_z = np.sum(phi[:, :-1] * (sp.digamma(g1) - dg12) + tail * (sp.digamma(g2) - dg12))            # (:64-66)
    ev_v = (t - 1.0) * (sp.digamma(w1) - np.log(w2)) + (w1 / w2 - 1.0) * np.sum(sp.digamma(g2) - dg12)   # (:68-69)
    ev_a = s1 * np.log(s2) - sp.gammaln(s1) + (s1 - 1.0) * (sp.digamma(w1) - np.log(w2)) - s2 * (w1 / w2)  # (:71-72)
    ent = np.sum(multinomial_entropy(phi)) + np.sum(beta_entropy(g1, g2)) + gamma_entropy(w1, w2)    # (:75-77)
    return float(-(ev_z + ev_v + ev_a + ent))                                       # (:80-88)


# ---------------------------------------------------------------------------------------------------------------
# Per-output ELBO reduction: src/models/dp_gp_lvm.py:100-154
# ---------------------------------------------------------------------------------------------------------------

def mix_hyperparameters(phi, gamma_atoms, alpha_atoms, beta_atoms):
    """dp_gp_lvm.py:100-102: gamma = phi gamma_at [D,Q], alpha = phi alpha_at [D,1], beta = phi beta_at [D,1]."""
    phi = np.asarray(phi, dtype=np.float64)
    return phi @ np.asarray(gamma_atoms, dtype=np.float64), \
        phi @ np.asarray(alpha_atoms, dtype=np.float64).reshape(-1, 1), \
        phi @ np.asarray(beta_atoms, dtype=np.float64).reshape(-1, 1)


def fhat_terms(y, z, mu, s, gamma, alpha, beta, jitter=GP_DEFAULT_JITTER, psi2_fn=None, return_parts=False):
    """dp_gp_lvm.py:108-145, one row of five terms per output dim d (f_hat = terms.sum()):
        t0 = 1/2 N (log beta_d - log 2 pi)                (:138)
        t1 = - sum_i log (L_A,d)_ii                       (:129,139)
        t2 = 1/2 beta_d ( tr(L^-1 Psi2 L^-T) - alpha_d N ) (:140-142)
        t3 = - 1/2 beta_d y_d^T y_d                       (:143-144)
        t4 = 1/2 beta_d^2 || L_A^-1 L^-1 Psi1^T y_d ||^2    (:132-136,145; = 1/2 (beta y_d)^T c^T c (beta y_d))
    The reference forms c [D,M,N] and c^T c [D,N,N]; only c y_d is needed, so only the M-vector Psi1^T y_d is solved."""
    gamma, alpha, beta = _hyp(gamma, alpha, beta)
    y, z, mu, s = (np.asarray(a, dtype=np.float64) for a in (y, z, mu, s))
    n_, d_ = y.shape
    m_ = z.shape[0]
    kuu = ard_rbf_gram(z, None, gamma, alpha, beta, include_noise=False, include_jitter=True, jitter=jitter)  # (:115)
    p2 = (psi2_fn or psi2)(z, mu, s, gamma, alpha)                                 # (:110)
    v = psi1T_y(z, mu, s, gamma, alpha, y)                                         # [D,M]
    terms = np.empty((d_, 5))
    l_uu = np.empty((d_, m_, m_))
    l_a = np.empty((d_, m_, m_))
    t2m = np.empty((d_, m_, m_))
    for d in range(d_):
        l = np.linalg.cholesky(kuu[d])                                             # (:116)
        x = sla.solve_triangular(l, p2[d], lower=True)                             # (:118)
        t2 = sla.solve_triangular(l, x.T, lower=True).T                            # (:119-121)
        a = beta[d] * t2 + np.eye(m_)                                              # (:124-126)
        la = np.linalg.cholesky(a)                                                 # (:127)
        cy = sla.solve_triangular(la, sla.solve_triangular(l, v[d], lower=True), lower=True)   # (:132-133) applied to y_d
        terms[d, 0] = 0.5 * n_ * (np.log(beta[d]) - LOG_2PI)
        terms[d, 1] = -np.sum(np.log(np.diag(la)))
        terms[d, 2] = 0.5 * beta[d] * (np.trace(t2) - alpha[d] * n_)
        terms[d, 3] = -0.5 * beta[d] * np.dot(y[:, d], y[:, d])
        terms[d, 4] = 0.5 * beta[d] ** 2 * np.dot(cy, cy)
        l_uu[d], l_a[d], t2m[d] = l, la, t2
    if return_parts:
        return terms, dict(k_uu=kuu, psi_2=p2, psi1T_y=v, l_uu=l_uu, l_a=l_a, t2=t2m)
    return terms


def objective(y, z, mu, s, phi, gamma_atoms, alpha_atoms, beta_atoms, g1, g2, w1, w2, s1, s2,
              jitter=GP_DEFAULT_JITTER, psi2_fn=None):
    """dp_gp_lvm.py:148-154: objective = DP objective - (f_hat - KL) - hyper-prior log-likelihood."""
    gamma, alpha, beta = mix_hyperparameters(phi, gamma_atoms, alpha_atoms, beta_atoms)
    f_hat = fhat_terms(y, z, mu, s, gamma, alpha, beta, jitter=jitter, psi2_fn=psi2_fn).sum()
    return dp_objective(phi, g1, g2, w1, w2, s1, s2) - (f_hat - kl_qx(mu, s)) \
        - hyperprior(gamma_atoms, alpha_atoms, beta_atoms)
