"""
TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT PATH.

PyTorch (fp64, CPU, autograd) restatement of the DP-GP-LVM objective as a function of the reference's eleven RAW trainable
variables, in the restructured algebra of the HIP path (B = K_uu + beta Psi2 form, Psi1 only as
Psi1^T y).  Its purpose is the gradient oracle for the backward pass of the fused ELBO: the
reference trains on tf.gradients(objective) (test/synthetic_data_hard_test.py:143-155).

Parity status: PINNED.  ``oracle/gen_golden_grad.py`` (container-only) runs the reference's own, unmodified
``dp_gp_lvm(...)`` constructor under an eager PyTorch stand-in for TensorFlow, differentiates the objective it builds and
checks this module's objective (rtol 1e-11) and all eleven gradients (rtol 1e-7) against it before writing
``tests/golden/grad_ref_*.npz``; ``tests/test_oracle_grad.py`` re-checks this module against those fixtures.
Only ``tests/`` may import it.

Raw variables (creation order in the reference, dp_gp_lvm.py:63-94, dirichlet_process.py:40-59, utils/types.py:40-72):
    x_mean [N,Q]; x_var_raw [N,Q] (S = softplus); x_u [M,Q]; dp_logits [D/mask,T] (phi = softmax, rows repeated mask_size
    times); gamma_1_raw, gamma_2_raw [T-1] (softplus); w_1_raw, w_2_raw (softplus); gamma_atoms_raw [T,Q], alpha_atoms_raw
    [T,1], beta_atoms_raw [T,1] (softplus).
"""

import numpy as np
import torch

NAMES = ['x_mean', 'x_var_raw', 'x_u', 'dp_logits', 'gamma_1_raw', 'gamma_2_raw', 'w_1_raw', 'w_2_raw',
         'gamma_atoms_raw', 'alpha_atoms_raw', 'beta_atoms_raw']
GP_DEFAULT_JITTER = 1.0e-8   # src/utils/constants.py:96
LOG_2PI = float(np.log(2.0 * np.pi))


def _softplus(x):
    return torch.nn.functional.softplus(x, beta=1.0, threshold=1.0e6)     # utils/types.py:57,70


def _log_normal_log_pdf(x):                                              # distributions/log_normal.py:24-39
    lx = torch.log(x)
    return -lx - 0.5 * (LOG_2PI + lx * lx)


def _beta_entropy(a, b):                                                 # distributions/beta.py:8-19
    t = a + b
    return torch.lgamma(a) + torch.lgamma(b) - torch.lgamma(t) - (a - 1.0) * torch.digamma(a) \
        - (b - 1.0) * torch.digamma(b) + (t - 2.0) * torch.digamma(t)


def _gamma_entropy(a, b):                                                # distributions/gamma.py:8-17
    return a - torch.log(b) + torch.lgamma(a) + (1.0 - a) * torch.digamma(a)


def dp_objective(phi, g1, g2, w1, w2, s1, s2):
    """dirichlet_process.py:64-88 (see oracle/dpgp_oracle.py:dp_objective for the line-by-line map)."""
    t = phi.shape[1]
    dg12 = torch.digamma(g1 + g2)
    tail = (torch.flip(torch.cumsum(torch.flip(phi, dims=(1,)), dim=1), dims=(1,)) - phi)[:, :-1]
    ev_z = torch.sum(phi[:, :-1] * (torch.digamma(g1) - dg12) + tail * (torch.digamma(g2) - dg12))
    ev_v = (t - 1.0) * (torch.digamma(w1) - torch.log(w2)) + (w1 / w2 - 1.0) * torch.sum(torch.digamma(g2) - dg12)
    ev_a = s1 * np.log(s2) - float(torch.lgamma(torch.tensor(float(s1), dtype=torch.float64))) \
        + (s1 - 1.0) * (torch.digamma(w1) - torch.log(w2)) - s2 * (w1 / w2)
    ent = torch.sum(-torch.sum(phi * torch.log(phi), dim=-1)) + torch.sum(_beta_entropy(g1, g2)) + _gamma_entropy(w1, w2)
    return -(ev_z + ev_v + ev_a + ent)


def psi_pieces(y, z, mu, s, gamma, alpha, jitter=GP_DEFAULT_JITTER, chunk=64):
    """K_uu + jitter I [D,M,M] (rbf_kernel.py:58-93), Psi2 [D,M,M] (rbf_kernel.py:164-199) and Psi1^T y [D,M]
    (rbf_kernel.py:135-161 contracted with y), the Psi statistics streamed over n-chunks."""
    n, d = y.shape
    m = z.shape[0]
    zd = z[:, None, :] - z[None, :, :]                                                       # [M,M,Q]
    zbar = 0.5 * (z[:, None, :] + z[None, :, :])
    sq = torch.einsum('dq,ijq->dij', gamma, zd * zd)
    k_uu = alpha[:, None, None] * torch.exp(-0.5 * sq) + jitter * torch.eye(m, dtype=y.dtype)
    v = torch.zeros((d, m), dtype=y.dtype)
    p2 = torch.zeros((d, m, m), dtype=y.dtype)
    t1 = 0.25 * sq
    for n0 in range(0, n, chunk):
        mu_c, s_c, y_c = mu[n0:n0 + chunk], s[n0:n0 + chunk], y[n0:n0 + chunk]
        den1 = gamma[:, None, :] * s_c[None] + 1.0                                           # [D,c,Q]
        e1 = torch.einsum('cmq,dcq->dcm', (mu_c[:, None, :] - z[None]) ** 2, gamma[:, None, :] / den1) \
            + torch.sum(torch.log(den1), dim=-1)[:, :, None]
        p1 = alpha[:, None, None] * torch.exp(-0.5 * e1)                                     # [D,c,M]
        v = v + torch.einsum('dcm,cd->dm', p1, y_c)
        den2 = 2.0 * gamma[:, None, :] * s_c[None] + 1.0                                     # [D,c,Q]
        num = (mu_c[:, None, None, :] - zbar[None]) ** 2                                     # [c,M,M,Q]
        e2 = torch.einsum('cijq,dcq->dcij', num, gamma[:, None, :] / den2)
        lp = 2.0 * torch.log(alpha)[:, None, None, None] - (0.5 * torch.sum(torch.log(den2), dim=-1)[:, :, None, None]
                                                             + t1[:, None] + e2)
        p2 = p2 + torch.sum(torch.exp(lp), dim=1)
    return k_uu, p2, v


def fhat_from_pieces(k_uu, p2, v, alpha, beta, yy, n):
    """f_hat of dp_gp_lvm.py:108-145 in the B = K_uu + beta Psi2 form, five terms per output dim:
        1/2 N (log beta - log 2 pi), -(log|L_B| - log|L_K|), 1/2 beta (<K^-1, Psi2> - alpha N), -1/2 beta y^T y,
        1/2 beta^2 || L_B^-1 Psi1^T y ||^2."""
    l_k = torch.linalg.cholesky(k_uu)
    b = k_uu + beta[:, None, None] * p2
    l_b = torch.linalg.cholesky(b)
    logdet_k = torch.sum(torch.log(torch.diagonal(l_k, dim1=-2, dim2=-1)), dim=-1)
    logdet_b = torch.sum(torch.log(torch.diagonal(l_b, dim1=-2, dim2=-1)), dim=-1)
    tr = torch.sum(torch.cholesky_solve(p2, l_k).diagonal(dim1=-2, dim2=-1), dim=-1)         # <K^-1, Psi2>
    c = torch.linalg.solve_triangular(l_b, v[:, :, None], upper=False)[:, :, 0]
    return torch.stack([0.5 * n * (torch.log(beta) - LOG_2PI), -(logdet_b - logdet_k), 0.5 * beta * (tr - alpha * n),
                        -0.5 * beta * yy, 0.5 * beta * beta * torch.sum(c * c, dim=-1)], dim=1)


def fhat(y, z, mu, s, gamma, alpha, beta, jitter=GP_DEFAULT_JITTER, chunk=64):
    k_uu, p2, v = psi_pieces(y, z, mu, s, gamma, alpha, jitter=jitter, chunk=chunk)
    return fhat_from_pieces(k_uu, p2, v, alpha, beta, torch.sum(y * y, dim=0), y.shape[0])


def chain_adjoints(y, z, mu, s, gamma, alpha, beta, jitter=GP_DEFAULT_JITTER):
    """What stage A of the HIP backward pass (dpgp_elbo_grad_chain) must return, by autograd (NumPy in / out):
    g_psi2, g_kuu (symmetrised d f_hat / d Psi2, d K_uu), g_v, and the COMPLETE d f_hat / d alpha_d, d beta_d."""
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64))
    y, z, mu, s, gamma = t(y), t(z), t(mu), t(s), t(gamma)
    al = t(alpha).reshape(-1).clone().requires_grad_(True)
    be = t(beta).reshape(-1).clone().requires_grad_(True)
    k_uu, p2, v = psi_pieces(y, z, mu, s, gamma, al, jitter=jitter)
    kl, pl, vl = (a.detach().clone().requires_grad_(True) for a in (k_uu, p2, v))
    f = torch.sum(fhat_from_pieces(kl, pl, vl, al.detach(), be.detach(), torch.sum(y * y, dim=0), y.shape[0]))
    gk, gp, gv = torch.autograd.grad(f, [kl, pl, vl])
    total = torch.sum(fhat_from_pieces(k_uu, p2, v, al, be, torch.sum(y * y, dim=0), y.shape[0]))
    da, db = torch.autograd.grad(total, [al, be])
    sym = lambda a: 0.5 * (a + a.transpose(1, 2))
    return dict(g_kuu=sym(gk).numpy(), g_psi2=sym(gp).numpy(), g_v=gv.numpy(), d_alpha=da.numpy(), d_beta=db.numpy(),
                k_uu=k_uu.detach().numpy(), psi2=p2.detach().numpy(), v=v.detach().numpy())


def fhat_input_gradients(y, z, mu, s, gamma, alpha, beta, jitter=GP_DEFAULT_JITTER):
    """d f_hat / d (mu, s, z, gamma, alpha, beta) by autograd (NumPy in / out): what stages A + B of the HIP backward pass
    must deliver together."""
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64)).clone().requires_grad_(True)
    yt = torch.as_tensor(np.asarray(y, dtype=np.float64))
    zt, mt, st, gt = t(z), t(mu), t(s), t(gamma)
    at, bt = t(np.asarray(alpha).reshape(-1)), t(np.asarray(beta).reshape(-1))
    f = torch.sum(fhat(yt, zt, mt, st, gt, at, bt, jitter=jitter))
    g = torch.autograd.grad(f, [mt, st, zt, gt, at, bt])
    return dict(zip(['d_mu', 'd_s', 'd_z', 'd_gamma', 'd_alpha', 'd_beta'], [a.numpy() for a in g]), f_hat=float(f))


def objective(y, raw, s_1=1.0, s_2=1.0, mask_size=1, jitter=GP_DEFAULT_JITTER):
    """dp_gp_lvm.py:100-154 as a function of the raw variables; returns (objective, dict of pieces)."""
    mu, z = raw['x_mean'], raw['x_u']
    s = _softplus(raw['x_var_raw'])
    phi = torch.softmax(raw['dp_logits'], dim=-1)                                            # dirichlet_process.py:40-51
    if mask_size != 1:
        phi = torch.repeat_interleave(phi, int(mask_size), dim=0)
    g1, g2 = _softplus(raw['gamma_1_raw']).reshape(-1), _softplus(raw['gamma_2_raw']).reshape(-1)
    w1, w2 = _softplus(raw['w_1_raw']).reshape(()), _softplus(raw['w_2_raw']).reshape(())
    gat, aat, bat = (_softplus(raw[k]) for k in ('gamma_atoms_raw', 'alpha_atoms_raw', 'beta_atoms_raw'))
    gamma, alpha, beta = phi @ gat, (phi @ aat)[:, 0], (phi @ bat)[:, 0]                      # dp_gp_lvm.py:100-102
    terms = fhat(y, z, mu, s, gamma, alpha, beta, jitter=jitter)
    kl = 0.5 * (torch.sum(mu * mu) + torch.sum(s - torch.log(s)) - mu.shape[0] * mu.shape[1])   # gp_expressions.py:10-24
    hyper = sum(torch.sum(_log_normal_log_pdf(a)) for a in (gat, aat, bat))                  # dp_gp_lvm.py:96-98
    dp = dp_objective(phi, g1, g2, w1, w2, float(s_1), float(s_2))
    obj = dp - (torch.sum(terms) - kl) - hyper                                               # dp_gp_lvm.py:148-154
    return obj, dict(fhat_terms=terms, kl=kl, hyperprior=hyper, dp_objective=dp, gamma=gamma, alpha=alpha, beta=beta,
                     s=s, phi=phi)


def objective_and_gradients(y, raw_values, s_1=1.0, s_2=1.0, mask_size=1, jitter=GP_DEFAULT_JITTER):
    """NumPy in, NumPy out: objective (float) and d objective / d raw variable for the eleven raw variables."""
    yt = torch.as_tensor(np.asarray(y), dtype=torch.float64)
    raw = {k: torch.tensor(np.asarray(raw_values[k], dtype=np.float64), dtype=torch.float64, requires_grad=True)
           for k in NAMES}
    obj, _ = objective(yt, raw, s_1=s_1, s_2=s_2, mask_size=mask_size, jitter=jitter)
    grads = torch.autograd.grad(obj, [raw[k] for k in NAMES], allow_unused=True)
    out = {k: (np.zeros_like(np.asarray(raw_values[k], dtype=np.float64)) if g is None else g.numpy().copy())
           for k, g in zip(NAMES, grads)}
    return float(obj), out


# --------------------------------------------------------------------------------------------------------------------
# Over-T formulation (reference dp_gp_lvm.py:513-676): the kernel batch is the T atoms instead of the
# D mixed hyper-parameter rows, the mixture weights phi enter outside the kernel.  Pinned by oracle/gen_golden_t.py
# (the reference's own dp_gp_lvm_t constructor under the PyTorch stand-in): objective rtol 1e-11, gradients rtol 1e-7.
# --------------------------------------------------------------------------------------------------------------------
def psi_pieces_t(y, z, mu, s, gamma, alpha, jitter=GP_DEFAULT_JITTER, chunk=64):
    """As psi_pieces for T hyper-parameter rows, but Psi1 contracted with ALL columns of y: V [T,M,D] = Psi1_t^T Y."""
    n, d = y.shape
    t, m = gamma.shape[0], z.shape[0]
    zd = z[:, None, :] - z[None, :, :]
    zbar = 0.5 * (z[:, None, :] + z[None, :, :])
    sq = torch.einsum('dq,ijq->dij', gamma, zd * zd)
    k_uu = alpha[:, None, None] * torch.exp(-0.5 * sq) + jitter * torch.eye(m, dtype=y.dtype)
    v = torch.zeros((t, m, d), dtype=y.dtype)
    p2 = torch.zeros((t, m, m), dtype=y.dtype)
    t1 = 0.25 * sq
    for n0 in range(0, n, chunk):
        mu_c, s_c, y_c = mu[n0:n0 + chunk], s[n0:n0 + chunk], y[n0:n0 + chunk]
        den1 = gamma[:, None, :] * s_c[None] + 1.0
        e1 = torch.einsum('cmq,dcq->dcm', (mu_c[:, None, :] - z[None]) ** 2, gamma[:, None, :] / den1) \
            + torch.sum(torch.log(den1), dim=-1)[:, :, None]
        p1 = alpha[:, None, None] * torch.exp(-0.5 * e1)                                     # [T,c,M]
        v = v + torch.einsum('tcm,cd->tmd', p1, y_c)
        den2 = 2.0 * gamma[:, None, :] * s_c[None] + 1.0
        num = (mu_c[:, None, None, :] - zbar[None]) ** 2
        e2 = torch.einsum('cijq,dcq->dcij', num, gamma[:, None, :] / den2)
        lp = 2.0 * torch.log(alpha)[:, None, None, None] - (0.5 * torch.sum(torch.log(den2), dim=-1)[:, :, None, None]
                                                             + t1[:, None] + e2)
        p2 = p2 + torch.sum(torch.exp(lp), dim=1)
    return k_uu, p2, v


def fhat_t(y, z, mu, s, phi, gamma, alpha, beta, jitter=GP_DEFAULT_JITTER, chunk=64):
    """f_hat of dp_gp_lvm.py:617-667 in the B_t = K_t + beta_t Psi2_t form:
        -1/2 N D log 2 pi + sum_td phi_td [ 1/2 (N log beta_t + beta_t (<K_t^-1, Psi2_t> - N alpha_t)) - (log|L_B| - log|L_K|) ]
        - 1/2 sum_td phi_td beta_t y_d^T y_d + 1/2 sum_td phi_td beta_t^2 || L_B,t^-1 Psi1_t^T y_d ||^2     (phi [D,T])"""
    n, d = y.shape
    k_uu, p2, v = psi_pieces_t(y, z, mu, s, gamma, alpha, jitter=jitter, chunk=chunk)
    l_k = torch.linalg.cholesky(k_uu)
    l_b = torch.linalg.cholesky(k_uu + beta[:, None, None] * p2)
    logdet_k = torch.sum(torch.log(torch.diagonal(l_k, dim1=-2, dim2=-1)), dim=-1)
    logdet_b = torch.sum(torch.log(torch.diagonal(l_b, dim1=-2, dim2=-1)), dim=-1)
    tr = torch.sum(torch.cholesky_solve(p2, l_k).diagonal(dim1=-2, dim2=-1), dim=-1)
    c = torch.linalg.solve_triangular(l_b, v, upper=False)                                   # [T,M,D]
    quad = beta[:, None] ** 2 * torch.sum(c * c, dim=1)                                      # [T,D]
    per_t = 0.5 * (n * torch.log(beta) + beta * (tr - alpha * n)) - (logdet_b - logdet_k)    # [T]
    phit = phi.transpose(0, 1)                                                               # [T,D]
    yy = torch.sum(y * y, dim=0)
    return -0.5 * n * d * LOG_2PI + torch.sum(phit * per_t[:, None]) - 0.5 * torch.sum(phit * beta[:, None] * yy[None, :]) \
        + 0.5 * torch.sum(phit * quad)


def objective_t(y, raw, s_1=1.0, s_2=1.0, mask_size=1, jitter=GP_DEFAULT_JITTER):
    """dp_gp_lvm.py:560-676 (dp_gp_lvm_t) as a function of the same eleven raw variables."""
    mu, z = raw['x_mean'], raw['x_u']
    s = _softplus(raw['x_var_raw'])
    phi = torch.softmax(raw['dp_logits'], dim=-1)
    if mask_size != 1:
        phi = torch.repeat_interleave(phi, int(mask_size), dim=0)
    g1, g2 = _softplus(raw['gamma_1_raw']).reshape(-1), _softplus(raw['gamma_2_raw']).reshape(-1)
    w1, w2 = _softplus(raw['w_1_raw']).reshape(()), _softplus(raw['w_2_raw']).reshape(())
    gat, aat, bat = (_softplus(raw[k]) for k in ('gamma_atoms_raw', 'alpha_atoms_raw', 'beta_atoms_raw'))
    f = fhat_t(y, z, mu, s, phi, gat, aat[:, 0], bat[:, 0], jitter=jitter)
    kl = 0.5 * (torch.sum(mu * mu) + torch.sum(s - torch.log(s)) - mu.shape[0] * mu.shape[1])
    hyper = sum(torch.sum(_log_normal_log_pdf(a)) for a in (gat, aat, bat))
    dp = dp_objective(phi, g1, g2, w1, w2, float(s_1), float(s_2))
    return dp - (f - kl) - hyper, dict(f_hat=f, kl=kl, hyperprior=hyper, dp_objective=dp)


def objective_t_and_gradients(y, raw_values, s_1=1.0, s_2=1.0, mask_size=1, jitter=GP_DEFAULT_JITTER):
    yt = torch.as_tensor(np.asarray(y), dtype=torch.float64)
    raw = {k: torch.tensor(np.asarray(raw_values[k], dtype=np.float64), dtype=torch.float64, requires_grad=True)
           for k in NAMES}
    obj, _ = objective_t(yt, raw, s_1=s_1, s_2=s_2, mask_size=mask_size, jitter=jitter)
    grads = torch.autograd.grad(obj, [raw[k] for k in NAMES], allow_unused=True)
    out = {k: (np.zeros_like(np.asarray(raw_values[k], dtype=np.float64)) if g is None else g.numpy().copy())
           for k, g in zip(NAMES, grads)}
    return float(obj), out


# --------------------------------------------------------------------------------------------------------------------
# Bayesian GP-LVM (reference gaussian_process.py:132-270): one kernel for all output dims = the over-T f_hat with a single
# atom and phi = 1; objective = -(f_hat - KL + log-normal hyper-prior of gamma, alpha, beta).  Pinned by
# oracle/gen_golden_bgplvm.py against the reference's own constructor (objective 1e-11, gradients 1e-7).
# --------------------------------------------------------------------------------------------------------------------
BGPLVM_NAMES = ['gamma_raw', 'alpha_raw', 'beta_raw', 'x_mean', 'x_u', 'x_var_raw']


def objective_bgplvm(y, raw, jitter=GP_DEFAULT_JITTER):
    mu, z = raw['x_mean'], raw['x_u']
    s = _softplus(raw['x_var_raw'])
    gam, al, be = _softplus(raw['gamma_raw']), _softplus(raw['alpha_raw']), _softplus(raw['beta_raw'])
    phi = torch.ones((y.shape[1], 1), dtype=y.dtype)
    f = fhat_t(y, z, mu, s, phi, gam, al[:, 0], be[:, 0], jitter=jitter)
    kl = 0.5 * (torch.sum(mu * mu) + torch.sum(s - torch.log(s)) - mu.shape[0] * mu.shape[1])
    hyper = sum(torch.sum(_log_normal_log_pdf(a)) for a in (gam, al, be))
    return -(f - kl + hyper)


def objective_bgplvm_and_gradients(y, raw_values, jitter=GP_DEFAULT_JITTER):
    yt = torch.as_tensor(np.asarray(y), dtype=torch.float64)
    raw = {k: torch.tensor(np.asarray(raw_values[k], dtype=np.float64), dtype=torch.float64, requires_grad=True)
           for k in BGPLVM_NAMES}
    obj = objective_bgplvm(yt, raw, jitter=jitter)
    grads = torch.autograd.grad(obj, [raw[k] for k in BGPLVM_NAMES], allow_unused=True)
    return float(obj), {k: (np.zeros_like(np.asarray(raw_values[k], dtype=np.float64)) if g is None else g.numpy().copy())
                        for k, g in zip(BGPLVM_NAMES, grads)}


# --------------------------------------------------------------------------------------------------------------------
# Manifold relevance determination (reference src/models/gaussian_process.py:551-664): V views share
# q(X); every view has its own B = 1 kernel and its own inducing inputs.  objective = -(sum_v f_hat_v - KL + sum_v prior_v),
# each f_hat_v being the Bayesian GP-LVM's (:236-258 = :619-651).  Pinned by oracle/gen_golden_mrd.py.
# --------------------------------------------------------------------------------------------------------------------
def mrd_names(num_views):
    v = range(num_views)
    return ['gamma_raw_%d' % i for i in v] + ['alpha_raw_%d' % i for i in v] + ['beta_raw_%d' % i for i in v] + \
        ['x_mean', 'x_var_raw'] + ['x_u_%d' % i for i in v]


def objective_mrd(views, raw, jitter=GP_DEFAULT_JITTER):
    mu, s = raw['x_mean'], _softplus(raw['x_var_raw'])
    kl = 0.5 * (torch.sum(mu * mu) + torch.sum(s - torch.log(s)) - mu.shape[0] * mu.shape[1])
    total = -kl
    for i, y in enumerate(views):
        gam, al, be = (_softplus(raw['%s_raw_%d' % (k, i)]) for k in ('gamma', 'alpha', 'beta'))
        phi = torch.ones((y.shape[1], 1), dtype=y.dtype)
        total = total + fhat_t(y, raw['x_u_%d' % i], mu, s, phi, gam, al[:, 0], be[:, 0], jitter=jitter) + \
            sum(torch.sum(_log_normal_log_pdf(a)) for a in (gam, al, be))
    return -total


def objective_mrd_and_gradients(views, raw_values, jitter=GP_DEFAULT_JITTER):
    vt = [torch.as_tensor(np.asarray(y), dtype=torch.float64) for y in views]
    names = mrd_names(len(views))
    raw = {k: torch.tensor(np.asarray(raw_values[k], dtype=np.float64), dtype=torch.float64, requires_grad=True) for k in names}
    obj = objective_mrd(vt, raw, jitter=jitter)
    grads = torch.autograd.grad(obj, [raw[k] for k in names], allow_unused=True)
    return float(obj), {k: (np.zeros_like(np.asarray(raw_values[k], dtype=np.float64)) if g is None else g.numpy().copy())
                        for k, g in zip(names, grads)}
