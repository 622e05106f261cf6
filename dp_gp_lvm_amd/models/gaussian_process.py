"""
Bayesian GP-LVM — mirror of the reference's ``bayesian_gp_lvm`` factory (src/models/gaussian_process.py:132-270; SURVEY.md
§8(f) row 4: the B = 1 model, which needs no new kernels).  One ARD-RBF kernel (gamma [1 x Q], alpha, beta [1 x 1]) serves
all D output dims, so the f_hat of gaussian_process.py:236-258 is the over-T f_hat of ``dp_gp_lvm_t`` with a single atom and
phi = 1; the objective is  -(f_hat - KL(q(X)||p(X)) + kernel.prior_log_likelihood)  (:263-269).  This wrapper therefore
builds a one-atom ``dp_gp_lvm_t`` (library operators for the forward, its autograd.Function for the backward) and drops the
(constant) DP terms.  The reference's stochastic variant (``num_latent_samples > 0``: Monte-Carlo psi statistics through
tensorflow_probability) is not built.
"""
import numpy as np
import torch
import torch.nn.functional as F

from ..kernels.interfaces.kernel import KernelHyperparameters
from ..kernels.rbf_kernel import k_ard_rbf
from ..utils.constants import GP_LVM_DEFAULT_LATENT_DIMENSIONS, GP_LVM_DEFAULT_NUM_INDUCING_POINTS, GP_INIT_GAMMA, \
    GP_INIT_ALPHA, GP_INIT_BETA
from ..utils.expressions import principal_component_analysis as pca
from .dp_gp_lvm import dp_gp_lvm_t
from .interfaces.trainable import Trainable


def bayesian_gp_lvm(y_train, kernel=None, num_latent_dims=GP_LVM_DEFAULT_LATENT_DIMENSIONS,
                    num_inducing_points=GP_LVM_DEFAULT_NUM_INDUCING_POINTS, num_latent_samples=0,
                    device=None, precision=None, initial_values=None):
    """
    :param y_train: [N x D] numpy array.  :param kernel: optional k_ard_rbf with batch size 1 whose hyper-parameter VALUES
    initialise the model's own trainable ones.  :param num_latent_dims: Q.  :param num_inducing_points: M (< N).
    :param num_latent_samples: must be 0 (closed-form psi statistics).
    Extensions: device, precision ('mixed' | 'f64'), initial_values (x_mean, x_var, x_u, gamma, alpha, beta: values).
    """
    num_samples, num_dimensions = np.shape(y_train)
    assert isinstance(num_latent_dims, int), 'Number of latent dimensions must be an integer.'
    assert 0 < num_latent_dims < num_dimensions, \
        'Number of latent dimensions must be postive and less than the dimensionality of the observed data.'
    assert isinstance(num_inducing_points, int), 'Number of inducing points must be an integer.'
    assert 0 < num_inducing_points < num_samples, \
        'Number of inducing points must be positive and less than the number of observations in the observed data.'
    assert isinstance(num_latent_samples, int), 'Number of latent space samples must be an integer.'
    if num_latent_samples:
        raise NotImplementedError('stochastic psi statistics (gaussian_process.py:189-214) are not built')
    iv = dict(initial_values or {})
    if kernel is not None:
        hp = kernel.hyperparameters
        iv.setdefault('gamma', hp[KernelHyperparameters.ARD_WEIGHTS].detach().cpu().numpy())
        iv.setdefault('alpha', hp[KernelHyperparameters.SIGNAL_VARIANCE].detach().cpu().numpy())
        iv.setdefault('beta', hp[KernelHyperparameters.NOISE_PRECISION].detach().cpu().numpy())
    q = num_latent_dims
    inner_iv = dict(gamma_atoms=np.asarray(iv.get('gamma', np.full((1, q), GP_INIT_GAMMA)), dtype=np.float64).reshape(1, q),
                    alpha_atoms=np.asarray(iv.get('alpha', GP_INIT_ALPHA), dtype=np.float64).reshape(1, 1),
                    beta_atoms=np.asarray(iv.get('beta', GP_INIT_BETA), dtype=np.float64).reshape(1, 1),
                    x_var=np.asarray(iv.get('x_var', np.full((num_samples, q), 0.5)), dtype=np.float64),   # (:218: 0.5, not 1)
                    phi_logits=np.zeros((num_dimensions, 1)))
    for k in ('x_mean', 'x_u'):
        if k in iv:
            inner_iv[k] = iv[k]
    inner = dp_gp_lvm_t(y_train, num_latent_dims=num_latent_dims, num_inducing_points=num_inducing_points, truncation_level=1,
                        device=device, precision=precision, initial_values=inner_iv)
    names = ('x_mean', 'x_var', 'x_u', 'gamma_atoms', 'alpha_atoms', 'beta_atoms')
    raw = {k: inner.raw[k] for k in names}

    def _gradients():
        g = inner.gradients()                                 # (the DP terms of the one-atom model do not touch these six)
        return {k: g[k] for k in names}

    def _optimise(num_iterations, learning_rate=0.01, callback=None):
        opt = torch.optim.Adam(list(raw.values()), lr=learning_rate)
        for it in range(num_iterations):
            g = _gradients()
            for k, p_ in raw.items():
                p_.grad = g[k].reshape(p_.shape)
            opt.step()
            if callback is not None:
                callback(it)

    class BayesianGPLVM(Trainable):
        """Accessors as in the reference (gaussian_process.py:276-340)."""
        raw_variables = raw

        @property
        def kernel(self):
            return k_ard_rbf(gamma=F.softplus(raw['gamma_atoms']), alpha=F.softplus(raw['alpha_atoms']),
                             beta=F.softplus(raw['beta_atoms']))

        @property
        def ard_weights(self):
            return F.softplus(raw['gamma_atoms'])

        @property
        def signal_variance(self):
            return F.softplus(raw['alpha_atoms'])

        @property
        def noise_precision(self):
            return F.softplus(raw['beta_atoms'])

        @property
        def inducing_input(self):
            return raw['x_u']

        @property
        def q_x(self):
            return raw['x_mean'], torch.diag_embed(F.softplus(raw['x_var']))

        @property
        def objective(self):
            """-(f_hat - KL + hyper-prior) (gaussian_process.py:263-269): 0-d fp64 device tensor."""
            t = inner.objective_terms                           # (objective_t, f_hat, KL, DP objective, hyper-prior)
            return t[0] - t[3]

        gradients = staticmethod(_gradients)
        optimise = staticmethod(_optimise)

    return BayesianGPLVM()


def manifold_relevance_determination(views_train, num_latent_dims=GP_LVM_DEFAULT_LATENT_DIMENSIONS,
                                     num_inducing_points=GP_LVM_DEFAULT_NUM_INDUCING_POINTS,
                                     device=None, precision=None, initial_values=None):
    """
    Manifold relevance determination — mirror of the reference's factory (src/models/gaussian_process.py:551-664): V views
    [N x D_v] share q(X); every view has its own B = 1 ARD-RBF kernel and its own M inducing inputs, and
        objective = -( sum_v f_hat_v - KL(q(X)||p(X)) + sum_v kernel_v.prior_log_likelihood )                      (:653-664)
    with f_hat_v the Bayesian GP-LVM's f_hat of view v (:619-651 = :236-258).  Like ``bayesian_gp_lvm`` it needs no new kernels:
    every view is a one-atom ``dp_gp_lvm_t`` (library operators forward, its streaming stage B backward) whose q(X) tensors
    are the SAME storage; the shared KL is counted once.
    Extensions: device, precision ('mixed' | 'f64'), initial_values (x_mean, x_var [N x Q]; gamma, alpha, beta, x_u: lists
    with one entry per view).
    """
    num_views = len(views_train)
    shapes = np.array([np.shape(v) for v in views_train])
    num_samples = [shapes[v][0] for v in range(num_views)]
    num_dimensions = [int(shapes[v][1]) for v in range(num_views)]
    assert np.size(np.unique(num_samples)) == 1, 'Each view must have the same number of observations.'
    num_samples = int(num_samples[0])
    assert 0 < num_latent_dims < np.sum(num_dimensions), \
        'Number of latent dimensions must be postive and less than the dimensionality of the observed data.'
    assert 0 < num_inducing_points < num_samples, \
        'Number of inducing points must be positive and less than the number of observations in the observed data.'
    iv = dict(initial_values or {})
    q, m = num_latent_dims, num_inducing_points
    x_init = np.asarray(iv['x_mean'], dtype=np.float64) if 'x_mean' in iv else \
        pca(np.hstack([np.asarray(v) for v in views_train]), num_latent_dimensions=q)                          # (:591)
    x_var = np.asarray(iv.get('x_var', np.ones((num_samples, q))), dtype=np.float64)                            # (:593: 1.0)
    inner = []
    for v in range(num_views):
        x_u = np.asarray(iv['x_u'][v], dtype=np.float64) if 'x_u' in iv else \
            np.random.permutation(x_init)[:m] + np.random.normal(loc=0.0, scale=0.01, size=(m, q))              # (:599-601)
        pick = lambda key, default, shape: np.asarray(iv[key][v] if key in iv else default, dtype=np.float64).reshape(shape)
        inner.append(dp_gp_lvm_t(np.asarray(views_train[v]), num_latent_dims=q, num_inducing_points=m, truncation_level=1,
                                 device=device, precision=precision, _view_of_many=True,
                                 initial_values=dict(x_mean=x_init, x_var=x_var, x_u=x_u,
                                                     gamma_atoms=pick('gamma', np.full((1, q), GP_INIT_GAMMA), (1, q)),
                                                     alpha_atoms=pick('alpha', GP_INIT_ALPHA, (1, 1)),
                                                     beta_atoms=pick('beta', GP_INIT_BETA, (1, 1)),
                                                     phi_logits=np.zeros((num_dimensions[v], 1)))))
    x_mean_t, x_var_raw = inner[0].raw['x_mean'], inner[0].raw['x_var']
    for mv in inner[1:]:                                     # one q(X): same storage in every view's model
        mv.raw['x_mean'].data = x_mean_t.data
        mv.raw['x_var'].data = x_var_raw.data
    per_view = ('x_u', 'gamma_atoms', 'alpha_atoms', 'beta_atoms')
    raw = dict(x_mean=x_mean_t, x_var=x_var_raw)
    for v, mv in enumerate(inner):
        for k in per_view:
            raw['%s_%d' % (k, v)] = mv.raw[k]

    def _objective():
        terms = [mv.objective_terms for mv in inner]          # (objective_t, f_hat, KL, DP objective, hyper-prior) per view
        return sum(t[0] - t[3] for t in terms) - (num_views - 1) * terms[0][2]

    def _gradients():
        g = [mv.gradients() for mv in inner]
        s_ = F.softplus(x_var_raw)
        out = dict(x_mean=sum(gv['x_mean'] for gv in g) - (num_views - 1) * x_mean_t,          # KL counted once:
                   x_var=sum(gv['x_var'] for gv in g) -                                           # gp_expressions.py:10-24
                   (num_views - 1) * 0.5 * (1.0 - 1.0 / s_) * torch.sigmoid(x_var_raw))
        for v, gv in enumerate(g):
            for k in per_view:
                out['%s_%d' % (k, v)] = gv[k]
        return out

    def _optimise(num_iterations, learning_rate=0.01, callback=None):
        opt = torch.optim.Adam(list(raw.values()), lr=learning_rate)
        for it in range(num_iterations):
            g = _gradients()
            bad = ~torch.stack([torch.isfinite(v).all() for v in g.values()]).all()
            for mv in inner:
                bad = bad | (mv.cholesky_info != 0)
            if bool(bad):
                eff = precision or 'f64'                        # (None resolves to the reference's fp64 in the models it builds)
                raise FloatingPointError('iteration %d: failed Cholesky factorisation or non-finite gradient (precision=%r)%s'
                                         % (it, eff, '' if eff == 'f64' else '; use precision="f64"'))
            for k, p_ in raw.items():
                p_.grad = g[k].reshape(p_.shape)
            opt.step()
            if callback is not None:
                callback(it)

    class ManifoldRelevanceDetermination(Trainable):
        """Accessors as in the reference (gaussian_process.py:667-727)."""
        raw_variables = raw

        @property
        def number_of_views(self):
            return num_views

        @property
        def kernels(self):
            return [mv.kernel for mv in inner]

        @property
        def ard_weights(self):
            return [F.softplus(mv.raw['gamma_atoms']) for mv in inner]

        @property
        def signal_variance(self):
            return [F.softplus(mv.raw['alpha_atoms']) for mv in inner]

        @property
        def noise_precision(self):
            return [F.softplus(mv.raw['beta_atoms']) for mv in inner]

        @property
        def inducing_input(self):
            return [mv.raw['x_u'] for mv in inner]

        @property
        def q_x(self):
            return x_mean_t, torch.diag_embed(F.softplus(x_var_raw))

        @property
        def objective(self):
            return _objective()

        gradients = staticmethod(_gradients)
        optimise = staticmethod(_optimise)

    return ManifoldRelevanceDetermination()
