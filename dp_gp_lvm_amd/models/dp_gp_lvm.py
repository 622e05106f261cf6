"""
DP-GP-LVM — mirror of the reference's ``dp_gp_lvm`` factory (src/models/dp_gp_lvm.py:22-510) with the same signature,
assertions and accessors; ``.objective`` is evaluated by the HIP library (libdpgp_hip.so):

    dpgp_model_prepare  ->  dpgp_elbo_fhat  ->  [one packed all-reduce when D is sharded]  ->  dpgp_model_finalize

i.e. five kernel launches and at most one collective per evaluation, with no host arithmetic in between.
Where the reference returns lazy TensorFlow nodes, accessors here return torch tensors computed from the current
parameter values, and ``objective`` re-evaluates on every access.

Multi-GPU: the per-output-dimension terms are independent given (q(X), Z, atoms), so the D outputs are sharded over the
ranks of ``process_group`` (rank r keeps columns r*D/W .. (r+1)*D/W of Y and the matching rows of phi); q(X), Z and the
atoms are replicated.  The only exchange per evaluation is a sum all-reduce of the two scalars (f_hat, DP objective).
"""
import os
import numpy as np
import torch
import torch.nn.functional as F

from .. import _lib, ops
from ..kernels.interfaces.kernel import KernelHyperparameters
from ..kernels.rbf_kernel import k_ard_rbf
from ..utils.constants import GP_LVM_DEFAULT_LATENT_DIMENSIONS, GP_LVM_DEFAULT_NUM_INDUCING_POINTS, \
    DP_DEFAULT_TRUNCATION_LEVEL, DP_DEFAULT_ALPHA_PRIOR_PARAMS, GP_INIT_GAMMA, GP_INIT_ALPHA, GP_INIT_BETA, \
    GP_DEFAULT_JITTER
from ..utils.expressions import principal_component_analysis as pca
from ..utils.types import TORCH_DTYPE, create_positive_variable, default_device, register_variable
from .dirichlet_process import dirichlet_process
from .interfaces.trainable import Trainable


def shard_bounds(num_dimensions, rank, world_size):
    """Contiguous, balanced split of the D output dims: rank r owns [lo, hi)."""
    base, rem = divmod(num_dimensions, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _adam(params, learning_rate):
    """torch.optim.Adam on the raw variables (plumbing: it only applies the update).  The fused implementation where this build of
    torch has it for fp64 device tensors — one launch instead of a dozen element-wise ones per iteration (DPGP_FUSED_ADAM=0: the default
    one)."""
    if os.environ.get('DPGP_FUSED_ADAM', '1') != '0':
        try:
            return torch.optim.Adam(params, lr=learning_rate, fused=True)
        except (RuntimeError, TypeError, ValueError):
            pass
    return torch.optim.Adam(params, lr=learning_rate)


def dp_gp_lvm(y_train,
              num_latent_dims=GP_LVM_DEFAULT_LATENT_DIMENSIONS,
              num_inducing_points=GP_LVM_DEFAULT_NUM_INDUCING_POINTS,
              truncation_level=DP_DEFAULT_TRUNCATION_LEVEL,
              alpha_prior_params=DP_DEFAULT_ALPHA_PRIOR_PARAMS,
              mask_size=1,
              device=None, precision=None, process_group=None, initial_values=None, backward_precision=None,
              psi_algo='auto', _shard_of=None):
    """
    :param y_train: [N x D] numpy array, columns normalised to zero mean / unit variance (dp_gp_lvm.py:30-32).
    :param num_latent_dims: Q.  :param num_inducing_points: M.  :param truncation_level: T.
    :param alpha_prior_params: Gamma prior (s_1, s_2) on the DP concentration.  :param mask_size: see the reference.
    Extensions (keyword-only in spirit, all optional):
    :param device: torch device of the parameters (default: current GPU).
    :param precision: 'mixed' (psi-statistics fp32 on the matrix cores, Cholesky chain fp64), 'f64' or 'f32'.
    :param process_group: torch.distributed group over which the D output dims are sharded (None: single GPU).
    :param backward_precision: precision of the streaming stage (B) of the backward pass; default: `precision`.  'mixed' with
           precision='f64' = fp64 forward and dense adjoints (accurate B^-1, K^-1 however ill-conditioned K_uu is), the
           streaming stage on the matrix pipe (gradients to ~1e-4): the training configuration (see optimise()).
    :param psi_algo: 'auto' (fp32 psi-statistics on f16 hi/lo-split MFMA operands), 'mfma_f32' (exact fp32 products on
           v_mfma_f32_16x16x4_f32: no f16 range limit, slower) or 'plain' (VALU cross-check kernels); forward evaluation only.
    :param initial_values: dict of post-initialisation parameter VALUES (x_mean, x_var, x_u, phi_logits, gamma_atoms,
           alpha_atoms, beta_atoms, gamma_1, gamma_2, w_1, w_2) that replace the random/PCA initialisation — used by the
           parity tests and the benchmark, which must not depend on PCA sign conventions or NumPy's global RNG.
    """
    num_samples, num_dimensions = np.shape(y_train)
    assert 0 < num_latent_dims <= num_dimensions, \
        'Number of latent dimensions must be postive and less than the dimensionality of the observed data.'
    assert 0 < num_inducing_points <= num_samples, \
        'Number of inducing points must be positive and less than the number of observations in the observed data.'
    assert 0 < truncation_level <= min(num_samples, num_dimensions), \
        'The truncation level must be positive and less than the dimensionality of the observed data and ' \
        'less than the number of observations.'
    if precision is None:
        # the default-constructed model is the one that trains like the reference (fp64 arithmetic, src/utils/types.py:13-14):
        # fp64 forward pass and dense adjoints, the streaming stage of the backward pass on the matrix pipe.
        # precision='mixed' (fp32 psi-statistics) is the fast scoring mode: its conditioning guard stops a long Adam run.
        precision, backward_precision = 'f64', (backward_precision or 'mixed')
    assert precision in ('f32', 'mixed', 'f64'), "precision must be one of 'f32', 'mixed', 'f64'"
    assert backward_precision in (None, 'mixed', 'f64', 'mixed_fast'), "backward_precision must be None, 'mixed', 'mixed_fast' or 'f64'"
    # stage B behind an fp64 forward pass (the training configuration): the patch form of the Psi2 term, which keeps its accuracy
    # where the adjoints cancel (include/dpgp.h, DPGP_PREC_MIXED_PATCH); behind a mixed forward pass the faster pair-tile form
    # precision='f32' (the psi-statistics as exact fp32 products, fp32 chain) is a scoring precision: its gradients are those of the
    # mixed evaluation — the same function to the mixed tolerance, fp64 dense adjoints — on a second, lazily built workspace
    grad_precision = 'mixed' if precision == 'f32' else precision
    stage_b_precision = backward_precision or grad_precision
    if precision == 'f64' and backward_precision == 'mixed':
        stage_b_precision = 'mixed_patch'
    assert psi_algo in _lib.ALGO, 'psi_algo must be one of %s' % sorted(_lib.ALGO)
    assert truncation_level <= 64, 'truncation levels above 64 are not supported by the HIP model kernels (PREP_MAX_T)'
    device = default_device() if device is None else torch.device(device)
    iv = dict(initial_values or {})

    def _t(a):
        return torch.as_tensor(np.asarray(a, dtype=np.float64), dtype=TORCH_DTYPE, device=device).contiguous()

    def _raw_pos(name, init, shape):
        if name in iv:
            v = np.asarray(iv[name], dtype=np.float64).reshape(shape)
            return _t(np.log(np.expm1(v)))
        return create_positive_variable(init, shape, device=device)

    # ---- variational parameters (dp_gp_lvm.py:62-74) ----
    if 'x_mean' in iv:
        x_init = np.asarray(iv['x_mean'], dtype=np.float64)
    else:
        x_init = pca(np.asarray(y_train), num_latent_dimensions=num_latent_dims)
    x_mean = _t(x_init)                                                        # [N x Q]
    x_var_raw = _raw_pos('x_var', 1.0, (num_samples, num_latent_dims))         # softplus(raw) = diag of q(X) covariance
    if 'x_u' in iv:
        x_u = _t(iv['x_u'])
    else:
        x_u = _t(np.random.permutation(x_init)[:num_inducing_points] +
                 np.random.normal(loc=0.0, scale=0.01, size=(num_inducing_points, num_latent_dims)))   # [M x Q]

    # ---- DP over the D output dims (:77-80) and the atoms of the kernel hyper-parameters (:84-94) ----
    dp_model = dirichlet_process(num_samples=num_dimensions, alpha_prior_params=alpha_prior_params,
                                 truncation_level=truncation_level, mask_size=mask_size, device=device)
    for key, name in (('logits', 'phi_logits'), ('gamma_1', 'gamma_1'), ('gamma_2', 'gamma_2')):
        if name in iv:
            v = np.asarray(iv[name], dtype=np.float64)
            dp_model.raw[key].copy_(_t(v if key == 'logits' else np.log(np.expm1(v))).reshape(dp_model.raw[key].shape))
    for i, name in enumerate(('w_1', 'w_2')):
        if name in iv:
            dp_model.raw['w'][i] = float(np.log(np.expm1(float(iv[name]))))
    gamma_atoms_raw = _raw_pos('gamma_atoms', GP_INIT_GAMMA, (truncation_level, num_latent_dims))
    sig_var_atoms_raw = _raw_pos('alpha_atoms', GP_INIT_ALPHA, (truncation_level, 1))
    beta_atoms_raw = _raw_pos('beta_atoms', GP_INIT_BETA, (truncation_level, 1))

    # the reference's creation order of its trainable tf.Variables (dp_gp_lvm.py:63-94, dirichlet_process.py:40-59)
    for v_ in (x_mean, x_var_raw, x_u, dp_model.raw['logits'], dp_model.raw['gamma_1'], dp_model.raw['gamma_2'],
               dp_model.raw['w'], gamma_atoms_raw, sig_var_atoms_raw, beta_atoms_raw):
        register_variable(v_, trainable=True)

    # ---- D-sharding ----
    if process_group is not None:
        import torch.distributed as dist
        rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
    elif _shard_of is not None:
        # test hook: behave as rank r of w WITHOUT a communicator — objective terms / gradients come back as this
        # rank's PARTIAL values (what would enter the all-reduce), so that one GPU can check the sharding arithmetic
        dist, (rank, world) = None, _shard_of
    else:
        dist, rank, world = None, 0, 1
    sharded = process_group is not None          # (a 1-rank group still goes through pack -> all_reduce -> finalize)
    d_lo, d_hi = shard_bounds(num_dimensions, rank, world)
    d_local = d_hi - d_lo
    assert d_local > 0, 'more ranks than output dimensions'
    y_local = _t(np.asarray(y_train)[:, d_lo:d_hi])                            # [N x D_local]

    # ---- device buffers of one objective evaluation (allocated once) ----
    f64 = dict(dtype=TORCH_DTYPE, device=device)
    buf = dict(gamma=torch.empty((d_local, num_latent_dims), **f64), alpha=torch.empty((d_local, 1), **f64),
               beta=torch.empty((d_local, 1), **f64), s=torch.empty((num_samples, num_latent_dims), **f64),
               phi=torch.empty((d_local, truncation_level), **f64),
               scal=torch.zeros(_lib.lib().dpgp_model_scal_count(d_local), **f64),
               red=torch.zeros(2, **f64), out=torch.zeros(5, **f64))
    workspace = ops.ElboWorkspace(d_local, num_samples, num_inducing_points, num_latent_dims, precision, device)
    s_1, s_2 = dp_model.prior

    # one training step through dpgp_elbo_step (mixed precision, M <= 128, Q <= 20: DESIGN.md section 7.1): the forward's psi2 dispatch is
    # replaced by the first pass of stage B (DPGP_FUSED_STEP=0: the three separate calls, e.g. for bench.py's per-stage breakdown)
    # fp64 forward + 'mixed' stage B: pair-tile or patch form by the conditioning of K_uu (see _gradients; DPGP_ADAPTIVE_STAGE_B=0: always
    # the patch form)
    DPGP_GUARD_REL = 2.0e-3                                  # (include/dpgp.h)
    adaptive_stage_b = (precision == 'f64' and backward_precision == 'mixed' and num_latent_dims <= 20 and
                        os.environ.get('DPGP_ADAPTIVE_STAGE_B', '1') != '0')
    step_ok = (grad_precision == 'mixed' and stage_b_precision in ('mixed', 'mixed_fast') and psi_algo == 'auto' and
               num_latent_dims <= 20 and os.environ.get('DPGP_FUSED_STEP', '1') != '0')
    fused_step = step_ok and ops.elbo_step_supported(num_inducing_points, num_latent_dims)
    split_step = step_ok and not fused_step            # (M > 128: the two halves of the step around the host-composed stage A)
    step_state = {}

    def grad_workspace():
        if precision != 'f32':
            return workspace
        if 'ws_mixed' not in step_state:
            step_state['ws_mixed'] = ops.ElboWorkspace(d_local, num_samples, num_inducing_points, num_latent_dims, 'mixed', device)
        return step_state['ws_mixed']

    def evaluate(events=None, out=None, _local_part_only=False, _step=False, _half_step=False, _grad=False):
        """One objective evaluation; returns the device tensor out[5] = (objective, f_hat, KL, DP objective, hyper-prior).
        Five launches: prepare, front (KL, y'y, K_uu tiles, operand constants, pair factors), psi1T_y,
        psi2 (+ the K_uu tasks), chain_b (+ the final reduction: sum, pack, finalize); D sharded: the all-reduce and the
        finalising launch follow."""
        lib, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
        r = dp_model.raw
        out = buf['out'] if out is None else out
        _lib.check(lib.dpgp_model_prepare(
            d_local, truncation_level, num_latent_dims, num_samples, d_lo, mask_size, r['logits'].data_ptr(),
            gamma_atoms_raw.data_ptr(), sig_var_atoms_raw.data_ptr(), beta_atoms_raw.data_ptr(), x_var_raw.data_ptr(),
            r['gamma_1'].data_ptr(), r['gamma_2'].data_ptr(), r['w'].data_ptr(), s_1, s_2, 1 if rank == 0 else 0,
            buf['gamma'].data_ptr(), buf['alpha'].data_ptr(), buf['beta'].data_ptr(), buf['s'].data_ptr(),
            buf['phi'].data_ptr(), buf['scal'].data_ptr(), st), 'dpgp_model_prepare')
        red = buf['red']
        # single GPU: the last kernel of the fused ELBO also packs and finalises; sharded: it packs, then one all-reduce
        if (_step or _half_step) and 'buf' not in step_state:
            step_state['buf'] = ops.ElboStepBuffers(d_local, num_samples, num_inducing_points, num_latent_dims, device)
        ws_ = grad_workspace() if (_step or _half_step or _grad) else workspace
        if _half_step:
            ops.elbo_fhat_step(y_local, x_u, x_mean, buf['s'], buf['gamma'], buf['alpha'], buf['beta'], ws_, step_state['buf'],
                               jitter=GP_DEFAULT_JITTER, model_tail=(buf['scal'], red, None if sharded else out))
        elif _step:
            step_state['grads'] = ops.elbo_step(y_local, x_u, x_mean, buf['s'], buf['gamma'], buf['alpha'], buf['beta'], ws_,
                                                step_state['buf'], jitter=GP_DEFAULT_JITTER,
                                                model_tail=(buf['scal'], red, None if sharded else out),
                                                stage_b=stage_b_precision)[1]
        else:
            ops.elbo_fhat(y_local, x_u, x_mean, buf['s'], buf['gamma'], buf['alpha'], buf['beta'],
                          jitter=GP_DEFAULT_JITTER, prec=grad_precision if _grad else precision, algo=psi_algo, workspace=ws_, events=events,
                          model_tail=(buf['scal'], red, None if sharded else out))
        if sharded and not _local_part_only:
            _exchange_and_finalise(out, ws_)
        return out

    def _exchange_and_finalise(out, ws_=None):
        lib, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
        dist.all_reduce(buf['red'], op=dist.ReduceOp.SUM, group=process_group)    # the only exchange: 2 fp64 scalars
        _lib.check(lib.dpgp_model_finalize(buf['red'].data_ptr(), (ws_ or workspace).sums[1:2].data_ptr(), buf['scal'][1:2].data_ptr(),
                                           out.data_ptr(), st), 'dpgp_model_finalize')

    graph_state = {}

    def _evaluate_graph(out=None):
        """evaluate() replayed from a HIP graph (captured on first use): one hipGraphLaunch instead of five kernel launches
        and their argument marshalling on the host — at small per-GPU shares (D / 8 output dims) the host side of the eager
        path is as long as the kernels.  The graph reads the raw variables in place, so optimiser updates are seen.  With a
        process group the graph holds this rank's part (prepare ... pack); the 2-scalar all-reduce and the finalising kernel
        follow as ordinary calls (a collective inside a captured graph is transport-specific; this form works with any).
        Returns the same device tensor as evaluate(); `out` (optional, [5]) receives a copy on the stream."""
        if 'graph' not in graph_state:
            evaluate()                                   # (first call outside the capture: function attributes, warm-up)
            torch.cuda.synchronize()
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph):
                evaluate(_local_part_only=True)
            graph_state['graph'] = gph
        graph_state['graph'].replay()
        if sharded:
            _exchange_and_finalise(buf['out'])
        if out is not None:
            out.copy_(buf['out'], non_blocking=True)
        return buf['out']

    grad_names = ('x_mean', 'x_var', 'x_u', 'dp_logits', 'dp_gamma_1', 'dp_gamma_2', 'dp_w', 'gamma_atoms', 'alpha_atoms',
                  'beta_atoms')

    grad_state = {}

    def _gradients(events=None):
        """d objective / d (raw trainable variables) — what tf.gradients(objective, trainable variables) gives the
        reference's optimiser (test/synthetic_data_hard_test.py:143-155).  dpgp_elbo_step (one call: forward, stage A, stage B;
        mixed precision, M <= 128, Q <= 20) or one forward evaluation, then dpgp_elbo_grad_chain and dpgp_elbo_grad_psi; then
        dpgp_model_backward; sharded over D, the per-GPU
        partial gradients are packed into one buffer and sum-all-reduced.  Returns {name: tensor} keyed like ``raw``.
        events: optional list of 4 torch.cuda.Event(enable_timing=True), recorded after the forward evaluation, stage A,
        stage B and the chain rule to the raw variables (bench.py's breakdown)."""
        lib, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
        gws = grad_workspace()
        mark = (lambda i: events[i].record()) if events is not None else (lambda i: None)
        r = dp_model.raw
        if fused_step and events is None:
            evaluate(_step=True)
            dmu, ds, dz, dg, dab, _ = step_state['grads']
        elif split_step and events is None:
            evaluate(_half_step=True)
            gp, wk, gv, dab, _ = ops.elbo_grad_chain(buf['alpha'], buf['beta'], gws, jitter=GP_DEFAULT_JITTER, z=x_u,
                                                     gamma=buf['gamma'], psi2_slabs=1)
            dmu, ds, dz, dg = ops.elbo_grad_psi_step(y_local, x_u, x_mean, buf['s'], buf['gamma'], buf['alpha'], gp, wk, gv, gws,
                                                     step_state['buf'], stage_b=stage_b_precision)
        else:
            evaluate(_grad=True)
            mark(0)
            gp, wk, gv, dab, _ = ops.elbo_grad_chain(buf['alpha'], buf['beta'], gws, jitter=GP_DEFAULT_JITTER, z=x_u,
                                                     gamma=buf['gamma'])
            mark(1)
            form = stage_b_precision
            if adaptive_stage_b:
                # the training configuration (fp64 forward, matrix-pipe stage B): the pair-tile form of the Psi2 term (4.6 ms at config 3)
                # while K_uu is well conditioned, the patch form (7.5 ms) once it is not — the pair form accumulates the a'- and b-weighted
                # exponentials separately and loses accuracy where they cancel (include/dpgp.h, DPGP_PREC_MIXED_PATCH: 7e-3 against 2e-3 of
                # the largest gradient entry at the ill-conditioned fixture, whose guard bound is 80 x the threshold).  The measure is the
                # conditioning-guard bound the forward evaluation computes in every precision (one host read per step: 10 ms of fp64
                # forward pass are in front of it); the switch sits at a tenth of the threshold a mixed forward pass is flagged at.
                bound = float(gws.guard.max())
                form = 'mixed' if bound <= 0.1 * DPGP_GUARD_REL * num_samples else 'mixed_patch'
                grad_state['stage_b_form'] = form
            dmu, ds, dz, dg = ops.elbo_grad_psi(y_local, x_u, x_mean, buf['s'], buf['gamma'], buf['alpha'], gp, wk, gv,
                                                prec=form)
            mark(2)
        rows = r['logits'].shape[0]
        sizes = [num_samples * num_latent_dims, num_samples * num_latent_dims, num_inducing_points * num_latent_dims,
                 rows * truncation_level, max(truncation_level - 1, 1), max(truncation_level - 1, 1), 2,
                 truncation_level * num_latent_dims, truncation_level, truncation_level]
        flat = torch.zeros(sum(sizes) + 1, **f64)          # (+1: this rank's trouble flag, summed over ranks with the rest)
        parts = list(torch.split(flat, sizes + [1]))[:-1]
        t_ = truncation_level
        _lib.check(lib.dpgp_model_backward(
            d_local, t_, num_latent_dims, num_samples, num_inducing_points, d_lo, mask_size, rows, r['logits'].data_ptr(),
            gamma_atoms_raw.data_ptr(), sig_var_atoms_raw.data_ptr(), beta_atoms_raw.data_ptr(), x_var_raw.data_ptr(),
            r['gamma_1'].data_ptr(), r['gamma_2'].data_ptr(), r['w'].data_ptr(), x_mean.data_ptr(), buf['phi'].data_ptr(),
            s_1, s_2, 1 if rank == 0 else 0, dmu.data_ptr(), ds.data_ptr(), dz.data_ptr(), dg.data_ptr(), dab.data_ptr(),
            *[p_.data_ptr() for p_ in parts], st), 'dpgp_model_backward')
        mark(3)
        # trouble flag of the LOCAL output dims (a Cholesky / conditioning flag of the forward evaluation or a non-finite
        # partial gradient), reduced with the gradients: every rank sees the same decision (optimise() branches on it)
        _lib.check(lib.dpgp_trouble_flag(flat.numel() - 1, flat.data_ptr(), d_local, gws.info.data_ptr(),
                                         flat[-1:].data_ptr(), st), 'dpgp_trouble_flag')
        if sharded:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)      # one packed exchange (N Q x 2 + M Q + ...)
        grad_state['flag'] = flat[-1]
        shapes = [x_mean.shape, x_var_raw.shape, x_u.shape, r['logits'].shape, r['gamma_1'].shape, r['gamma_2'].shape,
                  r['w'].shape, gamma_atoms_raw.shape, sig_var_atoms_raw.shape, beta_atoms_raw.shape]
        return {k: p_[:int(np.prod(sh))].reshape(sh) for k, p_, sh in zip(grad_names, parts, shapes)}

    def _optimise(num_iterations, learning_rate=0.01, callback=None):
        """Adam on the raw variables with the HIP gradients (the reference: tf.train.AdamOptimizer(...).minimize(objective),
        test/synthetic_data_hard_test.py:143-155).  torch.optim.Adam only applies the update (plumbing).

        Every iteration checks the flags of the evaluation — failed factorisation, the conditioning guard of the fp32 Psi2
        (DPGP_INFO_ILL_CONDITIONED, include/dpgp.h), non-finite gradients — as ONE value that travels with the packed gradient
        all-reduce, so all ranks of a sharded run take the same branch, and raises FloatingPointError (the reference's
        tf.cholesky raises InvalidArgumentError there) instead of updating parameters with meaningless gradients.
        precision='mixed' carries Psi2 in fp32: once training drives K_uu towards singularity (long length scales: its small
        eigenvalues approach the 1e-8 jitter) that is no longer the reference's objective and the guard fires (DESIGN.md
        section 5).  The training configuration that follows the reference's fp64 arithmetic all the way is
        precision='f64', backward_precision='mixed': fp64 forward and dense adjoints, the streaming stage of the backward pass
        on the matrix pipe.  Returns {'iterations', 'precision', 'backward_precision'}."""
        params = dict(x_mean=x_mean, x_var=x_var_raw, x_u=x_u, dp_logits=dp_model.raw['logits'],
                      dp_gamma_1=dp_model.raw['gamma_1'], dp_gamma_2=dp_model.raw['gamma_2'], dp_w=dp_model.raw['w'],
                      gamma_atoms=gamma_atoms_raw, alpha_atoms=sig_var_atoms_raw, beta_atoms=beta_atoms_raw)
        opt = _adam(list(params.values()), learning_rate)
        for it in range(num_iterations):
            g = _gradients()
            if float(grad_state['flag']) != 0.0:                                 # identical on every rank
                raise FloatingPointError(
                    'iteration %d: failed Cholesky factorisation, ill-conditioning flag or non-finite gradient in '
                    'precision=%r (info codes of this rank: %s).  The fp32 psi-statistics of precision="mixed" stop being a '
                    'substitute for the reference\'s fp64 once K_uu is nearly singular: build the model with '
                    'precision="f64", backward_precision="mixed".'
                    % (it, precision, sorted(set(workspace.info.unique().tolist()) - {0})))
            for k, p_ in params.items():
                p_.grad = g[k].reshape(p_.shape)                                 # (views of this step's own packed buffer: no copy)
            opt.step()
            if callback is not None:
                callback(it)
        return {'iterations': num_iterations, 'precision': precision,
                'backward_precision': backward_precision or precision}

    def _mixed():
        phi = dp_model.assignments                                               # [D x T], all output dims
        return (phi @ F.softplus(gamma_atoms_raw), phi @ F.softplus(sig_var_atoms_raw), phi @ F.softplus(beta_atoms_raw))

    pred_state = {}

    def _init_test_latents(y_test, y_ref, use_pca, x_test_mean, x_test_var):
        """q(X*) as dp_gp_lvm.py:246-262: mean from PCA of y_test or from the nearest training neighbour (L2 over the given
        columns) plus N(0, 0.01^2) noise, variances 1 — unless values are handed in."""
        n_t = y_test.shape[0]
        if x_test_mean is not None:
            init = np.asarray(x_test_mean.detach().cpu() if torch.is_tensor(x_test_mean) else x_test_mean, dtype=np.float64)
        elif use_pca:
            init = pca(y_test, num_latent_dimensions=num_latent_dims)
        else:
            d2 = ((y_ref[:, None, :] - y_test[None, :, :]) ** 2).sum(-1)            # [N x N*]  (utils/expressions.py:28-44)
            init = x_mean.detach().cpu().numpy()[np.argmin(d2, axis=0)] + \
                np.random.normal(scale=0.01, size=(n_t, num_latent_dims))
        if x_test_var is not None:
            var = np.asarray(x_test_var.detach().cpu() if torch.is_tensor(x_test_var) else x_test_var, dtype=np.float64)
        else:
            var = np.ones((n_t, num_latent_dims))
        xt_, st__ = _t(init), _t(var.reshape(n_t, num_latent_dims))
        # q(X*): the reference creates these two as NON-trainable tf.Variables (dp_gp_lvm.py:258-262), which is what
        # get_prediction_variables() hands to the test-time optimiser (test/frey_faces_prediction.py:165-171)
        pred_state['x_test_mean'] = register_variable(xt_, trainable=False)
        pred_state['x_test_var_raw'] = register_variable(torch.log(torch.expm1(st__)), trainable=False)
        return xt_, st__

    def _fhat_on(y_t, xt, st_, dims=None):
        """f_hat (fused ELBO, dpgp_elbo_fhat) of the first `dims` output dims on (y_t, q(X*)) with the trained kernel and
        inducing inputs, and KL(q(X*) || p(X*)); uses the mixed hyper-parameters of the last evaluate()."""
        dd = num_dimensions if dims is None else dims
        ws_t = ops.ElboWorkspace(dd, y_t.shape[0], num_inducing_points, num_latent_dims, precision, device)
        terms_t, sums, info = ops.elbo_fhat(y_t, x_u, xt, st_, buf['gamma'][:dd].contiguous(), buf['alpha'][:dd].contiguous(),
                                      buf['beta'][:dd].contiguous(), jitter=GP_DEFAULT_JITTER, prec=precision, workspace=ws_t)
        pred_state['terms'] = terms_t.clone()
        return sums[0].clone(), sums[1].clone()

    class DP_GP_LVM(Trainable):
        """Accessors as in the reference (dp_gp_lvm.py:161-231,502-508)."""
        raw = dict(x_mean=x_mean, x_var=x_var_raw, x_u=x_u, gamma_atoms=gamma_atoms_raw, alpha_atoms=sig_var_atoms_raw,
                   beta_atoms=beta_atoms_raw, **{'dp_' + k: v for k, v in dp_model.raw.items()})
        shard = (d_lo, d_hi)

        @property
        def assignments(self):
            return dp_model.assignments

        @property
        def dp(self):
            return dp_model

        @property
        def dp_atoms(self):
            return F.softplus(gamma_atoms_raw), F.softplus(sig_var_atoms_raw), F.softplus(beta_atoms_raw)

        @property
        def kernel(self):
            g, a, b = _mixed()
            return k_ard_rbf(gamma=g, alpha=a, beta=b)                           # batch size D (:105)

        @property
        def ard_weights(self):
            return _mixed()[0]

        @property
        def signal_variance(self):
            return _mixed()[1]

        @property
        def noise_precision(self):
            return _mixed()[2]

        @property
        def inducing_input(self):
            return x_u

        @property
        def q_x(self):
            return x_mean, torch.diag_embed(F.softplus(x_var_raw))               # mean [N x Q], covariance [N x Q x Q]

        @property
        def objective(self):
            """dp.objective - (f_hat - KL) - hyper-prior (dp_gp_lvm.py:154): 0-d fp64 device tensor."""
            return evaluate()[0].clone()

        @property
        def objective_terms(self):
            """(objective, f_hat, KL, DP objective, hyper-prior log-likelihood) of one evaluation, as a device tensor."""
            return evaluate().clone()

        @property
        def per_dimension_terms(self):
            """[D_local x 5] f_hat terms and the info flags of the last evaluation (0 fine, > 0 failed factorisation,
            DPGP_INFO_ILL_CONDITIONED = -2: fp32 Psi2 no longer trustworthy for this output dim)."""
            return workspace.terms, workspace.info

        @property
        def conditioning_guard(self):
            """[D_local] bound on what the rounding of an fp32 Psi2 can move each output dim's f_hat terms by (last
            evaluation; computed in every precision mode; flagged in info when > DPGP_GUARD_REL * N in mixed / f32)."""
            return workspace.guard

        @property
        def last_stage_b_form(self):
            """'mixed' (pair-tile form) or 'mixed_patch' (patch form): what the last gradients() of the training configuration
            (precision='f64', backward_precision='mixed') chose for the Psi2 term by the conditioning-guard bound; None otherwise."""
            return grad_state.get('stage_b_form')

        evaluate_ = staticmethod(evaluate)
        evaluate_graph = staticmethod(_evaluate_graph)

        @staticmethod
        def partial_pack():
            """(f_hat, DP objective) shares of the local output dims after one evaluation: the 2-vector that is
            sum-all-reduced when D is sharded (tests of the sharding arithmetic)."""
            evaluate()
            return buf['red'].clone()

        gradients = staticmethod(_gradients)
        optimise = staticmethod(_optimise)

        @property
        def prediction_terms(self):
            """[D* x 5] f_hat terms of the test points in the last predict_* call (the observed dims for missing data)."""
            return pred_state.get('terms')

        @staticmethod
        def _reference_defect(num_test_points):
            """What the reference's own prediction bounds contain on top of the bound (dp_gp_lvm.py:292, :409): there
            `tf.trace(...)` is [D] while psi_0_test and beta are [D x 1], so beta * (trace - psi_0) broadcasts to [D x D] and the
            reduce_sum adds  1/2 sum_ij beta_i (tr_j - psi0_i) - 1/2 sum_i beta_i (tr_i - psi0_i)."""
            terms_t = pred_state['terms']
            dd = terms_t.shape[0]
            al, be = buf['alpha'][:dd, 0], buf['beta'][:dd, 0]
            psi0 = al * num_test_points
            tr = 2.0 * terms_t[:, 2] / be + psi0                                  # term 2 = beta/2 (tr - alpha N*)
            return 0.5 * torch.sum(be[:, None] * (tr[None, :] - psi0[:, None])) - 0.5 * torch.sum(be * (tr - psi0))

        @staticmethod
        def predict_new_latent_variables(y_test, use_pca=False, x_test_mean=None, x_test_var=None, reference_compat=False):
            """Mirror of dp_gp_lvm.py:233-309: q(X*) for fully observed test data y_test [N* x D] and the prediction lower bound
                f_hat + f_hat_test - KL(q(X)) - KL(q(X*)),    test log-likelihood = f_hat_test - KL(q(X*)),
            (without the reference's broadcasting defect in its beta (trace - psi_0) term, dp_gp_lvm.py:292 — see
            oracle/gen_golden_predict.py; `reference_compat=True` returns the reference's own numbers, defect included)
            where f_hat_test is the SAME fused ELBO evaluated on (y_test, q(X*)) with the trained kernel and inducing inputs
            (one more dpgp_elbo_fhat call).  Returns (prediction_lower_bound, x_test_mean [N* x Q], x_test_covar [N* x Q x Q],
            test_log_likelihood) at the initial q(X*): nearest training neighbour + N(0, 0.01^2) noise, or PCA of y_test
            (`use_pca`), or the given x_test_mean / x_test_var (values; what a caller's optimiser of q(X*) passes back in)."""
            assert not sharded and world == 1, 'prediction paths run on one GPU'
            y_test = np.asarray(y_test, dtype=np.float64)
            num_test_points, test_dims = np.shape(y_test)
            assert test_dims == num_dimensions, \
                'Observed dimensionality for prediction must be equal to the dimensionality of the training data.'
            xt, st_ = _init_test_latents(y_test, np.asarray(y_train), use_pca, x_test_mean, x_test_var)
            out = evaluate().clone()                                                 # (objective, f_hat, KL, DP, hyper)
            f_hat_test, kl_test = _fhat_on(_t(y_test), xt, st_)
            lower_bound = out[1] + f_hat_test - out[2] - kl_test
            test_ll = f_hat_test - kl_test
            if reference_compat:                                                 # the reference's own numbers (see _reference_defect)
                extra = DP_GP_LVM._reference_defect(num_test_points)
                lower_bound, test_ll = lower_bound + extra, test_ll + extra
            return lower_bound, xt, torch.diag_embed(st_), test_ll

        @staticmethod
        def test_latent_gradients(y_test, x_test_mean, x_test_var):
            """d (f_hat_test - KL(q(X*))) / d (x_test_mean, x_test_var) with the trained model fixed — what TensorFlow's autograd
            gives a caller of the reference who optimises q(X*) on the bounds returned by predict_*; y_test [N* x Do], Do <= D
            (the first Do output dims).  Uses the backward pass of the fused ELBO (stages A and B) on the test points."""
            assert not sharded and world == 1, 'prediction paths run on one GPU'
            y_t = _t(np.asarray(y_test, dtype=np.float64))
            dd = y_t.shape[1]
            xt, st_ = _t(x_test_mean.detach().cpu().numpy() if torch.is_tensor(x_test_mean) else x_test_mean), \
                _t(x_test_var.detach().cpu().numpy() if torch.is_tensor(x_test_var) else x_test_var)
            evaluate()
            gam, al, be = buf['gamma'][:dd].contiguous(), buf['alpha'][:dd].contiguous(), buf['beta'][:dd].contiguous()
            ws_t = ops.ElboWorkspace(dd, y_t.shape[0], num_inducing_points, num_latent_dims, precision, device)
            ops.elbo_fhat(y_t, x_u, xt, st_, gam, al, be, jitter=GP_DEFAULT_JITTER, prec=precision, workspace=ws_t)
            gp, wk, gv, _, _ = ops.elbo_grad_chain(al, be, ws_t, jitter=GP_DEFAULT_JITTER, z=x_u, gamma=gam)
            dmu, ds, _, _ = ops.elbo_grad_psi(y_t, x_u, xt, st_, gam, al, gp, wk, gv, prec=precision)
            return dmu - xt, ds - 0.5 * (1.0 - 1.0 / st_)                        # minus the KL gradient (gp_expressions.py:10-24)

        @staticmethod
        def optimise_test_latents(y_test, num_iterations=200, learning_rate=0.01, use_pca=False, x_test_mean=None,
                                  x_test_var=None):
            """Adam on q(X*) (mean and softplus-parametrised variances) maximising f_hat_test - KL(q(X*)) for test points
            observed in their first Do output dims; returns (x_test_mean, x_test_var) to hand to predict_*."""
            y_test = np.asarray(y_test, dtype=np.float64)
            xt, st_ = _init_test_latents(y_test, np.asarray(y_train)[:, :y_test.shape[1]], use_pca, x_test_mean, x_test_var)
            raw = torch.log(torch.expm1(st_))
            opt = torch.optim.Adam([xt, raw], lr=learning_rate)
            for _ in range(num_iterations):
                sv = F.softplus(raw)
                g_mu, g_s = DP_GP_LVM.test_latent_gradients(y_test, xt, sv)
                xt.grad, raw.grad = -g_mu, -g_s * torch.sigmoid(raw)
                opt.step()
            return xt, F.softplus(raw)

        @staticmethod
        def predict_missing_data(y_test, use_pca=False, x_test_mean=None, x_test_var=None, reference_compat=False):
            """Mirror of dp_gp_lvm.py:311-500: y_test [N* x Do] holds the FIRST Do output dims of the test points; returns
            (missing_data_lower_bound, x_test_mean, x_test_covar, predicted_mean [N* x Du], predicted_covar [Du x N* x N*])
            for the remaining Du = D - Do dims at the initial q(X*) (see predict_new_latent_variables).  Composed of the
            library's operators (Psi statistics at q(X*), batched Cholesky / triangular solves) and plain fp64 GEMMs."""
            assert not sharded and world == 1, 'prediction paths run on one GPU'
            y_test = np.asarray(y_test, dtype=np.float64)
            num_test_points, num_observed_dims = np.shape(y_test)
            assert num_observed_dims < num_dimensions, \
                'Observed dimensionality for missing data scenario must be less than total ' \
                'dimensionality of training data.'
            do, m_ = num_observed_dims, num_inducing_points
            xt, st_ = _init_test_latents(y_test, np.asarray(y_train)[:, :do], use_pca, x_test_mean, x_test_var)
            out = evaluate().clone()
            gam, al, be = buf['gamma'], buf['alpha'][:, 0], buf['beta'][:, 0]
            f_hat_test, kl_test = _fhat_on(_t(y_test), xt, st_, dims=do)
            lower_bound = out[1] + f_hat_test - out[2] - kl_test
            if reference_compat:
                lower_bound = lower_bound + DP_GP_LVM._reference_defect(num_test_points)
            # predictive mean / covariance of the unobserved dims (:426-498), output dims do .. D-1 only
            gu, au, bu = gam[do:].contiguous(), al[do:].contiguous(), be[do:].contiguous()
            s_train = F.softplus(x_var_raw)
            psi_1 = ops.psi1(x_u, x_mean, s_train, gu, au)                          # [Du x N x M]
            psi_2 = ops.psi2(x_u, x_mean, s_train, gu, au)
            psi_1t = ops.psi1(x_u, xt, st_, gu, au)                                  # [Du x N* x M]
            psi_2t = ops.psi2(x_u, xt, st_, gu, au)
            k_uu = ops.ard_rbf_gram(x_u, None, gu, au, bu, include_noise=False, include_jitter=True, jitter=GP_DEFAULT_JITTER)
            l_uu, _ = ops.potrf_batched(k_uu)
            h = ops.trsm_batched(l_uu, psi_2)
            a_mat = bu[:, None, None] * ops.trsm_batched(l_uu, h.transpose(1, 2).contiguous()).transpose(1, 2) + \
                torch.eye(m_, dtype=TORCH_DTYPE, device=device)
            l_a, _ = ops.potrf_batched(a_mat.contiguous())
            c = ops.trsm_batched(l_a, ops.trsm_batched(l_uu, psi_1.transpose(1, 2).contiguous()))          # [Du x M x N]
            c_pred = ops.trsm_batched(l_a, ops.trsm_batched(l_uu, psi_1t.transpose(1, 2).contiguous()))   # [Du x M x N*]
            y_u = _t(np.asarray(y_train)[:, do:]).transpose(0, 1).contiguous()[:, :, None]                 # [Du x N x 1]
            cy = ops.matmul(c, y_u)                                                # [Du x M x 1]
            predicted_mean = (bu[:, None] * ops.matmul(c_pred.transpose(1, 2), cy)[:, :, 0]).transpose(0, 1)   # [N* x Du]
            eye = torch.eye(m_, dtype=TORCH_DTYPE, device=device).expand(num_dimensions - do, m_, m_).contiguous()
            l_uu_inv, l_a_inv = ops.trsm_batched(l_uu, eye), ops.trsm_batched(l_a, eye)
            ainv = ops.matmul(l_a_inv.transpose(1, 2), l_a_inv)                    # A^-1
            g = psi_2t - ops.matmul(psi_1t.transpose(1, 2), psi_1t)               # [Du x M x M]
            scale_yu = ops.matmul(ops.matmul(ops.matmul(l_uu_inv.transpose(1, 2), ops.matmul(ainv, l_uu_inv)),
                                                 psi_1.transpose(1, 2)), y_u)       # [Du x M x 1]
            yu_var = bu * bu * ops.matmul(scale_yu.transpose(1, 2), ops.matmul(g, scale_yu))[:, 0, 0]
            tr_term = torch.diagonal(ops.matmul(ops.matmul(l_uu_inv.transpose(1, 2), ops.matmul(eye - ainv, l_uu_inv)),
                                                  psi_2t), dim1=-2, dim2=-1).sum(-1)
            psi_0t = ops.psi0(num_test_points, au)[:, 0]
            predicted_covar = yu_var[:, None, None] + (psi_0t + 1.0 / bu + tr_term)[:, None, None] * \
                torch.eye(num_test_points, dtype=TORCH_DTYPE, device=device)
            return lower_bound, xt, torch.diag_embed(st_), predicted_mean, predicted_covar

    return DP_GP_LVM()


def dp_gp_lvm_t(y_train,
                num_latent_dims=GP_LVM_DEFAULT_LATENT_DIMENSIONS,
                num_inducing_points=GP_LVM_DEFAULT_NUM_INDUCING_POINTS,
                truncation_level=DP_DEFAULT_TRUNCATION_LEVEL,
                alpha_prior_params=DP_DEFAULT_ALPHA_PRIOR_PARAMS,
                mask_size=1,
                seed=0,
                device=None, precision=None, initial_values=None, _view_of_many=False, process_group=None):
    """
    Over-T formulation — mirror of the reference's ``dp_gp_lvm_t`` factory (src/models/dp_gp_lvm.py:513-676), SURVEY.md
    §8(f) row 3: the kernel batch is the T atoms, the mixture weights phi [T x D] enter outside the kernel, so an evaluation
    needs T Psi2's instead of D.  A different objective from ``dp_gp_lvm`` away from equal atoms (equal at the reference's
    initialisation, test/unittests/dpgplvm_unitttests.py:544-548).  Objective and gradients, composed of the library's
    operators in the B_t = K_t + beta_t Psi2_t algebra of DESIGN.md §2:

        Psi1 [T,N,M], Psi2 [T,M,M], K_uu [T,M,M]       dpgp_psi1 / dpgp_psi2 / dpgp_ard_rbf_gram      (:611-620)
        L_K, L_B = chol(K), chol(K + beta Psi2)         dpgp_potrf_batched                             (:620,633)
        <K^-1, Psi2> by two triangular solves           dpgp_trsm_batched                              (:622-629)
        V = Psi1^T Y  [T,M,D]                           dpgp_gemm_strided_f64 (fp64 MFMA, ops.matmul)
        C = L_B^-1 V, 1/2 sum_td phi_td beta_t^2 |C_td|^2   dpgp_trsm_batched                          (:638-658)

    precision: 'f64', or 'mixed' = the Psi statistics in fp32 (f16-split MFMA kernels), everything after them in fp64.
    process_group: a torch.distributed group -> the D output dims are sharded over its ranks as in ``dp_gp_lvm``
    (shard_bounds): the T-atom chain is replicated (it does not depend on D), V = Psi1^T Y, the solves and every sum over d
    run on the local columns; an evaluation exchanges ONE scalar (the local f_hat), a gradient evaluation ONE packed
    all-reduce of all raw-variable gradients plus the trouble flag of optimise().
    """
    num_samples, num_dimensions = np.shape(y_train)
    # (_view_of_many: one view of a multi-view model, whose own check is against the views' total dimensionality)
    assert 0 < num_latent_dims < num_dimensions or _view_of_many, \
        'Number of latent dimensions must be postive and less than the dimensionality of the observed data.'
    assert 0 < num_inducing_points <= num_samples, \
        'Number of inducing points must be positive and less than or equal to the number of observations in the ' \
        'observed data.'
    assert 0 < truncation_level <= min(num_samples, num_dimensions), \
        'The truncation level must be positive and less than or equal to the dimensionality of the observed data and ' \
        'less than or equal to the number of observations.'
    assert isinstance(seed, int) and seed >= 0, 'Seed must be a 32-bit unsigned integer, i.e., 0 <= seed <= 2^32 - 1.'
    precision = 'f64' if precision is None else precision     # (default: the reference's arithmetic, as in dp_gp_lvm)
    assert precision in ('mixed', 'f64'), "precision must be 'mixed' or 'f64'"
    np.random.seed(seed=seed)
    device = default_device() if device is None else torch.device(device)
    iv = dict(initial_values or {})

    def _t(a):
        return torch.as_tensor(np.asarray(a, dtype=np.float64), dtype=TORCH_DTYPE, device=device).contiguous()

    def _raw_pos(name, init, shape):
        if name in iv:
            return _t(np.log(np.expm1(np.asarray(iv[name], dtype=np.float64).reshape(shape))))
        return create_positive_variable(init, shape, device=device)

    x_init = np.asarray(iv['x_mean'], dtype=np.float64) if 'x_mean' in iv else \
        pca(np.asarray(y_train), num_latent_dimensions=num_latent_dims)
    x_mean = _t(x_init)
    x_var_raw = _raw_pos('x_var', 1.0, (num_samples, num_latent_dims))
    x_u = _t(iv['x_u']) if 'x_u' in iv else \
        _t(np.random.permutation(x_init)[:num_inducing_points] +
           np.random.normal(loc=0.0, scale=0.01, size=(num_inducing_points, num_latent_dims)))
    dp_model = dirichlet_process(num_samples=num_dimensions, alpha_prior_params=alpha_prior_params,
                                 truncation_level=truncation_level, mask_size=mask_size, device=device)
    for key, name in (('logits', 'phi_logits'), ('gamma_1', 'gamma_1'), ('gamma_2', 'gamma_2')):
        if name in iv:
            v = np.asarray(iv[name], dtype=np.float64)
            dp_model.raw[key].copy_(_t(v if key == 'logits' else np.log(np.expm1(v))).reshape(dp_model.raw[key].shape))
    for i, name in enumerate(('w_1', 'w_2')):
        if name in iv:
            dp_model.raw['w'][i] = float(np.log(np.expm1(float(iv[name]))))
    gamma_atoms_raw = _raw_pos('gamma_atoms', GP_INIT_GAMMA, (truncation_level, num_latent_dims))
    sig_var_atoms_raw = _raw_pos('alpha_atoms', GP_INIT_ALPHA, (truncation_level, 1))
    beta_atoms_raw = _raw_pos('beta_atoms', GP_INIT_BETA, (truncation_level, 1))
    sharded = process_group is not None
    if sharded:
        import torch.distributed as dist
        rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
        assert world <= num_dimensions, 'more ranks than output dimensions'
    else:
        rank, world = 0, 1
    d_lo, d_hi = shard_bounds(num_dimensions, rank, world)
    y_dev = _t(np.asarray(y_train)[:, d_lo:d_hi])                              # the local columns
    yy = torch.sum(y_dev * y_dev, dim=0)                                       # [D_local]
    from ..distributions.log_normal import log_pdf as log_normal_log_pdf
    from ..distributions.beta import entropy as beta_dist_entropy
    from ..distributions.gamma import entropy as gamma_dist_entropy
    from ..distributions.multinomial import entropy as multinomial_dist_entropy
    s_1, s_2 = dp_model.prior
    n_, d_, m_ = num_samples, d_hi - d_lo, num_inducing_points             # (d_: the LOCAL output dims; the sums over d are local)
    mp_ = 16 * ((m_ + 15) // 16)
    last_info = [torch.zeros((), dtype=torch.int32, device=device)]
    # the fused forward pass (csrc/elbo.hip, dpgp_elbo_fhat_t): M <= 128 and at least T local output dims; otherwise, and
    # whenever gradients are wanted, f_hat is composed of the library's operators (_FHatT)
    fused_t = ops.ElboTWorkspace(truncation_level, d_, n_, m_, num_latent_dims, precision, device) \
        if (ops.elbo_fhat_t_supported(m_) and d_ >= truncation_level and device.type == 'cuda'
            and os.environ.get('DPGP_FUSED_T', '1') != '0') else None       # (DPGP_FUSED_T=0: cross-checks only)

    def _chain(x_u_, x_mean_, s_, gat, aat, bat):
        """Psi statistics and the Cholesky factors of the T atoms (library operators)."""
        pt = torch.float32 if precision == 'mixed' else TORCH_DTYPE
        cast = lambda a: a.to(pt).contiguous()
        psi_1 = ops.psi1(cast(x_u_), cast(x_mean_), cast(s_), cast(gat), cast(aat)).to(TORCH_DTYPE)   # [T x N x M]
        psi_2 = ops.psi2(cast(x_u_), cast(x_mean_), cast(s_), cast(gat), cast(aat)).to(TORCH_DTYPE)   # [T x M x M]
        k_uu = ops.ard_rbf_gram(x_u_, None, gat, aat, bat, include_noise=False, include_jitter=True,
                                jitter=GP_DEFAULT_JITTER)                                              # [T x M x M]
        l_k, info_k = ops.potrf_batched(k_uu)
        l_b, info_b = ops.potrf_batched(k_uu + bat[:, None, None] * psi_2)
        last_info[0] = torch.maximum(info_k.abs().max(), info_b.abs().max())
        return psi_1, psi_2, k_uu, l_k, l_b

    def _fhat_forward(x_mean_, s_, x_u_, gat, aat, bat, phit, for_backward=True):
        """f_hat of dp_gp_lvm.py:617-667 from the library's operators; returns (f_hat, what the backward pass needs).
        for_backward: K^-1 and B^-1 are formed once here (two inverses of the triangular factors + two products) and serve the trace,
        the solve against V and the whole backward pass — six latency-bound batched solves of T small matrices otherwise."""
        psi_1, psi_2, k_uu, l_k, l_b = _chain(x_u_, x_mean_, s_, gat, aat, bat)
        v = ops.matmul(psi_1.transpose(1, 2), y_dev)                       # [T x M x D]
        k_inv = b_inv = None
        if for_backward:
            li = ops.tril_inverse_batched(l_k)                             # (M a multiple of 128: the persistent solve, one launch)
            k_inv = ops.matmul(li.transpose(1, 2), li)
            li = ops.tril_inverse_batched(l_b)
            b_inv = ops.matmul(li.transpose(1, 2), li)
            tr = torch.sum(k_inv * psi_2, dim=(1, 2))                        # tr(L^-1 Psi2 L^-T) = <K^-1, Psi2>
            c = ops.matmul(li, v)                                          # L_B^-1 V
        else:
            h = ops.trsm_batched(l_k, psi_2)
            tr = torch.diagonal(ops.trsm_batched(l_k, h.transpose(1, 2).contiguous()), dim1=-2, dim2=-1).sum(-1)
            c = ops.trsm_batched(l_b, v)
        logdet = torch.log(torch.diagonal(l_b, dim1=-2, dim2=-1)).sum(-1) - \
            torch.log(torch.diagonal(l_k, dim1=-2, dim2=-1)).sum(-1)
        quad = bat[:, None] ** 2 * torch.sum(c * c, dim=1)                   # [T x D]
        per_t = 0.5 * (n_ * torch.log(bat) + bat * (tr - n_ * aat)) - logdet
        f_hat = -0.5 * n_ * d_ * np.log(2.0 * np.pi) + torch.sum(phit * per_t[:, None]) \
            - 0.5 * torch.sum(phit * bat[:, None] * yy[None, :]) + 0.5 * torch.sum(phit * quad)
        return f_hat, (x_mean_, s_, x_u_, gat, aat, bat, phit, psi_2, k_uu, k_inv, b_inv, v, per_t, quad, tr)

    def _fhat_backward(saved):
        """d f_hat / d (x_mean, S, x_u, gamma atoms, alpha atoms, beta atoms, phi^T): the adjoints of the per-atom dense algebra
        (torch on [T, M, M] arrays, T small) and the library's streaming stage B (dpgp_elbo_grad_psi_ex: Psi2 term on the matrix
        pipe, Psi1 term with the full adjoint beta^2 Y diag(phi_t) W_t^T, K_uu term)."""
        x_mean_, s_, x_u_, gat, aat, bat, phit, p2, k_uu, k_inv, b_inv, v, per_t, quad, tr = saved
        eye = torch.eye(m_, dtype=TORCH_DTYPE, device=device)
        st = phit.sum(dim=1)[:, None, None]                                  # s_t = sum_d phi_td
        be = bat[:, None, None]
        w = ops.matmul(b_inv, v)                                           # [T x M x D]
        wphi = w * phit[:, None, :]
        vwphi = torch.sum(v * wphi, dim=(1, 2))                              # sum_d phi_td v_td^T B^-1 v_td
        gb = -0.5 * st * b_inv - 0.5 * be * be * ops.matmul(wphi, w.transpose(1, 2))
        x = ops.matmul(ops.matmul(k_inv, p2), k_inv)
        gk = 0.5 * st * k_inv - 0.5 * st * be * x + gb
        gp = 0.5 * st * be * k_inv + be * gb
        wk = gk * (k_uu - GP_DEFAULT_JITTER * eye)
        g1 = be * be * ops.matmul(y_dev, wphi.transpose(1, 2))             # [T x N x M] adjoint of Psi1
        pad2 = (0, mp_ - m_, 0, mp_ - m_)
        dmu, ds, dz, dgam = ops.elbo_grad_psi(None, x_u_, x_mean_, s_, gat, aat,
                                              torch.nn.functional.pad(gp, pad2).contiguous(),
                                              torch.nn.functional.pad(wk, pad2).contiguous(), None,
                                              prec='mixed_patch' if precision == 'f64' else 'mixed',
                                              g_psi1=torch.nn.functional.pad(g1, (0, mp_ - m_)).contiguous())
        s_k, s_p = wk.sum(dim=(1, 2)), (gp * p2).sum(dim=(1, 2))
        s_gbp = (gb * p2).sum(dim=(1, 2))
        stv = st[:, 0, 0]
        d_alpha = -0.5 * bat * n_ * stv + (s_k + 2.0 * s_p + bat * bat * vwphi) / aat
        d_beta = stv * (0.5 * n_ / bat + 0.5 * (tr - aat * n_)) - 0.5 * torch.sum(phit * yy[None, :], dim=1) \
            + bat * vwphi + s_gbp
        d_phit = per_t[:, None] - 0.5 * bat[:, None] * yy[None, :] + 0.5 * quad
        return dmu, ds, dz, dgam, d_alpha, d_beta, d_phit

    class _FHatT(torch.autograd.Function):
        """f_hat as a differentiable function of (x_mean, S, x_u, gamma / alpha / beta atoms, phi^T) for torch autograd (the
        cross-check path of the gradients, DPGP_T_AUTOGRAD=1, and the objective's graph): _fhat_forward / _fhat_backward."""

        @staticmethod
        def forward(ctx, x_mean_, s_, x_u_, gat, aat, bat, phit):
            f_hat, saved = _fhat_forward(x_mean_, s_, x_u_, gat, aat, bat, phit, for_backward=any(ctx.needs_input_grad))
            ctx.save_for_backward(*[a for a in saved if a is not None])
            ctx.with_inverses = saved[9] is not None
            return f_hat

        @staticmethod
        def backward(ctx, g_out):
            assert ctx.with_inverses
            return tuple(g_out * g for g in _fhat_backward(ctx.saved_tensors))

    def _dp_objective(phi, g1, g2, w1, w2):
        """-ELBO of the DP (dirichlet_process.py:64-88) as a function of its arguments (for autograd)."""
        t_ = truncation_level
        dg12 = torch.digamma(g1 + g2)
        tail = (torch.flip(torch.cumsum(torch.flip(phi, [-1]), dim=-1), [-1]) - phi)[:, 0:-1]
        ev_z = torch.sum(phi[:, 0:-1] * (torch.digamma(g1) - dg12) + tail * (torch.digamma(g2) - dg12))
        ev_v = (t_ - 1.0) * (torch.digamma(w1) - torch.log(w2)) + ((w1 / w2) - 1.0) * torch.sum(torch.digamma(g2) - dg12)
        ev_a = s_1 * np.log(s_2) - float(torch.lgamma(torch.tensor(s_1, dtype=TORCH_DTYPE))) + \
            (s_1 - 1.0) * (torch.digamma(w1) - torch.log(w2)) - s_2 * (w1 / w2)
        return -(ev_z + ev_v + ev_a + torch.sum(multinomial_dist_entropy(phi)) + torch.sum(beta_dist_entropy(g1, g2)) +
                 gamma_dist_entropy(w1, w2))

    def _objective_of(r):
        """(objective, f_hat, KL, DP objective, hyper-prior) from a dict of raw variables (torch graph when they require grad)."""
        gat, aat, bat = F.softplus(r['gamma_atoms']), F.softplus(r['alpha_atoms'])[:, 0], F.softplus(r['beta_atoms'])[:, 0]
        s = F.softplus(r['x_var'])
        phi = torch.softmax(r['dp_logits'], dim=-1)
        if mask_size != 1:
            phi = torch.repeat_interleave(phi, mask_size, dim=0)
        phit = phi[d_lo:d_hi].transpose(0, 1).contiguous()                                        # local dims
        mu = r['x_mean']
        f_hat = _FHatT.apply(mu, s, r['x_u'], gat, aat, bat, phit)
        kl = 0.5 * (torch.sum(mu * mu) + torch.sum(s - torch.log(s)) - mu.shape[0] * mu.shape[1])     # gp_expressions.py:10-24
        hyper = torch.sum(log_normal_log_pdf(gat)) + torch.sum(log_normal_log_pdf(aat)) + torch.sum(log_normal_log_pdf(bat))
        w = F.softplus(r['dp_w'])
        dp_obj = _dp_objective(phi, F.softplus(r['dp_gamma_1']), F.softplus(r['dp_gamma_2']), w[0], w[1])
        return torch.stack([dp_obj - (f_hat - kl) - hyper, f_hat, kl, dp_obj, hyper])

    raw_vars = dict(x_mean=x_mean, x_var=x_var_raw, x_u=x_u, gamma_atoms=gamma_atoms_raw, alpha_atoms=sig_var_atoms_raw,
                    beta_atoms=beta_atoms_raw, **{'dp_' + k: v for k, v in dp_model.raw.items()})

    def _exchange(out):
        """Sharded: out[1] is the f_hat of the local output dims; everything else is replicated."""
        if sharded:
            f_tot = out[1].clone()
            dist.all_reduce(f_tot, op=dist.ReduceOp.SUM, group=process_group)
            out = torch.stack([out[3] - (f_tot - out[2]) - out[4], f_tot, out[2], out[3], out[4]])
        return out

    tbuf = {}
    if fused_t is not None:
        t_, q_ = truncation_level, num_latent_dims
        tbuf = dict(s=torch.empty((n_, q_), dtype=TORCH_DTYPE, device=device),
                    phi=torch.empty((d_, t_), dtype=TORCH_DTYPE, device=device),
                    atoms=torch.empty(t_ * q_ + 2 * t_, dtype=TORCH_DTYPE, device=device),
                    scal=torch.zeros(_lib.lib().dpgp_model_scal_count(d_), dtype=TORCH_DTYPE, device=device),
                    red=torch.zeros(2, dtype=TORCH_DTYPE, device=device), out=torch.zeros(5, dtype=TORCH_DTYPE, device=device))

    def _evaluate_fused(finish=True):
        """dpgp_model_prepare_t -> dpgp_elbo_fhat_t (the model-level tail rides in its last launch) -> [one packed all-reduce
        and dpgp_model_finalize when D is sharded]: eleven launches, no host arithmetic (round 2: ~200 launches of library
        operators and torch element-wise kernels)."""
        lib, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
        r = dp_model.raw
        t_, q_ = truncation_level, num_latent_dims
        _lib.check(lib.dpgp_model_prepare_t(
            d_, t_, q_, n_, d_lo, mask_size, r['logits'].data_ptr(), gamma_atoms_raw.data_ptr(), sig_var_atoms_raw.data_ptr(),
            beta_atoms_raw.data_ptr(), x_var_raw.data_ptr(), r['gamma_1'].data_ptr(), r['gamma_2'].data_ptr(), r['w'].data_ptr(),
            s_1, s_2, 1 if rank == 0 else 0, tbuf['s'].data_ptr(), tbuf['phi'].data_ptr(), tbuf['atoms'].data_ptr(),
            tbuf['scal'].data_ptr(), st), 'dpgp_model_prepare_t')
        at = tbuf['atoms']
        _, _, sums_t, info_t = ops.elbo_fhat_t(
            y_dev, yy, x_u, x_mean, tbuf['s'], at[:t_ * q_].view(t_, q_), at[t_ * q_:t_ * q_ + t_], at[t_ * q_ + t_:],
            tbuf['phi'].t(), jitter=GP_DEFAULT_JITTER, prec=precision, workspace=fused_t,
            model_tail=(tbuf['scal'], tbuf['red'], None if sharded else tbuf['out']))
        last_info[0] = info_t                 # (per atom; cholesky_info reduces it when asked: no launch per evaluation)
        return _finish_fused() if finish else tbuf['out']

    def _finish_fused():
        if sharded:
            lib, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
            dist.all_reduce(tbuf['red'], op=dist.ReduceOp.SUM, group=process_group)     # (f_hat, DP objective): 2 fp64 scalars
            _lib.check(lib.dpgp_model_finalize(tbuf['red'].data_ptr(), fused_t.sums[1:2].data_ptr(),
                                               tbuf['scal'][1:2].data_ptr(), tbuf['out'].data_ptr(), st), 'dpgp_model_finalize')
        return tbuf['out']

    def evaluate():
        with torch.no_grad():
            out = _evaluate_fused() if fused_t is not None else _exchange(_objective_of(raw_vars))
        return out, last_info[0]

    graph = {}

    def _objective_terms_graph():
        """The same evaluation replayed from a HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm): this composed objective
        is ~200 small launches, i.e. launch-bound; capturing them once removes the per-launch host cost.  The raw variables
        are read in place, so updates by an optimiser are seen by the replay."""
        if 'g' not in graph:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):
                    evaluate()
            cur.wait_stream(side)
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_):
                with torch.no_grad():
                    graph['out'] = _evaluate_fused(finish=False) if fused_t is not None else _objective_of(raw_vars)
            graph['g'] = g_
        graph['g'].replay()
        if fused_t is not None:
            return _finish_fused().clone()
        return _exchange(graph['out'].clone())              # (the all-reduce of a sharded model runs eagerly behind the replay)

    def _local_flat_autograd():
        """This rank's packed gradient (all raw variables, then the trouble flag) — everything in front of the exchange — by torch
        autograd over the whole objective (~540 launches: the DP / KL / prior terms and their derivatives are ~400 of them)."""
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in raw_vars.items()}
        terms = _objective_of(leaves)
        # sharded: this rank's share of the objective = -(local f_hat) + (replicated terms) / world; the shares sum to it
        obj = terms[0] if not sharded else -terms[1] + (terms[3] + terms[2] - terms[4]) / world
        grads = torch.autograd.grad(obj, list(leaves.values()), allow_unused=True)
        grads = [torch.zeros_like(v) if g is None else g for v, g in zip(leaves.values(), grads)]
        flat = torch.cat([g.reshape(-1) for g in grads] + [torch.zeros(1, dtype=TORCH_DTYPE, device=device)])
        # trouble flag (failed factorisation / non-finite local gradient), reduced with the gradients: a collective decision
        flat[-1] = ((last_info[0] != 0) | ~torch.isfinite(flat[:-1]).all()).to(TORCH_DTYPE)
        return flat

    hbuf = {}

    def _local_flat():
        """The same with the model-level ends in the HIP library: dpgp_model_prepare_t (softplus / softmax transforms, phi), f_hat and
        its derivatives with respect to (x_mean, S, x_u, atoms, phi) from the library's operators (_fhat_forward / _fhat_backward),
        dpgp_model_backward_t (chain rule to the raw variables, KL, DP objective, hyper-prior): ~200 launches instead of ~540.
        Sharded: the replicated terms are added on rank 0 only (add_constants); the ranks' packed gradients sum to the gradient."""
        if os.environ.get('DPGP_T_AUTOGRAD', '0') == '1':
            return _local_flat_autograd()
        lib, st = _lib.lib(), torch.cuda.current_stream().cuda_stream
        r = dp_model.raw
        t_, q_ = truncation_level, num_latent_dims
        if not hbuf:
            f64 = dict(dtype=TORCH_DTYPE, device=device)
            hbuf.update(s=torch.empty((n_, q_), **f64), phi=torch.empty((d_, t_), **f64), atoms=torch.empty(t_ * q_ + 2 * t_, **f64),
                        scal=torch.zeros(_lib.lib().dpgp_model_scal_count(d_), **f64))
        with torch.no_grad():
            _lib.check(lib.dpgp_model_prepare_t(
                d_, t_, q_, n_, d_lo, mask_size, r['logits'].data_ptr(), gamma_atoms_raw.data_ptr(), sig_var_atoms_raw.data_ptr(),
                beta_atoms_raw.data_ptr(), x_var_raw.data_ptr(), r['gamma_1'].data_ptr(), r['gamma_2'].data_ptr(), r['w'].data_ptr(),
                s_1, s_2, 1 if rank == 0 else 0, hbuf['s'].data_ptr(), hbuf['phi'].data_ptr(), hbuf['atoms'].data_ptr(),
                hbuf['scal'].data_ptr(), st), 'dpgp_model_prepare_t')
            at = hbuf['atoms']
            gat, aat, bat = at[:t_ * q_].view(t_, q_), at[t_ * q_:t_ * q_ + t_], at[t_ * q_ + t_:]
            _, saved = _fhat_forward(x_mean, hbuf['s'], x_u, gat, aat, bat, hbuf['phi'].t().contiguous())
            dmu, ds, dz, dgam, d_alpha, d_beta, d_phit = _fhat_backward(saved)
            dab = torch.stack([d_alpha, d_beta], dim=1).contiguous()
            dphi = d_phit.t().contiguous()
            sizes = {k: v.numel() for k, v in raw_vars.items()}
            flat = torch.zeros(sum(sizes.values()) + 1, dtype=TORCH_DTYPE, device=device)
            parts, o = {}, 0
            for k, nel in sizes.items():
                parts[k] = flat[o:o + nel]
                o += nel
            rows = r['logits'].shape[0]
            _lib.check(lib.dpgp_model_backward_t(
                d_, t_, q_, n_, m_, d_lo, mask_size, rows, r['logits'].data_ptr(), gamma_atoms_raw.data_ptr(),
                sig_var_atoms_raw.data_ptr(), beta_atoms_raw.data_ptr(), x_var_raw.data_ptr(), r['gamma_1'].data_ptr(),
                r['gamma_2'].data_ptr(), r['w'].data_ptr(), x_mean.data_ptr(), hbuf['phi'].data_ptr(), s_1, s_2, 1 if rank == 0 else 0,
                dmu.contiguous().data_ptr(), ds.contiguous().data_ptr(), dz.contiguous().data_ptr(), dgam.contiguous().data_ptr(),
                dab.data_ptr(), dphi.data_ptr(), parts['x_mean'].data_ptr(), parts['x_var'].data_ptr(), parts['x_u'].data_ptr(),
                parts['dp_logits'].data_ptr(), parts['dp_gamma_1'].data_ptr(), parts['dp_gamma_2'].data_ptr(), parts['dp_w'].data_ptr(),
                parts['gamma_atoms'].data_ptr(), parts['alpha_atoms'].data_ptr(), parts['beta_atoms'].data_ptr(), st),
                'dpgp_model_backward_t')
            flat[-1] = ((last_info[0] != 0) | ~torch.isfinite(flat[:-1]).all()).to(TORCH_DTYPE)
        return flat

    grad_graph = {}

    def _gradients(graph=None):
        """d objective / d raw variable for all raw variables: torch autograd around the library-backed f_hat (above).
        graph=True (the default of optimise(); DPGP_GRAPH_T=0 turns it off): the local part — ~250 launches of library operators,
        element-wise kernels and their autograd, host-bound at ~4.6 ms per call whatever the problem size — is captured ONCE into a
        HIP graph (torch.cuda.CUDAGraph) and replayed; the raw variables are read in place, so optimiser updates are seen."""
        use_graph = bool(graph) and os.environ.get('DPGP_GRAPH_T', '1') != '0'
        if use_graph:
            if 'g' not in grad_graph:
                cur = torch.cuda.current_stream()
                side = torch.cuda.Stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    for _ in range(2):
                        _local_flat()
                cur.wait_stream(side)
                g_ = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_):
                    grad_graph['flat'] = _local_flat()
                grad_graph['g'] = g_
            grad_graph['g'].replay()
            flat = grad_graph['flat'].clone()
        else:
            flat = _local_flat()
        if sharded:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)       # ONE packed exchange
        grad_flag[0] = flat[-1]
        out, o = {}, 0
        for (k, v) in raw_vars.items():
            out[k] = flat[o:o + v.numel()].reshape(v.shape)
            o += v.numel()
        return out

    grad_flag = [torch.zeros((), dtype=TORCH_DTYPE, device=device)]

    def _optimise(num_iterations, learning_rate=0.01, callback=None):
        """Adam on the raw variables (the reference: tf.train.AdamOptimizer(...).minimize(objective))."""
        opt = torch.optim.Adam(list(raw_vars.values()), lr=learning_rate)
        for it in range(num_iterations):
            g = _gradients(graph=True)
            # potrf replaces a failing pivot by 1 and goes on: a failed factorisation would give finite, meaningless
            # gradients.  The reference's tf.cholesky raises; so does this.
            bad = (grad_flag[0] != 0) | ~torch.stack([torch.isfinite(v).all() for v in g.values()]).all()
            if bool(bad):
                raise FloatingPointError('iteration %d: failed Cholesky factorisation or non-finite gradient (precision=%r); '
                                         'use precision="f64"' % (it, precision))
            for k, p_ in raw_vars.items():
                p_.grad = g[k].reshape(p_.shape)
            opt.step()
            if callback is not None:
                callback(it)

    class DP_GP_LVM_T(Trainable):
        """Accessors as in the reference (dp_gp_lvm.py:679-740); the kernel has batch size T here."""
        raw = dict(x_mean=x_mean, x_var=x_var_raw, x_u=x_u, gamma_atoms=gamma_atoms_raw, alpha_atoms=sig_var_atoms_raw,
                   beta_atoms=beta_atoms_raw, **{'dp_' + k: v for k, v in dp_model.raw.items()})

        @property
        def assignments(self):
            return dp_model.assignments

        @property
        def dp(self):
            return dp_model

        @property
        def dp_atoms(self):
            return F.softplus(gamma_atoms_raw), F.softplus(sig_var_atoms_raw), F.softplus(beta_atoms_raw)

        @property
        def kernel(self):
            return k_ard_rbf(gamma=F.softplus(gamma_atoms_raw), alpha=F.softplus(sig_var_atoms_raw),
                             beta=F.softplus(beta_atoms_raw))

        @property
        def inducing_input(self):
            return x_u

        @property
        def q_x(self):
            return x_mean, torch.diag_embed(F.softplus(x_var_raw))

        @property
        def objective(self):
            return evaluate()[0][0].clone()

        @property
        def objective_terms(self):
            # (a copy: the fused path evaluates into one persistent buffer, which the next evaluation overwrites)
            return evaluate()[0].clone()

        @property
        def cholesky_info(self):
            info = evaluate()[1]
            return info.abs().max() if info.dim() else info

        shard = (d_lo, d_hi)

        objective_terms_graph = staticmethod(_objective_terms_graph)
        gradients = staticmethod(_gradients)
        optimise = staticmethod(_optimise)

    return DP_GP_LVM_T()
