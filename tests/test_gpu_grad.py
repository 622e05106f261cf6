"""GPU tests of the backward pass of the fused ELBO (first version), against the autograd gradient oracle
(oracle/dpgp_oracle_torch.py, pinned by tests/golden/grad_ref_*.npz = gradients of the reference's own objective).
Stage A: dpgp_elbo_grad_chain — adjoints of the per-output dense algebra (dp_gp_lvm.py:108-145 differentiated).
Tolerances: fp64 chain on fp64 psi statistics 1e-8; mixed (fp32 psi statistics) 2e-4 relative to the largest entry."""
import numpy as np
import pytest
import torch

from conftest import golden
from dp_gp_lvm_amd import ops
from oracle import dpgp_oracle_torch as ot

pytestmark = pytest.mark.gpu


def softplus(x):
    return np.logaddexp(0.0, x)


def point(name):
    g = golden(name)
    e = np.exp(g['dp_logits'] - g['dp_logits'].max(axis=1, keepdims=True))
    phi = e / e.sum(axis=1, keepdims=True)
    return dict(y=g['y'], z=g['x_u'], mu=g['x_mean'], s=softplus(g['x_var_raw']), gamma=phi @ softplus(g['gamma_atoms_raw']),
                alpha=(phi @ softplus(g['alpha_atoms_raw']))[:, 0], beta=(phi @ softplus(g['beta_atoms_raw']))[:, 0])


@pytest.mark.parametrize('fixture', ['grad_ref_40_6_12_3_T4', 'grad_ref_60_10_15_4_T5'])
@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_chain_adjoints(dev, fixture, prec):
    p = point(fixture)
    ref = ot.chain_adjoints(p['y'], p['z'], p['mu'], p['s'], p['gamma'], p['alpha'], p['beta'])
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    n, d = p['y'].shape
    m, q = p['z'].shape
    w = ops.ElboWorkspace(d, n, m, q, prec, dev)
    ops.elbo_fhat(t(p['y']), t(p['z']), t(p['mu']), t(p['s']), t(p['gamma']), t(p['alpha']), t(p['beta']), prec=prec, workspace=w)
    gp, wk, gv, dab, info = ops.elbo_grad_chain(t(p['alpha']), t(p['beta']), w)
    assert int(info.abs().max()) == 0
    tol = 1e-8 if prec == 'f64' else 2e-4
    low = np.tril(np.ones((m, m), dtype=bool))

    def close(got, want, what):
        np.testing.assert_allclose(got, want, rtol=tol, atol=tol * np.abs(want).max(), err_msg=what)
    close(gp.cpu().numpy()[:, :m, :m][:, low], ref['g_psi2'][:, low], 'd f_hat / d Psi2')
    k_scaled = ref['k_uu'] - 1e-8 * np.eye(m)
    close(wk.cpu().numpy()[:, :m, :m][:, low], (ref['g_kuu'] * k_scaled)[:, low], '(d f_hat / d K_uu) .* (K_uu - jitter I)')
    close(gv.cpu().numpy()[:, :m], ref['g_v'], 'd f_hat / d Psi1^T y')
    close(dab.cpu().numpy()[:, 0], ref['d_alpha'], 'd f_hat / d alpha')
    close(dab.cpu().numpy()[:, 1], ref['d_beta'], 'd f_hat / d beta')


@pytest.mark.parametrize('fixture', ['grad_ref_40_6_12_3_T4', 'grad_ref_60_10_15_4_T5'])
@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_fhat_gradients_wrt_kernel_inputs(dev, fixture, prec):
    """Stages A + B: d f_hat / d (mu, S, z, gamma, alpha, beta) against autograd of the oracle."""
    p = point(fixture)
    ref = ot.fhat_input_gradients(p['y'], p['z'], p['mu'], p['s'], p['gamma'], p['alpha'], p['beta'])
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    n, d = p['y'].shape
    m, q = p['z'].shape
    w = ops.ElboWorkspace(d, n, m, q, prec, dev)
    args = [t(p[k]) for k in ('y', 'z', 'mu', 's', 'gamma', 'alpha', 'beta')]
    ops.elbo_fhat(*args, prec=prec, workspace=w)
    gp, wk, gv, dab, info = ops.elbo_grad_chain(args[5], args[6], w)
    dmu, ds, dz, dg = ops.elbo_grad_psi(args[0], args[1], args[2], args[3], args[4], args[5], gp, wk, gv, prec=prec)
    tol = 1e-8 if prec == 'f64' else 3e-4

    def close(got, want, what):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=tol, atol=tol * np.abs(want).max(), err_msg=what)
    close(dmu, ref['d_mu'], 'd f_hat / d mu')
    close(ds, ref['d_s'], 'd f_hat / d S')
    close(dz, ref['d_z'], 'd f_hat / d z')
    close(dg, ref['d_gamma'], 'd f_hat / d gamma')
    close(dab[:, 0], ref['d_alpha'], 'd f_hat / d alpha')
    close(dab[:, 1], ref['d_beta'], 'd f_hat / d beta')


REF2RAW = dict(x_mean='x_mean', x_var_raw='x_var', x_u='x_u', dp_logits='dp_logits', gamma_1_raw='dp_gamma_1',
               gamma_2_raw='dp_gamma_2', gamma_atoms_raw='gamma_atoms', alpha_atoms_raw='alpha_atoms', beta_atoms_raw='beta_atoms')


def build_model(g, dev, prec, **kw):
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    sp = softplus
    return dp_gp_lvm(g['y'], num_latent_dims=g['x_mean'].shape[1], num_inducing_points=g['x_u'].shape[0],
                     truncation_level=g['dp_logits'].shape[1], alpha_prior_params=np.array([float(g['s_1']), float(g['s_2'])]),
                     device=dev, precision=prec,
                     initial_values=dict(x_mean=g['x_mean'], x_var=sp(g['x_var_raw']), x_u=g['x_u'], phi_logits=g['dp_logits'],
                                         gamma_atoms=sp(g['gamma_atoms_raw']), alpha_atoms=sp(g['alpha_atoms_raw']),
                                         beta_atoms=sp(g['beta_atoms_raw']), gamma_1=sp(g['gamma_1_raw']),
                                         gamma_2=sp(g['gamma_2_raw']), w_1=float(sp(g['w_1_raw'])),
                                         w_2=float(sp(g['w_2_raw']))), **kw)


@pytest.mark.parametrize('fixture', ['grad_ref_40_6_12_3_T4', 'grad_ref_60_10_15_4_T5'])
@pytest.mark.parametrize('prec', ['f64', 'mixed', 'f32'])
def test_model_gradients_match_the_reference(dev, fixture, prec):
    """model.gradients() against tf.gradients of the reference's own objective (tests/golden/grad_ref_*.npz).  precision='f32' scores
    in exact fp32 products; its gradients are those of the mixed evaluation (same tolerance), and the objective it reports stays its
    own."""
    g = golden(fixture)
    model = build_model(g, dev, prec)
    before = float(model.objective) if prec == 'f32' else None
    got = model.gradients()
    if prec == 'f32':
        assert float(model.objective) == before
    tol = 1e-7 if prec == 'f64' else 5e-4
    for ref_name, raw_name in REF2RAW.items():
        want = g['grad_' + ref_name]
        have = got[raw_name].cpu().numpy().reshape(want.shape)
        np.testing.assert_allclose(have, want, rtol=tol, atol=tol * max(1.0, np.abs(want).max()), err_msg=ref_name)
    w = got['dp_w'].cpu().numpy()
    np.testing.assert_allclose(w, [float(g['grad_w_1_raw']), float(g['grad_w_2_raw'])], rtol=tol, atol=tol)


def test_adam_on_hip_gradients_decreases_the_objective(dev):
    g = golden('grad_ref_40_6_12_3_T4')
    model = build_model(g, dev, 'mixed')
    before = float(model.objective)
    model.optimise(30, learning_rate=0.02)
    after = float(model.objective)
    assert np.isfinite(after) and after < before - 1.0, (before, after)


def test_sharded_gradient_path_with_one_rank(dev):
    """pack -> all_reduce -> unpack of the gradients with a 1-rank RCCL group equals the single-GPU gradients."""
    import os
    import torch.distributed as dist
    g = golden('grad_ref_40_6_12_3_T4')
    created = False
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        created = True
    try:
        a = build_model(g, dev, 'f64').gradients()
        b = build_model(g, dev, 'f64', process_group=dist.group.WORLD).gradients()
        for k in a:
            np.testing.assert_array_equal(a[k].cpu().numpy(), b[k].cpu().numpy())
    finally:
        if created:
            dist.destroy_process_group()


def test_training_trajectory_matches_the_oracle(dev):
    """Ten Adam steps (lr 0.01) with the HIP gradients (fp64) against ten torch.optim.Adam steps on the autograd oracle
    from the same raw variables: the objectives along the way agree to 1e-7."""
    g = golden('grad_ref_40_6_12_3_T4')
    model = build_model(g, dev, 'f64')
    hip_traj = []
    model.optimise(10, learning_rate=0.01, callback=lambda it: hip_traj.append(float(model.objective)))
    raw = {k: torch.tensor(np.asarray(g[k], dtype=np.float64), requires_grad=True) for k in ot.NAMES}
    y = torch.as_tensor(g['y'])
    opt = torch.optim.Adam(list(raw.values()), lr=0.01)
    ref_traj = []
    for _ in range(10):
        opt.zero_grad()
        obj, _ = ot.objective(y, raw, s_1=float(g['s_1']), s_2=float(g['s_2']))
        obj.backward()
        opt.step()
        with torch.no_grad():
            ref_traj.append(float(ot.objective(y, raw, s_1=float(g['s_1']), s_2=float(g['s_2']))[0]))
    np.testing.assert_allclose(hip_traj, ref_traj, rtol=1e-7)
    assert hip_traj[-1] < hip_traj[0]


@pytest.mark.parametrize('world', [2, 3])
def test_sharding_arithmetic_on_one_gpu(dev, world):
    """Rank r of `world` without a communicator (test hook _shard_of): the partial (f_hat, DP objective) pairs and the
    partial gradients of all ranks add up to the single-GPU values — what the all-reduces of the sharded path compute."""
    g = golden('grad_ref_60_10_15_4_T5')
    full = build_model(g, dev, 'f64')
    terms = full.objective_terms.cpu().numpy()                      # objective, f_hat, KL, DP objective, hyper-prior
    want = full.gradients()
    pack = sum(build_model(g, dev, 'f64', _shard_of=(r, world)).partial_pack().cpu().numpy() for r in range(world))
    np.testing.assert_allclose(pack, [terms[1], terms[3]], rtol=1e-12)
    parts = [build_model(g, dev, 'f64', _shard_of=(r, world)).gradients() for r in range(world)]
    for k in want:
        got = sum(p_[k].cpu().numpy() for p_ in parts)
        ref = want[k].cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(ref).max()), err_msg=k)


@pytest.mark.parametrize('shape', [(33, 5, 17, 5, 1, 1), (70, 9, 33, 9, 3, 3), (80, 8, 64, 6, 4, 2), (45, 14, 20, 13, 3, 1),
                                   (160, 6, 140, 5, 3, 1), (300, 9, 200, 8, 2, 3), (300, 6, 256, 5, 2, 1)])
@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_model_gradients_odd_shapes(dev, shape, prec):
    """Shapes off the tile sizes (M not a multiple of 16, Q not a multiple of 4, T = 1, mask_size > 1) against the pinned
    autograd oracle at random raw variables."""
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    n, d, m, q, t, mask = shape
    rng = np.random.default_rng(n + d)
    y = rng.standard_normal((n, d))
    y = (y - y.mean(0)) / y.std(0)
    # M > 128: the inducing inputs are spread over four length scales, so that K_uu stays well conditioned and the mixed
    # evaluation is NOT flagged by the conditioning guard — its gradients are then held to the stated tolerance like every
    # other shape (round 2 drew them at unit scale, the guard fired and the test returned without comparing anything).
    raw = dict(x_mean=rng.standard_normal((n, q)), x_var_raw=0.3 * rng.standard_normal((n, q)),
               x_u=(4.0 if m > 128 else 1.0) * rng.standard_normal((m, q)), dp_logits=rng.standard_normal((d // mask, t)),
               gamma_1_raw=rng.standard_normal(max(t - 1, 0)), gamma_2_raw=rng.standard_normal(max(t - 1, 0)),
               w_1_raw=np.array(0.4), w_2_raw=np.array(0.7), gamma_atoms_raw=0.5 * rng.standard_normal((t, q)),
               alpha_atoms_raw=0.5 * rng.standard_normal((t, 1)), beta_atoms_raw=0.5 * rng.standard_normal((t, 1)) + 1.0)
    obj, ref = ot.objective_and_gradients(y, raw, s_1=1.0, s_2=1.0, mask_size=mask)
    sp = softplus
    model = dp_gp_lvm(y, num_latent_dims=q, num_inducing_points=m, truncation_level=t, alpha_prior_params=np.array([1.0, 1.0]),
                      mask_size=mask, device=dev, precision=prec,
                      initial_values=dict(x_mean=raw['x_mean'], x_var=sp(raw['x_var_raw']), x_u=raw['x_u'],
                                          phi_logits=raw['dp_logits'], gamma_atoms=sp(raw['gamma_atoms_raw']),
                                          alpha_atoms=sp(raw['alpha_atoms_raw']), beta_atoms=sp(raw['beta_atoms_raw']),
                                          gamma_1=sp(raw['gamma_1_raw']), gamma_2=sp(raw['gamma_2_raw']),
                                          w_1=float(sp(raw['w_1_raw'])), w_2=float(sp(raw['w_2_raw']))))
    tol = 1e-7 if prec == 'f64' else 1e-3
    np.testing.assert_allclose(float(model.objective), obj, rtol=1e-7 if prec == 'f64' else 2e-5)
    got = model.gradients()
    info = model.per_dimension_terms[1].cpu().numpy()
    assert not info.any(), 'no evaluation of this test may be flagged (conditioning guard) or fail: info = %s' % np.unique(info)
    for ref_name, raw_name in REF2RAW.items():
        want = ref[ref_name]
        if want.size == 0:
            continue
        have = got[raw_name].cpu().numpy().reshape(-1)[:want.size].reshape(want.shape)
        np.testing.assert_allclose(have, want, rtol=tol, atol=tol * max(1.0, np.abs(want).max()), err_msg=ref_name)
    np.testing.assert_allclose(got['dp_w'].cpu().numpy(), [float(ref['w_1_raw']), float(ref['w_2_raw'])], rtol=tol, atol=tol)


@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_stage_a_beyond_128_inducing_points_one_call_equals_the_composed_one(dev, prec, monkeypatch):
    """M a multiple of 128: dpgp_elbo_grad_chain_big (one library call on [D][M][M] matrices) against the same adjoints composed on the
    host from the batched operators (DPGP_STAGE_A_COMPOSED=1); both from the workspace of one forward evaluation."""
    rng = np.random.default_rng(5)
    n, d, m, q = 300, 5, 256, 4
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    y = rng.standard_normal((n, d))
    z, mu, s = 4.0 * rng.standard_normal((m, q)), rng.standard_normal((n, q)), 0.3 + 0.3 * rng.random((n, q))
    gamma, alpha, beta = 0.5 + rng.random((d, q)), 0.5 + rng.random(d), 0.5 + rng.random(d)
    w = ops.ElboWorkspace(d, n, m, q, prec, dev)
    args = [t(a) for a in (y, z, mu, s, gamma, alpha, beta)]
    ops.elbo_fhat(*args, prec=prec, workspace=w)
    one = ops.elbo_grad_chain(args[5], args[6], w, z=args[1], gamma=args[4])
    monkeypatch.setenv('DPGP_STAGE_A_COMPOSED', '1')
    ref = ops.elbo_grad_chain(args[5], args[6], w, z=args[1], gamma=args[4])
    assert int(one[4].abs().max()) == 0 and int(ref[4].abs().max()) == 0
    for name, a, b in zip(('g_psi2', 'w_kuu', 'g_v', 'd_alpha_beta'), one, ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=1e-9 * float(b.abs().max()), err_msg=name)


def _stage_b_problem(dev, shape):
    n, d, m, q = shape
    rng = np.random.default_rng(n + m)
    y = rng.standard_normal((n, d))
    z = rng.standard_normal((m, q)) * 1.5
    mu = rng.standard_normal((n, q))
    s = np.exp(0.3 * rng.standard_normal((n, q)))
    gamma = np.exp(0.4 * rng.standard_normal((d, q))) * 0.5
    alpha = np.exp(0.2 * rng.standard_normal(d))
    beta = np.exp(0.2 * rng.standard_normal(d)) * 2.0
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    return [t(a) for a in (y, z, mu, s, gamma, alpha, beta)]


@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_fused_stage_a_refuses_more_than_128_inducing_points(dev, prec):
    """M = 130: the HIP stage A keeps B in LDS (M <= 128) — without z and gamma for the host-side composition the call must
    say so instead of computing something else."""
    n, d, m, q = 150, 5, 130, 7
    args = _stage_b_problem(dev, (n, d, m, q))
    w = ops.ElboWorkspace(d, n, m, q, prec, dev)
    ops.elbo_fhat(*args, prec=prec, workspace=w)
    with pytest.raises(ValueError):
        ops.elbo_grad_chain(args[5], args[6], w)


@pytest.mark.parametrize('shape', [(300, 16, 100, 10), (500, 8, 128, 12), (200, 6, 70, 4), (200, 6, 64, 20),
                                   (120, 5, 40, 30), (150, 4, 96, 17), (90, 3, 33, 29), (130, 7, 128, 13)])
def test_matrix_pipe_stage_b_against_fp64(dev, shape):
    """Mixed precision: the Psi2 term of stage B runs on the matrix pipe (psi2_grad_kernel), Psi1 in the reduction-free
    kernels.  Against the fp64 kernel (itself at 1e-8 of the oracle's autograd, tests above) on the same stage-A adjoints."""
    n, d, m, q = shape
    args = _stage_b_problem(dev, shape)
    out, adj, flagged = {}, {}, {}
    for prec in ('f64', 'mixed'):
        w = ops.ElboWorkspace(d, n, m, q, prec, dev)
        ops.elbo_fhat(*args, prec=prec, workspace=w)
        flagged[prec] = bool((w.info == -2).any())              # conditioning guard of the forward evaluation
        assert int(w.info.clamp(min=0).max()) == 0
        gp, wk, gv, dab, info = ops.elbo_grad_chain(args[5], args[6], w)
        assert int(info.abs().max()) == 0
        adj[prec] = (gp, wk, gv)
        out[prec] = [a.cpu().numpy() for a in ops.elbo_grad_psi(args[0], args[1], args[2], args[3], args[4], args[5], gp, wk, gv, prec=prec)]
    # (1) the streaming stage itself: both kernels on the SAME (fp64) adjoints
    same = [a.cpu().numpy() for a in ops.elbo_grad_psi(args[0], args[1], args[2], args[3], args[4], args[5], *adj['f64'], prec='mixed')]
    for name, want, got in zip(('d mu', 'd S', 'd z', 'd gamma'), out['f64'], same):
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-4 * np.abs(want).max(), err_msg=name + ' (same adjoints)')
    # (2) end to end in mixed precision (fp32 Psi2 -> adjoints -> stage B), wherever the forward evaluation is not flagged as
    #     ill-conditioned: random Z with M = 70 ... 128 points in 4 ... 30 latent dims amplifies the rounding of Psi2
    assert not flagged['f64']                                    # (the bound is computed, never flagged, in fp64)
    #     (the mixed-precision gradient tolerance stated for the model, 5e-4 of the largest entry: test_model_gradients_*)
    if not flagged['mixed']:
        for name, want, got in zip(('d mu', 'd S', 'd z', 'd gamma'), out['f64'], out['mixed']):
            np.testing.assert_allclose(got, want, rtol=0, atol=5e-4 * np.abs(want).max(), err_msg=name)


@pytest.mark.parametrize('shape', [(150, 4, 130, 5), (260, 3, 200, 7), (300, 2, 257, 14), (1, 3, 200, 4)])
def test_stage_b_beyond_128_inducing_points(dev, shape):
    """Stage B alone for M > 128 (K_uu term by kuu_grad_kernel, Psi1 kernels over row blocks, the Psi2 term in the pair-tile
    or the patch form), fed with the stage-A adjoints of the oracle; against autograd of the oracle.  The last shape is ONE
    observation against 200 inducing points — the prediction paths' M > N (the K_uu term's partial sums used to
    overrun the workspace region sized by N)."""
    n, d, m, q = shape
    rng = np.random.default_rng(m)
    y = rng.standard_normal((n, d))
    z = rng.standard_normal((m, q)) * 2.0
    mu = rng.standard_normal((n, q)) * 1.5
    s = np.exp(0.3 * rng.standard_normal((n, q)))
    gamma = np.exp(0.3 * rng.standard_normal((d, q))) * 1.5
    alpha = np.exp(0.2 * rng.standard_normal(d))
    beta = np.exp(0.2 * rng.standard_normal(d)) * 2.0
    ref = ot.fhat_input_gradients(y, z, mu, s, gamma, alpha, beta)
    adj = ot.chain_adjoints(y, z, mu, s, gamma, alpha, beta)
    mp = 16 * ((m + 15) // 16)
    pad = lambda a: np.pad(a, ((0, 0), (0, mp - m), (0, mp - m)))
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    gp, wk = t(pad(adj['g_psi2'])), t(pad(adj['g_kuu'] * (adj['k_uu'] - 1e-8 * np.eye(m))))
    gv = t(np.pad(adj['g_v'], ((0, 0), (0, mp - m))))
    dmu, ds, dz, dg = ops.elbo_grad_psi(t(y), t(z), t(mu), t(s), t(gamma), t(alpha), gp, wk, gv, prec='mixed')
    for name, got, want in (('d mu', dmu, ref['d_mu']), ('d S', ds, ref['d_s']), ('d z', dz, ref['d_z']), ('d gamma', dg, ref['d_gamma'])):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=3e-4 * np.abs(want).max(), err_msg=name)
    # fp64: the M <= 128 kernel over pairs of 64-point blocks (ops._elbo_grad_psi_f64_blocks)
    dmu, ds, dz, dg = ops.elbo_grad_psi(t(y), t(z), t(mu), t(s), t(gamma), t(alpha), gp, wk, gv, prec='f64')
    for name, got, want in (('d mu', dmu, ref['d_mu']), ('d S', ds, ref['d_s']), ('d z', dz, ref['d_z']), ('d gamma', dg, ref['d_gamma'])):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=1e-10 * np.abs(want).max(), err_msg=name + ' (f64)')


def _ill_conditioned(g, scale):
    """The fixture's problem with the ARD weights scaled down: long length scales make the inducing points redundant and
    K_uu nearly singular (what Adam does to the BASELINE configs after ~200 iterations)."""
    g2 = {k: g[k] for k in g.files} if hasattr(g, 'files') else dict(g)
    g2['gamma_atoms_raw'] = np.log(np.expm1(softplus(np.asarray(g['gamma_atoms_raw'])) * scale))
    return g2


def test_optimise_checks_flags_every_iteration(dev):
    """optimise() reads ONE flag per iteration (Cholesky / conditioning guard / non-finite gradient; it travels with the packed
    gradient all-reduce) and raises instead of stepping on meaningless gradients; precision='f64' with the streaming stage
    of the backward pass on the matrix pipe is the configuration that keeps going where the fp32 Psi2 is flagged."""
    g = golden('grad_ref_60_10_15_4_T5')
    model = build_model(g, dev, 'mixed')
    before = float(model.objective)
    stats = model.optimise(12, learning_rate=0.01)
    assert stats == {'iterations': 12, 'precision': 'mixed', 'backward_precision': 'mixed'}
    assert float(model.objective) < before
    assert int((model.per_dimension_terms[1] != 0).sum()) == 0
    bad = _ill_conditioned(g, 1e-5)
    mixed = build_model(bad, dev, 'mixed')
    with pytest.raises(FloatingPointError):
        mixed.optimise(3)
    assert set(mixed.per_dimension_terms[1].unique().tolist()) - {0} != set()       # flagged, not silently wrong
    robust = build_model(bad, dev, 'f64', backward_precision='mixed')
    before = float(robust.objective)
    assert robust.optimise(5, learning_rate=0.01)['backward_precision'] == 'mixed'
    assert int((robust.per_dimension_terms[1] != 0).sum()) == 0
    assert float(robust.objective) < before


@pytest.mark.parametrize('fixture', ['grad_ref_40_6_12_3_T4', 'grad_ref_60_10_15_4_T5'])
def test_fp64_forward_with_matrix_pipe_stage_b(dev, fixture):
    """precision='f64', backward_precision='mixed' (what optimise() falls back to): fp64 forward and dense adjoints, the
    streaming stage on the matrix pipe; against the reference's tf.gradients."""
    g = golden(fixture)
    model = build_model(g, dev, 'f64', backward_precision='mixed')
    np.testing.assert_allclose(float(model.objective), float(g['objective']), rtol=1e-10)
    got = model.gradients()
    for ref_name, raw_name in REF2RAW.items():
        want = g['grad_' + ref_name]
        have = got[raw_name].cpu().numpy().reshape(want.shape)
        np.testing.assert_allclose(have, want, rtol=5e-4, atol=5e-4 * max(1.0, np.abs(want).max()), err_msg=ref_name)
    # the form of the Psi2 term follows the conditioning-guard bound of the forward evaluation (tests/test_gpu_illcond.py: the patch form
    # at 80 x the threshold); whichever it is, the gradients above hold the mixed tolerance
    bound = float(model.conditioning_guard.max())
    assert model.last_stage_b_form == ('mixed' if bound <= 0.1 * 2.0e-3 * g['y'].shape[0] else 'mixed_patch')


def test_training_configuration_switches_between_the_two_forms_of_stage_b(dev, monkeypatch):
    """Well-conditioned K_uu (spread-out inducing inputs, short length scales): the pair-tile form; and the same model with
    DPGP_ADAPTIVE_STAGE_B=0: the patch form — both within the mixed tolerance of the all-fp64 gradients."""
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    rng = np.random.default_rng(11)
    n, d, m, q, t = 300, 12, 40, 4, 3
    y = rng.standard_normal((n, d))
    y = (y - y.mean(0)) / y.std(0)
    iv = dict(x_mean=rng.standard_normal((n, q)), x_var=0.3 + 0.2 * rng.random((n, q)), x_u=3.0 * rng.standard_normal((m, q)),
              phi_logits=rng.standard_normal((d, t)), gamma_atoms=1.0 + rng.random((t, q)), alpha_atoms=1.0 + rng.random((t, 1)),
              beta_atoms=1.0 + rng.random((t, 1)), gamma_1=1.0 + rng.random(t - 1), gamma_2=1.0 + rng.random(t - 1), w_1=1.3, w_2=0.8)
    kw = dict(num_latent_dims=q, num_inducing_points=m, truncation_level=t, device=dev, initial_values=iv)
    ref = dp_gp_lvm(y, precision='f64', backward_precision='f64', **kw).gradients()
    a = dp_gp_lvm(y, precision='f64', backward_precision='mixed', **kw)
    ga = a.gradients()
    assert a.last_stage_b_form == 'mixed', float(a.conditioning_guard.max())
    monkeypatch.setenv('DPGP_ADAPTIVE_STAGE_B', '0')
    b = dp_gp_lvm(y, precision='f64', backward_precision='mixed', **kw)
    gb = b.gradients()
    assert b.last_stage_b_form is None
    for k, want in ref.items():
        w = want.cpu().numpy()
        for got in (ga[k], gb[k]):
            np.testing.assert_allclose(got.cpu().numpy(), w, rtol=0, atol=5e-4 * max(np.abs(w).max(), 1e-300), err_msg=k)


@pytest.mark.parametrize('shape', [(40, 6, 12, 3), (33, 5, 17, 5), (200, 6, 70, 4), (300, 16, 100, 10), (260, 3, 200, 7)])
def test_pair_tile_and_patch_forms_of_stage_b_agree(dev, shape):
    """The Psi2 term of stage B exists in two forms (include/dpgp.h: DPGP_PREC_MIXED = pair tiles, psi2_pairs_grad.hip;
    DPGP_PREC_MIXED_PATCH = per-observation patches, psi2_grad_kernel): same adjoints in, the four gradients must agree to the
    mixed tolerance (measured 1e-5 ... 4e-5 of the largest entry on well-conditioned adjoints; the test feeds a smooth positive
    definite adjoint — with a random indefinite one the two differ by 5e-3 through cancellation)."""
    n, d, m, q = shape
    rng = np.random.default_rng(m)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    y = rng.standard_normal((n, d))
    z = rng.standard_normal((m, q)) * 2.0
    mu = rng.standard_normal((n, q)) * 1.5
    s = np.exp(0.3 * rng.standard_normal((n, q)))
    gamma = np.exp(0.3 * rng.standard_normal((d, q))) * 1.5
    alpha = np.exp(0.2 * rng.standard_normal(d))
    mp = 16 * ((m + 15) // 16)
    dist = ((z[:, None, :] - z[None, :, :]) ** 2).sum(-1)
    g = np.zeros((d, mp, mp))
    g[:, :m, :m] = np.exp(-0.1 * dist)[None] * (1.0 + 0.1 * rng.standard_normal((d, 1, 1)))
    wk, gv = np.zeros((d, mp, mp)), np.zeros((d, mp))
    args = (t(y), t(z), t(mu), t(s), t(gamma), t(alpha), t(g), t(wk), t(gv))
    pair = ops.elbo_grad_psi(*args, prec='mixed')
    patch = ops.elbo_grad_psi(*args, prec='mixed_patch')
    for name, a, b in zip(('d mu', 'd S', 'd z', 'd gamma'), pair, patch):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=2e-4 * float(b.abs().max()), err_msg=name)


def test_pair_tile_stage_b_range_guard_poisons_the_gradients(dev):
    """Observations ~100 length scales from the inducing inputs: the f16-split exponent of the pair-tile kernels is out of range
    (|c''| > 8192, psi2_pairs.hip); the gradients must come back NaN — never finite garbage."""
    n, d, m, q = 64, 4, 12, 3
    rng = np.random.default_rng(5)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    z = rng.standard_normal((m, q))
    mu = rng.standard_normal((n, q))
    mu[7] += 300.0                                              # one observation far away in every latent dim
    s = np.full((n, q), 0.01)
    gamma = np.full((d, q), 4.0)
    alpha = np.ones(d)
    mp = 16
    g = np.zeros((d, mp, mp))
    g[:, :m, :m] = 1.0
    out = ops.elbo_grad_psi(t(rng.standard_normal((n, d))), t(z), t(mu), t(s), t(gamma), t(alpha), t(g), t(np.zeros((d, mp, mp))),
                            t(np.zeros((d, mp))), prec='mixed')
    assert all(bool(torch.isnan(o).any()) for o in out)


@pytest.mark.parametrize('shape', [(300, 16, 100, 10), (500, 8, 128, 12), (200, 6, 70, 4), (200, 6, 64, 20), (150, 4, 96, 17),
                                   (257, 9, 33, 15), (64, 3, 16, 3)])
def test_training_step_equals_the_three_calls(dev, shape):
    """dpgp_elbo_step (Psi2 from the first pass of stage B, stage A, the rest of stage B in one call) against dpgp_elbo_fhat_ex +
    dpgp_elbo_grad_chain + dpgp_elbo_grad_psi on the same inputs, mixed precision: the f_hat terms to the mixed tolerance of the
    forward tests (2e-5 of the largest term; the two Psi2 kernels round differently), the gradients of BOTH paths to the stated
    mixed-precision gradient tolerance (5e-4 of the largest entry) of the all-fp64 gradients."""
    n, d, m, q = shape
    args = _stage_b_problem(dev, shape)
    args[1] = args[1] * 0.6                                      # (inducing inputs inside the cloud of latent means: well-conditioned K_uu)
    w3 = ops.ElboWorkspace(d, n, m, q, 'mixed', dev)
    terms3, sums3, info3 = [a.clone() for a in ops.elbo_fhat(*args, prec='mixed', workspace=w3)]
    gp, wk, gv, dab3, infog3 = ops.elbo_grad_chain(args[5], args[6], w3)
    g3 = [a.cpu().numpy() for a in ops.elbo_grad_psi(args[0], args[1], args[2], args[3], args[4], args[5], gp, wk, gv, prec='mixed')]
    w1 = ops.ElboWorkspace(d, n, m, q, 'mixed', dev)
    b1 = ops.ElboStepBuffers(d, n, m, q, dev)
    assert ops.elbo_step_supported(m, q)
    for rep in range(2):                                         # (twice: the buffers are reused by every step)
        (terms1, sums1, info1), (dmu, ds, dz, dg, dab1, infog1) = ops.elbo_step(*args, workspace=w1, buffers=b1)
    # output dims that the conditioning guard flags (info = -2: the fp32 rounding of Psi2 is amplified beyond the mixed tolerance by a
    # nearly singular K_uu — random inducing inputs in few latent dims) are exempt in both paths, as in the forward tests
    ok = ((info1 == 0) & (info3 == 0)).cpu().numpy()
    assert int(info1.clamp(min=0).max()) == 0 and int(info3.clamp(min=0).max()) == 0
    assert torch.equal(info1, info3), 'the two paths flag different output dims'
    t3 = terms3.cpu().numpy()
    assert np.isfinite(terms1.cpu().numpy()).all()
    if ok.any():
        np.testing.assert_allclose(terms1.cpu().numpy()[ok], t3[ok], rtol=0, atol=2e-5 * np.abs(t3[ok]).max())
    if ok.all():
        np.testing.assert_allclose(sums1.cpu().numpy(), sums3.cpu().numpy(), rtol=2e-6)
        assert int(infog1.abs().max()) == 0 and int(infog3.abs().max()) == 0
        w64 = ops.ElboWorkspace(d, n, m, q, 'f64', dev)
        ops.elbo_fhat(*args, prec='f64', workspace=w64)
        gp, wk, gv, dab64, _ = ops.elbo_grad_chain(args[5], args[6], w64)
        g64 = [a.cpu().numpy() for a in ops.elbo_grad_psi(args[0], args[1], args[2], args[3], args[4], args[5], gp, wk, gv, prec='f64')]
        for path, dab, grads in (('step', dab1, [a.cpu().numpy() for a in (dmu, ds, dz, dg)]), ('three calls', dab3, g3)):
            np.testing.assert_allclose(dab.cpu().numpy(), dab64.cpu().numpy(), rtol=0, atol=5e-4 * float(dab64.abs().max()), err_msg=path)
            for name, got, want in zip(('d mu', 'd S', 'd z', 'd gamma'), grads, g64):
                np.testing.assert_allclose(got, want, rtol=0, atol=5e-4 * np.abs(want).max(), err_msg='%s, %s' % (name, path))


def test_model_gradients_fused_step_equals_separate_calls(dev, monkeypatch):
    """dp_gp_lvm.gradients() through dpgp_elbo_step (default) and through the three separate calls (DPGP_FUSED_STEP=0): same
    objective (2e-6) and gradients (2e-4 of the largest entry) of all raw variables."""
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    from dp_gp_lvm_amd.utils.synthetic import make_problem
    p = make_problem(1)
    def build():
        return dp_gp_lvm(p['y'], num_latent_dims=p['mu'].shape[1], num_inducing_points=p['z'].shape[0],
                         truncation_level=p['phi'].shape[1], alpha_prior_params=np.array([p['s1'], p['s2']]), device=dev,
                         precision='mixed',
                         initial_values=dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']),
                                             gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'],
                                             gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2']))
    out = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('DPGP_FUSED_STEP', mode)
        mdl = build()
        g = mdl.gradients()
        out[mode] = ({k: v.cpu().numpy().copy() for k, v in g.items()}, mdl.objective_terms.cpu().numpy())
    np.testing.assert_allclose(out['1'][1], out['0'][1], rtol=2e-6)
    for k in out['0'][0]:
        want = out['0'][0][k]
        np.testing.assert_allclose(out['1'][0][k], want, rtol=0, atol=2e-4 * max(np.abs(want).max(), 1e-12), err_msg=k)


@pytest.mark.parametrize('shape', [(300, 16, 100, 10), (500, 8, 128, 12), (200, 6, 64, 20), (150, 4, 96, 17)])
def test_fast_stage_b_within_its_stated_tolerance(dev, shape):
    """DPGP_PREC_MIXED_FAST (opt-in: the second products of the pair-tile stage B take 11-bit exponentials, include/dpgp.h): against
    the fp64 stage B on the same fp64 adjoints — 2e-3 of the largest entry at these small N (a few hundred observations per sum;
    the default mixed mode: 2e-4), where its rounding errors average least."""
    n, d, m, q = shape
    args = _stage_b_problem(dev, shape)
    w = ops.ElboWorkspace(d, n, m, q, 'f64', dev)
    ops.elbo_fhat(*args, prec='f64', workspace=w)
    gp, wk, gv, _, info = ops.elbo_grad_chain(args[5], args[6], w)
    assert int(info.abs().max()) == 0
    want = [a.cpu().numpy() for a in ops.elbo_grad_psi(args[0], args[1], args[2], args[3], args[4], args[5], gp, wk, gv, prec='f64')]
    got = [a.cpu().numpy() for a in ops.elbo_grad_psi(args[0], args[1], args[2], args[3], args[4], args[5], gp, wk, gv, prec='mixed_fast')]
    for name, a, b in zip(('d mu', 'd S', 'd z', 'd gamma'), got, want):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-3 * np.abs(b).max(), err_msg=name)


def test_fast_stage_b_at_a_baseline_shape(dev):
    """The same mode where it is meant to be used — BASELINE config 2 (N = 2000, D = 64, M = 128, Q = 10): the gradients of all raw
    variables through dpgp_elbo_step with backward_precision='mixed_fast' against the default mixed step: 5e-5 of the largest
    entry of each (measured ~3e-6; the stated mixed-precision gradient tolerance is 5e-4), same objective."""
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    from dp_gp_lvm_amd.utils.synthetic import make_problem
    p = make_problem(2)
    out = {}
    for bp in ('mixed', 'mixed_fast'):
        mdl = dp_gp_lvm(p['y'], num_latent_dims=p['mu'].shape[1], num_inducing_points=p['z'].shape[0],
                        truncation_level=p['phi'].shape[1], alpha_prior_params=np.array([p['s1'], p['s2']]), device=dev,
                        precision='mixed', backward_precision=bp,
                        initial_values=dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi']),
                                            gamma_atoms=p['gamma_atoms'], alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'],
                                            gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2']))
        g = mdl.gradients()
        out[bp] = ({k: v.cpu().numpy().copy() for k, v in g.items()}, float(mdl.objective))
    assert out['mixed'][1] == out['mixed_fast'][1]                 # (the pass that yields Psi2 keeps both halves)
    for k, want in out['mixed'][0].items():
        np.testing.assert_allclose(out['mixed_fast'][0][k], want, rtol=0, atol=5e-5 * max(np.abs(want).max(), 1e-12), err_msg=k)
