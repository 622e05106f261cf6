"""
GPU parity tests of the model object dp_gp_lvm(...) (mirror of /root/reference/src/models/dp_gp_lvm.py:22-231) built
from the golden fixtures' post-initialisation parameter values: objective and its breakdown against the values the
reference's own source produced (tests/golden/dpgplvm_*.npz: TestDPGPLVM / TestT1 / TestD2T1 of
test/unittests/dpgplvm_unitttests.py, including the reference's naive known-answer objective) and accessors.
Tolerance: objective rel 1e-9 (prec f64), 2e-5 (mixed), 5e-4 (f32).
"""
import numpy as np
import pytest
import torch

from conftest import golden
from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
from dp_gp_lvm_amd.models.dirichlet_process import dirichlet_process
from dp_gp_lvm_amd.kernels.rbf_kernel import k_ard_rbf
from dp_gp_lvm_amd.kernels.interfaces.kernel import Kernel, KernelHyperparameters
from dp_gp_lvm_amd.utils.synthetic import make_problem

pytestmark = pytest.mark.gpu
RTOL = {'f64': 1e-9, 'mixed': 2e-5, 'f32': 5e-4}


def build(g, dev, prec, **kw):
    return dp_gp_lvm(g['y'], num_latent_dims=g['mu'].shape[1], num_inducing_points=g['z'].shape[0],
                     truncation_level=g['phi'].shape[1], alpha_prior_params=np.array([float(g['s1']), float(g['s2'])]),
                     device=dev, precision=prec,
                     initial_values=dict(x_mean=g['mu'], x_var=g['s'], x_u=g['z'], phi_logits=np.log(g['phi']),
                                         gamma_atoms=g['gamma_atoms'], alpha_atoms=g['alpha_atoms'],
                                         beta_atoms=g['beta_atoms'], gamma_1=g['g1'], gamma_2=g['g2'],
                                         w_1=float(g['w1']), w_2=float(g['w2'])), **kw)


@pytest.mark.parametrize('fixture', ['dpgplvm_50_10_25_3_T8', 'dpgplvm_T1_d5', 'dpgplvm_d2'])
@pytest.mark.parametrize('prec', ['f64', 'mixed', 'f32'])
def test_objective_matches_reference_model(dev, fixture, prec):
    g = golden(fixture)
    model = build(g, dev, prec)
    obj, fhat, kl, dpo, hyp = model.objective_terms.cpu().numpy()
    rt = RTOL[prec]
    np.testing.assert_allclose(obj, float(g['objective']), rtol=rt)
    np.testing.assert_allclose(obj, float(g['objective_naive']), rtol=max(rt, 1e-7))   # the reference's known answer
    np.testing.assert_allclose(fhat, g['fhat_per_d'].sum(), rtol=rt)
    np.testing.assert_allclose(kl, float(g['kl']), rtol=1e-12)
    np.testing.assert_allclose(dpo, float(g['dp_objective']), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(hyp, float(g['hyperprior']), rtol=1e-12, atol=1e-12)
    assert float(model.objective) == obj                                   # re-evaluation is deterministic
    # accessors (dp_gp_lvm.py:161-231)
    np.testing.assert_allclose(model.assignments.cpu().numpy(), g['phi'], rtol=1e-12)
    np.testing.assert_allclose(model.ard_weights.cpu().numpy(), g['gamma'], rtol=1e-12)
    np.testing.assert_allclose(model.signal_variance.cpu().numpy(), g['alpha'], rtol=1e-12)
    np.testing.assert_allclose(model.noise_precision.cpu().numpy(), g['beta'], rtol=1e-12)
    mean, covar = model.q_x
    assert covar.shape == (g['mu'].shape[0], g['mu'].shape[1], g['mu'].shape[1])
    np.testing.assert_allclose(torch.diagonal(covar, dim1=-2, dim2=-1).cpu().numpy(), g['s'], rtol=1e-12)
    np.testing.assert_allclose(float(model.dp.objective), float(g['dp_objective']), rtol=1e-10, atol=1e-10)
    gat, aat, bat = model.dp_atoms
    np.testing.assert_allclose(gat.cpu().numpy(), g['gamma_atoms'], rtol=1e-12)


@pytest.mark.parametrize('fixture', ['grad_ref_40_6_12_3_T4', 'grad_ref_60_10_15_4_T5'])
@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_objective_at_the_gradient_fixture_points(dev, fixture, prec):
    """The HIP forward pass at the points where tests/golden/grad_ref_*.npz hold the reference's own gradients (groundwork
    for the backward pass: the forward value must agree there first).  The fixtures store the reference's RAW variables."""
    g = golden(fixture)
    sp = lambda x: np.logaddexp(0.0, x)                                     # utils/types.py:57
    model = dp_gp_lvm(g['y'], num_latent_dims=g['x_mean'].shape[1], num_inducing_points=g['x_u'].shape[0],
                      truncation_level=g['dp_logits'].shape[1], alpha_prior_params=np.array([float(g['s_1']), float(g['s_2'])]),
                      device=dev, precision=prec,
                      initial_values=dict(x_mean=g['x_mean'], x_var=sp(g['x_var_raw']), x_u=g['x_u'], phi_logits=g['dp_logits'],
                                          gamma_atoms=sp(g['gamma_atoms_raw']), alpha_atoms=sp(g['alpha_atoms_raw']),
                                          beta_atoms=sp(g['beta_atoms_raw']), gamma_1=sp(g['gamma_1_raw']),
                                          gamma_2=sp(g['gamma_2_raw']), w_1=float(sp(g['w_1_raw'])),
                                          w_2=float(sp(g['w_2_raw']))))
    np.testing.assert_allclose(float(model.objective), float(g['objective']), rtol=RTOL[prec])


def test_model_kernel_object_matches_reference(dev):
    """model.kernel is a Kernel with batch D whose operators reproduce the reference's K_uu / psi1 / psi2."""
    g = golden('dpgplvm_50_10_25_3_T8')
    model = build(g, dev, 'f64')
    k = model.kernel
    assert isinstance(k, Kernel) and set(k.hyperparameters) == {KernelHyperparameters.ARD_WEIGHTS,
                                                                KernelHyperparameters.SIGNAL_VARIANCE,
                                                                KernelHyperparameters.NOISE_PRECISION}
    mean, covar = model.q_x
    z = model.inducing_input
    np.testing.assert_allclose(k.covariance_matrix(z, None, include_noise=False, include_jitter=True).cpu().numpy(),
                               g['k_uu'], rtol=1e-10)
    np.testing.assert_allclose(k.psi_1(z, mean, covar).cpu().numpy(), g['psi_1'], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(k.psi_2(z, mean, covar).cpu().numpy(), g['psi_2'], rtol=1e-10, atol=1e-300)
    np.testing.assert_allclose(k.psi_0(z, mean, covar).cpu().numpy(), g['alpha'] * g['mu'].shape[0], rtol=1e-12)
    np.testing.assert_allclose(float(k.prior_log_likelihood), float(k.prior_log_likelihood))


def test_constructor_assertions_and_default_init(dev):
    """Same AssertionErrors as the reference (dp_gp_lvm.py:53-59); default construction (PCA + random DP) evaluates."""
    y = make_problem(1)['y']
    with pytest.raises(AssertionError):
        dp_gp_lvm(y, num_latent_dims=0, device=dev)
    with pytest.raises(AssertionError):
        dp_gp_lvm(y, num_latent_dims=4, num_inducing_points=y.shape[0] + 1, device=dev)
    with pytest.raises(AssertionError):
        dp_gp_lvm(y, num_latent_dims=4, num_inducing_points=20, truncation_level=y.shape[1] + 1, device=dev)
    np.random.seed(1)
    model = dp_gp_lvm(y, num_latent_dims=4, num_inducing_points=20, truncation_level=8, device=dev, precision='f64')
    o1 = float(model.objective)
    assert np.isfinite(o1)
    terms, info = model.per_dimension_terms
    assert int(info.abs().max()) == 0
    with pytest.raises(AssertionError):                                    # all D dims observed is not a missing-data problem
        model.predict_missing_data(y)
    lower_bound, x_test_mean, x_test_covar, mean, covar = model.predict_missing_data(y[:9, :8])
    assert mean.shape == (9, y.shape[1] - 8) and covar.shape == (y.shape[1] - 8, 9, 9) and bool(torch.isfinite(lower_bound))


def test_mask_size_groups_adjacent_dims(dev):
    y = make_problem(1)['y']
    np.random.seed(3)
    model = dp_gp_lvm(y, num_latent_dims=4, num_inducing_points=20, truncation_level=4, mask_size=3, device=dev,
                      precision='f64')
    phi = model.assignments.cpu().numpy()
    assert phi.shape == (12, 4)
    np.testing.assert_array_equal(phi[0], phi[2])
    np.testing.assert_array_equal(phi[3], phi[5])
    assert not np.array_equal(phi[0], phi[3])
    obj, fhat, kl, dpo, hyp = model.objective_terms.cpu().numpy()
    np.testing.assert_allclose(dpo, float(model.dp.objective), rtol=1e-10)     # HIP prepare kernel vs torch DP objective


def test_dirichlet_process_standalone(dev):
    np.random.seed(1)
    dp = dirichlet_process(num_samples=10, truncation_level=20, alpha_prior_params=np.array([1.1, 0.9]), device=dev)
    from oracle import dpgp_oracle as orc
    g1, g2 = dp.q_v
    w1, w2 = dp.q_alpha
    ref = orc.dp_objective(dp.q_z.cpu().numpy(), g1.cpu().numpy(), g2.cpu().numpy(), float(w1), float(w2), 1.1, 0.9)
    np.testing.assert_allclose(float(dp.objective), ref, rtol=1e-10)


def test_objective_replayed_from_a_hip_graph_follows_the_parameters(dev):
    """evaluate_graph(): the captured evaluation gives the eager numbers bit for bit, and sees in-place parameter updates."""
    import torch
    g = golden('dpgplvm_50_10_25_3_T8')
    model = build(g, dev, 'mixed')
    eager = model.objective_terms.cpu().numpy()
    replay = model.evaluate_graph().clone().cpu().numpy()
    np.testing.assert_array_equal(replay, eager)
    with torch.no_grad():
        model.raw['x_mean'].mul_(1.01)
        model.raw['gamma_atoms'].add_(0.05)
    eager2 = model.objective_terms.cpu().numpy()
    out = torch.zeros(5, dtype=torch.float64, device=dev)
    model.evaluate_graph(out=out)
    np.testing.assert_array_equal(out.cpu().numpy(), eager2)
    assert eager2[0] != eager[0]


def test_per_gpu_shares_of_config_3_agree_with_the_whole(dev):
    """The split of the observations / pair tiles over workgroups depends on the number of output dims a GPU holds (psi2_nsplit,
    pairs_geom: one round of workgroups for D = 64 / 128 / 256 / 512): the per-dim ELBO terms of the first D output dims of
    BASELINE config 3 must not depend on it (mixed precision: fp32 sums in a different association, 2e-6 of the largest term)."""
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    from dp_gp_lvm_amd.utils.synthetic import make_problem, CONFIGS
    n, dfull, m, q = CONFIGS[3]
    p = make_problem(3)
    t = p['phi'].shape[1]

    def terms(d):
        init = dict(x_mean=p['mu'], x_var=p['s'], x_u=p['z'], phi_logits=np.log(p['phi'][:d]), gamma_atoms=p['gamma_atoms'],
                    alpha_atoms=p['alpha_atoms'], beta_atoms=p['beta_atoms'], gamma_1=p['g1'], gamma_2=p['g2'], w_1=p['w1'], w_2=p['w2'])
        model = dp_gp_lvm(p['y'][:, :d], num_latent_dims=q, num_inducing_points=m, truncation_level=t,
                          alpha_prior_params=np.array([p['s1'], p['s2']]), device=dev, initial_values=init, precision='mixed')
        float(model.objective)                                     # (per_dimension_terms: of the LAST evaluation)
        tm, info = model.per_dimension_terms
        assert int(info.abs().max()) == 0
        return tm.cpu().numpy()

    whole = terms(dfull)
    for d in (64, 128, 256):
        part = terms(d)
        np.testing.assert_allclose(part, whole[:d], rtol=0, atol=2e-6 * np.abs(whole).max(), err_msg='D = %d' % d)
