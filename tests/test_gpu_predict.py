"""GPU tests of the prediction paths (reference src/models/dp_gp_lvm.py:233-500) against fixtures produced
by the reference's own methods (oracle/gen_golden_predict.py -> tests/golden/predict_ref_*.npz).  The reference's lower
bounds carry a broadcasting defect (see the generator's header); the product computes the defect-free bound, and the test
reproduces the reference's number by adding the defect, computed from the product's own per-dimension terms."""
import numpy as np
import pytest
import torch

from conftest import golden
from test_gpu_model_t import build

pytestmark = pytest.mark.gpu
FIXTURES = ['predict_ref_40_6_12_3_T4', 'predict_ref_60_10_15_4_T5']


def defect(model, n_test):
    terms = model.prediction_terms.cpu().numpy()
    dd = terms.shape[0]
    al = model.signal_variance.cpu().numpy().reshape(-1)[:dd]
    be = model.noise_precision.cpu().numpy().reshape(-1)[:dd]
    tr = 2.0 * terms[:, 2] / be + al * n_test
    psi0 = al * n_test
    return 0.5 * float(np.sum(be[:, None] * (tr[None, :] - psi0[:, None]))) - 0.5 * float(np.sum(be * (tr - psi0)))


@pytest.mark.parametrize('fixture', FIXTURES)
@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_predict_new_latent_variables(dev, fixture, prec):
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    g = golden(fixture)
    model = build(dp_gp_lvm, g, dev, prec)
    n_test = g['y_test'].shape[0]
    lb, xm, xc, ll = model.predict_new_latent_variables(g['y_test'], x_test_mean=g['new_x_test_mean'])
    tol = 1e-9 if prec == 'f64' else 2e-5
    np.testing.assert_allclose(float(ll), float(g['new_test_log_likelihood_clean']), rtol=tol)
    np.testing.assert_allclose(float(lb), float(g['new_lower_bound_clean']), rtol=tol)
    np.testing.assert_allclose(float(ll) + defect(model, n_test), float(g['new_test_log_likelihood']), rtol=tol)   # the reference's number
    np.testing.assert_allclose(float(lb) + defect(model, n_test), float(g['new_lower_bound']), rtol=tol)
    np.testing.assert_allclose(xm.cpu().numpy(), g['new_x_test_mean'], rtol=0, atol=1e-15)
    np.testing.assert_allclose(xc.cpu().numpy(), g['new_x_test_covar'], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize('fixture', FIXTURES)
@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_predict_missing_data(dev, fixture, prec):
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    g = golden(fixture)
    model = build(dp_gp_lvm, g, dev, prec)
    do, n_test = int(g['n_observed']), g['y_test'].shape[0]
    lb, xm, xc, mean, covar = model.predict_missing_data(g['y_test'][:, :do], x_test_mean=g['missing_x_test_mean'])
    tol = 1e-9 if prec == 'f64' else 2e-5
    np.testing.assert_allclose(float(lb), float(g['missing_lower_bound_clean']), rtol=tol)
    np.testing.assert_allclose(float(lb) + defect(model, n_test), float(g['missing_lower_bound']), rtol=tol)
    assert mean.shape == g['predicted_mean'].shape and covar.shape == g['predicted_covar'].shape
    np.testing.assert_allclose(mean.cpu().numpy(), g['predicted_mean'], rtol=0, atol=1e-8 * np.abs(g['predicted_mean']).max())
    np.testing.assert_allclose(covar.cpu().numpy(), g['predicted_covar'], rtol=0, atol=1e-8 * np.abs(g['predicted_covar']).max())


def test_default_initialisation_of_test_latents(dev):
    """Nearest training neighbour + N(0, 0.01^2) (dp_gp_lvm.py:251-256), PCA on request, fewer test points than inducing points."""
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    g = golden(FIXTURES[0])
    model = build(dp_gp_lvm, g, dev, 'f64')
    y, x_mean = g['y'], g['x_mean']
    y_test = y[[3, 17, 5]] + 1e-3
    np.random.seed(0)
    lb, xm, xc, ll = model.predict_new_latent_variables(y_test)
    assert xm.shape == (3, x_mean.shape[1]) and bool(torch.isfinite(lb)) and bool(torch.isfinite(ll))
    assert np.abs(xm.cpu().numpy() - x_mean[[3, 17, 5]]).max() < 0.06
    lb2, xm2, _, _ = model.predict_new_latent_variables(y[:20], use_pca=True)
    assert xm2.shape == (20, x_mean.shape[1]) and bool(torch.isfinite(lb2))
    with pytest.raises(AssertionError):
        model.predict_new_latent_variables(y_test[:, :4])
    with pytest.raises(AssertionError):
        model.predict_missing_data(y_test)


def test_test_latent_gradients_and_optimisation(dev):
    """d (f_hat_test - KL) / d q(X*) from the HIP backward pass against central differences of the HIP forward (fp64), and
    Adam on q(X*) improves the test log-likelihood."""
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    g = golden(FIXTURES[1])
    model = build(dp_gp_lvm, g, dev, 'f64')
    y_test = g['y_test']
    xm, xv = g['new_x_test_mean'].copy(), 0.5 + np.random.default_rng(0).random(g['new_x_test_mean'].shape)
    g_mu, g_s = (a.cpu().numpy() for a in model.test_latent_gradients(y_test, xm, xv))

    def ll(xm_, xv_):
        return float(model.predict_new_latent_variables(y_test, x_test_mean=xm_, x_test_var=xv_)[3])
    rng = np.random.default_rng(1)
    for _ in range(4):
        e_mu, e_s = rng.standard_normal(xm.shape), rng.standard_normal(xv.shape)
        h = 1e-5
        fd = (ll(xm + h * e_mu, xv + h * e_s) - ll(xm - h * e_mu, xv - h * e_s)) / (2 * h)
        an = float(np.sum(g_mu * e_mu) + np.sum(g_s * e_s))
        assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (fd, an)
    before = ll(xm, np.ones_like(xm))
    xo, vo = model.optimise_test_latents(y_test, num_iterations=60, learning_rate=0.02, x_test_mean=xm)
    after = ll(xo.cpu().numpy(), vo.cpu().numpy())
    assert after > before + 1.0, (before, after)
    # fewer test points than inducing points (N* = 7 < M = 12)
    g0 = golden(FIXTURES[0])
    m0 = build(dp_gp_lvm, g0, dev, 'mixed')
    x0, v0 = m0.optimise_test_latents(g0['y_test'], num_iterations=5, x_test_mean=g0['new_x_test_mean'])
    assert x0.shape == g0['new_x_test_mean'].shape and bool(torch.isfinite(x0).all()) and bool((v0 > 0).all())
    # missing-data flavour: only the first Do dims observed
    do = int(g['n_observed'])
    gm2, gs2 = model.test_latent_gradients(y_test[:, :do], xm, xv)
    assert gm2.shape == xm.shape and bool(torch.isfinite(gm2).all()) and bool(torch.isfinite(gs2).all())
