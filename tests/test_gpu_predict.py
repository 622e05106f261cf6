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


@pytest.mark.parametrize('fixture', FIXTURES)
def test_reference_compat_returns_the_references_own_bounds(dev, fixture):
    """reference_compat=True: the bounds exactly as the reference's graph computes them (with the [D x D] broadcast of
    dp_gp_lvm.py:292 / :409), against the numbers its own methods produced (tests/golden/predict_ref_*.npz)."""
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    g = golden(fixture)
    model = build(dp_gp_lvm, g, dev, 'f64')
    lb, _, _, ll = model.predict_new_latent_variables(g['y_test'], x_test_mean=g['new_x_test_mean'], reference_compat=True)
    np.testing.assert_allclose(float(lb), float(g['new_lower_bound']), rtol=1e-9)
    np.testing.assert_allclose(float(ll), float(g['new_test_log_likelihood']), rtol=1e-9)
    do = int(g['n_observed'])
    lb2 = model.predict_missing_data(g['y_test'][:, :do], x_test_mean=g['missing_x_test_mean'], reference_compat=True)[0]
    np.testing.assert_allclose(float(lb2), float(g['missing_lower_bound']), rtol=1e-9)


def test_variable_split_and_mvn_scoring(dev):
    """get_training_variables() / get_prediction_variables() (reference src/utils/types.py:21-37) and mvn_log_pdf
    (src/distributions/normal.py:14-36) as test/frey_faces_prediction.py:164-171,232-238 uses them."""
    from scipy.stats import multivariate_normal
    from dp_gp_lvm_amd.distributions.normal import mvn_log_pdf
    from dp_gp_lvm_amd.models.dp_gp_lvm import dp_gp_lvm
    from dp_gp_lvm_amd.utils import types as ty
    ty.reset_variable_collections()
    g = golden(FIXTURES[1])
    model = build(dp_gp_lvm, g, dev, 'f64')
    train = ty.get_training_variables()
    assert len(train) == 10 and ty.get_prediction_variables() == []      # (the reference's w_1, w_2 are one 2-vector here)
    assert {id(v) for v in train} == {id(v) for v in model.raw.values()}
    do, n_test = int(g['n_observed']), g['y_test'].shape[0]
    _, xm, _, mean, covar = model.predict_missing_data(g['y_test'][:, :do], x_test_mean=g['missing_x_test_mean'])
    pred = ty.get_prediction_variables()
    assert len(pred) == 2 and pred[0].shape == xm.shape and pred[1].shape == xm.shape
    assert len(ty.get_training_variables()) == 10
    # ground-truth log-likelihood of the unobserved dims under the predictive posterior, one dim at a time
    y_u = g['y_test'][:, do:]
    total = 0.0
    for du in range(y_u.shape[1]):
        got = mvn_log_pdf(torch.as_tensor(y_u[:, du][None, :], device=dev), mean[:, du][None, :], covar[du])
        want = multivariate_normal.logpdf(y_u[:, du], mean=mean[:, du].cpu().numpy(), cov=covar[du].cpu().numpy())
        np.testing.assert_allclose(got.cpu().numpy(), [want], rtol=1e-10)
        total += float(got[0])
    assert np.isfinite(total)
    bad = -torch.eye(n_test, dtype=torch.float64, device=dev)
    assert bool(torch.isnan(mvn_log_pdf(torch.zeros((1, n_test), dtype=torch.float64, device=dev),
                                        torch.zeros((1, n_test), dtype=torch.float64, device=dev), bad)).all())
    ty.reset_variable_collections()


def test_training_example_writes_the_reference_result_schema(dev, tmp_path):
    """examples/train_synthetic.py end to end (train, predict held-out rows, save) and the file read back by ResultKeys."""
    import subprocess
    import sys
    import os
    from dp_gp_lvm_amd.utils.constants import ResultKeys
    from dp_gp_lvm_amd.utils.results import load_results
    out = str(tmp_path / 'dp_gp_lvm_synthetic_test.npz')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(repo, 'examples', 'train_synthetic.py'), '--n', '80', '--d', '9', '--m', '12',
                    '--q', '3', '--t', '4', '--iters', '30', '--predict', '6', '--out', out], check=True, timeout=600,
                   stdout=subprocess.DEVNULL)
    res = load_results(out)
    expect = set(ResultKeys) - {ResultKeys.ORIGINAL_DATA, ResultKeys.RANDOMISED_DATA, ResultKeys.NORMALISED_DATA}
    assert expect <= set(res)
    assert res[ResultKeys.TRAINING_DATA].shape == (74, 9) and res[ResultKeys.TEST_DATA].shape == (6, 9)
    assert res[ResultKeys.TEST_INPUT_MEAN].shape == (6, 3) and res[ResultKeys.TEST_INPUT_COVAR].shape == (6, 3, 3)
    assert res[ResultKeys.DP_ASSIGNMENTS].shape == (9, 4) and res[ResultKeys.Q_V_A].shape == (3,)
    assert res[ResultKeys.ARD_WEIGHTS].shape == (9, 3) and np.all(res[ResultKeys.NOISE_PRECISION] > 0)
