"""CPU test of the prediction fixtures (tests/golden/predict_ref_*.npz, written by oracle/gen_golden_predict.py from the
reference's own predict_new_latent_variables / predict_missing_data): the pinned NumPy oracle reproduces the reference's lower
bounds once the reference's broadcasting defect (generator header) is added, and the defect-free values the product is held to."""
import glob
import os

import numpy as np
import pytest

from oracle import dpgp_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, 'predict_ref_*.npz')))


def softplus(x):
    return np.logaddexp(0.0, x)


def test_prediction_fixtures_exist():
    assert len(FIXTURES) >= 2


@pytest.mark.parametrize('path', FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_oracle_reproduces_the_reference_bounds(path):
    g = np.load(path)
    phi = np.exp(g['dp_logits'] - g['dp_logits'].max(axis=1, keepdims=True))
    phi /= phi.sum(axis=1, keepdims=True)
    gamma, alpha, beta = orc.mix_hyperparameters(phi, softplus(g['gamma_atoms_raw']), softplus(g['alpha_atoms_raw']),
                                                 softplus(g['beta_atoms_raw']))
    al, be = np.asarray(alpha).reshape(-1), np.asarray(beta).reshape(-1)
    z, y, y_test, do = g['x_u'], g['y'], g['y_test'], int(g['n_observed'])
    n_t = y_test.shape[0]

    def defect(terms, a_, b_):
        tr = 2.0 * terms[:, 2] / b_ + a_ * n_t
        psi0 = a_ * n_t
        return 0.5 * float(np.sum(b_[:, None] * (tr[None, :] - psi0[:, None]))) - 0.5 * float(np.sum(b_ * (tr - psi0)))
    s_tr = softplus(g['x_var_raw'])
    f_tr = orc.fhat_terms(y, z, g['x_mean'], s_tr, gamma, alpha, beta).sum() - orc.kl_qx(g['x_mean'], s_tr)
    xt = g['new_x_test_mean']
    st = np.stack([np.diag(c) for c in g['new_x_test_covar']])
    terms = orc.fhat_terms(y_test, z, xt, st, gamma, alpha, beta)
    ll = terms.sum() - orc.kl_qx(xt, st)
    np.testing.assert_allclose(ll, float(g['new_test_log_likelihood_clean']), rtol=1e-11)
    np.testing.assert_allclose(ll + defect(terms, al, be), float(g['new_test_log_likelihood']), rtol=1e-10)
    np.testing.assert_allclose(f_tr + ll + defect(terms, al, be), float(g['new_lower_bound']), rtol=1e-10)
    xm = g['missing_x_test_mean']
    sm = np.stack([np.diag(c) for c in g['missing_x_test_covar']])
    t_obs = orc.fhat_terms(y_test[:, :do], z, xm, sm, gamma[:do], alpha[:do], beta[:do])
    f_obs = t_obs.sum() - orc.kl_qx(xm, sm)
    np.testing.assert_allclose(f_tr + f_obs, float(g['missing_lower_bound_clean']), rtol=1e-11)
    np.testing.assert_allclose(f_tr + f_obs + defect(t_obs, al[:do], be[:do]), float(g['missing_lower_bound']), rtol=1e-10)
    assert g['predicted_mean'].shape == (n_t, y.shape[1] - do) and g['predicted_covar'].shape == (y.shape[1] - do, n_t, n_t)
