"""
TEST INFRASTRUCTURE, CONTAINER-ONLY: fixtures for the prediction paths.
Run as ``python oracle/gen_golden_predict.py`` in the build container (needs /root/reference; never runs on the GPU box).

Builds the reference's own ``dp_gp_lvm(...)`` under the NumPy stand-in for TensorFlow at steered (non-degenerate) variable
values, exactly as oracle/gen_golden_grad.py does, and calls the reference's own ``predict_new_latent_variables(y_test)``
(src/models/dp_gp_lvm.py:233-309) and ``predict_missing_data(y_test[:, :Do])`` (:311-500) on random test points.  The
reference has no known-answer tests for these (its test_prediction methods are empty), so the fixtures are the reference's
outputs themselves, checked before writing against the pinned NumPy oracle (oracle/dpgp_oracle.py): f_hat(y_test, q(X*)) - KL
with the model's mixed hyper-parameters equals the reference's test log-likelihood UP TO A DEFECT OF THE REFERENCE: in both
prediction methods `tf.trace(...)` has shape [D] while `psi_0_test` and `beta` are [D,1], so
`beta * (tf.trace(...) - psi_0_test)` broadcasts to [D,D] (dp_gp_lvm.py:292, :411-412; the training objective uses a
keepdims sum, :124-126, and is correct).  The reference therefore adds
    defect = 1/2 sum_ij beta_i (tr_j - psi0_i) - 1/2 sum_i beta_i (tr_i - psi0_i)
to its lower bounds.  The product computes the bound without the defect; fixtures store the reference's numbers and the
defect-free ones, and the relation between them is asserted here and in the tests.
Fixtures are data only: y, the raw variable values, y_test, the q(X*) the reference initialised (its nearest-neighbour +
noise draw), and the returned values.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
import numpy as np                                                           # noqa: E402
from oracle import gen_golden_t as gt                                        # noqa: E402

CASES = {   # name: ((N, D, M, Q, T, mask_size, seed), N*, Do)
    'predict_ref_40_6_12_3_T4': ((40, 6, 12, 3, 4, 1, 31), 7, 4),
    'predict_ref_60_10_15_4_T5': ((60, 10, 15, 4, 5, 1, 32), 20, 6),
}


def main():
    from oracle import dpgp_oracle as orc
    for name, (case, n_test, n_obs) in CASES.items():
        tf, model, variables, y = gt.build('standin', case, factory='dp_gp_lvm')
        vals = [np.asarray(v).copy() for v in variables]
        rng = np.random.default_rng(case[-1] + 500)
        # test points: noisy copies of training rows, so that the nearest-neighbour initialisation is meaningful
        y_test = y[rng.choice(y.shape[0], n_test, replace=False)] + 0.1 * rng.standard_normal((n_test, y.shape[1]))
        np.random.seed(case[-1])
        lb, xt_mean, xt_covar, test_ll = (np.asarray(a) for a in model.predict_new_latent_variables(y_test))
        np.random.seed(case[-1] + 1)
        mlb, xm_mean, xm_covar, pmean, pcovar = (np.asarray(a) for a in model.predict_missing_data(y_test[:, :n_obs]))
        # independent check of the test log-likelihood through the pinned oracle
        gamma, alpha, beta = (np.asarray(a) for a in (model.ard_weights, model.signal_variance, model.noise_precision))
        z = np.asarray(model.inducing_input)
        s_t = np.stack([np.diag(c) for c in xt_covar])
        def defect(terms_, al_, be_, n_t):
            tr = 2.0 * terms_[:, 2] / be_ + al_ * n_t                                   # <K^-1, Psi2*> per output dim
            psi0 = al_ * n_t
            return 0.5 * float(np.sum(be_[:, None] * (tr[None, :] - psi0[:, None]))) - 0.5 * float(np.sum(be_ * (tr - psi0)))
        al1, be1 = alpha.reshape(-1), beta.reshape(-1)
        terms = orc.fhat_terms(y_test, z, xt_mean, s_t, gamma, alpha, beta)
        clean_ll = terms.sum() - orc.kl_qx(xt_mean, s_t)
        np.testing.assert_allclose(clean_ll + defect(terms, al1, be1, n_test), float(test_ll), rtol=1e-10)
        s_m = np.stack([np.diag(c) for c in xm_covar])
        mu_tr, cov_tr = model.q_x
        s_tr = np.stack([np.diag(c) for c in np.asarray(cov_tr)])
        f_tr = orc.fhat_terms(y, z, np.asarray(mu_tr), s_tr, gamma, alpha, beta).sum() - orc.kl_qx(np.asarray(mu_tr), s_tr)
        t_obs = orc.fhat_terms(y_test[:, :n_obs], z, xm_mean, s_m, gamma[:n_obs], alpha[:n_obs], beta[:n_obs])
        f_obs = t_obs.sum() - orc.kl_qx(xm_mean, s_m)
        np.testing.assert_allclose(f_tr + f_obs + defect(t_obs, al1[:n_obs], be1[:n_obs], n_test), float(mlb), rtol=1e-10)
        np.testing.assert_allclose(f_tr + clean_ll + defect(terms, al1, be1, n_test), float(lb), rtol=1e-10)
        np.savez_compressed(os.path.join(gt.OUT, name + '.npz'), y=y, y_test=y_test, n_observed=n_obs, s_1=1.0, s_2=1.0,
                            mask_size=case[5], **dict(zip(gt.NAMES, vals)),
                            new_lower_bound=float(lb), new_x_test_mean=xt_mean, new_x_test_covar=xt_covar,
                            new_test_log_likelihood=float(test_ll), new_test_log_likelihood_clean=float(clean_ll),
                            new_lower_bound_clean=float(f_tr + clean_ll), missing_lower_bound_clean=float(f_tr + f_obs),
                            missing_lower_bound=float(mlb), missing_x_test_mean=xm_mean, missing_x_test_covar=xm_covar,
                            predicted_mean=pmean, predicted_covar=pcovar)
        print('wrote %s: lower bound %.10f, test log-likelihood %.10f; missing-data lower bound %.10f, predicted mean %s, '
              'covariance %s' % (name, float(lb), float(test_ll), float(mlb), pmean.shape, pcovar.shape))


if __name__ == '__main__':
    main()
