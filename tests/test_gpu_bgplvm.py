"""GPU tests of the Bayesian GP-LVM wrapper (reference src/models/gaussian_process.py:132-270) against
fixtures from the reference's own constructor and its tf.gradients (oracle/gen_golden_bgplvm.py)."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
FIXTURES = ['bgplvm_ref_40_6_12_3', 'bgplvm_ref_70_9_20_4']
REF2RAW = dict(gamma_raw='gamma_atoms', alpha_raw='alpha_atoms', beta_raw='beta_atoms', x_mean='x_mean', x_u='x_u', x_var_raw='x_var')


def softplus(x):
    return np.logaddexp(0.0, x)


def build(g, dev, prec):
    from dp_gp_lvm_amd.models.gaussian_process import bayesian_gp_lvm
    return bayesian_gp_lvm(g['y'], num_latent_dims=g['x_mean'].shape[1], num_inducing_points=g['x_u'].shape[0], device=dev,
                           precision=prec, initial_values=dict(x_mean=g['x_mean'], x_var=softplus(g['x_var_raw']), x_u=g['x_u'],
                                                               gamma=softplus(g['gamma_raw']), alpha=softplus(g['alpha_raw']),
                                                               beta=softplus(g['beta_raw'])))


@pytest.mark.parametrize('fixture', FIXTURES)
@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_objective_and_gradients_match_the_reference(dev, fixture, prec):
    g = golden(fixture)
    model = build(g, dev, prec)
    np.testing.assert_allclose(float(model.objective), float(g['objective']), rtol=1e-10 if prec == 'f64' else 2e-6)
    got = model.gradients()
    for ref_name, raw_name in REF2RAW.items():
        want = g['grad_' + ref_name]
        have = got[raw_name].cpu().numpy().reshape(want.shape)
        np.testing.assert_allclose(have, want, rtol=5e-4, atol=5e-4 * max(1.0, np.abs(want).max()), err_msg=ref_name)


def test_default_construction_kernel_argument_and_adam(dev):
    from dp_gp_lvm_amd.kernels.rbf_kernel import k_ard_rbf
    from dp_gp_lvm_amd.models.gaussian_process import bayesian_gp_lvm
    g = golden(FIXTURES[1])
    y = g['y']
    np.random.seed(2)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)
    kern = k_ard_rbf(gamma=t(np.full((1, 4), 0.7)), alpha=t([[1.3]]), beta=t([[2.0]]))
    model = bayesian_gp_lvm(y, kernel=kern, num_latent_dims=4, num_inducing_points=15, device=dev)
    np.testing.assert_allclose(model.ard_weights.cpu().numpy(), 0.7, rtol=1e-12)
    np.testing.assert_allclose(float(model.noise_precision), 2.0, rtol=1e-12)
    np.testing.assert_allclose(torch.diagonal(model.q_x[1], dim1=-2, dim2=-1).cpu().numpy(), 0.5, rtol=1e-12)   # (:218)
    before = float(model.objective)
    model.optimise(25, learning_rate=0.01)
    assert float(model.objective) < before - 1.0
    with pytest.raises(NotImplementedError):
        bayesian_gp_lvm(y, num_latent_dims=4, num_inducing_points=15, num_latent_samples=10, device=dev)
    with pytest.raises(AssertionError):
        bayesian_gp_lvm(y, num_latent_dims=4, num_inducing_points=y.shape[0], device=dev)
