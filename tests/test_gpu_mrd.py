"""GPU tests of the manifold-relevance-determination wrapper (reference
src/models/gaussian_process.py:551-664) against fixtures from the reference's own constructor and its tf.gradients
(oracle/gen_golden_mrd.py -> tests/golden/mrd_ref_*.npz)."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
FIXTURES = ['mrd_ref_50_2views_12_3', 'mrd_ref_60_3views_15_4']


def softplus(x):
    return np.logaddexp(0.0, x)


def build(g, dev, prec):
    from dp_gp_lvm_amd.models.gaussian_process import manifold_relevance_determination
    nv = int(g['num_views'])
    views = [g['view_%d' % i] for i in range(nv)]
    iv = dict(x_mean=g['x_mean'], x_var=softplus(g['x_var_raw']), x_u=[g['x_u_%d' % i] for i in range(nv)],
              gamma=[softplus(g['gamma_raw_%d' % i]) for i in range(nv)], alpha=[softplus(g['alpha_raw_%d' % i]) for i in range(nv)],
              beta=[softplus(g['beta_raw_%d' % i]) for i in range(nv)])
    return manifold_relevance_determination(views, num_latent_dims=g['x_mean'].shape[1], num_inducing_points=g['x_u_0'].shape[0],
                                            device=dev, precision=prec, initial_values=iv), views


@pytest.mark.parametrize('fixture', FIXTURES)
@pytest.mark.parametrize('prec', ['f64', 'mixed'])
def test_objective_and_gradients_match_the_reference(dev, fixture, prec):
    g = golden(fixture)
    model, views = build(g, dev, prec)
    assert model.number_of_views == len(views) == len(model.kernels)
    np.testing.assert_allclose(float(model.objective), float(g['objective']), rtol=1e-10 if prec == 'f64' else 2e-6)
    got = model.gradients()
    ref2raw = dict(x_mean='x_mean', x_var_raw='x_var')
    for i in range(len(views)):
        ref2raw.update({'gamma_raw_%d' % i: 'gamma_atoms_%d' % i, 'alpha_raw_%d' % i: 'alpha_atoms_%d' % i,
                        'beta_raw_%d' % i: 'beta_atoms_%d' % i, 'x_u_%d' % i: 'x_u_%d' % i})
    for ref_name, raw_name in ref2raw.items():
        want = g['grad_' + ref_name]
        have = got[raw_name].cpu().numpy().reshape(want.shape)
        np.testing.assert_allclose(have, want, rtol=5e-4, atol=5e-4 * max(1.0, np.abs(want).max()), err_msg=ref_name)


def test_default_construction_and_adam(dev):
    from dp_gp_lvm_amd.models.gaussian_process import manifold_relevance_determination
    g = golden(FIXTURES[0])
    views = [g['view_0'], g['view_1']]
    np.random.seed(3)
    model = manifold_relevance_determination(views, num_latent_dims=3, num_inducing_points=10, device=dev, precision='f64')
    assert [tuple(z.shape) for z in model.inducing_input] == [(10, 3), (10, 3)]
    np.testing.assert_allclose(torch.diagonal(model.q_x[1], dim1=-2, dim2=-1).cpu().numpy(), 1.0, rtol=1e-12)   # (:593)
    before = float(model.objective)
    model.optimise(25, learning_rate=0.01)
    assert float(model.objective) < before - 1.0
    with pytest.raises(AssertionError):
        manifold_relevance_determination([views[0], views[1][:-1]], num_latent_dims=3, num_inducing_points=10, device=dev)
    with pytest.raises(AssertionError):
        manifold_relevance_determination(views, num_latent_dims=12, num_inducing_points=10, device=dev)
