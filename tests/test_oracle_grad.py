"""CPU tests of the gradient oracle (oracle/dpgp_oracle_torch.py) against the gradient fixtures that
oracle/gen_golden_grad.py produced by differentiating the reference's own objective (tests/golden/grad_ref_*.npz).
Groundwork for the backward pass of the fused ELBO: the HIP gradients will be checked against this
oracle, which is pinned here."""
import glob
import os

import numpy as np
import pytest

from oracle import dpgp_oracle as orc
from oracle import dpgp_oracle_torch as ot

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, 'grad_ref_*.npz')))


def softplus(x):
    return np.logaddexp(0.0, x)


def test_gradient_fixtures_exist():
    assert len(FIXTURES) >= 2


@pytest.mark.parametrize('path', FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_torch_oracle_reproduces_reference_objective_and_gradients(path):
    g = np.load(path)
    obj, grads = ot.objective_and_gradients(g['y'], {k: g[k] for k in ot.NAMES}, s_1=float(g['s_1']), s_2=float(g['s_2']),
                                            mask_size=int(g['mask_size']))
    np.testing.assert_allclose(obj, float(g['objective']), rtol=1e-11)
    for k in ot.NAMES:
        ref = g['grad_' + k]
        np.testing.assert_allclose(grads[k], ref, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(ref).max()), err_msg=k)


@pytest.mark.parametrize('path', FIXTURES, ids=[os.path.basename(p) for p in FIXTURES])
def test_numpy_oracle_agrees_at_the_gradient_fixture_points(path):
    """The forward oracle that checks the HIP path (oracle/dpgp_oracle.py) evaluated at the same raw variables."""
    g = np.load(path)
    e = np.exp(g['dp_logits'] - g['dp_logits'].max(axis=1, keepdims=True))
    phi = e / e.sum(axis=1, keepdims=True)
    obj = orc.objective(g['y'], g['x_u'], g['x_mean'], softplus(g['x_var_raw']), phi, softplus(g['gamma_atoms_raw']),
                        softplus(g['alpha_atoms_raw']), softplus(g['beta_atoms_raw']), softplus(g['gamma_1_raw']),
                        softplus(g['gamma_2_raw']), float(softplus(g['w_1_raw'])), float(softplus(g['w_2_raw'])),
                        float(g['s_1']), float(g['s_2']))
    np.testing.assert_allclose(obj, float(g['objective']), rtol=1e-10)


@pytest.mark.parametrize('path', FIXTURES[:1], ids=[os.path.basename(p) for p in FIXTURES[:1]])
def test_gradient_is_consistent_with_finite_differences_of_the_numpy_oracle(path):
    g = np.load(path)
    raw = {k: g[k].astype(np.float64) for k in ot.NAMES}
    _, grads = ot.objective_and_gradients(g['y'], raw, s_1=float(g['s_1']), s_2=float(g['s_2']))

    def f(r):
        return ot.objective_and_gradients(g['y'], r, s_1=float(g['s_1']), s_2=float(g['s_2']))[0]
    rs = np.random.default_rng(3)
    dirs = {k: rs.standard_normal(np.shape(v)) for k, v in raw.items()}
    h = 1e-6
    fd = (f({k: raw[k] + h * dirs[k] for k in raw}) - f({k: raw[k] - h * dirs[k] for k in raw})) / (2 * h)
    an = sum(float(np.sum(grads[k] * dirs[k])) for k in raw)
    assert abs(fd - an) <= 1e-6 * max(1.0, abs(an))


# ---- over-T formulation: fixtures from the reference's own dp_gp_lvm_t (oracle/gen_golden_t.py) ----
FIXTURES_T = sorted(glob.glob(os.path.join(GOLDEN, 'model_t_ref_*.npz')))


def test_over_t_fixtures_exist():
    assert len(FIXTURES_T) >= 2


@pytest.mark.parametrize('path', FIXTURES_T, ids=[os.path.basename(p) for p in FIXTURES_T])
def test_torch_oracle_reproduces_the_over_t_reference(path):
    g = np.load(path)
    obj, grads = ot.objective_t_and_gradients(g['y'], {k: g[k] for k in ot.NAMES}, s_1=float(g['s_1']), s_2=float(g['s_2']),
                                              mask_size=int(g['mask_size']))
    np.testing.assert_allclose(obj, float(g['objective']), rtol=1e-11)
    for k in ot.NAMES:
        ref = g['grad_' + k]
        np.testing.assert_allclose(grads[k], ref, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(ref).max()), err_msg=k)
    # the reference's known answer (dpgplvm_unitttests.py:544-548): with equal atoms the over-T and the over-D objectives agree
    init = {k: g['init_' + k] for k in ot.NAMES}
    o_t, _ = ot.objective_t_and_gradients(g['y'], init, s_1=float(g['s_1']), s_2=float(g['s_2']))
    o_d, _ = ot.objective_and_gradients(g['y'], init, s_1=float(g['s_1']), s_2=float(g['s_2']))
    np.testing.assert_allclose(o_t, float(g['objective_init']), rtol=1e-11)
    np.testing.assert_allclose(o_d, o_t, rtol=1e-9)
    # away from equal atoms they are different models
    o_d2, _ = ot.objective_and_gradients(g['y'], {k: g[k] for k in ot.NAMES}, s_1=float(g['s_1']), s_2=float(g['s_2']))
    np.testing.assert_allclose(o_d2, float(g['objective_over_d']), rtol=1e-10)
    assert abs(o_d2 - obj) > 1e-3


# ---- Bayesian GP-LVM: fixtures from the reference's own bayesian_gp_lvm (oracle/gen_golden_bgplvm.py) ----
FIXTURES_B = sorted(glob.glob(os.path.join(GOLDEN, 'bgplvm_ref_*.npz')))


@pytest.mark.parametrize('path', FIXTURES_B, ids=[os.path.basename(p) for p in FIXTURES_B])
def test_torch_oracle_reproduces_the_bgplvm_reference(path):
    g = np.load(path)
    obj, grads = ot.objective_bgplvm_and_gradients(g['y'], {k: g[k] for k in ot.BGPLVM_NAMES})
    np.testing.assert_allclose(obj, float(g['objective']), rtol=1e-11)
    for k in ot.BGPLVM_NAMES:
        ref = g['grad_' + k]
        np.testing.assert_allclose(grads[k], ref, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(ref).max()), err_msg=k)
    assert len(FIXTURES_B) >= 2


ILLCOND = sorted(glob.glob(os.path.join(GOLDEN, 'illcond_ref_*.npz')))


@pytest.mark.parametrize('path', ILLCOND, ids=[os.path.basename(p) for p in ILLCOND])
def test_oracles_at_the_ill_conditioned_point(path):
    """tests/golden/illcond_ref_*.npz (oracle/gen_golden_illcond.py): the reference's own objective and tf.gradients after
    Adam has driven K_uu towards singularity.  Both oracles must reproduce them (they are the checkers of the GPU tests at
    this point), and the fixture must really be in the regime where the fp32 path's guard fires."""
    g = np.load(path)
    assert len(ILLCOND) >= 1 and g['kuu_condition'].max() > 1e5
    n = g['y'].shape[0]
    assert g['guard'].max() > 10 * 2.0e-3 * n                       # 10 x DPGP_GUARD_REL N (include/dpgp.h)
    obj, grads = ot.objective_and_gradients(g['y'], {k: g[k] for k in ot.NAMES}, s_1=float(g['s_1']), s_2=float(g['s_2']),
                                            mask_size=int(g['mask_size']))
    np.testing.assert_allclose(obj, float(g['objective']), rtol=1e-9)
    for k in ot.NAMES:
        ref = g['grad_' + k]
        np.testing.assert_allclose(grads[k], ref, rtol=0, atol=1e-6 * max(1e-300, np.abs(ref).max()), err_msg=k)
    e = np.exp(g['dp_logits'] - g['dp_logits'].max(axis=1, keepdims=True))
    phi = e / e.sum(axis=1, keepdims=True)
    obj_np = orc.objective(g['y'], g['x_u'], g['x_mean'], softplus(g['x_var_raw']), phi, softplus(g['gamma_atoms_raw']),
                           softplus(g['alpha_atoms_raw']), softplus(g['beta_atoms_raw']), softplus(g['gamma_1_raw']),
                           softplus(g['gamma_2_raw']), float(softplus(g['w_1_raw'])), float(softplus(g['w_2_raw'])),
                           float(g['s_1']), float(g['s_2']))
    np.testing.assert_allclose(obj_np, float(g['objective']), rtol=1e-9)
