"""GPU tests at an ILL-CONDITIONED operating point (tests/golden/illcond_ref_*.npz, oracle/gen_golden_illcond.py: the
reference's own objective and tf.gradients after Adam has driven K_uu towards singularity, cond ~1e5, the conditioning-guard
bound ~80 x its threshold).  Every precision mode must either match the reference to its stated tolerance or raise its flag —
never diverge silently:
    f64                      objective 1e-8, gradients 1e-5 of the largest entry, no flag
    f64 + mixed backward     (the training configuration) objective 1e-8, gradients 2e-3 of the largest entry, no flag
    mixed                    flagged (info = DPGP_INFO_ILL_CONDITIONED = -2) and optimise() raises; where a dim is NOT flagged its
                             terms match the fp64 terms to the mixed tolerance
The guard values themselves are checked against the oracle's restatement of the bound."""
import numpy as np
import pytest
import torch

from conftest import golden
from test_gpu_grad import REF2RAW, build_model

pytestmark = pytest.mark.gpu
FIXTURE = 'illcond_ref_300_8_64_3_T4'
ILL_CONDITIONED = -2            # DPGP_INFO_ILL_CONDITIONED (include/dpgp.h)


def _check_gradients(model, g, tol):
    got = model.gradients()
    for ref_name, raw_name in REF2RAW.items():
        want = g['grad_' + ref_name]
        np.testing.assert_allclose(got[raw_name].cpu().numpy().reshape(want.shape), want, rtol=0,
                                   atol=tol * np.abs(want).max(), err_msg=ref_name)
    want_w = np.array([float(g['grad_w_1_raw']), float(g['grad_w_2_raw'])])
    np.testing.assert_allclose(got['dp_w'].cpu().numpy().reshape(-1), want_w, rtol=0, atol=tol * np.abs(want_w).max())


@pytest.mark.parametrize('backward', [None, 'mixed'])
def test_fp64_matches_the_reference_where_kuu_is_nearly_singular(dev, backward):
    g = golden(FIXTURE)
    model = build_model(g, dev, 'f64', backward_precision=backward)
    np.testing.assert_allclose(float(model.objective), float(g['objective']), rtol=1e-8)
    terms, info = model.per_dimension_terms
    assert int(info.abs().max()) == 0
    # the bound is computed in every mode; with an fp64 Psi2 it is the oracle's number
    np.testing.assert_allclose(model.conditioning_guard.cpu().numpy(), g['guard'], rtol=1e-6)
    _check_gradients(model, g, 1e-5 if backward is None else 2e-3)
    # the training configuration picks the form of the Psi2 term by this bound: here (80 x the threshold) the patch form
    assert model.last_stage_b_form == (None if backward is None else 'mixed_patch')


def test_mixed_precision_raises_its_flag_instead_of_diverging(dev):
    g = golden(FIXTURE)
    n = g['y'].shape[0]
    mixed, ref = build_model(g, dev, 'mixed'), build_model(g, dev, 'f64')
    obj_mixed, obj_ref = float(mixed.objective), float(ref.objective)
    terms, info = (a.cpu().numpy() for a in mixed.per_dimension_terms)
    terms64 = ref.per_dimension_terms[0].cpu().numpy()
    guard = mixed.conditioning_guard.cpu().numpy()
    np.testing.assert_allclose(guard, g['guard'], rtol=1e-3)             # (fp32 Psi2 inside the norm)
    flagged = info == ILL_CONDITIONED
    assert np.array_equal(flagged, guard > 2.0e-3 * n) and flagged.any()
    assert set(np.unique(info)) <= {0, ILL_CONDITIONED}                   # the factorisations themselves still succeed here
    ok = ~flagged
    if ok.any():                                                          # unflagged dims keep the mixed tolerance
        np.testing.assert_allclose(terms[ok], terms64[ok], rtol=0, atol=2e-5 * np.abs(terms64[ok]).max())
    assert np.isfinite(obj_mixed) and abs(obj_mixed - obj_ref) <= 1e-3 * abs(obj_ref)   # flagged terms are numbers, not NaN
    with pytest.raises(FloatingPointError):
        mixed.optimise(2)
    before = {k: v.clone() for k, v in mixed.raw.items()}
    with pytest.raises(FloatingPointError):
        mixed.optimise(1)
    for k, v in mixed.raw.items():                                        # no update was applied on the flagged step
        assert torch.equal(v, before[k]), k


def test_training_configuration_keeps_descending_from_the_ill_conditioned_point(dev):
    g = golden(FIXTURE)
    model = build_model(g, dev, 'f64', backward_precision='mixed')
    before = float(model.objective)
    model.optimise(40, learning_rate=0.01)
    after = float(model.objective)
    assert int(model.per_dimension_terms[1].abs().max()) == 0
    assert np.isfinite(after) and after < before
