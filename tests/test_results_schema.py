"""CPU tests of the host-side pieces around the hot path that the reference's experiment scripts rely on (SURVEY.md 8f rows 2
and 4): the result-file schema (src/utils/constants.py:38-72) and the trainable / prediction variable collections
(src/utils/types.py:21-37)."""
import numpy as np
import torch

from dp_gp_lvm_amd.utils import types as ty
from dp_gp_lvm_amd.utils.constants import DataSetKeys, ResultKeys
from dp_gp_lvm_amd.utils.results import collect_results, load_results, save_results

# the key strings of the reference's files, re-typed from src/utils/constants.py:17-72
REFERENCE_RESULT_KEYS = {
    'ORIGINAL_DATA': 'original_data', 'RANDOMISED_DATA': 'randomised_data', 'NORMALISED_DATA': 'normalised_data',
    'TRAINING_DATA': 'y_train', 'TRAINING_INPUT_MEAN': 'x_mean', 'TRAINING_INPUT_COVAR': 'x_covar', 'INDUCING_INPUT': 'x_u',
    'TEST_DATA': 'y_test', 'TEST_INPUT_MEAN': 'x_test_mean', 'TEST_INPUT_COVAR': 'x_test_covar', 'ARD_WEIGHTS': 'ard_weights',
    'SIGNAL_VARIANCE': 'signal_variance', 'NOISE_PRECISION': 'noise_precision', 'DP_ASSIGNMENTS': 'assignments',
    'Q_ALPHA_W1': 'q_alpha_w1', 'Q_ALPHA_W2': 'q_alpha_w2', 'Q_V_A': 'q_v_a', 'Q_V_B': 'q_v_b', 'ARD_WEIGHTS_ATOMS': 'gamma_atoms',
    'SIGNAL_VARIANCE_ATOMS': 'alpha_atoms', 'NOISE_PRECISION_ATOMS': 'beta_atoms'}
REFERENCE_DATASET_KEYS = {
    'FULL_DATA_SET': 'full_data_set', 'TRAINING_DATA': 'training_data', 'TEST_DATA': 'test_data',
    'OBSERVED_TEST_DATA': 'observed_test_data', 'UNOBSERVED_TEST_DATA': 'unobserved_test_data',
    'NUM_OBSERVATIONS': 'num_observations', 'NUM_DIMENSIONS': 'num_dimensions', 'NUM_TRAINING_SAMPLES': 'num_training_samples',
    'NUM_TEST_SAMPLES': 'num_test_samples', 'NUM_OBSERVED_DIMENSIONS': 'num_observed_dimensions',
    'NUM_UNOBSERVED_DIMENSIONS': 'num_unobserved_dimensions'}


def test_key_enums_are_the_references():
    assert {k.name: k.value for k in ResultKeys} == REFERENCE_RESULT_KEYS
    assert {k.name: k.value for k in DataSetKeys} == REFERENCE_DATASET_KEYS


class _FakeDP:
    q_alpha = (torch.tensor(1.5), torch.tensor(2.5))
    q_v = (torch.ones(3), 2.0 * torch.ones(3))


class _FakeModel:
    """The accessor surface of dp_gp_lvm (reference dp_gp_lvm.py:161-231) on CPU tensors."""
    n, d, m, q, t = 6, 5, 3, 2, 4
    q_x = (torch.zeros(6, 2), torch.eye(2).expand(6, 2, 2))
    dp_atoms = (torch.ones(4, 2), torch.ones(4, 1), 2.0 * torch.ones(4, 1))
    assignments = torch.full((5, 4), 0.25)
    ard_weights, signal_variance, noise_precision = torch.ones(5, 2), torch.ones(5, 1), 2.0 * torch.ones(5, 1)
    inducing_input = torch.zeros(3, 2)
    dp = _FakeDP()


def test_result_file_round_trip(tmp_path):
    y = np.arange(30.0).reshape(6, 5)
    path = str(tmp_path / 'dp_gp_lvm_synthetic_test.npz')
    written = save_results(path, _FakeModel(), y, y_test=y[:2], x_test_mean=np.zeros((2, 2)), x_test_covar=np.zeros((2, 2, 2)),
                           extra={'train_opt_time': 1.25})
    model_keys = set(REFERENCE_RESULT_KEYS.values()) - {'original_data', 'randomised_data', 'normalised_data'}
    assert model_keys <= set(written)                     # every model / test key of the schema is written
    back = load_results(path)
    for k in ResultKeys:
        if k.value in model_keys:
            np.testing.assert_array_equal(back[k], written[k.value])
    assert back[ResultKeys.Q_ALPHA_W1] == 1.5 and back[ResultKeys.Q_V_B].shape == (3,)
    assert back[ResultKeys.TRAINING_INPUT_COVAR].shape == (6, 2, 2) and float(back['train_opt_time']) == 1.25
    # over-T models have no mixed per-output accessors: the same mixing phi @ atoms is written
    over_t = _FakeModel()
    over_t.__class__ = type('T', (), {k: v for k, v in vars(_FakeModel).items()
                                       if k not in ('ard_weights', 'signal_variance', 'noise_precision')})
    res = collect_results(over_t, y)
    np.testing.assert_allclose(res['noise_precision'], 2.0 * np.ones((5, 1)))


def test_training_and_prediction_variable_collections():
    ty.reset_variable_collections()
    a, b = ty.register_variable(torch.zeros(2)), ty.register_variable(torch.ones(3))
    c = ty.register_variable(torch.full((4,), 2.0), trainable=False)
    train, pred = ty.get_training_variables(), ty.get_prediction_variables()
    assert [id(v) for v in train] == [id(a), id(b)]        # creation order, as tf.get_collection gives it
    assert [id(v) for v in pred] == [id(c)]
    ty.reset_variable_collections()
    assert ty.get_training_variables() == [] and ty.get_prediction_variables() == []
