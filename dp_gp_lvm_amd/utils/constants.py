"""Constants of the DP-GP-LVM path, same names and values as the reference's src/utils/constants.py:17-72,85-121.
The two Enum classes are the key names of the reference's data-set and result .npz files (constants.py:17-72): files written
with them can be read by the reference's analyse_*.py scripts and vice versa."""
from enum import Enum

import numpy as np


class DataSetKeys(Enum):
    """Array names in data-set files (reference: src/utils/constants.py:17-35)."""
    FULL_DATA_SET = 'full_data_set'
    TRAINING_DATA = 'training_data'
    TEST_DATA = 'test_data'
    OBSERVED_TEST_DATA = 'observed_test_data'
    UNOBSERVED_TEST_DATA = 'unobserved_test_data'
    NUM_OBSERVATIONS = 'num_observations'
    NUM_DIMENSIONS = 'num_dimensions'
    NUM_TRAINING_SAMPLES = 'num_training_samples'
    NUM_TEST_SAMPLES = 'num_test_samples'
    NUM_OBSERVED_DIMENSIONS = 'num_observed_dimensions'
    NUM_UNOBSERVED_DIMENSIONS = 'num_unobserved_dimensions'


class ResultKeys(Enum):
    """Array names in result files of converged models (reference: src/utils/constants.py:38-72)."""
    ORIGINAL_DATA = 'original_data'
    RANDOMISED_DATA = 'randomised_data'
    NORMALISED_DATA = 'normalised_data'
    TRAINING_DATA = 'y_train'
    TRAINING_INPUT_MEAN = 'x_mean'
    TRAINING_INPUT_COVAR = 'x_covar'
    INDUCING_INPUT = 'x_u'
    TEST_DATA = 'y_test'
    TEST_INPUT_MEAN = 'x_test_mean'
    TEST_INPUT_COVAR = 'x_test_covar'
    ARD_WEIGHTS = 'ard_weights'
    SIGNAL_VARIANCE = 'signal_variance'
    NOISE_PRECISION = 'noise_precision'
    DP_ASSIGNMENTS = 'assignments'
    Q_ALPHA_W1 = 'q_alpha_w1'
    Q_ALPHA_W2 = 'q_alpha_w2'
    Q_V_A = 'q_v_a'
    Q_V_B = 'q_v_b'
    ARD_WEIGHTS_ATOMS = 'gamma_atoms'
    SIGNAL_VARIANCE_ATOMS = 'alpha_atoms'
    NOISE_PRECISION_ATOMS = 'beta_atoms'


OPT_DEFAULT_LEARNING_RATE = 0.05
OPT_DEFAULT_ITERS = 901
OPT_MAX_ITERS = 2501

GP_DEFAULT_JITTER = 1.0e-8
GP_INIT_GAMMA = 1.0
GP_INIT_ALPHA = 1.0
GP_INIT_BETA = 1.0

GP_LVM_DEFAULT_LATENT_DIMENSIONS = 10
GP_LVM_DEFAULT_NUM_INDUCING_POINTS = 25

DP_DEFAULT_ALPHA_PRIOR_PARAMS = np.array([1.0, 1.0], dtype=np.float64)
DP_DEFAULT_TRUNCATION_LEVEL = 8
