"""Result files of converged models in the reference's .npz schema (src/utils/constants.py:38-72, written e.g. by
test/frey_faces_prediction.py:242-249 and read by its analyse_*.py scripts): every array is stored under the string value of
its ``ResultKeys`` member, so files written here load in the reference's tooling and vice versa."""
import numpy as np

from .constants import ResultKeys


def _np(a):
    return a.detach().cpu().numpy() if hasattr(a, 'detach') else np.asarray(a)


def collect_results(model, y_train, y_test=None, x_test_mean=None, x_test_covar=None, extra=None):
    """{ResultKeys value: numpy array} of a trained dp_gp_lvm / dp_gp_lvm_t model (the accessors of dp_gp_lvm.py:161-231)."""
    x_mean, x_covar = model.q_x
    gat, aat, bat = model.dp_atoms
    phi = model.assignments
    if hasattr(model, 'ard_weights'):                    # over-D model: mixed per-output hyper-parameters
        ard, sig, noise = model.ard_weights, model.signal_variance, model.noise_precision
    else:                                                # over-T model: the mixture is outside the kernel; same mixing
        ard, sig, noise = phi @ gat, phi @ aat, phi @ bat
    w_1, w_2 = model.dp.q_alpha
    v_a, v_b = model.dp.q_v
    out = {
        ResultKeys.TRAINING_DATA: y_train, ResultKeys.TRAINING_INPUT_MEAN: x_mean, ResultKeys.TRAINING_INPUT_COVAR: x_covar,
        ResultKeys.INDUCING_INPUT: model.inducing_input, ResultKeys.ARD_WEIGHTS: ard, ResultKeys.SIGNAL_VARIANCE: sig,
        ResultKeys.NOISE_PRECISION: noise, ResultKeys.DP_ASSIGNMENTS: phi, ResultKeys.Q_ALPHA_W1: w_1,
        ResultKeys.Q_ALPHA_W2: w_2, ResultKeys.Q_V_A: v_a, ResultKeys.Q_V_B: v_b, ResultKeys.ARD_WEIGHTS_ATOMS: gat,
        ResultKeys.SIGNAL_VARIANCE_ATOMS: aat, ResultKeys.NOISE_PRECISION_ATOMS: bat,
    }
    if y_test is not None:
        out[ResultKeys.TEST_DATA] = y_test
    if x_test_mean is not None:
        out[ResultKeys.TEST_INPUT_MEAN] = x_test_mean
    if x_test_covar is not None:
        out[ResultKeys.TEST_INPUT_COVAR] = x_test_covar
    res = {k.value: _np(v) for k, v in out.items()}
    for k, v in (extra or {}).items():
        res[k.value if isinstance(k, ResultKeys) else str(k)] = _np(v)
    return res


def save_results(path, model, y_train, **kw):
    res = collect_results(model, y_train, **kw)
    np.savez(path, **res)
    return res


def load_results(path):
    """{ResultKeys member: array} for the members present in the file, plus the remaining arrays under their file names."""
    values = {k.value: k for k in ResultKeys}
    with np.load(path) as f:
        return {values.get(name, name): f[name] for name in f.files}
