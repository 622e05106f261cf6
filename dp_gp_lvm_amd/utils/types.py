"""
Data types and positive-variable parameterisation, mirroring the reference's src/utils/types.py:13-72 on torch tensors:
parameters are fp64 (the reference's TF_DTYPE) and positive quantities are softplus(raw variable).
"""
import numpy as np
import torch

TORCH_DTYPE = torch.float64   # reference: TF_DTYPE = tf.float64 (types.py:13)
NP_DTYPE = np.float64


def default_device():
    """Parameters live on the GPU; there is no CPU execution path for the operators."""
    if not torch.cuda.is_available():
        raise RuntimeError('dp_gp_lvm_amd needs an AMD GPU (ROCm): its operators are HIP kernels with no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def inverse_softplus(value):
    """log(exp(v) - 1): the raw value whose softplus is v (types.py:52)."""
    return np.log(np.expm1(value))


def create_positive_variable(initial_value, shape=None, device=None):
    """Raw fp64 variable whose softplus equals ``initial_value`` everywhere (types.py:40-57). Returns the RAW tensor;
    the positive value is torch.nn.functional.softplus(raw)."""
    assert initial_value > 0, 'Initial value must be positive.'
    raw = inverse_softplus(initial_value) * np.ones(shape=() if shape is None else shape, dtype=NP_DTYPE)
    return torch.as_tensor(raw, dtype=TORCH_DTYPE, device=device)


def create_random_positive_variable(shape, device=None):
    """Raw variable ~ N(0,1) (types.py:60-72)."""
    return torch.as_tensor(np.random.standard_normal(size=shape), dtype=TORCH_DTYPE, device=device)


# ---- variable collections: the reference's tf.GraphKeys.TRAINABLE_VARIABLES / GLOBAL_VARIABLES (types.py:21-37) ----
# A model constructor registers its raw variables as trainable; the prediction methods register q(X*) (mean and raw
# variance) as non-trainable — exactly the split the reference's experiment scripts rely on
# (test/frey_faces_prediction.py:164-171: one Adam over get_training_variables(), a second over get_prediction_variables()).
_COLLECTIONS = {'trainable': [], 'global': []}


def register_variable(tensor, trainable=True):
    """Add a raw variable (torch tensor, updated in place by optimisers) to the collections; returns it."""
    _COLLECTIONS['global'].append(tensor)
    if trainable:
        _COLLECTIONS['trainable'].append(tensor)
    return tensor


def reset_variable_collections():
    """tf.reset_default_graph() for the two collections."""
    _COLLECTIONS['trainable'].clear()
    _COLLECTIONS['global'].clear()


def get_training_variables():
    """List of the trainable raw variables of every model built since the last reset (types.py:21-26)."""
    return list(_COLLECTIONS['trainable'])


def get_prediction_variables():
    """The non-trainable variables: q(X*) of the prediction methods, optimised at test time (types.py:29-37)."""
    train = {id(v) for v in _COLLECTIONS['trainable']}
    return [v for v in _COLLECTIONS['global'] if id(v) not in train]
