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
