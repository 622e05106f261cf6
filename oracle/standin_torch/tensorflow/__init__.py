"""
TEST INFRASTRUCTURE, CONTAINER-ONLY.  Eager PyTorch (fp64, CPU) module that answers to the name ``tensorflow``.

Same purpose and same rules as ``oracle/standin/tensorflow`` (the NumPy stand-in): it lets the reference's *own,
unmodified* source run without TensorFlow 1.15.  This variant computes every "graph node" as a torch tensor, so the
reference's objective carries an autograd graph and ``tf.gradients`` (what ``tf.train.AdamOptimizer.minimize`` builds,
test/synthetic_data_hard_test.py:143-155) is available: ``oracle/gen_golden_grad.py`` uses it to write gradient fixtures
for the backward pass of the fused ELBO.  It contains no DP-GP-LVM logic: each function is the textbook
meaning of the TensorFlow op of the same name.  Nothing in ``dp_gp_lvm_amd/``, ``bench.py`` or the GPU tests imports it.
"""

import numpy as _np
import torch as _th

float64 = _th.float64
float32 = _th.float32
int32 = _th.int32
int64 = _th.int64

_TRAINABLE = []
_GLOBAL = []


class GraphKeys:
    TRAINABLE_VARIABLES = 'trainable_variables'
    GLOBAL_VARIABLES = 'variables'


def get_collection(key):
    return list(_TRAINABLE if key == GraphKeys.TRAINABLE_VARIABLES else _GLOBAL)


def reset_default_graph():
    del _TRAINABLE[:]
    del _GLOBAL[:]


def _arr(x, dtype=None):
    if isinstance(x, _th.Tensor):
        return x if dtype is None or x.dtype == dtype else x.to(dtype)
    a = _np.asarray(x)
    if dtype is None:
        dtype = _th.float64 if a.dtype.kind == 'f' else None
    return _th.as_tensor(a, dtype=dtype)


def Variable(initial_value=None, dtype=None, trainable=True, **_):
    v = _arr(initial_value, dtype).detach().clone()
    if v.dtype.is_floating_point:
        v.requires_grad_(bool(trainable))
    _GLOBAL.append(v)
    if trainable:
        _TRAINABLE.append(v)
    return v


def constant(value, dtype=None, **_):
    return _arr(value, dtype)


def set_random_seed(_seed):
    return None


def global_variables_initializer():
    return None


def gradients(ys, xs):
    """d(sum ys)/d xs for a list of (variable) tensors, as numpy-convertible tensors; None where unconnected (as TF)."""
    ys = ys if isinstance(ys, (list, tuple)) else [ys]
    total = sum(_th.sum(y) for y in ys)
    return list(_th.autograd.grad(total, list(xs), allow_unused=True))


class Session:
    def __init__(self, *a, **k):
        pass

    def run(self, fetches, feed_dict=None):
        if isinstance(fetches, (list, tuple)):
            return type(fetches)(self.run(f) for f in fetches)
        if fetches is None:
            return None
        a = fetches.detach().numpy() if isinstance(fetches, _th.Tensor) else _np.array(fetches)
        return a[()] if a.ndim == 0 else _np.array(a)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def as_default(self):
        return self


class nn:
    @staticmethod
    def softplus(x):
        return _th.nn.functional.softplus(_arr(x), beta=1.0, threshold=1.0e6)     # no linear shortcut: exact log(1+e^x)

    @staticmethod
    def softmax(x, axis=-1):
        return _th.softmax(_arr(x), dim=axis)


def expand_dims(x, axis):
    return _th.unsqueeze(_arr(x), axis)


def squeeze(x, axis=None):
    x = _arr(x)
    return _th.squeeze(x) if axis is None else _th.squeeze(x, axis)


def sqrt(x):
    return _th.sqrt(_arr(x))


def square(x):
    return _th.square(_arr(x))


def exp(x):
    return _th.exp(_arr(x))


def log(x):
    return _th.log(_arr(x))


def reciprocal(x):
    return 1.0 / _arr(x)


def negative(x):
    return -_arr(x)


def multiply(a, b):
    return _arr(a) * _arr(b)


def squared_difference(a, b):
    return _th.square(_arr(a) - _arr(b))


def digamma(x):
    return _th.digamma(_arr(x))


def lgamma(x):
    return _th.lgamma(_arr(x))


def _axes(axis):
    return axis if axis is None or isinstance(axis, int) else tuple(int(a) for a in axis)


def reduce_sum(x, axis=None, keepdims=False):
    if isinstance(x, (list, tuple)):
        x = _th.stack([_arr(e) for e in x])
    x = _arr(x)
    return _th.sum(x) if axis is None else _th.sum(x, dim=_axes(axis), keepdim=keepdims)


def reduce_mean(x, axis=None, keepdims=False):
    x = _arr(x)
    return _th.mean(x) if axis is None else _th.mean(x, dim=_axes(axis), keepdim=keepdims)


def transpose(x, perm=None):
    x = _arr(x)
    if perm is None:
        perm = list(range(x.dim()))[::-1]
    return x.permute(*[int(p) for p in perm])


def _t(x):
    return _th.transpose(x, -1, -2)


def matmul(a, b, transpose_a=False, transpose_b=False):
    a, b = _arr(a), _arr(b)
    if transpose_a:
        a = _t(a)
    if transpose_b:
        b = _t(b)
    return _th.matmul(a, b)


def eye(num_rows, num_columns=None, batch_shape=None, dtype=float64):
    e = _th.eye(int(num_rows), int(num_rows) if num_columns is None else int(num_columns), dtype=dtype)
    if batch_shape is not None:
        e = e.expand(*[int(b) for b in batch_shape], *e.shape).clone()
    return e


def shape(x, out_type=int32):
    return _np.array(tuple(_arr(x).shape), dtype=_np.int64)


def _shape_tuple(shape):
    return tuple(int(s) for s in _np.atleast_1d(_np.asarray(shape)))


def ones(shape, dtype=float64):
    return _th.ones(_shape_tuple(shape), dtype=dtype)


def zeros(shape, dtype=float64):
    return _th.zeros(_shape_tuple(shape), dtype=dtype)


def ones_like(x):
    return _th.ones_like(_arr(x))


def zeros_like(x):
    return _th.zeros_like(_arr(x))


def matrix_diag(x):
    return _th.diag_embed(_arr(x))


def matrix_diag_part(x):
    return _th.diagonal(_arr(x), dim1=-2, dim2=-1)


def diag_part(x):
    return _th.diagonal(_arr(x))


def trace(x):
    return _th.sum(_th.diagonal(_arr(x), dim1=-2, dim2=-1), dim=-1)


def cholesky(x):
    return _th.linalg.cholesky(_arr(x))


def matrix_triangular_solve(matrix, rhs, lower=True, adjoint=False):
    matrix, rhs = _arr(matrix), _arr(rhs)
    if adjoint:
        return _th.linalg.solve_triangular(_t(matrix), rhs, upper=bool(lower))
    return _th.linalg.solve_triangular(matrix, rhs, upper=not lower)


def slice(x, begin, size):  # noqa: A001 - mirrors the TF name
    x = _arr(x)
    idx = tuple(_np.s_[int(b):(None if int(s) == -1 else int(b) + int(s))] for b, s in zip(begin, size))
    return x[idx]


def tile(x, multiples):
    return _arr(x).repeat(*[int(m) for m in multiples])


def cumsum(x, axis=0, exclusive=False, reverse=False):
    x = _arr(x)
    if reverse:
        x = _th.flip(x, dims=(axis,))
    c = _th.cumsum(x, dim=axis)
    if exclusive:
        c = c - x
    if reverse:
        c = _th.flip(c, dims=(axis,))
    return c


def one_hot(indices, depth, dtype=float64):
    return _th.eye(int(depth), dtype=dtype)[_th.as_tensor(_np.asarray(indices, dtype=_np.int64))]


def norm(x, axis=None):
    x = _arr(x)
    return _th.linalg.norm(x) if axis is None else _th.linalg.norm(x, dim=axis)


def argmin(x, axis=None):
    return _th.argmin(_arr(x), dim=axis)


def map_fn(fn, elems, dtype=None):
    return _th.stack([_arr(fn(e)) for e in elems])
