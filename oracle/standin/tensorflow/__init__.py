"""
TEST INFRASTRUCTURE, CONTAINER-ONLY.  Eager NumPy/SciPy module that answers to the name ``tensorflow``.

TensorFlow 1.15 (the reference's numeric backend, requirements.txt:29) is not installed in this image and has no
Python 3.10 wheel, so the reference's Python files cannot be imported as they are.  This module provides plain
NumPy/SciPy definitions of the ~45 ``tf.*`` names the reference's ``src/`` tree and ``test/unittests`` use, so that the
reference's *own, unmodified* source can be executed (every "graph node" is simply an ndarray computed on the spot).
It is used by ``oracle/gen_golden.py`` only: to pin ``oracle/`` against the reference and to write ``tests/golden/*.npz``.

It contains no DP-GP-LVM logic: each function is the textbook meaning of the TensorFlow op of the same name.
Nothing in ``dp_gp_lvm_amd/``, ``bench.py`` or the GPU tests imports it, and without ``/root/reference`` it is inert.
Linear algebra is LAPACK (NumPy/SciPy) instead of Eigen, so fp64 results can differ from real TF in the last digits.
"""

import numpy as _np
import scipy.linalg as _sla
import scipy.special as _sp

float64 = _np.float64
float32 = _np.float32
int32 = _np.int32
int64 = _np.int64

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


class _Var(_np.ndarray):
    """ndarray subclass so variables are hashable by identity (the reference puts them in sets)."""
    def __hash__(self):
        return id(self)


def Variable(initial_value=None, dtype=None, trainable=True, **_):
    v = _np.array(initial_value, dtype=dtype).view(_Var)
    _GLOBAL.append(v)
    if trainable:
        _TRAINABLE.append(v)
    return v


def constant(value, dtype=None, **_):
    return _np.array(value, dtype=dtype)


def set_random_seed(_seed):
    return None


def global_variables_initializer():
    return None


class Session:
    def __init__(self, *a, **k):
        pass

    def run(self, fetches, feed_dict=None):
        # real TF hands back plain ndarrays / NumPy scalars, never Variable objects
        if isinstance(fetches, (list, tuple)):
            return type(fetches)(self.run(f) for f in fetches)
        if fetches is None:
            return None
        a = _np.array(fetches)
        return a[()] if a.ndim == 0 else a

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def as_default(self):
        return self


def _arr(x):
    return _np.asarray(x)


class nn:
    @staticmethod
    def softplus(x):
        return _np.logaddexp(0.0, _arr(x))

    @staticmethod
    def softmax(x, axis=-1):
        x = _arr(x)
        e = _np.exp(x - _np.max(x, axis=axis, keepdims=True))
        return e / _np.sum(e, axis=axis, keepdims=True)


def expand_dims(x, axis):
    return _np.expand_dims(_arr(x), axis)


def squeeze(x, axis=None):
    return _np.squeeze(_arr(x), axis=axis)


def sqrt(x):
    return _np.sqrt(_arr(x))


def square(x):
    return _np.square(_arr(x))


def exp(x):
    return _np.exp(_arr(x))


def log(x):
    return _np.log(_arr(x))


def reciprocal(x):
    return 1.0 / _arr(x)


def negative(x):
    return -_arr(x)


def multiply(a, b):
    return _arr(a) * _arr(b)


def squared_difference(a, b):
    return _np.square(_arr(a) - _arr(b))


def digamma(x):
    return _sp.digamma(_arr(x))


def lgamma(x):
    return _sp.gammaln(_arr(x))


def reduce_sum(x, axis=None, keepdims=False):
    if isinstance(x, (list, tuple)):
        x = _np.stack([_arr(e) for e in x])
    return _np.sum(_arr(x), axis=axis, keepdims=keepdims)


def reduce_mean(x, axis=None, keepdims=False):
    return _np.mean(_arr(x), axis=axis, keepdims=keepdims)


def transpose(x, perm=None):
    return _np.transpose(_arr(x), perm)


def _t(x):
    return _np.swapaxes(x, -1, -2)


def matmul(a, b, transpose_a=False, transpose_b=False):
    a, b = _arr(a), _arr(b)
    if transpose_a:
        a = _t(a)
    if transpose_b:
        b = _t(b)
    return _np.matmul(a, b)


def eye(num_rows, num_columns=None, batch_shape=None, dtype=float64):
    e = _np.eye(int(num_rows), None if num_columns is None else int(num_columns), dtype=dtype)
    if batch_shape is not None:
        e = _np.broadcast_to(e, tuple(int(b) for b in batch_shape) + e.shape).copy()
    return e


def shape(x, out_type=int32):
    return _np.array(_np.shape(x), dtype=out_type)


def ones(shape, dtype=float64):
    return _np.ones(tuple(int(s) for s in _np.atleast_1d(shape)), dtype=dtype)


def zeros(shape, dtype=float64):
    return _np.zeros(tuple(int(s) for s in _np.atleast_1d(shape)), dtype=dtype)


def ones_like(x):
    return _np.ones_like(_arr(x))


def zeros_like(x):
    return _np.zeros_like(_arr(x))


def matrix_diag(x):
    x = _arr(x)
    out = _np.zeros(x.shape + (x.shape[-1],), dtype=x.dtype)
    idx = _np.arange(x.shape[-1])
    out[..., idx, idx] = x
    return out


def matrix_diag_part(x):
    return _np.diagonal(_arr(x), axis1=-2, axis2=-1).copy()


def diag_part(x):
    return _np.diagonal(_arr(x)).copy()


def trace(x):
    return _np.trace(_arr(x), axis1=-2, axis2=-1)


def cholesky(x):
    return _np.linalg.cholesky(_arr(x))


def matrix_triangular_solve(matrix, rhs, lower=True, adjoint=False):
    matrix, rhs = _arr(matrix), _arr(rhs)
    batch = _np.broadcast_shapes(matrix.shape[:-2], rhs.shape[:-2])
    mb = _np.broadcast_to(matrix, batch + matrix.shape[-2:])
    rb = _np.broadcast_to(rhs, batch + rhs.shape[-2:])
    out = _np.empty(batch + rhs.shape[-2:], dtype=_np.result_type(matrix, rhs))
    for i in _np.ndindex(*batch):
        out[i] = _sla.solve_triangular(mb[i], rb[i], lower=lower, trans='T' if adjoint else 'N')
    return out


def slice(x, begin, size):  # noqa: A001 - mirrors the TF name
    x = _arr(x)
    idx = tuple(_np.s_[int(b):(None if int(s) == -1 else int(b) + int(s))] for b, s in zip(begin, size))
    return x[idx]


def tile(x, multiples):
    return _np.tile(_arr(x), tuple(int(m) for m in multiples))


def cumsum(x, axis=0, exclusive=False, reverse=False):
    x = _arr(x)
    if reverse:
        x = _np.flip(x, axis=axis)
    c = _np.cumsum(x, axis=axis)
    if exclusive:
        c = c - x
    if reverse:
        c = _np.flip(c, axis=axis)
    return c


def one_hot(indices, depth, dtype=float64):
    return _np.eye(int(depth), dtype=dtype)[_np.asarray(indices, dtype=int)]


def norm(x, axis=None):
    return _np.linalg.norm(_arr(x), axis=axis)


def argmin(x, axis=None):
    return _np.argmin(_arr(x), axis=axis)


def map_fn(fn, elems, dtype=None):
    return _np.stack([fn(e) for e in elems])
