"""Host-side initialisation helpers (reference: src/utils/expressions.py:47-76). NumPy only; not on the hot path."""
import numpy as np
from scipy.sparse.linalg import eigsh


def principal_component_analysis(x, num_latent_dimensions):
    """
    Leading eigenvectors of X X^T rescaled so that the mean column standard deviation is 1 — what the reference's
    principal_component_analysis returns (expressions.py:47-76).  The reference calls ARPACK with a random start vector,
    so its eigenvector SIGNS change from run to run; here the start vector is fixed and every column is
    oriented so that its largest-magnitude entry is positive, which makes the initialisation reproducible.
    """
    assert isinstance(x, np.ndarray)
    assert x.ndim == 2
    n, d = x.shape
    assert 0 < num_latent_dimensions < min(n, d), \
        'Number of latent dimensions must be greater than zero and less than the minimum of the number of ' \
        'observations and the number of observed dimensions.'
    g = np.dot(x, x.T)
    if n < (num_latent_dimensions + 2):
        w, v = np.linalg.eigh(g)
        v = v[:, ::-1][:, :num_latent_dimensions]
    else:
        w, v = eigsh(g, k=num_latent_dimensions, which='LA', v0=np.ones(n))
        v = v[:, np.argsort(-w)]
    v = v * np.sign(v[np.argmax(np.abs(v), axis=0), np.arange(v.shape[1])])
    return v / np.mean(v.std(axis=0, ddof=1))
