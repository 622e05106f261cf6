"""Multivariate normal scoring of predictions (reference: src/distributions/normal.py:14-36), on the library's batched
Cholesky / triangular-solve operators (dpgp_potrf_batched, dpgp_trsm_batched) instead of tf.cholesky /
tf.matrix_triangular_solve.  Used as in test/frey_faces_prediction.py:232-238: log-likelihood of the held-out ground truth
under the predictive mean / covariance that predict_missing_data returns."""
import math

import torch

from .. import ops
from ..utils.types import TORCH_DTYPE


def _chol(covariance):
    c = torch.as_tensor(covariance, dtype=TORCH_DTYPE)
    while c.dim() > 2 and c.shape[0] == 1:
        c = c[0]                                                  # tf.squeeze
    assert c.dim() == 2 and c.shape[0] == c.shape[1], 'covariance must be [D x D]'
    l_, info = ops.potrf_batched(c[None].contiguous())
    return l_, info


def mvn_log_pdf(x, mean, covariance):
    """Log-likelihood of the rows of x [B x D] under N(mean [1 x D], covariance [D x D]); returns a B-vector.
    A covariance that is not positive definite gives NaN (tf.cholesky raises there)."""
    x = torch.as_tensor(x, dtype=TORCH_DTYPE, device=covariance.device if torch.is_tensor(covariance) else None)
    mean = torch.as_tensor(mean, dtype=TORCH_DTYPE, device=x.device)
    l_, info = _chol(covariance)
    num_dims = l_.shape[-1]
    diff = (x - mean).transpose(0, 1).contiguous()                # [D x B]
    alpha = ops.trsm_batched(l_, diff[None].contiguous())[0].transpose(0, 1)          # [B x D]
    beta = torch.sum(torch.log(torch.diagonal(l_[0])))
    out = -0.5 * (torch.sum(alpha * alpha, dim=-1) + num_dims * math.log(2.0 * math.pi)) - beta
    return torch.where(info[0] == 0, out, torch.full_like(out, float('nan')))
