"""KL(q(X) || p(X)) (reference: src/models/expressions/gp_expressions.py:10-24), evaluated by the HIP reduction kernel."""
import torch

from ... import ops


def calculate_kl_divergence_standard_prior(x_mean, x_covar):
    """x_mean [N,Q]; x_covar [N,Q,Q] (only its diagonal is used, gp_expressions.py:20) or the diagonal itself [N,Q]."""
    x_var = torch.diagonal(x_covar, dim1=-2, dim2=-1) if x_covar.dim() == 3 else x_covar
    return ops.kl_qx(x_mean, x_var)
