"""
Truncated stick-breaking Dirichlet process, variational objective — mirror of the reference's
src/models/dirichlet_process.py:17-136 on torch tensors (fp64).  It is O(D T) scalar arithmetic: standalone it is
evaluated with a few torch ops on the GPU; inside dp_gp_lvm(...).objective the same quantity comes out of the HIP
``dpgp_model_prepare`` kernel (tests compare the two).
"""
import numpy as np
import torch
import torch.nn.functional as F

from ..distributions.beta import entropy as beta_dist_entropy
from ..distributions.gamma import entropy as gamma_dist_entropy
from ..distributions.multinomial import entropy as multinomial_dist_entropy
from ..utils.constants import DP_DEFAULT_ALPHA_PRIOR_PARAMS, DP_DEFAULT_TRUNCATION_LEVEL
from ..utils.types import TORCH_DTYPE, create_positive_variable, create_random_positive_variable, default_device
from .interfaces.trainable import Trainable


def dirichlet_process(num_samples, alpha_prior_params=DP_DEFAULT_ALPHA_PRIOR_PARAMS,
                      truncation_level=DP_DEFAULT_TRUNCATION_LEVEL, mask_size=1, device=None):
    device = default_device() if device is None else device
    s_1, s_2 = float(alpha_prior_params[0]), float(alpha_prior_params[1])
    # q(Z): phi = softmax(logits) [N x T]; with a mask, groups of mask_size adjacent samples share a row (:39-51)
    if mask_size == 1:
        mask_depth = num_samples
    else:
        assert num_samples % mask_size == 0, 'mask_size must divide the number of samples.'
        mask_depth = num_samples // mask_size
    logits = torch.as_tensor(np.random.standard_normal((mask_depth, truncation_level)), dtype=TORCH_DTYPE,
                             device=device)
    # q(V): Beta(gamma_1, gamma_2), T-1 sticks; q(alpha): Gamma(w_1, w_2) initialised at the prior (:54-59)
    gamma_1_raw = create_random_positive_variable(truncation_level - 1, device=device)
    gamma_2_raw = create_random_positive_variable(truncation_level - 1, device=device)
    w_raw = torch.stack([create_positive_variable(s_1, device=device), create_positive_variable(s_2, device=device)])

    class DirichletProcess(Trainable):
        # raw (unconstrained) variables, for optimisers and for the fused HIP objective
        raw = dict(logits=logits, gamma_1=gamma_1_raw, gamma_2=gamma_2_raw, w=w_raw)
        prior = (s_1, s_2)
        mask = mask_size

        @property
        def assignments(self):
            phi = torch.softmax(logits, dim=-1)
            return phi if mask_size == 1 else torch.repeat_interleave(phi, mask_size, dim=0)

        @property
        def q_z(self):
            return self.assignments

        @property
        def q_v(self):
            return F.softplus(gamma_1_raw), F.softplus(gamma_2_raw)

        @property
        def q_alpha(self):
            return F.softplus(w_raw[0]), F.softplus(w_raw[1])

        @property
        def objective(self):
            """-ELBO of the DP (:64-88)."""
            phi = self.assignments
            gamma_1, gamma_2 = self.q_v
            w_1, w_2 = self.q_alpha
            t = truncation_level
            dg12 = torch.digamma(gamma_1 + gamma_2)
            tail = (torch.flip(torch.cumsum(torch.flip(phi, [-1]), dim=-1), [-1]) - phi)[:, 0:-1]
            ev_z = torch.sum(phi[:, 0:-1] * (torch.digamma(gamma_1) - dg12) + tail * (torch.digamma(gamma_2) - dg12))
            ev_v = (t - 1.0) * (torch.digamma(w_1) - torch.log(w_2)) + \
                ((w_1 / w_2) - 1.0) * torch.sum(torch.digamma(gamma_2) - dg12)
            ev_a = s_1 * np.log(s_2) - float(torch.lgamma(torch.tensor(s_1, dtype=TORCH_DTYPE))) + \
                (s_1 - 1.0) * (torch.digamma(w_1) - torch.log(w_2)) - s_2 * (w_1 / w_2)
            elbo = ev_z + ev_v + ev_a + torch.sum(multinomial_dist_entropy(phi)) + \
                torch.sum(beta_dist_entropy(gamma_1, gamma_2)) + gamma_dist_entropy(w_1, w_2)
            return -elbo

    return DirichletProcess()
