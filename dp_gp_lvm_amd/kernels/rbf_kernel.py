"""
ARD-RBF kernel object — the reference's k_ard_rbf (src/kernels/rbf_kernel.py:26-203) with its closures dispatching to
the HIP kernels of libdpgp_hip.so instead of TensorFlow ops.  Same argument meaning and return shapes:
  gamma [B x Q], alpha [B x 1], beta [B x 1]  ->  Kernel with
  covariance_matrix -> [B x N0 x N1], covariance_diag -> [B x N], psi_0 -> [B x 1], psi_1 -> [B x N x M], psi_2 -> [B x M x M].
The arithmetic type is the dtype of the tensors passed in (float32 or float64).
"""
import torch

from .. import ops
from ..distributions.log_normal import log_pdf as log_normal_log_pdf
from ..utils.constants import GP_DEFAULT_JITTER
from .interfaces.kernel import Kernel, KernelHyperparameters


def k_rbf(gamma, alpha, beta):
    raise NotImplementedError          # as in the reference (rbf_kernel.py:13-23)


def _variances(latent_input_covariance):
    """The reference API takes q(X)'s covariance as [N x Q x Q] and reads only its diagonal (rbf_kernel.py:151,182);
    the [N x Q] diagonal itself is accepted too."""
    c = latent_input_covariance
    return torch.diagonal(c, dim1=-2, dim2=-1) if c.dim() == 3 else c


def k_ard_rbf(gamma, alpha, beta):
    hyperparameters_dict = {KernelHyperparameters.ARD_WEIGHTS: gamma,
                            KernelHyperparameters.SIGNAL_VARIANCE: alpha,
                            KernelHyperparameters.NOISE_PRECISION: beta}
    hyperpriors_dict = {KernelHyperparameters.ARD_WEIGHTS: log_normal_log_pdf,
                        KernelHyperparameters.SIGNAL_VARIANCE: log_normal_log_pdf,
                        KernelHyperparameters.NOISE_PRECISION: log_normal_log_pdf}

    def covariance_matrix_func(input_0, input_1=None, include_noise=False, include_jitter=False):
        # noise / jitter only when input_1 is None, even if the same array is passed twice (rbf_kernel.py:80,86)
        return ops.ard_rbf_gram(input_0, input_1, gamma, alpha, beta, include_noise=include_noise,
                                include_jitter=include_jitter, jitter=GP_DEFAULT_JITTER)

    def covariance_diagonal_func(input_0, include_noise=False, include_jitter=False):
        return ops.ard_rbf_diag(input_0.shape[0], alpha, beta, include_noise=include_noise,
                                include_jitter=include_jitter, jitter=GP_DEFAULT_JITTER)

    def calculate_psi_0(inducing_input, latent_input_mean, latent_input_covariance):
        return ops.psi0(latent_input_mean.shape[0], alpha)

    def calculate_psi_1(inducing_input, latent_input_mean, latent_input_covariance):
        return ops.psi1(inducing_input, latent_input_mean, _variances(latent_input_covariance), gamma, alpha)

    def calculate_psi_2(inducing_input, latent_input_mean, latent_input_covariance):
        return ops.psi2(inducing_input, latent_input_mean, _variances(latent_input_covariance), gamma, alpha)

    return Kernel(covar_matrix_func=covariance_matrix_func, covar_diag_func=covariance_diagonal_func,
                  hyperparameter_dict=hyperparameters_dict, hyperprior_func_dict=hyperpriors_dict,
                  psi_0_func=calculate_psi_0, psi_1_func=calculate_psi_1, psi_2_func=calculate_psi_2)


def k_mahalanobis_rbf(weights, gamma, alpha, beta):
    raise NotImplementedError          # as in the reference (rbf_kernel.py:206-218)
