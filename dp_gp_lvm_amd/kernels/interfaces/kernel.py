"""
Kernel operator interface — same names, keywords and error behaviour as the reference's
src/kernels/interfaces/kernel.py:12-280, on torch (ROCm) tensors instead of TensorFlow graph nodes.
"""
from abc import ABC, abstractmethod
from enum import Enum

import torch


class KernelHyperparameters(Enum):
    ARD_WEIGHTS = 'gamma'
    SIGNAL_VARIANCE = 'alpha'
    NOISE_PRECISION = 'beta'
    FREQUENCY = 'freq'
    PERIOD = 'period'
    LENGTH_SCALES = 'l'
    LINEAR_WEIGHTS = 'W'


class AbstractKernel(ABC):
    @property
    @abstractmethod
    def prior_log_likelihood(self):
        pass

    @property
    @abstractmethod
    def noise_precision(self):
        pass

    @property
    @abstractmethod
    def hyperparameters(self):
        pass

    @abstractmethod
    def covariance_matrix(self, input_0, input_1=None, include_noise=False, include_jitter=False):
        pass

    @abstractmethod
    def covariance_diag(self, input_0, include_noise=False, include_jitter=False):
        pass

    @abstractmethod
    def psi_0(self, inducing_input, latent_input_mean, latent_input_covariance):
        pass

    @abstractmethod
    def psi_1(self, inducing_input, latent_input_mean, latent_input_covariance):
        pass

    @abstractmethod
    def psi_2(self, inducing_input, latent_input_mean, latent_input_covariance):
        pass


class Kernel(AbstractKernel):
    """Generic kernel object around closures (kernel.py:129-280)."""

    def __init__(self, covar_matrix_func, covar_diag_func, hyperparameter_dict, hyperprior_func_dict,
                 psi_0_func=None, psi_1_func=None, psi_2_func=None):
        assert callable(covar_matrix_func), 'Covariance matrix function must be callable.'
        assert callable(covar_diag_func), 'Covariance diagonal function must be callable.'
        assert isinstance(hyperparameter_dict, dict)
        assert isinstance(hyperprior_func_dict, dict)
        assert all([isinstance(hp, KernelHyperparameters) for hp in hyperparameter_dict.keys()]), \
            'All dictionary keys must be of type KernelHyperparameters enumeration.'
        assert set(hyperparameter_dict.keys()) == set(hyperprior_func_dict.keys()), \
            'Both dictionaries must have the same keys.'
        assert all([callable(prior_func) for prior_func in hyperprior_func_dict.values()]), \
            'All hyperprior functions must be callable.'
        self._covar_matrix_func = covar_matrix_func
        self._covar_diag_func = covar_diag_func
        self._hyperparameter_dict = hyperparameter_dict
        self._hyperprior_dict = hyperprior_func_dict
        self._psi_0_func = psi_0_func
        self._psi_1_func = psi_1_func
        self._psi_2_func = psi_2_func
        # sum of the hyper-prior log-likelihoods of the kernel's own hyperparameters (kernel.py:177-178)
        self.hyperprior_log_likelihood = sum(torch.sum(hyperprior_func_dict[hp](hyperparameter_dict[hp]))
                                             for hp in hyperparameter_dict.keys())

    @property
    def prior_log_likelihood(self):
        return self.hyperprior_log_likelihood

    @property
    def noise_precision(self):
        return self._hyperparameter_dict[KernelHyperparameters.NOISE_PRECISION]

    @property
    def hyperparameters(self):
        return self._hyperparameter_dict

    def covariance_matrix(self, input_0, input_1=None, include_noise=False, include_jitter=False):
        return self._covar_matrix_func(input_0=input_0, input_1=input_1, include_noise=include_noise,
                                       include_jitter=include_jitter)

    def covariance_diag(self, input_0, include_noise=False, include_jitter=False):
        return self._covar_diag_func(input_0=input_0, include_noise=include_noise, include_jitter=include_jitter)

    def psi_0(self, inducing_input, latent_input_mean, latent_input_covariance):
        if self._psi_0_func is not None:
            return self._psi_0_func(inducing_input=inducing_input, latent_input_mean=latent_input_mean,
                                    latent_input_covariance=latent_input_covariance)
        raise NotImplementedError

    def psi_1(self, inducing_input, latent_input_mean, latent_input_covariance):
        if self._psi_1_func is not None:
            return self._psi_1_func(inducing_input=inducing_input, latent_input_mean=latent_input_mean,
                                    latent_input_covariance=latent_input_covariance)
        raise NotImplementedError

    def psi_2(self, inducing_input, latent_input_mean, latent_input_covariance):
        if self._psi_2_func is not None:
            return self._psi_2_func(inducing_input=inducing_input, latent_input_mean=latent_input_mean,
                                    latent_input_covariance=latent_input_covariance)
        raise NotImplementedError
