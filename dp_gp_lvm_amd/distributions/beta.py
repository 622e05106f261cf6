"""Beta entropy (reference: src/distributions/beta.py:8-19)."""
import torch


def entropy(alpha, beta):
    t = alpha + beta
    return torch.lgamma(alpha) + torch.lgamma(beta) - torch.lgamma(t) - (alpha - 1.0) * torch.digamma(alpha) - \
        (beta - 1.0) * torch.digamma(beta) + (t - 2.0) * torch.digamma(t)
