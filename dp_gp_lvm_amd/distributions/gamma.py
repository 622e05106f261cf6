"""Gamma entropy, shape alpha / rate beta (reference: src/distributions/gamma.py:8-17)."""
import torch


def entropy(alpha, beta):
    return alpha - torch.log(beta) + torch.lgamma(alpha) + (1.0 - alpha) * torch.digamma(alpha)
