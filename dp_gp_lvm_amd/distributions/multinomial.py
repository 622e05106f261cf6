"""Multinomial entropy (reference: src/distributions/multinomial.py:8-16)."""
import torch


def entropy(probs):
    return -torch.sum(probs * torch.log(probs), dim=-1)
