"""Log-normal log-density (reference: src/distributions/log_normal.py:24-39), torch one-liner on the host side of the API."""
import math

import torch


def log_pdf(x, mean=None, var=None):
    if mean is None:
        mean = torch.zeros_like(x)
    if var is None:
        var = torch.ones_like(x)
    return -torch.log(x) - 0.5 * (torch.log(2.0 * math.pi * var) + (torch.log(x) - mean) ** 2 / var)
