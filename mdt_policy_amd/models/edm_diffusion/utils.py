"""Helpers of the EDM wrappers (reference: mdt/models/edm_diffusion/utils.py:146-203)."""
from __future__ import annotations

import math

import torch


def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    """Right-pad ``x`` with singleton dims up to ``target_dims`` (reference utils.py:146-151)."""
    missing = target_dims - x.ndim
    if missing < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * missing]


def rand_log_normal(shape, loc=0., scale=1., device="cpu", dtype=torch.float32):
    """Log-normal noise levels (reference utils.py:154-156)."""
    return (torch.randn(shape, device=device, dtype=dtype) * scale + loc).exp()


def rand_log_logistic(shape, loc=0., scale=1., min_value=0., max_value=float("inf"), device="cpu",
                      dtype=torch.float32):
    """Optionally truncated log-logistic noise levels -- the training sigma density (reference utils.py:159-166):
    inverse-CDF sampling in float64 between the CDF values of the truncation bounds."""
    lo = torch.as_tensor(min_value, device=device, dtype=torch.float64)
    hi = torch.as_tensor(max_value, device=device, dtype=torch.float64)
    cdf_lo = ((lo.log() - loc) / scale).sigmoid()
    cdf_hi = ((hi.log() - loc) / scale).sigmoid()
    u = torch.rand(shape, device=device, dtype=torch.float64) * (cdf_hi - cdf_lo) + cdf_lo
    return (u.logit() * scale + loc).exp().to(dtype)


def rand_log_uniform(shape, min_value, max_value, device="cpu", dtype=torch.float32):
    """Log-uniform noise levels (reference utils.py:169-173)."""
    lo, hi = math.log(min_value), math.log(max_value)
    return (torch.rand(shape, device=device, dtype=dtype) * (hi - lo) + lo).exp()


def rand_uniform(shape, min_value, max_value, device="cpu", dtype=torch.float32):
    return torch.rand(shape, device=device, dtype=dtype) * (max_value - min_value) + min_value
