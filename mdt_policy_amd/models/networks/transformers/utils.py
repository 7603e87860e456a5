"""Parameter container of the resampler's feed-forward layer.

Reference: mdt/models/networks/transformers/utils.py:16-29 (``feed_forward_layer``): Sequential(LayerNorm, bias-less
Linear, activation, bias-less Linear) -- the Sequential indices (0, 1, 3) are part of the state_dict names.
"""
from __future__ import annotations

from torch import nn


def feed_forward_layer(dim: int, mult: int = 4, activation: str = "gelu") -> nn.Sequential:
    if activation != "gelu":
        raise NotImplementedError(f"activation={activation!r}: the HIP resampler implements 'gelu' (the reference default)")
    inner_dim = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner_dim, bias=False), nn.GELU(),
                         nn.Linear(inner_dim, dim, bias=False))
