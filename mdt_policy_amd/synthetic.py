"""Deterministic synthetic weights and inputs, keyed by parameter name.

The golden fixtures under tests/golden/ hold only the reference's *outputs*; the weights and inputs
that produced them are regenerated anywhere (this container, the GPU box) from this counter-based
generator, so nothing of the reference has to travel.  numpy Philox is a counter-based bit generator
whose stream is fixed by (seed, crc32(name)).

Two weight profiles:
  * ``"init"``  - the distributions of the reference's ``_init_weights``
                  (reference: mdt/models/networks/mdtv_transformer.py:197-206): Linear W ~ N(0, 0.02),
                  biases 0, LayerNorm weight 1 / bias 0, pos_emb ~ N(0, 0.02).  Used by bench.py.
  * ``"rich"``  - trained-like magnitudes (W ~ N(0, 1/fan_in), non-zero biases, non-unit LN gains) so
                  that every bias/gain/modulation term is exercised by the parity tests.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def normal(name: str, shape: Tuple[int, ...], seed: int = 0, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """float32 N(mean, std) tensor fully determined by (seed, name, shape)."""
    x = _rng(seed, name).standard_normal(size=tuple(shape), dtype=np.float64)
    return (x * std + mean).astype(np.float32)


def uniform(name: str, shape: Tuple[int, ...], seed: int = 0, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    x = _rng(seed, name).random(size=tuple(shape), dtype=np.float64)
    return (x * (hi - lo) + lo).astype(np.float32)


def param_tensor(name: str, shape: Tuple[int, ...], seed: int = 0, profile: str = "rich") -> np.ndarray:
    """Value for one state_dict entry, chosen by its name/shape."""
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit(".", 1)[-1]
    if name.endswith("rotary_pos_emb.freqs"):
        raise ValueError("rotary freqs are a deterministic buffer, not synthetic")
    if leaf in ("latents", "time_pos_emb"):  # Perceiver resampler: torch.randn initialised (perceiver_resampler.py:105-106)
        return normal(name, shape, seed, std=1.0)
    if len(shape) == 4 and leaf == "weight":  # Conv2d patch embedding (out, c, ph, pw): a Linear over c*ph*pw inputs
        fan_in = int(np.prod(shape[1:]))
        return normal(name, shape, seed, std=0.02 if profile == "init" else 1.0 / np.sqrt(fan_in))
    if len(shape) == 4:  # ctx_dec_pe (1, 2, 1, d)
        return normal(name, shape, seed, std=0.02 if profile == "init" else 0.1)
    if len(shape) == 1 and leaf == "gamma":  # LayerScale
        if profile == "init":
            return np.full(shape, 0.1, np.float32)
        return normal(name, shape, seed, std=0.05, mean=0.3)
    if len(shape) == 3:  # pos_emb (1, T, d)
        return normal(name, shape, seed, std=0.02 if profile == "init" else 0.1)
    if len(shape) == 2:  # Linear weight (out, in)
        if profile == "init":
            return normal(name, shape, seed, std=0.02)
        return normal(name, shape, seed, std=1.0 / np.sqrt(shape[1]))
    if len(shape) == 1 and leaf == "bias":
        if profile == "init":
            return np.zeros(shape, np.float32)
        return normal(name, shape, seed, std=0.1)
    if len(shape) == 1 and leaf in ("weight", "g"):  # LayerNorm / RMSNorm gain
        if profile == "init":
            return np.ones(shape, np.float32)
        return normal(name, shape, seed, std=0.1, mean=1.0)
    raise ValueError(f"don't know how to synthesise {name} {shape}")


def fill_state_dict(named_shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0,
                    profile: str = "rich") -> Dict[str, np.ndarray]:
    """Synthesise every (name, shape) except rotary frequency buffers."""
    out = {}
    for name, shape in named_shapes:
        if name.endswith("rotary_pos_emb.freqs"):
            continue
        out[name] = param_tensor(name, shape, seed, profile)
    return out


def sampler_inputs(batch: int, cfg: dict, seed: int = 1, arch: str = "mdtv") -> Dict[str, np.ndarray]:
    """Synthetic tokens of SURVEY.md 8(d): state ~ N(0,1), goal ~ N(0,1), noise ~ N(0,1)."""
    d_obs = int(cfg["obs_dim"])
    out = {
        "goal": normal("goal", (batch, 1, int(cfg["goal_dim"])), seed),
        "noise": normal("noise", (batch, int(cfg["action_seq_len"]), int(cfg["action_dim"])), seed),
    }
    if arch == "mdtv":
        out["state_images"] = normal("state_images", (batch, int(cfg["n_obs_token"]), d_obs), seed)
    else:
        out["static"] = normal("static", (batch, 1, d_obs), seed)
        out["gripper"] = normal("gripper", (batch, 1, d_obs), seed)
    return out


def loss_inputs(batch: int, cfg: dict, seed: int = 3, sigma_data: float = 0.5,
                sigma_min: float = 0.001, sigma_max: float = 80.0) -> Dict[str, np.ndarray]:
    """Training-shaped inputs: actions ~ U(-1,1), sigma ~ truncated log-logistic (reference:
    mdt/models/edm_diffusion/utils.py:159-166 with loc=ln(sigma_data), scale=0.5), noise ~ N(0,1)."""
    ta, a = int(cfg["action_seq_len"]), int(cfg["action_dim"])
    u = _rng(seed, "sigma_u").random(size=(batch,), dtype=np.float64)
    loc, scale = np.log(sigma_data), 0.5
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    lo, hi = sig((np.log(sigma_min) - loc) / scale), sig((np.log(sigma_max) - loc) / scale)
    u = u * (hi - lo) + lo
    sigma = np.exp(np.log(u / (1.0 - u)) * scale + loc).astype(np.float32)
    return {
        "actions": uniform("actions", (batch, ta, a), seed, -1.0, 1.0),
        "noise_train": normal("noise_train", (batch, ta, a), seed),
        "sigma": sigma,
    }
