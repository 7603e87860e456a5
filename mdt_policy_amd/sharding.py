"""Batch-sharded sampling over the GPUs of one node (BASELINE config C4).

Every action chunk is independent, so the request shards by batch with NO exchange inside the sampler loop:
weights are replicated (90 MB fp32), rank r runs the fused encoder + DDIM loop on its contiguous slice, and the
sampled actions (B_local x 10 x 7 fp32 = 71.7 KB at B_local = 256) are combined by ONE all-gather -- RCCL over
xGMI on MI355X (``backend="nccl"`` is RCCL on ROCm), gloo in the CPU tests.  One process per GPU.

The reference never shards a sampling batch (rollouts are B = 1 per rank, mdt/rollout/rollout_long_horizon.py:42-78);
this is the multi-GPU form of ``MDTVAgent.denoise_actions`` (mdt/models/mdtv_agent.py:523-550).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) of ``total`` items for ``rank`` (first total % world ranks get one more)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _slice_state(state: dict, lo: int, hi: int) -> dict:
    return {k: (v[lo:hi] if torch.is_tensor(v) else v) for k, v in state.items()}


def all_gather_actions(local: torch.Tensor, total: Optional[int] = None, group=None) -> torch.Tensor:
    """One collective: concatenate every rank's (B_r, Ta, A) block along the batch, in rank order.
    Equal shards use all_gather_into_tensor (a single flat collective); ragged shards are padded to the
    largest shard and trimmed."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    rank = dist.get_rank(group)
    if total is None:
        sizes = [None] * world
        dist.all_gather_object(sizes, int(local.shape[0]), group=group)
        total = sum(sizes)
    bounds = [shard_bounds(total, r, world) for r in range(world)]
    sizes = [hi - lo for lo, hi in bounds]
    if sizes[rank] != local.shape[0]:
        raise ValueError(f"rank {rank}: local batch {local.shape[0]} does not match its shard size {sizes[rank]}")
    local = local.contiguous()
    if len(set(sizes)) == 1:
        out = local.new_empty((total,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    mx = max(sizes)
    padded = local.new_zeros((mx,) + tuple(local.shape[1:]))
    padded[: local.shape[0]] = local
    out = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def sample_sharded(sample_fn: Callable, state: dict, x_T: torch.Tensor, goal: torch.Tensor, sigmas, group=None,
                   gather: bool = True) -> torch.Tensor:
    """Run ``sample_fn(state, x_T, goal, sigmas)`` on this rank's contiguous slice of a replicated request and
    all-gather the sampled actions.  ``sample_fn`` is e.g. ``lambda s, x, g, sig: sample_ddim(model, s, x, g, sig)``."""
    if not dist.is_available() or not dist.is_initialized():
        return sample_fn(state, x_T, goal, sigmas)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    total = x_T.shape[0]
    lo, hi = shard_bounds(total, rank, world)
    local = sample_fn(_slice_state(state, lo, hi), x_T[lo:hi], goal[lo:hi], sigmas)
    return all_gather_actions(local, total, group) if gather else local
