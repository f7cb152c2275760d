"""Batch sharding across the GPUs of a node (one process per GPU, ``torch.distributed`` over RCCL).

Every op of the path is independent per image: the batch is cut into contiguous,
balanced slices, each rank runs the native ops on its slice with no data-path collective, and only an
optional ``all_gather`` (RCCL over xGMI when the tensors are on HIP devices) reassembles outputs for a
single-device consumer.  Per-sample matrices / sigmas shard with the batch; shared ones (leading
dimension 1) are replicated.  The same code runs on CPU tensors with the ``gloo`` backend (tests).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist

__all__ = ["gather_batch", "shard_batch", "shard_bounds", "sharded_apply"]


def shard_bounds(n: int, world_size: int, rank: int) -> tuple[int, int]:
    """[lo, hi) of rank's contiguous slice of n items; the first (n % world_size) ranks get one more."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} out of range for world size {world_size}")
    q, r = divmod(n, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_batch(tensors: Sequence[Optional[torch.Tensor]], batch: int, world_size: int, rank: int):
    """Slice every tensor whose leading dimension equals ``batch``; pass shared ones (leading dim 1 or
    non-tensors) through unchanged."""
    lo, hi = shard_bounds(batch, world_size, rank)
    out = []
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.dim() > 0 and t.shape[0] == batch and batch != 1:
            out.append(t[lo:hi])
        else:
            out.append(t)
    return out


def gather_batch(local: torch.Tensor, batch: int, group=None) -> torch.Tensor:
    """all_gather the per-rank slices back into a (batch, ...) tensor on every rank (uneven slices are
    padded to the largest one for the collective)."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_bounds(batch, world, r) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    n_local = local.shape[0]
    if n_local < max_n:
        pad = torch.zeros(max_n - n_local, *local.shape[1:], dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    gathered = torch.empty(world * max_n, *local.shape[1:], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)
    if all(hi - lo == max_n for lo, hi in sizes):
        return gathered
    return torch.cat([gathered[r * max_n : r * max_n + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def sharded_apply(op: Callable[..., torch.Tensor], *tensors, gather: bool = True, group=None) -> torch.Tensor:
    """Run ``op`` on this rank's slice of the batch (first tensor's leading dim) and optionally gather.

    ``op`` is any per-sample-independent callable, e.g.
    ``lambda x, M: gaussian_blur2d(warp_perspective(x, M, (512, 512)), (5, 5), (1.5, 1.5))``.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return op(*tensors)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    batch = tensors[0].shape[0]
    local = op(*shard_batch(tensors, batch, world, rank))
    return gather_batch(local, batch, group) if gather else local
