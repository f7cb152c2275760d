"""Batch sharding across the GPUs of a node (one process per GPU, ``torch.distributed`` over RCCL).

Every op of the path is independent per image: the batch is cut into contiguous,
balanced slices, each rank runs the native ops on its slice with no data-path collective, and only an
optional exchange (RCCL over xGMI when the tensors are on HIP devices) reassembles outputs for a
single-device consumer.  Per-sample matrices / sigmas shard with the batch; shared ones (leading
dimension 1) are replicated.  The same code runs on CPU tensors with the ``gloo`` backend (tests).

The exchange costs more than the compute it follows (config 2 on 8 GPUs: 0.16 ms of kernels per rank against
0.66 ms for 100 MB over one xGMI link), so

* ``gather_batch`` writes straight into the final ``(batch, ...)`` tensor (no padding, no ``cat``): one
  ``all_gather_into_tensor`` when the slices are even, otherwise - or with ``mode="p2p"`` - world-1 point-to-point
  sends and receives posted as ONE group (``batch_isend_irecv``: on RCCL every peer link carries its own copy at
  the same time, which is what a fully connected xGMI node wants; a ring is bound by one link and 7 hops);
* ``sharded_apply(..., chunks=k)`` cuts the rank's slice into k sub-batches and exchanges sub-batch i while
  sub-batch i+1 is being computed (the exchange runs on the process group's own stream).
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


def _is_batched(k: int, t, batch: int, batched: Optional[Sequence[int]]) -> bool:
    if batched is not None:
        if k in batched:
            if not (isinstance(t, torch.Tensor) and t.dim() > 0 and t.shape[0] == batch):
                raise ValueError(f"argument {k} was declared batched but its leading dimension is not {batch}")
            return True
        return False
    if isinstance(t, torch.Tensor) and t.dim() == 1 and t.shape[0] == batch and batch > 4:
        # a (B,) vector - per-sample angle / sigma / scale, or a coincidence? It is NOT sliced by the shape rule (a fill_value of shape
        # (3,) is not a batch of three); replicated, it meets its rank's slice inside the op or in gather_batch with an error that does
        # not name the cause.  Sizes up to 4 (channel-like) are left alone; anything longer must be declared.
        raise ValueError(
            f"argument {k} is a one-dimensional tensor whose length equals the batch size {batch}: it is ambiguous whether it is a per-sample vector. "
            f"List the per-sample arguments in batched= (e.g. batched=(0, {k})), or pass batched=(...) without it to replicate it.")
    return isinstance(t, torch.Tensor) and t.dim() >= 2 and t.shape[0] == batch and batch != 1


def _slice_batched(tensors, batch: int, lo: int, hi: int, batched: Optional[Sequence[int]]):
    return [t[lo:hi] if _is_batched(k, t, batch, batched) else t for k, t in enumerate(tensors)]


def shard_batch(tensors: Sequence[Optional[torch.Tensor]], batch: int, world_size: int, rank: int,
                batched: Optional[Sequence[int]] = None):
    """Slice the batched tensors to this rank's contiguous range; pass everything else through unchanged.

    ``batched`` = positions (in ``tensors``) of the arguments that carry the batch on their leading dimension.
    Without it the rule is by shape: a tensor with at least two dimensions whose leading one equals ``batch`` is
    batched (images, (B,3,3) matrices, (B,2) sigmas); one-dimensional tensors never are - a ``fill_value`` of shape
    (3,) is not a batch of three - so per-sample vectors must be named through ``batched``.  A one-dimensional tensor whose length
    equals a batch size above 4 is refused as ambiguous instead of being silently replicated."""
    lo, hi = shard_bounds(batch, world_size, rank)
    return _slice_batched(tensors, batch, lo, hi, batched)


def _global_rank(group, r: int) -> int:
    return dist.get_global_rank(group, r) if group is not None else r


def _peer_exchange(out: torch.Tensor, spans: Sequence[tuple[int, int]], group) -> list:
    """Direct exchange: this rank's rows ``out[spans[rank]]`` (already in place) go to every peer and every peer's rows
    land in their final place in ``out``.  All sends / receives are posted as one group; returns the requests."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = spans[rank]
    mine = out[lo:hi]
    ops = []
    for step in range(1, world):  # staggered: rank r starts with r+1, so no two ranks open on the same target
        dst, src = (rank + step) % world, (rank - step) % world
        if hi > lo:
            ops.append(dist.P2POp(dist.isend, mine, _global_rank(group, dst), group))
        slo, shi = spans[src]
        if shi > slo:
            ops.append(dist.P2POp(dist.irecv, out[slo:shi], _global_rank(group, src), group))
    return dist.batch_isend_irecv(ops) if ops else []


def _check_mode(mode: str) -> None:
    if mode not in ("all_gather", "p2p"):
        raise ValueError(f"mode must be 'all_gather' or 'p2p', got {mode!r}")


def gather_batch(local: torch.Tensor, batch: int, group=None, mode: str = "all_gather") -> torch.Tensor:
    """Reassemble the per-rank slices into a (batch, ...) tensor on every rank.

    ``mode="all_gather"``: one ``all_gather_into_tensor`` straight into the result when the slices are even (uneven
    slices use the peer exchange: nothing is padded or concatenated).  ``mode="p2p"``: always the direct peer exchange."""
    _check_mode(mode)
    world = dist.get_world_size(group)
    if world == 1:
        return local
    rank = dist.get_rank(group)
    spans = [shard_bounds(batch, world, r) for r in range(world)]
    if local.shape[0] != spans[rank][1] - spans[rank][0]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, expected {spans[rank][1] - spans[rank][0]} of {batch}")
    out = torch.empty(batch, *local.shape[1:], dtype=local.dtype, device=local.device)
    if mode == "all_gather" and batch % world == 0:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    out[spans[rank][0] : spans[rank][1]].copy_(local)
    for req in _peer_exchange(out, spans, group):
        req.wait()
    return out


def sharded_apply(op: Callable[..., torch.Tensor], *tensors, gather: bool = True, group=None,
                  batched: Optional[Sequence[int]] = None, chunks: int = 1, mode: str = "all_gather") -> torch.Tensor:
    """Run ``op`` on this rank's slice of the batch (first tensor's leading dim) and optionally gather.

    ``op`` is any per-sample-independent callable, e.g.
    ``lambda x, M: gaussian_blur2d(warp_perspective(x, M, (512, 512)), (5, 5), (1.5, 1.5))``.
    ``batched``: see :func:`shard_batch`.  ``chunks`` > 1 (with ``gather``): the slice is processed in that many
    sub-batches and sub-batch i is exchanged peer-to-peer while sub-batch i+1 is computed."""
    _check_mode(mode)
    if not (dist.is_available() and dist.is_initialized()):
        return op(*tensors)
    if not (isinstance(tensors[0], torch.Tensor) and tensors[0].dim() > 0):
        raise ValueError("the first argument must be a batched tensor")
    if batched is not None and 0 not in batched:
        raise ValueError("argument 0 defines the batch and must be listed in `batched`")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    batch = tensors[0].shape[0]
    lo, hi = shard_bounds(batch, world, rank)
    if not gather or world == 1:
        return op(*_slice_batched(tensors, batch, lo, hi, batched))
    if chunks <= 1:
        return gather_batch(op(*_slice_batched(tensors, batch, lo, hi, batched)), batch, group, mode)

    # ---- chunked: sub-batch i travels while sub-batch i+1 is computed ----
    # The exchange is collective, so every rank runs the same number of rounds; in round c rank r contributes the
    # rows shard_bounds(len_r, chunks, c) of its slice (possibly none).
    bounds = [shard_bounds(batch, world, r) for r in range(world)]
    if min(h - l for l, h in bounds) < 1:
        raise ValueError(f"batch {batch} leaves a rank of {world} without rows; gather unchunked instead")
    out: Optional[torch.Tensor] = None
    pending: list = []
    for c in range(chunks):
        spans = []
        for rlo, rhi in bounds:
            clo, chi = shard_bounds(rhi - rlo, chunks, c)
            spans.append((rlo + clo, rlo + chi))
        clo, chi = spans[rank]
        if chi > clo:
            part = op(*_slice_batched(tensors, batch, clo, chi, batched))
            if out is None:
                out = torch.empty(batch, *part.shape[1:], dtype=part.dtype, device=part.device)
            out[clo:chi].copy_(part)
        elif out is None:  # the first sub-batch is never empty (slices hold >= 1 row and the remainder goes first)
            raise AssertionError("empty first sub-batch")
        # requests are queued behind the work above on the current stream, the next round's kernels are not behind them
        pending.extend(_peer_exchange(out, spans, group))
    for req in pending:
        req.wait()
    assert out is not None
    return out
