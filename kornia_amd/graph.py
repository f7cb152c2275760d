"""HIP-graph capture of whole steps of the native path.

Every native op is an allocation-free, stream-ordered launch sequence with no host synchronisation, so a forward or a
forward + backward step over fixed shapes captures into one HIP graph.  A replay costs one launch on the host instead of
~10 Python -> ctypes -> HIP round trips per op: in the launch-bound regime (small batches - BASELINE config 5 per GPU,
config 3's 224 x 224 images) the eager step is host-bound at ~200 us and the GPU clocks down between its kernels; the
replayed graph runs them back to back.

    step = kornia_amd.graph.capture(lambda x, M: K.gaussian_blur2d(K.warp_perspective(x, M, (S, S)), (5, 5), (1.5, 1.5)), x, M)
    y = step(x_new, M_new)          # copies the inputs into the captured buffers, replays, returns the captured outputs

    def train(x, H, target):        # a step that back-propagates inside
        (g,) = torch.autograd.grad(F.l1_loss(K.homography_warp(x, H, (256, 256)), target), H)
        return g
    step = kornia_amd.graph.capture(train, x, H.requires_grad_(), target)

Rules of capture (HIP's, not ours): the tensors that change between replays must already live on the device (host tensors and
non-tensor arguments are baked into the graph), shapes are frozen, and the function must
not synchronise (``.item()``, ``.tolist()`` on device data, host tensors turned into device tensors inside).  The returned tensors are the graph's own buffers: copy them if they must survive the
next replay.
"""
from __future__ import annotations

from typing import Any, Callable, Sequence

import torch

__all__ = ["GraphedStep", "capture"]


def _tensors(tree: Any) -> list:
    """The device tensors of a (nested) argument list; host tensors (e.g. ColorJitter's sampled ``order``) are constants of the graph."""
    if isinstance(tree, torch.Tensor):
        return [tree] if tree.is_cuda else []
    if isinstance(tree, (list, tuple)):
        return [t for x in tree for t in _tensors(x)]
    if isinstance(tree, dict):
        return [t for x in tree.values() for t in _tensors(x)]
    return []


class GraphedStep:
    """A captured step: call it with new values for the tensor arguments (same shapes / dtypes / devices)."""

    def __init__(self, fn: Callable, *args: Any, warmup: int = 2, no_grad: bool = False) -> None:
        tensors = _tensors(args)
        if not tensors:
            raise ValueError("capture() needs at least one tensor argument on a HIP device")
        self._device = tensors[0].device
        self._fn = fn
        self._no_grad = no_grad

        def clone(t):
            if not t.is_cuda:
                return t  # host tensors are baked into the graph as they are
            c = t.detach().clone()
            return c.requires_grad_(t.requires_grad)

        self._static = self._map(args, clone)
        self._static_flat = _tensors(self._static)
        self._graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=self._device)
        side.wait_stream(torch.cuda.current_stream(self._device))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):  # library load, tap caches, allocator warm-up, autograd graph shapes
                self._run()
        torch.cuda.current_stream(self._device).wait_stream(side)
        with torch.cuda.graph(self._graph):
            self._out = self._run()

    @staticmethod
    def _map(tree: Any, f: Callable) -> Any:
        if isinstance(tree, torch.Tensor):
            return f(tree)
        if isinstance(tree, (list, tuple)):
            return type(tree)(GraphedStep._map(x, f) for x in tree)
        if isinstance(tree, dict):
            return {k: GraphedStep._map(v, f) for k, v in tree.items()}
        return tree

    def _run(self):
        if self._no_grad:
            with torch.no_grad():
                return self._fn(*self._static)
        return self._fn(*self._static)

    def __call__(self, *args: Any):
        new = _tensors(args)
        if len(new) != len(self._static_flat):
            raise ValueError(f"expected {len(self._static_flat)} tensor arguments, got {len(new)}")
        with torch.no_grad():
            for dst, src in zip(self._static_flat, new):
                if dst.shape != src.shape or dst.dtype != src.dtype:
                    raise ValueError(f"captured with {tuple(dst.shape)} {dst.dtype}, called with {tuple(src.shape)} {src.dtype}")
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
        self._graph.replay()
        return self._out

    def replay(self):
        """Replay on the captured buffers as they are (``inputs`` gives access to them)."""
        self._graph.replay()
        return self._out

    @property
    def inputs(self) -> Sequence[torch.Tensor]:
        return self._static_flat


def capture(fn: Callable, *args: Any, warmup: int = 2, no_grad: bool = False) -> GraphedStep:
    """Capture ``fn(*args)`` into a HIP graph; see the module docstring."""
    return GraphedStep(fn, *args, warmup=warmup, no_grad=no_grad)
