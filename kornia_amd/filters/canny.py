"""Canny edge detector with its filtering front half on the native kernels.

Reference behaviour mirrored: kornia/filters/canny.py:32-161 (canny), :164-244 (Canny).  The Gaussian blur and the Sobel
derivatives - where the pixels are touched by 5x5 and 3x3 windows - are km_filter2d_sep_fwd and km_spatial_gradient_fwd;
magnitude, direction binning, non-maximum suppression, thresholds and hysteresis are pointwise / 3x3-neighbour selections
written here as shifted-view comparisons: the reference's two fixed-kernel convolutions (8 one-hot difference kernels for
the suppression, 8 one-hot kernels for the hysteresis) compute exactly ``m - m[neighbour k]`` and ``edges[neighbour k]``.
"""
from __future__ import annotations

import math
from typing import Union

import torch
import torch.nn.functional as F
from torch import nn

from .. import _native as N
from ..core.check import KORNIA_CHECK, KORNIA_CHECK_IS_TENSOR, KORNIA_CHECK_SHAPE
from .gaussian import gaussian_blur2d
from .sobel import spatial_gradient

__all__ = ["Canny", "canny"]

# neighbour k of the suppression / hysteresis stacks, as (dy, dx): east, south-east, south, ... counter-clockwise in image
# coordinates (kornia/filters/kernels.py:943-976)
_NEIGHBOURS = ((0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1))


def _check_thresholds(low_threshold: float, high_threshold: float) -> None:
    KORNIA_CHECK(low_threshold <= high_threshold,
                 f"Invalid input thresholds. low_threshold should be smaller than the high_threshold. Got: {low_threshold}>{high_threshold}")
    KORNIA_CHECK(0 < low_threshold < 1, f"Invalid low threshold. Should be in range (0, 1). Got: {low_threshold}")
    KORNIA_CHECK(0 < high_threshold < 1, f"Invalid high threshold. Should be in range (0, 1). Got: {high_threshold}")


def _to_gray(image: torch.Tensor) -> torch.Tensor:
    """kornia/color/gray.py:92-105 for floating-point RGB: r*0.299, then two addcmul."""
    w = torch.tensor([0.299, 0.587, 0.114], device=image.device, dtype=image.dtype)
    r, g, b = image.unbind(dim=-3)
    out = torch.addcmul(torch.addcmul(r * w[0], g, w[1]), b, w[2])
    return out.unsqueeze(-3)


def _neighbour_views(x: torch.Tensor) -> list[torch.Tensor]:
    """The eight zero-padded neighbours of every pixel of a (B,1,H,W) map, in `_NEIGHBOURS` order."""
    H, W = x.shape[-2:]
    p = F.pad(x, (1, 1, 1, 1))
    return [p[..., 1 + dy : 1 + dy + H, 1 + dx : 1 + dx + W] for dy, dx in _NEIGHBOURS]


def canny(input: torch.Tensor, low_threshold: float = 0.1, high_threshold: float = 0.2, kernel_size=(5, 5), sigma=(1, 1),
          hysteresis: bool = True, eps: float = 1e-6) -> tuple[torch.Tensor, torch.Tensor]:
    r"""Returns ``(magnitude, edges)``, both (B,1,H,W): the gradient magnitude after non-maximum suppression and the edge map
    (1 strong; with ``hysteresis=False`` 0.5 marks weak edges, otherwise weak edges connected to strong ones are promoted)."""
    KORNIA_CHECK_IS_TENSOR(input)
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    _check_thresholds(low_threshold, high_threshold)
    dtype = input.dtype
    if input.shape[1] == 3:
        input = _to_gray(input)

    blurred = gaussian_blur2d(input, kernel_size, sigma)
    grads = spatial_gradient(blurred, normalized=False)
    if dtype == torch.float32 and grads.is_cuda and not (torch.is_grad_enabled() and grads.requires_grad):
        return _canny_tail_native(grads, low_threshold, high_threshold, hysteresis, eps)
    # differentiable / half-precision calls: the same arithmetic as tensor expressions (autograd through the magnitude)
    gx, gy = grads[:, :, 0], grads[:, :, 1]
    magnitude = torch.sqrt(gx * gx + gy * gy + eps)
    direction = (torch.atan2(gy, gx) * (4 / math.pi)).round()  # -4 .. 4, multiples of 45 degrees

    # non-maximum suppression: a pixel survives when it exceeds both neighbours along its gradient direction
    diffs = torch.cat([magnitude - n for n in _neighbour_views(magnitude)], dim=1)  # (B,8,H,W)
    ahead = torch.gather(diffs, 1, (direction % 8).long())
    behind = torch.gather(diffs, 1, ((direction + 4) % 8).long())
    magnitude = magnitude * (torch.minimum(ahead, behind) > 0.0)

    edges = ((magnitude > low_threshold) * 0.5 + (magnitude > high_threshold) * 0.5).to(dtype)
    if hysteresis:
        previous = -torch.ones_like(edges)
        promoted = edges
        while ((previous - edges).abs() != 0).any():
            weak = (edges == 0.5).float()
            strong = (edges == 1).float()
            touches_strong = torch.stack([n == 1 for n in _neighbour_views(edges)], 0).any(0).to(dtype)
            promoted = touches_strong * weak + strong
            previous = edges.clone()
            edges = promoted + (promoted == 0) * weak * 0.5
        edges = promoted
    return magnitude, edges


def _canny_tail_native(grads: torch.Tensor, low: float, high: float, hysteresis: bool, eps: float) -> tuple[torch.Tensor, torch.Tensor]:
    """Magnitude, direction binning, suppression and thresholds in one launch (km_canny_nms_fwd); the hysteresis as sweeps of
    block-local fixed points (km_canny_hysteresis_sweep), repeated until a sweep promotes nothing.  The reference's loop reads a
    flag back once per promotion step (canny.py:160); this one once per sweep - a few per image whatever the edge length."""
    B, _, _, H, W = grads.shape
    grads = grads.contiguous()
    lib = N.lib()
    dev = grads.device
    magnitude = torch.empty(B, 1, H, W, device=dev, dtype=torch.float32)
    edges = torch.empty_like(magnitude)
    with N.device_guard(dev):
        stream = N.stream_ptr(dev)
        N.check(lib.km_canny_nms_fwd(grads.data_ptr(), magnitude.data_ptr(), edges.data_ptr(), B, H, W, float(low), float(high), float(eps), stream),
                "km_canny_nms_fwd")
        if hysteresis:
            out = torch.empty_like(edges)
            flag = torch.zeros(1, device=dev, dtype=torch.int32)
            while True:
                N.check(lib.km_canny_hysteresis_sweep(edges.data_ptr(), out.data_ptr(), flag.data_ptr(), B, H, W, stream), "km_canny_hysteresis_sweep")
                if int(flag.item()) == 0:
                    break
                flag.zero_()
            edges = out
    return magnitude, edges


class Canny(nn.Module):
    def __init__(self, low_threshold: float = 0.1, high_threshold: float = 0.2, kernel_size=(5, 5), sigma: Union[tuple, torch.Tensor] = (1, 1),
                 hysteresis: bool = True, eps: float = 1e-6) -> None:
        super().__init__()
        _check_thresholds(low_threshold, high_threshold)
        self.kernel_size = kernel_size
        self.sigma = sigma
        self.low_threshold = low_threshold
        self.high_threshold = high_threshold
        self.hysteresis = hysteresis
        self.eps = eps

    def __repr__(self) -> str:
        return (f"{type(self).__name__}(low_threshold={self.low_threshold}, high_threshold={self.high_threshold}, kernel_size={self.kernel_size}, "
                f"sigma={self.sigma}, hysteresis={self.hysteresis}, eps={self.eps})")

    def forward(self, input: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        return canny(input, self.low_threshold, self.high_threshold, self.kernel_size, self.sigma, self.hysteresis, self.eps)
