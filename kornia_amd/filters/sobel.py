"""spatial_gradient / sobel and their modules (kornia/filters/sobel.py:31-72, :137-173, :176-229,
:288-335) on csrc/km_gradient.hip: every derivative channel from one pass over the taps, replicate
border as an index clamp (no padded copy), and - for ``sobel`` without autograd - the magnitude
fused so the (B,C,2,H,W) stack is never written."""
from __future__ import annotations

import ctypes
from functools import lru_cache

import torch
from torch import nn

from .. import _native as N
from ..core.check import KORNIA_CHECK_IS_TENSOR, KORNIA_CHECK_SHAPE
from .kernels import get_spatial_gradient_kernel2d, normalize_kernel2d

__all__ = ["Sobel", "SpatialGradient", "sobel", "spatial_gradient"]


@lru_cache(maxsize=64)
def _host_kernel(mode: str, order: int, normalized: bool, dtype: torch.dtype):
    """(n_out,k,k) derivative stack built on the host in the INPUT dtype (as the reference does on the
    input's device), widened to the compute dtype; returned with a ctypes view for by-value passing."""
    k = get_spatial_gradient_kernel2d(mode, order, dtype=dtype)
    if normalized:
        k = normalize_kernel2d(k)
    k = k.to(N.compute_dtype(dtype)).contiguous()
    return k, int(k.shape[0]), int(k.shape[1])


class _SpatialGradientFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, mode: str, order: int, normalized: bool):
        xc = x.detach().contiguous()
        B, C, H, W = xc.shape
        k, n_out, kS = _host_kernel(mode, order, normalized, x.dtype)
        out = torch.empty(B, C, n_out, H, W, device=x.device, dtype=x.dtype)
        with N.device_guard(x.device):
            N.check(N.lib().km_spatial_gradient_fwd(xc.data_ptr(), k.data_ptr(), out.data_ptr(), None, B, C, H, W, n_out, kS,
                                                    ctypes.c_double(0.0), N.dtype_code(x.dtype), N.stream_ptr(x.device)),
                    "km_spatial_gradient_fwd")
        ctx.cfg = (mode, order, normalized, (B, C, H, W), x.dtype)
        return out

    @staticmethod
    def backward(ctx, gout: torch.Tensor):
        mode, order, normalized, (B, C, H, W), dtype = ctx.cfg
        k, n_out, kS = _host_kernel(mode, order, normalized, dtype)
        g = gout.detach().to(dtype).contiguous()
        gx = torch.empty(B, C, H, W, device=g.device, dtype=dtype)
        with N.device_guard(g.device):
            N.check(N.lib().km_spatial_gradient_bwd(g.data_ptr(), k.data_ptr(), gx.data_ptr(), B, C, H, W, n_out, kS,
                                                    N.dtype_code(dtype), N.stream_ptr(g.device)), "km_spatial_gradient_bwd")
        return gx, None, None, None


def spatial_gradient(input: torch.Tensor, mode: str = "sobel", order: int = 1, normalized: bool = True) -> torch.Tensor:
    r"""Image derivatives of ``input`` (B,C,H,W): (B,C,2,H,W) = [d/dx, d/dy] for ``order=1``,
    (B,C,3,H,W) = [dxx, dxy, dyy] for ``order=2``; ``mode``: ``'sobel' | 'diff'``."""
    KORNIA_CHECK_IS_TENSOR(input)
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    _host_kernel(mode, order, normalized, input.dtype)  # validates mode / order like the reference
    N.require_device(input, "input")
    return _SpatialGradientFunction.apply(input, mode, order, normalized)


def sobel(input: torch.Tensor, normalized: bool = True, eps: float = 1e-6) -> torch.Tensor:
    r"""Sobel edge magnitude ``sqrt(gx^2 + gy^2 + eps)`` per channel, (B,C,H,W) -> (B,C,H,W)."""
    KORNIA_CHECK_IS_TENSOR(input)
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    N.require_device(input, "input")
    if torch.is_grad_enabled() and input.requires_grad:
        edges = spatial_gradient(input, normalized=normalized)
        gx, gy = edges[:, :, 0], edges[:, :, 1]
        return torch.sqrt(gx * gx + gy * gy + eps)
    xc = input.contiguous()
    B, C, H, W = xc.shape
    k, n_out, kS = _host_kernel("sobel", 1, normalized, input.dtype)
    mag = torch.empty_like(xc)
    with N.device_guard(xc.device):
        N.check(N.lib().km_spatial_gradient_fwd(xc.data_ptr(), k.data_ptr(), None, mag.data_ptr(), B, C, H, W, n_out, kS,
                                                ctypes.c_double(float(eps)), N.dtype_code(xc.dtype), N.stream_ptr(xc.device)),
                "km_spatial_gradient_fwd")
    return mag


class SpatialGradient(nn.Module):
    r"""Module form of :func:`spatial_gradient`."""

    def __init__(self, mode: str = "sobel", order: int = 1, normalized: bool = True) -> None:
        super().__init__()
        self.normalized: bool = normalized
        self.order: int = order
        self.mode: str = mode

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(order={self.order}, normalized={self.normalized}, mode={self.mode})"

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return spatial_gradient(input, self.mode, self.order, self.normalized)


class Sobel(nn.Module):
    r"""Module form of :func:`sobel`."""

    def __init__(self, normalized: bool = True, eps: float = 1e-6) -> None:
        super().__init__()
        self.normalized: bool = normalized
        self.eps: float = eps

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(normalized={self.normalized})"

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return sobel(input, self.normalized, self.eps)
