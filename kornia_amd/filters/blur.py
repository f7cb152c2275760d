"""box_blur / laplacian on the native filters (filter2d callers).

Reference behaviour mirrored: kornia/filters/blur.py:28-76 (box_blur), :79-151 (BoxBlur), kornia/filters/laplacian.py:30-63
(laplacian), :66-118 (Laplacian), kernels kornia/filters/kernels.py:299-335 (box), :780-847 (laplacian).  The taps are O(k)
host tensors; the filtering is km_filter2d_fwd (register-tiled for 3x3 / 5x5 / 7x7) or km_filter2d_sep_fwd.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from ..core.check import KORNIA_CHECK_IS_TENSOR
from .filter import filter2d, filter2d_separable
from .kernels import _check_kernel_size, _unpack_2d_ks, normalize_kernel2d

__all__ = ["BoxBlur", "Laplacian", "UnsharpMask", "box_blur", "unsharp_mask", "get_box_kernel1d", "get_box_kernel2d", "get_laplacian_kernel1d", "get_laplacian_kernel2d",
           "laplacian"]


def get_box_kernel1d(kernel_size: int, *, device=None, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """(1, k) filled with 1/k."""
    return torch.tensor(1.0 / kernel_size, device=device, dtype=dtype).expand(1, kernel_size)


def get_box_kernel2d(kernel_size, *, device=None, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """(1, ky, kx) filled with 1/(ky*kx)."""
    ky, kx = _unpack_2d_ks(kernel_size)
    return torch.tensor(1.0 / (kx * ky), device=device, dtype=dtype).expand(1, ky, kx)


def get_laplacian_kernel1d(kernel_size: int, *, device=None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """(k,) ones with the centre set so that the taps sum to zero."""
    _check_kernel_size(kernel_size)
    k = torch.ones(kernel_size, device=device, dtype=dtype)
    k[kernel_size // 2] = 1 - kernel_size
    return k


def get_laplacian_kernel2d(kernel_size, *, device=None, dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """(ky, kx) ones with the centre set so that the taps sum to zero."""
    ky, kx = _unpack_2d_ks(kernel_size)
    _check_kernel_size((ky, kx))
    k = torch.ones((ky, kx), device=device, dtype=dtype)
    k[ky // 2, kx // 2] = 1 - k.sum()
    return k


def box_blur(input: torch.Tensor, kernel_size, border_type: str = "reflect", separable: bool = False) -> torch.Tensor:
    """Mean filter over a ``kernel_size`` window."""
    KORNIA_CHECK_IS_TENSOR(input)
    if separable:
        ky, kx = _unpack_2d_ks(kernel_size)
        return filter2d_separable(input, get_box_kernel1d(kx, device=input.device, dtype=input.dtype),
                                  get_box_kernel1d(ky, device=input.device, dtype=input.dtype), border_type)
    return filter2d(input, get_box_kernel2d(kernel_size, device=input.device, dtype=input.dtype), border_type)


def laplacian(input: torch.Tensor, kernel_size, border_type: str = "reflect", normalized: bool = True) -> torch.Tensor:
    """Laplacian of the image (taps optionally L1-normalised)."""
    kernel = get_laplacian_kernel2d(kernel_size, device=input.device, dtype=input.dtype)[None, ...]
    if normalized:
        kernel = normalize_kernel2d(kernel)
    return filter2d(input, kernel, border_type)


def unsharp_mask(input: torch.Tensor, kernel_size, sigma, border_type: str = "reflect") -> torch.Tensor:
    """``blur + 2 * (input - blur)`` with a Gaussian blur (kornia/filters/unsharp.py:28-61: ``torch.lerp(blur, input, 2)``).
    The blur is the native kernel; the lerp is one elementwise pass left to PyTorch (not yet folded into the blur's epilogue)."""
    from .gaussian import gaussian_blur2d

    return torch.lerp(gaussian_blur2d(input, kernel_size, sigma, border_type), input, weight=2.0)


class UnsharpMask(nn.Module):
    """Module form of :func:`unsharp_mask`."""

    def __init__(self, kernel_size, sigma, border_type: str = "reflect") -> None:
        super().__init__()
        self.kernel_size = kernel_size
        self.sigma = sigma
        self.border_type = border_type

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return unsharp_mask(input, self.kernel_size, self.sigma, self.border_type)


class BoxBlur(nn.Module):
    """Module form of :func:`box_blur`."""

    def __init__(self, kernel_size, border_type: str = "reflect", separable: bool = False) -> None:
        super().__init__()
        self.kernel_size = kernel_size
        self.border_type = border_type
        self.separable = separable

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(kernel_size={self.kernel_size}, border_type={self.border_type}, separable={self.separable})"

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return box_blur(input, self.kernel_size, self.border_type, self.separable)


class Laplacian(nn.Module):
    """Module form of :func:`laplacian`."""

    def __init__(self, kernel_size, border_type: str = "reflect", normalized: bool = True) -> None:
        super().__init__()
        self.kernel_size = kernel_size
        self.border_type = border_type
        self.normalized = normalized

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(kernel_size={self.kernel_size}, normalized={self.normalized}, border_type={self.border_type})"

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return laplacian(input, self.kernel_size, self.border_type, self.normalized)
