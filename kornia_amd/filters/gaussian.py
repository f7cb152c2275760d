"""gaussian_blur2d / GaussianBlur2d (kornia/filters/gaussian.py:32-195) on the fused separable kernel."""
from __future__ import annotations

from functools import lru_cache
from typing import Union

import torch
from torch import nn

from .. import _native as N
from ..core.check import KORNIA_CHECK, KORNIA_CHECK_IS_TENSOR, KORNIA_CHECK_SHAPE
from .filter import filter2d, filter2d_separable
from .kernels import _check_kernel_size, _unpack_2d_ks, get_gaussian_kernel1d, get_gaussian_kernel2d

__all__ = ["GaussianBlur2d", "gaussian_blur2d"]


@lru_cache(maxsize=256)
def _cached_taps(ky: int, kx: int, sigma: tuple, dtype: torch.dtype, device: torch.device):
    """1-D Gaussian taps for a Python-float sigma: evaluated once on the HOST with the reference's
    own formula (bit-identical to its CPU result) and kept resident on the device, so the steady
    state of ``gaussian_blur2d(x, k, (sy, sx))`` is exactly one kernel launch and no H2D copy."""
    s = torch.tensor([sigma], dtype=dtype)  # (1,2) on CPU, rounded to the input dtype like gaussian.py:96
    kernel_x = get_gaussian_kernel1d(kx, s[:, 1].view(1, 1))
    kernel_y = get_gaussian_kernel1d(ky, s[:, 0].view(1, 1))
    return kernel_x.to(device), kernel_y.to(device)


def gaussian_blur2d(
    input: torch.Tensor,
    kernel_size: Union[tuple[int, int], int],
    sigma: Union[tuple[float, float], torch.Tensor],
    border_type: str = "reflect",
    separable: bool = True,
) -> torch.Tensor:
    r"""Blur ``input`` (B,C,H,W) with a Gaussian of size ``kernel_size`` (int or (ky, kx), odd) and
    standard deviation ``sigma`` = (sigma_y, sigma_x) floats or a (B,2) tensor (per-sample blur).

    ``border_type``: ``'constant' | 'reflect' | 'replicate' | 'circular'``; ``separable=False`` applies
    the full 2-D kernel instead of two 1-D passes.

    A tuple ``sigma`` is validated on the host (no device sync); a tensor ``sigma`` is checked with the
    reference's ``(sigma > 0).all()`` test, which synchronises when it lives on the device.
    """
    KORNIA_CHECK_IS_TENSOR(input)
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    _check_kernel_size(kernel_size, min_value=0)

    if isinstance(sigma, tuple):
        KORNIA_CHECK(len(sigma) == 2, "Shape dimension mismatch: expected sigma of shape ['B', '2']")
        positive = all(float(s) > 0 for s in sigma)
        KORNIA_CHECK(positive, "sigma must be positive" if positive else f"sigma must be positive, got {sigma}")
        if separable and N.on_device(input):
            ky, kx = _unpack_2d_ks(kernel_size)
            kernel_x, kernel_y = _cached_taps(ky, kx, (float(sigma[0]), float(sigma[1])), input.dtype, input.device)
            return filter2d_separable(input, kernel_x, kernel_y, border_type)
        sigma = torch.tensor([sigma], device=input.device, dtype=input.dtype)
    else:
        KORNIA_CHECK_IS_TENSOR(sigma)
        sigma = sigma.to(device=input.device, dtype=input.dtype)
        KORNIA_CHECK_SHAPE(sigma, ["B", "2"])
        if not torch.compiler.is_compiling():
            positive = bool((sigma > 0).all())
            KORNIA_CHECK(positive, "sigma must be positive" if positive else f"sigma must be positive, got {sigma}")

    if separable:
        ky, kx = _unpack_2d_ks(kernel_size)
        bs = sigma.shape[0]
        kernel_x = get_gaussian_kernel1d(kx, sigma[:, 1].view(bs, 1))
        kernel_y = get_gaussian_kernel1d(ky, sigma[:, 0].view(bs, 1))
        return filter2d_separable(input, kernel_x, kernel_y, border_type)
    kernel = get_gaussian_kernel2d(kernel_size, sigma)
    return filter2d(input, kernel, border_type)


class GaussianBlur2d(nn.Module):
    r"""Module form of :func:`gaussian_blur2d` (same arguments)."""

    def __init__(
        self,
        kernel_size: Union[tuple[int, int], int],
        sigma: Union[tuple[float, float], torch.Tensor],
        border_type: str = "reflect",
        separable: bool = True,
    ) -> None:
        super().__init__()
        self.kernel_size = kernel_size
        self.sigma = sigma
        self.border_type = border_type
        self.separable = separable

    def __repr__(self) -> str:
        return (
            f"{self.__class__.__name__}"
            f"(kernel_size={self.kernel_size}, "
            f"sigma={self.sigma}, "
            f"border_type={self.border_type}, "
            f"separable={self.separable})"
        )

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return gaussian_blur2d(input, self.kernel_size, self.sigma, self.border_type, self.separable)
