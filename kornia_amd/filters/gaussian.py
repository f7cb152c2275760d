"""gaussian_blur2d / GaussianBlur2d (kornia/filters/gaussian.py:32-195) on the fused separable kernel."""
from __future__ import annotations

from functools import lru_cache
from typing import Union

import torch
from torch import nn

from .. import _native as N
from ..core.check import KORNIA_CHECK, KORNIA_CHECK_IS_TENSOR, KORNIA_CHECK_SHAPE, device_value_checks
from .filter import filter2d, filter2d_separable
from .kernels import _check_kernel_size, _unpack_2d_ks, get_gaussian_kernel1d, get_gaussian_kernel2d

__all__ = ["GaussianBlur2d", "gaussian_blur2d"]


@lru_cache(maxsize=256)
def _cached_taps(ky: int, kx: int, sigma: tuple, dtype: torch.dtype, device: torch.device):
    """1-D Gaussian taps for a Python-float sigma: evaluated once on the HOST with the reference's
    own formula (bit-identical to its CPU result) and kept resident on the device, so the steady
    state of ``gaussian_blur2d(x, k, (sy, sx))`` is exactly one kernel launch and no H2D copy."""
    # built outside inference mode: a cached inference tensor could not be saved for backward by a later
    # differentiable call with the same key (the reference has no such dependence on call history)
    with torch.inference_mode(False), torch.no_grad():
        s = torch.tensor([sigma], dtype=dtype)  # (1,2) on CPU, rounded to the input dtype like gaussian.py:96
        kernel_x = get_gaussian_kernel1d(kx, s[:, 1].view(1, 1))
        kernel_y = get_gaussian_kernel1d(ky, s[:, 0].view(1, 1))
        return kernel_x.to(device), kernel_y.to(device)


def _check_host_sigma(sigma: tuple) -> tuple[float, float]:
    """A tuple sigma is validated on the host: no tensor is built and nothing synchronises."""
    KORNIA_CHECK(len(sigma) == 2, "Shape dimension mismatch: expected sigma of shape ['B', '2']")
    sy, sx = float(sigma[0]), float(sigma[1])
    KORNIA_CHECK(sy > 0 and sx > 0, f"sigma must be positive, got {sigma}")
    return sy, sx


def _check_tensor_sigma(sigma: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """A tensor sigma is validated where its values live on the host (the reference's ``bool((sigma > 0).all())``,
    gaussian.py:103-108).  A sigma that is already on the device is NOT read back: that would drain the stream in the
    middle of an augmentation pipeline; the Gaussian only uses sigma^2, so a negative entry then acts as
    its magnitude, and a zero gives the limit of the Gaussian for sigma -> 0 - the identity kernel (km_gaussian_taps_fwd) - instead of
    the reference's exception; a NaN sigma gives NaN taps.  (INTEGRATION.md lists the divergence.)  ``kornia_amd.core.check.set_device_value_checks(True)`` restores the
    reference's synchronising check (the reference's own tests run with it)."""
    KORNIA_CHECK_IS_TENSOR(sigma)
    on_host = sigma.device.type == "cpu"
    sigma_in = sigma
    sigma = sigma.to(device=like.device, dtype=like.dtype)
    KORNIA_CHECK_SHAPE(sigma, ["B", "2"])
    if not torch.compiler.is_compiling():
        if on_host:  # the values are host data: test them there (rounded to the input dtype like the reference does)
            positive = bool((sigma_in.to(like.dtype) > 0).all())
            KORNIA_CHECK(positive, "sigma must be positive" if positive else f"sigma must be positive, got {sigma_in}")
        elif device_value_checks():
            positive = bool((sigma > 0).all())
            KORNIA_CHECK(positive, "sigma must be positive" if positive else f"sigma must be positive, got {sigma}")
    return sigma


def gaussian_blur2d(input: torch.Tensor, kernel_size: Union[tuple[int, int], int], sigma: Union[tuple[float, float], torch.Tensor],
                    border_type: str = "reflect", separable: bool = True) -> torch.Tensor:
    r"""Blur ``input`` (B,C,H,W) with a Gaussian of size ``kernel_size`` (int or (ky, kx), odd) and
    standard deviation ``sigma`` = (sigma_y, sigma_x) floats or a (B,2) tensor (per-sample blur).

    ``border_type``: ``'constant' | 'reflect' | 'replicate' | 'circular'``; ``separable=False`` applies
    the full 2-D kernel instead of two 1-D passes.
    """
    KORNIA_CHECK_IS_TENSOR(input)
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    _check_kernel_size(kernel_size, min_value=0)
    ky, kx = _unpack_2d_ks(kernel_size)

    if isinstance(sigma, tuple):
        host_sigma = _check_host_sigma(sigma)
        if separable and N.on_device(input):  # steady state: one launch, taps resident on the device
            taps_x, taps_y = _cached_taps(ky, kx, host_sigma, input.dtype, input.device)
            return filter2d_separable(input, taps_x, taps_y, border_type)
        sigma = torch.tensor([sigma], device=input.device, dtype=input.dtype)
    else:
        sigma = _check_tensor_sigma(sigma, input)
        if (separable and N.on_device(input) and N.on_device(sigma) and input.dtype in (torch.float32, torch.bfloat16, torch.float16) and kx <= 64 and ky <= 64
                and sigma.dim() == 2 and sigma.shape[-1] == 2 and sigma.shape[0] in (1, input.shape[0])  # (what gaussian_taps takes; anything else: the tensor path below)
                and not (torch.is_grad_enabled() and sigma.requires_grad)):
            # per-sample sigma already on the device: both tap vectors in ONE launch (km_gaussian_taps_fwd) instead of the ~16
            # elementwise launches of two get_gaussian_kernel1d calls - the call was host-bound on them (265 us of enqueue for
            # 256 x 3 x 224^2).  Taps are evaluated in float32 (the device's expf: within 2 ulp of the reference's) and rounded to the
            # image dtype by filter2d_separable like any kernel (kornia/filters/filter.py:126).
            from ..augmentation import gaussian_taps

            taps_x, taps_y = gaussian_taps(sigma, (ky, kx))
            return filter2d_separable(input, taps_x, taps_y, border_type)

    if not separable:
        return filter2d(input, get_gaussian_kernel2d(kernel_size, sigma), border_type)
    per_sample = sigma.shape[0]
    # sigma[:, 0] is the vertical deviation, sigma[:, 1] the horizontal one (gaussian.py:111-114)
    taps_x = get_gaussian_kernel1d(kx, sigma[:, 1].view(per_sample, 1))
    taps_y = get_gaussian_kernel1d(ky, sigma[:, 0].view(per_sample, 1))
    return filter2d_separable(input, taps_x, taps_y, border_type)


class GaussianBlur2d(nn.Module):
    r"""Module form of :func:`gaussian_blur2d` (same arguments)."""

    _FIELDS = ("kernel_size", "sigma", "border_type", "separable")

    def __init__(self, kernel_size: Union[tuple[int, int], int], sigma: Union[tuple[float, float], torch.Tensor], border_type: str = "reflect",
                 separable: bool = True) -> None:
        super().__init__()
        self.kernel_size, self.sigma, self.border_type, self.separable = kernel_size, sigma, border_type, separable

    def __repr__(self) -> str:
        return f"{type(self).__name__}(" + ", ".join(f"{name}={getattr(self, name)}" for name in self._FIELDS) + ")"

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return gaussian_blur2d(input, *(getattr(self, name) for name in self._FIELDS))
