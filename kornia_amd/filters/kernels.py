"""Filter-tap builders of the hot path (kornia/filters/kernels.py).

These stay in PyTorch on purpose: they are O(kernel size) work and they
define the tap values bit for bit (including taps computed in bf16 for bf16 inputs); the native
kernels receive the taps as data.  ``normalize_kernel2d`` :68-74, ``gaussian`` :77-120,
``get_gaussian_kernel1d`` :552-584, ``get_gaussian_kernel2d`` :661-715, sobel/diff stacks
:357-396 and :470-529.
"""
from __future__ import annotations

from typing import Optional, Union

import torch

from ..core.check import KORNIA_CHECK, KORNIA_CHECK_IS_TENSOR, KORNIA_CHECK_SHAPE

__all__ = [
    "gaussian",
    "get_gaussian_kernel1d",
    "get_gaussian_kernel2d",
    "get_spatial_gradient_kernel2d",
    "normalize_kernel2d",
]


def _check_kernel_size(kernel_size, min_value: int = 0, allow_even: bool = False) -> None:
    if isinstance(kernel_size, int):
        kernel_size = (kernel_size,)
    fmt = "even or odd" if allow_even else "odd"
    for size in kernel_size:
        KORNIA_CHECK(
            isinstance(size, int) and (((size % 2 == 1) or allow_even) and size > min_value),
            f"Kernel size must be an {fmt} integer bigger than {min_value}. Gotcha {size} on {kernel_size}",
        )


def _unpack_2d_ks(kernel_size) -> tuple[int, int]:
    if isinstance(kernel_size, int):
        ky = kx = kernel_size
    else:
        KORNIA_CHECK(len(kernel_size) == 2, "2D Kernel size should have a length of 2.")
        ky, kx = kernel_size
    return int(ky), int(kx)


def normalize_kernel2d(input: torch.Tensor) -> torch.Tensor:
    """Divide by the L1 norm over the last two dims."""
    KORNIA_CHECK_SHAPE(input, ["*", "H", "W"])
    norm = input.abs().sum(dim=-1).sum(dim=-1)
    return input / (norm[..., None, None])


def gaussian(
    window_size: int,
    sigma: Union[torch.Tensor, float],
    *,
    mean: Optional[Union[torch.Tensor, float]] = None,
    device: Optional[torch.device] = None,
    dtype: Optional[torch.dtype] = None,
) -> torch.Tensor:
    """(B, window_size) normalised samples of exp(-(x-mean)^2 / (2 sigma^2)); sigma: float or (B,1)."""
    if isinstance(sigma, float):
        sigma = torch.tensor([[sigma]], device=device, dtype=dtype)
    KORNIA_CHECK_IS_TENSOR(sigma)
    KORNIA_CHECK_SHAPE(sigma, ["B", "1"])
    batch_size = sigma.shape[0]
    mean = float(window_size // 2) if mean is None else mean
    if isinstance(mean, float):
        mean = torch.tensor([[mean]], device=sigma.device, dtype=sigma.dtype)
    KORNIA_CHECK_IS_TENSOR(mean)
    KORNIA_CHECK_SHAPE(mean, ["B", "1"])
    x = (torch.arange(window_size, device=sigma.device, dtype=sigma.dtype) - mean).expand(batch_size, -1)
    if window_size % 2 == 0:
        x = x + 0.5
    gauss = torch.exp(-x.pow(2.0) / (2 * sigma.pow(2.0)))
    return gauss / gauss.sum(-1, keepdim=True)


def get_gaussian_kernel1d(
    kernel_size: int,
    sigma: Union[float, torch.Tensor],
    force_even: bool = False,
    *,
    device: Optional[torch.device] = None,
    dtype: Optional[torch.dtype] = None,
) -> torch.Tensor:
    """(B, kernel_size) Gaussian taps, e.g. (5, 1.5) -> [0.1201, 0.2339, 0.2921, 0.2339, 0.1201]."""
    _check_kernel_size(kernel_size, allow_even=force_even)
    return gaussian(kernel_size, sigma, device=device, dtype=dtype)


def get_gaussian_kernel2d(
    kernel_size,
    sigma,
    force_even: bool = False,
    *,
    device: Optional[torch.device] = None,
    dtype: Optional[torch.dtype] = None,
) -> torch.Tensor:
    """(B, ky, kx) outer product of the two 1-D kernels; sigma = (sigma_y, sigma_x) or (B,2)."""
    if isinstance(sigma, tuple):
        sigma = torch.tensor([sigma], device=device, dtype=dtype)
    KORNIA_CHECK_IS_TENSOR(sigma)
    KORNIA_CHECK_SHAPE(sigma, ["B", "2"])
    ksize_y, ksize_x = _unpack_2d_ks(kernel_size)
    sigma_y, sigma_x = sigma[:, 0, None], sigma[:, 1, None]
    kernel_y = get_gaussian_kernel1d(ksize_y, sigma_y, force_even, device=device, dtype=dtype)[..., None]
    kernel_x = get_gaussian_kernel1d(ksize_x, sigma_x, force_even, device=device, dtype=dtype)[..., None]
    return kernel_y * kernel_x.view(-1, 1, ksize_x)


# derivative operators (mathematical constants of the Sobel / central-difference operators)
_SOBEL_3 = ((-1.0, 0.0, 1.0), (-2.0, 0.0, 2.0), (-1.0, 0.0, 1.0))
_DIFF_3 = ((-0.0, 0.0, 0.0), (-1.0, 0.0, 1.0), (-0.0, 0.0, 0.0))
_SOBEL_5_XX = (
    (-1.0, 0.0, 2.0, 0.0, -1.0),
    (-4.0, 0.0, 8.0, 0.0, -4.0),
    (-6.0, 0.0, 12.0, 0.0, -6.0),
    (-4.0, 0.0, 8.0, 0.0, -4.0),
    (-1.0, 0.0, 2.0, 0.0, -1.0),
)
_SOBEL_5_XY = (
    (-1.0, -2.0, 0.0, 2.0, 1.0),
    (-2.0, -4.0, 0.0, 4.0, 2.0),
    (0.0, 0.0, 0.0, 0.0, 0.0),
    (2.0, 4.0, 0.0, -4.0, -2.0),
    (1.0, 2.0, 0.0, -2.0, -1.0),
)
_DIFF_XX = ((0.0, 0.0, 0.0), (1.0, -2.0, 1.0), (0.0, 0.0, 0.0))
_DIFF_XY = ((-1.0, 0.0, 1.0), (0.0, 0.0, 0.0), (1.0, 0.0, -1.0))


def get_spatial_gradient_kernel2d(
    mode: str, order: int, *, device: Optional[torch.device] = None, dtype: Optional[torch.dtype] = None
) -> torch.Tensor:
    """(2,k,k) [d/dx, d/dy] for order 1, (3,k,k) [dxx, dxy, dyy] for order 2; mode 'sobel' | 'diff'."""
    KORNIA_CHECK(mode.lower() in {"sobel", "diff"}, f"Mode should be `sobel` or `diff`. Got {mode}")
    KORNIA_CHECK(order in {1, 2}, f"Order should be 1 or 2. Got {order}")
    t = lambda rows: torch.tensor(rows, device=device, dtype=dtype)  # noqa: E731
    if mode == "sobel" and order == 1:
        kx = t(_SOBEL_3)
        return torch.stack([kx, kx.transpose(0, 1)])
    if mode == "sobel" and order == 2:
        gxx = t(_SOBEL_5_XX)
        return torch.stack([gxx, t(_SOBEL_5_XY), gxx.transpose(0, 1)])
    if mode == "diff" and order == 1:
        kx = t(_DIFF_3)
        return torch.stack([kx, kx.transpose(0, 1)])
    if mode == "diff" and order == 2:
        gxx = t(_DIFF_XX)
        return torch.stack([gxx, t(_DIFF_XY), gxx.transpose(0, 1)])
    raise NotImplementedError(f"Not implemented for order {order} on mode {mode}")
