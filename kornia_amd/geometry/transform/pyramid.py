"""Image pyramids on the native kernels.

Reference behaviour mirrored: kornia/geometry/transform/pyramid.py - pyrdown :409-453, pyrup :460-502, build_pyramid
:505-560, build_laplacian_pyramid :572-654, PyrDown :50-99, PyrUp :102-148.

  * pyrdown without a gradient to record is ONE launch, km_pyrdown_fwd (csrc/km_pyramid.hip): blur with the binomial
    kernel and bilinear decimation fused, the blurred image is never written (1.25 e instead of 3.25 e bytes per input
    element at factor 2);
  * pyrup without a gradient is km_resize_bilinear_fwd + the register-tiled 5x5 km_filter2d_fwd;
  * when autograd has to record the call, the differentiable composition of the reference is used: the native
    filter2d (its backward is native) around F.interpolate, whose backward is ATen's.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from ... import _native as N
from ...core.check import KORNIA_CHECK, KORNIA_CHECK_SHAPE
from ...filters.filter import _BORDER_CODE, _VALID_BORDERS, filter2d

__all__ = ["PyrDown", "PyrUp", "build_laplacian_pyramid", "build_pyramid", "pyrdown", "pyrup", "resize_bilinear"]


def _get_pyramid_gaussian_kernel() -> torch.Tensor:
    """(1, 5, 5) binomial kernel / 256 (pyramid.py:32-47)."""
    r = torch.tensor([1.0, 4.0, 6.0, 4.0, 1.0])
    return (r[:, None] * r[None, :] / 256.0)[None]


def _records_grad(t: torch.Tensor) -> bool:
    return torch.is_grad_enabled() and t.requires_grad


def _check_border(border_type: str) -> str:
    KORNIA_CHECK(str(border_type).lower() in _VALID_BORDERS, f"Invalid border, {border_type}. Expected one of {_VALID_BORDERS}")
    return str(border_type).lower()


def resize_bilinear(input: torch.Tensor, size, align_corners: bool = False) -> torch.Tensor:
    """``F.interpolate(input, size=size, mode='bilinear', align_corners=align_corners)`` (forward only) as one native launch."""
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    N.require_device(input, "input")
    oh, ow = int(size[0]), int(size[1])
    KORNIA_CHECK(oh > 0 and ow > 0, f"Input and output sizes should be greater than 0, got output {oh}x{ow}")
    x = input.contiguous()
    B, C, H, W = x.shape
    out = torch.empty(B, C, oh, ow, device=x.device, dtype=x.dtype)
    with N.device_guard(x.device):
        N.check(N.lib().km_resize_bilinear_fwd(x.data_ptr(), out.data_ptr(), B, C, H, W, oh, ow, int(bool(align_corners)),
                                               N.dtype_code(x.dtype), N.stream_ptr(x.device)), "km_resize_bilinear_fwd")
    return out


def pyrdown(input: torch.Tensor, border_type: str = "reflect", align_corners: bool = False, factor: float = 2.0) -> torch.Tensor:
    """Blur with the 5x5 binomial kernel, then resize to ``(int(H / factor), int(W // factor))``."""
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    border = _check_border(border_type)
    _, _, height, width = input.shape
    oh, ow = int(float(height) / factor), int(float(width) // factor)
    if _records_grad(input):
        x_blur = filter2d(input, _get_pyramid_gaussian_kernel(), border)
        return F.interpolate(x_blur, size=(oh, ow), mode="bilinear", align_corners=align_corners)
    N.require_device(input, "input")
    KORNIA_CHECK(oh > 0 and ow > 0, f"Input and output sizes should be greater than 0, got output {oh}x{ow}")
    x = input.contiguous()
    B, C, H, W = x.shape
    out = torch.empty(B, C, oh, ow, device=x.device, dtype=x.dtype)
    with N.device_guard(x.device):
        N.check(N.lib().km_pyrdown_fwd(x.data_ptr(), out.data_ptr(), B, C, H, W, oh, ow, _BORDER_CODE[border], int(bool(align_corners)),
                                       N.dtype_code(x.dtype), N.stream_ptr(x.device)), "km_pyrdown_fwd")
    return out


def pyrup(input: torch.Tensor, border_type: str = "reflect", align_corners: bool = False) -> torch.Tensor:
    """Resize to ``(2H, 2W)``, then blur with the 5x5 binomial kernel."""
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    border = _check_border(border_type)
    _, _, height, width = input.shape
    if _records_grad(input):
        x_up = F.interpolate(input, size=(height * 2, width * 2), mode="bilinear", align_corners=align_corners)
    else:
        x_up = resize_bilinear(input, (height * 2, width * 2), align_corners)
    return filter2d(x_up, _get_pyramid_gaussian_kernel(), border)


def build_pyramid(input: torch.Tensor, max_level: int, border_type: str = "reflect", align_corners: bool = False) -> list[torch.Tensor]:
    """``max_level`` levels, level 0 being the input itself, each next one a :func:`pyrdown` of the previous."""
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    KORNIA_CHECK(isinstance(max_level, int) or max_level < 0, f"Invalid max_level, it must be a positive integer. Got: {max_level}")
    pyramid = [input]
    for _ in range(max_level - 1):
        pyramid.append(pyrdown(pyramid[-1], border_type, align_corners))
    return pyramid


def is_powerof_two(x: int) -> bool:
    return bool(x) and (not (x & (x - 1)))


def find_next_powerof_two(x: int) -> int:
    return 1 << (x - 1).bit_length()


def build_laplacian_pyramid(input: torch.Tensor, max_level: int, border_type: str = "reflect", align_corners: bool = False) -> list[torch.Tensor]:
    """Band-pass residuals ``gaussian[i] - pyrup(gaussian[i + 1])`` followed by the last Gaussian level; the input is
    reflect-padded to the next power of two only when neither side is one already (the reference's rule)."""
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    KORNIA_CHECK(isinstance(max_level, int) or max_level < 0, f"Invalid max_level, it must be a positive integer. Got: {max_level}")
    h, w = input.shape[2], input.shape[3]
    if not (is_powerof_two(w) or is_powerof_two(h)):
        input = F.pad(input, (0, find_next_powerof_two(w) - w, 0, find_next_powerof_two(h) - h), "reflect")
    gaussian = build_pyramid(input, max_level, border_type, align_corners)
    laplacian = [gaussian[i] - pyrup(gaussian[i + 1], border_type, align_corners) for i in range(max_level - 1)]
    laplacian.append(gaussian[-1])
    return laplacian


class PyrDown(nn.Module):
    def __init__(self, border_type: str = "reflect", align_corners: bool = False, factor: float = 2.0) -> None:
        super().__init__()
        self.border_type = border_type
        self.align_corners = align_corners
        self.factor = factor

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return pyrdown(input, self.border_type, self.align_corners, self.factor)


class PyrUp(nn.Module):
    def __init__(self, border_type: str = "reflect", align_corners: bool = False) -> None:
        super().__init__()
        self.border_type = border_type
        self.align_corners = align_corners

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return pyrup(input, self.border_type, self.align_corners)
