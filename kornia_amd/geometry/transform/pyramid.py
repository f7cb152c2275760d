"""Image pyramids on the native kernels.

Reference behaviour mirrored: kornia/geometry/transform/pyramid.py - pyrdown :409-453, pyrup :460-502, build_pyramid
:505-560, build_laplacian_pyramid :572-654, PyrDown :50-99, PyrUp :102-148.

  * pyrdown is ONE launch, km_pyrdown_fwd (csrc/km_pyramid.hip): blur with the binomial kernel and bilinear decimation
    fused, the blurred image is never written (1.25 e instead of 3.25 e bytes per input element at factor 2);
  * pyrup is km_resize_bilinear_fwd + the register-tiled 5x5 km_filter2d_fwd;
  * backward: both ops are linear, nothing of the forward is saved; the adjoint of the resize is km_resize_bilinear_bwd (a
    deterministic gather, no atomics), the adjoint of the blur km_filter2d_bwd_input - no ATen kernel in the path.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from ... import _native as N
from ...core.check import KORNIA_CHECK, KORNIA_CHECK_SHAPE
from ...filters.filter import _BORDER_CODE, _VALID_BORDERS, filter2d, filter2d_separable
from ...filters.gaussian import gaussian_blur2d

__all__ = ["PyrDown", "PyrUp", "ScalePyramid", "build_laplacian_pyramid", "build_pyramid", "pyrdown", "pyrup", "resize_bilinear"]


def _get_pyramid_gaussian_kernel() -> torch.Tensor:
    """(1, 5, 5) binomial kernel / 256 (pyramid.py:32-47)."""
    r = torch.tensor([1.0, 4.0, 6.0, 4.0, 1.0])
    return (r[:, None] * r[None, :] / 256.0)[None]


def _check_border(border_type: str) -> str:
    KORNIA_CHECK(str(border_type).lower() in _VALID_BORDERS, f"Invalid border, {border_type}. Expected one of {_VALID_BORDERS}")
    return str(border_type).lower()


def _resize_adjoint(gy: torch.Tensor, shape, oh: int, ow: int, align: int, dtype) -> torch.Tensor:
    """km_resize_bilinear_bwd: (B,C,oh,ow) gradient -> (B,C,H,W), one launch, written completely."""
    B, C, H, W = shape
    g = gy.detach().to(dtype).contiguous()
    gx = torch.empty(B, C, H, W, device=g.device, dtype=dtype)
    with N.device_guard(g.device):
        N.check(N.lib().km_resize_bilinear_bwd(g.data_ptr(), gx.data_ptr(), B, C, H, W, oh, ow, align, N.dtype_code(dtype), N.stream_ptr(g.device)),
                "km_resize_bilinear_bwd")
    return gx


class _ResizeBilinearFunction(torch.autograd.Function):
    """Native forward (km_resize_bilinear_fwd) and adjoint (km_resize_bilinear_bwd)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, oh: int, ow: int, align: int):
        xc = x.detach().contiguous()
        B, C, H, W = xc.shape
        out = torch.empty(B, C, oh, ow, device=xc.device, dtype=xc.dtype)
        with N.device_guard(xc.device):
            N.check(N.lib().km_resize_bilinear_fwd(xc.data_ptr(), out.data_ptr(), B, C, H, W, oh, ow, align, N.dtype_code(xc.dtype),
                                                   N.stream_ptr(xc.device)), "km_resize_bilinear_fwd")
        ctx.cfg = ((B, C, H, W), oh, ow, align, xc.dtype)
        return out

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        shape, oh, ow, align, dtype = ctx.cfg
        return _resize_adjoint(gy, shape, oh, ow, align, dtype), None, None, None


class _PyrDownFunction(torch.autograd.Function):
    """Fused native forward (km_pyrdown_fwd).  Backward: the resize's adjoint (km_resize_bilinear_bwd) followed by the adjoint of the
    5x5 blur (km_filter2d_bwd_input) - both native, both linear, so nothing of the forward needs to be kept."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, oh: int, ow: int, border: int, align: int):
        xc = x.detach().contiguous()
        B, C, H, W = xc.shape
        out = torch.empty(B, C, oh, ow, device=xc.device, dtype=xc.dtype)
        with N.device_guard(xc.device):
            N.check(N.lib().km_pyrdown_fwd(xc.data_ptr(), out.data_ptr(), B, C, H, W, oh, ow, border, align, N.dtype_code(xc.dtype),
                                           N.stream_ptr(xc.device)), "km_pyrdown_fwd")
        ctx.cfg = ((B, C, H, W), oh, ow, border, align, xc.dtype)
        return out

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        (B, C, H, W), oh, ow, border, align, dtype = ctx.cfg
        g_blur = _resize_adjoint(gy, (B, C, H, W), oh, ow, align, dtype)
        taps = _get_pyramid_gaussian_kernel().to(device=gy.device, dtype=N.compute_dtype(dtype)).contiguous()
        gx = torch.empty_like(g_blur)
        with N.device_guard(gy.device):
            N.check(N.lib().km_filter2d_bwd_input(g_blur.data_ptr(), taps.data_ptr(), gx.data_ptr(), B, C, H, W, 1, 5, 5, border, 1,
                                                  N.dtype_code(dtype), N.stream_ptr(gy.device)), "km_filter2d_bwd_input")
        return gx, None, None, None, None


def resize_bilinear(input: torch.Tensor, size, align_corners: bool = False) -> torch.Tensor:
    """``F.interpolate(input, size=size, mode='bilinear', align_corners=align_corners)`` with the forward as one native launch."""
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    N.require_device(input, "input")
    oh, ow = int(size[0]), int(size[1])
    KORNIA_CHECK(oh > 0 and ow > 0, f"Input and output sizes should be greater than 0, got output {oh}x{ow}")
    return _ResizeBilinearFunction.apply(input, oh, ow, int(bool(align_corners)))


def pyrdown(input: torch.Tensor, border_type: str = "reflect", align_corners: bool = False, factor: float = 2.0) -> torch.Tensor:
    """Blur with the 5x5 binomial kernel, then resize to ``(int(H / factor), int(W // factor))``."""
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    border = _check_border(border_type)
    _, _, height, width = input.shape
    oh, ow = int(float(height) / factor), int(float(width) // factor)
    N.require_device(input, "input")
    KORNIA_CHECK(oh > 0 and ow > 0, f"Input and output sizes should be greater than 0, got output {oh}x{ow}")
    return _PyrDownFunction.apply(input, oh, ow, _BORDER_CODE[border], int(bool(align_corners)))


def pyrup(input: torch.Tensor, border_type: str = "reflect", align_corners: bool = False) -> torch.Tensor:
    """Resize to ``(2H, 2W)``, then blur with the 5x5 binomial kernel."""
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    border = _check_border(border_type)
    _, _, height, width = input.shape
    x_up = resize_bilinear(input, (height * 2, width * 2), align_corners)
    if N.lib().km_config_get(b"pyrdown_separable") == 1:
        # opt-in with the separable pyrdown (csrc/km_pyramid.hip): the rank-1 binomial kernel as the fused 5 + 5 tap blur
        taps = torch.tensor([[1.0, 4.0, 6.0, 4.0, 1.0]], device=input.device, dtype=input.dtype) / 16.0
        return filter2d_separable(x_up, taps, taps, border)
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


def _interpolate_bilinear(x: torch.Tensor, size, align_corners: bool) -> torch.Tensor:
    return resize_bilinear(x, size, align_corners)


def _odd_at_most(n: int) -> int:
    return n if n % 2 == 1 else n - 1


class ScalePyramid(nn.Module):
    r"""Gaussian scale space: per octave ``n_levels + extra_levels`` progressively blurred images of one resolution, the next
    octave starting from the level with twice the initial sigma, decimated by two (pyramid.py:151-400).

    Returns ``(pyr, sigmas, pixel_dists)``: per octave a ``(B, C, L, H_o, W_o)`` stack, the ``(B, L)`` nominal sigmas in octave
    pixels and the ``(B, L)`` pixel spacings relative to the input.  Every blur is the native separable filter
    (reflect border, the kernels precomputed once per module), every decimation the native bilinear resize."""

    def __init__(self, n_levels: int = 3, init_sigma: float = 1.6, min_size: int = 15, double_image: bool = False, extra_levels: int = 3) -> None:
        super().__init__()
        self.n_levels, self.extra_levels = n_levels, extra_levels
        self.init_sigma, self.min_size, self.double_image = init_sigma, min_size, double_image
        self.border = min_size // 2 - 1
        self.sigma_step = 2 ** (1.0 / float(n_levels))
        # taps of the first blur (input sigma 0.5, or 1.0 after doubling -> init_sigma) and of every level-to-level increment
        start = 1.0 if double_image else 0.5
        first = self._incremental_sigma(start, init_sigma) if init_sigma > start else None
        self.register_buffer("_gk_init", None if first is None else self._make_gaussian_kernel1d(first, self.get_kernel_size(first)))
        sigma = init_sigma
        for lvl in range(n_levels + extra_levels - 1):
            delta = sigma * math.sqrt(self.sigma_step**2 - 1.0)
            self.register_buffer(f"_gk_{lvl}", self._make_gaussian_kernel1d(delta, self.get_kernel_size(delta)))
            sigma *= self.sigma_step

    def __repr__(self) -> str:
        names = ("n_levels", "init_sigma", "min_size", "extra_levels", "border", "sigma_step", "double_image")
        return f"{type(self).__name__}(" + ", ".join(f"{n}={getattr(self, n)}" for n in names) + ")"

    @staticmethod
    def _incremental_sigma(have: float, want: float) -> float:
        """Gaussians compose in quadrature: the blur that takes an image of sigma ``have`` to sigma ``want``."""
        return max(math.sqrt(want**2 - have**2), 0.01)

    @staticmethod
    def _make_gaussian_kernel1d(sigma: float, ksize: int) -> torch.Tensor:
        offsets = torch.arange(ksize, dtype=torch.float64) - ksize // 2
        taps = torch.exp(-0.5 * offsets**2 / sigma**2)
        return (taps / taps.sum()).float()

    def get_kernel_size(self, sigma: float) -> int:
        """Odd size covering +-4 sigma."""
        ksize = int(2.0 * 4.0 * sigma + 1.0)
        return ksize if ksize % 2 == 1 else ksize + 1

    def _blur_fast(self, x: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
        taps = kernel.to(device=x.device, dtype=x.dtype)[None]
        return filter2d_separable(x, taps, taps, "reflect")

    def _blur(self, x: torch.Tensor, taps: Optional[torch.Tensor], sigma: float, wanted_size: int) -> torch.Tensor:
        """The precomputed taps when they fit into the image, otherwise a Gaussian clipped to the largest odd size that does."""
        smallest = min(x.shape[2], x.shape[3])
        if taps is not None and wanted_size <= smallest:
            return self._blur_fast(x, taps)
        k = min(wanted_size, _odd_at_most(smallest))
        return gaussian_blur2d(x, (k, k), (sigma, sigma))

    def get_first_level(self, input: torch.Tensor) -> tuple[torch.Tensor, float, float]:
        """``(level, sigma, pixel_distance)`` of the image every octave descends from."""
        level, sigma, pixel_distance = input, 0.5, 1.0
        if self.double_image:
            level = _interpolate_bilinear(input, (input.shape[2] * 2, input.shape[3] * 2), True)
            sigma, pixel_distance = 1.0, 0.5
        if self.init_sigma > sigma:
            delta = self._incremental_sigma(sigma, self.init_sigma)
            level = self._blur(level, self._gk_init, delta, self.get_kernel_size(delta))
            sigma = self.init_sigma
        return level, sigma, pixel_distance

    def forward(self, x: torch.Tensor) -> tuple[list[torch.Tensor], list[torch.Tensor], list[torch.Tensor]]:
        batch, per_octave = x.shape[0], self.n_levels + self.extra_levels
        ratio = math.sqrt(self.sigma_step**2 - 1.0)
        level, first_sigma, pixel_distance = self.get_first_level(x)
        stacks, sigmas, pixel_dists = [], [], []
        while True:
            levels = [level]
            sig = torch.full((batch, per_octave), first_sigma if not stacks else self.init_sigma, device=x.device, dtype=x.dtype)
            octave_sigma = self.init_sigma  # nominal sigma of levels[0] in this octave's pixels
            for idx in range(1, per_octave):
                taps = getattr(self, f"_gk_{idx - 1}")
                prev = levels[-1]
                smallest = min(prev.shape[2], prev.shape[3])
                if taps.shape[0] <= smallest:
                    levels.append(self._blur_fast(prev, taps))
                else:
                    k = _odd_at_most(smallest)
                    levels.append(gaussian_blur2d(prev, (k, k), (octave_sigma * ratio,) * 2))
                octave_sigma *= self.sigma_step
                sig[:, idx] = octave_sigma
            stacks.append(torch.stack(levels, 2))
            sigmas.append(sig)
            pixel_dists.append(torch.full((batch, per_octave), pixel_distance, device=x.device, dtype=x.dtype))
            seed = levels[-self.extra_levels]  # the level with twice the octave's initial sigma
            half = (seed.shape[2] // 2, seed.shape[3] // 2)
            if min(half) <= self.min_size:
                return stacks, sigmas, pixel_dists
            level = _interpolate_bilinear(seed, half, True)
            pixel_distance *= 2.0
