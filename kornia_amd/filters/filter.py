"""filter2d / filter2d_separable on the gfx950 kernels (csrc/km_filter.hip).

Reference: kornia/filters/filter.py:54-152 (filter2d), :155-207 (filter2d_separable), :31-51
(_compute_padding).  No padded copy is materialised and the separable case runs both passes in
ONE launch with the intermediate held in LDS.  Differentiable wrt the input and the kernel(s).
"""
from __future__ import annotations

import torch

from .. import _native as N
from ..core.check import KORNIA_CHECK, KORNIA_CHECK_IS_TENSOR, KORNIA_CHECK_SHAPE
from .kernels import normalize_kernel2d

__all__ = ["filter2d", "filter2d_separable"]

_VALID_BORDERS = {"constant", "reflect", "replicate", "circular"}
_VALID_PADDING = {"valid", "same"}
_VALID_BEHAVIOUR = {"conv", "corr"}
_BORDER_CODE = {"constant": 0, "reflect": 1, "replicate": 2, "circular": 3}


def _compute_padding(kernel_size: list[int]) -> list[int]:
    """[left, right, top, bottom] for (kH, kW): front = (k-1)//2, rear = the rest."""
    if len(kernel_size) < 2:
        raise AssertionError(kernel_size)
    out = []
    for k in reversed(kernel_size):
        front = (k - 1) // 2
        out += [front, (k - 1) - front]
    return out


def _check_pad_fits(border: str, kH: int, kW: int, H: int, W: int) -> None:
    # the errors F.pad raises in the reference
    pl, pr, pt, pb = _compute_padding([kH, kW])
    if border == "reflect" and (max(pl, pr) >= W or max(pt, pb) >= H):
        raise RuntimeError(
            "Padding size should be less than the corresponding input dimension, "
            f"but got: padding ({pl}, {pr}) at dimension 3 of input {[H, W]}"
        )
    if border == "circular" and (max(pl, pr) > W or max(pt, pb) > H):
        raise RuntimeError("Padding value causes wrapping around more than once.")


class _Filter2dFunction(torch.autograd.Function):
    """x (B,C,H,W) data dtype; k (Bk,kH,kW) prepared taps in the compute dtype."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, k: torch.Tensor, border: int, same: int):
        xc = x.detach().contiguous()
        kc = k.detach().contiguous()
        B, C, H, W = xc.shape
        Bk, kH, kW = kc.shape
        out = torch.empty(B, C, H if same else H - kH + 1, W if same else W - kW + 1, device=x.device, dtype=x.dtype)
        with N.device_guard(x.device):
            N.check(N.lib().km_filter2d_fwd(xc.data_ptr(), kc.data_ptr(), out.data_ptr(), B, C, H, W, Bk, kH, kW, border, same,
                                            N.dtype_code(x.dtype), N.stream_ptr(x.device)), "km_filter2d_fwd")
        ctx.save_for_backward(xc, kc)
        ctx.cfg = (border, same)
        return out

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        xc, kc = ctx.saved_tensors
        border, same = ctx.cfg
        B, C, H, W = xc.shape
        Bk, kH, kW = kc.shape
        g = gy.detach().to(xc.dtype).contiguous()
        gx = gk = None
        lib = N.lib()
        with N.device_guard(xc.device):
            stream = N.stream_ptr(xc.device)
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(xc)
                N.check(lib.km_filter2d_bwd_input(g.data_ptr(), kc.data_ptr(), gx.data_ptr(), B, C, H, W, Bk, kH, kW, border, same,
                                                  N.dtype_code(xc.dtype), stream), "km_filter2d_bwd_input")
            if ctx.needs_input_grad[1]:
                gk64 = torch.zeros(Bk, kH, kW, device=xc.device, dtype=torch.float64)
                N.check(lib.km_filter2d_bwd_kernel(g.data_ptr(), xc.data_ptr(), gk64.data_ptr(), B, C, H, W, Bk, kH, kW, border,
                                                   same, N.dtype_code(xc.dtype), stream), "km_filter2d_bwd_kernel")
                gk = gk64.to(kc.dtype)
        return gx, gk, None, None


def _filter2d_sep_launch(x: torch.Tensor, kx: torch.Tensor, ky: torch.Tensor, border: int, same: int):
    """The forward launch of the fused separable filter: ``(out, kx, ky)`` (the contiguous kernels it read, for a backward)."""
    xc = x.detach().contiguous()
    kxc, kyc = kx.detach().contiguous(), ky.detach().contiguous()
    B, C, H, W = xc.shape
    Bk, kW = kxc.shape
    kH = kyc.shape[1]
    out = torch.empty(B, C, H if same else H - kH + 1, W if same else W - kW + 1, device=x.device, dtype=x.dtype)
    with N.device_guard(x.device):
        N.check(N.lib().km_filter2d_sep_fwd(xc.data_ptr(), kxc.data_ptr(), kyc.data_ptr(), out.data_ptr(), B, C, H, W, Bk, kH, kW,
                                            border, same, N.dtype_code(x.dtype), N.stream_ptr(x.device)), "km_filter2d_sep_fwd")
    return out, kxc, kyc


class _Filter2dSepFunction(torch.autograd.Function):
    """Fused separable filter; differentiable wrt the input only (kernels that need gradients go
    through two `_Filter2dFunction` passes instead)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, kx: torch.Tensor, ky: torch.Tensor, border: int, same: int):
        out, kxc, kyc = _filter2d_sep_launch(x, kx, ky, border, same)
        ctx.save_for_backward(kxc, kyc)
        ctx.cfg = (border, same, tuple(x.shape), x.dtype)
        return out

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        kxc, kyc = ctx.saved_tensors
        border, same, (B, C, H, W), dtype = ctx.cfg
        Bk, kW = kxc.shape
        kH = kyc.shape[1]
        g = gy.detach().to(dtype).contiguous()
        gx = torch.empty(B, C, H, W, device=g.device, dtype=dtype)
        lib = N.lib()
        code, stream = N.dtype_code(dtype), N.stream_ptr(g.device)
        with N.device_guard(g.device):
            if lib.km_filter2d_sep_supported(kH, kW, same, code) & 2:
                N.check(lib.km_filter2d_sep_bwd_input(g.data_ptr(), kxc.data_ptr(), kyc.data_ptr(), gx.data_ptr(), B, C, H, W, Bk, kH,
                                                      kW, border, same, code, stream), "km_filter2d_sep_bwd_input")
            else:
                # large kernels: the fused adjoint does not fit in LDS - adjoint of the column pass, then of the row pass
                Wo = W if same else W - kW + 1
                gt = torch.empty(B, C, H, Wo, device=g.device, dtype=dtype)
                N.check(lib.km_filter2d_bwd_input(g.data_ptr(), kyc.data_ptr(), gt.data_ptr(), B, C, H, Wo, Bk, kH, 1, border, same, code,
                                                  stream), "km_filter2d_bwd_input")
                N.check(lib.km_filter2d_bwd_input(gt.data_ptr(), kxc.data_ptr(), gx.data_ptr(), B, C, H, W, Bk, 1, kW, border, same, code,
                                                  stream), "km_filter2d_bwd_input")
        return gx, None, None, None, None


def _prepare_kernel(kernel: torch.Tensor, input: torch.Tensor, normalized: bool, behaviour: str) -> torch.Tensor:
    """filter.py:122-129 - flip for 'conv', cast to the input dtype/device, optional L1 normalisation -
    then widen to the compute dtype (values stay rounded to the input dtype)."""
    if str(behaviour).lower() == "conv":
        k = kernel.flip((-2, -1)).to(device=input.device, dtype=input.dtype)
    else:
        k = kernel.to(device=input.device, dtype=input.dtype)
    if normalized:
        k = normalize_kernel2d(k)
    return k.to(N.compute_dtype(input.dtype))


def _validate(input, kernel, kshape, border_type, padding, behaviour="corr"):
    KORNIA_CHECK_IS_TENSOR(input)
    KORNIA_CHECK_SHAPE(input, ["B", "C", "H", "W"])
    KORNIA_CHECK_IS_TENSOR(kernel)
    KORNIA_CHECK_SHAPE(kernel, kshape)
    KORNIA_CHECK(
        str(border_type).lower() in _VALID_BORDERS,
        f"Invalid border, {border_type}. Expected one of {_VALID_BORDERS}",
    )
    KORNIA_CHECK(
        str(padding).lower() in _VALID_PADDING,
        f"Invalid padding mode, {padding}. Expected one of {_VALID_PADDING}",
    )
    KORNIA_CHECK(
        str(behaviour).lower() in _VALID_BEHAVIOUR,
        f"Invalid padding mode, {behaviour}. Expected one of {_VALID_BEHAVIOUR}",
    )


def _check_kernel_batch(Bk: int, B: int, C: int) -> None:
    # the reference's input.view(-1, Bk*C, H, W) (filter.py:142) needs Bk to divide B
    if B % Bk != 0:
        raise RuntimeError(f"shape '[-1, {Bk * C}, ...]' is invalid for input of batch size {B}: kernel batch must divide it")


def filter2d(
    input: torch.Tensor,
    kernel: torch.Tensor,
    border_type: str = "reflect",
    normalized: bool = False,
    padding: str = "same",
    behaviour: str = "corr",
) -> torch.Tensor:
    r"""Correlate (``behaviour='conv'``: convolve) every channel of ``input`` (B,C,H,W) with
    ``kernel`` (1,kH,kW) or (B,kH,kW).

    ``border_type``: ``'constant' | 'reflect' | 'replicate' | 'circular'`` (for ``padding='same'``);
    ``padding='valid'`` shrinks the output to (H-kH+1, W-kW+1); ``normalized`` L1-normalises the kernel.
    """
    _validate(input, kernel, ["B", "H", "W"], border_type, padding, behaviour)
    N.require_device(input, "input")
    B, C, H, W = input.shape
    k = _prepare_kernel(kernel, input, normalized, behaviour)
    Bk, kH, kW = k.shape
    _check_kernel_batch(Bk, B, C)
    same = int(str(padding).lower() == "same")
    border = str(border_type).lower()
    if same:
        _check_pad_fits(border, kH, kW, H, W)
    return _Filter2dFunction.apply(input, k, _BORDER_CODE[border], same)


def filter2d_separable_taps(input: torch.Tensor, taps_x: torch.Tensor, taps_y: torch.Tensor, border_type: str = "reflect") -> torch.Tensor:
    """:func:`filter2d_separable` ('same' padding, correlation) for taps that are ALREADY what ``filter2d`` would make of them: (1|B, k)
    contiguous float32 device tensors whose values are rounded to the image's dtype (``augmentation.gaussian_taps(..., round_to=...)``).
    Skips the cast round trip of the kernels (two ATen launches each for a 16-bit image); everything else - validation, the fused
    launch, the autograd node - is ``filter2d_separable``'s."""
    ok = (isinstance(taps_x, torch.Tensor) and isinstance(taps_y, torch.Tensor) and taps_x.dtype == torch.float32 and taps_y.dtype == torch.float32
          and taps_x.dim() == 2 and taps_y.dim() == 2 and taps_x.shape[0] == taps_y.shape[0] and taps_x.is_contiguous() and taps_y.is_contiguous()
          and isinstance(input, torch.Tensor) and input.dim() == 4 and N.on_device(input) and taps_x.device == input.device and taps_y.device == input.device
          and input.dtype in (torch.float32, torch.bfloat16, torch.float16) and not (taps_x.requires_grad or taps_y.requires_grad))
    if ok:
        kW, kH = taps_x.shape[1], taps_y.shape[1]
        ok = bool(N.lib().km_filter2d_sep_supported(kH, kW, 1, N.dtype_code(input.dtype)) & 1)
    if not ok:
        return filter2d_separable(input, taps_x, taps_y, border_type)
    _validate(input, taps_x[..., None, :], ["B", "H", "W"], border_type, "same")
    B, C, H, W = input.shape
    _check_kernel_batch(taps_x.shape[0], B, C)
    border = str(border_type).lower()
    _check_pad_fits(border, kH, kW, H, W)
    if not (torch.is_grad_enabled() and input.requires_grad):
        return _filter2d_sep_launch(input, taps_x, taps_y, _BORDER_CODE[border], 1)[0]  # (no autograd node to build: ~8 us of host per call)
    return _Filter2dSepFunction.apply(input, taps_x, taps_y, _BORDER_CODE[border], 1)


def filter2d_separable(
    input: torch.Tensor,
    kernel_x: torch.Tensor,
    kernel_y: torch.Tensor,
    border_type: str = "reflect",
    normalized: bool = False,
    padding: str = "same",
) -> torch.Tensor:
    r"""Filter ``input`` (B,C,H,W) with ``kernel_x`` (1|B, kW) along x and then ``kernel_y`` (1|B, kH)
    along y - the composition ``filter2d(filter2d(x, kx[..., None, :]), ky[..., None])`` of the
    reference, executed as one fused launch when the kernels fit the LDS tile."""
    fused = (
        isinstance(input, torch.Tensor) and isinstance(kernel_x, torch.Tensor) and isinstance(kernel_y, torch.Tensor)
        and input.dim() == 4 and kernel_x.dim() == 2 and kernel_y.dim() == 2
        and kernel_x.shape[0] == kernel_y.shape[0]
        and not (kernel_x.requires_grad or kernel_y.requires_grad)
        and N.on_device(input)
    )
    if fused:
        _validate(input, kernel_x[..., None, :], ["B", "H", "W"], border_type, padding)
        B, C, H, W = input.shape
        kW, kH = kernel_x.shape[1], kernel_y.shape[1]
        same = int(str(padding).lower() == "same")
        fused = bool(N.lib().km_filter2d_sep_supported(kH, kW, same, N.dtype_code(input.dtype)) & 1)
    if not fused:
        out_x = filter2d(input, kernel_x[..., None, :], border_type, normalized, padding)
        return filter2d(out_x, kernel_y[..., None], border_type, normalized, padding)
    kx = _prepare_kernel(kernel_x[..., None, :], input, normalized, "corr")[:, 0, :]
    ky = _prepare_kernel(kernel_y[..., None], input, normalized, "corr")[:, :, 0]
    _check_kernel_batch(kx.shape[0], B, C)
    border = str(border_type).lower()
    if same:
        _check_pad_fits(border, kH, kW, H, W)
    return _Filter2dSepFunction.apply(input, kx, ky, _BORDER_CODE[border], same)
