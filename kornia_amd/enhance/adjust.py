"""ColorJitter's four adjustments on one fused gfx950 kernel.

Reference functions mirrored (names, argument meaning, output range):
  adjust_brightness_accumulative            kornia/enhance/adjust.py:542-593   clamp(x * f, 0, 1)
  adjust_contrast_with_mean_subtraction     kornia/enhance/adjust.py:414-469   clamp(x * f + mean_gray * (1 - f), 0, 1)
  adjust_saturation_with_gray_subtraction   kornia/enhance/adjust.py:80-134    clamp((1 - f) * gray + f * x, 0, 1)
  adjust_hue                                kornia/enhance/adjust.py:212-254   rgb -> hsv, h = fmod(h + f, 2 pi), hsv -> rgb
and ``color_jitter`` = the sequence ColorJitter.apply_transform runs
(kornia/augmentation/_2d/intensity/color_jitter.py:126-159) in ONE pass over the image.

Differentiable like the reference's op sequences: ``km_color_jitter_bwd`` gives the gradients wrt the image and wrt per-image
factor tensors (contrast couples every pixel of an image through the mean: one extra reduction pass).
RGB ``(B,3,H,W)`` / ``(3,H,W)`` inputs on a HIP device; there is no PyTorch/CPU fallback.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Sequence, Union

import torch

from .. import _native as N

__all__ = [
    "adjust_brightness_accumulative",
    "adjust_contrast_with_mean_subtraction",
    "adjust_hue",
    "adjust_saturation_with_gray_subtraction",
    "color_jitter",
    "color_jitter_from_table",
]

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3
_NEUTRAL = (1.0, 1.0, 1.0, 0.0)
Factor = Union[float, torch.Tensor, None]


def _factor_column(f: Factor, B: int, device, neutral: float) -> torch.Tensor:
    if f is None:
        return torch.full((B,), neutral, device=device, dtype=torch.float32)
    if isinstance(f, (int, float)):
        return torch.full((B,), float(f), device=device, dtype=torch.float32)
    if not isinstance(f, torch.Tensor):
        raise TypeError(f"Factor should be float or torch.Tensor. Got {type(f)}")
    f = f.to(device=device, dtype=torch.float32).reshape(-1)
    if f.numel() == 1:
        return f.expand(B)
    if f.numel() != B:
        raise ValueError(f"factor has {f.numel()} elements, expected 1 or the batch size {B}")
    return f


def _color_launch(x: torch.Tensor, params: torch.Tensor, enable: Optional[torch.Tensor], apply: Optional[torch.Tensor], stages: tuple,
                  gray_ws: Optional[torch.Tensor] = None):
    """The forward launch of the fused colour kernel: ``(out, x, params, gray_sum)`` (the contiguous operands it read, for a backward)."""
    B, _, H, W = x.shape
    dev = x.device
    x = x.contiguous()
    params = params.contiguous()
    out = torch.empty_like(x)
    # (gray_ws: a (B,) float64 workspace that the launch which made `params` has ALREADY zeroed - km_color_params_ws_fwd - instead of a fill launch here)
    gray_sum = (gray_ws if gray_ws is not None else torch.zeros(B, device=dev, dtype=torch.float64)) if CONTRAST in stages else None
    arr = (ctypes.c_int * max(len(stages), 1))(*stages)
    with N.device_guard(dev):
        if apply is None:
            N.check(N.lib().km_color_jitter_fwd(x.data_ptr(), out.data_ptr(), params.data_ptr(), N.ptr(gray_sum), N.ptr(enable), arr, len(stages), B, H, W,
                                                N.dtype_code(x.dtype), N.stream_ptr(dev)), "km_color_jitter_fwd")
        else:
            N.check(N.lib().km_color_jitter_fwd_masked(x.data_ptr(), out.data_ptr(), params.data_ptr(), N.ptr(gray_sum), N.ptr(enable), apply.data_ptr(), arr,
                                                       len(stages), B, H, W, N.dtype_code(x.dtype), N.stream_ptr(dev)), "km_color_jitter_fwd_masked")
    return out, x, params, gray_sum


class _ColorJitterFunction(torch.autograd.Function):
    """``km_color_jitter_fwd(_masked)`` / ``km_color_jitter_bwd``: gradients wrt the image and the (B,4) parameter table."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, params: torch.Tensor, enable: Optional[torch.Tensor], apply: Optional[torch.Tensor], stages: tuple,
                gray_ws: Optional[torch.Tensor] = None):
        out, x, params, gray_sum = _color_launch(x, params, enable, apply, stages, gray_ws)
        ctx.stages = stages
        ctx.save_for_backward(x, params, gray_sum, enable, apply)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy: torch.Tensor):
        x, params, gray_sum, enable, apply = ctx.saved_tensors
        stages = ctx.stages
        B, _, H, W = x.shape
        dev = x.device
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        gsum = torch.zeros(B, device=dev, dtype=torch.float64) if CONTRAST in stages else None
        gparams = torch.zeros(B, 4, device=dev, dtype=torch.float64) if ctx.needs_input_grad[1] else None
        arr = (ctypes.c_int * max(len(stages), 1))(*stages)
        with N.device_guard(dev):
            N.check(N.lib().km_color_jitter_bwd(x.data_ptr(), gy.data_ptr(), gx.data_ptr(), params.data_ptr(), N.ptr(gray_sum), N.ptr(gsum), N.ptr(gparams),
                                                N.ptr(enable), N.ptr(apply), arr, len(stages), B, H, W, N.dtype_code(x.dtype), N.stream_ptr(dev)), "km_color_jitter_bwd")
        return (gx if ctx.needs_input_grad[0] else None), (gparams.to(params.dtype) if gparams is not None else None), None, None, None, None


def _run(image: torch.Tensor, factors: Sequence[Factor], stages: Sequence[int], enable: Optional[torch.Tensor] = None,
         apply: Optional[torch.Tensor] = None) -> torch.Tensor:
    N.require_device(image, "image")
    if not isinstance(image, torch.Tensor):
        raise TypeError(f"Input type is not a torch.Tensor. Got {type(image)}")
    if image.dim() < 3 or image.shape[-3] != 3:
        raise ValueError(f"Input size must have a shape of (*, 3, H, W). Got {image.shape}")
    if image.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise TypeError(f"color adjustments run in float32 / bfloat16 / float16 on the native path. Got {image.dtype}")
    stages = [int(s) for s in stages]
    if len(stages) > 4 or any(s not in (0, 1, 2, 3) for s in stages) or stages.count(CONTRAST) > 1:
        raise ValueError(f"`order` entries must be in 0..3 (brightness, contrast, saturation, hue), contrast at most once. Got {stages}")
    shape = image.shape
    x = image.reshape(-1, 3, shape[-2], shape[-1])
    B = x.shape[0]
    dev = x.device
    params = torch.stack([_factor_column(f, B, dev, n) for f, n in zip(factors, _NEUTRAL)], dim=1)
    if enable is not None:
        enable = N.flags(enable, dev, 4)
    if apply is not None:  # the augmentation layer's per-sample switch: samples whose entry is 0 are copied by the same launch
        apply = N.flags(apply, dev, B)
    return _ColorJitterFunction.apply(x, params, enable, apply, tuple(stages)).reshape(shape)


def color_jitter_from_table(image: torch.Tensor, params: torch.Tensor, enable: Optional[torch.Tensor], apply: Optional[torch.Tensor],
                            stages: Sequence[int], gray_ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The fused kernel on an already assembled parameter table (``km_color_params_fwd``): params (B,4) float32 - brightness,
    contrast, saturation factors and the hue shift in RADIANS; enable (4) / apply (B) uint8 device tensors or None; gray_ws: a (B,)
    float64 device workspace ALREADY ZEROED by the launch that made the table (``km_color_params_ws_fwd``), or None."""
    N.require_device(image, "image")
    if image.dim() != 4 or image.shape[1] != 3 or image.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError(f"expected a (B,3,H,W) float32 / bfloat16 / float16 image. Got {tuple(image.shape)} {image.dtype}")
    stages = tuple(int(s) for s in stages)
    if len(stages) > 4 or any(s not in (0, 1, 2, 3) for s in stages) or stages.count(CONTRAST) > 1:
        raise ValueError(f"`order` entries must be in 0..3 (brightness, contrast, saturation, hue), contrast at most once. Got {list(stages)}")
    if params.dtype != torch.float32 or tuple(params.shape) != (image.shape[0], 4):
        raise ValueError("params must be (B,4) float32")
    if gray_ws is not None and (gray_ws.dtype != torch.float64 or gray_ws.numel() != image.shape[0] or gray_ws.device != image.device or not gray_ws.is_contiguous()):
        raise ValueError("gray_ws must be a contiguous (B,) float64 tensor on the image's device")
    if not (torch.is_grad_enabled() and (image.requires_grad or params.requires_grad)):
        return _color_launch(image.detach(), params.detach(), enable, apply, stages, gray_ws)[0]  # (no autograd node to build: ~8 us of host per call)
    return _ColorJitterFunction.apply(image, params, enable, apply, stages, gray_ws)


def adjust_brightness_accumulative(image: torch.Tensor, factor: Union[float, torch.Tensor], clip_output: bool = True) -> torch.Tensor:
    """``clamp(image * factor, 0, 1)`` with a per-image factor."""
    if not clip_output:
        raise NotImplementedError("clip_output=False is not on the native path")
    return _run(image, (factor, None, None, None), (BRIGHTNESS,))


def adjust_contrast_with_mean_subtraction(image: torch.Tensor, factor: Union[float, torch.Tensor]) -> torch.Tensor:
    """``clamp(image * factor + mean * (1 - factor), 0, 1)``, mean = per-image mean of the grayscale image."""
    return _run(image, (None, factor, None, None), (CONTRAST,))


def adjust_saturation_with_gray_subtraction(image: torch.Tensor, factor: Union[float, torch.Tensor]) -> torch.Tensor:
    """``clamp((1 - factor) * gray(image) + factor * image, 0, 1)``."""
    return _run(image, (None, None, factor, None), (SATURATION,))


def adjust_hue(image: torch.Tensor, factor: Union[float, torch.Tensor]) -> torch.Tensor:
    """Shift the hue channel by ``factor`` radians (in [-pi, pi]) through HSV."""
    return _run(image, (None, None, None, factor), (HUE,))


def color_jitter(image: torch.Tensor, brightness_factor: Factor = None, contrast_factor: Factor = None, saturation_factor: Factor = None,
                 hue_factor: Factor = None, order: Optional[Sequence[int]] = None, enable: Optional[torch.Tensor] = None,
                 apply: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The ColorJitter sequence in one kernel.

    Factors follow ``ColorJitterGenerator`` (per-image tensors of shape (B,) or floats; ``None`` skips the stage):
    brightness / contrast / saturation are multiplicative, ``hue_factor`` is in turns (the module multiplies it by
    2 pi before ``adjust_hue``, color_jitter.py:147).  ``order`` is the application order as stage ids
    0 brightness, 1 contrast, 2 saturation, 3 hue (default 0,1,2,3); stages whose factor is ``None`` are dropped.
    ``enable``: optional (4,) device tensor indexed by stage id; a zero entry skips that stage - this is how the
    module's ``(factor != neutral).any()`` guards are honoured without a host synchronisation.
    ``apply``: optional (B,) device tensor, the augmentation layer's per-sample switch - a sample whose entry is zero is returned
    unchanged by the same launch (``_AugmentationBase._blend_by_prob``, kornia/augmentation/base.py:348-393, without its extra pass).
    """
    given = {BRIGHTNESS: brightness_factor, CONTRAST: contrast_factor, SATURATION: saturation_factor, HUE: hue_factor}
    order = [0, 1, 2, 3] if order is None else [int(i) for i in (order.tolist() if isinstance(order, torch.Tensor) else order)]
    stages = [s for s in order if given.get(s) is not None]
    hue = hue_factor
    if hue is not None:
        hue = hue * (2.0 * math.pi) if not isinstance(hue, torch.Tensor) else hue.float() * (2.0 * math.pi)
    return _run(image, (brightness_factor, contrast_factor, saturation_factor, hue), stages, enable, apply)
