"""The augmentation layer's callers of the hot path, parameter-driven.

The reference's ``RandomAffine`` / ``ColorJitter`` / ``RandomGaussianBlur`` (kornia/augmentation/_2d/geometric/affine.py:125-162,
_2d/intensity/color_jitter.py:126-159, _2d/intensity/gaussian_blur.py:95-114) split every call into *sample parameters*
(host-side random generators) and *apply them* (``compute_transformation`` + ``apply_transform`` + the ``batch_prob``
blend of ``_AugmentationBase.transform_inputs``, augmentation/base.py:348-393).  This module is the second half, taking the
parameter dictionaries the reference's generators produce (or a replay of them, ``AugmentationSequential(x, params=...)``):

* :func:`random_affine` - parameters (B,...) -> pixel matrix -> normalise / invert in ONE launch (``km_affine_params_chain_fwd``:
  the prologue of the warp) -> sample; under autograd the same through ``km_affine_matrix2d_fwd`` + ``warp_affine``;
* :func:`color_jitter` - the four adjustments in the sampled order, one fused kernel (+ one reduction pass for the contrast mean);
* :func:`random_gaussian_blur` - per-sample sigma -> taps (``km_gaussian_taps_fwd``, one launch) -> fused separable blur;
* the per-sample apply probability (``batch_prob``, base.py:348-393) rides INSIDE the three launches - the warp copies the samples
  that are not transformed (``km_warp2d_fwd_masked``), the colour kernel passes them through (``km_color_jitter_fwd_masked``), the
  blur gives them the identity kernel (``km_gaussian_taps_fwd``) - so ``torch.where``'s extra pass per stage does not exist;
  :func:`select_samples` (one pass that reads only the kept side, 2e instead of 3e) serves every other augmentation through
  ``patch()``; nothing of this runs when the parameters carry no draw (p = 1);
* :func:`apply_sequence` - the three stages in the order of BASELINE config 3.  It captures into a HIP graph
  (``kornia_amd.graph.capture``) when the parameters are device tensors: nothing in it synchronises.

Parameters stay in float32 whatever the image dtype (the reference rounds them to the image dtype first, which costs a
third of a pixel in bfloat16).  No host synchronisation anywhere: the apply masks are device data.
"""
from __future__ import annotations

import math
from typing import Any, Mapping, Optional, Sequence

import torch

from . import _native as N
from .enhance.adjust import color_jitter as _color_jitter, color_jitter_from_table
from .filters.filter import filter2d_separable, filter2d_separable_taps
from .filters.gaussian import gaussian_blur2d
from .geometry.transform.builders import get_affine_matrix2d, get_perspective_transform
from .geometry.transform.imgwarp import COORD_PERSPECTIVE, _warp, _warp_affine_from_chain, warp_affine, warp_perspective

__all__ = ["affine_chain", "affine_matrix", "apply_sequence", "color_jitter", "gaussian_taps", "random_affine", "random_gaussian_blur", "random_perspective",
           "select_samples"]


def _p(params: Mapping[str, Any], key: str, device) -> torch.Tensor:
    return torch.as_tensor(params[key], dtype=torch.float32).to(device=device, dtype=torch.float32)


def _apply_mask(params: Mapping[str, Any], device) -> Optional[torch.Tensor]:
    """(B,) bool ``batch_prob > 0.5`` (base.py:380), or None when the parameters carry no probability draw."""
    if "batch_prob" not in params or params["batch_prob"] is None:
        return None
    return torch.atleast_1d(torch.as_tensor(params["batch_prob"]).to(device) > 0.5)


def _prob(params: Mapping[str, Any], device, B: int) -> Optional[torch.Tensor]:
    """``batch_prob`` as a contiguous (B,) float32 device tensor for the parameter kernels (which threshold it), or None."""
    if "batch_prob" not in params or params["batch_prob"] is None:
        return None
    p = torch.as_tensor(params["batch_prob"]).to(device=device, dtype=torch.float32).reshape(-1).contiguous()
    if p.numel() != B:
        raise ValueError(f"batch_prob has {p.numel()} entries, expected the batch size {B}")
    return p


def select_samples(transformed: torch.Tensor, original: torch.Tensor, apply: Optional[torch.Tensor]) -> torch.Tensor:
    """``torch.where(apply[:, None, None, None], transformed, original)`` (base.py:348-361) as one native pass that reads only the
    side it keeps; ``apply`` (B,) bool on the device, ``None`` = every sample was transformed."""
    if apply is None:
        return transformed
    if (transformed.shape != original.shape or transformed.dtype != original.dtype or transformed.dtype not in (torch.float32, torch.bfloat16, torch.float16)
            or apply.shape[0] != transformed.shape[0] or (torch.is_grad_enabled() and (transformed.requires_grad or original.requires_grad))):
        return torch.where(apply.view(-1, *([1] * (transformed.dim() - 1))), transformed, original)
    t, o = transformed.contiguous(), original.contiguous()
    flags = apply.to(device=t.device, dtype=torch.uint8).contiguous()
    out = torch.empty_like(t)
    B = t.shape[0]
    with N.device_guard(t.device):
        N.check(N.lib().km_select_samples_fwd(t.data_ptr(), o.data_ptr(), flags.data_ptr(), out.data_ptr(), B, t.numel() // max(B, 1),
                                              N.dtype_code(t.dtype), N.stream_ptr(t.device)), "km_select_samples_fwd")
    return out


def gaussian_taps(sigma: torch.Tensor, kernel_size, apply: Optional[torch.Tensor] = None, batch_prob: Optional[torch.Tensor] = None,
                  round_to: Optional[torch.dtype] = None) -> tuple:
    """Per-sample 1-D Gaussian taps from ``sigma`` (B,2) = (sigma_y, sigma_x) - or (B,): the same sigma for both axes -:
    ``(taps_x (B,kx), taps_y (B,ky))`` in float32, one launch (``km_gaussian_taps_fwd`` / ``km_gaussian_taps_dtype_fwd``) for the ~16
    elementwise launches of the reference's two ``get_gaussian_kernel1d`` calls.
    ``apply`` (B,) bool: a sample whose entry is False gets the identity kernel (odd sizes), so the blur returns it unchanged - bit for
    bit when the image is finite (0 * inf is NaN).  ``batch_prob`` (B,) float: the same switch from the augmentation layer's draw itself
    (``> 0.5``), thresholded inside the launch.  ``round_to``: the image dtype - the taps come out rounded to it (still float32 values), what
    ``filter2d``'s cast of its kernel to the input dtype does (filter.py:126).  A sigma of 0 gives the identity kernel too (the sigma -> 0
    limit), NaN gives NaN taps."""
    ky, kx = (kernel_size, kernel_size) if isinstance(kernel_size, int) else (int(kernel_size[0]), int(kernel_size[1]))
    s = sigma.detach().to(torch.float32).contiguous()
    B = s.shape[0]
    tx = torch.empty(B, kx, device=s.device, dtype=torch.float32)
    ty = torch.empty(B, ky, device=s.device, dtype=torch.float32)
    if s.dim() == 1 or batch_prob is not None or round_to not in (None, torch.float32):
        if apply is not None:
            raise ValueError("gaussian_taps: with a (B,) sigma, a `batch_prob` or a 16-bit `round_to` the per-sample switch is `batch_prob` (the layer's draw, "
                             "thresholded inside the launch); `apply` (flags) goes with a (B,2) float32 sigma alone")
        if s.dim() not in (1, 2) or (s.dim() == 2 and s.shape[1] != 2):
            raise ValueError("gaussian_taps: sigma must be (B,) or (B,2)")
        prob = None if batch_prob is None else batch_prob.detach().to(device=s.device, dtype=torch.float32).reshape(-1).contiguous()
        if prob is not None and prob.numel() != B:
            raise ValueError(f"batch_prob has {prob.numel()} entries, expected the batch size {B}")
        with N.device_guard(s.device):
            N.check(N.lib().km_gaussian_taps_dtype_fwd(s.data_ptr(), int(s.dim() == 2), N.ptr(prob), tx.data_ptr(), ty.data_ptr(), B, kx, ky,
                                                       N.dtype_code(round_to or torch.float32), N.stream_ptr(s.device)), "km_gaussian_taps_dtype_fwd")
        return tx, ty
    flags = None if apply is None else N.flags(apply, s.device, B)
    with N.device_guard(s.device):
        N.check(N.lib().km_gaussian_taps_fwd(s.data_ptr(), N.ptr(flags), tx.data_ptr(), ty.data_ptr(), B, kx, ky, N.stream_ptr(s.device)), "km_gaussian_taps_fwd")
    return tx, ty


def affine_matrix(params: Mapping[str, Any], device) -> torch.Tensor:
    """RandomAffine.compute_transformation (affine.py:125-141): (B,3,3) float32 pixel matrix from the sampled
    ``translations, center, scale, angle, shear_x, shear_y`` (shears in degrees)."""
    d2r = math.pi / 180.0
    return get_affine_matrix2d(_p(params, "translations", device), _p(params, "center", device), _p(params, "scale", device),
                               _p(params, "angle", device), _p(params, "shear_x", device) * d2r, _p(params, "shear_y", device) * d2r)


def affine_chain(params: Mapping[str, Any], device, height: int, width: int, with_matrix: bool = False):
    """The sampled parameters -> ``(m, M, apply)`` in ONE launch (``km_affine_params_chain_fwd`` = compute_transformation +
    warp_affine's normalise / invert chain + the ``batch_prob > 0.5`` switch): m (B,9) float32, the normalised dst->src matrix the
    warp kernel reads for a same-size warp of a (height, width) image; M (B,3,3) the pixel matrix (the module's
    ``transform_matrix``) when ``with_matrix``, else None; apply (B) uint8 when the parameters carry a probability draw, else None."""
    device = torch.device(device)
    t, c, sc, ang, sx, sy = (_p(params, k, device).contiguous() for k in ("translations", "center", "scale", "angle", "shear_x", "shear_y"))
    B = ang.numel()
    if t.shape != (B, 2) or c.shape != (B, 2) or sc.shape != (B, 2) or sx.numel() != B or sy.numel() != B:
        raise ValueError("translations / center / scale must be (B,2) and angle / shear_x / shear_y (B,)")
    prob = _prob(params, device, B)
    m = torch.empty(B, 9, device=device, dtype=torch.float32)
    M = torch.empty(B, 3, 3, device=device, dtype=torch.float32) if with_matrix else None
    apply = torch.empty(B, device=device, dtype=torch.uint8) if prob is not None else None
    with N.device_guard(device):
        N.check(N.lib().km_affine_params_chain_fwd(t.data_ptr(), c.data_ptr(), sc.data_ptr(), ang.data_ptr(), sx.data_ptr(), sy.data_ptr(), N.ptr(prob),
                                                   N.ptr(M), m.data_ptr(), N.ptr(apply), B, int(height), int(width), int(height), int(width),
                                                   N.stream_ptr(device)), "km_affine_params_chain_fwd")
    return m, M, apply


def random_affine(input: torch.Tensor, params: Mapping[str, Any], resample: str = "bilinear", align_corners: bool = False,
                  padding_mode: str = "zeros", fill_value: Optional[torch.Tensor] = None) -> torch.Tensor:
    """RandomAffine.apply_transform + the batch_prob blend (affine.py:143-162, base.py:380-393)."""
    N.require_device(input, "input")
    if input.dim() == 4 and input.dtype in (torch.float32, torch.bfloat16, torch.float16) and not (torch.is_grad_enabled() and input.requires_grad):
        # parameters -> normalised inverse matrix (+ the per-sample switch) in one launch, the switch applied inside the warp's own launch
        if padding_mode == "fill" and fill_value is None:
            fill_value = torch.zeros(input.shape[1], device=input.device, dtype=input.dtype)
        m, _, apply = affine_chain(params, input.device, input.shape[-2], input.shape[-1])
        return _warp_affine_from_chain(input, m, resample, padding_mode, align_corners, fill_value, apply)
    mask = _apply_mask(params, input.device)
    M = affine_matrix(params, input.device)
    size = (input.shape[-2], input.shape[-1])
    out = warp_affine(input, M[:, :2, :], size, resample, padding_mode, align_corners, fill_value)
    return select_samples(out, input, mask)


def random_perspective(input: torch.Tensor, params: Mapping[str, Any], resample: str = "bilinear", align_corners: bool = False) -> torch.Tensor:
    """RandomPerspective.compute_transformation + apply_transform + the batch_prob blend (perspective.py:92-115, base.py:380-393):
    ``start_points`` / ``end_points`` (B,4,2) -> homography (``km_perspective_transform_fwd``, one launch) -> normalise / invert (one
    launch) -> warp with the per-sample switch inside its launch (``km_warp2d_fwd_masked``: samples whose draw failed are copied)."""
    N.require_device(input, "input")
    dev = input.device
    M = get_perspective_transform(_p(params, "start_points", dev), _p(params, "end_points", dev))
    size = (input.shape[-2], input.shape[-1])
    mask = _apply_mask(params, dev)
    if mask is not None and input.dim() == 4 and mask.numel() == input.shape[0] and not (torch.is_grad_enabled() and input.requires_grad):
        return _warp(input, M, size, COORD_PERSPECTIVE, 1, resample, "zeros", align_corners, torch.zeros(3), mask)
    return select_samples(warp_perspective(input, M, size, resample, "zeros", align_corners), input, mask)


def color_jitter(input: torch.Tensor, params: Mapping[str, Any], order: Optional[Sequence[int]] = None) -> torch.Tensor:
    """ColorJitter.apply_transform (color_jitter.py:126-159): brightness / contrast / saturation / hue in ``params['order']``
    (or the module's fixed ``order``), each stage skipped when its factors are all neutral."""
    N.require_device(input, "input")
    dev = input.device
    bf, cf, sf, hf = (_p(params, k, dev) for k in ("brightness_factor", "contrast_factor", "saturation_factor", "hue_factor"))
    if order is None:
        order = torch.as_tensor(params["order"]).tolist()  # sampled on the host by the reference's generator
    order = [int(i) for i in order]
    B = input.shape[0] if input.dim() == 4 else -1
    if (input.dim() == 4 and input.shape[1] == 3 and all(f.dim() == 1 and f.numel() == B for f in (bf, cf, sf, hf))
            and not (torch.is_grad_enabled() and any(f.requires_grad for f in (bf, cf, sf, hf)))):
        # factor table, stage switches and the per-sample switch in one launch (km_color_params_fwd), then the fused kernel
        prob = _prob(params, dev, B)
        table = torch.empty(B, 4, device=dev, dtype=torch.float32)
        enable = torch.empty(4, device=dev, dtype=torch.uint8)
        apply = torch.empty(B, device=dev, dtype=torch.uint8) if prob is not None else None
        bf, cf, sf, hf = (f.contiguous() for f in (bf, cf, sf, hf))
        gray_ws = torch.empty(B, device=dev, dtype=torch.float64)  # the contrast stage's accumulators: zeroed by the same launch
        with N.device_guard(dev):
            N.check(N.lib().km_color_params_ws_fwd(bf.data_ptr(), cf.data_ptr(), sf.data_ptr(), hf.data_ptr(), N.ptr(prob), table.data_ptr(), enable.data_ptr(),
                                                   N.ptr(apply), gray_ws.data_ptr(), B, N.stream_ptr(dev)), "km_color_params_ws_fwd")
        return color_jitter_from_table(input, table, enable, apply, order, gray_ws)
    enable = torch.stack([(bf != 0).any(), (cf != 1).any(), (sf != 1).any(), (hf != 0).any()])
    return _color_jitter(input, bf, cf, sf, hf, order, enable=enable, apply=_apply_mask(params, dev))


def random_gaussian_blur(input: torch.Tensor, params: Mapping[str, Any], kernel_size=(5, 5), border_type: str = "reflect",
                         separable: bool = True) -> torch.Tensor:
    """RandomGaussianBlur.apply_transform (gaussian_blur.py:95-114): per-sample ``sigma`` (B,), same in both directions."""
    N.require_device(input, "input")
    sigma1 = _p(params, "sigma", input.device)
    if separable and input.dtype in (torch.float32, torch.bfloat16, torch.float16) and sigma1.dim() == 1:
        # taps in float32 from the float32 sigma (the reference rounds sigma to the image dtype first), rounded to the image dtype inside the
        # same launch - filter2d_separable's cast of any kernel (kornia/filters/filter.py:126) - and handed to the fused filter as they are
        ky, kx = (kernel_size, kernel_size) if isinstance(kernel_size, int) else (int(kernel_size[0]), int(kernel_size[1]))
        has_prob = "batch_prob" in params and params["batch_prob"] is not None
        if not has_prob or (kx % 2 == 1 and ky % 2 == 1):
            # the switch rides in the taps: a sample that is not blurred gets the identity kernel: 1 * x + 0 * neighbours = x bit for bit
            # for FINITE images (an inf / NaN pixel of an untouched sample spreads NaN over its neighbourhood, -0.0 comes back as +0.0:
            # patch() therefore keeps the select pass for RandomGaussianBlur; this entry function states the precondition)
            taps_x, taps_y = gaussian_taps(sigma1, kernel_size, batch_prob=torch.as_tensor(params["batch_prob"]) if has_prob else None, round_to=input.dtype)
            return filter2d_separable_taps(input, taps_x, taps_y, border_type)
        taps_x, taps_y = gaussian_taps(sigma1, kernel_size, round_to=input.dtype)
        return select_samples(filter2d_separable_taps(input, taps_x, taps_y, border_type), input, _apply_mask(params, input.device))
    sigma = sigma1.unsqueeze(-1).expand(-1, 2)
    out = gaussian_blur2d(input, kernel_size, sigma.to(input.dtype), border_type, separable)
    return select_samples(out, input, _apply_mask(params, input.device))


def apply_sequence(input: torch.Tensor, affine: Mapping[str, Any], jitter: Mapping[str, Any], blur: Mapping[str, Any],
                   kernel_size=(5, 5)) -> torch.Tensor:
    """BASELINE config 3: RandomAffine -> ColorJitter -> RandomGaussianBlur with replayed parameters."""
    return random_gaussian_blur(color_jitter(random_affine(input, affine), jitter), blur, kernel_size)
