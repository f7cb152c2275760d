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

__all__ = ["AugmentationSequential", "ColorJitter", "ParamItem", "RandomAffine", "RandomGaussianBlur", "affine_chain", "affine_matrix", "apply_sequence",
           "color_jitter", "gaussian_taps", "random_affine", "random_gaussian_blur", "random_perspective", "select_samples"]


def _p(params: Mapping[str, Any], key: str, device) -> torch.Tensor:
    v = params[key]
    if type(v) is torch.Tensor and v.dtype is torch.float32 and v.device == device:  # (the modules' own device views: nothing to convert)
        return v
    return torch.as_tensor(v, dtype=torch.float32).to(device=device, dtype=torch.float32)


def _apply_mask(params: Mapping[str, Any], device) -> Optional[torch.Tensor]:
    """(B,) bool ``batch_prob > 0.5`` (base.py:380), or None when the parameters carry no probability draw."""
    if "batch_prob" not in params or params["batch_prob"] is None:
        return None
    return torch.atleast_1d(torch.as_tensor(params["batch_prob"]).to(device) > 0.5)


def _prob(params: Mapping[str, Any], device, B: int) -> Optional[torch.Tensor]:
    """``batch_prob`` as a contiguous (B,) float32 device tensor for the parameter kernels (which threshold it), or None."""
    if "batch_prob" not in params or params["batch_prob"] is None:
        return None
    p = params["batch_prob"]
    if not (type(p) is torch.Tensor and p.dtype is torch.float32 and p.device == device and p.dim() == 1 and p.is_contiguous()):
        p = torch.as_tensor(p).to(device=device, dtype=torch.float32).reshape(-1).contiguous()
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


# =====================================================================================================================================
# The modules (round 6): BASELINE config 3 AS IT IS WRITTEN - ``AugmentationSequential(RandomAffine(...), ColorJitter(...),
# RandomGaussianBlur(...))(x)`` - with the PARAMETER SAMPLING inside the call.
#
# The functions above take parameter dictionaries somebody else sampled; on a machine without Kornia (the GPU box of this project: the
# reference tree may not travel) that left config 3's public spelling without an implementation and its sampling - which SURVEY.md 8(f) calls
# the real wall time of the layer - outside every timing.  These classes are the missing half, written against the behaviour of
# kornia/augmentation/base.py:179-272 (``__batch_prob_generator__``, ``forward_parameters``), random_generator/_2d/affine.py:161-213,
# random_generator/_2d/color_jitter.py:97-110, random_generator/_2d/gaussian_blur.py:75-79 and container/augment.py:431-500:
#
#   * the draws come from torch's GLOBAL CPU generator in the reference's order - batch_prob first (only for 0 < p < 1), then each quantity as
#     ``low + torch.rand(n) * (high - low)`` in float32 (torch.distributions.Uniform.rsample), a degenerate range still consuming its draw,
#     ``torch.randperm(4)`` last for the colour order - so ``torch.manual_seed(s)`` followed by the same pipeline gives THE SAME parameters as
#     Kornia does (tests/golden/aug_modules.npz: the reference's ``_params`` for seeded calls, compared entry for entry);
#   * a module's draws land in ONE host buffer and cross to the device as ONE copy (the reference moves every parameter tensor on its own);
#     ``_params`` exposes host views with the reference's keys, so a replay through Kornia - or of Kornia's through this - works;
#   * the apply step is the functions above: the per-sample probability switch inside the launches, no blend pass, no host synchronisation.
# Only what config 3 spells: 4-D (or 3-D) image tensors, ``data_keys=["input"]``, no ``random_apply``; everything else raises.
from collections import namedtuple

ParamItem = namedtuple("ParamItem", ["name", "data"])  # (module name, parameter dictionary): what ``AugmentationSequential._params`` holds


def _range_pair(value, name: str, center: float, bounds, scalar_ok: bool = True) -> torch.Tensor:
    """(low, high) as a float32 tensor: a single number v means (center - v, center + v) clamped to ``bounds`` - computed in float32 tensor
    arithmetic, as the reference's ``_range_bound`` does, so that the bounds are the same bits - a pair is taken as given."""
    t = value.detach().to(torch.float32).cpu() if isinstance(value, torch.Tensor) else torch.tensor(value, dtype=torch.float32)
    if t.dim() == 0:
        if not scalar_ok or float(t) < 0:
            raise ValueError(f"If {name} is a single number, it must be non negative. Got {t}.")
        t = (t.repeat(2) * torch.tensor([-1.0, 1.0]) + center).clamp(bounds[0], bounds[1])
    if t.shape != (2,):
        raise ValueError(f"{name} must be a number or a (low, high) pair. Got {tuple(t.shape)}.")
    if not (bounds[0] <= float(t[0]) <= float(t[1]) <= bounds[1]):
        raise ValueError(f"{name} out of bounds. Expected inside {bounds} and low <= high, got {t.tolist()}.")
    return t


class _Draws:
    """The host buffer of one module's draws: ``floats`` float32 values in one (pinned) allocation, handed out as contiguous pieces."""

    def __init__(self, floats: int):
        self.buf = torch.empty(max(int(floats), 1), dtype=torch.float32, pin_memory=torch.cuda.is_available())
        self.k = 0

    def piece(self, *shape: int) -> torch.Tensor:
        n = 1
        for v in shape:
            n *= int(v)
        out = self.buf[self.k:self.k + n]
        if len(shape) > 1:
            out = out.view(*shape)
        self.k += n
        return out

    @staticmethod
    def uniforms(lo: torch.Tensor, span: torch.Tensor, B: int, same_on_batch: bool) -> torch.Tensor:
        """k consecutive draws at once, (k, B): ``lo[i] + torch.rand(B) * span[i]`` for i = 0 .. k-1 IN THAT ORDER.  ONE call of the generator
        for what the reference draws in k calls: torch's CPU generator fills a float32 tensor element by element, so ``torch.rand(k * B)`` is
        the concatenation of k ``torch.rand(B)`` (tests/test_gpu_aug_modules.py::test_one_call_of_the_generator_is_k_calls pins that) - and the
        host's share of a call is what bounds config 3 as it is written."""
        k = lo.numel()
        r = torch.rand(k * (1 if same_on_batch else B), dtype=torch.float32).view(k, -1)
        v = lo.view(k, 1) + r * span.view(k, 1)
        return v.expand(k, B) if same_on_batch else v

    @staticmethod
    def uniform(lo_hi: torch.Tensor, B: int, same_on_batch: bool, out: torch.Tensor) -> torch.Tensor:
        """``low + torch.rand(n) * (high - low)`` in float32 from the global CPU generator (torch.distributions.Uniform.rsample), one value
        repeated over the batch for ``same_on_batch``, written to ``out`` (B values, any stride)."""
        r = torch.rand(1 if same_on_batch else B, dtype=torch.float32)
        v = lo_hi[0] + r * (lo_hi[1] - lo_hi[0])
        out.copy_(v.expand(B) if same_on_batch else v)
        return out


class _RandomOp(torch.nn.Module):
    """What the three modules share: p / same_on_batch / keepdim, the probability draw, one host buffer -> one device copy, replay."""

    _FLOATS_PER_SAMPLE = 1  # float32 values the module draws / derives per sample, batch_prob included

    def __init__(self, p: float, same_on_batch: bool, keepdim: bool, p_batch: float = 1.0):
        super().__init__()
        self.p, self.p_batch, self.same_on_batch, self.keepdim = float(p), float(p_batch), bool(same_on_batch), bool(keepdim)
        # (per-call state lives in a plain dict: torch.nn.Module.__setattr__ costs ~10 us per assignment, and the host's share bounds the call)
        object.__setattr__(self, "_st", {"params": {}, "host_buf": None, "dev_buf": None})

    @property
    def _params(self) -> dict:
        return self._st["params"]

    @property
    def _host_buf(self):
        return self._st["host_buf"]

    @property
    def _dev_buf(self):
        return self._st["dev_buf"]

    @_dev_buf.setter
    def _dev_buf(self, v):
        self._st["dev_buf"] = v

    def _sample(self, d: _Draws, shape, params: dict) -> None:
        raise NotImplementedError

    def _apply(self, x: torch.Tensor, params: dict) -> torch.Tensor:
        raise NotImplementedError

    def _batch_prob(self, B: int, out: torch.Tensor) -> torch.Tensor:
        # base.py:179-215: a batch-level gate (p_batch) first, then the per-sample gates; certain outcomes consume nothing
        gate = 1.0
        if 0.0 < self.p_batch < 1.0:
            gate = float((torch.rand(1) < self.p_batch).item())
        elif self.p_batch <= 0.0:
            gate = 0.0
        if self.p >= 1.0 or self.p <= 0.0:  # (certain outcomes: one fill of the piece, no draw)
            out.fill_(gate if self.p >= 1.0 else 0.0)
            return out
        elif self.same_on_batch:
            e = (torch.rand(1) < self.p).to(torch.float32).expand(B)
        else:
            e = (torch.rand(B) < self.p).to(torch.float32)
        out.copy_(e * gate)
        return out

    def forward_parameters(self, batch_shape, draws: Optional[_Draws] = None) -> dict:
        """Sample this call's parameters on the host (the reference's keys; every float tensor a piece of ONE buffer - the container's,
        when it hands one in: then the whole pipeline's draws cross to the device as one copy)."""
        B = int(batch_shape[0])
        d = draws if draws is not None else _Draws(self._FLOATS_PER_SAMPLE * B)
        params: dict = {"batch_prob": self._batch_prob(B, d.piece(B))}
        self._sample(d, batch_shape, params)
        shp = tuple(int(v) for v in batch_shape)
        st = self._st.get("shape_t")
        if st is None or st[0] != shp:  # (the same small tensor for every call at this shape: torch.tensor(...) is 4 us a time)
            st = self._st["shape_t"] = (shp, torch.tensor(shp, dtype=torch.long))
        params["forward_input_shape"] = st[1]
        self._st["host_buf"], self._st["dev_buf"] = d.buf, None
        return params

    def _device_params(self, params: Mapping[str, Any], device, own: bool) -> dict:
        """The parameters as the apply step wants them: this module's own sample crosses to the device as ONE copy of the draw buffer (the
        float tensors of ``params`` are pieces of it); foreign parameters (a replay) go tensor by tensor, as the functions above take them."""
        out = dict(params)
        buf = self._host_buf if own else None
        dev = self._dev_buf if own else None
        if buf is None:
            # a replay: host float tensors that are pieces of ONE allocation (this package's own `_params` are) cross as one copy too
            fl = [v for v in params.values() if isinstance(v, torch.Tensor) and v.dtype == torch.float32 and v.numel() and v.device.type == "cpu" and v.is_contiguous()]
            if len(fl) > 1 and all(v.untyped_storage().data_ptr() == fl[0].untyped_storage().data_ptr() for v in fl):
                buf = torch.empty(0, dtype=torch.float32).set_(fl[0].untyped_storage())
                dev = None
        if buf is not None:
            if dev is None:
                dev = buf.to(device, non_blocking=True)
            base = buf.data_ptr()
            # where the float tensors of `params` sit in the buffer: the same for every call of this module at this batch size (the pieces are
            # handed out in a fixed order) - found once, checked by (key, address offset) since, and each device-side view made by ONE as_strided
            # (a slice + a view per tensor were 30 us of a call whose host share bounds it: profiles/r06/run20_*)
            lkey = "layout" if own else "layout_replay"
            lay = self._st.get(lkey)
            if lay is not None and (lay[0] != buf.numel() or len(lay[1]) > len(params)):
                lay = None
            if lay is not None:
                for k, off, shape, strides in lay[1]:
                    v = params.get(k)
                    if v is None or v.data_ptr() - base != 4 * off or v.shape != shape:
                        lay = None
                        break
            if lay is None:
                found = []
                for k, v in params.items():
                    if isinstance(v, torch.Tensor) and v.dtype == torch.float32 and v.numel() and v.device.type == "cpu" and v.is_contiguous():
                        off = (v.data_ptr() - base) // 4
                        if 0 <= off and off + v.numel() <= buf.numel():
                            found.append((k, off, v.shape, v.stride()))
                lay = self._st[lkey] = (buf.numel(), found)
            for k, off, shape, strides in lay[1]:
                out[k] = dev.as_strided(shape, strides, off)
        if self.p >= 1.0 and self.p_batch >= 1.0:
            out["batch_prob"] = None  # every sample is transformed: no switch in the launches at all
        return out

    def forward(self, input: torch.Tensor, params: Optional[Mapping[str, Any]] = None, _own: bool = False) -> torch.Tensor:
        N.require_device(input, "input")
        if input.dim() not in (3, 4):
            raise ValueError(f"expected a (B, C, H, W) or (C, H, W) image tensor, got {tuple(input.shape)}")
        x = input.unsqueeze(0) if input.dim() == 3 else input
        own = params is None or _own  # (_own: the container sampled these through this module's forward_parameters a moment ago)
        if params is None:
            params = self.forward_parameters(x.shape)
        elif not _own:
            self._st["host_buf"] = None
        self._st["params"] = dict(params)
        out = self._apply(x, self._device_params(self._st["params"], x.device, own))
        return out[0] if (input.dim() == 3 and self.keepdim) else out


class RandomAffine(_RandomOp):
    """``kornia.augmentation.RandomAffine`` (kornia/augmentation/_2d/geometric/affine.py:33-162) on the native path: the same constructor,
    the same parameter draws, parameters -> matrix -> normalise / invert in one launch, the warp with the probability switch inside it."""

    _FLOATS_PER_SAMPLE = 10  # batch_prob, angle, shear x / y, scale (2), translation (2), centre (2)

    def __init__(self, degrees, translate=None, scale=None, shear=None, resample="BILINEAR", same_on_batch: bool = False, align_corners: bool = False,
                 padding_mode="ZEROS", fill_value=None, p: float = 0.5, keepdim: bool = False) -> None:
        super().__init__(p, same_on_batch, keepdim)
        self.degrees = _range_pair(degrees, "degrees", 0.0, (-360.0, 360.0))
        self.translate = None
        if translate is not None:
            t = torch.as_tensor(translate, dtype=torch.float32)
            if t.shape != (2,) or not bool(((t >= 0) & (t <= 1)).all()):
                raise ValueError(f"translate must be two fractions in [0, 1]. Got {translate}.")
            self.translate = t
        self.scale = None
        if scale is not None:
            s = torch.as_tensor(scale, dtype=torch.float32)
            if s.shape not in ((2,), (4,)) or not bool((s >= 0).all()):
                raise ValueError(f"'scale' expected to be either 2 or 4 non-negative elements. Got {scale}")
            self.scale = s
        self.shear = None
        if shear is not None:
            sh = torch.as_tensor(shear, dtype=torch.float32)
            if sh.dim() == 0:
                self.shear = torch.stack([_range_pair(sh, "shear-x", 0.0, (-360.0, 360.0)), torch.zeros(2)])
            elif sh.shape == (2,):
                self.shear = torch.stack([_range_pair(sh, "shear-x", 0.0, (-360.0, 360.0)), torch.zeros(2)])
            elif sh.shape == (4,):
                self.shear = torch.stack([_range_pair(sh[:2], "shear-x", 0.0, (-360.0, 360.0)), _range_pair(sh[2:], "shear-y", 0.0, (-360.0, 360.0))])
            elif sh.shape == (2, 2):
                self.shear = sh
            else:
                raise ValueError(f"shear must be a number, a pair, four numbers or a 2 x 2 tensor. Got {shear}.")
        self.resample = str(getattr(resample, "name", resample)).lower()
        self.padding_mode = str(getattr(padding_mode, "name", padding_mode)).lower()
        self.align_corners = bool(align_corners)
        self.fill_value = fill_value
        self._lo = self._span = None

    def _ranges(self):
        """(lo, span) of the module's draws in the reference's order - angle, scale x (, scale y), translation x, y, shear x, y - as float32 vectors,
        formed once (``high - low`` in float32, as torch.distributions.Uniform does)."""
        if getattr(self, "_lo", None) is None:
            pairs = [self.degrees]
            if self.scale is not None:
                pairs.append(self.scale[:2])
                if self.scale.numel() == 4:
                    pairs.append(self.scale[2:])
            if self.translate is not None:
                pairs += [torch.stack([-self.translate[0], self.translate[0]]), torch.stack([-self.translate[1], self.translate[1]])]
            if self.shear is not None:
                pairs += [self.shear[0], self.shear[1]]
            p = torch.stack(pairs).to(torch.float32)
            self._lo, self._span = p[:, 0].contiguous(), (p[:, 1] - p[:, 0]).contiguous()
        return self._lo, self._span

    def _sample(self, d: _Draws, shape, params: dict) -> None:
        # random_generator/_2d/affine.py:161-213: angle, scale (x, then y when four numbers were given), translation x, y, shear x, y
        B, H, W = int(shape[0]), int(shape[-2]), int(shape[-1])
        angle, shx, shy = d.piece(B), d.piece(B), d.piece(B)
        scale, trans, center = d.piece(B, 2), d.piece(B, 2), d.piece(B, 2)
        lo, span = self._ranges()
        geo = self._st.get("geo")
        if geo is None or geo[2] != (H, W):  # ((W, H) and the centre (W / 2 - 0.5, H / 2 - 0.5) as float32 rows, formed once per image size)
            geo = self._st["geo"] = (torch.tensor([float(W), float(H)], dtype=torch.float32), torch.tensor([W / 2.0 - 0.5, H / 2.0 - 0.5], dtype=torch.float32), (H, W))
        v = d.uniforms(lo, span, B, self.same_on_batch)  # (k, B): every draw of the module from ONE call of the generator
        angle.copy_(v[0])
        k = 1
        if self.scale is not None:
            if self.scale.numel() == 4:
                scale.copy_(v[k:k + 2].t())
                k += 2
            else:
                scale.copy_(v[k:k + 1].t().expand(B, 2))
                k += 1
        else:
            scale.fill_(1.0)
        if self.translate is not None:
            trans.copy_(v[k:k + 2].t())
            trans.mul_(geo[0])  # (x by W, y by H: the same float32 products as two column-wise multiplications)
            k += 2
        else:
            trans.zero_()
        center.copy_(geo[1])
        if self.shear is not None:
            shx.copy_(v[k])
            shy.copy_(v[k + 1])
        else:
            shx.zero_()
            shy.zero_()
        params.update(translations=trans, center=center, scale=scale, angle=angle, shear_x=shx, shear_y=shy)

    def _apply(self, x: torch.Tensor, params: dict) -> torch.Tensor:
        fill = self.fill_value
        if fill is not None and not isinstance(fill, torch.Tensor):
            fill = torch.full((x.shape[1],), float(fill))
        return random_affine(x, params, self.resample, self.align_corners, self.padding_mode, fill)

    @property
    def transform_matrix(self) -> Optional[torch.Tensor]:
        """(B,3,3) pixel matrix of the last call (identity for the samples whose probability draw failed), computed on demand."""
        if not self._params:
            return None
        shp = self._params["forward_input_shape"].tolist()
        dev = "cuda"
        _, M, apply = affine_chain(self._params, dev, shp[-2], shp[-1], with_matrix=True)
        if apply is not None:
            M = torch.where(apply.bool().view(-1, 1, 1), M, torch.eye(3, device=M.device).expand_as(M))
        return M


class ColorJitter(_RandomOp):
    """``kornia.augmentation.ColorJitter`` (kornia/augmentation/_2d/intensity/color_jitter.py:34-159): brightness, contrast, saturation and hue
    factors per sample, applied in a random (or the given) order by ONE fused kernel (+ the reduction pass of the contrast mean)."""

    _FLOATS_PER_SAMPLE = 5  # batch_prob, brightness, contrast, hue, saturation

    def __init__(self, brightness=0.0, contrast=0.0, saturation=0.0, hue=0.0, same_on_batch: bool = False, p: float = 1.0, keepdim: bool = False,
                 order: Optional[Sequence[int]] = None) -> None:
        super().__init__(p, same_on_batch, keepdim)
        inf = float("inf")
        self.brightness = _range_pair(brightness, "brightness", 1.0, (0.0, inf))
        self.contrast = _range_pair(contrast, "contrast", 1.0, (0.0, inf))
        self.saturation = _range_pair(saturation, "saturation", 1.0, (0.0, inf))
        self.hue = _range_pair(hue, "hue", 0.0, (-0.5, 0.5))
        if order is not None:
            order = tuple(int(i) for i in order)
            if not set(order) <= {0, 1, 2, 3}:
                raise ValueError(f"`order` entries must be in 0..3 (brightness, contrast, saturation, hue). Got {order}")
        self._fixed_order = order

    def _sample(self, d: _Draws, shape, params: dict) -> None:
        # random_generator/_2d/color_jitter.py:97-110: brightness, contrast, HUE, saturation, then the order of the four stages
        B = int(shape[0])
        if getattr(self, "_lo", None) is None:
            p = torch.stack([self.brightness, self.contrast, self.hue, self.saturation])
            self._lo, self._span = p[:, 0].contiguous(), (p[:, 1] - p[:, 0]).contiguous()
        out = d.piece(4, B)
        out.copy_(d.uniforms(self._lo, self._span, B, self.same_on_batch))
        params["brightness_factor"], params["contrast_factor"], params["hue_factor"], params["saturation_factor"] = out[0], out[1], out[2], out[3]
        params["order"] = torch.randperm(4, dtype=torch.long)

    def _apply(self, x: torch.Tensor, params: dict) -> torch.Tensor:
        return color_jitter(x, params, self._fixed_order)


class RandomGaussianBlur(_RandomOp):
    """``kornia.augmentation.RandomGaussianBlur`` (kornia/augmentation/_2d/intensity/gaussian_blur.py:31-114): one sigma per sample, the
    taps of every sample in one launch, the fused separable blur."""

    _FLOATS_PER_SAMPLE = 2  # batch_prob, sigma

    def __init__(self, kernel_size, sigma, border_type: str = "reflect", separable: bool = True, same_on_batch: bool = False, p: float = 0.5,
                 keepdim: bool = False) -> None:
        super().__init__(p, same_on_batch, keepdim)
        self.kernel_size = (kernel_size, kernel_size) if isinstance(kernel_size, int) else (int(kernel_size[0]), int(kernel_size[1]))
        s = torch.as_tensor(sigma, dtype=torch.float32)
        if s.shape != (2,):
            raise TypeError(f"sigma must be a (min, max) pair. Got {sigma}.")
        if float(s[1]) < float(s[0]):
            raise TypeError(f"sigma_max should be higher than sigma_min: {sigma} passed.")
        if float(s[0]) < 0:
            raise ValueError(f"sigma out of bounds. Expected inside (0, inf), got {s.tolist()}.")
        self.sigma = s
        self.border_type = str(getattr(border_type, "name", border_type)).lower()
        self.separable = bool(separable)

    def _sample(self, d: _Draws, shape, params: dict) -> None:
        params["sigma"] = d.uniform(self.sigma, int(shape[0]), self.same_on_batch, d.piece(int(shape[0])))

    def _apply(self, x: torch.Tensor, params: dict) -> torch.Tensor:
        return random_gaussian_blur(x, params, self.kernel_size, self.border_type, self.separable)


class AugmentationSequential(torch.nn.Module):
    """``kornia.augmentation.AugmentationSequential`` for image tensors (kornia/augmentation/container/augment.py:431-500): every child samples
    its parameters and transforms the previous child's output; ``params=`` replays a list of ``ParamItem(name, data)`` (this container's
    ``_params`` - or Kornia's own: the names and keys are the reference's)."""

    def __init__(self, *args: torch.nn.Module, data_keys=("input",), same_on_batch: Optional[bool] = None, keepdim: Optional[bool] = None,
                 random_apply=False, random_apply_weights=None, transformation_matrix_mode: str = "silent", extra_args=None) -> None:
        super().__init__()
        keys = [str(getattr(k, "name", k)).lower() for k in (data_keys or ("input",))]
        if keys not in (["input"], ["image"], ["0"]):
            raise NotImplementedError(f"only image tensors are supported here (data_keys={list(data_keys)}); use Kornia's container with kornia_amd.patch() for masks / boxes / keypoints")
        if random_apply not in (False, None) or random_apply_weights is not None:
            raise NotImplementedError("random_apply is not supported here; use Kornia's container with kornia_amd.patch()")
        for i, m in enumerate(args):
            if not isinstance(m, _RandomOp):
                raise NotImplementedError(f"child {i} ({type(m).__name__}) is not one of this package's modules (RandomAffine, ColorJitter, RandomGaussianBlur)")
            if same_on_batch is not None:
                m.same_on_batch = bool(same_on_batch)
            if keepdim is not None:
                m.keepdim = bool(keepdim)
            self.add_module(f"{type(m).__name__}_{i}", m)
        # (per-call state as plain attributes: torch.nn.Module.__setattr__ costs ~5 us per assignment; the children are fixed after construction)
        object.__setattr__(self, "_params", [])
        object.__setattr__(self, "_draws", None)
        object.__setattr__(self, "_kids", list(self.named_children()))

    def forward_parameters(self, batch_shape) -> list:
        """One ``ParamItem`` per child, sampled in order (a same-size pipeline: every child sees the input's shape); the draws of ALL children
        in one host buffer."""
        B = int(batch_shape[0])
        children = self._kids
        d = _Draws(sum(m._FLOATS_PER_SAMPLE for _, m in children) * B)
        items = [ParamItem(name, m.forward_parameters(batch_shape, d)) for name, m in children]
        object.__setattr__(self, "_draws", d)
        return items

    def forward(self, input: torch.Tensor, params: Optional[Sequence[ParamItem]] = None) -> torch.Tensor:
        children = self._kids
        if params is not None and len(params) != len(children):
            raise ValueError(f"{len(params)} parameter items for {len(children)} children")
        own = params is None
        if own:
            shape = input.shape if input.dim() == 4 else (1, *input.shape)
            params = self.forward_parameters(shape)
            dev = self._draws.buf.to(input.device, non_blocking=True)  # the whole pipeline's draws: ONE copy
            for _, m in children:
                m._dev_buf = dev
        else:
            # a replay of parameters whose float tensors are pieces of ONE host allocation (this container's own `_params` are): one copy for the
            # whole pipeline here too, handed to the children the way their own draws are
            st_ptr, store = None, None
            for it in params:
                for v in it.data.values():
                    if type(v) is torch.Tensor and v.dtype is torch.float32 and v.device.type == "cpu" and v.numel():
                        sp = v.untyped_storage().data_ptr()
                        if st_ptr is None:
                            st_ptr, store = sp, v.untyped_storage()
                        elif sp != st_ptr:
                            st_ptr = 0
                            break
                if st_ptr == 0:
                    break
            if st_ptr:
                buf = torch.empty(0, dtype=torch.float32).set_(store)
                dev = buf.to(input.device, non_blocking=True)
                for _, m in children:
                    m._st["host_buf"] = buf
                    m._dev_buf = dev
                own = True
        out, used = input, []
        for i, (name, m) in enumerate(children):
            out = m(out, params[i].data, _own=own)
            used.append(ParamItem(name, m._params))
        object.__setattr__(self, "_params", used)
        return out
