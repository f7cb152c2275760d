"""The augmentation layer's callers of the hot path, parameter-driven.

The reference's ``RandomAffine`` / ``ColorJitter`` / ``RandomGaussianBlur`` (kornia/augmentation/_2d/geometric/affine.py:125-162,
_2d/intensity/color_jitter.py:126-159, _2d/intensity/gaussian_blur.py:95-114) split every call into *sample parameters*
(host-side random generators) and *apply them* (``compute_transformation`` + ``apply_transform`` + the ``batch_prob``
blend of ``_AugmentationBase.transform_inputs``, augmentation/base.py:348-393).  This module is the second half, taking the
parameter dictionaries the reference's generators produce (or a replay of them, ``AugmentationSequential(x, params=...)``):

* :func:`random_affine` - parameters (B,...) -> pixel matrix (``km_affine_matrix2d_fwd``) -> normalise / invert -> sample,
  with the per-sample apply mask folded into the matrix (a skipped sample gets the identity, which the sampler reproduces
  bit for bit when source and destination sizes agree);
* :func:`color_jitter` - the four adjustments in the sampled order, one fused kernel;
* :func:`random_gaussian_blur` - per-sample sigma -> taps -> separable blur;
* :func:`apply_sequence` - the three in the order of BASELINE config 3.

Parameters stay in float32 whatever the image dtype (the reference rounds them to the image dtype first, which costs a
third of a pixel in bfloat16).  No host synchronisation anywhere: the apply masks are device data.
"""
from __future__ import annotations

import math
from typing import Any, Mapping, Optional, Sequence

import torch

from . import _native as N
from .enhance.adjust import color_jitter as _color_jitter
from .filters.gaussian import gaussian_blur2d
from .geometry.transform.builders import get_affine_matrix2d
from .geometry.transform.imgwarp import warp_affine

__all__ = ["apply_sequence", "color_jitter", "random_affine", "random_gaussian_blur"]


def _p(params: Mapping[str, Any], key: str, device) -> torch.Tensor:
    return torch.as_tensor(params[key], dtype=torch.float32).to(device=device, dtype=torch.float32)


def _apply_mask(params: Mapping[str, Any], device) -> Optional[torch.Tensor]:
    """(B,) bool ``batch_prob > 0.5`` (base.py:380), or None when the parameters carry no probability draw."""
    if "batch_prob" not in params or params["batch_prob"] is None:
        return None
    return torch.atleast_1d(torch.as_tensor(params["batch_prob"]).to(device) > 0.5)


def affine_matrix(params: Mapping[str, Any], device) -> torch.Tensor:
    """RandomAffine.compute_transformation (affine.py:125-141): (B,3,3) float32 pixel matrix from the sampled
    ``translations, center, scale, angle, shear_x, shear_y`` (shears in degrees)."""
    d2r = math.pi / 180.0
    return get_affine_matrix2d(_p(params, "translations", device), _p(params, "center", device), _p(params, "scale", device),
                               _p(params, "angle", device), _p(params, "shear_x", device) * d2r, _p(params, "shear_y", device) * d2r)


def random_affine(input: torch.Tensor, params: Mapping[str, Any], resample: str = "bilinear", align_corners: bool = False,
                  padding_mode: str = "zeros", fill_value: Optional[torch.Tensor] = None) -> torch.Tensor:
    """RandomAffine.apply_transform + the batch_prob blend (affine.py:143-162, base.py:380-393)."""
    N.require_device(input, "input")
    M = affine_matrix(params, input.device)
    mask = _apply_mask(params, input.device)
    out = warp_affine(input, M[:, :2, :], (input.shape[-2], input.shape[-1]), resample, padding_mode, align_corners, fill_value)
    if mask is None:
        return out
    return torch.where(mask.view(-1, 1, 1, 1), out, input)


def color_jitter(input: torch.Tensor, params: Mapping[str, Any], order: Optional[Sequence[int]] = None) -> torch.Tensor:
    """ColorJitter.apply_transform (color_jitter.py:126-159): brightness / contrast / saturation / hue in ``params['order']``
    (or the module's fixed ``order``), each stage skipped when its factors are all neutral."""
    N.require_device(input, "input")
    dev = input.device
    bf, cf, sf, hf = (_p(params, k, dev) for k in ("brightness_factor", "contrast_factor", "saturation_factor", "hue_factor"))
    if order is None:
        order = torch.as_tensor(params["order"]).tolist()  # sampled on the host by the reference's generator
    enable = torch.stack([(bf != 0).any(), (cf != 1).any(), (sf != 1).any(), (hf != 0).any()])
    out = _color_jitter(input, bf, cf, sf, hf, [int(i) for i in order], enable=enable)
    mask = _apply_mask(params, dev)
    return out if mask is None else torch.where(mask.view(-1, 1, 1, 1), out, input)


def random_gaussian_blur(input: torch.Tensor, params: Mapping[str, Any], kernel_size=(5, 5), border_type: str = "reflect",
                         separable: bool = True) -> torch.Tensor:
    """RandomGaussianBlur.apply_transform (gaussian_blur.py:95-114): per-sample ``sigma`` (B,), same in both directions."""
    N.require_device(input, "input")
    sigma = _p(params, "sigma", input.device).unsqueeze(-1).expand(-1, 2)
    out = gaussian_blur2d(input, kernel_size, sigma, border_type, separable)
    mask = _apply_mask(params, input.device)
    return out if mask is None else torch.where(mask.view(-1, 1, 1, 1), out, input)


def apply_sequence(input: torch.Tensor, affine: Mapping[str, Any], jitter: Mapping[str, Any], blur: Mapping[str, Any],
                   kernel_size=(5, 5)) -> torch.Tensor:
    """BASELINE config 3: RandomAffine -> ColorJitter -> RandomGaussianBlur with replayed parameters."""
    return random_gaussian_blur(color_jitter(random_affine(input, affine), jitter), blur, kernel_size)
