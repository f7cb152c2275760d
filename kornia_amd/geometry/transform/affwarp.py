"""affine / rotate / translate / scale / shear on the native warp (callers of warp_affine).

Reference behaviour mirrored: kornia/geometry/transform/affwarp.py:136-193 (affine), :257-327 (rotate), :401-454
(translate), :455-521 (scale), :522-575 (shear) - same argument meaning, defaults (``align_corners=True`` except
``shear``: False), error types, and the same matrices: rotation / scaling about the image centre ``((W-1)/2, (H-1)/2)``
through ``get_rotation_matrix2d``, translation and shear as ``I + offset``.  The matrix step is a few O(B) tensor
expressions, the resampling is ``warp_affine`` (one chain launch + one warp launch).
"""
from __future__ import annotations

from typing import Optional

import torch

from .builders import get_rotation_matrix2d
from .imgwarp import warp_affine

__all__ = ["affine", "rotate", "scale", "shear", "translate"]


def _center(tensor: torch.Tensor) -> torch.Tensor:
    if not 2 <= tensor.dim() <= 4:
        raise AssertionError(f"Must be a 3D tensor as HW, CHW and BCHW. Got {tensor.shape}.")
    h, w = tensor.shape[-2:]
    return torch.tensor([float(w - 1) / 2, float(h - 1) / 2], device=tensor.device, dtype=tensor.dtype)


def _batch(tensor: torch.Tensor) -> int:
    return tensor.shape[0] if tensor.dim() == 4 else 1


def _check_image(tensor, what="tensor"):
    if not isinstance(tensor, torch.Tensor):
        raise TypeError(f"Input {what} type is not a torch.Tensor. Got {type(tensor)}")
    if tensor.dim() not in (3, 4):
        raise ValueError(f"Invalid tensor shape, we expect CxHxW or BxCxHxW. Got: {tensor.shape}")


def affine(tensor: torch.Tensor, matrix: torch.Tensor, mode: str = "bilinear", padding_mode: str = "zeros",
           align_corners: bool = True) -> torch.Tensor:
    """Warp ``(C,H,W)`` / ``(B,C,H,W)`` by the source->destination pixel affine ``matrix`` (B,2,3); output keeps the input size.
    A single image is broadcast over a batch of matrices and vice versa."""
    unbatched = tensor.dim() == 3
    if unbatched:
        tensor = tensor.unsqueeze(0)
    if tensor.shape[0] == 1 and matrix.shape[0] != 1:
        tensor = tensor.expand(matrix.shape[0], -1, -1, -1)
    matrix = matrix.expand(tensor.shape[0], -1, -1)
    out = warp_affine(tensor, matrix, (tensor.shape[-2], tensor.shape[-1]), mode, padding_mode, align_corners)
    return out.squeeze(0) if unbatched else out


def rotate(tensor: torch.Tensor, angle: torch.Tensor, center: Optional[torch.Tensor] = None, mode: str = "bilinear",
           padding_mode: str = "zeros", align_corners: bool = True) -> torch.Tensor:
    """Rotate anti-clockwise by ``angle`` degrees about ``center`` (default: the image centre)."""
    _check_image(tensor)
    if not isinstance(angle, torch.Tensor):
        raise TypeError(f"Input angle type is not a torch.Tensor. Got {type(angle)}")
    if center is not None and not isinstance(center, torch.Tensor):
        raise TypeError(f"Input center type is not a torch.Tensor. Got {type(center)}")
    B = _batch(tensor)
    if center is None:
        center = _center(tensor)
    angle = angle.expand(B)
    center = center.expand(B, -1)
    M = get_rotation_matrix2d(center, angle, torch.ones_like(center))
    return affine(tensor, M, mode, padding_mode, align_corners)


def translate(tensor: torch.Tensor, translation: torch.Tensor, mode: str = "bilinear", padding_mode: str = "zeros",
              align_corners: bool = True) -> torch.Tensor:
    """Shift by ``translation`` (B,2) = (dx, dy) pixels."""
    _check_image(tensor)
    if not isinstance(translation, torch.Tensor):
        raise TypeError(f"Input translation type is not a torch.Tensor. Got {type(translation)}")
    t = translation.reshape(-1, 2)
    M = torch.zeros(t.shape[0], 2, 3, device=t.device, dtype=t.dtype)
    M[:, 0, 0] = 1
    M[:, 1, 1] = 1
    M = torch.cat([M[:, :, :2], M[:, :, 2:] + t[:, :, None]], dim=-1)
    return affine(tensor, M, mode, padding_mode, align_corners)


def scale(tensor: torch.Tensor, scale_factor: torch.Tensor, center: Optional[torch.Tensor] = None, mode: str = "bilinear",
          padding_mode: str = "zeros", align_corners: bool = True) -> torch.Tensor:
    """Scale by ``scale_factor`` ((B,) isotropic or (B,2) = (sx, sy)) about ``center`` (default: the image centre)."""
    if not isinstance(tensor, torch.Tensor):
        raise TypeError(f"Input tensor type is not a torch.Tensor. Got {type(tensor)}")
    if not isinstance(scale_factor, torch.Tensor):
        raise TypeError(f"Input scale_factor type is not a torch.Tensor. Got {type(scale_factor)}")
    if scale_factor.dim() == 1:
        scale_factor = scale_factor.repeat(1, 2)  # the reference's isotropic convention: (1, 2 * n) -> expanded below
    B = _batch(tensor)
    if center is None:
        center = _center(tensor)
    center = center.expand(B, -1)
    scale_factor = scale_factor.expand(B, 2)
    M = get_rotation_matrix2d(center, torch.zeros(B, device=scale_factor.device, dtype=scale_factor.dtype), scale_factor)
    return affine(tensor, M, mode, padding_mode, align_corners)


def shear(tensor: torch.Tensor, shear: torch.Tensor, mode: str = "bilinear", padding_mode: str = "zeros",
          align_corners: bool = False) -> torch.Tensor:
    """Shear by ``shear`` (B,2) = (shx, shy): ``x' = x + shx * y``, ``y' = shy * x + y``."""
    _check_image(tensor)
    if not isinstance(shear, torch.Tensor):
        raise TypeError(f"Input shear type is not a torch.Tensor. Got {type(shear)}")
    s = shear.reshape(-1, 2)
    one, zero = torch.ones_like(s[:, 0]), torch.zeros_like(s[:, 0])
    M = torch.stack([one, s[:, 0], zero, s[:, 1], one, zero], dim=-1).reshape(-1, 2, 3)
    return affine(tensor, M, mode, padding_mode, align_corners)
