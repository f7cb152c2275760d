"""Pixel <-> normalised coordinate transforms of the warp path.

Mirrors kornia/geometry/conversions.py: normal_transform_pixel (:1729-1763),
normalize_homography (:1691-1726), convert_affinematrix_to_homography (:342-378),
convert_points_to/from_homogeneous (:247-339).  ``normalize_homography`` runs the whole
``N_dst @ (M @ inv(N_src))`` chain as ONE HIP launch (csrc/km_chain.hip) instead of ~15 torch ops.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from .. import _native as N

__all__ = [
    "convert_affinematrix_to_homography",
    "convert_points_from_homogeneous",
    "convert_points_to_homogeneous",
    "normal_transform_pixel",
    "normalize_homography",
]


def normal_transform_pixel(
    height: int,
    width: int,
    eps: float = 1e-14,
    device: Optional[torch.device] = None,
    dtype: Optional[torch.dtype] = None,
) -> torch.Tensor:
    """(1,3,3) matrix taking pixel coordinates to [-1, 1] (host-side constant, no kernel)."""
    wd = eps if width == 1 else width - 1.0
    hd = eps if height == 1 else height - 1.0
    m = torch.tensor([[2.0 / wd, 0.0, -1.0], [0.0, 2.0 / hd, -1.0], [0.0, 0.0, 1.0]], device=device, dtype=dtype)
    return m.unsqueeze(0)


class _ChainFunction(torch.autograd.Function):
    """(B,rows,3) pixel matrix -> (B,3,3) normalised matrix (`want_inverse`: its inverse)."""

    @staticmethod
    def forward(ctx, M: torch.Tensor, src_size, dst_size, want_inverse: bool):
        N.require_device(M, "M")
        cdt = N.compute_dtype(M.dtype)
        Mc = M.detach().to(cdt).contiguous()
        B, rows = Mc.shape[0], Mc.shape[1]
        out = torch.empty(B, 9, device=M.device, dtype=cdt)
        with N.device_guard(M.device):
            rc = N.lib().km_homography_chain_fwd(
                Mc.data_ptr(), rows, None if want_inverse else out.data_ptr(), out.data_ptr() if want_inverse else None,
                B, int(src_size[0]), int(src_size[1]), int(dst_size[0]), int(dst_size[1]), N.dtype_code(cdt),
                N.stream_ptr(M.device))
        N.check(rc, "km_homography_chain_fwd")
        ctx.save_for_backward(Mc)
        ctx.cfg = (src_size, dst_size, want_inverse, M.dtype)
        return out.view(B, 3, 3).to(M.dtype)

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        (Mc,) = ctx.saved_tensors
        src_size, dst_size, want_inverse, in_dtype = ctx.cfg
        if not want_inverse:
            # A = Nd M Nsi is linear in M: gM = Nd^T gA Nsi^T (tiny; PyTorch ops are fine here)
            cdt = Mc.dtype
            Ns = normal_transform_pixel(src_size[0], src_size[1]).to(device=Mc.device, dtype=cdt)
            Nd = normal_transform_pixel(dst_size[0], dst_size[1]).to(device=Mc.device, dtype=cdt)
            gM = Nd.transpose(-1, -2) @ g.to(cdt) @ torch.linalg.inv(Ns).transpose(-1, -2)
            return gM[:, : Mc.shape[1], :].to(in_dtype), None, None, None
        B, rows = Mc.shape[0], Mc.shape[1]
        gm = g.detach().to(torch.float64).contiguous().view(B, 9)
        gM = torch.empty_like(Mc)
        with N.device_guard(Mc.device):
            rc = N.lib().km_homography_chain_bwd(
                Mc.data_ptr(), rows, gm.data_ptr(), gM.data_ptr(), B, int(src_size[0]), int(src_size[1]),
                int(dst_size[0]), int(dst_size[1]), N.dtype_code(Mc.dtype), N.stream_ptr(Mc.device))
        N.check(rc, "km_homography_chain_bwd")
        return gM.to(in_dtype), None, None, None


def normalize_homography(
    dst_pix_trans_src_pix: torch.Tensor, dsize_src: tuple[int, int], dsize_dst: tuple[int, int]
) -> torch.Tensor:
    """Normalise a (B,3,3) pixel homography to [-1, 1] coordinates: N_dst @ (M @ inv(N_src))."""
    if not isinstance(dst_pix_trans_src_pix, torch.Tensor):
        raise TypeError(f"Input type is not a torch.Tensor. Got {type(dst_pix_trans_src_pix)}")
    if not (len(dst_pix_trans_src_pix.shape) == 3 or dst_pix_trans_src_pix.shape[-2:] == (3, 3)):
        raise ValueError(f"Input dst_pix_trans_src_pix must be a Bx3x3 tensor. Got {dst_pix_trans_src_pix.shape}")
    return _ChainFunction.apply(dst_pix_trans_src_pix, tuple(dsize_src), tuple(dsize_dst), False)


def convert_affinematrix_to_homography(A: torch.Tensor) -> torch.Tensor:
    """(B,2,3) -> (B,3,3) by appending the row [0, 0, 1]."""
    if not isinstance(A, torch.Tensor):
        raise TypeError(f"Input type is not a torch.Tensor. Got {type(A)}")
    if not (len(A.shape) == 3 and A.shape[-2:] == (2, 3)):
        raise ValueError(f"Input matrix must be a Bx2x3 tensor. Got {A.shape}")
    H = F.pad(A, [0, 0, 0, 1], "constant", value=0.0)
    H[..., -1, -1] += 1.0
    return H


def convert_points_from_homogeneous(points: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """(*,N,D) -> (*,N,D-1): divide by w + eps where |w| > eps, pass through otherwise."""
    if not isinstance(points, torch.Tensor):
        raise TypeError(f"Input type is not a torch.Tensor. Got {type(points)}")
    if len(points.shape) < 2:
        raise ValueError(f"Input must be at least a 2D tensor. Got {points.shape}")
    z = points[..., -1:]
    scale = torch.where(torch.abs(z) > eps, 1.0 / (z + eps), torch.ones_like(z))
    return scale * points[..., :-1]


def convert_points_to_homogeneous(points: torch.Tensor) -> torch.Tensor:
    """(*,N,D) -> (*,N,D+1) by appending 1."""
    if not isinstance(points, torch.Tensor):
        raise TypeError(f"Input type is not a torch.Tensor. Got {type(points)}")
    if len(points.shape) < 2:
        raise ValueError(f"Input must be at least a 2D tensor. Got {points.shape}")
    return F.pad(points, [0, 1], "constant", 1.0)


def normalize_pixel_coordinates(pixel_coordinates: torch.Tensor, height: int, width: int, eps: float = 1e-8) -> torch.Tensor:
    """(*,2) pixel (x, y) -> [-1, 1]: ``2 / (size - 1).clamp(eps) * p - 1`` (conversions.py:1455-1500).
    Elementwise O(N) tensor expression kept in PyTorch: it defines the bits of the grid ``remap`` samples with."""
    if pixel_coordinates.shape[-1] != 2:
        raise ValueError(f"Input pixel_coordinates must be of shape (*, 2). Got {pixel_coordinates.shape}")
    hw = torch.stack(
        [
            torch.tensor(width, device=pixel_coordinates.device, dtype=pixel_coordinates.dtype),
            torch.tensor(height, device=pixel_coordinates.device, dtype=pixel_coordinates.dtype),
        ]
    )
    factor = torch.tensor(2.0, device=pixel_coordinates.device, dtype=pixel_coordinates.dtype) / (hw - 1).clamp(eps)
    return factor * pixel_coordinates - 1
