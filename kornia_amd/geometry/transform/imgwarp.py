"""warp_perspective / warp_affine / homography_warp / warp_grid on the gfx950 kernels.

Same signatures, defaults, argument meaning and error behaviour as the reference
(kornia/geometry/transform/imgwarp.py:69-174, :177-290, :323-353, :1476-1549).  What differs is the
execution: one HIP launch for the 3x3 chain and one for coordinate generation + sampling, instead
of ~25 elementwise launches that build a (B,h,w,2) grid in HBM followed by ``F.grid_sample``; the
backward takes both gradients from ONE read of grad_out when both are wanted (persistent tile-owner kernel,
csrc/km_warp_bwd_fused.hip: bilinear, zeros / border / reflection / fill padding), otherwise one launch per gradient
(image: tile-owner scatter; matrix: forward-shaped reduction), plus the tiny chain adjoint.

There is no PyTorch/CPU fallback: tensors must live on a HIP device.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch

from ... import _native as N
from ..linalg import transform_points

__all__ = ["grid_sample", "homography_warp", "remap", "warp_affine", "warp_grid", "warp_perspective"]

COORD_PERSPECTIVE, COORD_AFFINE, COORD_HOMOGRAPHY = 0, 1, 2
_INTERP = {"nearest": 0, "bilinear": 1, "bicubic": 2}
_PAD = {"zeros": 0, "border": 1, "reflection": 2, "fill": 3}


class _WarpCfg(NamedTuple):
    dsize: tuple
    coord_mode: int
    norm_coords: int
    interp: int
    pad: int
    align: int


def _mode_codes(mode: str, padding_mode: str) -> tuple[int, int]:
    # F.grid_sample's own error messages for unknown modes
    if mode not in _INTERP:
        raise ValueError(
            f"nn.functional.grid_sample(): expected mode to be 'bilinear', 'nearest' or 'bicubic', but got: '{mode}'"
        )
    if padding_mode not in _PAD:
        raise ValueError(
            "nn.functional.grid_sample(): expected padding_mode to be 'zeros', 'border', or 'reflection', "
            f"but got: '{padding_mode}'"
        )
    return _INTERP[mode], _PAD[padding_mode]


class _Warp2dFunction(torch.autograd.Function):
    """src (B,C,H,W), mat: pixel matrix (B_M,3,3)/(B_M,2,3) [perspective/affine] or normalised
    dst->src homography (B_M,3,3) [homography]; fill: (C,) compute-dtype tensor or None."""

    @staticmethod
    def forward(ctx, src: torch.Tensor, mat: torch.Tensor, fill: Optional[torch.Tensor], cfg: _WarpCfg, apply: Optional[torch.Tensor] = None):
        lib = N.lib()
        dev = src.device
        cdt = N.compute_dtype(src.dtype)
        x = src.detach().contiguous()
        Mc = mat.detach().to(device=dev, dtype=cdt).contiguous()
        B, C, H, W = x.shape
        h, w = cfg.dsize
        B_M = Mc.shape[0]
        stream = N.stream_ptr(dev)
        out = torch.empty(B, C, h, w, device=dev, dtype=src.dtype)
        with N.device_guard(dev):
            if cfg.coord_mode == COORD_HOMOGRAPHY:
                m = Mc.view(B_M, 9)
            else:
                m = torch.empty(B_M, 9, device=dev, dtype=cdt)
                N.check(lib.km_homography_chain_fwd(Mc.data_ptr(), Mc.shape[1], None, m.data_ptr(), B_M, H, W, h, w,
                                                    N.dtype_code(cdt), stream), "km_homography_chain_fwd")
            if apply is None:
                N.check(lib.km_warp2d_fwd(x.data_ptr(), m.data_ptr(), out.data_ptr(), B, C, H, W, h, w, B_M, cfg.coord_mode,
                                          cfg.norm_coords, cfg.interp, cfg.pad, cfg.align, N.ptr(fill), N.dtype_code(src.dtype),
                                          stream), "km_warp2d_fwd")
            else:  # the augmentation layer's per-sample switch, folded into the launch (forward only: _warp refuses it under autograd)
                N.check(lib.km_warp2d_fwd_masked(x.data_ptr(), m.data_ptr(), out.data_ptr(), apply.data_ptr(), B, C, H, W, h, w, B_M, cfg.coord_mode,
                                                 cfg.norm_coords, cfg.interp, cfg.pad, cfg.align, N.ptr(fill), N.dtype_code(src.dtype),
                                                 stream), "km_warp2d_fwd_masked")
        ctx.save_for_backward(x, Mc, m, fill)
        ctx.cfg = cfg
        ctx.mat_dtype = mat.dtype
        return out

    @staticmethod
    def backward(ctx, gout: torch.Tensor):
        x, Mc, m, fill = ctx.saved_tensors
        gsrc, gmat = _warp2d_backward(gout, x, Mc, m, fill, ctx.cfg, ctx.mat_dtype, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gsrc, gmat, None, None, None


def _warp2d_backward(gout: torch.Tensor, x: torch.Tensor, Mc: torch.Tensor, m: torch.Tensor, fill, cfg: _WarpCfg, mat_dtype, need_src: bool, need_mat: bool):
    """(grad wrt src, grad wrt the caller's matrix) of a 2-D warp: km_warp2d_bwd_ws (+ the chain adjoint).  Shared by ``_Warp2dFunction`` and
    the fused warp + blur op (warp_blur.py)."""
    lib = N.lib()
    dev = x.device
    cdt = Mc.dtype
    B, C, H, W = x.shape
    h, w = cfg.dsize
    B_M = Mc.shape[0]
    g = gout.detach().to(x.dtype).contiguous()
    stream = N.stream_ptr(dev)
    gmat = None
    # both gradients wanted - or the image gradient under border / reflection padding, where the tile-owner kernel replaces a scatter with
    # global atomics: a workspace lets the library take them from one read of grad_out (include/kornia_amd.h)
    ws, ws_bytes = None, 0
    if need_src and (need_mat or cfg.pad in (_PAD["border"], _PAD["reflection"])):
        ws_bytes = int(lib.km_warp2d_bwd_workspace_bytes(B, C, H, W, h, w, cfg.interp, cfg.pad, N.dtype_code(x.dtype)))
        if ws_bytes > 0:
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    gsrc = None
    if need_src:
        zero = ws is None and lib.km_warp2d_bwd_needs_zero_init(cfg.interp, cfg.pad, N.dtype_code(x.dtype))  # (the tile owners write every pixel)
        gsrc = (torch.zeros if zero else torch.empty)(B, C, H, W, device=dev, dtype=cdt)
    # (on the one-read path the first launch zeroes the fp64 accumulators itself: one fill launch less per step)
    gm = (torch.empty if ws is not None else torch.zeros)(B_M, 9, device=dev, dtype=torch.float64) if need_mat else None
    with N.device_guard(dev):
        N.check(lib.km_warp2d_bwd_ws(g.data_ptr(), x.data_ptr(), m.data_ptr(), N.ptr(gsrc), N.ptr(gm), B, C, H, W, h, w,
                                     B_M, cfg.coord_mode, cfg.norm_coords, cfg.interp, cfg.pad, cfg.align, N.ptr(fill),
                                     N.dtype_code(x.dtype), N.ptr(ws), ws_bytes, stream), "km_warp2d_bwd_ws")
        if need_mat:
            if cfg.coord_mode == COORD_HOMOGRAPHY:
                gmat = gm.view(B_M, 3, 3).to(mat_dtype)
            else:
                gM = torch.empty_like(Mc)
                N.check(lib.km_homography_chain_bwd(Mc.data_ptr(), Mc.shape[1], gm.data_ptr(), gM.data_ptr(), B_M, H,
                                                    W, h, w, N.dtype_code(cdt), stream), "km_homography_chain_bwd")
                gmat = gM.to(mat_dtype)
    if gsrc is not None and gsrc.dtype != x.dtype:
        gsrc = gsrc.to(x.dtype)
    return gsrc, gmat


class _GridSampleFunction(torch.autograd.Function):
    """input (B,C,H,W), grid (B_G,h,w,2) normalised (x, y) in the input dtype, B_G in {1, B}."""

    @staticmethod
    def forward(ctx, input: torch.Tensor, grid: torch.Tensor, interp: int, pad: int, align: int):
        lib = N.lib()
        dev = input.device
        x = input.detach().contiguous()
        gr = grid.detach().contiguous()
        B, C, H, W = x.shape
        B_G, h, w, _ = gr.shape
        out = torch.empty(B, C, h, w, device=dev, dtype=x.dtype)
        with N.device_guard(dev):
            N.check(lib.km_grid_sample2d_fwd(x.data_ptr(), gr.data_ptr(), out.data_ptr(), B, C, H, W, h, w, B_G, interp, pad, align,
                                             N.dtype_code(x.dtype), N.stream_ptr(dev)), "km_grid_sample2d_fwd")
        ctx.save_for_backward(x, gr)
        ctx.cfg = (interp, pad, align)
        return out

    @staticmethod
    def backward(ctx, gout: torch.Tensor):
        x, gr = ctx.saved_tensors
        interp, pad, align = ctx.cfg
        lib = N.lib()
        dev = x.device
        cdt = N.compute_dtype(x.dtype)
        B, C, H, W = x.shape
        B_G, h, w, _ = gr.shape
        g = gout.detach().to(x.dtype).contiguous()
        gsrc = torch.zeros(B, C, H, W, device=dev, dtype=cdt) if ctx.needs_input_grad[0] else None
        ggrid = torch.empty(B, h, w, 2, device=dev, dtype=cdt) if ctx.needs_input_grad[1] else None
        with N.device_guard(dev):
            N.check(lib.km_grid_sample2d_bwd(g.data_ptr(), x.data_ptr(), gr.data_ptr(), N.ptr(gsrc), N.ptr(ggrid), B, C, H, W, h, w, B_G,
                                             interp, pad, align, N.dtype_code(x.dtype), N.stream_ptr(dev)), "km_grid_sample2d_bwd")
        if gsrc is not None and gsrc.dtype != x.dtype:
            gsrc = gsrc.to(x.dtype)
        if ggrid is not None:
            if B_G == 1 and B > 1:
                ggrid = ggrid.sum(0, keepdim=True)
            ggrid = ggrid.to(gr.dtype)
        return gsrc, ggrid, None, None, None


def grid_sample(input: torch.Tensor, grid: torch.Tensor, mode: str = "bilinear", padding_mode: str = "zeros",
                align_corners: Optional[bool] = None) -> torch.Tensor:
    """``F.grid_sample`` for 4-D inputs on the native sampler (call sites imgwarp.py:702, homography_warper.py:182).

    ``grid`` is ``(B,h,w,2)`` (or ``(1,h,w,2)``, shared by the batch without being expanded in memory),
    normalised ``(x, y)``, same dtype as ``input``; differentiable in both arguments.
    """
    N.require_device(input, "input")
    if input.dim() != 4 or grid.dim() != 4 or grid.shape[-1] != 2:
        raise ValueError(f"grid_sample: expected 4-D input and a (B,h,w,2) grid, got {tuple(input.shape)} and {tuple(grid.shape)}")
    if grid.dtype != input.dtype:
        raise RuntimeError(f"grid_sampler(): expected input and grid to have same dtype, but input has {input.dtype} and grid has {grid.dtype}")
    if grid.device != input.device:
        raise RuntimeError(f"grid_sampler(): expected input and grid to be on same device, but input is on {input.device} and grid is on {grid.device}")
    if not (grid.shape[0] == input.shape[0] or grid.shape[0] == 1):
        raise RuntimeError(
            f"grid_sampler(): expected grid and input to have same batch size, but got input with sizes {list(input.shape)} "
            f"and grid with sizes {list(grid.shape)}"
        )
    interp, pad = _mode_codes(mode, padding_mode)
    if pad == _PAD["fill"]:
        raise ValueError("nn.functional.grid_sample(): expected padding_mode to be 'zeros', 'border', or 'reflection', but got: 'fill'")
    return _GridSampleFunction.apply(input, grid, interp, pad, int(bool(align_corners)))


def remap(image: torch.Tensor, map_x: torch.Tensor, map_y: torch.Tensor, mode: str = "bilinear", padding_mode: str = "zeros",
          align_corners: Optional[bool] = None, normalized_coordinates: bool = False) -> torch.Tensor:
    r"""``dst(x, y) = src(map_x(x, y), map_y(x, y))`` (kornia/geometry/transform/imgwarp.py:625-702).

    ``image`` (B,C,H,W); ``map_x`` / ``map_y`` (B,H,W) or (1,H,W) in pixels unless ``normalized_coordinates``;
    ``align_corners=None`` resolves to False.  A single map is shared by the batch without being expanded in HBM.
    """
    from ...core.check import KORNIA_CHECK_SHAPE
    from ..conversions import normalize_pixel_coordinates

    KORNIA_CHECK_SHAPE(image, ["B", "C", "H", "W"])
    KORNIA_CHECK_SHAPE(map_x, ["B", "H", "W"])
    KORNIA_CHECK_SHAPE(map_y, ["B", "H", "W"])
    _, _, height, width = image.shape
    map_xy = torch.stack([map_x, map_y], -1)
    if not normalized_coordinates:
        map_xy = normalize_pixel_coordinates(map_xy, height, width)
    return grid_sample(image, map_xy.to(image.dtype), mode, padding_mode, bool(align_corners))


def _prepare_fill(fill_value: torch.Tensor, C: int, device, cdt) -> torch.Tensor:
    """imgwarp.py:308-313: scalar or (1,)/(C,) fill broadcast over channels."""
    f = fill_value.detach().to(device=device, dtype=cdt).reshape(-1)
    if f.numel() == 1:
        f = f.expand(C)
    elif f.numel() != C:
        raise RuntimeError(
            f"The size of tensor a ({C}) must match the size of tensor b ({f.numel()}) at non-singleton dimension 1"
        )
    return f.contiguous()


def _warp(src, mat, dsize, coord_mode, norm_coords, mode, padding_mode, align_corners, fill_value, apply=None):
    N.require_device(src, "src")
    interp, pad = _mode_codes(mode, padding_mode)
    B, C = src.shape[0], src.shape[1]
    B_M = mat.shape[0]
    if not (B_M == B or (B_M == 1 and coord_mode == COORD_AFFINE)):
        # what F.grid_sample reports when the (B_M,h,w,2) grid meets a (B,...) input
        raise RuntimeError(
            f"grid_sampler(): expected grid and input to have same batch size, but got input with sizes {list(src.shape)} "
            f"and grid with sizes {[B_M, int(dsize[0]), int(dsize[1]), 2]}"
        )
    fill = None
    if pad == _PAD["fill"]:
        fill = _prepare_fill(fill_value, C, src.device, N.compute_dtype(src.dtype))
    cfg = _WarpCfg((int(dsize[0]), int(dsize[1])), coord_mode, int(bool(norm_coords)), interp, pad, int(bool(align_corners)))
    if apply is not None:
        if torch.is_grad_enabled() and (src.requires_grad or mat.requires_grad):
            raise RuntimeError("the per-sample switch of the warp is forward-only")
        if tuple(cfg.dsize) != tuple(src.shape[-2:]) or apply.numel() != B:
            raise ValueError("the per-sample switch needs dsize == the source size and one entry per sample")
        apply = N.flags(apply, src.device, B)
    return _Warp2dFunction.apply(src, mat, fill, cfg, apply)


def _warp_affine_from_chain(src: torch.Tensor, m: torch.Tensor, mode: str, padding_mode: str, align_corners: bool, fill_value, apply) -> torch.Tensor:
    """Forward-only ``warp_affine`` to the source's own size from an already normalised and inverted (B,9) float32 matrix
    (``km_affine_params_chain_fwd``): the augmentation layer's apply step, no autograd node, one launch."""
    N.require_device(src, "src")
    interp, pad = _mode_codes(mode, padding_mode)
    B, C, H, W = src.shape
    dev = src.device
    if src.dtype not in (torch.float32, torch.bfloat16, torch.float16) or m.dtype != torch.float32 or tuple(m.shape) != (B, 9):
        raise TypeError("the chained warp takes float32 / bfloat16 / float16 images and a (B,9) float32 matrix")
    fill = _prepare_fill(fill_value, C, dev, torch.float32) if pad == _PAD["fill"] else None
    x = src.detach().contiguous()
    out = torch.empty_like(x)
    if apply is not None:
        apply = N.flags(apply, dev, B)
    with N.device_guard(dev):
        if apply is None:
            N.check(N.lib().km_warp2d_fwd(x.data_ptr(), m.data_ptr(), out.data_ptr(), B, C, H, W, H, W, B, COORD_AFFINE, 1, interp, pad,
                                          int(bool(align_corners)), N.ptr(fill), N.dtype_code(x.dtype), N.stream_ptr(dev)), "km_warp2d_fwd")
        else:
            N.check(N.lib().km_warp2d_fwd_masked(x.data_ptr(), m.data_ptr(), out.data_ptr(), apply.data_ptr(), B, C, H, W, H, W, B, COORD_AFFINE, 1, interp,
                                                 pad, int(bool(align_corners)), N.ptr(fill), N.dtype_code(x.dtype), N.stream_ptr(dev)), "km_warp2d_fwd_masked")
    return out


def warp_perspective(
    src: torch.Tensor,
    M: torch.Tensor,
    dsize: tuple[int, int],
    mode: str = "bilinear",
    padding_mode: str = "zeros",
    align_corners: bool = True,
    fill_value: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    r"""Warp ``src`` (B,C,H,W) with the source->destination **pixel** homography ``M`` (B,3,3):
    ``dst(x, y) = src(M^-1 (x, y, 1))``, output size ``dsize = (h, w)``.

    ``mode``: ``'bilinear' | 'nearest' | 'bicubic'``; ``padding_mode``: ``'zeros' | 'border' |
    'reflection' | 'fill'`` (``fill_value``: tensor of shape (3,), RGB only).  Differentiable wrt
    ``src`` and ``M``.
    """
    if not isinstance(src, torch.Tensor):
        raise TypeError(f"Input src type is not a torch.Tensor. Got {type(src)}")
    if not isinstance(M, torch.Tensor):
        raise TypeError(f"Input M type is not a torch.Tensor. Got {type(M)}")
    if not len(src.shape) == 4:
        raise ValueError(f"Input src must be a BxCxHxW torch.Tensor. Got {src.shape}")
    if not (len(M.shape) == 3 and M.shape[-2:] == (3, 3)):
        raise ValueError(f"Input M must be a Bx3x3 torch.Tensor. Got {M.shape}")
    if fill_value is None:
        fill_value = torch.zeros(3)
    if padding_mode == "fill" and fill_value.shape != torch.Size([3]):
        raise ValueError(f"Padding_tensor only supported for 3 channels. Got {fill_value.shape}")
    return _warp(src, M, dsize, COORD_PERSPECTIVE, 1, mode, padding_mode, align_corners, fill_value)


def warp_affine(
    src: torch.Tensor,
    M: torch.Tensor,
    dsize: tuple[int, int],
    mode: str = "bilinear",
    padding_mode: str = "zeros",
    align_corners: bool = True,
    fill_value: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    r"""Warp ``src`` (B,C,H,W) with the source->destination **pixel** affine matrix ``M`` (B,2,3)
    (a (1,2,3) matrix is shared by the whole batch).  Same modes as :func:`warp_perspective`;
    ``fill_value`` has shape (C,) or (1,)."""
    if not isinstance(src, torch.Tensor):
        raise TypeError(f"Input src type is not a torch.Tensor. Got {type(src)}")
    if not isinstance(M, torch.Tensor):
        raise TypeError(f"Input M type is not a torch.Tensor. Got {type(M)}")
    if not len(src.shape) == 4:
        raise ValueError(f"Input src must be a BxCxHxW torch.Tensor. Got {src.shape}")
    if not (len(M.shape) == 3 or M.shape[-2:] == (2, 3)):
        raise ValueError(f"Input M must be a Bx2x3 torch.Tensor. Got {M.shape}")
    # the reference's loose check above lets a wrong-shaped M through to
    # convert_affinematrix_to_homography, which raises this (conversions.py:375-376)
    if not (len(M.shape) == 3 and M.shape[-2:] == (2, 3)):
        raise ValueError(f"Input matrix must be a Bx2x3 tensor. Got {M.shape}")
    if padding_mode == "fill" and fill_value is None:
        fill_value = torch.zeros(src.shape[1], device=src.device, dtype=src.dtype)
    return _warp(src, M, dsize, COORD_AFFINE, 1, mode, padding_mode, align_corners, fill_value)


def warp_grid(grid: torch.Tensor, src_homo_dst: torch.Tensor) -> torch.Tensor:
    """Transform a (1,H,W,2) (or (N,H,W,2)) coordinate grid by destination->source homographies
    (N,3,3) / (N,1,3,3); returns (N,H,W,2)."""
    batch_size = src_homo_dst.size(0)
    _, height, width, _ = grid.size()
    grid = grid.expand(batch_size, -1, -1, -1)
    if len(src_homo_dst.shape) == 3:
        src_homo_dst = src_homo_dst.view(batch_size, 1, 3, 3)
    flow = transform_points(src_homo_dst, grid.to(src_homo_dst))
    return flow.view(batch_size, height, width, 2)


def homography_warp(
    patch_src: torch.Tensor,
    src_homo_dst: torch.Tensor,
    dsize: tuple[int, int],
    mode: str = "bilinear",
    padding_mode: str = "zeros",
    align_corners: bool = False,
    normalized_coordinates: bool = True,
    normalized_homography: bool = True,
) -> torch.Tensor:
    r"""Warp (N,C,H,W) patches by (N,3,3) homographies.

    ``normalized_homography=True`` (default): ``src_homo_dst`` maps destination->source in
    normalised [-1, 1] coordinates (pixel indices when ``normalized_coordinates=False``) and is
    applied to the base grid directly.  ``normalized_homography=False``: it is a source->destination
    pixel homography handled exactly like :func:`warp_perspective` with ``mode='bilinear'`` and
    ``align_corners=True``.
    """
    if not src_homo_dst.device == patch_src.device:
        raise TypeError(
            f"Patch and homography must be on the same device. Got patch.device: {patch_src.device} "
            f"src_H_dst.device: {src_homo_dst.device}."
        )
    if normalized_homography:
        if not (len(src_homo_dst.shape) == 3 and src_homo_dst.shape[-2:] == (3, 3)):
            raise ValueError(f"Input src_homo_dst must be a Nx3x3 torch.Tensor. Got {src_homo_dst.shape}")
        return _warp(patch_src, src_homo_dst, dsize, COORD_HOMOGRAPHY, normalized_coordinates, mode, padding_mode,
                     align_corners, None)
    return warp_perspective(patch_src, src_homo_dst, dsize, mode="bilinear", padding_mode=padding_mode, align_corners=True)
