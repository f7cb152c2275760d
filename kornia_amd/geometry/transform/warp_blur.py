"""warp + separable blur as ONE forward launch (csrc/km_warp_blur.hip): ``gaussian_blur2d(warp_perspective(src, M, dsize), k, sigma)`` and the
affine form, bit-identical to the two calls, without the warped image ever reaching HBM (5e instead of 9e bytes per element
over forward + backward - "report it if fusion is added").

Not part of the reference's API: the reference composes ``kornia.geometry.transform.warp_perspective`` (imgwarp.py:69-174) /
``warp_affine`` (:177-290) with ``kornia.filters.gaussian_blur2d`` (gaussian.py:32-120) as two ops, and so does the headline benchmark.  This
is the op a pipeline that knows it wants both calls can use instead (RandomAffine / RandomPerspective followed by RandomGaussianBlur); the
backward is the blur adjoint followed by the warp's one-read backward - two launches that already exist.
"""
from __future__ import annotations

import torch

from ... import _native as N
from ...filters.filter import _BORDER_CODE, filter2d_separable
from ...filters.gaussian import _cached_taps, _check_host_sigma
from ...filters.kernels import _check_kernel_size, _unpack_2d_ks
from .imgwarp import COORD_AFFINE, COORD_PERSPECTIVE, _mode_codes, _warp, _warp2d_backward, _WarpCfg

__all__ = ["warp_affine_blur", "warp_perspective_blur"]


class _WarpBlurFunction(torch.autograd.Function):
    """src (B,C,H,W); mat: pixel matrix (B,3,3) / (B_M,2,3); kx, ky (1,K) float taps already rounded to the image dtype."""

    @staticmethod
    def forward(ctx, src: torch.Tensor, mat: torch.Tensor, kx: torch.Tensor, ky: torch.Tensor, cfg: _WarpCfg, border: int):
        lib = N.lib()
        dev = src.device
        cdt = N.compute_dtype(src.dtype)
        x = src.detach().contiguous()
        Mc = mat.detach().to(device=dev, dtype=cdt).contiguous()
        B, C, H, W = x.shape
        h, w = cfg.dsize
        B_M = Mc.shape[0]
        K = kx.shape[1]
        stream = N.stream_ptr(dev)
        out = torch.empty(B, C, h, w, device=dev, dtype=src.dtype)
        with N.device_guard(dev):
            m = torch.empty(B_M, 9, device=dev, dtype=cdt)
            N.check(lib.km_homography_chain_fwd(Mc.data_ptr(), Mc.shape[1], None, m.data_ptr(), B_M, H, W, h, w, N.dtype_code(cdt), stream),
                    "km_homography_chain_fwd")
            N.check(lib.km_warp2d_blur_fwd(x.data_ptr(), m.data_ptr(), kx.data_ptr(), ky.data_ptr(), out.data_ptr(), B, C, H, W, h, w, B_M, kx.shape[0],
                                           cfg.coord_mode, cfg.norm_coords, cfg.align, K, border, N.dtype_code(src.dtype), stream), "km_warp2d_blur_fwd")
        ctx.save_for_backward(x, Mc, m, kx, ky)
        ctx.cfg, ctx.border, ctx.mat_dtype = cfg, border, mat.dtype
        return out

    @staticmethod
    def backward(ctx, gy: torch.Tensor):
        x, Mc, m, kx, ky = ctx.saved_tensors
        cfg: _WarpCfg = ctx.cfg
        B, C = x.shape[0], x.shape[1]
        h, w = cfg.dsize
        K = kx.shape[1]
        g = gy.detach().to(x.dtype).contiguous()
        gw = torch.empty(B, C, h, w, device=g.device, dtype=x.dtype)
        with N.device_guard(g.device):  # the blur's adjoint (km_blur_fast.hip), then the warp's own backward
            N.check(N.lib().km_filter2d_sep_bwd_input(g.data_ptr(), kx.data_ptr(), ky.data_ptr(), gw.data_ptr(), B, C, h, w, kx.shape[0], K, K,
                                                      ctx.border, 1, N.dtype_code(x.dtype), N.stream_ptr(g.device)), "km_filter2d_sep_bwd_input")
        gsrc, gmat = _warp2d_backward(gw, x, Mc, m, None, cfg, ctx.mat_dtype, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gsrc, gmat, None, None, None, None


def _warp_blur(src, M, dsize, coord_mode, kernel_size, sigma, border_type, mode, padding_mode, align_corners, two_ops):
    _check_kernel_size(kernel_size, min_value=0)
    ky, kx = _unpack_2d_ks(kernel_size)
    fusable = (
        isinstance(src, torch.Tensor) and isinstance(M, torch.Tensor) and src.dim() == 4 and N.on_device(src) and isinstance(sigma, tuple)
        and kx == ky and str(border_type).lower() in _BORDER_CODE and src.dtype in (torch.float32, torch.bfloat16, torch.float16)
        and M.dim() == 3 and (M.shape[0] == src.shape[0] or (M.shape[0] == 1 and coord_mode == COORD_AFFINE))
        # the matrix shape of THIS op (anything else takes the two calls, whose own argument checks raise what the reference raises)
        and tuple(M.shape[-2:]) == ((2, 3) if coord_mode == COORD_AFFINE else (3, 3))
        and isinstance(dsize, (tuple, list)) and len(dsize) == 2
    )
    if fusable:
        interp, pad = _mode_codes(mode, padding_mode)
        B, C, H, W = src.shape
        fusable = bool(N.lib().km_warp2d_blur_supported(C, H, W, int(dsize[0]), int(dsize[1]), interp, pad, kx, _BORDER_CODE[str(border_type).lower()],
                                                        N.dtype_code(src.dtype)))
    if not fusable:
        return two_ops()
    host_sigma = _check_host_sigma(sigma)
    taps_x, taps_y = _cached_taps(ky, kx, host_sigma, src.dtype, src.device)  # (1,K) in the image dtype, as gaussian_blur2d builds them
    cfg = _WarpCfg((int(dsize[0]), int(dsize[1])), coord_mode, 1, interp, pad, int(bool(align_corners)))
    return _WarpBlurFunction.apply(src, M, taps_x.float().contiguous(), taps_y.float().contiguous(), cfg, _BORDER_CODE[str(border_type).lower()])


def warp_perspective_blur(src: torch.Tensor, M: torch.Tensor, dsize: tuple[int, int], kernel_size, sigma: tuple[float, float],
                          border_type: str = "reflect", mode: str = "bilinear", padding_mode: str = "zeros", align_corners: bool = True) -> torch.Tensor:
    """``gaussian_blur2d(warp_perspective(src, M, dsize, mode, padding_mode, align_corners), kernel_size, sigma, border_type)`` - the same bits -
    in one forward launch when the modes allow it (bilinear + zeros, grey / RGB, square odd kernel of 3, 5 or 7, tuple sigma); the two calls
    otherwise.  Differentiable wrt ``src`` and ``M``."""
    from ...filters.gaussian import gaussian_blur2d
    from .imgwarp import warp_perspective

    return _warp_blur(src, M, dsize, COORD_PERSPECTIVE, kernel_size, sigma, border_type, mode, padding_mode, align_corners,
                      lambda: gaussian_blur2d(warp_perspective(src, M, dsize, mode, padding_mode, align_corners), kernel_size, sigma, border_type))


def warp_affine_blur(src: torch.Tensor, M: torch.Tensor, dsize: tuple[int, int], kernel_size, sigma: tuple[float, float],
                     border_type: str = "reflect", mode: str = "bilinear", padding_mode: str = "zeros", align_corners: bool = True) -> torch.Tensor:
    """``gaussian_blur2d(warp_affine(src, M, dsize, ...), kernel_size, sigma, border_type)`` in one forward launch (see :func:`warp_perspective_blur`)."""
    from ...filters.gaussian import gaussian_blur2d
    from .imgwarp import warp_affine

    return _warp_blur(src, M, dsize, COORD_AFFINE, kernel_size, sigma, border_type, mode, padding_mode, align_corners,
                      lambda: gaussian_blur2d(warp_affine(src, M, dsize, mode, padding_mode, align_corners), kernel_size, sigma, border_type))
