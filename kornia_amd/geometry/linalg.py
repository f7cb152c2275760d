"""transform_points (kornia/geometry/linalg.py:183-239) on the native kernel csrc/km_points.hip."""
from __future__ import annotations

import torch

from .. import _native as N
from ..core.check import KORNIA_CHECK_IS_TENSOR

__all__ = ["transform_points"]


class _TransformPointsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, T: torch.Tensor, pts: torch.Tensor):
        # T (B_T,D+1,D+1), pts (B,N,D), same compute dtype, contiguous
        B, Np, D = pts.shape
        out = torch.empty_like(pts)
        with N.device_guard(pts.device):
            rc = N.lib().km_transform_points_fwd(T.data_ptr(), pts.data_ptr(), out.data_ptr(), B, Np, D, T.shape[0],
                                                 N.dtype_code(pts.dtype), N.stream_ptr(pts.device))
        N.check(rc, "km_transform_points_fwd")
        ctx.save_for_backward(T, pts)
        return out

    @staticmethod
    def backward(ctx, g: torch.Tensor):
        T, pts = ctx.saved_tensors
        B, Np, D = pts.shape
        g = g.contiguous()
        gpts = torch.empty_like(pts) if ctx.needs_input_grad[1] else None
        gT = torch.zeros(T.shape[0], (D + 1) * (D + 1), device=pts.device, dtype=torch.float64) if ctx.needs_input_grad[0] else None
        with N.device_guard(pts.device):
            rc = N.lib().km_transform_points_bwd(g.data_ptr(), T.data_ptr(), pts.data_ptr(), N.ptr(gpts), N.ptr(gT), B, Np, D,
                                                 T.shape[0], N.dtype_code(pts.dtype), N.stream_ptr(pts.device))
        N.check(rc, "km_transform_points_bwd")
        return (None if gT is None else gT.view_as(T).to(T.dtype)), gpts


def transform_points(trans_01: torch.Tensor, points_1: torch.Tensor) -> torch.Tensor:
    """Apply (B,D+1,D+1) transforms to (B,N,D) points (leading dims are flattened, a transform batch
    of 1 broadcasts); homogeneous divide uses the reference's ``w + 1e-8`` convention."""
    KORNIA_CHECK_IS_TENSOR(trans_01)
    KORNIA_CHECK_IS_TENSOR(points_1)
    if not trans_01.shape[0] == points_1.shape[0] and trans_01.shape[0] != 1:
        raise ValueError(
            f"Input batch size must be the same for both tensors or 1. Got {trans_01.shape} and {points_1.shape}"
        )
    if not trans_01.shape[-1] == (points_1.shape[-1] + 1):
        raise ValueError(f"Last input dimensions must differ by one unit Got{trans_01} and {points_1}")
    if points_1.shape[-2] == 0:
        return points_1
    N.require_device(points_1, "points_1")
    N.require_device(trans_01, "trans_01")
    D = points_1.shape[-1]
    if D not in (2, 3):
        raise ValueError(f"kornia_amd.transform_points supports 2-D and 3-D points, got D={D}")
    shape_inp = list(points_1.shape)
    points_dtype = points_1.dtype
    # the reference computes in the transform's dtype and returns the points' dtype (linalg.py:227-239)
    cdt = N.compute_dtype(trans_01.dtype)
    pts = points_1.reshape(-1, shape_inp[-2], D).to(cdt).contiguous()
    T = trans_01.reshape(-1, D + 1, D + 1).to(cdt).contiguous()
    if T.shape[0] != 1 and T.shape[0] != pts.shape[0]:
        T = torch.repeat_interleave(T, repeats=int(pts.shape[0] // T.shape[0]), dim=0)
    out = _TransformPointsFunction.apply(T, pts)
    return out.reshape(shape_inp).to(points_dtype)
