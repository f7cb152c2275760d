"""Matrix builders that feed the warp path.

These are O(B) scalar formulas - a few hundred flops per matrix - so they stay as tensor expressions on
whatever device the inputs live on (they are differentiable through autograd like the reference's).
Each one is written in closed form instead of the reference's chain of eye_like / slice-assign / bmm
launches, so a batch of matrices costs a handful of elementwise kernels.

Reference behaviour mirrored (names, argument meaning, shapes, errors):
  get_perspective_transform   kornia/geometry/transform/imgwarp.py:459-525  (Heckbert square->quad twice)
  get_rotation_matrix2d       kornia/geometry/transform/imgwarp.py:529-622
  get_affine_matrix2d         kornia/geometry/transform/imgwarp.py:746-787
  get_translation_matrix2d    kornia/geometry/transform/imgwarp.py:790-812
  get_shear_matrix2d          kornia/geometry/transform/imgwarp.py:815-869
  angle_to_rotation_matrix    kornia/geometry/conversions.py:1652-1688
  deg2rad                     kornia/geometry/conversions.py:118-148   (float32 pi, cast to input dtype)
"""
from __future__ import annotations

from typing import Optional

import torch

from ...core.check import KORNIA_CHECK, KORNIA_CHECK_SHAPE

__all__ = [
    "angle_to_rotation_matrix",
    "deg2rad",
    "get_affine_matrix2d",
    "get_perspective_transform",
    "get_rotation_matrix2d",
    "get_shear_matrix2d",
    "get_translation_matrix2d",
]

# the reference keeps pi as a float32 tensor and casts it to the input dtype (conversions.py:148,
# constants.py:25), so fp64 angles also see the float32-rounded constant
_PI_F32 = float(torch.tensor(3.14159265358979323846, dtype=torch.float32))


def deg2rad(tensor: torch.Tensor) -> torch.Tensor:
    """Degrees -> radians as ``x * pi_f32 / 180`` in the dtype of ``x``."""
    if not isinstance(tensor, torch.Tensor):
        raise TypeError(f"Input type is not a torch.Tensor. Got {type(tensor)}")
    if tensor.is_floating_point():
        return tensor * _PI_F32 / 180.0
    # integer input: the reference truncates pi to the integer dtype first (documented defect, kept)
    return tensor * int(_PI_F32) / 180.0


def angle_to_rotation_matrix(angle: torch.Tensor) -> torch.Tensor:
    """(*) degrees -> (*,2,2) ``[[cos, sin], [-sin, cos]]``."""
    rad = deg2rad(angle)
    c, s = torch.cos(rad), torch.sin(rad)
    return torch.stack([c, s, -s, c], dim=-1).reshape(*angle.shape, 2, 2)


def _square_to_quad(q: torch.Tensor):
    """Coefficients (a..h) of the projective map taking the unit square's corners (0,0),(1,0),(1,1),(0,1)
    to the quad ``q`` (B,4,2), after Heckbert, "Fundamentals of Texture Mapping and Image Warping" (1989) §2.
    Returned as a tuple of (B,) tensors for the matrix [[a,b,c],[d,e,f],[g,h,1]]."""
    x0, y0 = q[..., 0, 0], q[..., 0, 1]
    x1, y1 = q[..., 1, 0], q[..., 1, 1]
    x2, y2 = q[..., 2, 0], q[..., 2, 1]
    x3, y3 = q[..., 3, 0], q[..., 3, 1]
    ex1, ex2 = x1 - x2, x3 - x2
    ey1, ey2 = y1 - y2, y3 - y2
    sx = (x0 - x1) + (x2 - x3)
    sy = (y0 - y1) + (y2 - y3)
    det = ex1 * ey2 - ex2 * ey1
    g = (sx * ey2 - ex2 * sy) / det
    h = (ex1 * sy - sx * ey1) / det
    a = (x1 - x0) + g * x1
    b = (x3 - x0) + h * x3
    d = (y1 - y0) + g * y1
    e = (y3 - y0) + h * y3
    return a, b, x0, d, e, y0, g, h


def get_perspective_transform(points_src: torch.Tensor, points_dst: torch.Tensor) -> torch.Tensor:
    """(B,4,2),(B,4,2) -> (B,3,3) pixel homography taking ``points_src`` onto ``points_dst``, ``H[2,2] == 1``.

    ``H = Q_dst · adj(Q_src)`` with Q the square->quad maps; the adjugate replaces the inverse because the
    result is renormalised by ``H[2,2]`` anyway.  Half precision is computed in fp32 and cast back.
    """
    KORNIA_CHECK_SHAPE(points_src, ["B", "4", "2"])
    KORNIA_CHECK_SHAPE(points_dst, ["B", "4", "2"])
    KORNIA_CHECK(points_src.shape == points_dst.shape, "Source data shape must match Destination data shape.")
    KORNIA_CHECK(points_src.dtype == points_dst.dtype, "Source data type must match Destination data type.")
    dtype = points_src.dtype
    work = dtype if dtype in (torch.float32, torch.float64) else torch.float32
    from ... import _native as N

    if N.on_device(points_src) and N.on_device(points_dst) and not (torch.is_grad_enabled() and (points_src.requires_grad or points_dst.requires_grad)):
        if N.is_built() and dtype in (torch.float32, torch.float64, torch.bfloat16, torch.float16):
            # one launch (km_perspective_transform_fwd, same formulas); inputs that need gradients take the expression below
            ps, pd = points_src.detach().to(work).contiguous(), points_dst.detach().to(work).contiguous()
            out = torch.empty(ps.shape[0], 3, 3, device=ps.device, dtype=work)
            with N.device_guard(ps.device):
                N.check(N.lib().km_perspective_transform_fwd(ps.data_ptr(), pd.data_ptr(), out.data_ptr(), ps.shape[0], N.dtype_code(work),
                                                             N.stream_ptr(ps.device)), "km_perspective_transform_fwd")
            return out.to(dtype)
    a, b, c, d, e, f, g, h = _square_to_quad(points_src.to(work))
    A, Bq, C, D, E, Fq, G, Hq = _square_to_quad(points_dst.to(work))
    # adjugate of [[a,b,c],[d,e,f],[g,h,1]]
    j00, j01, j02 = e - f * h, c * h - b, b * f - c * e
    j10, j11, j12 = f * g - d, a - c * g, c * d - a * f
    j20, j21, j22 = d * h - e * g, b * g - a * h, a * e - b * d
    rows = [
        A * j00 + Bq * j10 + C * j20, A * j01 + Bq * j11 + C * j21, A * j02 + Bq * j12 + C * j22,
        D * j00 + E * j10 + Fq * j20, D * j01 + E * j11 + Fq * j21, D * j02 + E * j12 + Fq * j22,
        G * j00 + Hq * j10 + j20, G * j01 + Hq * j11 + j21, G * j02 + Hq * j12 + j22,
    ]
    Hm = torch.stack(rows, dim=-1)
    Hm = Hm / Hm[..., 8:9]
    return Hm.reshape(*points_src.shape[:-2], 3, 3).to(dtype)


def get_rotation_matrix2d(center: torch.Tensor, angle: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """(B,2) centre, (B,) degrees (CCW-positive), (B,2) scale -> (B,2,3):  T(c) · R(angle) · S(scale) · T(-c)."""
    if not isinstance(center, torch.Tensor):
        raise TypeError(f"Input center type is not a torch.Tensor. Got {type(center)}")
    if not isinstance(angle, torch.Tensor):
        raise TypeError(f"Input angle type is not a torch.Tensor. Got {type(angle)}")
    if not isinstance(scale, torch.Tensor):
        raise TypeError(f"Input scale type is not a torch.Tensor. Got {type(scale)}")
    if not (len(center.shape) == 2 and center.shape[1] == 2):
        raise ValueError(f"Input center must be a Bx2 torch.Tensor. Got {center.shape}")
    if not len(angle.shape) == 1:
        raise ValueError(f"Input angle must be a B torch.Tensor. Got {angle.shape}")
    if not (len(scale.shape) == 2 and scale.shape[1] == 2):
        raise ValueError(f"Input scale must be a Bx2 torch.Tensor. Got {scale.shape}")
    if not (center.shape[0] == angle.shape[0] == scale.shape[0]):
        raise ValueError(
            f"Inputs must have same batch size dimension. Got center {center.shape}, angle {angle.shape} and scale "
            f"{scale.shape}"
        )
    if not (center.device == angle.device == scale.device) or not (center.dtype == angle.dtype == scale.dtype):
        raise ValueError(
            f"Inputs must have same device Got center ({center.device}, {center.dtype}), angle ({angle.device}, "
            f"{angle.dtype}) and scale ({scale.device}, {scale.dtype})"
        )
    rad = deg2rad(angle)
    c, s = torch.cos(rad), torch.sin(rad)
    cx, cy = center[:, 0], center[:, 1]
    m00, m01 = c * scale[:, 0], s * scale[:, 1]
    m10, m11 = -s * scale[:, 0], c * scale[:, 1]
    tx = cx - (m00 * cx + m01 * cy)
    ty = cy - (m10 * cx + m11 * cy)
    return torch.stack([m00, m01, tx, m10, m11, ty], dim=-1).reshape(-1, 2, 3)


def _to_homography(A: torch.Tensor) -> torch.Tensor:
    last = torch.zeros_like(A[:, :1, :])
    last[:, 0, 2] = 1.0
    return torch.cat([A, last], dim=1)


def get_translation_matrix2d(translations: torch.Tensor) -> torch.Tensor:
    """(B,2) -> (B,3,3) pure translation."""
    B = translations.shape[0]
    H = torch.eye(3, device=translations.device, dtype=translations.dtype).repeat(B, 1, 1)
    H[:, :2, 2] = translations
    return H


def get_shear_matrix2d(
    center: torch.Tensor, sx: Optional[torch.Tensor] = None, sy: Optional[torch.Tensor] = None
) -> torch.Tensor:
    """(B,2) centre, shear angles in radians -> (B,3,3) shear about ``center``."""
    B = center.size(0)
    sx = torch.zeros(B, device=center.device, dtype=center.dtype) if sx is None else sx
    sy = torch.zeros(B, device=center.device, dtype=center.dtype) if sy is None else sy
    x, y = center[:, 0], center[:, 1]
    tx, ty = torch.tan(sx), torch.tan(sy)
    one = torch.ones_like(sx)
    A = torch.stack([one, -tx, tx * y, -ty, one + tx * ty, ty * (x - tx * y)], dim=-1).reshape(-1, 2, 3)
    return _to_homography(A)


def _affine_matrix2d_native(translations, center, scale, angle, sx, sy):
    from ... import _native as N

    tensors = [t for t in (translations, center, scale, angle, sx, sy) if t is not None]
    if not all(isinstance(t, torch.Tensor) and N.on_device(t) for t in tensors) or not N.is_built():
        return None
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        return None
    dtype = translations.dtype
    if dtype not in (torch.float32, torch.float64, torch.bfloat16, torch.float16) or any(t.dtype != dtype for t in tensors):
        return None
    B = translations.shape[0]
    if translations.shape != (B, 2) or center.shape != (B, 2) or scale.shape != (B, 2) or angle.shape != (B,):
        return None
    if any(t is not None and t.shape != (B,) for t in (sx, sy)):
        return None
    cdt = torch.float64 if dtype == torch.float64 else torch.float32
    dev = translations.device
    args = [None if t is None else t.detach().to(cdt).contiguous() for t in (translations, center, scale, angle, sx, sy)]
    out = torch.empty(B, 3, 3, device=dev, dtype=cdt)
    with N.device_guard(dev):
        N.check(N.lib().km_affine_matrix2d_fwd(*[N.ptr(t) for t in args], out.data_ptr(), B, N.dtype_code(cdt), N.stream_ptr(dev)),
                "km_affine_matrix2d_fwd")
    return out.to(dtype)


def get_affine_matrix2d(
    translations: torch.Tensor,
    center: torch.Tensor,
    scale: torch.Tensor,
    angle: torch.Tensor,
    sx: Optional[torch.Tensor] = None,
    sy: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """(B,3,3) pixel affine: rotation by ``-angle`` (clockwise-positive) and scale about ``center``, then
    translation, optionally right-multiplied by the shear about ``center``.

    HIP tensors that need no gradient are built by ``km_affine_matrix2d_fwd`` (one launch, the reference's own
    operation sequence); anything else takes the closed-form tensor expression below."""
    native = _affine_matrix2d_native(translations, center, scale, angle, sx, sy)
    if native is not None:
        return native
    A = get_rotation_matrix2d(center, -angle, scale)
    A = torch.cat([A[..., :2], A[..., 2:] + translations[..., None]], dim=-1)
    H = _to_homography(A)
    if sx is not None or sy is not None:
        H = H @ get_shear_matrix2d(center, sx, sy)
    return H
