"""Optimisation-based image registration on the native kernels (the consumer of BASELINE config 5).

Reference behaviour mirrored: kornia/geometry/transform/image_registrator.py - BaseModel :33-68, Homography :71-101,
Similarity :104-151, ImageRegistrator :154-326.

What is native here
  * the pyramids of both images: one fused blur + decimation launch per level (pyramid.py -> km_pyrdown_fwd);
  * the loss of one level and its gradient wrt the model, :func:`masked_warp_loss`: ONE launch (km_warp_masked_loss,
    csrc/km_warp_loss.hip) instead of two warps, an elementwise loss, a compare, a ``masked_select`` (device sync) and a
    mean, plus the autograd walk back through all of them.  It applies when the warper is the HomographyWarper and the
    loss is ``F.l1_loss`` or ``F.mse_loss``; anything else runs the reference's composition on the native warps.
"""
from __future__ import annotations

from typing import Callable, Optional, Union

import torch
import torch.nn.functional as F
from torch import nn, optim

from ... import _native as N
from ..conversions import convert_affinematrix_to_homography
from .builders import angle_to_rotation_matrix
from .homography_warper import HomographyWarper
from .imgwarp import COORD_HOMOGRAPHY
from .pyramid import build_pyramid

__all__ = ["BaseModel", "Homography", "ImageRegistrator", "Similarity", "masked_warp_loss"]

_LOSS_KIND = {"l1": 0, "mse": 1}


class _MaskedWarpLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, dst, mat, kind: int, align: int, norm: int, threshold: float):
        lib = N.lib()
        dev = src.device
        x, d = src.detach().contiguous(), dst.detach().contiguous()
        m = mat.detach().to(device=dev, dtype=torch.float32).contiguous().view(-1, 9)
        B, C, H, W = x.shape
        h, w = d.shape[-2:]
        B_M = m.shape[0]
        if B == 0:  # the mean of an empty selection: NaN, and no gradient (what the reference's masked_select(...).mean() gives for an empty batch)
            ctx.save_for_backward(torch.zeros(B_M, 9, device=dev, dtype=torch.float64))
            ctx.mat_shape, ctx.mat_dtype = mat.shape, mat.dtype
            return torch.full((), float("nan"), device=dev, dtype=src.dtype)
        acc = torch.zeros(B, 11, device=dev, dtype=torch.float64)  # per image: loss sum, selected count, d sum / d mat
        with N.device_guard(dev):
            N.check(lib.km_warp_masked_loss(x.data_ptr(), d.data_ptr(), m.data_ptr(), acc.data_ptr(), B, C, H, W, h, w, B_M,
                                            COORD_HOMOGRAPHY, norm, align, kind, float(threshold), N.dtype_code(x.dtype),
                                            N.stream_ptr(dev)), "km_warp_masked_loss")
        # the sums over the batch, the mean and the unit gradient: one small launch (was ~nine torch ops around an 86 us kernel)
        loss64 = torch.empty((), device=dev, dtype=torch.float64)
        loss32 = torch.empty((), device=dev, dtype=torch.float32) if src.dtype == torch.float32 else None  # (0-dim, returned as is: a view would forbid the caller's in-place ops)
        gm_unit = torch.empty(B_M, 9, device=dev, dtype=torch.float64)
        with N.device_guard(dev):
            N.check(lib.km_warp_masked_loss_finish(acc.data_ptr(), B, B_M, loss64.data_ptr(), N.ptr(loss32), gm_unit.data_ptr(), N.stream_ptr(dev)),
                    "km_warp_masked_loss_finish")
        ctx.save_for_backward(gm_unit)
        ctx.mat_shape, ctx.mat_dtype = mat.shape, mat.dtype
        # 0 / 0 = nan when nothing is selected, like the mean of an empty selection
        return loss32 if loss32 is not None else loss64.to(src.dtype)

    @staticmethod
    def backward(ctx, gout):
        (gm_unit,) = ctx.saved_tensors
        if (gout.dtype in (torch.float32, torch.float64) and ctx.mat_dtype in (torch.float32, torch.float64) and gout.numel() == 1
                and N.on_device(gout)):
            # (d loss / d mat) * grad_output in fp64, rounded once to the matrix dtype: one launch
            dev = gm_unit.device
            g1 = gout.detach().contiguous()
            gm = torch.empty(gm_unit.shape, device=dev, dtype=ctx.mat_dtype)
            with N.device_guard(dev):
                N.check(N.lib().km_scale_f64(gm_unit.data_ptr(), g1.data_ptr(), N.dtype_code(g1.dtype), gm.data_ptr(), N.dtype_code(ctx.mat_dtype),
                                             gm_unit.numel(), N.stream_ptr(dev)), "km_scale_f64")
            return None, None, gm.view(ctx.mat_shape), None, None, None, None
        gm = gm_unit.view(-1, 3, 3) * gout.to(torch.float64)
        return None, None, gm.to(ctx.mat_dtype).view(ctx.mat_shape), None, None, None, None


def masked_warp_loss(src: torch.Tensor, dst: torch.Tensor, src_homo_dst: torch.Tensor, loss: str = "l1", align_corners: bool = False,
                     normalized_coordinates: bool = True, threshold: Optional[float] = 0.9) -> torch.Tensor:
    r"""``loss_fn(homography_warp(src, H), dst, reduction='none').masked_select(homography_warp(ones, H) > threshold).mean()``
    as one launch, differentiable wrt ``src_homo_dst`` ((B,3,3) or (1,3,3), destination->source in normalised coordinates).

    ``loss``: ``'l1'`` or ``'mse'``.  The images are constants of the optimisation: no gradient is produced for them.
    ``threshold=None`` drops the mask: the result is ``loss_fn(homography_warp(src, H), dst)`` with the default mean
    reduction over every element (BASELINE config 5's learned-homography step: forward and ``H.grad`` in one launch)."""
    if loss not in _LOSS_KIND:
        raise ValueError(f"loss must be one of {sorted(_LOSS_KIND)}, got {loss!r}")
    if not (isinstance(src, torch.Tensor) and isinstance(dst, torch.Tensor) and src.dim() == 4 and dst.dim() == 4):
        raise ValueError("src and dst must be BxCxHxW tensors")
    if src.shape[:2] != dst.shape[:2] or src.dtype != dst.dtype:
        raise ValueError(f"src and dst must agree in batch, channels and dtype, got {src.shape} {src.dtype} / {dst.shape} {dst.dtype}")
    if not (src_homo_dst.dim() == 3 and src_homo_dst.shape[-2:] == (3, 3) and src_homo_dst.shape[0] in (1, src.shape[0])):
        raise ValueError(f"src_homo_dst must be a Bx3x3 or 1x3x3 tensor, got {src_homo_dst.shape}")
    if torch.is_grad_enabled() and (src.requires_grad or dst.requires_grad):
        raise RuntimeError("masked_warp_loss differentiates wrt the homography only; detach the images")
    N.require_device(src, "src")
    N.require_device(dst, "dst")
    if threshold is None:
        threshold = -1.0  # the warped ones image is >= 0 everywhere
    return _MaskedWarpLoss.apply(src, dst, src_homo_dst, _LOSS_KIND[loss], int(bool(align_corners)), int(bool(normalized_coordinates)), threshold)


class BaseModel(nn.Module):
    """A learnable geometric model: ``forward()`` gives the matrix, ``forward_inverse()`` the matrix of the opposite direction."""

    def reset_model(self) -> None:
        raise NotImplementedError

    def forward(self) -> torch.Tensor:
        raise NotImplementedError

    def forward_inverse(self) -> torch.Tensor:
        raise NotImplementedError


class Homography(BaseModel):
    """3x3 matrix, 8 degrees of freedom; ``forward()`` is (1,3,3) normalised by its last entry."""

    def __init__(self) -> None:
        super().__init__()
        self.model = nn.Parameter(torch.eye(3))

    def __repr__(self) -> str:
        return f"{type(self).__name__}({self.model})"

    def reset_model(self) -> None:
        nn.init.eye_(self.model)

    def forward(self) -> torch.Tensor:
        return (self.model / self.model[2, 2])[None]

    def forward_inverse(self) -> torch.Tensor:
        return torch.inverse(self.model)[None]


class Similarity(BaseModel):
    """Rotation / scale / shift, each either optimised or held at its neutral value."""

    _NEUTRAL = {"rot": lambda: torch.zeros(1), "shift": lambda: torch.zeros(1, 2, 1), "scale": lambda: torch.ones(1)}

    def __init__(self, rotation: bool = True, scale: bool = True, shift: bool = True) -> None:
        super().__init__()
        for name, learn in (("rot", rotation), ("shift", shift), ("scale", scale)):
            value = self._NEUTRAL[name]()
            if learn:
                setattr(self, name, nn.Parameter(value))
            else:
                self.register_buffer(name, value)

    def __repr__(self) -> str:
        return f"{type(self).__name__}(angle = {self.rot},               \n shift={self.shift}, \n scale={self.scale})"

    def reset_model(self) -> None:
        with torch.no_grad():
            for name, make in self._NEUTRAL.items():
                getattr(self, name).copy_(make())

    def forward(self) -> torch.Tensor:
        linear = self.scale * angle_to_rotation_matrix(self.rot)
        return convert_affinematrix_to_homography(torch.cat([linear, self.shift], dim=2))

    def forward_inverse(self) -> torch.Tensor:
        return torch.inverse(self.forward())


# model_type string -> which of (rotation, scale, shift) a Similarity optimises; "homography" is the full 3x3 matrix
_NAMED_MODELS = {"similarity": (True, True, True), "translation": (False, False, True), "rotation": (True, False, False), "scale": (False, True, False)}


class ImageRegistrator(nn.Module):
    r"""Coarse-to-fine gradient descent on a geometric model that warps ``src_img`` onto ``dst_img``.

    Same arguments, defaults and return values as the reference's class; ``register`` keeps the reference's stopping rule
    (one ``loss.item()`` per iteration for the tolerance test)."""

    known_models = ["homography", "similarity", "translation", "scale", "rotation"]

    def __init__(self, model_type: Union[str, BaseModel] = "homography", optimizer: type = optim.Adam, loss_fn: Callable[..., torch.Tensor] = F.l1_loss,
                 pyramid_levels: int = 5, lr: float = 1e-3, num_iterations: int = 100, tolerance: float = 1e-4, warper: Optional[type] = None,
                 allow_shape_mismatch: bool = False) -> None:
        super().__init__()
        if isinstance(model_type, str):
            key = model_type.lower()
            if key not in self.known_models:
                raise ValueError(f"{model_type} is not supported. Try {self.known_models}")
            self.model = Homography() if key == "homography" else Similarity(*_NAMED_MODELS[key])
            self.warper = HomographyWarper
        else:
            if warper is None:
                raise ValueError("You must supply warper together with custom model")
            self.model, self.warper = model_type, warper
        self.pyramid_levels, self.num_iterations = pyramid_levels, num_iterations
        self.optimizer, self.lr, self.loss_fn = optimizer, lr, loss_fn
        self.tolerance, self.allow_shape_mismatch = tolerance, allow_shape_mismatch

    # ---- one level ---------------------------------------------------------------------------------------------------
    def _fused_loss_kind(self, img_src: torch.Tensor, img_dst: torch.Tensor, transform_model: torch.Tensor) -> Optional[str]:
        kind = "l1" if self.loss_fn is F.l1_loss else ("mse" if self.loss_fn is F.mse_loss else None)
        ok = (
            kind is not None and self.warper is HomographyWarper and img_src.dim() == 4 and N.on_device(img_src) and N.on_device(img_dst)
            and img_src.dtype in (torch.float32, torch.bfloat16, torch.float16) and img_src.shape[-1] >= 2
            and transform_model.dim() == 3 and transform_model.shape[-2:] == (3, 3) and transform_model.shape[0] in (1, img_src.shape[0])
            and not (torch.is_grad_enabled() and (img_src.requires_grad or img_dst.requires_grad))
        )
        return kind if ok else None

    def get_single_level_loss(self, img_src: torch.Tensor, img_dst: torch.Tensor, transform_model: torch.Tensor) -> torch.Tensor:
        """Warp ``img_src`` onto ``img_dst`` with ``transform_model`` and return the loss over the pixels the warp covers."""
        if img_src.shape != img_dst.shape:
            raise ValueError(f"Cannot register images of different shapes                             {img_src.shape} {img_dst.shape:} ")
        kind = self._fused_loss_kind(img_src, img_dst, transform_model)
        if kind is not None:
            return masked_warp_loss(img_src, img_dst, transform_model, kind)
        # any other warper / loss: the reference's composition on the native warps
        warp = self.warper(*img_dst.shape[-2:])
        covered = warp(torch.ones_like(img_src), transform_model) > 0.9
        return self.loss_fn(warp(img_src, transform_model), img_dst, reduction="none").masked_select(covered).mean()

    def _symmetric_loss(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return self.get_single_level_loss(a, b, self.model()) + self.get_single_level_loss(b, a, self.model.forward_inverse())

    def reset_model(self) -> None:
        self.model.reset_model()

    # ---- the whole pyramid ---------------------------------------------------------------------------------------------
    def register(self, src_img: torch.Tensor, dst_img: torch.Tensor, verbose: bool = False, output_intermediate_models: bool = False):
        r"""Estimate the transformation that warps ``src_img`` into ``dst_img``; returns the model matrix ((1,3,3) for the named
        models), and the per-level models as well when ``output_intermediate_models``."""
        self.reset_model()
        if src_img.shape != dst_img.shape:
            if not self.allow_shape_mismatch:
                raise ValueError(f"Cannot register images of different shapes {src_img.shape} {dst_img.shape}. Consider setting `allow_shape_mismatch = True`")
            src_img = F.interpolate(src_img, size=dst_img.shape[-2:], mode="bilinear", align_corners=False)
        opt = self.optimizer(self.model.parameters(), lr=self.lr)
        coarse_to_fine = list(zip(reversed(build_pyramid(src_img, self.pyramid_levels)), reversed(build_pyramid(dst_img, self.pyramid_levels))))
        last = 1e10  # carried across levels, like the reference
        per_level = []
        for level_src, level_dst in coarse_to_fine:
            for it in range(self.num_iterations):
                opt.zero_grad()
                loss = self._symmetric_loss(level_src, level_dst)
                value = loss.item()
                if abs(value - last) < self.tolerance:
                    break
                last = value
                loss.backward()
                if verbose and it % 10 == 0:
                    print(f"Loss = {value:.4f}, iter={it}")
                opt.step()
            if output_intermediate_models:
                per_level.append(self.model().detach().clone())
        return (self.model(), per_level) if output_intermediate_models else self.model()

    def warp_src_into_dst(self, src_img: torch.Tensor) -> torch.Tensor:
        return self.warper(*src_img.shape[-2:])(src_img, self.model())

    def warp_dst_inro_src(self, dst_img: torch.Tensor) -> torch.Tensor:  # the reference's spelling
        return self.warper(*dst_img.shape[-2:])(dst_img, self.model.forward_inverse())
