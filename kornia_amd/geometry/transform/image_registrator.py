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

from typing import Any, Callable, Optional, Union

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
        acc = torch.zeros(B, 11, device=dev, dtype=torch.float64)  # per image: loss sum, selected count, d sum / d mat
        with N.device_guard(dev):
            N.check(lib.km_warp_masked_loss(x.data_ptr(), d.data_ptr(), m.data_ptr(), acc.data_ptr(), B, C, H, W, h, w, B_M,
                                            COORD_HOMOGRAPHY, norm, align, kind, float(threshold), N.dtype_code(x.dtype),
                                            N.stream_ptr(dev)), "km_warp_masked_loss")
        total = acc[:, :2].sum(0)
        gsum = acc[:, 2:] if B_M == B else acc[:, 2:].sum(0, keepdim=True)
        ctx.save_for_backward(gsum, total)
        ctx.mat_shape, ctx.mat_dtype = mat.shape, mat.dtype
        return (total[0] / total[1]).to(src.dtype)  # 0 / 0 = nan when nothing is selected, like the mean of an empty selection

    @staticmethod
    def backward(ctx, gout):
        gsum, total = ctx.saved_tensors
        gm = (gsum / total[1]).view(-1, 3, 3) * gout.to(torch.float64)
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
        self.reset_model()

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.model})"

    def reset_model(self) -> None:
        torch.nn.init.eye_(self.model)

    def forward(self) -> torch.Tensor:
        return torch.unsqueeze(self.model / self.model[2, 2], dim=0)

    def forward_inverse(self) -> torch.Tensor:
        return torch.unsqueeze(torch.inverse(self.model), dim=0)


class Similarity(BaseModel):
    """Rotation / scale / shift, each either optimised or held at its neutral value."""

    def __init__(self, rotation: bool = True, scale: bool = True, shift: bool = True) -> None:
        super().__init__()
        if rotation:
            self.rot = nn.Parameter(torch.zeros(1))
        else:
            self.register_buffer("rot", torch.zeros(1))
        if shift:
            self.shift = nn.Parameter(torch.zeros(1, 2, 1))
        else:
            self.register_buffer("shift", torch.zeros(1, 2, 1))
        if scale:
            self.scale = nn.Parameter(torch.ones(1))
        else:
            self.register_buffer("scale", torch.ones(1))
        self.reset_model()

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(angle = {self.rot},               \n shift={self.shift}, \n scale={self.scale})"

    def reset_model(self) -> None:
        torch.nn.init.zeros_(self.rot)
        torch.nn.init.zeros_(self.shift)
        torch.nn.init.ones_(self.scale)

    def forward(self) -> torch.Tensor:
        rot = self.scale * angle_to_rotation_matrix(self.rot)
        return convert_affinematrix_to_homography(torch.cat([rot, self.shift], dim=2))

    def forward_inverse(self) -> torch.Tensor:
        return torch.inverse(self.forward())


class ImageRegistrator(nn.Module):
    r"""Coarse-to-fine gradient descent on a geometric model that warps ``src_img`` onto ``dst_img``.

    Same arguments, defaults and return values as the reference's class.  ``register`` keeps the reference's control flow
    (one ``loss.item()`` per iteration for the tolerance test)."""

    known_models = ["homography", "similarity", "translation", "scale", "rotation"]

    def __init__(self, model_type: Union[str, BaseModel] = "homography", optimizer: type = optim.Adam, loss_fn: Callable[..., torch.Tensor] = F.l1_loss,
                 pyramid_levels: int = 5, lr: float = 1e-3, num_iterations: int = 100, tolerance: float = 1e-4, warper: Optional[type] = None,
                 allow_shape_mismatch: bool = False) -> None:
        super().__init__()
        if not isinstance(model_type, str):
            if warper is None:
                raise ValueError("You must supply warper together with custom model")
            self.warper = warper
            self.model = model_type
        elif model_type.lower() == "homography":
            self.warper, self.model = HomographyWarper, Homography()
        elif model_type.lower() == "similarity":
            self.warper, self.model = HomographyWarper, Similarity(True, True, True)
        elif model_type.lower() == "translation":
            self.warper, self.model = HomographyWarper, Similarity(False, False, True)
        elif model_type.lower() == "rotation":
            self.warper, self.model = HomographyWarper, Similarity(True, False, False)
        elif model_type.lower() == "scale":
            self.warper, self.model = HomographyWarper, Similarity(False, True, False)
        else:
            raise ValueError(f"{model_type} is not supported. Try {self.known_models}")
        self.pyramid_levels = pyramid_levels
        self.optimizer = optimizer
        self.lr = lr
        self.loss_fn = loss_fn
        self.num_iterations = num_iterations
        self.tolerance = tolerance
        self.allow_shape_mismatch = allow_shape_mismatch

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
        _height, _width = img_dst.shape[-2:]
        warper = self.warper(_height, _width)
        img_src_to_dst = warper(img_src, transform_model)
        loss = self.loss_fn(img_src_to_dst, img_dst, reduction="none")
        ones_tensor = warper(torch.ones_like(img_src), transform_model)
        return loss.masked_select(ones_tensor > 0.9).mean()

    def reset_model(self) -> None:
        self.model.reset_model()

    def register(self, src_img: torch.Tensor, dst_img: torch.Tensor, verbose: bool = False, output_intermediate_models: bool = False):
        r"""Estimate the transformation that warps ``src_img`` into ``dst_img``; returns the model matrix ((1,3,3) for the named
        models), and the per-level models as well when ``output_intermediate_models``."""
        self.reset_model()
        if src_img.shape != dst_img.shape:
            if not self.allow_shape_mismatch:
                raise ValueError(f"Cannot register images of different shapes {src_img.shape} {dst_img.shape}. Consider setting `allow_shape_mismatch = True`")
            src_img = F.interpolate(src_img, size=dst_img.shape[-2:], mode="bilinear", align_corners=False)
        _opt_args: dict[str, Any] = {"lr": self.lr}
        opt = self.optimizer(self.model.parameters(), **_opt_args)
        img_src_pyr = build_pyramid(src_img, self.pyramid_levels)[::-1]
        img_dst_pyr = build_pyramid(dst_img, self.pyramid_levels)[::-1]
        prev_loss = 1e10
        aux_models = []
        if len(img_dst_pyr) != len(img_src_pyr):
            raise ValueError("Cannot register images of different sizes")
        for img_src_level, img_dst_level in zip(img_src_pyr, img_dst_pyr):
            for i in range(self.num_iterations):
                opt.zero_grad()
                loss = self.get_single_level_loss(img_src_level, img_dst_level, self.model())
                loss = loss + self.get_single_level_loss(img_dst_level, img_src_level, self.model.forward_inverse())
                current_loss = loss.item()
                if abs(current_loss - prev_loss) < self.tolerance:
                    break
                prev_loss = current_loss
                loss.backward()
                if verbose and (i % 10 == 0):
                    print(f"Loss = {current_loss:.4f}, iter={i}")
                opt.step()
            if output_intermediate_models:
                aux_models.append(self.model().clone().detach())
        if output_intermediate_models:
            return self.model(), aux_models
        return self.model()

    def warp_src_into_dst(self, src_img: torch.Tensor) -> torch.Tensor:
        _height, _width = src_img.shape[-2:]
        return self.warper(_height, _width)(src_img, self.model())

    def warp_dst_inro_src(self, dst_img: torch.Tensor) -> torch.Tensor:  # the reference's spelling
        _height, _width = dst_img.shape[-2:]
        return self.warper(_height, _width)(dst_img, self.model.forward_inverse())
