"""crop_by_transform_mat / crop_by_boxes / crop_and_resize / center_crop on the native warps.

Reference behaviour mirrored: kornia/geometry/transform/crop2d.py:41-124 (crop_and_resize), :125-208 (center_crop),
:209-298 (crop_by_boxes), :299-404 (crop_by_transform_mat); box size from kornia/geometry/bbox.py infer_bbox_shape
(``x1 - x0 + 1``, ``y2 - y0 + 1``).  Boxes are four ``(x, y)`` corners in the order top-left, top-right, bottom-right,
bottom-left.  The homography is ``get_perspective_transform`` (one launch on HIP tensors), the resampling is
``warp_perspective`` / ``warp_affine``.
"""
from __future__ import annotations

from typing import Tuple

import torch

from .builders import get_perspective_transform
from .imgwarp import warp_affine, warp_perspective

__all__ = ["center_crop", "crop_and_resize", "crop_by_boxes", "crop_by_transform_mat"]


def crop_by_transform_mat(input_tensor: torch.Tensor, transform: torch.Tensor, out_size: Tuple[int, int], mode: str = "bilinear",
                          padding_mode: str = "zeros", align_corners: bool = True) -> torch.Tensor:
    """Crop with a source->destination transform: ``(B,2,3)`` takes ``warp_affine``, ``(B,3,3)`` ``warp_perspective``
    (with the destination-side reparametrisation the reference applies for ``align_corners=False``)."""
    T = transform.expand(input_tensor.shape[0], -1, -1).to(device=input_tensor.device, dtype=input_tensor.dtype)
    if transform.shape[-2:] == (2, 3):
        return warp_affine(input_tensor, T, out_size, mode=mode, padding_mode=padding_mode, align_corners=align_corners)
    h_out, w_out = out_size
    if not align_corners and (h_out == 1 or w_out == 1):
        return warp_affine(input_tensor, T[:, :2, :], out_size, mode=mode, padding_mode=padding_mode, align_corners=align_corners)
    if not align_corners:
        corr = torch.tensor([[w_out / (w_out - 1.0), 0.0, -0.5], [0.0, h_out / (h_out - 1.0), -0.5], [0.0, 0.0, 1.0]],
                            device=T.device, dtype=T.dtype)
        T = corr.unsqueeze(0) @ T
    return warp_perspective(input_tensor, T, out_size, mode=mode, padding_mode=padding_mode, align_corners=align_corners)


def crop_by_boxes(input_tensor: torch.Tensor, src_box: torch.Tensor, dst_box: torch.Tensor, mode: str = "bilinear",
                  padding_mode: str = "zeros", align_corners: bool = True, validate_boxes: bool = True) -> torch.Tensor:
    """Warp the quadrilaterals ``src_box`` (B,4,2) onto the axis-aligned ``dst_box`` (B,4,2); all crops of a batch share one size."""
    if input_tensor.dim() != 4:
        raise AssertionError(f"Only torch.Tensor with shape (B, C, H, W) supported. Got {input_tensor.shape}.")
    dst_trans_src = get_perspective_transform(src_box.to(input_tensor), dst_box.to(input_tensor))
    widths = dst_box[:, 1, 0] - dst_box[:, 0, 0] + 1
    heights = dst_box[:, 2, 1] - dst_box[:, 0, 1] + 1
    if not ((heights == heights[0]).all() and (widths == widths[0]).all()):
        raise AssertionError(f"Cropping height, width and depth must be exact same in a batch. Got height {heights} and width {widths}.")
    return crop_by_transform_mat(input_tensor, dst_trans_src, (int(heights[0].item()), int(widths[0].item())), mode=mode,
                                 padding_mode=padding_mode, align_corners=align_corners)


def _dst_box(size, like: torch.Tensor, n: int) -> torch.Tensor:
    dst_h, dst_w = size
    return torch.tensor([[[0, 0], [dst_w - 1, 0], [dst_w - 1, dst_h - 1], [0, dst_h - 1]]], device=like.device,
                        dtype=like.dtype).expand(n, -1, -1)


def _check_crop_args(input_tensor, size):
    if not isinstance(input_tensor, torch.Tensor):
        raise TypeError(f"Input torch.tensor type is not a torch.Tensor. Got {type(input_tensor)}")
    if not isinstance(size, (tuple, list)) or len(size) != 2:
        raise ValueError(f"Input size must be a tuple/list of length 2. Got {size}")
    if input_tensor.dim() != 4:
        raise AssertionError(f"Only torch.Tensor with shape (B, C, H, W) supported. Got {input_tensor.shape}.")


def crop_and_resize(input_tensor: torch.Tensor, boxes: torch.Tensor, size: Tuple[int, int], mode: str = "bilinear",
                    padding_mode: str = "zeros", align_corners: bool = True) -> torch.Tensor:
    """Extract the quadrilaterals ``boxes`` (B,4,2) and resample each to ``size`` = (h, w)."""
    _check_crop_args(input_tensor, size)
    if not isinstance(boxes, torch.Tensor):
        raise TypeError(f"Input boxes type is not a torch.Tensor. Got {type(boxes)}")
    points_src = boxes.to(input_tensor)
    return crop_by_boxes(input_tensor, points_src, _dst_box(size, input_tensor, points_src.shape[0]), mode, padding_mode, align_corners)


def center_crop(input_tensor: torch.Tensor, size: Tuple[int, int], mode: str = "bilinear", padding_mode: str = "zeros",
                align_corners: bool = True) -> torch.Tensor:
    """Crop the centred ``size`` = (h, w) window."""
    _check_crop_args(input_tensor, size)
    dst_h, dst_w = size
    src_h, src_w = input_tensor.shape[-2:]
    start_x, start_y = src_w / 2 - dst_w / 2, src_h / 2 - dst_h / 2
    end_x, end_y = start_x + dst_w - 1, start_y + dst_h - 1
    points_src = torch.tensor([[[start_x, start_y], [end_x, start_y], [end_x, end_y], [start_x, end_y]]],
                              device=input_tensor.device, dtype=input_tensor.dtype)
    return crop_by_boxes(input_tensor, points_src, _dst_box(size, input_tensor, 1), mode, padding_mode, align_corners)
