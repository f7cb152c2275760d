from .affwarp import affine, rotate, scale, shear, translate
from .builders import (
    angle_to_rotation_matrix,
    deg2rad,
    get_affine_matrix2d,
    get_perspective_transform,
    get_rotation_matrix2d,
    get_shear_matrix2d,
    get_translation_matrix2d,
)
from .crop2d import center_crop, crop_and_resize, crop_by_boxes, crop_by_transform_mat
from .homography_warper import HomographyWarper
from .pyramid import PyrDown, PyrUp, ScalePyramid, build_laplacian_pyramid, build_pyramid, pyrdown, pyrup, resize_bilinear
from .image_registrator import BaseModel, Homography, ImageRegistrator, Similarity, masked_warp_loss
from .imgwarp import grid_sample, homography_warp, remap, warp_affine, warp_grid, warp_perspective
from .warp_blur import warp_affine_blur, warp_perspective_blur

__all__ = [
    "warp_affine_blur",
    "warp_perspective_blur",
    "BaseModel",
    "Homography",
    "ImageRegistrator",
    "Similarity",
    "masked_warp_loss",
    "HomographyWarper",
    "PyrDown",
    "PyrUp",
    "ScalePyramid",
    "build_laplacian_pyramid",
    "build_pyramid",
    "pyrdown",
    "pyrup",
    "resize_bilinear",
    "affine",
    "center_crop",
    "crop_and_resize",
    "crop_by_boxes",
    "crop_by_transform_mat",
    "rotate",
    "scale",
    "shear",
    "translate",
    "angle_to_rotation_matrix",
    "deg2rad",
    "get_affine_matrix2d",
    "get_perspective_transform",
    "get_rotation_matrix2d",
    "get_shear_matrix2d",
    "get_translation_matrix2d",
    "grid_sample",
    "homography_warp",
    "remap",
    "warp_affine",
    "warp_grid",
    "warp_perspective",
]
