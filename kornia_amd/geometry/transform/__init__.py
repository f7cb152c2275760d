from .builders import (
    angle_to_rotation_matrix,
    deg2rad,
    get_affine_matrix2d,
    get_perspective_transform,
    get_rotation_matrix2d,
    get_shear_matrix2d,
    get_translation_matrix2d,
)
from .homography_warper import HomographyWarper
from .imgwarp import grid_sample, homography_warp, remap, warp_affine, warp_grid, warp_perspective

__all__ = [
    "HomographyWarper",
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
