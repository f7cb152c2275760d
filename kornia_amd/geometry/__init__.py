from . import conversions, grid, linalg, transform
from .conversions import (
    convert_affinematrix_to_homography,
    convert_points_from_homogeneous,
    convert_points_to_homogeneous,
    normal_transform_pixel,
    normalize_homography,
    normalize_pixel_coordinates,
)
from .grid import create_meshgrid
from .linalg import transform_points
from .transform import (
    HomographyWarper,
    angle_to_rotation_matrix,
    deg2rad,
    get_affine_matrix2d,
    get_perspective_transform,
    get_rotation_matrix2d,
    get_shear_matrix2d,
    get_translation_matrix2d,
    grid_sample,
    homography_warp,
    remap,
    warp_affine,
    warp_grid,
    warp_perspective,
)
