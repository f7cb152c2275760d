from .homography_warper import HomographyWarper
from .imgwarp import homography_warp, warp_affine, warp_grid, warp_perspective

__all__ = ["HomographyWarper", "homography_warp", "warp_affine", "warp_grid", "warp_perspective"]
