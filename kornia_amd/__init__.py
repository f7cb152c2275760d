"""kornia_amd - MI355X (gfx950 / CDNA4) native implementation of Kornia's geometric-warp and
separable-filter hot path, behind Kornia's own Python API.

    import kornia_amd as K
    y = K.filters.gaussian_blur2d(K.geometry.transform.warp_perspective(x, M, (512, 512)), (5, 5), (1.5, 1.5))

The ops run hand-written HIP kernels through a C-ABI shared library (``include/kornia_amd.h``,
``kornia_amd/lib/libkornia_amd.so``, built by ``python -m kornia_amd.build``).  There is no
PyTorch/CPU fallback: tensors must be on a HIP device and the library must be built.
"""
from . import augmentation, core, enhance, filters, geometry, graph
from ._native import NativeLibraryError, is_built, library_path
from .filters import (
    GaussianBlur2d,
    Sobel,
    SpatialGradient,
    filter2d,
    filter2d_separable,
    gaussian_blur2d,
    sobel,
    spatial_gradient,
)
from .geometry import (
    HomographyWarper,
    get_affine_matrix2d,
    get_perspective_transform,
    get_rotation_matrix2d,
    get_shear_matrix2d,
    grid_sample,
    homography_warp,
    normalize_homography,
    remap,
    transform_points,
    warp_affine,
    warp_grid,
    warp_perspective,
)

from .kornia_patch import is_patched, patch, unpatch

__version__ = "0.1.0"
