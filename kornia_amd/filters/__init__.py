from .blur import (
    BoxBlur,
    Laplacian,
    UnsharpMask,
    box_blur,
    get_box_kernel1d,
    get_box_kernel2d,
    get_laplacian_kernel1d,
    get_laplacian_kernel2d,
    laplacian,
    unsharp_mask,
)
from .canny import Canny, canny
from .filter import filter2d, filter2d_separable
from .gaussian import GaussianBlur2d, gaussian_blur2d
from .kernels import (
    gaussian,
    get_gaussian_kernel1d,
    get_gaussian_kernel2d,
    get_spatial_gradient_kernel2d,
    normalize_kernel2d,
)
from .sobel import Sobel, SpatialGradient, sobel, spatial_gradient

# reference aliases (kornia/filters/filter.py:460-548)
def correlate2d(input, kernel, border_type="reflect", normalized=False, padding="same"):
    """filter2d with behaviour='corr' (kornia/filters/filter.py:460-502: no `behaviour` argument of its own)"""
    return filter2d(input, kernel, border_type, normalized, padding, behaviour="corr")


def convolve2d(input, kernel, border_type="reflect", normalized=False, padding="same"):
    """filter2d with behaviour='conv', the kernel flipped (kornia/filters/filter.py:505-548)"""
    return filter2d(input, kernel, border_type, normalized, padding, behaviour="conv")
