from .adjust import (
    adjust_brightness_accumulative,
    adjust_contrast_with_mean_subtraction,
    adjust_hue,
    adjust_saturation_with_gray_subtraction,
    color_jitter,
)

__all__ = [
    "adjust_brightness_accumulative",
    "adjust_contrast_with_mean_subtraction",
    "adjust_hue",
    "adjust_saturation_with_gray_subtraction",
    "color_jitter",
]
