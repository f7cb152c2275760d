"""Drop-in activation for an installed Kornia: ``kornia_amd.patch()`` rebinds Kornia's hot-path
functions - in every ``kornia.*`` module that imported them by value (augmentation, affwarp, crop2d,
pyramid, tracking, contrib ... ~40 modules) - to dispatchers that run the native
gfx950 kernels for HIP tensors and Kornia's own implementation otherwise.

The fallback inside a dispatcher is *Kornia's* function (CPU tensors, float64 where unsupported,
tracing / torch.compile / export, exotic arguments): it is not part of the kornia_amd product path,
which itself never falls back.  ``unpatch()`` restores the originals; each dispatcher keeps the
original reachable as ``__wrapped__`` (``torch.jit.script`` users must script that one).
"""
from __future__ import annotations

import functools
import importlib
import sys
from typing import Callable

import torch

from . import filters as _f
from . import geometry as _g

_NATIVE = {
    "kornia.geometry.transform.imgwarp": {
        "warp_perspective": _g.warp_perspective,
        "warp_affine": _g.warp_affine,
        "homography_warp": _g.homography_warp,
        "warp_grid": _g.warp_grid,
        "remap": _g.remap,
    },
    "kornia.geometry.linalg": {"transform_points": _g.transform_points},
    "kornia.geometry.conversions": {"normalize_homography": _g.normalize_homography},
    "kornia.filters.filter": {"filter2d": _f.filter2d, "filter2d_separable": _f.filter2d_separable},
    "kornia.filters.gaussian": {"gaussian_blur2d": _f.gaussian_blur2d},
    "kornia.filters.sobel": {"spatial_gradient": _f.spatial_gradient, "sobel": _f.sobel},
}
_SUPPORTED = (torch.float32, torch.float64, torch.bfloat16, torch.float16)
_patched: dict = {}  # id(original) -> (original, dispatcher)


def _use_native(args, kwargs) -> bool:
    if torch.jit.is_tracing() or torch.jit.is_scripting() or torch.compiler.is_compiling():
        return False
    tensors = [a for a in list(args) + list(kwargs.values()) if isinstance(a, torch.Tensor)]
    if not tensors:
        return False
    return all(t.is_cuda and t.dtype in _SUPPORTED for t in tensors if t.is_floating_point()) and any(t.is_cuda for t in tensors)


def _dispatcher(original: Callable, native: Callable) -> Callable:
    @functools.wraps(original)
    def wrapper(*args, **kwargs):
        if _use_native(args, kwargs):
            return native(*args, **kwargs)
        return original(*args, **kwargs)

    wrapper.__wrapped__ = original
    wrapper.__kornia_amd_native__ = native
    return wrapper


def patch() -> int:
    """Activate the native path inside Kornia. Returns the number of rebound module attributes."""
    if _patched:
        return 0
    import kornia  # noqa: F401 - must be importable

    for mod_name, table in _NATIVE.items():
        mod = importlib.import_module(mod_name)
        for name, native in table.items():
            original = getattr(mod, name)
            _patched[id(original)] = (original, _dispatcher(original, native))
    count = 0
    for mod_name, mod in list(sys.modules.items()):
        if mod is None or not (mod_name == "kornia" or mod_name.startswith("kornia.")):
            continue
        for attr, value in list(vars(mod).items()):
            hit = _patched.get(id(value))
            if hit is not None and value is hit[0]:
                setattr(mod, attr, hit[1])
                count += 1
    return count


def unpatch() -> int:
    """Restore Kornia's own functions."""
    if not _patched:
        return 0
    by_wrapper = {id(w): o for o, w in _patched.values()}
    count = 0
    for mod_name, mod in list(sys.modules.items()):
        if mod is None or not (mod_name == "kornia" or mod_name.startswith("kornia.")):
            continue
        for attr, value in list(vars(mod).items()):
            orig = by_wrapper.get(id(value))
            if orig is not None:
                setattr(mod, attr, orig)
                count += 1
    _patched.clear()
    return count


def is_patched() -> bool:
    return bool(_patched)
