"""ctypes binding of the C-ABI library (``include/kornia_amd.h``).

The product path has NO fallback: if the shared library is missing or cannot be loaded, or a
tensor does not live on a HIP device, the ops raise.  PyTorch is used for device memory, streams
and autograd only.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_void_p
from typing import Optional

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KORNIA_AMD_LIB") or os.path.join(_PKG, "lib", "libkornia_amd.so")  # env override: A/B builds
ABI_VERSION = 3

KM_F32, KM_F64, KM_BF16, KM_F16 = 0, 1, 2, 3
_DTYPE_CODES = {torch.float32: KM_F32, torch.float64: KM_F64, torch.bfloat16: KM_BF16, torch.float16: KM_F16}


class NativeLibraryError(RuntimeError):
    """The HIP extension is missing, stale, or reported an error."""


_P = c_void_p
_I = c_int
# name -> argtypes; every entry point returns int (0 ok, <0 bad argument, >0 hipError_t)
_PROTOTYPES = {
    "km_homography_chain_fwd": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "km_homography_chain_bwd": [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "km_perspective_transform_fwd": [_P, _P, _P, _I, _I, _P],
    "km_affine_matrix2d_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "km_warp2d_fwd": [_P, _P, _P] + [_I] * 12 + [_P, _I, _P],
    "km_warp2d_bwd": [_P, _P, _P, _P, _P] + [_I] * 12 + [_P, _I, _P],
    "km_warp2d_bwd_ws": [_P, _P, _P, _P, _P] + [_I] * 12 + [_P, _I, _P, ctypes.c_longlong, _P],
    "km_warp2d_bwd_workspace_bytes": [_I] * 9,
    "km_warp2d_bwd_needs_zero_init": [_I, _I, _I],
    "km_grid_sample2d_fwd": [_P, _P, _P] + [_I] * 10 + [_I, _P],
    "km_grid_sample2d_bwd": [_P, _P, _P, _P, _P] + [_I] * 10 + [_I, _P],
    "km_filter2d_fwd": [_P, _P, _P] + [_I] * 10 + [_P],
    "km_filter2d_bwd_input": [_P, _P, _P] + [_I] * 10 + [_P],
    "km_filter2d_bwd_kernel": [_P, _P, _P] + [_I] * 10 + [_P],
    "km_filter2d_sep_fwd": [_P, _P, _P, _P] + [_I] * 10 + [_P],
    "km_filter2d_sep_bwd_input": [_P, _P, _P, _P] + [_I] * 10 + [_P],
    "km_spatial_gradient_fwd": [_P, _P, _P, _P] + [_I] * 6 + [c_double, _I, _P],
    "km_spatial_gradient_bwd": [_P, _P, _P] + [_I] * 6 + [_I, _P],
    "km_filter2d_sep_supported": [_I, _I, _I, _I],
    "km_color_jitter_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "km_pyrdown_fwd": [_P, _P] + [_I] * 9 + [_P],
    "km_resize_bilinear_fwd": [_P, _P] + [_I] * 8 + [_P],
    "km_resize_bilinear_bwd": [_P, _P] + [_I] * 8 + [_P],
    "km_warp_masked_loss": [_P, _P, _P, _P] + [_I] * 11 + [c_double, _I, _P],
    "km_warp_masked_loss_finish": [_P, _I, _I, _P, _P, _P, _P],
    "km_scale_f64": [_P, _P, _I, _P, _I, ctypes.c_longlong, _P],
    "km_stream_copy": [_P, _P, ctypes.c_longlong, _I, _P],
    "km_gaussian_taps_fwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "km_gaussian_taps_dtype_fwd": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "km_warp2d_fwd_masked": [_P, _P, _P, _P] + [_I] * 12 + [_P, _I, _P],
    "km_color_jitter_fwd_masked": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "km_affine_params_chain_fwd": [_P] * 10 + [_I] * 5 + [_P],
    "km_color_params_fwd": [_P] * 8 + [_I, _P],
    "km_color_params_ws_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "km_color_jitter_bwd": [_P] * 10 + [_I] * 5 + [_P],
    "km_select_samples_fwd": [_P, _P, _P, _P, _I, ctypes.c_longlong, _I, _P],
    "km_canny_nms_fwd": [_P, _P, _P, _I, _I, _I, c_double, c_double, c_double, _P],
    "km_canny_hysteresis_sweep": [_P, _P, _P, _I, _I, _I, _P],
    "km_warp2d_blur_fwd": [_P, _P, _P, _P, _P] + [_I] * 14 + [_P],
    "km_warp2d_blur_supported": [_I] * 10,
    "km_transform_points_fwd": [_P, _P, _P] + [_I] * 4 + [_I, _P],
    "km_transform_points_bwd": [_P, _P, _P, _P, _P] + [_I] * 4 + [_I, _P],
}

_lib: Optional[ctypes.CDLL] = None


def library_path() -> str:
    return LIB_PATH


def is_built() -> bool:
    return os.path.exists(LIB_PATH)


def lib() -> ctypes.CDLL:
    """Load (once) and return the C-ABI library; raises NativeLibraryError when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"kornia_amd native library not found at {LIB_PATH}. Build it with `python -m kornia_amd.build` "
            "(hipcc, --offload-arch=gfx950). There is no CPU/PyTorch fallback."
        )
    try:
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover - depends on the runtime environment
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    handle.km_abi_version.restype = c_int
    handle.km_last_error.restype = c_char_p
    got = handle.km_abi_version()
    if got != ABI_VERSION:
        raise NativeLibraryError(f"{LIB_PATH} has ABI version {got}, expected {ABI_VERSION}; rebuild it")
    for name, argtypes in _PROTOTYPES.items():
        fn = getattr(handle, name, None)
        if fn is None:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}; rebuild it")
        fn.argtypes = argtypes
        fn.restype = ctypes.c_longlong if name == "km_warp2d_bwd_workspace_bytes" else c_int
    handle.km_device_info.argtypes = [c_char_p, c_int]
    handle.km_device_info.restype = c_int
    handle.km_set_traversal.argtypes = [c_int]
    handle.km_set_traversal.restype = c_int
    handle.km_config_set.argtypes = [c_char_p, c_int]
    handle.km_config_set.restype = c_int
    handle.km_config_get.argtypes = [c_char_p]
    handle.km_config_get.restype = c_int
    _lib = handle
    return handle


def exported_symbols() -> list[str]:
    return ["km_abi_version", "km_last_error", "km_device_info", "km_set_traversal", "km_config_set", "km_config_get", *_PROTOTYPES.keys()]


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().km_last_error().decode("utf-8", "replace")
        raise NativeLibraryError(f"{what} failed (rc={rc}): {msg}")


def dtype_code(dtype: torch.dtype) -> int:
    try:
        return _DTYPE_CODES[dtype]
    except KeyError:
        raise TypeError(f"kornia_amd supports float32/float64/bfloat16/float16 tensors, got {dtype}") from None


def compute_dtype(dtype: torch.dtype) -> torch.dtype:
    """fp64 data is processed in fp64, everything else in fp32."""
    return torch.float64 if dtype == torch.float64 else torch.float32


def require_device(t: torch.Tensor, name: str) -> None:
    """The native path only runs on HIP tensors - fail loudly otherwise (no silent fallback)."""
    if not t.is_cuda:
        raise NativeLibraryError(
            f"kornia_amd: `{name}` must be a tensor on a HIP (cuda) device, got device={t.device}. "
            "The MI355X-native path has no CPU fallback."
        )


def on_device(t: torch.Tensor) -> bool:
    """True when ``t`` lives in HIP device memory (the only place the native kernels can read)."""
    return t.is_cuda


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device: torch.device) -> int:
    """The current HIP stream of ``device`` as the integer the C ABI takes.  Through torch's raw accessor when it exists (0.3 us; the public
    ``torch.cuda.current_stream(device).cuda_stream`` builds a Stream object: 5 us - six of them per call of a three-module augmentation
    pipeline, whose host share is what bounds it: profiles/r06/run20_*)."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    __slots__ = ()

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(device: torch.device):
    """``torch.cuda.device(device)`` only when it is not already the current device (the common case costs ~1 us
    instead of two device switches per native call)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def flags(t, device, n=None):
    """A per-sample / per-stage switch as the contiguous uint8 array the kernels read (zero / non-zero): a bool tensor is reinterpreted
    in place and a uint8 tensor - what the parameter kernels write - is taken as it is (no launch), anything else goes through ``!= 0``."""
    import torch
    t = t.detach().to(device=device).reshape(-1)
    if n is not None and t.numel() != n:
        raise ValueError(f"expected {n} switch entries, got {t.numel()}")
    t = t.contiguous()
    if t.dtype == torch.uint8:
        return t
    return t.view(torch.uint8) if t.dtype == torch.bool else t.ne(0).view(torch.uint8)
