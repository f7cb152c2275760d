"""TEST INFRASTRUCTURE ONLY: `emulated_device()` lets the *unchanged* Python host layer of kornia_amd and the *unchanged*
GPU parity tests run in a container without a GPU, against the host build of the kernels (build_emu.py).

Inside the context:
  * kornia_amd._native hands out the emulated library, accepts host tensors and passes a null stream;
  * "cuda" means host memory for tensor factories, .cuda(), .to(...): a TorchFunctionMode rewrites the device.
Outside it nothing changes: the package has no CPU path (tests/test_abi_and_host.py::test_no_cpu_fallback)."""
from __future__ import annotations

import contextlib
import os
import sys

import torch
from torch.overrides import TorchFunctionMode

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _is_cuda_dev(d) -> bool:
    if isinstance(d, torch.device):
        return d.type == "cuda"
    if isinstance(d, str):
        return d.startswith("cuda")
    return False


class _CudaIsHost(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        name = getattr(func, "__name__", "")
        to_dev_kw = name == "to" and _is_cuda_dev(kwargs.get("device"))
        if _is_cuda_dev(kwargs.get("device")):
            kwargs["device"] = "cpu"
        # a transfer to the device is a new tensor (same autograd semantics as the real copy), never an alias
        if name == "cuda" and args and isinstance(args[0], torch.Tensor):
            return args[0].clone()
        moved = False
        if name == "to" and len(args) >= 2 and _is_cuda_dev(args[1]):
            args = (args[0], torch.device("cpu"), *args[2:])
            moved = True
        moved = moved or to_dev_kw
        out = func(*args, **kwargs)
        if name == "to" and moved and out is args[0]:
            out = out.clone()
        return out


class _HostStream:
    cuda_stream = 0


@contextlib.contextmanager
def emulated_device():
    import emu_lib

    from kornia_amd import _native as N

    h = emu_lib.lib()
    saved = {k: getattr(N, k) for k in ("lib", "_lib", "require_device", "on_device", "stream_ptr", "device_guard", "is_built")}
    saved_cuda = {k: getattr(torch.cuda, k) for k in ("synchronize", "current_stream", "manual_seed_all")}
    N._lib = h
    N.lib = lambda: h
    N.require_device = lambda t, name: None
    N.on_device = lambda t: True
    N.stream_ptr = lambda device: 0
    N.device_guard = lambda device: N._NO_GUARD
    N.is_built = lambda: True
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: _HostStream()
    torch.cuda.manual_seed_all = lambda *a, **k: None
    try:
        with _CudaIsHost():
            yield h
    finally:
        for k, v in saved.items():
            setattr(N, k, v)
        for k, v in saved_cuda.items():
            setattr(torch.cuda, k, v)
