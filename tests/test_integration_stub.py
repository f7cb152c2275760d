"""CPU: the reference-side binding INTEGRATION.md shows (section B: `kornia/core/_backend_amd.py`) is EXECUTED as written - the code
block is taken from the document - against the host build of the shipped kernels (tests/emu, same C ABI): the ABI version it asserts
is the library's, the argument lists of `km_homography_chain_fwd` / `km_warp2d_fwd` are the header's, the dtype codes are the
library's, and its forward returns what `kornia_amd.warp_perspective` returns, bit for bit."""
import ctypes
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):  # pragma: no cover
    pytest.skip("ROCm clang++ (host compiler of the emulated build) not found", allow_module_level=True)


def _stub_source() -> str:
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    hits = [b for b in blocks if b.lstrip().startswith("# kornia/core/_backend_amd.py")]
    assert len(hits) == 1, "INTEGRATION.md section B: the _backend_amd.py block was not found (or is no longer unique)"
    return hits[0]


def test_the_documented_binding_runs_against_the_shipped_abi(monkeypatch):
    import build_emu
    from _util import flagship_homographies
    from mode import emulated_device

    import kornia_amd as K
    from kornia_amd import _native as N

    monkeypatch.setenv("KORNIA_AMD_LIB", build_emu.build())
    ns = {"__name__": "kornia.core._backend_amd"}
    exec(compile(_stub_source(), "INTEGRATION.md:_backend_amd.py", "exec"), ns)
    # what the stub hard-codes is what the library and the package say
    assert ns["lib"]().km_abi_version() == N.ABI_VERSION == 3
    assert {dt: N.dtype_code(dt) for dt in ns["_DT"]} == ns["_DT"]
    with pytest.raises(RuntimeError):  # a refused call surfaces the library's message through the stub's check()
        ns["check"](ns["lib"]().km_warp2d_fwd(None, None, None, 1, 1, 4, 4, 4, 4, 1, 0, 1, 1, 0, 1, None, 0, None))

    g = torch.Generator().manual_seed(5)
    B, C, H, W, h, w = 2, 3, 40, 52, 36, 44
    x = torch.rand(B, C, H, W, generator=g)
    M = flagship_homographies(B, H, W, h, w, g, jitter=3.0)
    with emulated_device():  # (the stub asks torch for the current stream: a null stream on the host build)
        for mode, interp in (("bilinear", 1), ("nearest", 0), ("bicubic", 2)):
            for pad_name, pad in (("zeros", 0), ("border", 1), ("reflection", 2)):
                for align in (True, False):
                    y_stub = ns["_Warp2d"].apply(x, M, (h, w), 0, interp, pad, align)
                    y_pkg = K.warp_perspective(x, M, (h, w), mode=mode, padding_mode=pad_name, align_corners=align)
                    assert torch.equal(y_stub, y_pkg), (mode, pad_name, align)
