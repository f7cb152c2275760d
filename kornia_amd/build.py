"""Builds ``kornia_amd/lib/libkornia_amd.so`` for gfx950 with hipcc (no GPU required).

Usage: ``python -m kornia_amd.build [--force]`` or ``kornia_amd.build.build()``.
The library is built in-tree so that it travels with a source snapshot; it is git-ignored.
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
OBJ_DIR = os.path.join(_PKG, "lib", "obj")
LIB_PATH = os.path.join(_PKG, "lib", "libkornia_amd.so")

ARCH = "gfx950"
# -ffp-contract=off is part of the numerics contract (csrc/km_common.h): fused multiply-adds
# appear only where the source asks for them, and fp32 divide/sqrt stay correctly rounded.
HIPCC_FLAGS = [
    f"--offload-arch={ARCH}",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall",
    "-Wno-unused-function",
]


# per-file additions to HIPCC_FLAGS (the reason is in the header of each file)
FILE_FLAGS = {"km_warp_cubic.hip": ["-fno-slp-vectorize"], "km_warp_gm_box.hip": ["-fno-slp-vectorize"], "km_warp_bwd_fused.hip": ["-fno-slp-vectorize"], "km_warp_blur.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (ROCm toolchain is required to build kornia_amd)")
    return exe


def sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(".h"))


def _compile(src: str, force: bool, verbose: bool) -> tuple[str, bool]:
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[: -len(".hip")] + ".o")
    newest = max(os.path.getmtime(src), _headers_mtime(), os.path.getmtime(__file__))
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    cmd = [_hipcc(), *HIPCC_FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{res.stdout}\n{res.stderr}")
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)
    return obj, True


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every ``csrc/*.hip`` for gfx950 and link the shared library. Returns its path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, verbose), srcs))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or not os.path.exists(LIB_PATH):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
