"""TEST INFRASTRUCTURE ONLY: builds tests/_build/libkornia_amd_emu.so - the shipped kernel sources (kornia_amd/csrc/*.hip),
unmodified except for the substitutions listed below, compiled for the host against tests/emu/hip/hip_runtime.h.
The package never imports this module and never loads that library (tests/test_abi_and_host.py checks the former)."""
from __future__ import annotations

import concurrent.futures
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "kornia_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "emu")
# KM_EMU_ASAN=1: a second build with AddressSanitizer (run the tests with the runtime preloaded, see tests/emu/asan.sh):
# every out-of-bounds read or write of a kernel relative to the tensors it was handed is reported
ASAN = os.environ.get("KM_EMU_ASAN", "") not in ("", "0")
# KM_EMU_UBSAN=1: UndefinedBehaviorSanitizer build (signed overflow in index arithmetic, misaligned typed accesses, bad shifts ...)
UBSAN = os.environ.get("KM_EMU_UBSAN", "") not in ("", "0")
# KM_EMU_DEFS="-DKMB_UPDOWN=1 ...": extra preprocessor definitions (variant builds of the kernels, as profiles/build_variant2.sh makes for the
# device), in a build directory of their own
DEFS = os.environ.get("KM_EMU_DEFS", "").split()
_TAG = ("_asan" if ASAN else ("_ubsan" if UBSAN else "")) + ("_" + "".join(c if c.isalnum() else "_" for c in "".join(DEFS)) if DEFS else "")
OUT = os.path.join(ROOT, "tests", "_build", "emu" + _TAG)
LIB = os.path.join(ROOT, "tests", "_build", f"libkornia_amd_emu{_TAG}.so")
CXX = "/opt/rocm/lib/llvm/bin/clang++"

# (file, device-only text, host text): everything else is compiled exactly as shipped.  A substitution that no longer
# matches fails the build, so the list cannot rot silently.
SUBSTITUTIONS = [
    # dynamic LDS: `extern __shared__` has no host spelling
    ("*", "extern __shared__ __attribute__((aligned(16))) char smem_raw[];", "char* smem_raw = emu::dyn_smem();"),
]


def _rt(name: str) -> str:
    return subprocess.run([CXX, f"-print-file-name=libclang_rt.{name}-x86_64.so"], capture_output=True, text=True).stdout.strip()


def _flags() -> list[str]:
    flags = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unknown-pragmas",
             "-Wno-pass-failed", "-I", EMU, "-I", CSRC]
    try:
        cpu = open("/proc/cpuinfo").read()
    except OSError:
        cpu = ""
    flags += DEFS
    if ASAN:
        flags += ["-fsanitize=address", "-fno-omit-frame-pointer", "-g1"]
    if UBSAN:
        flags += ["-fsanitize=undefined", "-fno-sanitize=vptr,function", "-fno-sanitize-recover=undefined", "-g1"]
    if " fma " in cpu:
        flags.append("-mfma")
    if " f16c " in cpu:
        flags.append("-mf16c")
    return flags


def _translate(name: str) -> str:
    path = os.path.join(CSRC, name)
    text = open(path).read()
    for which, old, new in SUBSTITUTIONS:
        if which not in ("*", name):
            continue
        if which == name and old not in text:
            raise RuntimeError(f"emulation substitution for {name} no longer matches: {old!r}")
        text = text.replace(old, new)
    out = os.path.join(OUT, name[: -len(".hip")] + ".cpp")
    body = f'#line 1 "{path}"\n' + text
    if not os.path.exists(out) or open(out).read() != body:
        open(out, "w").write(body)
    return out


def _compile(src: str) -> str:
    obj = src[: -len(".cpp")] + ".o"
    deps = [src, os.path.join(EMU, "hip", "hip_runtime.h"), __file__] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(d) for d in deps):
        return obj
    res = subprocess.run([CXX, *_flags(), "-c", src, "-o", obj], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"host build of {src} failed:\n{res.stdout}\n{res.stderr[-6000:]}")
    return obj


def build() -> str:
    if not os.path.exists(CXX):
        raise RuntimeError(f"{CXX} not found")
    os.makedirs(OUT, exist_ok=True)
    srcs = [_translate(f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    rt = os.path.join(OUT, "emu_runtime.cpp")
    body = f'#line 1 "{os.path.join(EMU, "emu_runtime.cpp")}"\n' + open(os.path.join(EMU, "emu_runtime.cpp")).read()
    if not os.path.exists(rt) or open(rt).read() != body:
        open(rt, "w").write(body)
    srcs.append(rt)
    with concurrent.futures.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        objs = list(ex.map(_compile, srcs))
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        # -Bsymbolic: the library's own references (km_set_error, the per-file *_run helpers, the hip* stand-ins) must bind
        # to its own definitions even when the real libkornia_amd.so / libamdhip64.so are already loaded RTLD_GLOBAL
        res = subprocess.run([CXX, "-shared", "-fPIC", "-Wl,-Bsymbolic", *(["-fsanitize=address", "-shared-libasan"] if ASAN else []),
                              *(["-fsanitize=undefined", "-fno-sanitize=vptr,function", "-shared-libsan", "-Wl,-rpath," + os.path.dirname(_rt("ubsan_standalone"))] if UBSAN else []),
                              "-o", LIB, *objs], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"link failed:\n{res.stderr[-4000:]}")
    return LIB


if __name__ == "__main__":
    print(build())
