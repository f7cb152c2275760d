#!/usr/bin/env python3
"""Compares the gfx950 device ISA of the kernel sources between a git revision and the working tree, kernel by kernel
(no GPU needed):   profiles/isa_diff.py [REV=HEAD] [file.hip ...]
Local label numbers and comments are normalised, so adding a kernel to a file does not show up as a change of the others.
A refactor that must not change code generation prints "changed: none" for every file."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kornia_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "--cuda-device-only", "-S"]


def kernels(asm_path):
    out, cur = {}, None
    for line in open(asm_path):
        if line.strip().startswith(";") or "__hip_cuid" in line:
            continue
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None:
            if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
                cur = None
                continue
            line = re.sub(r"\.L([A-Za-z_]+)\d+_(\d+)", r".L\1_\2", line.rstrip())
            out[cur].append(re.sub(r";.*$", "", line).rstrip())
    return out


def main():
    args = sys.argv[1:]
    rev = args[0] if args and not args[0].endswith(".hip") else "HEAD"
    files = [a for a in args if a.endswith(".hip")] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(f"git archive {rev} kornia_amd/csrc | tar -x -C {tmp}", shell=True, cwd=ROOT, check=True)
        for f in files:
            old_src = os.path.join(tmp, "kornia_amd", "csrc", f)
            if not os.path.exists(old_src):
                print(f"{f}: new file")
                continue
            a, b = os.path.join(tmp, f + ".a.s"), os.path.join(tmp, f + ".b.s")
            procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", *FLAGS, src, "-o", dst], stderr=subprocess.DEVNULL)
                     for src, dst in ((old_src, a), (os.path.join(CSRC, f), b))]
            if any(p.wait() for p in procs):
                print(f"{f}: compile failed")
                continue
            ka, kb = kernels(a), kernels(b)
            changed = [k for k in ka if k in kb and ka[k] != kb[k]]
            print(f"{f}: {len(ka)} kernels, changed: {changed or 'none'}, removed: {[k for k in ka if k not in kb] or 'none'}, "
                  f"added: {[k for k in kb if k not in ka] or 'none'}")


if __name__ == "__main__":
    main()
