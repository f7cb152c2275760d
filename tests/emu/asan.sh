#!/bin/bash
# TEST INFRASTRUCTURE ONLY: runs (a selection of) the emulated GPU parity tests with the host build of the kernels compiled
# under AddressSanitizer.  Tensors are allocated by torch through the sanitizer's malloc, so a kernel that reads or writes
# one element outside a tensor it was handed aborts with a report.   Usage: tests/emu/asan.sh [pytest args]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export KM_EMU_ASAN=1
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:alloc_dealloc_mismatch=0:new_delete_type_mismatch=0:halt_on_error=1:abort_on_error=1
cd "$ROOT"
LD_PRELOAD="$RT" python -m pytest tests/test_emulated_kernels.py -q -x -p no:cacheprovider "$@"
