#!/bin/bash
# round 3, device run 6: (a) phases of the one-read backward's tile loop (s_memtime per wave, lib_prof); (b) forward: the hot path's stores were
# never streaming (the compile-time policy helper lost the non-temporal mark) - real streaming stores (default) against plain ones (lib_plainst);
# the box forward (KM_WARP_FWD_ALGO=box: source box of a 64 x 16 output tile through LDS) and its tile / pitch variants; (c) the step with each
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03
O=gpurun_out/r03/run6.txt
: > $O
V=$PWD/kornia_amd/lib/var
run() { echo "\$ $*" >> $O; timeout 300 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
KORNIA_AMD_LIB=$V/lib_prof.so run python profiles/time_bwd_phases.py
run python profiles/time_warp_kernels.py 20 fwd
KORNIA_AMD_LIB=$V/lib_plainst.so run python profiles/time_warp_kernels.py 20 fwd
KM_WARP_FWD_ALGO=box run python profiles/time_warp_kernels.py 20 fwd
KM_WARP_FWD_ALGO=box KORNIA_AMD_LIB=$V/lib_plainst.so run python profiles/time_warp_kernels.py 20 fwd
KM_WARP_FWD_ALGO=box KORNIA_AMD_LIB=$V/lib_boxth32.so run python profiles/time_warp_kernels.py 20 fwd
KM_WARP_FWD_ALGO=box KORNIA_AMD_LIB=$V/lib_boxp72.so run python profiles/time_warp_kernels.py 20 fwd
run python profiles/time_warp_kernels.py 20 fwd
KM_WARP_FWD_ALGO=box run python -m pytest tests/test_gpu_warp.py tests/test_gpu_config_parity.py tests/test_gpu_augmentation.py tests/test_gpu_golden.py -m gpu -x -q
run python -m pytest tests/test_gpu_warp.py tests/test_gpu_warp_fused.py -m gpu -x -q
run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
KORNIA_AMD_LIB=$V/lib_plainst.so run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
KM_WARP_FWD_ALGO=box run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
run python profiles/time_config3.py
KM_WARP_FWD_ALGO=box run python profiles/time_config3.py
grep -v "^{" $O | grep -v "^\s*$" | tail -80
grep "^{" $O | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['ms_per_step'], d['value'], {k:v['ms'] for k,v in d['ops'].items()})"
