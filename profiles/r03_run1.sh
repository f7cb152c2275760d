#!/bin/bash
# round 3, device run 1: the one-read backward - correctness on hardware, timing against the two launches, NT / SLOTS variants
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03
O=gpurun_out/r03/run1.txt
: > $O
run() { echo "\$ $*" >> $O; timeout 300 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
run python profiles/time_bwd_fused.py 20
for v in n1024s5 n1024s8 n512s12 n512s10; do
  KORNIA_AMD_LIB=$PWD/kornia_amd/lib/var/lib_$v.so run python profiles/time_bwd_fused.py 20
done
LAB_B=24 run python profiles/time_bwd_fused.py 10
LAB_B=7 run python profiles/time_bwd_fused.py 10
run python -m pytest tests/test_gpu_warp.py tests/test_gpu_config_parity.py tests/test_gpu_golden.py -m gpu -x -q
run python bench.py --steps 20 --warmup 5
tail -5 $O
