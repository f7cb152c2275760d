#!/bin/bash
# round 3, device run 2: the one-read backward after the wait-placement work - variants and ablations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03
O=gpurun_out/r03/run2.txt
: > $O
run() { echo "\$ $*" >> $O; timeout 300 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
run python profiles/time_bwd_fused.py 20
for v in th32 th32s8 src2 n512 abl1 abl2 abl3 abl4 th32abl4; do
  KORNIA_AMD_LIB=$PWD/kornia_amd/lib/var/lib_$v.so run python profiles/time_bwd_fused.py 20
done
LAB_B=24 run python profiles/time_bwd_fused.py 10
LAB_B=24 KORNIA_AMD_LIB=$PWD/kornia_amd/lib/var/lib_th32.so run python profiles/time_bwd_fused.py 10
run python -m pytest tests/test_gpu_warp.py tests/test_gpu_config_parity.py tests/test_gpu_golden.py -m gpu -x -q
KORNIA_AMD_LIB=$PWD/kornia_amd/lib/var/lib_th32.so run python -m pytest tests/test_gpu_warp.py tests/test_gpu_config_parity.py -m gpu -x -q
run python bench.py --steps 20 --warmup 5
grep -v "^{" $O | grep "fused\|passed\|failed\|rc" | tail -40
