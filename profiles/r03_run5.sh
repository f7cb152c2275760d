#!/bin/bash
# round 3, device run 5: late slots of the one-read backward looked at in the middle of the scatter (their latency was exposed at the top
# of the tile loop) - same-box A/B against the previous kernel (lib_head), KMO_EARLY / tile-height / source-request variants
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03
O=gpurun_out/r03/run5.txt
: > $O
run() { echo "\$ $*" >> $O; timeout 300 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
run python profiles/time_bwd_fused.py 20
for v in head e2 e4 e6 th32 srcat1 s5; do
  KORNIA_AMD_LIB=$PWD/kornia_amd/lib/var/lib_$v.so run python profiles/time_bwd_fused.py 20
done
run python profiles/time_bwd_fused.py 20
run python -m pytest tests/test_gpu_warp_fused.py tests/test_gpu_warp.py tests/test_gpu_config_parity.py -m gpu -x -q
run python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
grep -v "^{" $O | grep "fused\|passed\|failed\|rc" | tail -40
