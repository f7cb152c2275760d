#!/bin/bash
# round 3, device run 7: (a) one-read backward with the tile records precomputed by the boxes kernel, the next item described and the
# image-end sums taken in front of the barriers (lib_bwdhead = the kernel before); (b) box forward: shapes (default 64 x 32, lib_w16, lib_sq),
# rotated / minified inputs (the gather-rows fallback inside the box kernel against the gather kernel); (c) config 3 with the 16-bit box
# forward (lib_box16); (d) the whole device suite on the new defaults; (e) the step, new defaults against the round's starting point
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03
O=gpurun_out/r03/run7.txt
: > $O
V=$PWD/kornia_amd/lib/var
run() { echo "\$ $*" >> $O; timeout 600 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
KORNIA_AMD_LIB=$V/lib_prof.so run python profiles/time_bwd_phases.py
run python profiles/time_bwd_fused.py 20
KORNIA_AMD_LIB=$V/lib_bwdhead.so run python profiles/time_bwd_fused.py 20
run python profiles/time_bwd_fused.py 20
KORNIA_AMD_LIB=$V/lib_bwdhead.so run python profiles/time_bwd_fused.py 20
run python profiles/time_warp_kernels.py 20 fwd
KORNIA_AMD_LIB=$V/lib_w16.so run python profiles/time_warp_kernels.py 20 fwd
KORNIA_AMD_LIB=$V/lib_sq.so run python profiles/time_warp_kernels.py 20 fwd
KM_WARP_FWD_ALGO=rows run python profiles/time_warp_kernels.py 20 fwd
for rot in 5 20 45; do
  LAB_ROT=$rot run python profiles/time_warp_kernels.py 20 fwd
  LAB_ROT=$rot KORNIA_AMD_LIB=$V/lib_sq.so run python profiles/time_warp_kernels.py 20 fwd
  LAB_ROT=$rot KM_WARP_FWD_ALGO=rows run python profiles/time_warp_kernels.py 20 fwd
done
LAB_ROT=0 LAB_SCALE=0.5 run python profiles/time_warp_kernels.py 20 fwd
LAB_ROT=0 LAB_SCALE=0.5 KM_WARP_FWD_ALGO=rows run python profiles/time_warp_kernels.py 20 fwd
run python profiles/time_config3.py
KORNIA_AMD_LIB=$V/lib_box16.so run python profiles/time_config3.py
run python -m pytest tests -m gpu -x -q
run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
KM_WARP_FWD_ALGO=rows KORNIA_AMD_LIB=$V/lib_bwdhead.so run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
KM_WARP_FWD_ALGO=rows KORNIA_AMD_LIB=$V/lib_bwdhead.so run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
grep -v "^{" $O | grep -v "^\s*$\|amdgpu.ids" | tail -150
grep "^{" $O | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['ms_per_step'], d['value'], {k:v['ms'] for k,v in d['ops'].items()})"
