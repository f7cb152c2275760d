#!/bin/bash
# round 3, device run 9: the box forward with the path chosen from the wide box and the tilt of an output row (wide tile / gather rows / square halves)
# on the flagship, under rotations and minification, against the gather kernel (KM_WARP_FWD_ALGO=rows); config 3;
# the device suite; the step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03
O=gpurun_out/r03/run9.txt
: > $O
V=$PWD/kornia_amd/lib/var
run() { echo "\$ $*" >> $O; timeout 600 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
for cfg in "LAB_X=0" "LAB_ROT=5" "LAB_ROT=20" "LAB_ROT=45" "LAB_ROT=0 LAB_SCALE=0.8" "LAB_ROT=0 LAB_SCALE=0.5" "LAB_ROT=30 LAB_SCALE=1.3"; do
  env $cfg python profiles/time_warp_kernels.py 20 fwd 2>&1 | grep -v amdgpu | sed "s/^/[$cfg] hybrid  /" >> $O

  env $cfg KM_WARP_FWD_ALGO=rows python profiles/time_warp_kernels.py 20 fwd 2>&1 | grep -v amdgpu | sed "s/^/[$cfg] rows    /" >> $O
done
run python profiles/time_config3.py
KM_WARP_FWD_ALGO=rows run python profiles/time_config3.py
run python -m pytest tests -m gpu -x -q
run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
KM_WARP_FWD_ALGO=rows run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
grep -v "^{" $O | grep -v "^\s*$\|amdgpu.ids" | tail -120
grep "^{" $O | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['ms_per_step'], d['value'], {k:v['ms'] for k,v in d['ops'].items()})"
