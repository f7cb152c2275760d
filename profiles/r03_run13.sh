#!/bin/bash
# round 3, device run 13: check of the tree after the launcher's shrink rule (outputs smaller than their source -> the gather kernel): device suite,
# forward on a 512^2 -> 224^2 warp (box forced / default / rows), the step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03
O=gpurun_out/r03/run13.txt
: > $O
run() { echo "\$ $*" >> $O; timeout 600 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
run python -m pytest tests -m gpu -x -q
cat > /tmp/shrink.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, kornia_amd as K
from kornia_amd import _native as N
lib = N.lib()
x = torch.rand(256, 3, 512, 512, device="cuda")
M = torch.eye(3, device="cuda").repeat(256, 1, 1); M[:, 0, 0] = 224 / 512; M[:, 1, 1] = 224 / 512
for name, algo in (("default", 0), ("box forced", 3), ("rows", 4)):
    prev = lib.km_config_set(b"warp_fwd_algo", algo)
    t = bench.event_time_ms(lambda: K.warp_perspective(x, M, (224, 224)), 20)
    lib.km_config_set(b"warp_fwd_algo", prev)
    print(f"warp_perspective 256x3x512^2 -> 224^2 (scale 0.4375), {name}: {t:.4f} ms", flush=True)
PY
run python /tmp/shrink.py
run python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras
grep -v "^{" $O | grep -v "amdgpu.ids\|^\.\.\.\|UserWarning\|run_backward\|Docs:\|warnings summary\|test_gpu_graph" | tail -20
grep "^{" $O | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['ms_per_step'], d['value'], {k:v['ms'] for k,v in d['ops'].items()})"
