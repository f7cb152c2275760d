#!/bin/bash
# round 3, device run 17: the box matrix-gradient kernel (km_warp_gm_box_kernel, KM_WARP_GM_ALGO=lds) against the gather kernel: the kernel alone on the
# flagship and under rotation / minification, the two-launch backward, config 5's public path; its device test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r03
O=gpurun_out/r03/run17.txt
: > $O
run() { echo "\$ $*" >> $O; timeout 300 "$@" >> $O 2>&1; echo "[rc $?]" >> $O; }
run python -m pytest tests/test_gpu_warp.py -m gpu -x -q -k "box_matrix_gradient or 16_bit"
for cfg in "LAB_X=0" "LAB_ROT=5" "LAB_ROT=20" "LAB_ROT=45" "LAB_ROT=0 LAB_SCALE=0.8"; do
  env $cfg python profiles/time_warp_kernels.py 20 gm 2>&1 | grep -v amdgpu | sed "s/^/[$cfg] rows /" >> $O
  env $cfg KM_WARP_GM_ALGO=lds python profiles/time_warp_kernels.py 20 gm 2>&1 | grep -v amdgpu | sed "s/^/[$cfg] box  /" >> $O
done
cat > /tmp/c5.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench, kornia_amd as K
x = torch.rand(128, 3, 256, 256, device="cuda"); tgt = torch.rand(128, 3, 256, 256, device="cuda")
H = (torch.eye(3, device="cuda") + 0.01 * torch.randn(128, 3, 3, device="cuda")).requires_grad_()
def step():
    (gh,) = torch.autograd.grad(torch.nn.functional.l1_loss(K.homography_warp(x, H, (256, 256)), tgt), H)
print("config 5 public path: l1(homography_warp) + grad H: %.4f ms" % bench.event_time_ms(step, 30))
PY
run python /tmp/c5.py
KM_WARP_GM_ALGO=lds run python /tmp/c5.py
grep -v "^{" $O | grep -v "amdgpu.ids\|^\.\.\." | tail -30
