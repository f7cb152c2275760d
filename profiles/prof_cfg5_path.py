"""rocprofv3 target: config 5's PATH (128x3x256x256: homography_warp forward + backward wrt H with the upstream gradient given - what
bench.py's cfg5_*_fwd+gradH figure times), 200 eager steps: the launches of one step and their durations.
   device_run.sh <tag> <run> rocprof:profiles/prof_cfg5_path.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kornia_amd as K
dev = torch.device("cuda")
x = torch.rand(128, 3, 256, 256, device=dev)
H = (torch.eye(3, device=dev)[None] + 0.01 * torch.randn(128, 3, 3, device=dev)).requires_grad_()
go = torch.rand(128, 3, 256, 256, device=dev)
for _ in range(200):
    (gh,) = torch.autograd.grad(K.homography_warp(x, H, (256, 256)), H, go)
torch.cuda.synchronize()
print("lib=" + os.path.basename(os.environ.get("KORNIA_AMD_LIB", "default")), flush=True)
