"""rocprofv3 target: the matrix gradient alone at config 5's shape (128x3x256^2, homography_warp) and config 2's (256x3x512^2, warp_perspective), the box
form (km_warp_gm_box_kernel, default) and the gather kernel (km_warp_gm_kernel, warp_gm_algo = 3), 40 calls each: the kernels' own durations, without
the host's share of a launch-bound call.   device_run.sh <tag> <run> rocprof:profiles/prof_gm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import kornia_amd as K
from kornia_amd import _native as N
T = K.geometry.transform
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
for B, S, homog in ((128, 256, True), (256, 512, False)):
    x = torch.rand(B, 3, S, S, device=dev); go = torch.rand(B, 3, S, S, device=dev)
    if homog:
        M = (torch.eye(3)[None] + 0.01 * torch.randn(B, 3, 3, generator=g)).to(dev).requires_grad_()
        y = T.homography_warp(x, M, (S, S))
    else:
        M = bench.flagship_homographies(B, S, S, g).to(dev).requires_grad_()
        y = T.warp_perspective(x, M, (S, S))
    for algo in (0, 3):
        N.lib().km_config_set(b"warp_gm_algo", algo)
        for _ in range(40):
            torch.autograd.grad(y, M, go, retain_graph=True)
        torch.cuda.synchronize()
    N.lib().km_config_set(b"warp_gm_algo", 0)
    del x, go, y, M
print("lib=" + os.path.basename(os.environ.get("KORNIA_AMD_LIB", "default")), flush=True)
