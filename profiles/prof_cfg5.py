"""rocprofv3 target: config 5's per-GPU step (128x3x256x256, l1_loss(homography_warp(x, H), target) -> H.grad), 200 eager steps.
   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -o c5 -- python profiles/prof_cfg5.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kornia_amd as K
dev = torch.device("cuda")
x = torch.rand(128, 3, 256, 256, device=dev)
H = (torch.eye(3, device=dev)[None] + 0.01 * torch.randn(128, 3, 3, device=dev)).requires_grad_()
tgt = torch.rand(128, 3, 256, 256, device=dev)
for _ in range(200):
    (gh,) = torch.autograd.grad(torch.nn.functional.l1_loss(K.homography_warp(x, H, (256, 256)), tgt), H)
torch.cuda.synchronize()
