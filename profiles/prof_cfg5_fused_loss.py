"""rocprofv3 target: config 5's learned-homography step as ONE fused launch (masked_warp_loss with threshold=None: l1(homography_warp(x, H), target)
and H.grad), 128x3x256x256, 200 eager steps - which launches the step is made of (the kernel and the torch ops around it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kornia_amd as K
from kornia_amd.geometry.transform.image_registrator import masked_warp_loss
dev = torch.device("cuda")
x = torch.rand(128, 3, 256, 256, device=dev)
H = (torch.eye(3, device=dev)[None] + 0.01 * torch.randn(128, 3, 3, device=dev)).requires_grad_()
tgt = torch.rand(128, 3, 256, 256, device=dev)
for _ in range(200):
    (gh,) = torch.autograd.grad(masked_warp_loss(x, tgt, H, "l1", threshold=None), H)
torch.cuda.synchronize()
print("lib=default")
