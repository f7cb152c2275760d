"""Times one level of ImageRegistrator's loss + gradient on the device: the fused launch (km_warp_masked_loss) against the
reference's composition running on the native warps.  Usage on the GPU box:  python profiles/time_registration.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kornia_amd as K  # noqa: E402

T = K.geometry.transform
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
for B, C, S in ((1, 3, 512), (64, 3, 256), (512, 3, 256)):
    x = torch.rand(B, C, S, S, device=dev, generator=g)
    y = torch.rand(B, C, S, S, device=dev, generator=g)
    H = (torch.eye(3, device=dev)[None] + 0.05 * (torch.rand(B, 3, 3, device=dev, generator=g) - 0.5)).requires_grad_(True)
    fused_reg = T.ImageRegistrator("homography")
    comp_reg = T.ImageRegistrator("homography", loss_fn=lambda a, b, reduction: F.l1_loss(a, b, reduction=reduction))

    def step(reg):
        H.grad = None
        reg.get_single_level_loss(x, y, H).backward()

    tf = bench.event_time_ms(lambda: step(fused_reg), 10)
    tc = bench.event_time_ms(lambda: step(comp_reg), 10)
    e = x.numel() * 4 * 2
    print(f"level loss + grad wrt H, {B}x{C}x{S}x{S} fp32: fused {tf:.4f} ms ({e / tf / 1e6:.0f} GB/s on 2e B/element) | composition on native warps {tc:.4f} ms | x{tc / tf:.1f}")
    del x, y, H
