"""warp_affine backward under border / reflection padding (a RandomAffine option): the tile-owner kernel of the one-read backward against the generic
scatter with global atomics, 256x3x512^2 fp32, rotation 10 degrees + translation (a frame of ~40 px maps outside).  python profiles/time_bwd_border.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import kornia_amd as K
from kornia_amd import _native as N
lib = N.lib(); dev = torch.device('cuda')
B, C, S = 256, 3, 512
gg = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, C, S, S, device=dev, generator=gg); go = torch.rand(B, C, S, S, device=dev, generator=gg)
c_, s_ = math.cos(math.radians(10.0)), math.sin(math.radians(10.0)); cx = cy = (S - 1) / 2
A = torch.tensor([[c_, s_, (1 - c_) * cx - s_ * cy + 12.0], [-s_, c_, s_ * cx + (1 - c_) * cy - 7.0]], device=dev).repeat(B, 1, 1)
for pad in ("zeros", "border", "reflection"):
    for fused in (1, 0):
        prev = lib.km_config_set(b"warp_bwd_fused", fused)
        def f():
            xs, As = x.detach().requires_grad_(), A.detach().requires_grad_()
            K.warp_affine(xs, As, (S, S), padding_mode=pad).backward(go)
            return xs.grad, As.grad
        t = bench.event_time_ms(f, 10, 3)
        gx, gA = f(); torch.cuda.synchronize()
        lib.km_config_set(b"warp_bwd_fused", prev)
        print(f"padding_mode={pad:10s} one-read tile owners={'on ' if fused else 'off'}  warp_affine fwd+bwd {t:.4f} ms   |grad_x|max {gx.abs().max().item():.3f}", flush=True)
