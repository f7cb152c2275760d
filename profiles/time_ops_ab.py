"""Op times of one library (KORNIA_AMD_LIB) at the BASELINE shapes, for A/B runs of compiler flags / kernel variants that touch several kernels:
config 2's four ops, the fused warp + blur forward, config 5's forward + gradient wrt H, config 4's two ops.
  python profiles/time_ops_ab.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import kornia_amd as K
T = K.geometry.transform
dev = torch.device("cuda")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(0); gg = torch.Generator(device=dev).manual_seed(0)
B, S = 256, 512
x = torch.rand(B, 3, S, S, device=dev, generator=gg); M = bench.flagship_homographies(B, S, S, g).to(dev)
go = torch.rand(B, 3, S, S, device=dev, generator=gg)
res = {}
with torch.no_grad():
    res["warp_fwd"] = bench.event_time_ms(lambda: T.warp_perspective(x, M, (S, S)), iters)
    res["blur_fwd"] = bench.event_time_ms(lambda: K.gaussian_blur2d(x, (5, 5), (1.5, 1.5)), iters)
    res["warp_blur_fwd"] = bench.event_time_ms(lambda: T.warp_perspective_blur(x, M, (S, S), (5, 5), (1.5, 1.5)), iters)
xb = x.clone().requires_grad_()
yb = K.gaussian_blur2d(xb, (5, 5), (1.5, 1.5))
res["blur_bwd"] = bench.event_time_ms(lambda: torch.autograd.grad(yb, xb, go, retain_graph=True), iters)
xw, Mw = x.clone().requires_grad_(), M.clone().requires_grad_()
yw = T.warp_perspective(xw, Mw, (S, S))
res["warp_bwd"] = bench.event_time_ms(lambda: torch.autograd.grad(yw, (xw, Mw), go, retain_graph=True), iters)
del xb, yb, xw, yw, x, go
# config 5: homography_warp forward + gradient wrt H only
B5, S5 = 128, 256
x5 = torch.rand(B5, 3, S5, S5, device=dev, generator=gg)
H5 = (torch.eye(3)[None] + 0.01 * torch.randn(B5, 3, 3, generator=g)).to(dev).requires_grad_()
g5 = torch.rand(B5, 3, S5, S5, device=dev, generator=gg)
y5 = T.homography_warp(x5, H5, (S5, S5))
res["cfg5_gradH"] = bench.event_time_ms(lambda: torch.autograd.grad(y5, H5, g5, retain_graph=True), iters)
# config 4
x4 = torch.rand(64, 1, 1080, 1920, device=dev, generator=gg)
A4 = torch.tensor([[[0.99939, -0.0349, 3.0], [0.0349, 0.99939, -2.0]]], device=dev).repeat(64, 1, 1)
with torch.no_grad():
    res["cfg4_sobel"] = bench.event_time_ms(lambda: K.spatial_gradient(x4), iters)
    res["cfg4_bicubic"] = bench.event_time_ms(lambda: T.warp_affine(x4, A4, (1080, 1920), mode="bicubic"), iters)
print(f"lib={os.path.basename(os.environ.get('KORNIA_AMD_LIB', 'default'))}  " + "  ".join(f"{k} {v:.4f}" for k, v in res.items()), flush=True)
