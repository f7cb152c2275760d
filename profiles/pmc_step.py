"""The four hot launches of BASELINE config 2 (256x3x512^2 fp32: box forward, blur, blur adjoint, one-read backward) and transform_points at
2.1 GB (the control: 0.78 of peak), a few times each, for rocprofv3 --pmc passes (profiles/pmc_units.sh).  No timing here.
  python profiles/pmc_step.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import kornia_amd as K
T = K.geometry.transform
dev = torch.device("cuda")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator().manual_seed(0); gg = torch.Generator(device=dev).manual_seed(0)
B, S = 256, 512
x = torch.rand(B, 3, S, S, device=dev, generator=gg).requires_grad_()
M = bench.flagship_homographies(B, S, S, g).to(dev).requires_grad_()
go = torch.rand(B, 3, S, S, device=dev, generator=gg)
for _ in range(iters):
    y = K.gaussian_blur2d(T.warp_perspective(x, M, (S, S)), (5, 5), (1.5, 1.5))
    y.backward(go)
    x.grad = None; M.grad = None
del y
torch.cuda.synchronize()
pts = torch.rand(2048, 65536, 2, device=dev, generator=gg)  # 1.07 GB in, 1.07 GB out (bench.py other_configs)
Tm = (torch.eye(3)[None] + 0.01 * torch.randn(2048, 3, 3, generator=g)).to(dev)
with torch.no_grad():
    for _ in range(iters):
        out = K.transform_points(Tm, pts)
torch.cuda.synchronize()
print("pmc_step done", flush=True)
