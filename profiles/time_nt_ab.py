"""Config 2's four ops of one library (KORNIA_AMD_LIB), the blur pair also at the other strip heights: A/B of the cache-policy variants
(non-temporal LDS-DMA of the backward's source tile / of the forward's box, non-temporal loads of the blur rows only one strip reads).
  python profiles/time_nt_ab.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import kornia_amd as K
from kornia_amd import _native as N
T = K.geometry.transform
dev = torch.device("cuda")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(0); gg = torch.Generator(device=dev).manual_seed(0)
B, S = 256, 512
x = torch.rand(B, 3, S, S, device=dev, generator=gg); M = bench.flagship_homographies(B, S, S, g).to(dev)
go = torch.rand(B, 3, S, S, device=dev, generator=gg)
res = {}


def best(fn):  # min of three samples of `iters` calls
    return min(bench.event_time_ms(fn, iters) for _ in range(3))


with torch.no_grad():
    res["warp_fwd"] = best(lambda: T.warp_perspective(x, M, (S, S)))
    for rows in ("", "32"):
        if rows: N.lib().km_config_set(b"blur_rows", int(rows))
        res["blur_fwd" + (rows and "_r" + rows)] = best(lambda: K.gaussian_blur2d(x, (5, 5), (1.5, 1.5)))
        N.lib().km_config_set(b"blur_rows", 0)
xb = x.clone().requires_grad_()
yb = K.gaussian_blur2d(xb, (5, 5), (1.5, 1.5))
for rows in ("", "16"):
    if rows: N.lib().km_config_set(b"blur_rows", int(rows))
    res["blur_bwd" + (rows and "_r" + rows)] = best(lambda: torch.autograd.grad(yb, xb, go, retain_graph=True))
    N.lib().km_config_set(b"blur_rows", 0)
del xb, yb
xw, Mw = x.clone().requires_grad_(), M.clone().requires_grad_()
yw = T.warp_perspective(xw, Mw, (S, S))
res["warp_bwd"] = best(lambda: torch.autograd.grad(yw, (xw, Mw), go, retain_graph=True))
print(f"lib={os.path.basename(os.environ.get('KORNIA_AMD_LIB', 'default'))}  " + "  ".join(f"{k} {v:.4f}" for k, v in res.items()), flush=True)
