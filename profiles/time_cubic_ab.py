"""Bicubic warp_affine of BASELINE config 4 (64x1x1080x1920 fp32, 2 degrees + (3, -2) px) and an RGB case (64x3x512x512, 10 degrees), public API,
min of 5 x 30 calls (HIP events), + a checksum of each result.  KORNIA_AMD_LIB: variant library."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import kornia_amd as K
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
def rot(B, H, W, deg, tx=0.0, ty=0.0):
    a = math.radians(deg); cx, cy = (W - 1) / 2.0, (H - 1) / 2.0; ca, sa = math.cos(a), math.sin(a)
    return torch.tensor([[[ca, sa, (1 - ca) * cx - sa * cy + tx], [-sa, ca, sa * cx + (1 - ca) * cy + ty]]]).repeat(B, 1, 1).to(dev)
res = []
with torch.no_grad():
    for name, (B, C, H, W, deg) in (("cfg4 64x1x1080x1920 2deg", (64, 1, 1080, 1920, 2.0)), ("64x3x512x512 10deg", (64, 3, 512, 512, 10.0)), ("64x1x1080x1920 20deg", (64, 1, 1080, 1920, 20.0))):
        x = torch.rand(B, C, H, W, generator=g).to(dev)
        A = rot(B, H, W, deg, 3.0, -2.0)
        y = K.warp_affine(x, A, (H, W), mode="bicubic")
        cs = int((y.view(torch.int32).to(torch.int64) * 2654435761 % 4294967291).sum().item() % 4294967291)
        t = min(bench.event_time_ms(lambda: K.warp_affine(x, A, (H, W), mode="bicubic"), 30) for _ in range(5))
        res.append(f"{name} {t*1e3:6.1f} us {cs:x}")
        del x, y
print(f"lib={os.path.basename(os.environ.get('KORNIA_AMD_LIB', 'default'))}  " + "   ".join(res), flush=True)
