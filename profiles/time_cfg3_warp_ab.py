"""BASELINE config 3's warp (256x3x224^2, +-15 degrees affine, bf16 and fp32) and config 5's forward (128x3x256^2 homography_warp) through the public
API with preallocation-free loops, min of 5 x 50 calls (HIP events): A/B of forward-kernel variants.  KORNIA_AMD_LIB: variant library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import kornia_amd as K
import kornia_amd.augmentation as A
dev = torch.device("cuda")
B = 256
g = torch.Generator().manual_seed(0)
Pa = {"translations": (torch.rand(B, 2, generator=g) - 0.5) * 44.8, "center": torch.full((B, 2), 111.5), "scale": (0.8 + 0.4 * torch.rand(B, 1, generator=g)).expand(B, 2).contiguous(),
      "angle": (torch.rand(B, generator=g) - 0.5) * 30, "shear_x": (torch.rand(B, generator=g) - 0.5) * 10, "shear_y": torch.zeros(B)}
Pa = {k: v.to(dev) for k, v in Pa.items()}
x32 = torch.rand(B, 3, 224, 224, generator=g).to(dev)
res = []
with torch.no_grad():
    for name, x in (("bf16", x32.bfloat16()), ("fp32", x32)):
        t = min(bench.event_time_ms(lambda: A.random_affine(x, Pa), 50) for _ in range(5))
        y = A.random_affine(x, Pa)
        cs = int((y.float().view(torch.int32).to(torch.int64) * 2654435761 % 4294967291).sum().item() % 4294967291)
        res.append(f"cfg3 random_affine {name} {t*1e3:6.1f} us {cs:x}")
    x5 = torch.rand(128, 3, 256, 256, generator=g).to(dev)
    H5 = (torch.eye(3)[None] + 0.01 * torch.randn(128, 3, 3, generator=g)).to(dev)
    t = min(bench.event_time_ms(lambda: K.homography_warp(x5, H5, (256, 256)), 50) for _ in range(5))
    res.append(f"cfg5 homography_warp fwd {t*1e3:6.1f} us")
print(f"lib={os.path.basename(os.environ.get('KORNIA_AMD_LIB', 'default'))}  " + "   ".join(res), flush=True)
