"""Config 3's warp (256x3x224^2 bf16 / fp32 through augmentation.random_affine) by RANGE OF ANGLES (the rest of config 3's ranges unchanged, and alone):
which of the box forward's tile shapes the regions take - one wide 64 x 32 tile (box <= 80 x 40), two square halves (52 x 50 each), gather rows -
decides the time.   python profiles/time_cfg3_warp_angles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import kornia_amd.augmentation as A
dev = torch.device("cuda")
B = 256
g = torch.Generator().manual_seed(0)
x32 = torch.rand(B, 3, 224, 224, generator=g).to(dev)
with torch.no_grad():
    for deg, full in ((0.0, False), (5.0, False), (15.0, False), (30.0, False), (15.0, True)):
        Pa = {"translations": ((torch.rand(B, 2, generator=g) - 0.5) * 44.8) if full else torch.zeros(B, 2), "center": torch.full((B, 2), 111.5),
              "scale": ((0.8 + 0.4 * torch.rand(B, 1, generator=g)).expand(B, 2).contiguous()) if full else torch.ones(B, 2),
              "angle": (torch.rand(B, generator=g) - 0.5) * 2 * deg, "shear_x": ((torch.rand(B, generator=g) - 0.5) * 10) if full else torch.zeros(B), "shear_y": torch.zeros(B)}
        Pa = {k: v.to(dev) for k, v in Pa.items()}
        out = []
        for name, x in (("bf16", x32.bfloat16()), ("fp32", x32)):
            t = min(bench.event_time_ms(lambda: A.random_affine(x, Pa), 50) for _ in range(5))
            out.append(f"{name} {t * 1e3:6.1f} us")
        print(f"angles +-{deg:4.1f} deg {'+ scale 0.8-1.2, shear 5, translation 10 %' if full else 'alone':44s} " + "   ".join(out), flush=True)
