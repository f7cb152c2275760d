"""Times the pyramid kernels on the device (HIP events, C ABI called directly on pre-allocated buffers) next to the unfused
pipeline they replace.  Usage on the GPU box:  python profiles/time_pyramid.py > gpurun_out/time_pyramid.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kornia_amd as K  # noqa: E402
from kornia_amd import _native as N  # noqa: E402

dev = torch.device("cuda")
lib = N.lib()
stream = N.stream_ptr(dev)
T = K.geometry.transform
kern = T.pyramid._get_pyramid_gaussian_kernel().to(dev)
for B, C, S in ((64, 3, 512), (256, 3, 512), (16, 1, 1080)):
    H, W = (S, S) if S != 1080 else (1080, 1920)
    x = torch.rand(B, C, H, W, device=dev)
    y = torch.empty(B, C, H // 2, W // 2, device=dev)
    e = 4
    fused = bench.event_time_ms(lambda: N.check(lib.km_pyrdown_fwd(x.data_ptr(), y.data_ptr(), B, C, H, W, H // 2, W // 2, 1, 0, 0, stream), "pd"), 10)
    with torch.no_grad():
        two = bench.event_time_ms(lambda: torch.nn.functional.interpolate(K.filter2d(x, kern, "reflect"), size=(H // 2, W // 2), mode="bilinear", align_corners=False), 10)
    os.environ["KM_PYRDOWN_ALGO"] = "separable"
    sep = bench.event_time_ms(lambda: N.check(lib.km_pyrdown_fwd(x.data_ptr(), y.data_ptr(), B, C, H, W, H // 2, W // 2, 1, 0, 0, stream), "pd"), 10)
    del os.environ["KM_PYRDOWN_ALGO"]
    algo = (x.numel() + y.numel()) * e
    print(f"pyrdown {B}x{C}x{H}x{W} fp32: fused {fused:.4f} ms = {algo / fused / 1e6:.0f} GB/s algorithmic (1.25 e B/px) | separable variant {sep:.4f} ms | native blur + ATen resize {two:.4f} ms | x{two / fused:.2f}")
    up = torch.empty(B, C, H, W, device=dev)
    rs = bench.event_time_ms(lambda: N.check(lib.km_resize_bilinear_fwd(y.data_ptr(), up.data_ptr(), B, C, H // 2, W // 2, H, W, 0, 0, stream), "rs"), 10)
    with torch.no_grad():
        at = bench.event_time_ms(lambda: torch.nn.functional.interpolate(y, size=(H, W), mode="bilinear", align_corners=False), 10)
        pu = bench.event_time_ms(lambda: T.pyrup(y), 10)
        os.environ["KM_PYRDOWN_ALGO"] = "separable"
        pus = bench.event_time_ms(lambda: T.pyrup(y), 10)
        del os.environ["KM_PYRDOWN_ALGO"]
    print(f"resize x2 to {H}x{W}: native {rs:.4f} ms = {(x.numel() + y.numel()) * e / rs / 1e6:.0f} GB/s | ATen {at:.4f} ms ; pyrup (resize + 5x5 blur) {pu:.4f} ms, with the separable blur {pus:.4f} ms")
    del x, y, up
xb = torch.rand(64, 3, 512, 512, device=dev).bfloat16()
yb = torch.empty(64, 3, 256, 256, device=dev, dtype=torch.bfloat16)
ms = bench.event_time_ms(lambda: N.check(lib.km_pyrdown_fwd(xb.data_ptr(), yb.data_ptr(), 64, 3, 512, 512, 256, 256, 1, 0, 2, stream), "pd"), 10)
print(f"pyrdown 64x3x512x512 bf16: fused {ms:.4f} ms = {(xb.numel() + yb.numel()) * 2 / ms / 1e6:.0f} GB/s algorithmic")
