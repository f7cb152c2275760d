"""BASELINE config 4 on one GPU: 64x1x1080x1920 fp32 `spatial_gradient` and bicubic `warp_affine` (HIP-event time per call).
KM_WARP_FWD_ALGO=generic python profiles/time_config4.py   times the per-pixel gather kernel instead of the LDS-staged one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kornia_amd as K

dev = torch.device("cuda")

def t(name, fn, n=3 if os.environ.get('CFG4_PMC') else 20, bytes_=None):
    with torch.no_grad():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / n
    extra = f"   {bytes_ / ms / 1e9:6.2f} TB/s algorithmic" if bytes_ else ""
    print(f"{name:64s} {ms * 1e3:9.1f} us{extra}", flush=True)

algo = os.environ.get("KM_WARP_FWD_ALGO", "default")
SHAPES = ((64, 1, 1080, 1920),) if os.environ.get("CFG4_ONLY") else ((64, 1, 1080, 1920), (16, 3, 1080, 1920), (256, 3, 512, 512))
for (B, C, H, W) in SHAPES:
    x = torch.rand(B, C, H, W, device=dev)
    n = x.numel()
    ctr = torch.tensor([[(W - 1) / 2, (H - 1) / 2]], device=dev).repeat(B, 1)
    for ang in ((2.0,) if os.environ.get("CFG4_PMC") else (2.0, 20.0, 45.0)):
        R = K.get_rotation_matrix2d(ctr, torch.full((B,), ang, device=dev), torch.ones(B, 2, device=dev))
        t(f"[{algo}] warp_affine bicubic {B}x{C}x{H}x{W} rot {ang:4.1f}", lambda: K.warp_affine(x, R, (H, W), mode="bicubic"), bytes_=8 * n)
    if C == 1 and not os.environ.get("CFG4_PMC"):
        t(f"spatial_gradient {B}x{C}x{H}x{W}", lambda: K.spatial_gradient(x), bytes_=12 * n)
        xb = x.bfloat16()
        t(f"[{algo}] warp_affine bicubic bf16 rot 2", lambda: K.warp_affine(xb, R * 0 + K.get_rotation_matrix2d(ctr, torch.full((B,), 2.0, device=dev), torch.ones(B, 2, device=dev)), (H, W), mode="bicubic"), bytes_=4 * n)
    del x
