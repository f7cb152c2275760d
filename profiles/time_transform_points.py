"""transform_points well beyond the Infinity Cache (2048 x 65536 x 2 fp32: 1.07 GB in, 1.07 GB out), a fresh pair of buffers per call
out of a ring of 3, HIP-event timed.  The command the rocprofv3 rows of profiles/r03_transform_points_* were taken on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import kornia_amd as K
dev = torch.device("cuda")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B, Np = 2048, 65536
P = [torch.rand(B, Np, 2, device=dev) for _ in range(3)]
T = torch.eye(3, device=dev)[None].repeat(B, 1, 1) + 0.01 * torch.randn(B, 3, 3, device=dev)
k = [0]
def f():
    k[0] += 1
    return K.transform_points(T, P[k[0] % 3])
with torch.no_grad():
    ms = bench.event_time_ms(f, iters)
nbytes = 2 * P[0].numel() * 4
print(f"transform_points {B}x{Np}x2 fp32: {ms:.4f} ms, {nbytes / ms / 1e6:.0f} GB/s algorithmic (2e per coordinate), {nbytes / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s")
