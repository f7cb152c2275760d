"""Register-tiled 5 x 5 blur at BASELINE config 2 (256x3x512^2 fp32), forward and adjoint, strip heights 16 and 32, through the C ABI with HIP events;
prints a checksum of each result (a variant library must print the same ones: bit-identical).  A/B of variant libraries: KORNIA_AMD_LIB.
  python profiles/time_blur_ab.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kornia_amd import _native as N
from kornia_amd.filters.gaussian import _cached_taps
lib = N.lib(); dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
stream = N.stream_ptr(dev)
tag = os.path.basename(os.environ.get('KORNIA_AMD_LIB', 'default'))
B, C, H, W, K = 256, 3, 512, 512, 5
gg = torch.Generator(device=dev).manual_seed(0)
xs = [torch.rand(B, C, H, W, device=dev, generator=gg) for _ in range(3)]
ys = [torch.empty(B, C, H, W, device=dev) for _ in range(3)]
kx, ky = _cached_taps(K, K, (1.5, 1.5), torch.float32, dev)
kx, ky = kx.float().contiguous(), ky.float().contiguous()
k = [0]
def mk(bwd):
    fn = lib.km_filter2d_sep_bwd_input if bwd else lib.km_filter2d_sep_fwd
    def f():
        k[0] += 1
        i = k[0] % 3
        N.check(fn(xs[i].data_ptr(), kx.data_ptr(), ky.data_ptr(), ys[i].data_ptr(), B, C, H, W, 1, K, K, 1, 1, 0, stream), "blur")
    return f
def checksum(bwd):
    fn = lib.km_filter2d_sep_bwd_input if bwd else lib.km_filter2d_sep_fwd
    N.check(fn(xs[0].data_ptr(), kx.data_ptr(), ky.data_ptr(), ys[0].data_ptr(), B, C, H, W, 1, K, K, 1, 1, 0, stream), "blur")
    v = ys[0].view(torch.int32)
    return int((v.to(torch.int64) * 2654435761 % 4294967291).sum().item() % 4294967291)
out = []
for rows in (16, 32):
    lib.km_config_set(b"blur_rows", rows)  # (returns the previous value)
    cf, cb = checksum(False), checksum(True)
    tf = min(bench.event_time_ms(mk(False), iters, 1) for _ in range(3)); tb = min(bench.event_time_ms(mk(True), iters, 1) for _ in range(3))
    out.append(f"rows {rows}: fwd {tf*1e3:6.1f} us adj {tb*1e3:6.1f} us  sums {cf:x} {cb:x}")
print(f"lib={tag}  " + "   ".join(out), flush=True)
