"""Register-tiled K x K blur, forward and adjoint, over batch sizes / image sizes / kernel sizes (fp32 and bf16) through the C ABI with HIP events:
where does a strip height pay?  A/B of libraries built with -DKMB_ROWS_OVERRIDE=<rows>: KORNIA_AMD_LIB.   python profiles/time_blur_sweep.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kornia_amd import _native as N
from kornia_amd.filters.gaussian import _cached_taps
lib = N.lib(); dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
stream = N.stream_ptr(dev)
tag = os.path.basename(os.environ.get('KORNIA_AMD_LIB', 'default'))
cases = [(256, 3, 512, 512, 5, torch.float32), (256, 3, 512, 512, 3, torch.float32), (256, 3, 512, 512, 7, torch.float32), (256, 3, 512, 512, 9, torch.float32),
         (64, 3, 512, 512, 5, torch.float32), (16, 3, 512, 512, 5, torch.float32), (4, 3, 512, 512, 5, torch.float32), (256, 3, 224, 224, 5, torch.float32),
         (64, 1, 1080, 1920, 5, torch.float32), (8, 3, 2048, 2048, 5, torch.float32), (256, 3, 512, 512, 5, torch.bfloat16), (256, 3, 224, 224, 5, torch.bfloat16),
         (1024, 3, 224, 224, 5, torch.bfloat16)]
for (B, C, H, W, K, dt) in cases:
    code = 0 if dt == torch.float32 else 2
    gg = torch.Generator(device=dev).manual_seed(0)
    xs = [torch.rand(B, C, H, W, device=dev, generator=gg).to(dt) for _ in range(3)]
    ys = [torch.empty(B, C, H, W, device=dev, dtype=dt) for _ in range(3)]
    kx, ky = _cached_taps(K, K, (1.5, 1.5), dt, dev)
    kx, ky = kx.float().contiguous(), ky.float().contiguous()
    k = [0]
    def mk(bwd):
        fn = lib.km_filter2d_sep_bwd_input if bwd else lib.km_filter2d_sep_fwd
        def f():
            k[0] += 1
            i = k[0] % 3
            N.check(fn(xs[i].data_ptr(), kx.data_ptr(), ky.data_ptr(), ys[i].data_ptr(), B, C, H, W, 1, K, K, 1, 1, code, stream), "blur")
        return f
    tf = bench.event_time_ms(mk(False), iters, 3); tb = bench.event_time_ms(mk(True), iters, 3)
    print(f"lib={tag} {B}x{C}x{H}x{W} k={K} {str(dt)[6:]:9s} forward {tf*1e3:8.1f} us   adjoint {tb*1e3:8.1f} us", flush=True)
    del xs, ys
