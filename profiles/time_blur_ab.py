"""Register-tiled 5x5 blur, forward and adjoint, config 2 (256x3x512^2 fp32) and config 3's share (256x3x224^2 bf16), through the C ABI with HIP events;
three input / output sets rotated.  A/B of libraries: KORNIA_AMD_LIB.   python profiles/time_blur_ab.py [iters]"""
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
for (B, C, S, dt, code) in ((256, 3, 512, torch.float32, 0), (256, 3, 224, torch.bfloat16, 2)):
    gg = torch.Generator(device=dev).manual_seed(0)
    xs = [torch.rand(B, C, S, S, device=dev, generator=gg).to(dt) for _ in range(3)]
    ys = [torch.empty(B, C, S, S, device=dev, dtype=dt) for _ in range(3)]
    kx, ky = _cached_taps(5, 5, (1.5, 1.5), dt, dev)
    kx, ky = kx.float().contiguous(), ky.float().contiguous()
    k = [0]
    def mk(bwd):
        fn = lib.km_filter2d_sep_bwd_input if bwd else lib.km_filter2d_sep_fwd
        def f():
            k[0] += 1
            i = k[0] % 3
            N.check(fn(xs[i].data_ptr(), kx.data_ptr(), ky.data_ptr(), ys[i].data_ptr(), B, C, S, S, 1, 5, 5, 1, 1, code, stream), "blur")
        return f
    for bwd in (False, True, False, True):
        t = bench.event_time_ms(mk(bwd), iters, 5)
        nb = 2 * xs[0].element_size() * B * C * S * S
        print(f"lib={os.path.basename(os.environ.get('KORNIA_AMD_LIB', 'default'))} {B}x{C}x{S}x{S} {str(dt)[6:]} {'adjoint' if bwd else 'forward'}  {t:.4f} ms  {nb / t / 1e6:.0f} GB/s", flush=True)
    del xs, ys
