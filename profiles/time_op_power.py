"""Shader clock and board power right after ONE op of the hot step has looped for ~3 s (sysfs after the final synchronize, the protocol that gives
sane values in bench.py's `clocks`): which kernels pull the clock down / sit at the power cap.   python profiles/time_op_power.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kornia_amd import _native as N
from kornia_amd.filters.gaussian import _cached_taps
lib = N.lib(); dev = torch.device('cuda')
B, C, S = 256, 3, 512
g = torch.Generator().manual_seed(0); gg = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, C, S, S, device=dev, generator=gg); M = bench.flagship_homographies(B, S, S, g).to(dev)
go = torch.rand(B, C, S, S, device=dev, generator=gg)
out = torch.empty_like(x); gsrc = torch.empty_like(x); gm = torch.zeros(B, 9, device=dev, dtype=torch.float64)
stream = N.stream_ptr(dev)
m = torch.empty(B, 9, device=dev); N.check(lib.km_homography_chain_fwd(M.data_ptr(), 3, None, m.data_ptr(), B, S, S, S, S, 0, stream), "c")
kx, ky = _cached_taps(5, 5, (1.5, 1.5), torch.float32, dev)
nbytes = int(lib.km_warp2d_bwd_workspace_bytes(B, C, S, S, S, S, 1, 0, 0))
ws = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)
ops = {
    "idle (2 s sleep)": None,
    "warp_fwd": lambda: lib.km_warp2d_fwd(x.data_ptr(), m.data_ptr(), out.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, stream),
    "blur_fwd": lambda: lib.km_filter2d_sep_fwd(x.data_ptr(), kx.data_ptr(), ky.data_ptr(), out.data_ptr(), B, C, S, S, 1, 5, 5, 1, 1, 0, stream),
    "warp_bwd_fused": lambda: lib.km_warp2d_bwd_ws(go.data_ptr(), x.data_ptr(), m.data_ptr(), gsrc.data_ptr(), gm.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, ws.data_ptr(), nbytes, stream),
    "warp_bwd_gsrc_only": lambda: lib.km_warp2d_bwd(go.data_ptr(), x.data_ptr(), m.data_ptr(), gsrc.data_ptr(), None, B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, stream),
    "warp_bwd_gmat_only": lambda: lib.km_warp2d_bwd(go.data_ptr(), x.data_ptr(), m.data_ptr(), None, gm.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, stream),
    "copy": lambda: out.copy_(x),
}
for name, fn in ops.items():
    if fn is None:
        time.sleep(2.0)
        print(f"{name:22s} {bench.device_clocks(0)}", flush=True)
        continue
    fn(); torch.cuda.synchronize()
    n = 0
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < 3.0:
        for _ in range(100):
            fn()
        n += 100
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    c = bench.device_clocks(0)
    print(f"{name:22s} {e0.elapsed_time(e1) / n:.4f} ms  {c}", flush=True)
    time.sleep(1.0)
