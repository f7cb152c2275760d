"""One-read backward of warp_perspective (km_warp2d_bwd_ws with a workspace) against the two launches (without), config 2 shapes,
through the C ABI with HIP events; also checks that the two agree.
  python profiles/time_bwd_fused.py [iters]      LAB_B: batch (default 256), KORNIA_AMD_LIB: variant library"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kornia_amd import _native as N
lib = N.lib(); dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, C, S = int(os.environ.get("LAB_B", 256)), 3, 512
g = torch.Generator().manual_seed(0); gg = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, C, S, S, device=dev, generator=gg); M = bench.flagship_homographies(B, S, S, g).to(dev)
go = torch.rand(B, C, S, S, device=dev, generator=gg)
stream = N.stream_ptr(dev)
lib.km_set_traversal(1)
m = torch.empty(B, 9, device=dev); N.check(lib.km_homography_chain_fwd(M.data_ptr(), 3, None, m.data_ptr(), B, S, S, S, S, 0, stream), "c")
nbytes = int(lib.km_warp2d_bwd_workspace_bytes(B, C, S, S, S, S, 1, 0, 0))
ws = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)
res = {}
def mk(use_ws):
    gsrc = torch.empty_like(x); gm = torch.zeros(B, 9, device=dev, dtype=torch.float64)
    def f():
        gm.zero_()
        N.check(lib.km_warp2d_bwd_ws(go.data_ptr(), x.data_ptr(), m.data_ptr(), gsrc.data_ptr(), gm.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0,
                                     ws.data_ptr() if use_ws else None, nbytes if use_ws else 0, stream), "bwd")
    return f, gsrc, gm
f1, gs1, gm1 = mk(True); f0, gs0, gm0 = mk(False)
t1 = bench.event_time_ms(f1, iters); t0 = bench.event_time_ms(f0, iters)
torch.cuda.synchronize()
dgs = (gs1 - gs0).abs().max().item(); dgm = ((gm1 - gm0).abs().amax(dim=1) / gm0.abs().amax(dim=1)).max().item()
alg = 3 * 4 * B * C * S * S
print(f"lib={os.environ.get('KORNIA_AMD_LIB','default')} B={B} ws={nbytes}  fused {t1:.4f} ms ({alg/t1/1e6:.0f} GB/s, {alg/t1/8e6/10:.1f}% of 8 TB/s)   two launches {t0:.4f} ms   "
      f"|d gsrc| {dgs:.2e}  rel d gmat {dgm:.2e}", flush=True)
