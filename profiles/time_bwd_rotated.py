"""The backward of warp_perspective under rotations / scalings (boxes of the source tiles that do not fit the persistent loop's
registers go to the general launch): one-read form against the two launches, 256 x 3 x 512^2, HIP events.
  python profiles/time_bwd_rotated.py [iters]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kornia_amd import _native as N
lib = N.lib(); dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, C, S = int(os.environ.get("LAB_B", 256)), 3, 512
gg = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, C, S, S, device=dev, generator=gg); go = torch.rand(B, C, S, S, device=dev, generator=gg)
stream = N.stream_ptr(dev)
nbytes = int(lib.km_warp2d_bwd_workspace_bytes(B, C, S, S, S, S, 1, 0, 0))
ws = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)
gsrc = torch.empty_like(x); gm = torch.zeros(B, 9, device=dev, dtype=torch.float64)
c = (S - 1) / 2
for name, ang, sc in (("identity", 0.0, 1.0), ("5 deg", 5.0, 1.0), ("10 deg", 10.0, 1.0), ("20 deg", 20.0, 1.0), ("45 deg", 45.0, 1.0), ("scale 0.8", 0.0, 0.8), ("scale 1.3", 0.0, 1.3)):
    a = math.radians(ang); ca, sa = sc * math.cos(a), sc * math.sin(a)
    M = torch.tensor([[ca, sa, (1 - ca) * c - sa * c], [-sa, ca, sa * c + (1 - ca) * c], [0, 0, 1]], dtype=torch.float32).repeat(B, 1, 1).to(dev)
    m = torch.empty(B, 9, device=dev); N.check(lib.km_homography_chain_fwd(M.data_ptr(), 3, None, m.data_ptr(), B, S, S, S, S, 0, stream), "c")
    def f(use_ws):
        gm.zero_()
        N.check(lib.km_warp2d_bwd_ws(go.data_ptr(), x.data_ptr(), m.data_ptr(), gsrc.data_ptr(), gm.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0,
                                     ws.data_ptr() if use_ws else None, nbytes if use_ws else 0, stream), "bwd")
    t1 = bench.event_time_ms(lambda: f(True), iters); g1 = gsrc.clone(); m1 = gm.clone()
    t0 = bench.event_time_ms(lambda: f(False), iters)
    torch.cuda.synchronize()
    print(f"{name:10s} one read {t1:.4f} ms   two launches {t0:.4f} ms   |d gsrc| {(g1 - gsrc).abs().max().item():.2e}  rel d gmat {((m1 - gm).abs().max() / gm.abs().max()).item():.2e}", flush=True)
