"""Times the three warp kernels of config 2 one by one through the C ABI (HIP events): fwd / gm / scatter.
  python profiles/time_warp_kernels.py [iters] [which: fwd,gm,sc,blur]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kornia_amd import _native as N
lib = N.lib(); dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = (sys.argv[2] if len(sys.argv) > 2 else "fwd,gm,sc").split(",")
B, C, S = int(os.environ.get("LAB_B", 256)), 3, 512
g = torch.Generator().manual_seed(0); gg = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, C, S, S, device=dev, generator=gg); M = bench.flagship_homographies(B, S, S, g).to(dev)
if os.environ.get('LAB_IDENTITY'):  # identity / pure translation: the kernels' structure without the perspective access pattern
    M = torch.eye(3).repeat(B, 1, 1); M[:, 0, 2] = float(os.environ['LAB_IDENTITY']); M = M.to(dev)
if os.environ.get('LAB_ROT'):  # rotation about the image centre by LAB_ROT degrees, scale LAB_SCALE (default 1)
    import math
    th = math.radians(float(os.environ['LAB_ROT'])); sc = float(os.environ.get('LAB_SCALE', 1.0))
    ca, sa, c0 = sc * math.cos(th), sc * math.sin(th), (S - 1) / 2
    M = torch.tensor([[ca, sa, (1 - ca) * c0 - sa * c0], [-sa, ca, sa * c0 + (1 - ca) * c0], [0.0, 0.0, 1.0]]).repeat(B, 1, 1).to(dev)
go = torch.rand(B, C, S, S, device=dev, generator=gg)
stream = N.stream_ptr(dev)
lib.km_set_traversal(0 if os.environ.get('LAB_ALTERNATE') else 1)  # one kernel over one input: fixed direction (bench.py: kernel_roofline)
m = torch.empty(B, 9, device=dev); N.check(lib.km_homography_chain_fwd(M.data_ptr(), 3, None, m.data_ptr(), B, S, S, S, S, 0, stream), "c")
out = torch.empty_like(x); gsrc = torch.empty_like(x); gm = torch.zeros(B, 9, device=dev, dtype=torch.float64)
fns = {
 "fwd": lambda: N.check(lib.km_warp2d_fwd(x.data_ptr(), m.data_ptr(), out.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, stream), "wf"),
 "gm": lambda: N.check(lib.km_warp2d_bwd(go.data_ptr(), x.data_ptr(), m.data_ptr(), None, gm.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, stream), "gm"),
 "bwd": lambda: N.check(lib.km_warp2d_bwd(go.data_ptr(), x.data_ptr(), m.data_ptr(), gsrc.data_ptr(), gm.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, stream), "bwd"),
 "sc": lambda: N.check(lib.km_warp2d_bwd(go.data_ptr(), x.data_ptr(), m.data_ptr(), gsrc.data_ptr(), None, B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, stream), "sc"),
}
if "par" in which:  # the two backward launches on two streams (fork / join per iteration)
    s2 = torch.cuda.Stream(device=dev); s2p = s2.cuda_stream
    gm0 = lambda: N.check(lib.km_warp2d_bwd(go.data_ptr(), x.data_ptr(), m.data_ptr(), None, gm.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, s2p), "gm")
    def par():
        s2.wait_stream(torch.cuda.current_stream(dev))
        if os.environ.get("LAB_GM_FIRST"):
            gm0(); fns["sc"]()
        else:
            fns["sc"](); gm0()
        torch.cuda.current_stream(dev).wait_stream(s2)
    fns["par"] = par
if "copy" in which:
    fns["copy"] = lambda: out.copy_(x)
for k in which:
    print(k, round(bench.event_time_ms(fns[k], iters), 4), "ms", flush=True)
