"""Forward warp of BASELINE config 2 (256x3x512^2 fp32, flagship homographies) through the C ABI, HIP events, min of 3 x 30 launches into
preallocated outputs, + the same under 5 / 20 degree rotations and 0.8 x minification (the box kernel's other paths), + a checksum of each
result (variant libraries must agree).  KORNIA_AMD_LIB: variant library.   python profiles/time_fwd_ab.py [iters]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kornia_amd import _native as N
lib = N.lib(); dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, C, S = 256, 3, 512
g = torch.Generator().manual_seed(0); gg = torch.Generator(device=dev).manual_seed(0)
xs = [torch.rand(B, C, S, S, device=dev, generator=gg) for _ in range(2)]
outs = [torch.empty(B, C, S, S, device=dev) for _ in range(2)]
stream = N.stream_ptr(dev)
lib.km_set_traversal(1)
def rot(deg, sc=1.0):
    th = math.radians(deg); ca, sa, c0 = sc * math.cos(th), sc * math.sin(th), (S - 1) / 2
    return torch.tensor([[ca, sa, (1 - ca) * c0 - sa * c0], [-sa, ca, sa * c0 + (1 - ca) * c0], [0.0, 0.0, 1.0]]).repeat(B, 1, 1)
cases = {"flagship": bench.flagship_homographies(B, S, S, g), "rot5": rot(5), "rot20": rot(20), "x0.8": rot(0, 0.8)}
res = []
for name, M in cases.items():
    M = M.to(dev)
    m = torch.empty(B, 9, device=dev); N.check(lib.km_homography_chain_fwd(M.data_ptr(), 3, None, m.data_ptr(), B, S, S, S, S, 0, stream), "c")
    k = [0]
    def f():
        k[0] += 1
        i = k[0] & 1
        N.check(lib.km_warp2d_fwd(xs[i].data_ptr(), m.data_ptr(), outs[i].data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, stream), "wf")
    k[0] = 1; f()
    v = outs[0].view(torch.int32)
    cs = int((v.to(torch.int64) * 2654435761 % 4294967291).sum().item() % 4294967291)
    t = min(bench.event_time_ms(f, iters) for _ in range(3))
    res.append(f"{name} {t*1e3:6.1f} us {cs:x}")
print(f"lib={os.path.basename(os.environ.get('KORNIA_AMD_LIB', 'default'))}  " + "   ".join(res), flush=True)
