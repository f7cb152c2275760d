"""The fused warp + blur forward (km_warp2d_blur_fwd) against the two launches it replaces, config 2 (256x3x512^2 fp32), through the C ABI with HIP
events (three input / output sets rotated), and the whole fwd + bwd step through the public API both ways.   python profiles/time_warp_blur.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import kornia_amd as K
from kornia_amd import _native as N
from kornia_amd.filters.gaussian import _cached_taps
lib = N.lib(); dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, C, S = int(os.environ.get("LAB_B", 256)), 3, 512
g = torch.Generator().manual_seed(0); gg = torch.Generator(device=dev).manual_seed(0)
sets = []
for _ in range(3):
    x = torch.rand(B, C, S, S, device=dev, generator=gg); M = bench.flagship_homographies(B, S, S, g).to(dev)
    m = torch.empty(B, 9, device=dev)
    N.check(lib.km_homography_chain_fwd(M.data_ptr(), 3, None, m.data_ptr(), B, S, S, S, S, 0, N.stream_ptr(dev)), "c")
    sets.append((x, M, m, torch.empty_like(x), torch.empty_like(x), torch.rand(B, C, S, S, device=dev, generator=gg)))
kx, ky = _cached_taps(5, 5, (1.5, 1.5), torch.float32, dev)
stream = N.stream_ptr(dev)
k = [0]
def nxt():
    k[0] += 1
    return sets[k[0] % 3]
def fused():
    x, M, m, w, y, go = nxt()
    N.check(lib.km_warp2d_blur_fwd(x.data_ptr(), m.data_ptr(), kx.data_ptr(), ky.data_ptr(), y.data_ptr(), B, C, S, S, S, S, B, 1, 0, 1, 1, 5, 1, 0, stream), "wb")
def two():
    x, M, m, w, y, go = nxt()
    N.check(lib.km_warp2d_fwd(x.data_ptr(), m.data_ptr(), w.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, stream), "wf")
    N.check(lib.km_filter2d_sep_fwd(w.data_ptr(), kx.data_ptr(), ky.data_ptr(), y.data_ptr(), B, C, S, S, 1, 5, 5, 1, 1, 0, stream), "bf")
n_el = B * C * S * S
for name, fn in (("fused forward", fused), ("two launches", two), ("fused forward", fused), ("two launches", two)):
    t = bench.event_time_ms(fn, iters, 5)
    print(f"{name:14s} {t:.4f} ms   ({2 * 4 * n_el / t / 1e6:.0f} GB/s on its 2e; the two launches move 4e)", flush=True)
k[0] = 0; fused(); y1 = sets[1][4].clone(); k[0] = 0; two(); torch.cuda.synchronize()
print("bit-identical:", torch.equal(y1, sets[1][4]))
T = K.geometry.transform
def step(fn):
    def f():
        x, M, m, w, y, go = nxt()
        xs, Ms = x.detach().requires_grad_(), M.detach().requires_grad_()
        fn(xs, Ms).backward(go)
    return f
for name, fn in (("fused op fwd+bwd", lambda a, b_: T.warp_perspective_blur(a, b_, (S, S), (5, 5), (1.5, 1.5))),
                 ("two ops fwd+bwd", lambda a, b_: K.gaussian_blur2d(K.warp_perspective(a, b_, (S, S)), (5, 5), (1.5, 1.5)))) * 2:
    t = bench.event_time_ms(step(fn), iters, 5)
    print(f"{name:18s} {t:.4f} ms/step   {B * S * S / t / 1e3:.0f} Mpix/s", flush=True)
