"""The matrix gradient alone (grad wrt H / M only: the box form km_warp_gm_box_kernel by default since round 6, the gather kernel km_warp_gm_kernel
under warp_gm_algo = 3, the first LDS-staged km_warp_gm_lds_kernel under warp_gm_algo = 2) of one
library (KORNIA_AMD_LIB): config 5's shape through homography_warp, config 2's through warp_perspective.  Prints the time of both kernels and
their relative difference.   python profiles/time_gm_ab.py [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import kornia_amd as K
from kornia_amd import _native as N
T = K.geometry.transform
dev = torch.device("cuda")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
g = torch.Generator().manual_seed(0); gg = torch.Generator(device=dev).manual_seed(0)
out = []
for name, B, S, homog in (("cfg5 128x3x256^2 homography_warp", 128, 256, True), ("cfg2 256x3x512^2 warp_perspective", 256, 512, False)):
    x = torch.rand(B, 3, S, S, device=dev, generator=gg)
    go = torch.rand(B, 3, S, S, device=dev, generator=gg)
    if homog:
        M = (torch.eye(3)[None] + 0.01 * torch.randn(B, 3, 3, generator=g)).to(dev).requires_grad_()
        y = T.homography_warp(x, M, (S, S))
    else:
        M = bench.flagship_homographies(B, S, S, g).to(dev).requires_grad_()
        y = T.warp_perspective(x, M, (S, S))
    res, grads = {}, {}
    for algo, tag in ((3, "gather"), (2, "lds"), (0, "box")):
        N.lib().km_config_set(b"warp_gm_algo", algo)
        fn = lambda: torch.autograd.grad(y, M, go, retain_graph=True)
        res[tag] = min(bench.event_time_ms(fn, iters) for _ in range(3))
        grads[tag] = fn()[0].double()
    N.lib().km_config_set(b"warp_gm_algo", 0)
    rel = ((grads["gather"] - grads["box"]).abs().max() / grads["gather"].abs().max()).item()
    out.append(f"{name}: gather {res['gather'] * 1e3:.1f} us  lds {res['lds'] * 1e3:.1f} us  box {res['box'] * 1e3:.1f} us  rel diff box / gather {rel:.1e}")
    del x, go, y, M
print(f"lib={os.path.basename(os.environ.get('KORNIA_AMD_LIB', 'default'))}  " + "   ".join(out), flush=True)
