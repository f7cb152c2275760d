"""The one-read backward of config 2 (km_warp2d_bwd_ws, flagship homographies) 40 times, for rocprofv3 --kernel-trace --stats: per-kernel averages of
its three launches (boxes, persistent loop, general launch with the scan).  KORNIA_AMD_LIB: variant library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kornia_amd import _native as N
lib = N.lib(); dev = torch.device('cuda')
B, C, S = 256, 3, 512
g = torch.Generator().manual_seed(0); gg = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, C, S, S, device=dev, generator=gg); M = bench.flagship_homographies(B, S, S, g).to(dev)
go = torch.rand(B, C, S, S, device=dev, generator=gg)
stream = N.stream_ptr(dev)
m = torch.empty(B, 9, device=dev); N.check(lib.km_homography_chain_fwd(M.data_ptr(), 3, None, m.data_ptr(), B, S, S, S, S, 0, stream), "c")
nbytes = int(lib.km_warp2d_bwd_workspace_bytes(B, C, S, S, S, S, 1, 0, 0))
ws = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)
gsrc = torch.empty_like(x); gm = torch.empty(B, 9, device=dev, dtype=torch.float64)
for _ in range(40):
    N.check(lib.km_warp2d_bwd_ws(go.data_ptr(), x.data_ptr(), m.data_ptr(), gsrc.data_ptr(), gm.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0, ws.data_ptr(), nbytes, stream), "bwd")
torch.cuda.synchronize()
print(f"lib={os.path.basename(os.environ.get('KORNIA_AMD_LIB', 'default'))}  checksum gm {gm.sum().item():.6e} gsrc {gsrc.double().sum().item():.9e}", flush=True)
