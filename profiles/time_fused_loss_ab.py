"""Config 5's learned-homography step as the fused launch (masked_warp_loss, threshold=None: loss and H.grad), 128x3x256x256, of one library
(KORNIA_AMD_LIB): the whole step (eager, HIP events, min of 5 samples of 100 calls) and the kernel alone through the C ABI."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kornia_amd import _native as N
from kornia_amd.geometry.transform.image_registrator import masked_warp_loss
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(128, 3, 256, 256, device=dev, generator=g); tgt = torch.rand(128, 3, 256, 256, device=dev, generator=g)
H = (torch.eye(3, device=dev)[None] + 0.01 * torch.randn(128, 3, 3, device=dev, generator=g)).requires_grad_()
fn = lambda: torch.autograd.grad(masked_warp_loss(x, tgt, H, "l1", threshold=None), H)
step = min(bench.event_time_ms(fn, 100) for _ in range(5))
gh = fn()[0]
acc = torch.zeros(128, 11, device=dev, dtype=torch.float64)
m = H.detach().contiguous().view(-1, 9)
lib = N.lib()
kern = lambda: lib.km_warp_masked_loss(x.data_ptr(), tgt.data_ptr(), m.data_ptr(), acc.data_ptr(), 128, 3, 256, 256, 256, 256, 128, 2, 1, 0, 0, -1.0, 0, N.stream_ptr(dev))
k = min(bench.event_time_ms(kern, 100) for _ in range(5))
print(f"lib={os.path.basename(os.environ.get('KORNIA_AMD_LIB', 'default'))}  step {step * 1e3:.1f} us  kernel {k * 1e3:.1f} us  checksum gH {gh.double().abs().sum().item():.9e}", flush=True)
