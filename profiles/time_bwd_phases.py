"""Where the tile loop of the one-read backward spends its time: a variant library built with -DKMO_PROFILE sums, per wave, the shader cycles
(s_memtime) of each phase of the loop; this prints the per-tile means over waves and workers.
  KORNIA_AMD_LIB=kornia_amd/lib/var/lib_prof.so python profiles/time_bwd_phases.py      LAB_B: batch (default 256)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from kornia_amd import _native as N
lib = N.lib(); dev = torch.device('cuda')
raw = ctypes.CDLL(os.environ["KORNIA_AMD_LIB"])
B, C, S = int(os.environ.get("LAB_B", 256)), 3, 512
g = torch.Generator().manual_seed(0); gg = torch.Generator(device=dev).manual_seed(0)
x = torch.rand(B, C, S, S, device=dev, generator=gg); M = bench.flagship_homographies(B, S, S, g).to(dev)
go = torch.rand(B, C, S, S, device=dev, generator=gg)
stream = N.stream_ptr(dev)
lib.km_set_traversal(1)
m = torch.empty(B, 9, device=dev); N.check(lib.km_homography_chain_fwd(M.data_ptr(), 3, None, m.data_ptr(), B, S, S, S, S, 0, stream), "c")
nbytes = int(lib.km_warp2d_bwd_workspace_bytes(B, C, S, S, S, S, 1, 0, 0))
ws = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)
gsrc = torch.empty_like(x); gm = torch.zeros(B, 9, device=dev, dtype=torch.float64)
def f():
    gm.zero_()
    N.check(lib.km_warp2d_bwd_ws(go.data_ptr(), x.data_ptr(), m.data_ptr(), gsrc.data_ptr(), gm.data_ptr(), B, C, S, S, S, S, B, 0, 1, 1, 0, 1, None, 0,
                                 ws.data_ptr(), nbytes, stream), "bwd")
t = bench.event_time_ms(f, 10)
torch.cuda.synchronize()
NW, PH = 16, int(os.environ.get("LAB_PHASES", 16))   # (the table always has 16 wave slots per worker)
workers = int(os.environ.get("LAB_WORKERS", 256))   # persistent workgroups of the launch (256 CUs x KMO_WG_PER_CU)
waves = int(os.environ.get("LAB_WAVES", 16))         # waves per workgroup (KMO_NT / 64)
TH = int(os.environ.get("LAB_TH", 64))               # KMO_TH
out = (ctypes.c_ulonglong * (512 * NW * PH))()
raw.km_debug_fused_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = raw.km_debug_fused_profile(out, 512 * NW * PH)
a = np.frombuffer(out, dtype=np.uint64).reshape(512, NW, PH)[:workers, :waves].astype(np.float64)
tiles_per_worker = B * (S // 64) * ((S + TH - 1) // TH) / workers
names = ["wait for requests", "stage (src->LDS, max, tables)", "flush of previous tile", "barrier B1 (+ box records)", "describe next, gm commit, scale",
         "scatter (+ next requests)", "barrier B2", "loop tail (image end sums)",
         "  stage: source tile -> LDS", "  stage: maximum of |grad_out|", "  stage: coordinate tables", "  describe: next tile's fields",
         "  describe: walk, plane pointers", "  describe: matrix-gradient commit", "-", "-"]
# (phases 8-14 are parts of the stage and of the description: with them phase 1 holds nothing and phase 4 only the scale)
tot = a.sum(axis=2)
print(f"rc {rc}  kernel+boxes+general {t:.4f} ms  B={B}  tiles/worker {tiles_per_worker:.1f}  cycles per wave (mean) {tot.mean():.0f} = {tot.mean()/tiles_per_worker:.0f} per tile"
      f"  -> {tot.mean()/ (t*1e-3) / 1e9:.2f} GHz if the loop is the whole launch")
for k in range(PH):
    v = a[:, :, k] / tiles_per_worker
    print(f"  {names[k]:34s} mean {v.mean():8.0f}  min-wave {v.min():8.0f}  max-wave {v.max():8.0f}   {100*a[:,:,k].sum()/tot.sum():5.1f} %")
# the slowest / fastest wave of a workgroup in the scatter
sc = a[:, :, 5] / tiles_per_worker
print(f"  scatter: per-workgroup spread (max - min over its 16 waves), mean over workgroups: {(sc.max(axis=1)-sc.min(axis=1)).mean():.0f} cycles")
# per-WORKER totals (a persistent workgroup = a CU): the launch ends when the slowest worker does
wt = tot.max(axis=1)  # a workgroup ends with its slowest wave
order = np.argsort(wt)
print(f"  per worker (max over its waves), cycles: min {wt.min():.0f}  median {np.median(wt):.0f}  mean {wt.mean():.0f}  max {wt.max():.0f}   max / median {wt.max()/np.median(wt):.3f}")
print("  slowest workers: " + ", ".join(f"#{int(i)}: {wt[i]:.0f}" for i in order[-6:][::-1]) + "   fastest: " + ", ".join(f"#{int(i)}: {wt[i]:.0f}" for i in order[:4]))
xcd = np.array([wt[i::8].mean() for i in range(8)])
print("  mean per XCD (worker index mod 8): " + " ".join(f"{v:.0f}" for v in xcd))
# wall time per worker on the constant-rate counter (100 MHz): equal CYCLES on a slower-clocked XCD are a longer TIME
rt = (ctypes.c_ulonglong * (512 * 4))()
raw.km_debug_fused_profile_rt.argtypes = [ctypes.c_void_p, ctypes.c_int]
if raw.km_debug_fused_profile_rt(rt, 512 * 4) == 0:
    r = np.frombuffer(rt, dtype=np.uint64).reshape(512, 4)[:workers].astype(np.float64)
    dur = (r[:, 1] - r[:, 0]) / 100.0  # microseconds
    start = (r[:, 0] - r[:, 0].min()) / 100.0
    end = (r[:, 1] - r[:, 0].min()) / 100.0
    xcc = (r[:, 2].astype(np.int64) & 0xf)
    print(f"  wall time per worker (us): min {dur.min():.1f}  median {np.median(dur):.1f}  max {dur.max():.1f}   last start {start.max():.1f}  first end {end.min():.1f}  last end {end.max():.1f}")
    for xid in sorted(set(xcc.tolist())):
        sel = xcc == xid
        print(f"    XCC {xid}: {int(sel.sum()):3d} workers  duration mean {dur[sel].mean():7.1f} us  min {dur[sel].min():7.1f}  max {dur[sel].max():7.1f}   effective clock {wt[sel].mean() / dur[sel].mean() / 1e3:.3f} GHz")
