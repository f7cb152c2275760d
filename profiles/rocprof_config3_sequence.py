"""BASELINE config 3's per-GPU share (256x3x224^2 bf16: RandomAffine -> ColorJitter -> RandomGaussianBlur, device-resident parameters) and nothing
else, N times, for `rocprofv3 --kernel-trace --stats`: every row of the table with a call count that is a multiple of N belongs to the sequence.
  python profiles/rocprof_config3_sequence.py [N]     (device_run.sh stage rocprof:profiles/rocprof_config3_sequence.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kornia_amd.augmentation as A
dev = torch.device("cuda")
N_ = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = 256
g = torch.Generator().manual_seed(0)
x = torch.rand(B, 3, 224, 224, generator=g).bfloat16().to(dev)
def params(with_prob):
    Pa = {"translations": (torch.rand(B, 2, generator=g) - 0.5) * 44.8, "center": torch.full((B, 2), 111.5), "scale": (0.8 + 0.4 * torch.rand(B, 1, generator=g)).expand(B, 2).contiguous(),
          "angle": (torch.rand(B, generator=g) - 0.5) * 30, "shear_x": (torch.rand(B, generator=g) - 0.5) * 10, "shear_y": torch.zeros(B)}
    Pj = {"brightness_factor": 0.8 + 0.4 * torch.rand(B, generator=g), "contrast_factor": 0.8 + 0.4 * torch.rand(B, generator=g),
          "saturation_factor": 0.8 + 0.4 * torch.rand(B, generator=g), "hue_factor": (torch.rand(B, generator=g) - 0.5) * 0.2}
    Pb = {"sigma": 0.1 + 1.9 * torch.rand(B, generator=g)}
    if with_prob:
        for d in (Pa, Pj, Pb):
            d["batch_prob"] = (torch.rand(B, generator=g) < 0.7).float()
    return tuple({k: v.to(dev) for k, v in d.items()} for d in (Pa, Pj, Pb))
order = [0, 2, 3, 1]
sets = {"p=1": params(False), "p<1": params(True)}
torch.cuda.synchronize()
with torch.no_grad():
    for name, (Pa, Pj, Pb) in sets.items():
        for _ in range(N_):
            y = A.random_gaussian_blur(A.color_jitter(A.random_affine(x, Pa), Pj, order), Pb)
torch.cuda.synchronize()
print(f"lib=default  config 3 sequence: {N_} x (p = 1) + {N_} x (p < 1, the per-sample switch inside the launches); rows of the sequence have {2 * N_} calls", flush=True)
