"""Per-op breakdown of BASELINE config 3 (256 images per GPU, bf16, 224x224) on one GPU: HIP-event time and host wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kornia_amd as K
import kornia_amd.augmentation as A

dev = torch.device("cuda")
B = 256
x = torch.rand(B, 3, 224, 224, device=dev).bfloat16()
g = torch.Generator().manual_seed(0)
Pa = {"translations": (torch.rand(B, 2, generator=g) - 0.5) * 44.8, "center": torch.full((B, 2), 111.5), "scale": (0.8 + 0.4 * torch.rand(B, 1, generator=g)).expand(B, 2).contiguous(),
      "angle": (torch.rand(B, generator=g) - 0.5) * 30, "shear_x": (torch.rand(B, generator=g) - 0.5) * 10, "shear_y": torch.zeros(B), "batch_prob": torch.ones(B)}
Pj = {"brightness_factor": 0.8 + 0.4 * torch.rand(B, generator=g), "contrast_factor": 0.8 + 0.4 * torch.rand(B, generator=g), "saturation_factor": 0.8 + 0.4 * torch.rand(B, generator=g),
      "hue_factor": (torch.rand(B, generator=g) - 0.5) * 0.2, "order": torch.tensor([0, 2, 3, 1]), "batch_prob": torch.ones(B)}
Pb = {"sigma": 0.1 + 1.9 * torch.rand(B, generator=g), "batch_prob": torch.ones(B)}
dPa = {k: v.to(dev) for k, v in Pa.items()}
dPj = {k: (v.to(dev) if k != "order" else v) for k, v in Pj.items()}
dPb = {k: v.to(dev) for k, v in Pb.items()}
M = A.affine_matrix(dPa, dev)
sig = dPb["sigma"].unsqueeze(-1).expand(-1, 2)

def t(name, fn, n=20):
    with torch.no_grad():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0 = time.perf_counter(); e0.record()
        for _ in range(n): fn()
        e1.record(); wq = time.perf_counter() - w0
        e1.synchronize()
    print(f"{name:60s} gpu {e0.elapsed_time(e1)/n*1e3:9.1f} us   host-enqueue {wq/n*1e6:9.1f} us", flush=True)

w = K.warp_affine(x, M[:, :2], (224, 224), align_corners=False)
c = A.color_jitter(w, dPj)
t("affine_matrix (params->M)", lambda: A.affine_matrix(dPa, dev))
t("warp_affine bf16 256x3x224^2", lambda: K.warp_affine(x, M[:, :2], (224, 224), align_corners=False))
t("random_affine (matrix+warp+mask)", lambda: A.random_affine(x, dPa))
t("color_jitter", lambda: A.color_jitter(w, dPj))
def cj_fwd_bwd():
    with torch.enable_grad():
        wr = w.detach().requires_grad_()
        A.color_jitter(wr, dPj).backward(c)
t("color_jitter fwd + bwd (image gradient)", cj_fwd_bwd)
t("gaussian_blur2d tensor sigma (B,2)", lambda: K.gaussian_blur2d(c, (5, 5), sig))
t("gaussian_blur2d tuple sigma", lambda: K.gaussian_blur2d(c, (5, 5), (1.5, 1.5)))
t("random_gaussian_blur", lambda: A.random_gaussian_blur(c, dPb))
t("apply_sequence (device params)", lambda: A.apply_sequence(x, dPa, dPj, dPb))
t("apply_sequence (host params)", lambda: A.apply_sequence(x, Pa, Pj, Pb))
no_p = lambda d: {k: v for k, v in d.items() if k != "batch_prob"}
t("apply_sequence (device params, p=1: no mask)", lambda: A.apply_sequence(x, no_p(dPa), no_p(dPj), no_p(dPb)))
x32 = x.float()
t("apply_sequence fp32", lambda: A.apply_sequence(x32, dPa, dPj, dPb))
# ---- HIP-graph replay of the whole sequence (device parameters) ----
order = [0, 2, 3, 1]
def seq(xx, Pa_, Pj_, Pb_):
    return A.random_gaussian_blur(A.color_jitter(A.random_affine(xx, Pa_), Pj_, order), Pb_)
np_ = lambda d: {k: v for k, v in d.items() if k not in ("batch_prob", "order")}
step = K.graph.capture(seq, x, np_(dPa), np_(dPj), np_(dPb), no_grad=True)
t("graph replay: apply_sequence bf16 (p=1, no mask)", lambda: step.replay())
dPj2 = {k: v for k, v in dPj.items() if k != "order"}
step_m = K.graph.capture(seq, x, dPa, dPj2, dPb, no_grad=True)
t("graph replay: apply_sequence bf16 (with apply masks)", lambda: step_m.replay())
step32 = K.graph.capture(seq, x32, np_(dPa), np_(dPj), np_(dPb), no_grad=True)
t("graph replay: apply_sequence fp32 (p=1)", lambda: step32.replay())
t("eager again: apply_sequence bf16 (p=1)", lambda: seq(x, np_(dPa), np_(dPj), np_(dPb)))
