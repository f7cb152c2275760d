#!/usr/bin/env python
"""Headline benchmark: Mpix/s of one fwd+bwd step of

    y = gaussian_blur2d(warp_perspective(x, M, (512, 512)), (5, 5), (1.5, 1.5));  y.backward(grad_out)

with x (B,3,512,512) fp32 requiring grad and M (B,3,3) requiring grad - BASELINE.json configs[1]
(B = 256 per GPU; inputs follow benchmarks/geometry/flagship.py:89-107 of the reference:
uniform-noise images, image quad perturbed by 8*randn).  Batches shard across GPUs with no
data-path collective (weak scaling: every rank owns B images).

    python bench.py                      # 1 GPU, prints ONE JSON line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The JSON line carries `roofline` (dominant kernel, HIP-event timed on the launch stream, algorithmic
bytes from SURVEY.md 8(d)) and `cpu_baseline` (the reference's op sequence on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=256, help="images per GPU")
    p.add_argument("--size", type=int, default=512)
    p.add_argument("--channels", type=int, default=3)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-batch", type=int, default=16, help="images in the CPU baseline sample")
    p.add_argument("--gather", action="store_true", help="also time an RCCL all_gather of the outputs (reported separately)")
    return p.parse_args()


def flagship_homographies(B, H, W, gen):
    """benchmarks/geometry/flagship.py:89-107: M = get_perspective_transform(quad, quad + 8*randn)."""
    src = torch.tensor([[0.0, 0.0], [W - 1.0, 0.0], [W - 1.0, H - 1.0], [0.0, H - 1.0]]).expand(B, 4, 2).double()
    dst = src + 8.0 * torch.randn(B, 4, 2, generator=gen).double()
    A = torch.zeros(B, 8, 8, dtype=torch.float64)
    b = torch.zeros(B, 8, dtype=torch.float64)
    for k in range(4):
        x, y, u, v = src[:, k, 0], src[:, k, 1], dst[:, k, 0], dst[:, k, 1]
        A[:, 2 * k, 0], A[:, 2 * k, 1], A[:, 2 * k, 2] = x, y, 1.0
        A[:, 2 * k, 6], A[:, 2 * k, 7] = -u * x, -u * y
        A[:, 2 * k + 1, 3], A[:, 2 * k + 1, 4], A[:, 2 * k + 1, 5] = x, y, 1.0
        A[:, 2 * k + 1, 6], A[:, 2 * k + 1, 7] = -v * x, -v * y
        b[:, 2 * k], b[:, 2 * k + 1] = u, v
    h = torch.linalg.solve(A, b)
    return torch.cat([h, torch.ones(B, 1, dtype=torch.float64)], dim=1).view(B, 3, 3).float()


def event_time_ms(fn, iters):
    """Average duration of `fn` (launches on torch's current stream) measured with HIP events on that stream."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_roofline(x, M, go, size, iters):
    """Times each of the four hot kernels by calling the C ABI directly (pre-allocated buffers, no
    allocator / autograd in the timed loop) and returns per-kernel stats."""
    from kornia_amd import _native as N
    from kornia_amd.filters.gaussian import _cached_taps

    lib = N.lib()
    dev = x.device
    B, C, H, W = x.shape
    h = w = size
    stream = N.stream_ptr(dev)
    m = torch.empty(B, 9, device=dev)
    N.check(lib.km_homography_chain_fwd(M.data_ptr(), 3, None, m.data_ptr(), B, H, W, h, w, 0, stream), "chain")
    warped = torch.empty(B, C, h, w, device=dev)
    blurred = torch.empty_like(warped)
    gw = torch.empty_like(warped)
    gsrc = torch.zeros_like(x)
    gm = torch.zeros(B, 9, device=dev, dtype=torch.float64)
    kx, ky = _cached_taps(5, 5, (1.5, 1.5), torch.float32, dev)
    n_el = B * C * h * w
    e = 4

    def warp_fwd():
        N.check(lib.km_warp2d_fwd(x.data_ptr(), m.data_ptr(), warped.data_ptr(), B, C, H, W, h, w, B, 0, 1, 1, 0, 1, None, 0, stream), "wf")

    def blur_fwd():
        N.check(lib.km_filter2d_sep_fwd(warped.data_ptr(), kx.data_ptr(), ky.data_ptr(), blurred.data_ptr(), B, C, h, w, 1, 5, 5, 1, 1, 0, stream), "bf")

    def blur_bwd():
        N.check(lib.km_filter2d_sep_bwd_input(go.data_ptr(), kx.data_ptr(), ky.data_ptr(), gw.data_ptr(), B, C, h, w, 1, 5, 5, 1, 1, 0, stream), "bb")

    def warp_bwd_gsrc():  # tile-owner scatter: grad wrt the image
        N.check(lib.km_warp2d_bwd(gw.data_ptr(), x.data_ptr(), m.data_ptr(), gsrc.data_ptr(), None, B, C, H, W, h, w, B, 0, 1, 1, 0, 1, None, 0, stream), "wb")

    def warp_bwd_gmat():  # forward-shaped reduction: grad wrt the homography
        N.check(lib.km_warp2d_bwd(gw.data_ptr(), x.data_ptr(), m.data_ptr(), None, gm.data_ptr(), B, C, H, W, h, w, B, 0, 1, 1, 0, 1, None, 0, stream), "wm")

    def warp_bwd():  # the public op = both kernels back to back
        N.check(lib.km_warp2d_bwd(gw.data_ptr(), x.data_ptr(), m.data_ptr(), gsrc.data_ptr(), gm.data_ptr(), B, C, H, W, h, w, B, 0, 1, 1, 0, 1, None, 0, stream), "wb")

    # algorithmic bytes per launch = compulsory traffic of what the kernel computes (DESIGN.md "Roofline accounting"):
    # every tensor it must read once + every tensor it must write once, e bytes per element
    stats = {}
    for name, fn, nbytes in (
        ("km_warp_fwd_bz_kernel", warp_fwd, 2 * e * n_el),        # read src, write out
        ("km_blur_reg_kernel<fwd>", blur_fwd, 2 * e * n_el),      # read x, write y
        ("km_blur_reg_kernel<bwd>", blur_bwd, 2 * e * n_el),      # read grad_y, write grad_x
        ("km_warp_bwd_tiled_kernel", warp_bwd_gsrc, 2 * e * n_el),  # read grad_out, write grad_src
        ("km_warp_gm_kernel", warp_bwd_gmat, 2 * e * n_el),       # read grad_out, read src
    ):
        ms = event_time_ms(fn, iters)
        stats[name] = {"ms": round(ms, 4), "alg_bytes": nbytes, "GBps": round(nbytes / ms / 1e6, 1)}
    ms = event_time_ms(warp_bwd, iters)
    # the public op km_warp2d_bwd (both launches) against SURVEY 8(d)'s 3e: read grad_out, read src, write grad_src
    stats["op:km_warp2d_bwd"] = {"ms": round(ms, 4), "alg_bytes": 3 * e * n_el, "GBps": round(3 * e * n_el / ms / 1e6, 1)}
    return stats


def other_configs(dev):
    """Informational: the parity configurations of BASELINE.json (configs[2..4]) timed through the public Python API on one
    GPU (HIP events, 10 iterations).  Not part of `value`; failures here never affect the bench line."""
    import kornia_amd as K

    out = {}

    def t(fn):
        return round(event_time_ms(fn, 10), 4)

    try:
        with torch.no_grad():
            B = 256
            x = torch.rand(B, 3, 224, 224, device=dev).bfloat16()
            ang = (torch.rand(B, device=dev) - 0.5) * 30
            A = K.get_affine_matrix2d(torch.zeros(B, 2, device=dev), torch.full((B, 2), 111.5, device=dev), 0.8 + 0.4 * torch.rand(B, 2, device=dev), ang)
            f = [0.8 + 0.4 * torch.rand(B, device=dev) for _ in range(3)] + [(torch.rand(B, device=dev) - 0.5) * 0.2]
            sig = 0.1 + 1.9 * torch.rand(B, 2, device=dev)
            out["cfg3_bf16_256x3x224_affine+colorjitter+blur_ms"] = t(
                lambda: K.gaussian_blur2d(K.enhance.color_jitter(K.warp_affine(x, A[:, :2], (224, 224), align_corners=False), *f), (5, 5), sig))
            x = torch.rand(64, 1, 1080, 1920, device=dev)
            R = K.get_rotation_matrix2d(torch.tensor([[959.5, 539.5]], device=dev).repeat(64, 1), torch.full((64,), 2.0, device=dev), torch.ones(64, 2, device=dev))
            out["cfg4_64x1x1080x1920_spatial_gradient_ms"] = t(lambda: K.spatial_gradient(x))
            out["cfg4_64x1x1080x1920_warp_affine_bicubic_ms"] = t(lambda: K.warp_affine(x, R, (1080, 1920), mode="bicubic"))
        x = torch.rand(128, 3, 256, 256, device=dev)
        H = (torch.eye(3, device=dev)[None] + 0.01 * torch.randn(128, 3, 3, device=dev)).requires_grad_()
        go = torch.rand(128, 3, 256, 256, device=dev)

        def learn_h():
            (g,) = torch.autograd.grad(K.homography_warp(x, H, (256, 256)), H, go)
            return g

        out["cfg5_128x3x256x256_homography_warp_fwd+gradH_ms"] = t(learn_h)
    except Exception as e:  # informational only
        out["error"] = f"{type(e).__name__}: {e}"
    try:  # the pyramid / registration stack that consumes config 5
        T = K.geometry.transform
        with torch.no_grad():
            x = torch.rand(256, 3, 512, 512, device=dev)
            out["pyrdown_256x3x512x512_fused_ms"] = t(lambda: T.pyrdown(x))
            os.environ["KM_PYRDOWN_ALGO"] = "separable"  # the opt-in 5 + 5 tap evaluation (csrc/km_pyramid.hip), timed for the A/B decision
            try:
                out["pyrdown_256x3x512x512_separable_variant_ms"] = t(lambda: T.pyrdown(x))
            finally:
                del os.environ["KM_PYRDOWN_ALGO"]
            out["build_pyramid_5_levels_256x3x512x512_ms"] = t(lambda: T.build_pyramid(x, 5))
            del x
        xs = torch.rand(128, 3, 256, 256, device=dev)
        xd = torch.rand(128, 3, 256, 256, device=dev)
        H = (torch.eye(3, device=dev)[None] + 0.01 * torch.randn(128, 3, 3, device=dev)).requires_grad_()

        def level_loss():
            (g,) = torch.autograd.grad(T.masked_warp_loss(xs, xd, H), H)
            return g

        out["cfg5_128x3x256x256_masked_l1_loss+gradH_one_launch_ms"] = t(level_loss)

        def cfg5_step():
            (g,) = torch.autograd.grad(T.masked_warp_loss(xs, xd, H, threshold=None), H)
            return g

        out["cfg5_128x3x256x256_l1_loss(homography_warp)+gradH_one_launch_ms"] = t(cfg5_step)
    except Exception as e:  # informational only
        out["error_next_rows"] = f"{type(e).__name__}: {e}"
    try:  # the remaining callers of the path, small so that the default run stays short
        T = K.geometry.transform
        with torch.no_grad():
            x = torch.rand(64, 3, 256, 256, device=dev)
            out["pyrup_64x3x256x256_to_512_ms"] = t(lambda: T.pyrup(x))
            x = torch.rand(8, 1, 512, 512, device=dev)
            sp = T.ScalePyramid().to(dev)
            out["scale_pyramid_8x1x512x512_ms"] = t(lambda: sp(x))
            x = torch.rand(16, 3, 512, 512, device=dev)
            out["canny_16x3x512x512_ms"] = t(lambda: K.filters.canny(x))
    except Exception as e:  # informational only
        out["error_callers"] = f"{type(e).__name__}: {e}"
    return out


def cpu_baseline(size, channels, cpu_batch):
    """The reference's CPU path (its torch op sequence, oracle/torch_ref.py) on the host cores, on a
    bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch_ref  # test infrastructure: baseline leg only

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(cpu_batch, channels, size, size, generator=g)
    M = flagship_homographies(cpu_batch, size, size, g)
    go = torch.rand(cpu_batch, channels, size, size, generator=g)
    torch_ref.headline_step(x, M, go, (size, size))  # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        torch_ref.headline_step(x, M, go, (size, size))
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or reps >= 20:
            break
    mpix = cpu_batch * size * size * reps / dt / 1e6
    # second CPU figure: the plain-C oracle (OpenMP over the same host cores), fwd + bwd of the same sample
    import oracle as c_oracle

    t1 = time.perf_counter()
    w = c_oracle.warp_perspective(x, M, (size, size))
    c_oracle.gaussian_blur2d(w, (5, 5), (1.5, 1.5))
    gw = c_oracle.gaussian_blur2d_backward(go, w, (5, 5), (1.5, 1.5))
    c_oracle.warp_perspective_backward(gw, x, M, (size, size))
    c_dt = time.perf_counter() - t1
    return {
        "value": round(mpix, 3),
        "unit": "Mpix/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{reps} fwd+bwd steps of B={cpu_batch}x{channels}x{size}x{size} fp32 through the reference's PyTorch-CPU op sequence (oracle/torch_ref.py), {dt:.1f} s",
        "c_oracle_value": round(cpu_batch * size * size / c_dt / 1e6, 3),
        "c_oracle_note": f"plain-C oracle (OpenMP, {os.cpu_count()} threads), one fwd+bwd of the same sample in {c_dt:.2f} s",
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} must be launched with torch.distributed.run --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # test hooks (1-GPU box): BENCH_FORCE_DEVICE=0 maps every rank to one device, BENCH_DIST_BACKEND=gloo replaces RCCL
    force_dev = os.environ.get("BENCH_FORCE_DEVICE")
    dev_index = int(force_dev) if force_dev is not None else local_rank
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    import kornia_amd as K

    B, C, S = args.batch, args.channels, args.size
    gen = torch.Generator().manual_seed(1000 * rank)
    ggen = torch.Generator(device=dev).manual_seed(1000 * rank)
    x = torch.rand(B, C, S, S, device=dev, generator=ggen).requires_grad_()
    M = flagship_homographies(B, S, S, gen).to(dev).requires_grad_()
    go = torch.rand(B, C, S, S, device=dev, generator=ggen)

    def step():
        x.grad = None
        M.grad = None
        y = K.gaussian_blur2d(K.warp_perspective(x, M, (S, S)), (5, 5), (1.5, 1.5))
        y.backward(go)
        return y

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # a freshly provisioned box starts with cold clocks / lazily paged-in libraries: settle for ~1 s (untimed, bounded)
    # before the W warm-up steps the contract asks for
    t_settle = time.perf_counter()
    for _ in range(200):
        step()
        torch.cuda.synchronize()
        if time.perf_counter() - t_settle > 1.0:
            break
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    gather_ms = None
    if args.gather and dist is not None and backend == "nccl":
        outs = torch.empty(world * B, C, S, S, device=dev)
        dist.all_gather_into_tensor(outs, y.detach())
        barrier()
        t1 = time.perf_counter()
        dist.all_gather_into_tensor(outs, y.detach())
        barrier()
        gather_ms = (time.perf_counter() - t1) * 1e3
        del outs

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * S * S * args.steps / elapsed / 1e6

    if rank == 0:
        with torch.no_grad():
            kstats = kernel_roofline(x.detach(), M.detach(), go, S, max(5, min(args.steps, 20)))
        dom = max((k for k in kstats if not k.startswith("op:")), key=lambda k: kstats[k]["ms"])
        achieved = kstats[dom]["GBps"]
        # HBM traffic of the dominant kernel: rocprofv3 PMC cannot run inside this process, so the figure is the
        # committed measurement of this very command/config (profiles/r01_pmc_traffic.json: FETCH_SIZE x2 + WRITE_SIZE)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if (B, C, S) == (256, 3, 512) and os.path.exists(tpath):
            rocname = {"km_warp_bwd_tiled_kernel": "km_warp_bwd_tiled_kernel", "km_warp_gm_kernel": "km_warp_gm_kernel",
                       "km_warp_fwd_bz_kernel": "km_warp_fwd_bz_kernel", "km_blur_reg_kernel<fwd>": "km_blur_reg_kernel<float, 5, false>",
                       "km_blur_reg_kernel<bwd>": "km_blur_reg_kernel<float, 5, true>"}[dom]
            for kname, rec in json.load(open(tpath))["kernels"].items():
                if rocname in kname:
                    traffic = rec["hbm_bytes_per_launch"]
        roofline = {
            "bound": "hbm",
            "kernel": dom,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same command)" if traffic else None,
            "kernel_ms": kstats[dom]["ms"],
            "alg_bytes_per_launch": kstats[dom]["alg_bytes"],
        }
        alg_step_bytes = 36 * B * C * S * S
        result = {
            "metric": "Mpix/s fwd+bwd warp_perspective+GaussianBlur2d Bx3x512x512",
            "value": round(value, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"configs[1]: warp_perspective bilinear + GaussianBlur2d k=5 sigma=1.5, B={B}x{C}x{S}x{S} fp32 fwd+bwd (grad wrt image and homography), per GPU",
                "global_batch": world * B,
                "parallelism": f"batch-shard x{world}, no data-path collective",
            },
            "step_GBps_algorithmic": round(alg_step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "step_frac_of_hbm_peak": round(alg_step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": roofline,
            "kernels": kstats,
        }
        if gather_ms is not None:
            result["all_gather_ms"] = round(gather_ms, 3)
        if world == 1:
            result["other_configs"] = other_configs(x.device)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(S, C, args.cpu_batch)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)

    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
