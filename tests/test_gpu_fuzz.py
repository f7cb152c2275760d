"""Seeded random sweep of the bilinear warp (forward, grad wrt image, grad wrt matrix) against the CPU oracle: random
shapes, channel counts, rotations by any angle, scales 0.35x-3x, translations, mild perspective, both align_corners,
zeros / fill padding.  Exercises the owner-tile box logic (margins, bands, ragged tiles), the patch mapping of the
gather kernels and their wave-uniform fast paths on inputs nobody hand-picked."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _random_homography(g, B, H, W, h, w):
    """Pixel src->dst homographies: T(c_dst) R S T(-c_src) plus a small projective row."""
    ang = (torch.rand(B, generator=g) * 2 - 1) * math.pi
    sc = torch.exp((torch.rand(B, generator=g) * 2 - 1) * math.log(3.0)) * (min(h, w) / min(H, W))
    sc = sc.clamp(0.35 * min(h, w) / min(H, W), 3.0 * min(h, w) / min(H, W))
    tx = (torch.rand(B, 2, generator=g) - 0.5) * 0.3 * torch.tensor([w, h])
    M = torch.zeros(B, 3, 3)
    c, s = torch.cos(ang) * sc, torch.sin(ang) * sc
    cs, cd = torch.tensor([(W - 1) / 2, (H - 1) / 2]), torch.tensor([(w - 1) / 2, (h - 1) / 2])
    M[:, 0, 0], M[:, 0, 1], M[:, 1, 0], M[:, 1, 1] = c, -s, s, c
    M[:, :2, 2] = cd + tx - torch.einsum("bij,j->bi", M[:, :2, :2], cs)
    M[:, 2, 2] = 1.0
    M[:, 2, :2] = (torch.rand(B, 2, generator=g) - 0.5) * 0.4 / max(W, H)
    return M


@pytest.mark.parametrize("seed", range(20))
def test_random_warp_perspective_against_oracle(oracle, seed):
    import kornia_amd as K

    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    B, C = ri(1, 3), ri(1, 5)
    H, W, h, w = ri(8, 150), ri(8, 150), ri(8, 150), ri(8, 150)
    align = bool(ri(0, 1))
    pad = "fill" if (C == 3 and ri(0, 1)) else "zeros"
    fill = torch.rand(3, generator=g) if pad == "fill" else None
    x = torch.rand(B, C, H, W, generator=g)
    M = _random_homography(g, B, H, W, h, w)
    go = torch.rand(B, C, h, w, generator=g)

    xg, Mg = x.cuda().requires_grad_(), M.cuda().requires_grad_()
    y = K.warp_perspective(xg, Mg, (h, w), "bilinear", pad, align, None if fill is None else fill.cuda())
    ref = oracle.warp_perspective(x, M, (h, w), "bilinear", pad, align, fill)
    assert torch.equal(y.detach().cpu(), ref), f"forward max |d| {(y.detach().cpu() - ref).abs().max().item():.3e} (shape {tuple(x.shape)} -> {(h, w)})"
    y.backward(go.cuda())
    gs_o, gM32 = oracle.warp_perspective_backward(go, x, M, (h, w), "bilinear", pad, align, fill)
    # up to 4 * (h*w)/(H*W) contributions land on one source pixel under magnification
    atol = 1e-5 * max(1.0, 4.0 * h * w / (H * W))
    assert torch.allclose(xg.grad.cpu(), gs_o, atol=atol, rtol=1e-5), f"grad_src max |d| {(xg.grad.cpu() - gs_o).abs().max().item():.3e}"
    rel = ((Mg.grad.cpu() - gM32).abs().max() / gM32.abs().max().clamp_min(1e-20)).item()
    assert rel < 2e-4, f"grad_M relative error {rel:.2e}"


@pytest.mark.parametrize("seed", range(6))
def test_random_warp_affine_shared_and_per_sample(oracle, seed):
    import kornia_amd as K

    g = torch.Generator().manual_seed(2000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    B, C, H, W, h, w = ri(1, 4), ri(1, 4), ri(8, 120), ri(8, 120), ri(8, 120), ri(8, 120)
    align = bool(ri(0, 1))
    x = torch.rand(B, C, H, W, generator=g)
    A = _random_homography(g, B, H, W, h, w)[:, :2, :].contiguous()
    go = torch.rand(B, C, h, w, generator=g)
    xg, Ag = x.cuda().requires_grad_(), A.cuda().requires_grad_()
    y = K.warp_affine(xg, Ag, (h, w), "bilinear", "zeros", align)
    assert torch.equal(y.detach().cpu(), oracle.warp_affine(x, A, (h, w), "bilinear", "zeros", align))
    y.backward(go.cuda())
    gs_o, gA = oracle.warp_affine_backward(go, x, A, (h, w), "bilinear", "zeros", align)
    assert torch.allclose(xg.grad.cpu(), gs_o, atol=1e-5 * max(1.0, 4.0 * h * w / (H * W)), rtol=1e-5)
    assert ((Ag.grad.cpu() - gA).abs().max() / gA.abs().max().clamp_min(1e-20)).item() < 2e-4


@pytest.mark.parametrize("seed", range(12))
def test_random_filters_against_oracle(oracle, seed):
    """gaussian_blur2d (register-tiled or LDS kernel depending on W % 4), filter2d and spatial_gradient on random shapes."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(3000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    B, C = ri(1, 3), ri(1, 4)
    H, W = ri(10, 140), ri(3, 40) * (4 if ri(0, 1) else 3)
    ks = (3, 5, 7, 9)[ri(0, 3)]
    border = ("constant", "reflect", "replicate", "circular")[ri(0, 3)]
    sigma = (0.5 + 2.0 * float(torch.rand(1, generator=g)), 0.5 + 2.0 * float(torch.rand(1, generator=g)))
    x = torch.rand(B, C, H, W, generator=g)
    go = torch.rand(B, C, H, W, generator=g)
    xg = x.cuda().requires_grad_()
    y = K.gaussian_blur2d(xg, (ks, ks), sigma, border)
    assert torch.equal(y.detach().cpu(), oracle.gaussian_blur2d(x, (ks, ks), sigma, border)), (ks, border, tuple(x.shape))
    y.backward(go.cuda())
    assert torch.allclose(xg.grad.cpu(), oracle.gaussian_blur2d_backward(go, x, (ks, ks), sigma, border), atol=2e-6, rtol=1e-5)
    kf = min(ks, 7)
    k2 = torch.rand(1 if ri(0, 1) else B, kf, kf, generator=g) - 0.4
    assert torch.equal(K.filter2d(x.cuda(), k2.cuda(), border).cpu(), oracle.filter2d(x, k2, border))
    mode, order = ("sobel", "diff")[ri(0, 1)], ri(1, 2)
    assert torch.equal(K.spatial_gradient(x.cuda(), mode, order).cpu(), oracle.spatial_gradient(x, mode, order, True))


@pytest.mark.parametrize("seed", range(6))
def test_random_box_forward_against_the_gather_kernel(seed):
    """The box forward (one wide tile / gather rows under minification / two square halves / gather rows) against the gather kernel, bit for bit,
    on inputs nobody hand-picked: widths that are multiples of 4 from 4 to 132, heights from 1, output sizes around the tile sizes, fp32 / bf16 / f16
    storage, near-identity, rotated-and-scaled, projective and wild matrices (NaN compares equal to NaN), shared matrices, the three entry points."""
    import random

    import kornia_amd as K
    from kornia_amd import _native as N

    lib = N.lib()
    rnd = random.Random(100 + seed)
    g = torch.Generator().manual_seed(100 + seed)
    for _ in range(10):
        B, C = rnd.choice([1, 2, 3]), rnd.choice([1, 3])
        H, W = rnd.choice([1, 2, 3, 5, 8, 17, 33, 64, 65, 100]), 4 * rnd.choice([1, 2, 3, 5, 8, 16, 17, 25, 33])
        h, w = rnd.choice([1, 2, 7, 16, 31, 32, 33, 64, 70, 97]), rnd.choice([1, 3, 4, 31, 32, 33, 63, 64, 65, 96, 130])
        dtype = rnd.choice([torch.float32, torch.float32, torch.bfloat16, torch.float16])
        x = torch.rand(B, C, H, W, generator=g).to(dtype)
        kind = rnd.choice(["near", "rot", "proj", "wild", "shared"])
        M = torch.eye(3).repeat(B, 1, 1)
        if kind in ("near", "shared"):
            M[:, :2, :] += 0.05 * torch.randn(B, 2, 3, generator=g)
            M[:, :2, 2] += 3 * torch.randn(B, 2, generator=g)
        elif kind == "rot":
            th, s = torch.rand(B, generator=g) * 6.28, 0.3 + 2.5 * torch.rand(B, generator=g)
            M[:, 0, 0], M[:, 0, 1], M[:, 1, 0], M[:, 1, 1] = s * th.cos(), s * th.sin(), -s * th.sin(), s * th.cos()
            M[:, :2, 2] = torch.randn(B, 2, generator=g) * max(H, W) / 3
        elif kind == "proj":
            M[:, :2, :] += 0.1 * torch.randn(B, 2, 3, generator=g)
            M[:, 2, :2] = 0.01 * torch.randn(B, 2, generator=g)
        else:
            M = torch.randn(B, 3, 3, generator=g) * torch.tensor([1.0, 1.0, 30.0])
            M[:, 2] *= 0.02
            M[:, 2, 2] = 1 + 0.5 * torch.randn(B, generator=g)
        align = rnd.choice([True, False])
        fn = rnd.choice(["persp", "affine", "homog"])
        if kind == "shared":
            fn, M = "affine", M[:1]  # (only warp_affine takes a shared matrix, like the reference)
        outs = []
        for algo in (3, 4):
            prev = lib.km_config_set(b"warp_fwd_algo", algo)
            try:
                if fn == "persp":
                    o = K.warp_perspective(x.cuda(), M.cuda(), (h, w), "bilinear", "zeros", align)
                elif fn == "affine":
                    o = K.warp_affine(x.cuda(), M[:, :2].contiguous().cuda(), (h, w), "bilinear", "zeros", align)
                else:
                    o = K.homography_warp(x.cuda(), M.cuda(), (h, w), "bilinear", "zeros", align)
            finally:
                lib.km_config_set(b"warp_fwd_algo", prev)
            outs.append(torch.nan_to_num(o.float(), nan=12345.0, posinf=23456.0, neginf=-23456.0))
        assert torch.equal(outs[0], outs[1]), (kind, fn, (B, C, H, W), (h, w), dtype, align, (outs[0] - outs[1]).abs().max().item())


def _sweep_case(seed, big):
    """One case of the extended sweep, from its seed alone (a failure message names the seed: `_sweep_case(seed, True)` rebuilds it)."""
    import random

    from _util import flagship_homographies

    rnd = random.Random(seed)
    g = torch.Generator().manual_seed(seed)

    def size():
        if rnd.random() < 0.6:
            return max(8, 64 * rnd.randint(1, 8 if big else 3) + rnd.choice([-5, -4, -1, 0, 0, 1, 3, 4, 32]))
        return rnd.randint(30, 560 if big else 200)

    B, C = rnd.choice([1, 2, 3] if big else [1, 2]), rnd.choice([1, 2, 3, 3, 4, 5])
    H, W = size(), size()
    h, w = (H, W) if rnd.random() < 0.6 else (size(), size())
    pad = rnd.choice(["zeros", "zeros", "fill", "border", "reflection"])
    fn = rnd.choice(["persp", "persp", "affine", "homog"])
    if pad == "fill" and (C != 3 or fn == "homog"):  # (the reference's fill value is an RGB triple; homography_warp has no fill mode)
        pad = "zeros"
    align = rnd.choice([True, False])
    kind = rnd.choice(["near", "near", "rot"])
    if fn == "homog":
        # homography_warp takes the normalised dst -> src map: a perturbation of the identity
        M = torch.eye(3).repeat(B, 1, 1) + 0.03 * torch.randn(B, 3, 3, generator=g)
        M[:, 2, 2] = 1.0
    elif kind == "near":
        M = flagship_homographies(B, H, W, h, w, g, jitter=rnd.choice([1.0, 4.0, 8.0]))
    else:
        M = _random_homography(g, B, H, W, h, w)
    if fn == "affine":
        M = M[:, :2, :].contiguous()
    x = torch.rand(B, C, H, W, generator=g)
    go = torch.rand(B, C, h, w, generator=g)
    fill = torch.rand(3, generator=g) if pad == "fill" else None
    return dict(fn=fn, kind=kind, x=x, M=M, go=go, dsize=(h, w), pad=pad, align=align, fill=fill)


def _sweep_run(oracle, c):
    """The device result and the oracle's for one case: (y, ref, gx, gx_ref, gM, gM_ref)."""
    import kornia_amd as K

    x, M, go, (h, w), pad, align, fill = c["x"], c["M"], c["go"], c["dsize"], c["pad"], c["align"], c["fill"]
    xg, Mg = x.cuda().requires_grad_(), M.cuda().requires_grad_()
    fill_d = None if fill is None else fill.cuda()
    if c["fn"] == "persp":
        y = K.warp_perspective(xg, Mg, (h, w), "bilinear", pad, align, fill_d)
        ref = oracle.warp_perspective(x, M, (h, w), "bilinear", pad, align, fill)
        gxo, gMo = oracle.warp_perspective_backward(go, x, M, (h, w), "bilinear", pad, align, fill)
    elif c["fn"] == "affine":
        y = K.warp_affine(xg, Mg, (h, w), "bilinear", pad, align, fill_d)
        ref = oracle.warp_affine(x, M, (h, w), "bilinear", pad, align, fill)
        gxo, gMo = oracle.warp_affine_backward(go, x, M, (h, w), "bilinear", pad, align, fill)
    else:
        y = K.homography_warp(xg, Mg, (h, w), "bilinear", pad, align)
        ref = oracle.homography_warp(x, M, (h, w), "bilinear", pad, align)
        gxo, gMo = oracle.homography_warp_backward(go, x, M, (h, w), "bilinear", pad, align)
    y.backward(go.cuda())
    return y.detach().cpu(), ref, xg.grad.cpu(), gxo, Mg.grad.cpu(), gMo


def _sweep_forward_16(oracle, c, dt):
    import kornia_amd as K

    x, M, (h, w), pad, align = c["x"].to(dt), c["M"], c["dsize"], c["pad"], c["align"]
    if c["fn"] == "persp":
        return (K.warp_perspective(x.cuda(), M.cuda(), (h, w), "bilinear", pad, align).cpu(),
                oracle.warp_perspective(x.float(), M, (h, w), "bilinear", pad, align).to(dt))
    return (K.warp_affine(x.cuda(), M.cuda(), (h, w), "bilinear", pad, align).cpu(),
            oracle.warp_affine(x.float(), M, (h, w), "bilinear", pad, align).to(dt))


def test_extended_sweep_of_the_warps_at_tile_scale(oracle):
    """A time-bounded sweep at sizes where images span several 64 x 64 owner tiles / 64 x 32 forward regions (the seeded tests above stay
    below 150 pixels): sizes drawn around multiples of the tile sizes, near-identity maps (regular tiles: the persistent loop, LDS-DMA
    staging of ragged / unaligned tiles, runs of tiles) and rotated / scaled ones (general launch, multi-pass boxes), the three entry
    points, zeros / fill / border / reflection padding.  KM_FUZZ_SECONDS bounds it (default 4, and then at most three tiles a side - also
    what the host build of the kernels executes); KM_FUZZ_SEED picks the stream.  Every failure names the seed that rebuilds its case."""
    import os
    import time

    big = "KM_FUZZ_SECONDS" in os.environ
    budget = float(os.environ.get("KM_FUZZ_SECONDS", "4"))
    seed0 = int(os.environ.get("KM_FUZZ_SEED", "0"))
    t_end = time.time() + budget
    failures, n = [], 0
    while time.time() < t_end or n < 3:
        seed = seed0 * 100000 + n
        n += 1
        c = _sweep_case(seed, big)
        (h, w), (H, W) = c["dsize"], c["x"].shape[-2:]
        case = f"seed={seed} fn={c['fn']} kind={c['kind']} x={tuple(c['x'].shape)} -> {(h, w)} pad={c['pad']} align={c['align']}"
        y, ref, gx, gxo, gM, gMo = _sweep_run(oracle, c)
        if c["fn"] == "homog":
            # (transform_points' sum order is the BLAS kernel's in the reference, DESIGN.md 2: positions may differ by an ulp of the pixel coordinate)
            ok = torch.allclose(y, ref, atol=2e-7 * max(H, W, h, w) * 2, rtol=0)
        else:
            ok = torch.equal(y, ref)
        if not ok:
            failures.append(f"forward {case}: max |d| {(y - ref).abs().max().item():.3e}")
            continue
        scale = max(1.0, 4.0 * h * w / (H * W))
        if not torch.allclose(gx, gxo, atol=2e-5 * scale, rtol=1e-5):
            failures.append(f"grad_src {case}: max |d| {(gx - gxo).abs().max().item():.3e}")
        if seed % 3 == 0 and c["fn"] != "homog" and c["pad"] != "fill":
            # 16-bit storage: exactly the fp32 result of the rounded image, rounded to the storage type
            dt = torch.bfloat16 if seed % 2 else torch.float16
            y16, ref16 = _sweep_forward_16(oracle, c, dt)
            if not torch.equal(y16, ref16):
                failures.append(f"forward {dt} {case}: max |d| {(y16.float() - ref16.float()).abs().max().item():.3e}")
        rel = ((gM - gMo).abs().max() / gMo.abs().max().clamp_min(1e-20)).item()
        # (the oracle's fp32 matrix gradient is itself ~1e-4 accurate at these sizes: SURVEY.md App. C)
        if not rel < 2e-3:
            failures.append(f"grad_M {case}: rel {rel:.2e}")
    print(f"extended sweep: {n} cases in {budget:.0f} s budget, {len(failures)} failures")
    assert not failures, "\n".join(failures[:20])


def _filter_sweep_case(seed, big):
    import random

    rnd = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    hi = 640 if big else 150
    B, C = rnd.choice([1, 2, 3]), rnd.choice([1, 2, 3, 4])
    # heights around multiples of the strip heights (8 / 16 / 32 rows x 4 waves), widths around multiples of the 256-column blocks; widths that are
    # not multiples of 4 take the LDS kernels
    H = max(3, rnd.choice([rnd.randint(3, hi), 32 * rnd.randint(1, hi // 32) + rnd.choice([-3, -1, 0, 1, 2, 5])]))
    W = max(8, rnd.choice([4 * rnd.randint(2, hi // 4), 256 * rnd.randint(1, max(1, hi // 256)) + rnd.choice([-4, 0, 4, 8]), rnd.randint(8, hi)]))
    ks = rnd.choice([3, 5, 5, 7, 9])
    if rnd.random() < 0.15:
        ks = rnd.choice([11, 15, 23])  # the large separable kernel
    border = rnd.choice(["constant", "reflect", "reflect", "replicate", "circular"])
    if border == "reflect":
        H, W = max(H, ks // 2 + 1), max(W, ks // 2 + 1)  # F.pad's own limit: the reflection stays inside the image
    H, W = max(H, ks), max(W, ks)
    sigma = (0.4 + 2.5 * rnd.random(), 0.4 + 2.5 * rnd.random())
    x = torch.rand(B, C, H, W, generator=g)
    go = torch.rand(B, C, H, W, generator=g)
    return dict(x=x, go=go, ks=ks, border=border, sigma=sigma, mode=rnd.choice(["sobel", "diff"]), order=rnd.choice([1, 2]))


def test_extended_sweep_of_the_filters(oracle):
    """The same time-bounded sweep for the filter side: gaussian_blur2d forward and adjoint (register-tiled kernel at its three strip heights' edge cases, the
    LDS kernels for widths that are not multiples of 4, the large separable kernel), spatial_gradient, at sizes of several blocks a side with
    KM_FUZZ_SECONDS set (default: 4 seconds, up to 150 pixels - what the host build executes).  Bit-identical forward, adjoint within 2e-6."""
    import os
    import time

    import kornia_amd as K

    big = "KM_FUZZ_SECONDS" in os.environ
    budget = float(os.environ.get("KM_FUZZ_SECONDS", "4"))
    seed0 = int(os.environ.get("KM_FUZZ_SEED", "0"))
    t_end = time.time() + budget
    failures, n = [], 0
    while time.time() < t_end or n < 3:
        seed = 50000000 + seed0 * 100000 + n
        n += 1
        c = _filter_sweep_case(seed, big)
        x, go, ks, border, sigma = c["x"], c["go"], c["ks"], c["border"], c["sigma"]
        case = f"seed={seed} x={tuple(x.shape)} k={ks} border={border}"
        xg = x.cuda().requires_grad_()
        y = K.gaussian_blur2d(xg, (ks, ks), sigma, border)
        ref = oracle.gaussian_blur2d(x, (ks, ks), sigma, border)
        if not torch.equal(y.detach().cpu(), ref):
            failures.append(f"blur forward {case}: max |d| {(y.detach().cpu() - ref).abs().max().item():.3e}")
            continue
        y.backward(go.cuda())
        gref = oracle.gaussian_blur2d_backward(go, x, (ks, ks), sigma, border)
        if not torch.allclose(xg.grad.cpu(), gref, atol=2e-6, rtol=1e-5):
            failures.append(f"blur adjoint {case}: max |d| {(xg.grad.cpu() - gref).abs().max().item():.3e}")
        sg = K.spatial_gradient(x.cuda(), c["mode"], c["order"]).cpu()
        if not torch.equal(sg, oracle.spatial_gradient(x, c["mode"], c["order"], True)):
            failures.append(f"spatial_gradient {case} mode={c['mode']} order={c['order']}")
    print(f"extended filter sweep: {n} cases in {budget:.0f} s budget, {len(failures)} failures")
    assert not failures, "\n".join(failures[:20])
