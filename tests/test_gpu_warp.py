"""GPU parity of the warp path against the CPU oracle (called through the kornia-compatible API,
i.e. through the C ABI).  fp32 forward results are required to be BIT-IDENTICAL to the oracle, which
itself is bit-identical to the reference's CPU path (tests/golden, test_oracle_golden.py)."""
import math

import pytest
import torch

from _util import flagship_homographies, rotation_affines

pytestmark = pytest.mark.gpu

MODES = ["bilinear", "nearest", "bicubic"]
PADS = ["zeros", "border", "reflection", "fill"]


def _inputs(B=3, C=3, H=37, W=53, h=29, w=45, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, C, H, W, generator=g)
    M = flagship_homographies(B, H, W, h, w, g, jitter=4.0)
    A = rotation_affines(B, H, W, g)
    Hn = torch.eye(3)[None] + 0.05 * torch.randn(B, 3, 3, generator=g)
    go = torch.rand(B, C, h, w, generator=g)
    return x, M, A, Hn, go, (h, w)


def _fill(pad, n=3):
    return torch.tensor([0.1, 0.5, 0.9][:n]) if pad == "fill" else None


def test_chain_bit_exact(oracle):
    import kornia_amd as K

    g = torch.Generator().manual_seed(1)
    M = flagship_homographies(64, 512, 512, 384, 640, g)
    A_o, m_o = oracle.homography_chain(M, (512, 512), (384, 640))
    A = K.normalize_homography(M.cuda(), (512, 512), (384, 640)).cpu()
    assert torch.equal(A, A_o)
    from kornia_amd.geometry.conversions import _ChainFunction

    m = _ChainFunction.apply(M.cuda(), (512, 512), (384, 640), True).cpu()
    assert torch.equal(m, m_o)


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("pad", PADS)
@pytest.mark.parametrize("mode", MODES)
def test_warp_perspective_forward_bit_exact(oracle, mode, pad, align):
    import kornia_amd as K

    x, M, _, _, _, ds = _inputs()
    ref = oracle.warp_perspective(x, M, ds, mode, pad, align, _fill(pad))
    out = K.warp_perspective(x.cuda(), M.cuda(), ds, mode, pad, align, _fill(pad)).cpu()
    assert out.shape == ref.shape
    assert torch.equal(out, ref), f"max |d| = {(out - ref).abs().max().item():.3e}"


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("pad", PADS)
@pytest.mark.parametrize("mode", MODES)
def test_warp_affine_forward_bit_exact(oracle, mode, pad, align):
    import kornia_amd as K

    x, _, A, _, _, ds = _inputs()
    ref = oracle.warp_affine(x, A, ds, mode, pad, align, _fill(pad))
    out = K.warp_affine(x.cuda(), A.cuda(), ds, mode, pad, align, _fill(pad)).cpu()
    assert torch.equal(out, ref), f"max |d| = {(out - ref).abs().max().item():.3e}"


def test_warp_affine_shared_matrix(oracle):
    import kornia_amd as K

    x, _, A, _, _, ds = _inputs()
    ref = oracle.warp_affine(x, A[:1], ds)
    out = K.warp_affine(x.cuda(), A[:1].cuda(), ds).cpu()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("norm", [True, False])
@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("pad", ["zeros", "border", "reflection"])
@pytest.mark.parametrize("mode", MODES)
def test_homography_warp_forward_bit_exact(oracle, mode, pad, align, norm):
    import kornia_amd as K

    x, _, _, Hn, _, ds = _inputs()
    if not norm:  # pixel-index base grid: use a homography that keeps pixel coordinates in range
        Hn = torch.eye(3)[None] + 0.01 * (Hn - torch.eye(3)[None])
        Hn[:, :2, 2] = Hn[:, :2, 2] / 50.0 - 1.0
        Hn[:, :2, :2] = Hn[:, :2, :2] / 25.0
    ref = oracle.homography_warp(x, Hn, ds, mode, pad, align, norm)
    out = K.homography_warp(x.cuda(), Hn.cuda(), ds, mode, pad, align, norm).cpu()
    assert torch.equal(out, ref), f"max |d| = {(out - ref).abs().max().item():.3e}"


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-12)).item()


def _check_grads(gs, gM, gs_o, gM_o, gM64, tag):
    # grad_src: sums of <= a handful of products of O(1) values -> 1e-5 absolute
    assert torch.allclose(gs, gs_o, atol=1e-5, rtol=1e-5), f"{tag}: grad_src max |d| {(gs - gs_o).abs().max().item():.3e}"
    # Matrix gradient.  The fp32 oracle uses the SAME fp32 sampling positions as the kernel (bit-identical
    # coordinates => identical floor() decisions) and accumulates in fp64: tight tolerance.
    assert _rel(gM, gM_o) < 5e-5, f"{tag}: grad_M vs fp32 oracle rel err {_rel(gM, gM_o):.3e}"
    # Against the fp64 oracle only a loose bound is meaningful: d(out)/d(coord) of a bilinear sampler jumps
    # at integer coordinates, and fp32 coordinates land on the other side of such a jump for ~1e-4 of the
    # pixels of a noise image (the reference's own fp32 gradient is ~1e-1 off).
    assert _rel(gM, gM64) < 5e-2, f"{tag}: grad_M vs fp64 oracle rel err {_rel(gM, gM64):.3e}"


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("pad", PADS)
@pytest.mark.parametrize("mode", ["bilinear", "bicubic"])
def test_warp_perspective_backward(oracle, mode, pad, align):
    import kornia_amd as K

    x, M, _, _, go, ds = _inputs()
    xg, Mg = x.cuda().requires_grad_(), M.cuda().requires_grad_()
    K.warp_perspective(xg, Mg, ds, mode, pad, align, _fill(pad)).backward(go.cuda())
    gs_o, gM_o = oracle.warp_perspective_backward(go, x, M, ds, mode, pad, align, _fill(pad))
    f64 = None if pad != "fill" else _fill(pad).double()
    _, gM64 = oracle.warp_perspective_backward(go.double(), x.double(), M.double(), ds, mode, pad, align, f64)
    _check_grads(xg.grad.cpu(), Mg.grad.cpu(), gs_o, gM_o, gM64, f"persp {mode} {pad} {align}")


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("pad", ["zeros", "reflection", "fill"])
@pytest.mark.parametrize("mode", ["bilinear", "bicubic"])
def test_warp_affine_backward(oracle, mode, pad, align):
    import kornia_amd as K

    x, _, A, _, go, ds = _inputs()
    xg, Ag = x.cuda().requires_grad_(), A.cuda().requires_grad_()
    K.warp_affine(xg, Ag, ds, mode, pad, align, _fill(pad)).backward(go.cuda())
    gs_o, gA_o = oracle.warp_affine_backward(go, x, A, ds, mode, pad, align, _fill(pad))
    f64 = None if pad != "fill" else _fill(pad).double()
    _, gA64 = oracle.warp_affine_backward(go.double(), x.double(), A.double(), ds, mode, pad, align, f64)
    _check_grads(xg.grad.cpu(), Ag.grad.cpu(), gs_o, gA_o, gA64, f"affine {mode} {pad} {align}")


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("mode", ["bilinear", "bicubic"])
def test_homography_warp_backward(oracle, mode, align):
    import kornia_amd as K

    x, _, _, Hn, go, ds = _inputs()
    xg, Hg = x.cuda().requires_grad_(), Hn.cuda().requires_grad_()
    K.homography_warp(xg, Hg, ds, mode, "zeros", align).backward(go.cuda())
    gs_o, gH_o = oracle.homography_warp_backward(go, x, Hn, ds, mode, "zeros", align)
    _, gH64 = oracle.homography_warp_backward(go.double(), x.double(), Hn.double(), ds, mode, "zeros", align)
    _check_grads(xg.grad.cpu(), Hg.grad.cpu(), gs_o, gH_o, gH64, f"homog {mode} {align}")


def test_nearest_backward(oracle):
    import kornia_amd as K

    x, M, _, _, go, ds = _inputs()
    xg = x.cuda().requires_grad_()
    K.warp_perspective(xg, M.cuda(), ds, "nearest").backward(go.cuda())
    gs_o, _ = oracle.warp_perspective_backward(go, x, M, ds, "nearest")
    assert torch.allclose(xg.grad.cpu(), gs_o, atol=1e-5)


def test_fp64_matches_oracle_and_gradcheck(oracle):
    import kornia_amd as K

    x, M, A, Hn, go, ds = _inputs(B=2, C=2, H=9, W=11, h=7, w=8)
    out = K.warp_perspective(x.double().cuda(), M.double().cuda(), ds).cpu()
    ref = oracle.warp_perspective(x.double(), M.double(), ds)
    assert torch.allclose(out, ref, atol=1e-13)
    xd = x.double().cuda().requires_grad_()
    Md = M.double().cuda().requires_grad_()
    assert torch.autograd.gradcheck(lambda a, b: K.warp_perspective(a, b, ds), (xd, Md), nondet_tol=1e-8, fast_mode=True)
    Ad = A[:2].double().cuda().requires_grad_()
    assert torch.autograd.gradcheck(lambda a, b: K.warp_affine(a, b, ds, "bicubic", "border"), (xd, Ad), nondet_tol=1e-8, fast_mode=True)
    Hd = Hn[:2].double().cuda().requires_grad_()
    assert torch.autograd.gradcheck(lambda a, b: K.homography_warp(a, b, ds), (xd, Hd), nondet_tol=1e-8, fast_mode=True)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float16, 2e-3)])
def test_half_precision_against_fp32_oracle(oracle, dtype, tol):
    """bf16/f16 parity is defined against the fp32 oracle on the SAME (rounded) inputs, result rounded to
    the storage dtype (the reference computes its grid in bf16, which is not a usable pin)."""
    import kornia_amd as K

    x, M, _, _, _, ds = _inputs()
    xr = x.to(dtype).float()
    ref = oracle.warp_perspective(xr, M, ds)
    out = K.warp_perspective(x.to(dtype).cuda(), M.cuda(), ds).float().cpu()
    assert (out - ref).abs().max().item() <= tol
    assert torch.equal(out, ref.to(dtype).float())  # in fact: exactly the rounded fp32 result


def test_identity_is_exact_and_errors():
    import kornia_amd as K

    x = torch.rand(2, 3, 16, 20, generator=torch.Generator().manual_seed(0)).cuda()
    eye = torch.eye(3)[None].expand(2, 3, 3).contiguous().cuda()
    # the sampling positions of an identity warp are integers up to one fp32 ulp of the coordinate (2e-6 at x = 19)
    assert torch.allclose(K.warp_perspective(x, eye, (16, 20)), x, atol=5e-6)
    with pytest.raises(TypeError):
        K.warp_perspective(x, [1, 2, 3], (4, 4))
    with pytest.raises(ValueError):
        K.warp_perspective(x[0], eye, (4, 4))
    with pytest.raises(ValueError):
        K.warp_perspective(x, eye[:, :2], (4, 4))
    with pytest.raises(ValueError):
        K.warp_affine(x, eye, (4, 4))
    with pytest.raises(K.NativeLibraryError):
        K.warp_perspective(x.cpu(), eye.cpu(), (4, 4))


# ---- tile-owner backward (csrc/km_warp_bwd_tiled.hip): multi-tile sources, odd channel counts,
# ---- magnification / minification, and homographies whose vanishing line crosses the images
def _tiled_case(oracle, x, M, ds, kind="perspective", align=True, pad="zeros", seed=0):
    import kornia_amd as K

    g = torch.Generator().manual_seed(seed)
    go = torch.rand(x.shape[0], x.shape[1], *ds, generator=g)
    xg, Mg = x.cuda().requires_grad_(), M.cuda().requires_grad_()
    fill = _fill(pad, x.shape[1]) if pad == "fill" else None
    if kind == "perspective":
        if pad == "fill" and x.shape[1] != 3:
            pad, fill = "zeros", None
        K.warp_perspective(xg, Mg, ds, "bilinear", pad, align, fill).backward(go.cuda())
        gs_o, g32 = oracle.warp_perspective_backward(go, x, M, ds, "bilinear", pad, align, fill)
        _, g64 = oracle.warp_perspective_backward(go.double(), x.double(), M.double(), ds, "bilinear", pad, align, None if fill is None else fill.double())
    elif kind == "affine":
        K.warp_affine(xg, Mg, ds, "bilinear", pad, align, fill).backward(go.cuda())
        gs_o, g32 = oracle.warp_affine_backward(go, x, M, ds, "bilinear", pad, align, fill)
        _, g64 = oracle.warp_affine_backward(go.double(), x.double(), M.double(), ds, "bilinear", pad, align, None if fill is None else fill.double())
    else:
        K.homography_warp(xg, Mg, ds, "bilinear", "zeros", align).backward(go.cuda())
        gs_o, g32 = oracle.homography_warp_backward(go, x, M, ds, "bilinear", "zeros", align)
        _, g64 = oracle.homography_warp_backward(go.double(), x.double(), M.double(), ds, "bilinear", "zeros", align)
    return xg.grad.cpu(), Mg.grad.cpu(), gs_o, (g32, g64)


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("shape", [(2, 3, 100, 150, 90, 140), (2, 5, 70, 130, 33, 47), (1, 1, 40, 48, 200, 230), (2, 9, 33, 65, 64, 64),
                                   (1, 2, 24, 28, 300, 310), (2, 3, 300, 316, 20, 24), (1, 4, 130, 68, 129, 67)])
def test_tiled_backward_perspective(oracle, shape, align):
    B, C, H, W, h, w = shape
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, C, H, W, generator=g)
    M = flagship_homographies(B, H, W, h, w, g, jitter=6.0)
    gs, gM, gs_o, gM64 = _tiled_case(oracle, x, M, (h, w), "perspective", align, "fill" if C == 3 else "zeros")
    # magnification sums up to (h*w)/(H*W) * 4 contributions per source pixel: scale the tolerance
    atol = 1e-5 * max(1.0, 4.0 * h * w / (H * W))
    assert torch.allclose(gs, gs_o, atol=atol, rtol=1e-5), f"grad_src max |d| {(gs - gs_o).abs().max().item():.3e}"
    assert _rel(gM, gM64[0]) < 5e-5 and _rel(gM, gM64[1]) < 5e-2


@pytest.mark.parametrize("align", [True, False])
def test_tiled_backward_affine_and_homography(oracle, align):
    g = torch.Generator().manual_seed(12)
    x = torch.rand(3, 3, 90, 160, generator=g)
    A = rotation_affines(3, 90, 160, g)
    gs, gA, gs_o, gA64 = _tiled_case(oracle, x, A, (80, 150), "affine", align, "fill")
    assert torch.allclose(gs, gs_o, atol=1e-5, rtol=1e-5)
    assert _rel(gA, gA64[0]) < 5e-5 and _rel(gA, gA64[1]) < 5e-2
    gs, gA, gs_o, gA64 = _tiled_case(oracle, x, A[:1], (80, 150), "affine", align)  # shared matrix
    assert torch.allclose(gs, gs_o, atol=1e-5, rtol=1e-5)
    assert _rel(gA, gA64[0]) < 5e-5 and _rel(gA, gA64[1]) < 5e-2
    Hn = torch.eye(3)[None] + 0.08 * torch.randn(3, 3, 3, generator=g)
    gs, gH, gs_o, gH64 = _tiled_case(oracle, x, Hn, (80, 150), "homography", align)
    assert torch.allclose(gs, gs_o, atol=1e-5, rtol=1e-5)
    assert _rel(gH, gH64[0]) < 5e-5 and _rel(gH, gH64[1]) < 5e-2


def test_tiled_backward_vanishing_line_inside_image(oracle):
    """Homographies whose denominator changes sign inside the images: the affected tiles fall back to
    scanning the whole output; grad_src must still match the oracle everywhere."""
    g = torch.Generator().manual_seed(13)
    x = torch.rand(2, 2, 96, 130, generator=g)
    M = torch.eye(3)[None].repeat(2, 1, 1)
    M[0, 2, 0] = -1.0 / 60.0   # den = 1 - x/60 : pole at x = 60
    M[1, 2, 1] = -1.0 / 40.0
    M[1, 0, 1] = 0.2
    gs, gM, gs_o, gM64 = _tiled_case(oracle, x, M, (96, 130))
    finite = torch.isfinite(gs_o) & torch.isfinite(gs)
    assert finite.float().mean() > 0.999
    assert torch.allclose(gs[finite], gs_o[finite], atol=5e-4, rtol=1e-3), (gs[finite] - gs_o[finite]).abs().max()


def test_warp_adjoint_identity_at_full_size():
    """Config-2 sized warp: <W x, y> == <x, W^T y> for the linear map x -> warp(x, M) (fixed M)."""
    import kornia_amd as K

    gen = torch.Generator().manual_seed(0)
    M = flagship_homographies(16, 512, 512, 512, 512, gen).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(16, 3, 512, 512, device="cuda", generator=g).requires_grad_()
    y = torch.rand(16, 3, 512, 512, device="cuda", generator=g)
    Wx = K.warp_perspective(x, M, (512, 512))
    Wx.backward(y)
    lhs = (Wx.detach().double() * y.double()).sum()
    rhs = (x.detach().double() * x.grad.double()).sum()
    assert abs(lhs - rhs).item() / abs(lhs).item() < 1e-6
    # linearity in the image
    x2 = torch.rand(16, 3, 512, 512, device="cuda", generator=g)
    lin = K.warp_perspective(x.detach() + 2 * x2, M, (512, 512)) - (Wx.detach() + 2 * K.warp_perspective(x2, M, (512, 512)))
    assert lin.abs().max().item() < 1e-5


@pytest.mark.parametrize("scale", [1e-30, 1e-6, 1.0, 1e12, 1e30])
def test_tiled_backward_fixed_point_dynamic_range(oracle, scale):
    """grad_src is accumulated as int32 fixed point relative to the largest |grad_out| a tile sees: the error
    must stay ~1e-7 of that maximum whatever the absolute magnitude of the incoming gradient."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(21)
    x = torch.rand(2, 3, 70, 90, generator=g)
    M = flagship_homographies(2, 70, 90, 70, 90, g, jitter=4.0)
    go = (torch.rand(2, 3, 70, 90, generator=g) - 0.3) * scale
    xg = x.cuda().requires_grad_()
    K.warp_perspective(xg, M.cuda(), (70, 90)).backward(go.cuda())
    gs_o, _ = oracle.warp_perspective_backward(go, x, M, (70, 90))
    err = (xg.grad.cpu() - gs_o).abs().max().item()
    assert err <= 2e-6 * scale, f"abs err {err:.3e} vs scale {scale:.1e}"


def test_tiled_backward_nonfinite_gradients_propagate(oracle):
    import kornia_amd as K

    g = torch.Generator().manual_seed(22)
    x = torch.rand(1, 3, 70, 90, generator=g)
    M = flagship_homographies(1, 70, 90, 70, 90, g, jitter=2.0)
    go = torch.rand(1, 3, 70, 90, generator=g)
    go[0, 1, 30, 40] = float("inf")
    go[0, 2, 10, 10] = float("nan")
    xg = x.cuda().requires_grad_()
    K.warp_perspective(xg, M.cuda(), (70, 90)).backward(go.cuda())
    gs = xg.grad.cpu()
    gs_o, _ = oracle.warp_perspective_backward(go, x, M, (70, 90))
    assert torch.equal(torch.isnan(gs), torch.isnan(gs_o)) and torch.equal(torch.isinf(gs), torch.isinf(gs_o))
    fin = torch.isfinite(gs_o)
    assert torch.allclose(gs[fin], gs_o[fin], atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [1, 3])
def test_bicubic_lds_staged_kernel_is_bit_identical(oracle, C, dtype):
    """The LDS-staged bicubic forward (km_warp_fwd_cubic_kernel: zeros padding, C in {1, 3}, W % 4 == 0, 16-byte aligned source)
    against (a) the oracle and (b) the per-pixel gather kernel, reached by handing the same image over at an address that is not 16-byte
    aligned: small and large rotations, minification (source box larger than the LDS tile -> the gather path inside the staged kernel),
    magnification, a projective map, a map that leaves the image.
    Round 6: the staged kernel spends part of the tolerance BASELINE.json grants (1e-5): its taps and coefficients are fused multiply-adds
    (csrc/km_warp_cubic.hip, KMQ_FMA) - <= 2e-6 from the oracle and from the gather kernel, which keeps the reference's roundings and stays
    bit-identical to the oracle (the name of the test is from the rounds in which both were)."""
    import kornia_amd as K

    def same(a, b, what):
        if dtype == torch.float32:
            assert (a - b).abs().max().item() <= 2e-6, (what, (a - b).abs().max().item())
        else:  # 16-bit storage: two fp32 results 2e-6 apart round to the same 16-bit value except next to a rounding boundary
            d = (a.float() - b.float()).abs()
            assert d.max().item() <= 8e-3 and (d > 0).float().mean().item() <= 2e-3, (what, d.max().item(), (d > 0).float().mean().item())

    g = torch.Generator().manual_seed(41)
    B, H, W = 7, 72, 96
    x = torch.rand(B, C, H, W, generator=g).to(dtype)
    ang = torch.tensor([2.0, 31.0, -88.0, 5.0, 0.0, 170.0, 45.0]) * math.pi / 180
    sc = torch.tensor([1.0, 1.0, 1.1, 0.35, 2.7, 1.0, 0.9])
    a, b = sc * ang.cos(), sc * ang.sin()
    cx, cy = (W - 1) / 2, (H - 1) / 2
    A = torch.stack([torch.stack([a, b, (1 - a) * cx - b * cy + torch.tensor([0.3, -4.0, 2.0, 0.0, 1.5, 0.0, 40.0])], -1),
                     torch.stack([-b, a, b * cx + (1 - a) * cy + torch.tensor([-0.7, 3.0, 0.0, 2.5, 0.0, -1.0, -25.0])], -1)], 1)
    M = torch.cat([A, torch.tensor([[0.0, 0.0, 1.0]]).expand(B, 1, 3)], 1).clone()
    M[:, 2, 0] = torch.tensor([0.0, 1e-3, -2e-3, 0.0, 5e-4, 0.0, 2e-3])
    M[:, 2, 1] = torch.tensor([0.0, -1e-3, 1e-3, 2e-3, 0.0, 0.0, -1e-3])
    xd = x.cuda()
    off = torch.empty(x.numel() + 1, dtype=dtype, device="cuda")[1:].view_as(x)  # same values, 4- / 2-byte aligned only
    off.copy_(xd)
    assert off.data_ptr() % 16 != 0
    for ds in ((H, W), (50, 64), (33, 130)):
        for align in (True, False):
            got = K.warp_affine(xd, A.cuda(), ds, "bicubic", "zeros", align)
            gen = K.warp_affine(off, A.cuda(), ds, "bicubic", "zeros", align)
            same(got, gen, (ds, align, "affine"))
            gotp = K.warp_perspective(xd, M.cuda(), ds, "bicubic", "zeros", align)
            genp = K.warp_perspective(off, M.cuda(), ds, "bicubic", "zeros", align)
            same(gotp, genp, (ds, align, "perspective"))
            if dtype == torch.float32:
                oa, op = oracle.warp_affine(x, A, ds, "bicubic", "zeros", align, None), oracle.warp_perspective(x, M, ds, "bicubic", "zeros", align, None)
                assert torch.equal(gen.cpu(), oa) and torch.equal(genp.cpu(), op), (ds, align)  # the gather kernel: the reference's roundings
                same(got.cpu(), oa, (ds, align, "oracle"))
                same(gotp.cpu(), op, (ds, align, "oracle"))
    Hn = torch.eye(3)[None].repeat(B, 1, 1) + 0.05 * torch.randn(B, 3, 3, generator=g)
    goth = K.homography_warp(xd, Hn.cuda(), (H, W), "bicubic", "zeros", True)
    same(goth, K.homography_warp(off, Hn.cuda(), (H, W), "bicubic", "zeros", True), "homography")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [1, 3])
def test_box_forward_is_bit_identical(oracle, C, dtype):
    """The box forward (km_warp_fwd_box_kernel: the source box of a 64 x 16 output tile through LDS, bilinear + zeros) against the gather
    forward (KM_WARP_FWD_ALGO=rows) and, in fp32, the oracle, bit for bit: flagship homographies, small and large rotations (boxes that do
    not fit -> the gather path inside the kernel), minification, magnification, a map that leaves the image, output sizes that are not
    multiples of the tile, both align_corners, affine / homography coordinate modes, the per-sample switch."""
    import kornia_amd as K
    from kornia_amd import _native as N

    lib = N.lib()
    g = torch.Generator().manual_seed(43)
    B, H, W = 7, 72, 96
    x = torch.rand(B, C, H, W, generator=g).to(dtype)
    ang = torch.tensor([2.0, 31.0, -88.0, 5.0, 0.0, 170.0, 45.0]) * math.pi / 180
    sc = torch.tensor([1.0, 1.0, 1.1, 0.35, 2.7, 1.0, 0.9])
    a, b = sc * ang.cos(), sc * ang.sin()
    cx, cy = (W - 1) / 2, (H - 1) / 2
    A = torch.stack([torch.stack([a, b, (1 - a) * cx - b * cy + torch.tensor([0.3, -4.0, 2.0, 0.0, 1.5, 0.0, 40.0])], -1),
                     torch.stack([-b, a, b * cx + (1 - a) * cy + torch.tensor([-0.7, 3.0, 0.0, 2.5, 0.0, -1.0, -25.0])], -1)], 1)
    M = torch.cat([A, torch.tensor([[0.0, 0.0, 1.0]]).expand(B, 1, 3)], 1).clone()
    M[:, 2, 0] = torch.tensor([0.0, 1e-3, -2e-3, 0.0, 5e-4, 0.0, 2e-3])
    M[:, 2, 1] = torch.tensor([0.0, -1e-3, 1e-3, 2e-3, 0.0, 0.0, -1e-3])
    Mf = flagship_homographies(B, H, W, 150, 200, g, jitter=4.0)
    Hn = torch.eye(3)[None].repeat(B, 1, 1) + 0.05 * torch.randn(B, 3, 3, generator=g)
    xd = x.cuda()

    def both(fn):
        outs = []
        for algo in (3, 4):
            prev = lib.km_config_set(b"warp_fwd_algo", algo)
            try:
                outs.append(fn())
            finally:
                lib.km_config_set(b"warp_fwd_algo", prev)
        assert torch.equal(outs[0], outs[1])
        return outs[0]

    for ds in ((H, W), (50, 64), (33, 130), (150, 200)):
        for align in (True, False):
            got = both(lambda: K.warp_affine(xd, A.cuda(), ds, "bilinear", "zeros", align))
            gotp = both(lambda: K.warp_perspective(xd, M.cuda(), ds, "bilinear", "zeros", align))
            gotf = both(lambda: K.warp_perspective(xd, Mf.cuda(), ds, "bilinear", "zeros", align))
            if dtype == torch.float32:
                assert torch.equal(got.cpu(), oracle.warp_affine(x, A, ds, "bilinear", "zeros", align, None)), (ds, align)
                assert torch.equal(gotp.cpu(), oracle.warp_perspective(x, M, ds, "bilinear", "zeros", align, None)), (ds, align)
                assert torch.equal(gotf.cpu(), oracle.warp_perspective(x, Mf, ds, "bilinear", "zeros", align, None)), (ds, align)
    both(lambda: K.homography_warp(xd, Hn.cuda(), (H, W), "bilinear", "zeros", True))
    both(lambda: K.homography_warp(xd, Hn.cuda(), (60, 100), "bilinear", "zeros", False))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_16_bit_storage_is_the_fp32_result_rounded_once(dtype):
    """A 16-bit image gives EXACTLY the fp32 result of the same values rounded to the storage type, whichever kernel computes it (box forward,
    gather rows, generic sampler, the blur).  On the device the compiler used to fuse the last fma of the accumulation with the conversion
    (v_fma_mixlo_f16: the exact sum rounded to f16 once) in some kernels and not in others: 1 ulp apart at ties, ~1e-5 of the pixels."""
    import kornia_amd as K
    from kornia_amd import _native as N

    lib = N.lib()
    g = torch.Generator().manual_seed(105)
    x = torch.rand(3, 3, 72, 100, generator=g).to(dtype)
    A = torch.tensor([[[0.9480, -0.0047, 2.6100], [-0.0321, 0.9642, 1.1322]]]).repeat(3, 1, 1)
    A[1, :, :2] = torch.tensor([[0.81, 0.55], [-0.55, 0.81]])   # a rotation: the square tiles
    A[2] *= 0.6                                                   # minification: gather rows inside the box kernel
    for algo in (3, 4, 1):
        prev = lib.km_config_set(b"warp_fwd_algo", algo)
        try:
            got = K.warp_affine(x.cuda(), A.cuda(), (96, 136), "bilinear", "zeros", False)
            ref = K.warp_affine(x.float().cuda(), A.cuda(), (96, 136), "bilinear", "zeros", False).to(dtype)
        finally:
            lib.km_config_set(b"warp_fwd_algo", prev)
        assert torch.equal(got, ref), (algo, (got.float() - ref.float()).abs().max().item(), int((got != ref).sum()))
    got = K.gaussian_blur2d(x.cuda(), (5, 5), (1.5, 1.5))
    # (the reference materialises the row pass in the storage type: the fp32 run of the same values is not the comparison for the blur;
    # what is pinned here is that the result does not depend on the fusion - the strip height changes the code the compiler sees)
    for rows in (8, 16, 32):
        prev = lib.km_config_set(b"blur_rows", rows)
        try:
            again = K.gaussian_blur2d(x.cuda(), (5, 5), (1.5, 1.5))
        finally:
            lib.km_config_set(b"blur_rows", prev)
        assert torch.equal(got, again), rows
