"""GPU: randomized shapes for the pyramid kernels and the fused registration loss against the oracle (seeded; every case is
reproducible from its seed).  Runs on the host build of the kernels in the CPU suite (tests/test_emulated_kernels.py, also
under AddressSanitizer via tests/emu/asan.sh); sorts last because these shapes have not been on a device yet."""
import pytest
import torch

from _util import golden as _golden_np

pytestmark = pytest.mark.gpu


def golden(name):
    return {k: torch.from_numpy(v) for k, v in _golden_np(name).items()}


BORDERS = ["constant", "reflect", "replicate", "circular"]


def T():
    import kornia_amd as K

    return K.geometry.transform


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("seed", range(24))
def test_random_pyramid_shapes_against_oracle(oracle, seed):
    g = torch.Generator().manual_seed(9000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    B, C = ri(1, 3), ri(1, 4)
    H, W = ri(3, 140), ri(3, 300)
    if seed % 3 == 0:  # the register-tiled factor-2 path: even height, width a multiple of 4
        H, W = 2 * ri(2, 80), 4 * ri(2, 90)
    border = BORDERS[ri(0, 3)]
    align = bool(ri(0, 1))
    factor = [2.0, 2.0, 1.5, 3.0, 2.5][ri(0, 4)]
    dt = torch.float64 if seed % 5 == 4 else torch.float32
    x = (torch.rand(B, C, H, W, generator=g) * 3 - 1).to(dt)
    if int(float(H) / factor) > 0 and int(float(W) // factor) > 0:
        out = T().pyrdown(x.cuda(), border, align, factor).cpu()
        assert torch.equal(out, oracle.pyrdown(x, border, align, factor)), (seed, "pyrdown", x.shape, border, align, factor)
    if H * W <= 20000:
        up = T().pyrup(x.cuda(), border, align).cpu()
        assert torch.equal(up, oracle.pyrup(x, border, align)), (seed, "pyrup", x.shape, border, align)
    size = (ri(1, 150), ri(1, 150))
    rs = T().resize_bilinear(x.cuda(), size, align).cpu()
    assert torch.equal(rs, oracle.resize_bilinear(x, size, align)), (seed, "resize", x.shape, size, align)
    if seed % 4 in (0, 1):  # half-precision storage through the same shapes (8-byte / 4-byte vector paths)
        hd = torch.bfloat16 if seed % 8 < 4 else torch.float16
        xh = x.float().clamp(0, 1).to(hd)
        if int(float(H) / factor) > 0 and int(float(W) // factor) > 0:
            oh = T().pyrdown(xh.cuda(), border, align, factor).cpu()
            assert oh.dtype == hd and (oh.float() - oracle.pyrdown(xh.float(), border, align, factor)).abs().max().item() <= 2e-2
        rh = T().resize_bilinear(xh.cuda(), (2 * H, 2 * W) if W % 2 == 0 else size, False).cpu()
        assert (rh.float() - oracle.resize_bilinear(xh.float(), (2 * H, 2 * W) if W % 2 == 0 else size, False)).abs().max().item() <= 2e-2
    if seed % 2 == 0 and W % 2 == 0:  # the exact x2 path
        rs2 = T().resize_bilinear(x.cuda(), (2 * H, 2 * W), False).cpu()
        assert torch.equal(rs2, oracle.resize_bilinear(x, (2 * H, 2 * W), False)), (seed, "resize x2", x.shape)


@pytest.mark.parametrize("seed", range(12))
def test_random_masked_loss_against_oracle(oracle, seed):
    g = torch.Generator().manual_seed(7000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    B, C = ri(1, 3), [1, 3, 2, 4][ri(0, 3)]
    H, W = ri(4, 90), ri(4, 150)
    src, dst = torch.rand(B, C, H, W, generator=g), torch.rand(B, C, H, W, generator=g)
    amp = [0.02, 0.1, 0.4][ri(0, 2)]
    Hm = torch.eye(3)[None].repeat(B, 1, 1) + amp * (torch.rand(B, 3, 3, generator=g) - 0.5)
    if seed % 4 == 3:
        Hm = Hm[:1]
    kind = ["l1", "mse"][ri(0, 1)]
    align = bool(ri(0, 1))
    thr = [0.9, 0.5, None][ri(0, 2)]
    ref, gref = oracle.masked_warp_loss(src, dst, Hm, kind, align, True, -1.0 if thr is None else thr)
    Hg = Hm.cuda().requires_grad_(True)
    loss = T().masked_warp_loss(src.cuda(), dst.cuda(), Hg, kind, align, threshold=thr)
    if torch.isnan(ref):
        assert torch.isnan(loss)
        return
    loss.backward()
    assert abs(loss.item() - ref.item()) < 2e-6 * max(1.0, abs(ref.item())), (seed, loss.item(), ref.item())
    assert _rel(Hg.grad.cpu(), gref) < 2e-4, (seed, Hg.grad.cpu(), gref)


def test_pyramid_gradcheck_fp64():
    """pyrdown / pyrup / resize are linear: native forward, adjoint = ATen's resize adjoint + the native blur adjoint."""
    t = T()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 2, 9, 12, generator=g, dtype=torch.float64).cuda().requires_grad_(True)
    for border in ("reflect", "constant", "circular"):
        assert torch.autograd.gradcheck(lambda v: t.pyrdown(v, border), (x,), eps=1e-6, atol=1e-6, nondet_tol=1e-9)
        assert torch.autograd.gradcheck(lambda v: t.pyrdown(v, border, True, 1.5), (x,), eps=1e-6, atol=1e-6, nondet_tol=1e-9)
    xs = torch.rand(1, 1, 5, 6, generator=g, dtype=torch.float64).cuda().requires_grad_(True)
    assert torch.autograd.gradcheck(lambda v: t.pyrup(v, "replicate"), (xs,), eps=1e-6, atol=1e-6, nondet_tol=1e-9)
    assert torch.autograd.gradcheck(lambda v: t.resize_bilinear(v, (7, 11), True), (xs,), eps=1e-6, atol=1e-6, nondet_tol=1e-9)
    # fp32: the adjoint identity <A x, y> = <x, A^T y> of the fused forward / composed backward pair
    a = torch.rand(2, 3, 32, 48, generator=g).cuda().requires_grad_(True)
    y = torch.rand(2, 3, 16, 24, generator=g).cuda()
    out = t.pyrdown(a)
    (out * y).sum().backward()
    b = torch.rand(2, 3, 32, 48, generator=g).cuda()
    lhs = (t.pyrdown(b) * y).sum().item()
    rhs = (b * a.grad).sum().item()
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_gradient_path_matches_reference():
    """Gradient of pyrdown against the reference's autograd result (fused native forward, composed adjoint)."""
    d = golden("pyramid")
    x = d["x_even"].cuda().requires_grad_(True)
    out = T().pyrdown(x)
    assert torch.allclose(out.detach().cpu(), d["down_even_reflect_0"], atol=1e-6, rtol=0)
    (out * d["down_even_w"].cuda()).sum().backward()
    assert torch.allclose(x.grad.cpu(), d["down_even_gx"], atol=1e-5, rtol=1e-5)
    with torch.no_grad():
        assert torch.allclose(T().pyrdown(x).cpu(), out.detach().cpu(), atol=1e-6, rtol=0)


def test_scale_pyramid_backward():
    x = golden("scale_pyramid")["x"].cuda().requires_grad_(True)
    pyr, _, _ = T().ScalePyramid(n_levels=2, min_size=10).cuda()(x)
    sum(p.sum() for p in pyr).backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()


def test_reference_own_pyramid_cases():
    """The cases of the reference's tests/geometry/transform/test_pyramid.py (shapes :27-35, 61-74, 114-139; symmetry :76-81,
    151-162; blur order :141-149; conventions :41-56, 87-112; the octave-0 sigma example :174-189)."""
    t = T()
    z = lambda *s: torch.zeros(*s).cuda()  # noqa: E731
    assert t.PyrUp()(z(1, 2, 4, 4)).shape == (1, 2, 8, 8) and t.PyrUp()(z(2, 2, 4, 4)).shape == (2, 2, 8, 8)
    assert t.PyrDown()(z(1, 2, 4, 4)).shape == (1, 2, 2, 2) and t.PyrDown()(z(2, 2, 4, 4)).shape == (2, 2, 2, 2)
    assert t.PyrDown(factor=3.0)(z(1, 2, 9, 9)).shape == (1, 2, 3, 3)
    assert t.pyrdown(torch.rand(1, 1, 5, 5).cuda()).shape == (1, 1, 2, 2)  # floor, not OpenCV's ceil
    blob = z(1, 1, 6, 6)
    blob[:, :, 2:4, 2:4] = 1.0
    out = t.PyrDown()(blob).squeeze()
    assert torch.allclose(out, out.flip(0)) and torch.allclose(out, out.flip(1))
    x = torch.arange(0.0, 25.0).view(1, 1, 5, 5).cuda()
    for op in (t.pyrdown, t.pyrup):
        assert torch.allclose(op(x), op(x, align_corners=False), atol=1e-2) and torch.allclose(op(x), op(x, border_type="reflect"), atol=1e-2)
        assert not torch.allclose(op(x, align_corners=False), op(x, align_corners=True), atol=1e-2, rtol=1e-2)
        assert not torch.allclose(op(x, border_type="reflect"), op(x, border_type="constant"), atol=1e-2, rtol=1e-2)
    # ScalePyramid
    sp, _, _ = t.ScalePyramid(n_levels=3)(z(1, 1, 32, 32))
    assert sp[0].shape == (1, 1, 6, 32, 32)
    sp, _, _ = t.ScalePyramid(n_levels=3)(torch.rand(1, 1, 31, 31, generator=torch.Generator().manual_seed(0)).cuda())
    for level in sp:
        for img in level:
            per_blur = img.squeeze().reshape(3 + 3, -1).max(dim=1)[0]
            assert torch.argmax(per_blur).item() == 0  # every further blur lowers the maximum
    blob = z(1, 1, 16, 16)
    blob[..., 6:10, 6:10] = 1.0
    sp, _, _ = t.ScalePyramid(n_levels=3)(blob)
    for level in sp:
        for img in level:
            img = img.squeeze()
            assert torch.allclose(img, img.flip(1), atol=1e-6) and torch.allclose(img, img.flip(2), atol=1e-6)
    _, sigmas, _ = t.ScalePyramid(n_levels=1, init_sigma=0.25)(torch.rand(1, 1, 32, 32).cuda())
    assert torch.allclose(sigmas[0][0].cpu(), torch.tensor([0.5, 0.5, 1.0, 2.0]), atol=1e-3)
    assert torch.allclose(sigmas[1][0].cpu(), torch.tensor([0.25, 0.5, 1.0, 2.0]), atol=1e-3)
    # build_pyramid / build_laplacian_pyramid level shapes (:202-211, 230-239)
    img = torch.rand(2, 3, 64, 64).cuda()
    for levels in (2, 3, 4):
        for pyr in (t.build_pyramid(img, levels), t.build_laplacian_pyramid(img, levels)):
            assert len(pyr) == levels
            assert all(p.shape == (2, 3, 64 // 2**i, 64 // 2**i) for i, p in enumerate(pyr))


def test_separable_pyrdown_variant(oracle, monkeypatch):
    """pyrdown_separable (KM_PYRDOWN_ALGO=separable at load, km_config_set here) (opt-in, csrc/km_pyramid.hip): the 5 + 5 tap evaluation of the factor-2 path agrees with the
    25-tap chain to a few ulp, in every border mode and storage dtype; without the variable the default stays bit-identical."""
    from kornia_amd import _native as _N

    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, 3, 70, 264, generator=g) * 2 - 0.5
    for border in BORDERS:
        ref = oracle.pyrdown(x, border)
        _N.lib().km_config_set(b"pyrdown_separable", 1)
        sep = T().pyrdown(x.cuda(), border).cpu()
        up = T().pyrup(x[:, :, :20, :40].cuda(), border).cpu()
        _N.lib().km_config_set(b"pyrdown_separable", 0)
        assert torch.allclose(sep, ref, atol=1e-6, rtol=0) and not torch.equal(sep, ref)
        assert torch.allclose(up, oracle.pyrup(x[:, :, :20, :40], border), atol=1e-6, rtol=0)
        assert torch.equal(T().pyrdown(x.cuda(), border).cpu(), ref)
    for dt in (torch.bfloat16, torch.float16):
        xh = x.clamp(0, 1).to(dt)
        _N.lib().km_config_set(b"pyrdown_separable", 1)
        out = T().pyrdown(xh.cuda()).cpu()
        _N.lib().km_config_set(b"pyrdown_separable", 0)
        assert out.dtype == dt and (out.float() - oracle.pyrdown(xh.float())).abs().max().item() <= 1e-2
