"""GPU: image pyramid (km_pyrdown_fwd, km_resize_bilinear_fwd; kornia_amd/geometry/transform/pyramid.py) against the oracle
(bit for bit in fp32 / fp64) and against fixtures produced by the real reference (tests/golden/pyramid.npz; ATen's CPU
bilinear kernel groups its four products differently, hence 1e-6 there).  tests/test_emulated_kernels.py runs the same
cases on the host build of the kernels."""
import pytest
import torch

from _util import golden as _golden_np

pytestmark = pytest.mark.gpu
BORDERS = ["constant", "reflect", "replicate", "circular"]


def golden(name):
    return {k: torch.from_numpy(v) for k, v in _golden_np(name).items()}


def T():
    import kornia_amd as K

    return K.geometry.transform


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("border", BORDERS)
@pytest.mark.parametrize("shape,factor", [((2, 3, 32, 48), 2.0), ((1, 2, 37, 29), 2.0), ((1, 1, 6, 7), 2.0), ((2, 1, 64, 130), 2.0), ((1, 2, 200, 520), 2.0), ((1, 1, 70, 264), 2.0), ((3, 1, 4, 8), 2.0),
                                          ((1, 2, 37, 29), 1.5), ((1, 1, 45, 70), 3.0), ((1, 1, 9, 200), 4.0)])
def test_pyrdown_bit_exact_vs_oracle(oracle, shape, factor, border, align):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.rand(*shape, generator=g) * 2 - 0.5
    for dt in (torch.float32, torch.float64):
        ref = oracle.pyrdown(x.to(dt), border, align, factor)
        out = T().pyrdown(x.to(dt).cuda(), border, align, factor).cpu()
        assert out.shape == ref.shape and out.dtype == dt
        assert torch.equal(out, ref), (out - ref).abs().max()


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("size", [(7, 9), (64, 64), (20, 33), (1, 1), (50, 3), (42, 68)])
def test_resize_bilinear_bit_exact_vs_oracle(oracle, size, align):
    g = torch.Generator().manual_seed(size[0])
    x = torch.rand(2, 2, 21, 34, generator=g)
    for dt in (torch.float32, torch.float64):
        out = T().resize_bilinear(x.to(dt).cuda(), size, align).cpu()
        assert torch.equal(out, oracle.resize_bilinear(x.to(dt), size, align))
    ref = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=align)
    assert torch.allclose(T().resize_bilinear(x.cuda(), size, align).cpu(), ref, atol=2e-6, rtol=0)


@pytest.mark.parametrize("border", BORDERS)
def test_pyrup_bit_exact_vs_oracle(oracle, border):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 13, 18, generator=g)
    for align in (False, True):
        out = T().pyrup(x.cuda(), border, align).cpu()
        assert out.shape == (2, 3, 26, 36) and torch.equal(out, oracle.pyrup(x, border, align))
    big = torch.rand(1, 2, 70, 132, generator=g)  # several row strips and column tiles of the x2 kernel
    assert torch.equal(T().pyrup(big.cuda(), border).cpu(), oracle.pyrup(big, border))
    assert torch.equal(T().resize_bilinear(big.cuda(), (140, 264)).cpu(), oracle.resize_bilinear(big, (140, 264)))


def test_half_precision(oracle):
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 40, 56, generator=g)
    xo = torch.rand(1, 2, 37, 29, generator=g)
    for dt in (torch.bfloat16, torch.float16):
        out = T().pyrdown(xo.to(dt).cuda(), "replicate", True).cpu()  # general mapping
        assert out.dtype == dt and (out.float() - oracle.pyrdown(xo.to(dt).float(), "replicate", True)).abs().max().item() <= 1e-2
        xr = x.to(dt)
        out = T().pyrdown(xr.cuda()).cpu()
        assert out.dtype == dt and (out.float() - oracle.pyrdown(xr.float())).abs().max().item() <= 1e-2
        r2 = T().resize_bilinear(xr.cuda(), (80, 112)).cpu()
        assert r2.dtype == dt and (r2.float() - oracle.resize_bilinear(xr.float(), (80, 112))).abs().max().item() <= 1e-2
        up = T().pyrup(xr.cuda()).cpu()
        assert up.dtype == dt and (up.float() - oracle.pyrup(xr.float())).abs().max().item() <= 1e-2


def test_vs_reference_fixtures():
    d = golden("pyramid")
    t = T()
    for tag in ("even", "odd", "tiny"):
        x = d["x_" + tag].cuda()
        for border in BORDERS:
            for ac in (0, 1):
                assert torch.allclose(t.pyrdown(x, border, bool(ac)).cpu(), d[f"down_{tag}_{border}_{ac}"], atol=1e-6, rtol=0)
                if tag != "even":
                    assert torch.allclose(t.pyrup(x, border, bool(ac)).cpu(), d[f"up_{tag}_{border}_{ac}"], atol=1e-6, rtol=0)
        a, b = t.pyrdown(x, "reflect", False, 1.5).cpu(), d["down_" + tag + "_factor1p5"]
        assert a.shape == b.shape and torch.allclose(a, b, atol=1e-6, rtol=0)
        a, b = t.PyrDown("replicate", True, 3.0)(x).cpu(), d["down_" + tag + "_factor3"]
        assert a.shape == b.shape and torch.allclose(a, b, atol=1e-6, rtol=0)
    for i, lvl in enumerate(t.build_pyramid(d["x_even"].cuda(), 4)):
        assert torch.allclose(lvl.cpu(), d[f"pyr_even_{i}"], atol=1e-6, rtol=0)
    for i, lvl in enumerate(t.build_laplacian_pyramid(d["x_lap"].cuda(), 3)):
        assert torch.allclose(lvl.cpu(), d[f"lap_{i}"], atol=2e-6, rtol=0)
    assert torch.equal(t.pyrdown(d["lit_pyrdown_in"].cuda(), align_corners=True).cpu(), d["lit_pyrdown_out"])
    assert torch.allclose(t.PyrUp(align_corners=True)(d["lit_pyrup_in"].cuda()).cpu(), d["lit_pyrup_out"], atol=1e-6)


def test_argument_checks():
    from kornia_amd.core.exceptions import BaseError, ShapeError

    x = torch.rand(1, 1, 8, 8).cuda()
    with pytest.raises(ShapeError):
        T().pyrdown(x[0])
    with pytest.raises(BaseError, match="Invalid border"):
        T().pyrdown(x, "mirror")
    with pytest.raises(ShapeError):
        T().build_pyramid(x[0], 2)
    assert len(T().build_pyramid(x, 1)) == 1 and T().build_pyramid(x, 3)[-1].shape == (1, 1, 2, 2)
    assert T().pyrdown(torch.rand(0, 3, 8, 8).cuda()).shape == (0, 3, 4, 4)


def test_pyrdown_fused_equals_two_kernels_at_full_size():
    """BASELINE-size property: the fused launch equals the unfused native blur + ATen's device resize, a constant image
    stays constant, and the op is linear."""
    import kornia_amd as K

    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(16, 3, 512, 512, device="cuda", generator=g)
    y = T().pyrdown(x)
    two = torch.nn.functional.interpolate(K.filter2d(x, T().pyramid._get_pyramid_gaussian_kernel(), "reflect"), size=(256, 256), mode="bilinear", align_corners=False)
    assert y.shape == (16, 3, 256, 256) and torch.allclose(y, two, atol=1e-6, rtol=0)
    c = T().pyrdown(torch.full((2, 3, 512, 512), 0.625, device="cuda"))
    assert torch.equal(c, torch.full_like(c, 0.625))
    z = torch.rand(16, 3, 512, 512, device="cuda", generator=g)
    assert torch.allclose(T().pyrdown(x + 2 * z), y + 2 * T().pyrdown(z), atol=1e-5, rtol=0)


def test_scale_pyramid_vs_reference():
    """ScalePyramid (pyramid.py:151-400): octave stacks, nominal sigmas and pixel distances against the reference's outputs."""
    d = golden("scale_pyramid")
    x = d["x"].cuda()
    for tag, kw in (("default", {}), ("double", {"double_image": True, "n_levels": 2, "extra_levels": 2, "min_size": 20}), ("small_sigma", {"init_sigma": 0.4, "n_levels": 2, "min_size": 10})):
        sp = T().ScalePyramid(**kw).cuda()
        pyr, sig, pd = sp(x)
        assert len(pyr) == int(d[tag + "_n"]) and repr(sp)
        for o in range(len(pyr)):
            assert pyr[o].shape == d[f"{tag}_pyr_{o}"].shape
            assert torch.allclose(pyr[o].cpu(), d[f"{tag}_pyr_{o}"], atol=2e-6, rtol=0), (tag, o, (pyr[o].cpu() - d[f"{tag}_pyr_{o}"]).abs().max())
            assert torch.allclose(sig[o].cpu(), d[f"{tag}_sig_{o}"]) and torch.allclose(pd[o].cpu(), d[f"{tag}_pd_{o}"])


def test_resize_adjoint_matches_aten():
    """km_resize_bilinear_bwd (deterministic gather) against aten::upsample_bilinear2d_backward: up / down scaling, non-integer
    factors, both align_corners settings, a size-1 axis, 16-bit storage."""
    t = T()
    g = torch.Generator().manual_seed(31)
    for (H, W), (oh, ow) in (((12, 16), (24, 32)), ((24, 32), (12, 16)), ((9, 13), (14, 7)), ((17, 5), (5, 31)), ((1, 8), (4, 3)), ((6, 6), (6, 6))):
        for align in (False, True):
            x = torch.rand(2, 3, H, W, generator=g).cuda().requires_grad_()
            go = torch.rand(2, 3, oh, ow, generator=g).cuda()
            (got,) = torch.autograd.grad(t.resize_bilinear(x, (oh, ow), align), x, go)
            ref = torch.ops.aten.upsample_bilinear2d_backward(go, [oh, ow], [2, 3, H, W], align, None, None)
            assert got.shape == ref.shape and (got - ref).abs().max().item() <= 2e-6, ((H, W), (oh, ow), align)
    x = torch.rand(1, 2, 16, 16, generator=g).bfloat16().cuda().requires_grad_()
    (gb,) = torch.autograd.grad(t.resize_bilinear(x, (8, 8)), x, torch.ones(1, 2, 8, 8, dtype=torch.bfloat16).cuda())
    assert gb.dtype == torch.bfloat16 and torch.allclose(gb.float(), torch.full_like(gb.float(), 0.25))
