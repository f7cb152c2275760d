"""GPU parity of the explicit-grid sampler (km_grid_sample2d_*: F.grid_sample / remap) against
fixtures produced by the real reference (tests/golden/grid_sample.npz) and against the CPU oracle."""
import pytest
import torch

from _util import golden

pytestmark = pytest.mark.gpu
MODES = ["bilinear", "nearest", "bicubic"]
PADS = ["zeros", "border", "reflection"]


def _t(d, k):
    return torch.from_numpy(d[k])


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("pad", PADS)
@pytest.mark.parametrize("mode", MODES)
def test_grid_sample_matches_reference_fixture(mode, pad, align):
    import kornia_amd as K

    d = golden("grid_sample")
    tag = f"{mode}_{pad}_{int(align)}"
    x, grid = _t(d, "x").cuda().requires_grad_(), _t(d, "grid").cuda().requires_grad_()
    y = K.grid_sample(x, grid, mode, pad, align)
    if mode == "bicubic":
        assert torch.allclose(y.detach().cpu(), _t(d, "y_" + tag), atol=5e-7, rtol=0)
    else:
        assert torch.equal(y.detach().cpu(), _t(d, "y_" + tag))
    y.backward(_t(d, "go").cuda())
    assert torch.allclose(x.grad.cpu(), _t(d, "gx_" + tag), atol=3e-6, rtol=1e-6)
    assert torch.allclose(grid.grad.cpu(), _t(d, "gg_" + tag), atol=3e-5, rtol=1e-5)


def test_remap_matches_reference_fixture():
    import kornia_amd as K

    d = golden("grid_sample")
    x, mx, my, g = (_t(d, k).cuda() for k in ("x", "map_x", "map_y", "grid"))
    assert torch.equal(K.remap(x, mx, my).cpu(), _t(d, "remap"))
    assert torch.equal(K.remap(x, mx, my, align_corners=True).cpu(), _t(d, "remap_ac"))
    assert torch.equal(K.remap(x, mx[:1], my[:1], mode="nearest", padding_mode="border").cpu(), _t(d, "remap_bcast_nearest"))
    assert torch.equal(K.remap(x, g[..., 0], g[..., 1], padding_mode="reflection", normalized_coordinates=True).cpu(), _t(d, "remap_norm"))


@pytest.mark.parametrize("shape", [(3, 1, 37, 53, 29, 64), (2, 4, 16, 16, 70, 33), (1, 3, 5, 4, 1, 1)])
@pytest.mark.parametrize("mode", MODES)
def test_grid_sample_vs_oracle(oracle, mode, shape):
    import kornia_amd as K

    B, C, H, W, h, w = shape
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, C, H, W, generator=g)
    grid = torch.rand(B, h, w, 2, generator=g) * 2.4 - 1.2
    go = torch.rand(B, C, h, w, generator=g)
    for pad in PADS:
        ref = oracle.grid_sample(x, grid, mode, pad, False)
        xg, gg = x.cuda().requires_grad_(), grid.cuda().requires_grad_()
        out = K.grid_sample(xg, gg, mode, pad, False)
        assert torch.equal(out.detach().cpu(), ref) if mode != "bicubic" else torch.allclose(out.detach().cpu(), ref, atol=1e-6)
        out.backward(go.cuda())
        gx_o, gg_o = oracle.grid_sample_backward(go, x, grid, mode, pad, False)
        assert torch.allclose(xg.grad.cpu(), gx_o, atol=1e-5, rtol=1e-5)
        assert torch.allclose(gg.grad.cpu(), gg_o, atol=1e-4, rtol=1e-4)


def test_shared_grid_is_broadcast_and_its_gradient_summed(oracle):
    import kornia_amd as K

    g = torch.Generator().manual_seed(4)
    x = torch.rand(5, 2, 20, 24, generator=g)
    grid = torch.rand(1, 11, 13, 2, generator=g) * 2 - 1
    go = torch.rand(5, 2, 11, 13, generator=g)
    xg, gg = x.cuda().requires_grad_(), grid.cuda().requires_grad_()
    out = K.grid_sample(xg, gg, "bilinear", "zeros", True)
    assert torch.equal(out.detach().cpu(), oracle.grid_sample(x, grid.expand(5, -1, -1, -1), "bilinear", "zeros", True))
    out.backward(go.cuda())
    _, gg_o = oracle.grid_sample_backward(go, x, grid.expand(5, -1, -1, -1).contiguous(), "bilinear", "zeros", True)
    assert gg.grad.shape == (1, 11, 13, 2)
    assert torch.allclose(gg.grad.cpu(), gg_o.sum(0, keepdim=True), atol=1e-4, rtol=1e-4)


def test_half_precision_and_errors():
    import kornia_amd as K

    x = torch.rand(2, 3, 32, 32, device="cuda")
    grid = torch.rand(2, 16, 16, 2, device="cuda") * 2 - 1
    ref = K.grid_sample(x, grid)
    for dt in (torch.bfloat16, torch.float16):
        out = K.grid_sample(x.to(dt), grid.to(dt))
        assert out.dtype == dt and torch.allclose(out.float(), ref, atol=5e-2)
    with pytest.raises(RuntimeError):
        K.grid_sample(x, grid.double())
    with pytest.raises(RuntimeError):
        K.grid_sample(x, grid[:1].expand(3, -1, -1, -1))
    with pytest.raises(ValueError):
        K.grid_sample(x, grid, mode="cubic")
    with pytest.raises(Exception):
        K.remap(x, grid[..., 0], grid[0, ..., 1])


def test_remap_identity_at_full_size():
    """1080p identity map in pixel coordinates with align_corners=True reproduces the image exactly (size-independent property)."""
    import kornia_amd as K

    x = torch.rand(4, 1, 1080, 1920, device="cuda")
    ys, xs = torch.meshgrid(torch.arange(1080.0, device="cuda"), torch.arange(1920.0, device="cuda"), indexing="ij")
    out = K.remap(x, xs[None], ys[None], mode="nearest", align_corners=True)
    assert torch.equal(out, x)
    out = K.remap(x, xs[None], ys[None], mode="bilinear", align_corners=True)
    assert torch.allclose(out, x, atol=2e-4)
