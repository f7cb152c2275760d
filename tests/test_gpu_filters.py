"""GPU parity of the filter path (filter2d, filter2d_separable, gaussian_blur2d, spatial_gradient,
sobel, transform_points) against the CPU oracle, through the kornia-compatible API / C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BORDERS = ["constant", "reflect", "replicate", "circular"]


def _x(B=4, C=3, H=41, W=70, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(B, C, H, W, generator=g), g


@pytest.mark.parametrize("behaviour", ["corr", "conv"])
@pytest.mark.parametrize("padding", ["same", "valid"])
@pytest.mark.parametrize("kshape", [(1, 3, 3), (1, 5, 6), (1, 2, 2), (4, 3, 5), (2, 4, 3), (1, 1, 7)])
@pytest.mark.parametrize("border", BORDERS)
def test_filter2d_forward_bit_exact(oracle, border, kshape, padding, behaviour):
    import kornia_amd as K

    x, g = _x()
    k = torch.rand(*kshape, generator=g)
    ref = oracle.filter2d(x, k, border, False, padding, behaviour)
    out = K.filter2d(x.cuda(), k.cuda(), border, False, padding, behaviour).cpu()
    assert out.shape == ref.shape
    assert torch.equal(out, ref), f"max |d| = {(out - ref).abs().max().item():.3e}"
    # normalized=True: the L1 norm is a torch reduction evaluated on the device (summation order differs
    # from the CPU's by an ulp), so the taps - not the filter - differ in the last bit
    refn = oracle.filter2d(x, k, border, True, padding, behaviour)
    outn = K.filter2d(x.cuda(), k.cuda(), border, True, padding, behaviour).cpu()
    assert torch.allclose(outn, refn, atol=1e-6, rtol=0)


@pytest.mark.parametrize("padding", ["same", "valid"])
@pytest.mark.parametrize("kshape", [(1, 3, 3), (1, 5, 6), (1, 2, 2), (4, 3, 5), (2, 4, 3)])
@pytest.mark.parametrize("border", BORDERS)
def test_filter2d_backward(oracle, border, kshape, padding):
    import kornia_amd as K

    x, g = _x()
    k = torch.rand(*kshape, generator=g)
    xg, kg = x.cuda().requires_grad_(), k.cuda().requires_grad_()
    y = K.filter2d(xg, kg, border, False, padding)
    go = torch.rand(y.shape, generator=g)
    y.backward(go.cuda())
    gx_o, gk_o = oracle.filter2d_backward(go, x, k, border, False, padding)
    assert torch.allclose(xg.grad.cpu(), gx_o, atol=1e-5, rtol=1e-5), (xg.grad.cpu() - gx_o).abs().max()
    assert torch.allclose(kg.grad.cpu(), gk_o, atol=1e-2, rtol=1e-4), (kg.grad.cpu() - gk_o).abs().max()


@pytest.mark.parametrize("padding", ["same", "valid"])
@pytest.mark.parametrize("ks", [(5, 5), (3, 7), (1, 9), (4, 2), (11, 3)])
@pytest.mark.parametrize("Bk", [1, 4, 2])
@pytest.mark.parametrize("border", BORDERS)
def test_filter2d_separable_fused(oracle, border, Bk, ks, padding):
    import kornia_amd as K

    x, g = _x()
    kH, kW = ks
    kx, ky = torch.rand(Bk, kW, generator=g), torch.rand(Bk, kH, generator=g)
    ref = oracle.filter2d_separable(x, kx, ky, border, False, padding)
    xg = x.cuda().requires_grad_()
    out = K.filter2d_separable(xg, kx.cuda(), ky.cuda(), border, False, padding)
    assert torch.equal(out.detach().cpu(), ref), f"max |d| = {(out.detach().cpu() - ref).abs().max().item():.3e}"
    go = torch.rand(ref.shape, generator=g)
    out.backward(go.cuda())
    gx_o = oracle.filter2d_separable_backward(go, x, kx, ky, border, False, padding)
    assert torch.allclose(xg.grad.cpu(), gx_o, atol=1e-5, rtol=1e-5), (xg.grad.cpu() - gx_o).abs().max()


def test_filter2d_separable_kernel_grads_use_two_pass(oracle):
    import kornia_amd as K

    x, g = _x(B=2, C=2, H=12, W=15)
    kx = torch.rand(1, 5, generator=g).cuda().requires_grad_()
    ky = torch.rand(1, 3, generator=g).cuda().requires_grad_()
    y = K.filter2d_separable(x.cuda(), kx, ky)
    y.sum().backward()
    assert kx.grad is not None and ky.grad is not None and torch.isfinite(kx.grad).all()


@pytest.mark.parametrize("border", BORDERS)
def test_gaussian_blur2d(oracle, border):
    import kornia_amd as K

    x, g = _x()
    ref = oracle.gaussian_blur2d(x, (5, 5), (1.5, 1.5), border)
    assert torch.equal(K.gaussian_blur2d(x.cuda(), (5, 5), (1.5, 1.5), border).cpu(), ref)
    assert torch.equal(K.GaussianBlur2d(5, (1.5, 1.5), border)(x.cuda()).cpu(), ref)
    ref2 = oracle.gaussian_blur2d(x, (5, 5), (1.5, 1.5), border, separable=False)
    out2 = K.gaussian_blur2d(x.cuda(), (5, 5), (1.5, 1.5), border, separable=False).cpu()
    assert torch.allclose(out2, ref2, atol=2e-7)  # 2-D taps: exp() evaluated on the GPU vs CPU (1 ulp)
    sig = torch.rand(4, 2, generator=g) + 0.5
    ref3 = oracle.gaussian_blur2d(x, (3, 7), sig, border)
    out3 = K.gaussian_blur2d(x.cuda(), (3, 7), sig.cuda(), border).cpu()
    assert torch.allclose(out3, ref3, atol=3e-7)
    sig2 = torch.rand(2, 2, generator=g) + 0.5  # Bk = 2 divides B = 4 -> kernel index b % 2
    assert torch.allclose(K.gaussian_blur2d(x.cuda(), (3, 3), sig2.cuda(), border).cpu(), oracle.gaussian_blur2d(x, (3, 3), sig2, border), atol=3e-7)


def test_gaussian_blur2d_backward_and_errors(oracle):
    import kornia_amd as K
    from kornia_amd.core import BaseError, ShapeError, TypeCheckError

    x, g = _x()
    xg = x.cuda().requires_grad_()
    y = K.gaussian_blur2d(xg, (5, 5), (1.5, 1.5))
    go = torch.rand(y.shape, generator=g)
    y.backward(go.cuda())
    assert torch.allclose(xg.grad.cpu(), oracle.gaussian_blur2d_backward(go, x, (5, 5), (1.5, 1.5)), atol=1e-5)
    with pytest.raises(BaseError, match="sigma must be positive"):
        K.gaussian_blur2d(x.cuda(), (5, 5), (0.0, 1.0))
    with pytest.raises(BaseError, match="sigma must be positive"):  # host values are tested on the host
        K.gaussian_blur2d(x.cuda(), (5, 5), torch.tensor([[1.0, -1.0]]))
    # values already on the device are not read back by default (no stream drain) ...
    from kornia_amd.core.check import set_device_value_checks

    bad = torch.tensor([[1.0, -1.0]]).cuda()
    if bad.device.type != "cpu":  # (under the host build of the kernels "cuda" tensors are host tensors: checked like host data)
        K.gaussian_blur2d(x.cuda(), (5, 5), bad)
        # ... and a zero is the limit of the Gaussian for sigma -> 0, the identity kernel, not 0 / 0 taps (km_gaussian_taps_fwd)
        zero = torch.tensor([[0.0, 1.2]]).cuda()
        y0 = K.gaussian_blur2d(x.cuda(), (5, 5), zero)
        kx = K.filters.get_gaussian_kernel1d(5, torch.tensor([[1.2]])).cuda()
        ident = torch.tensor([[0.0, 0.0, 1.0, 0.0, 0.0]]).cuda()
        assert torch.isfinite(y0).all() and torch.allclose(y0, K.filter2d_separable(x.cuda(), kx, ident), atol=3e-7)
    old = set_device_value_checks(True)  # ... unless the reference's synchronising check is asked for
    try:
        with pytest.raises(BaseError, match="sigma must be positive"):
            K.gaussian_blur2d(x.cuda(), (5, 5), bad)
    finally:
        set_device_value_checks(old)
    with pytest.raises(BaseError, match="Kernel size must be"):
        K.gaussian_blur2d(x.cuda(), (4, 5), (1.0, 1.0))
    with pytest.raises(ShapeError):
        K.gaussian_blur2d(x.cuda()[0], (5, 5), (1.0, 1.0))
    with pytest.raises(TypeCheckError):
        K.filter2d([1, 2], torch.ones(1, 3, 3))
    with pytest.raises(BaseError, match="Invalid border, a. Ex"):
        K.filter2d(x.cuda(), torch.ones(1, 3, 3).cuda(), border_type="a")


@pytest.mark.parametrize("normalized", [True, False])
@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("mode", ["sobel", "diff"])
def test_spatial_gradient(oracle, mode, order, normalized):
    import kornia_amd as K

    x, g = _x()
    ref = oracle.spatial_gradient(x, mode, order, normalized)
    xg = x.cuda().requires_grad_()
    out = K.spatial_gradient(xg, mode, order, normalized)
    assert out.shape == ref.shape and out.is_contiguous()
    assert torch.equal(out.detach().cpu(), ref)
    go = torch.rand(ref.shape, generator=g)
    out.backward(go.cuda())
    assert torch.allclose(xg.grad.cpu(), oracle.spatial_gradient_backward(go, x, mode, order, normalized), atol=1e-4 if not normalized else 1e-5)


@pytest.mark.parametrize("shape", [(2, 3, 64, 72), (3, 1, 37, 128), (1, 2, 5, 8), (1, 1, 1, 264), (2, 1, 140, 260)])
@pytest.mark.parametrize("mode,order", [("sobel", 1), ("diff", 1), ("sobel", 2), ("diff", 2)])
def test_register_tiled_spatial_gradient(oracle, mode, order, shape):
    """W % 4 == 0 takes km_spatial_gradient_reg_kernel: bit-identical to the oracle, sobel magnitude fused."""
    import kornia_amd as K

    x, g = _x(*shape, seed=11)
    ref = oracle.spatial_gradient(x, mode, order, True)
    out = K.spatial_gradient(x.cuda(), mode, order, True)
    assert out.shape == ref.shape and out.is_contiguous()
    assert torch.equal(out.cpu(), ref), f"max |d| = {(out.cpu() - ref).abs().max().item():.3e}"
    # adjoint: rotated-stack interior (km_spatial_gradient_bwd_reg_kernel) + exact 1-pixel frame
    go = torch.rand(ref.shape, generator=g)
    xg = x.cuda().requires_grad_()
    K.spatial_gradient(xg, mode, order, True).backward(go.cuda())
    gx_o = oracle.spatial_gradient_backward(go, x, mode, order, True)
    assert torch.allclose(xg.grad.cpu(), gx_o, atol=1e-5, rtol=1e-5), (xg.grad.cpu() - gx_o).abs().max()
    if mode == "sobel" and order == 1:
        assert torch.equal(K.sobel(x.cuda()).cpu(), oracle.sobel(x))
        xb = x.bfloat16()
        ob = K.spatial_gradient(xb.cuda(), mode, order, True).float().cpu()
        assert torch.allclose(ob, oracle.spatial_gradient(xb.float(), mode, order, True), atol=2e-2)


@pytest.mark.parametrize("shape", [(2, 3, 64, 72), (3, 2, 37, 128), (2, 1, 5, 8), (1, 1, 140, 260)])
@pytest.mark.parametrize("K", [3, 5, 7])
@pytest.mark.parametrize("border", BORDERS)
def test_register_tiled_filter2d(oracle, border, K, shape):
    """Square odd 3/5/7 kernels with W % 4 == 0 take km_filter2d_reg_kernel: bit-identical to the oracle."""
    import kornia_amd as K_

    B, C, H, W = shape
    if border == "reflect" and (K - 1) // 2 >= min(H, W):
        pytest.skip("reflect pad wider than the image")
    x, g = _x(*shape, seed=5)
    for nb in (1, B):
        k = torch.rand(nb, K, K, generator=g) - 0.3
        ref = oracle.filter2d(x, k, border)
        out = K_.filter2d(x.cuda(), k.cuda(), border)
        assert torch.equal(out.cpu(), ref), f"max |d| = {(out.cpu() - ref).abs().max().item():.3e}"
    # adjoint: rotated-tap interior from the register-tiled kernel + exact pad-wide frame (km_filter2d_bwd_frame_kernel)
    if True:
        for nb in (1, B):
            k = torch.rand(nb, K, K, generator=g) - 0.3
            go = torch.rand(B, C, H, W, generator=g)
            xg, kg = x.cuda().requires_grad_(), k.cuda().requires_grad_()
            K_.filter2d(xg, kg, border).backward(go.cuda())
            gx_o, gk_o = oracle.filter2d_backward(go, x, k, border)
            assert torch.allclose(xg.grad.cpu(), gx_o, atol=2e-5, rtol=1e-5), (xg.grad.cpu() - gx_o).abs().max()
            assert torch.allclose(kg.grad.cpu(), gk_o, atol=2e-3, rtol=1e-4)
    kb = torch.rand(1, K, K, generator=g)
    ob = K_.filter2d(x.bfloat16().cuda(), kb.cuda(), border).float().cpu()
    assert torch.allclose(ob, oracle.filter2d(x.bfloat16().float(), kb.bfloat16().float(), border), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("shape", [(2, 3, 70, 90), (1, 2, 224, 224), (2, 1, 33, 150)])
@pytest.mark.parametrize("ksize", [(11, 11), (23, 23), (15, 10), (33, 13), (12, 31)])
@pytest.mark.parametrize("border", BORDERS)
def test_large_separable_kernels(oracle, border, ksize, shape):
    """k >= 10 takes km_filter_sep_big_fwd_kernel (sliding-window LDS kernel): bit-identical to the oracle; the adjoint runs
    as two generic passes when the fused one does not fit in LDS."""
    import kornia_amd as K_

    B, C, H, W = shape
    kH, kW = ksize
    if border == "reflect" and ((kH - 1) // 2 + (1 - kH % 2) >= H or (kW - 1) // 2 + (1 - kW % 2) >= W):
        pytest.skip("reflect pad wider than the image")
    x, g = _x(*shape, seed=13)
    kx, ky = torch.rand(1, kW, generator=g), torch.rand(B, kH, generator=g)[: (1 if kH % 2 else B)]
    kx = kx.expand(ky.shape[0], -1).contiguous()
    ref = oracle.filter2d_separable(x, kx, ky, border)
    xg = x.cuda().requires_grad_()
    out = K_.filter2d_separable(xg, kx.cuda(), ky.cuda(), border)
    assert torch.equal(out.detach().cpu(), ref), f"max |d| = {(out.detach().cpu() - ref).abs().max().item():.3e}"
    go = torch.rand(ref.shape, generator=g)
    out.backward(go.cuda())
    gx_o = oracle.filter2d_separable_backward(go, x, kx, ky, border)
    assert torch.allclose(xg.grad.cpu(), gx_o, atol=5e-5, rtol=1e-4), (xg.grad.cpu() - gx_o).abs().max()
    if ksize == (23, 23):
        refv = oracle.filter2d_separable(x, kx, ky, border, padding="valid")
        assert torch.equal(K_.filter2d_separable(x.cuda(), kx.cuda(), ky.cuda(), border, padding="valid").cpu(), refv)
        assert torch.equal(K_.gaussian_blur2d(x.cuda(), (23, 23), (2.3, 3.1), border).cpu(), oracle.gaussian_blur2d(x, (23, 23), (2.3, 3.1), border))


def test_sobel(oracle):
    import kornia_amd as K

    x, g = _x()
    ref = oracle.sobel(x)
    assert torch.equal(K.sobel(x.cuda()).cpu(), ref)
    assert torch.equal(K.Sobel()(x.cuda()).cpu(), ref)
    xg = x.cuda().requires_grad_()
    out = K.sobel(xg)
    assert torch.allclose(out.detach().cpu(), ref, atol=1e-6)
    out.sum().backward()
    assert torch.isfinite(xg.grad).all()


def test_transform_points(oracle):
    import kornia_amd as K

    g = torch.Generator().manual_seed(3)
    for D in (2, 3):
        T = torch.eye(D + 1)[None] + 0.1 * torch.randn(4, D + 1, D + 1, generator=g)
        for n_pts in (1000, 999, 1002, 3):  # whole 16-byte words per lane (vector kernel) and ragged counts (scalar kernel)
            P = torch.rand(4, n_pts, D, generator=g) * 2 - 1
            assert torch.equal(K.transform_points(T.cuda(), P.cuda()).cpu(), oracle.transform_points(T, P))
            assert torch.equal(K.transform_points(T[:1].cuda(), P.cuda()).cpu(), oracle.transform_points(T[:1], P))
        P = torch.rand(4, 1000, D, generator=g) * 2 - 1
        Td, Pd = T.double().cuda().requires_grad_(), P[:, :5].double().cuda().requires_grad_()
        assert torch.autograd.gradcheck(K.transform_points, (Td, Pd), nondet_tol=1e-8)
    assert K.transform_points(torch.eye(3)[None].cuda(), torch.zeros(1, 0, 2).cuda()).shape == (1, 0, 2)
    with pytest.raises(ValueError):
        K.transform_points(torch.eye(3)[None].expand(2, 3, 3).cuda(), torch.zeros(3, 4, 2).cuda())


def test_filters_fp64_gradcheck():
    import kornia_amd as K

    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 2, 9, 11, generator=g).double().cuda().requires_grad_()
    k = torch.rand(1, 3, 4, generator=g).double().cuda().requires_grad_()
    for border in BORDERS:
        assert torch.autograd.gradcheck(lambda a, b: K.filter2d(a, b, border), (x, k), nondet_tol=1e-8, fast_mode=True)
        assert torch.autograd.gradcheck(lambda a: K.gaussian_blur2d(a, (5, 3), (1.2, 0.8), border), (x,), fast_mode=True)
    assert torch.autograd.gradcheck(lambda a: K.spatial_gradient(a, "sobel", 2), (x,), fast_mode=True)
    assert torch.autograd.gradcheck(lambda a: K.sobel(a), (x,), fast_mode=True)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float16, 2e-3)])
def test_half_precision_blur(oracle, dtype, tol):
    import kornia_amd as K

    x, _ = _x()
    out = K.gaussian_blur2d(x.to(dtype).cuda(), (5, 5), (1.5, 1.5)).float().cpu()
    ref = oracle.gaussian_blur2d(x.to(dtype).float(), (5, 5), (1.5, 1.5))
    assert (out - ref).abs().max().item() <= tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("K,border,shape", [(5, "reflect", (2, 3, 70, 72)), (3, "constant", (3, 1, 37, 128)), (7, "replicate", (1, 2, 64, 264)), (9, "circular", (2, 1, 41, 16))])
def test_half_precision_blur_strip_heights(dtype, K, border, shape, monkeypatch):
    """The strip height of a launch is chosen from storage type, direction, kernel size and the number of waves (km_blur_fast.hip
    km_blur_rows: 32, 16 or 8 rows).  Same arithmetic per output pixel, so forward and adjoint are bit-identical between the three forms."""
    import kornia_amd as K_

    g = torch.Generator().manual_seed(11)
    x = torch.rand(*shape, generator=g).to(dtype).cuda()
    kx, ky = torch.rand(1, K, generator=g).cuda(), torch.rand(1, K, generator=g).cuda()
    go = torch.rand(*shape, generator=g).to(dtype).cuda()
    from kornia_amd import _native as N

    res = {}
    for rows in ("8", "16", "32"):
        prev = N.lib().km_config_set(b"blur_rows", int(rows))  # (the KM_BLUR_ROWS switch: read once at load, set explicitly here)
        try:
            xg = x.clone().requires_grad_()
            out = K_.filter2d_separable(xg, kx, ky, border)
            out.backward(go)
        finally:
            N.lib().km_config_set(b"blur_rows", prev)
        res[rows] = (out.detach(), xg.grad)
    assert torch.equal(res["8"][0], res["32"][0]) and torch.equal(res["8"][1], res["32"][1])
    assert torch.equal(res["16"][0], res["32"][0]) and torch.equal(res["16"][1], res["32"][1])
    ref = K_.filter2d_separable(x.float(), kx, ky, border)
    # against the fp32 evaluation: two roundings to the storage type (row pass, result), relative to the largest value
    assert (res["8"][0].float() - ref).abs().max().item() <= (1.2e-2 if dtype == torch.bfloat16 else 2e-3) * ref.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 3, 64, 72), (3, 1, 37, 128), (4, 2, 16, 264)])
@pytest.mark.parametrize("K", [3, 5, 7, 9])
@pytest.mark.parametrize("Bk", [1, "B"])
@pytest.mark.parametrize("border", BORDERS)
@pytest.mark.parametrize("rows", ["8", "16", "32"])
def test_register_tiled_blur_fast_path(oracle, border, Bk, K, shape, rows, monkeypatch):
    """W % 4 == 0, square odd K <= 9: served by csrc/km_blur_fast.hip (no LDS). Forward bit-exact;
    the adjoint against the oracle's scatter-form adjoint.  KM_BLUR_ROWS: each of the three strip heights (see
    test_half_precision_blur_strip_heights), against the oracle."""
    import kornia_amd as K_
    from kornia_amd import _native as N

    N.lib().km_config_set(b"blur_rows", int(rows))  # (the autouse fixture at the end of this file restores the default)

    B, C, H, W = shape
    g = torch.Generator().manual_seed(7)
    x = torch.rand(B, C, H, W, generator=g)
    nb = 1 if Bk == 1 else B
    kx, ky = torch.rand(nb, K, generator=g), torch.rand(nb, K, generator=g)
    ref = oracle.filter2d_separable(x, kx, ky, border)
    xg = x.cuda().requires_grad_()
    out = K_.filter2d_separable(xg, kx.cuda(), ky.cuda(), border)
    assert torch.equal(out.detach().cpu(), ref), f"max |d| = {(out.detach().cpu() - ref).abs().max().item():.3e}"
    go = torch.rand(ref.shape, generator=g)
    out.backward(go.cuda())
    gx_o = oracle.filter2d_separable_backward(go, x, kx, ky, border)
    assert torch.allclose(xg.grad.cpu(), gx_o, atol=2e-5, rtol=1e-5), (xg.grad.cpu() - gx_o).abs().max()


def test_blur_adjoint_identity_at_full_size():
    """Config-2 sized blur: adjoint identity <A x, y> == <x, A^T y> (size-independent property)."""
    import kornia_amd as K_

    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(8, 3, 512, 512, device="cuda", generator=g, dtype=torch.float32).requires_grad_()
    y = torch.rand(8, 3, 512, 512, device="cuda", generator=g)
    for border in BORDERS:
        x.grad = None
        Ax = K_.gaussian_blur2d(x, (5, 5), (1.5, 1.5), border)
        Ax.backward(y)
        lhs = (Ax.detach().double() * y.double()).sum()
        rhs = (x.detach().double() * x.grad.double()).sum()
        assert abs(lhs - rhs).item() / abs(lhs).item() < 1e-6
        ones = torch.ones(1, 1, 512, 512, device="cuda")
        if border != "constant":  # blur of a constant image is the same constant (taps sum to 1)
            assert torch.allclose(K_.gaussian_blur2d(ones, (5, 5), (1.5, 1.5), border), ones, atol=1e-6)


@pytest.fixture(autouse=True)
def _reset_launch_policy():
    """Tests above switch launch-policy entries of the library (km_config_set); every test leaves the defaults behind."""
    yield
    from kornia_amd import _native as N

    if N.is_built():
        try:
            N.lib().km_config_set(b"blur_rows", 0)
        except Exception:
            pass
