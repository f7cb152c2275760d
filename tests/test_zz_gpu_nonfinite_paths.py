"""GPU (and the host build of the kernels): matrices that are NOT maps - all zeros (singular), a NaN entry, an inf entry, a projective denominator that vanishes on
a column of the output, entries that overflow - through EVERY warp path the package has, not only the ones the reference fixture pins
(tests/test_gpu_golden.py::test_nonfinite_sampling_coordinates_*): padding modes the reference itself cannot be asked about (its CPU backward reads out of bounds
for a NaN grid under border / reflection), 16-bit storage, the matrix-gradient-only backward, the fused warp + blur, the fused registration loss, large images with
many owner tiles, batches that mix such matrices with ordinary ones.  What is asserted: the call returns (round 6's first device run of such a matrix FAULTED the
device: a tile box of wrapped width), the ordinary samples of the batch are what they are without their neighbours, and for zeros / fill padding the bilinear output
is NaN exactly where the position is not finite.  Runs last in the suite (a fault here would take the process with it)."""
import pytest
import torch

from _util import flagship_homographies

pytestmark = pytest.mark.gpu
nan, inf = float("nan"), float("inf")


def _bad_matrices(B, H, W, g):
    M = flagship_homographies(B, H, W, H, W, g, jitter=3.0)
    M[0] = 0.0
    M[1, 0, 2] = nan
    M[2, 1, 1] = inf
    M[3] = torch.linalg.inv(torch.tensor([[1.0, 0, 0], [0, 1, 0], [0.0625, 0, -0.0625 * (W - 1) / 2]], dtype=torch.float64)).float()
    M[4, 0, 0] = 3e38
    M[5, 2, 2] = -inf
    return M  # samples 6 .. are ordinary


@pytest.mark.parametrize("shape", [(8, 3, 40, 57), (8, 1, 150, 200), (8, 3, 256, 320)])
@pytest.mark.parametrize("pad", ["zeros", "border", "reflection", "fill"])
@pytest.mark.parametrize("mode", ["bilinear", "bicubic", "nearest"])
def test_every_padding_and_interpolation_forward_and_backward(shape, pad, mode):
    import kornia_amd as K

    B, C, H, W = shape
    if pad == "fill" and C != 3:
        pytest.skip("fill_value is RGB in the reference")
    g = torch.Generator().manual_seed(H + W)
    x = torch.rand(B, C, H, W, generator=g)
    M = _bad_matrices(B, H, W, g)
    go = torch.rand(B, C, H, W, generator=g) - 0.5
    kw = dict(mode=mode, padding_mode=pad)
    if pad == "fill":
        kw["fill_value"] = torch.tensor([0.1, 0.2, 0.3])
    for fn, Min in ((K.warp_perspective, M), (K.warp_affine, M[:, :2].contiguous())):
        xg, Mg = x.cuda().requires_grad_(), Min.cuda().requires_grad_()
        y = fn(xg, Mg, (H, W), **kw)
        y.backward(go.cuda())
        torch.cuda.synchronize()
        # the ordinary samples are untouched by their neighbours
        x2, M2 = x[6:].cuda().requires_grad_(), Min[6:].cuda().requires_grad_()
        y2 = fn(x2, M2, (H, W), **kw)
        y2.backward(go[6:].cuda())
        assert torch.equal(y[6:], y2) and torch.isfinite(y2).all()
        assert torch.allclose(xg.grad[6:], x2.grad, atol=1e-6, rtol=0) and torch.isfinite(x2.grad).all()
        assert torch.isfinite(Mg.grad[6:]).all()
        if pad in ("zeros", "fill"):
            assert torch.isfinite(xg.grad).all()  # a pixel whose position is not a number scatters nothing
        # (border / reflection bring an infinite position INTO the image - the clamp of +inf is the last column - with NaN bicubic coefficients:
        # the reference scatters NaN there too, where it does not read out of bounds first; only the ordinary samples are asserted)
        if mode == "bilinear" and pad in ("zeros", "fill"):
            assert y[0].isnan().all() and y[1].isnan().all()  # every position of a NaN matrix is NaN
            assert not torch.isfinite(Mg.grad[0]).any() and not torch.isfinite(Mg.grad[1]).any()
            if fn is K.warp_perspective and W % 2 == 1:
                col = y[3, :, :, (W - 1) // 2]
                assert col.isnan().all() and torch.isfinite(y[3, :, :, : (W - 1) // 2 - 1]).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_16_bit_storage_and_matrix_gradient_only(dtype):
    import kornia_amd as K

    g = torch.Generator().manual_seed(3)
    B, C, H, W = 8, 3, 96, 129
    x = torch.rand(B, C, H, W, generator=g)
    M = _bad_matrices(B, H, W, g)
    go = torch.rand(B, C, H, W, generator=g) - 0.5
    xg, Mg = x.to(dtype).cuda().requires_grad_(), M.cuda().requires_grad_()
    y = K.warp_perspective(xg, Mg, (H, W))
    y.backward(go.to(dtype).cuda())
    assert y[0].isnan().all() and torch.isfinite(y[6:].float()).all() and torch.isfinite(xg.grad.float()).all()
    # only the matrix needs a gradient (config 5's backward: km_warp_gm_kernel), all three coordinate modes
    for fn, Min in ((K.warp_perspective, M), (K.warp_affine, M[:, :2].contiguous()), (K.homography_warp, torch.eye(3).repeat(B, 1, 1) + 0.02 * (M - M[7]).clamp(-1, 1))):
        Mo = Min.cuda().requires_grad_()
        out = fn(x.cuda(), Mo, (H, W))
        out.backward(go.cuda())
        torch.cuda.synchronize()
        assert torch.isfinite(Mo.grad[6:]).all()
        assert not torch.isfinite(Mo.grad[1]).all()


def test_fused_warp_blur_and_registration_loss():
    import kornia_amd as K
    from kornia_amd.geometry.transform import masked_warp_loss, warp_perspective_blur

    g = torch.Generator().manual_seed(4)
    B, C, H, W = 8, 3, 70, 97
    x = torch.rand(B, C, H, W, generator=g)
    M = _bad_matrices(B, H, W, g)
    xg, Mg = x.cuda().requires_grad_(), M.cuda().requires_grad_()
    y = warp_perspective_blur(xg, Mg, (H, W), (5, 5), (1.5, 1.5))
    ref = K.gaussian_blur2d(K.warp_perspective(x.cuda(), M.cuda(), (H, W)), (5, 5), (1.5, 1.5))
    assert torch.equal(y.isnan(), ref.isnan()) and torch.equal(y[6:], ref[6:])
    y.backward(torch.rand(B, C, H, W, generator=g).cuda())
    assert torch.isfinite(xg.grad[6:]).all() and torch.isfinite(Mg.grad[6:]).all()
    # the registration loss: a NaN homography gives a NaN loss (no mask) and a NaN gradient for THAT homography only
    Hn = torch.eye(3).repeat(B, 1, 1) + 0.01 * torch.randn(B, 3, 3, generator=g)
    Hn[1, 0, 2] = nan
    Hn[2, 0, 0] = 3e38
    Hg = Hn.cuda().requires_grad_()
    loss = masked_warp_loss(x.cuda(), x.cuda(), Hg, "l1", threshold=None)
    loss.backward()
    assert loss.isnan() and torch.isfinite(Hg.grad[3:]).all() and not torch.isfinite(Hg.grad[1]).all()
    Hg2 = Hn.cuda().requires_grad_()
    loss2 = masked_warp_loss(x.cuda(), x.cuda(), Hg2, "mse", threshold=0.9)  # the mask drops the pixels whose warped ones are NaN: a finite loss
    loss2.backward()
    assert torch.isfinite(loss2) and torch.isfinite(Hg2.grad[3:]).all()
