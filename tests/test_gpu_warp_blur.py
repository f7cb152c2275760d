"""GPU: the fused warp + separable blur forward (csrc/km_warp_blur.hip, kornia_amd.geometry.transform.warp_perspective_blur / warp_affine_blur)
against the two calls it replaces - bit for bit - and against the oracle; gradients through the op.  Also runs on the host build of the
kernels (tests/test_emulated_kernels.py)."""
import math

import pytest
import torch

from _util import flagship_homographies, rotation_affines

pytestmark = pytest.mark.gpu

BORDERS = ["reflect", "constant", "replicate", "circular"]


def _rot(B, H, W, deg, scale=1.0):
    c_, s_ = scale * math.cos(math.radians(deg)), scale * math.sin(math.radians(deg))
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    return torch.tensor([[c_, s_, (1 - c_) * cx - s_ * cy], [-s_, c_, s_ * cx + (1 - c_) * cy]]).repeat(B, 1, 1)


@pytest.mark.parametrize("border", BORDERS)
@pytest.mark.parametrize("K", [3, 5, 7])
@pytest.mark.parametrize("shape", [(2, 3, 96, 160, 96, 160), (2, 1, 70, 132, 50, 100), (1, 3, 40, 36, 33, 65), (2, 3, 45, 37, 40, 70)])  # (the last: W % 4 != 0 - no 16-byte row loads, every pixel gathers)
def test_fused_forward_is_bit_identical_to_the_two_calls(oracle, shape, K, border):
    """Every border mode of the blur, tiles that hang over the right / bottom edge, kernels of 3 / 5 / 7, grey and RGB: torch.equal to
    gaussian_blur2d(warp_perspective(...)) - which is itself bit-identical to the oracle."""
    import kornia_amd as Km

    T = Km.geometry.transform
    B, C, H, W, h, w = shape
    assert Km._native.lib().km_warp2d_blur_supported(C, H, W, h, w, 1, 0, K, {"constant": 0, "reflect": 1, "replicate": 2, "circular": 3}[border], 0) == 1
    g = torch.Generator().manual_seed(K * 7 + h)
    x = torch.rand(B, C, H, W, generator=g)
    M = flagship_homographies(B, H, W, h, w, g, jitter=4.0)
    sig = (1.5, 0.9)
    y = T.warp_perspective_blur(x.cuda(), M.cuda(), (h, w), (K, K), sig, border)
    y2 = Km.filters.gaussian_blur2d(T.warp_perspective(x.cuda(), M.cuda(), (h, w)), (K, K), sig, border)
    assert torch.equal(y, y2), (y - y2).abs().max()
    yo = oracle.gaussian_blur2d(oracle.warp_perspective(x, M, (h, w)), (K, K), sig, border)
    assert torch.equal(y.cpu(), yo), (y.cpu() - yo).abs().max()


@pytest.mark.parametrize("deg,scale", [(5.0, 1.0), (30.0, 1.0), (45.0, 0.8), (2.0, 0.5), (0.0, 2.5)])
def test_rotations_and_scales_through_the_box_and_the_gather_fallback(deg, scale):
    """Beyond a few degrees (or under minification) the region's source box does not fit the LDS tile and the block gathers; the result
    never depends on which path ran."""
    import kornia_amd as Km

    T = Km.geometry.transform
    B, C, H, W = 2, 3, 128, 192
    x = torch.rand(B, C, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    A = _rot(B, H, W, deg, scale).cuda()
    for align in (True, False):
        y = T.warp_affine_blur(x, A, (H, W), (5, 5), (1.5, 1.5), "reflect", align_corners=align)
        y2 = Km.filters.gaussian_blur2d(T.warp_affine(x, A, (H, W), align_corners=align), (5, 5), (1.5, 1.5), "reflect")
        assert torch.equal(y, y2), (y - y2).abs().max()
    # a shared (1,2,3) matrix
    y = T.warp_affine_blur(x, A[:1], (H, W), (5, 5), (1.5, 1.5))
    assert torch.equal(y, Km.filters.gaussian_blur2d(T.warp_affine(x, A[:1], (H, W)), (5, 5), (1.5, 1.5)))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_16_bit_storage(dtype):
    """The warped values are rounded to the storage type before the blur reads them and the row pass once more, as in the two calls."""
    import kornia_amd as Km

    T = Km.geometry.transform
    B, C, H, W = 2, 3, 72, 136
    g = torch.Generator().manual_seed(3)
    x = torch.rand(B, C, H, W, generator=g).to(dtype).cuda()
    M = flagship_homographies(B, H, W, H, W, g, jitter=3.0).cuda()
    y = T.warp_perspective_blur(x, M, (H, W), (5, 5), (1.5, 1.5))
    y2 = Km.filters.gaussian_blur2d(T.warp_perspective(x, M, (H, W)), (5, 5), (1.5, 1.5))
    assert y.dtype == dtype and torch.equal(y, y2)


def test_gradients_and_the_modes_that_take_the_two_calls(oracle):
    """Backward = blur adjoint + the warp's own backward: the same gradients as the two calls.  Modes the fused launch does not cover fall
    through to the two calls (same results, no error)."""
    import kornia_amd as Km

    T = Km.geometry.transform
    B, C, H, W, h, w = 2, 3, 100, 140, 90, 150
    g = torch.Generator().manual_seed(8)
    x = torch.rand(B, C, H, W, generator=g)
    M = flagship_homographies(B, H, W, h, w, g, jitter=5.0)
    go = torch.rand(B, C, h, w, generator=g) - 0.4

    def grads(fn):
        xg, Mg = x.cuda().requires_grad_(), M.cuda().requires_grad_()
        fn(xg, Mg).backward(go.cuda())
        return xg.grad.cpu(), Mg.grad.cpu()

    gx, gM = grads(lambda a, m: T.warp_perspective_blur(a, m, (h, w), (5, 5), (1.5, 1.5)))
    gx2, gM2 = grads(lambda a, m: Km.filters.gaussian_blur2d(T.warp_perspective(a, m, (h, w)), (5, 5), (1.5, 1.5)))
    assert torch.equal(gx, gx2) and torch.equal(gM, gM2)  # the very same two launches
    gw = oracle.gaussian_blur2d_backward(go, oracle.warp_perspective(x, M, (h, w)), (5, 5), (1.5, 1.5))
    gxo, gMo = oracle.warp_perspective_backward(gw, x, M, (h, w))
    assert torch.allclose(gx, gxo, atol=1e-5, rtol=0)
    assert ((gM.double() - gMo.double()).abs().amax(dim=(-2, -1)) / gMo.double().abs().amax(dim=(-2, -1))).max().item() <= 5e-5
    # not fusable: bicubic, border padding, a 9 x 9 kernel, RGBA, a rectangular kernel
    xc, Mc = x.cuda(), M.cuda()
    for kw in (dict(mode="bicubic"), dict(padding_mode="border")):
        y = T.warp_perspective_blur(xc, Mc, (h, w), (5, 5), (1.5, 1.5), **kw)
        assert torch.equal(y, Km.filters.gaussian_blur2d(T.warp_perspective(xc, Mc, (h, w), **kw), (5, 5), (1.5, 1.5)))
    y = T.warp_perspective_blur(xc, Mc, (h, w), (9, 9), (2.0, 2.0))
    assert torch.equal(y, Km.filters.gaussian_blur2d(T.warp_perspective(xc, Mc, (h, w)), (9, 9), (2.0, 2.0)))
    y = T.warp_perspective_blur(xc, Mc, (h, w), (3, 5), (1.0, 1.0))
    assert torch.equal(y, Km.filters.gaussian_blur2d(T.warp_perspective(xc, Mc, (h, w)), (3, 5), (1.0, 1.0)))
    x4 = torch.rand(B, 4, H, W, generator=g).cuda()
    y = T.warp_perspective_blur(x4, Mc, (h, w), (5, 5), (1.5, 1.5))
    assert torch.equal(y, Km.filters.gaussian_blur2d(T.warp_perspective(x4, Mc, (h, w)), (5, 5), (1.5, 1.5)))


def test_full_size_flagship_forward():
    """Config-2 spatial size: 8 x 3 x 512 x 512, flagship homographies - bit-identical to the two calls."""
    import kornia_amd as Km

    T = Km.geometry.transform
    g = torch.Generator().manual_seed(0)
    B = 8 if Km._native.lib().km_device_info(None, 0) > 0 else 1
    x = torch.rand(B, 3, 512, 512, generator=g).cuda()
    M = flagship_homographies(B, 512, 512, 512, 512, g).cuda()
    y = T.warp_perspective_blur(x, M, (512, 512), (5, 5), (1.5, 1.5))
    assert torch.equal(y, Km.filters.gaussian_blur2d(T.warp_perspective(x, M, (512, 512)), (5, 5), (1.5, 1.5)))
