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
