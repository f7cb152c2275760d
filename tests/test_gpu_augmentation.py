"""GPU: the augmentation layer's geometric entry functions (kornia_amd.augmentation.random_affine / random_perspective) replaying the
parameters the reference's RandomAffine / RandomPerspective sampled with p < 1 (tests/golden/geometric_aug.npz, oracle/make_golden.py):
matrix from the parameters, warp, and the per-sample probability switch inside the warp's launch.  Also runs on the host build of
the kernels (tests/test_emulated_kernels.py)."""
import pytest
import torch

from _util import golden

pytestmark = pytest.mark.gpu


def _params(d, name):
    pre = name + "__"
    return {k[len(pre):]: torch.from_numpy(v) for k, v in d.items() if k.startswith(pre) and k not in (pre + "out", pre + "matrix")}


@pytest.mark.parametrize("name,kw", [("perspective", {}), ("perspective_nearest_align", dict(resample="nearest", align_corners=True))])
def test_random_perspective_replays_the_reference(name, kw):
    import kornia_amd.augmentation as A

    d = golden("geometric_aug")
    x, ref, p = torch.from_numpy(d["x"]), torch.from_numpy(d[name + "__out"]), _params(d, name)
    applied = p["batch_prob"] > 0.5
    assert 0 < int(applied.sum()) < x.shape[0]
    out = A.random_perspective(x.cuda(), {k: v.cuda() for k, v in p.items()}, **kw).cpu()
    assert torch.equal(out[~applied], x[~applied])  # the switch inside the warp's launch: untouched samples are copies
    if kw.get("resample") == "nearest":
        # a sampling position within rounding of a pixel boundary may pick the neighbour (the matrix comes from the parameters on both sides)
        assert ((out - ref).abs() > 1e-5).float().mean().item() < 2e-3
    else:
        # (the homography is rebuilt from the sampled corner points on both sides: a last-bit difference of the matrix, times the slope of
        # a noise image under a 0.4 distortion - SURVEY.md A.4 - is what the 2e-5 holds; with the reference's matrix the warp is bit-exact)
        assert torch.allclose(out, ref, atol=2e-5, rtol=0), (out - ref).abs().max()
    # the same through the autograd-capable composition (no switch in the launch: select pass)
    xg = x.cuda().requires_grad_()
    out2 = A.random_perspective(xg, {k: v.cuda() for k, v in p.items()}, **kw)
    assert torch.equal(out2.detach().cpu(), out)
    out2.sum().backward()
    assert torch.equal(xg.grad[~applied.cuda()].cpu(), torch.ones_like(x[~applied]))


@pytest.mark.parametrize("name,kw", [("affine", {}), ("affine_border", dict(padding_mode="border", align_corners=True))])
def test_random_affine_replays_the_reference(name, kw):
    import kornia_amd.augmentation as A

    d = golden("geometric_aug")
    x, ref, p = torch.from_numpy(d["x"]), torch.from_numpy(d[name + "__out"]), _params(d, name)
    applied = p["batch_prob"] > 0.5
    assert 0 < int(applied.sum()) < x.shape[0]
    out = A.random_affine(x.cuda(), {k: v.cuda() for k, v in p.items()}, **kw).cpu()
    assert torch.equal(out[~applied], x[~applied])
    assert torch.allclose(out, ref, atol=1e-5, rtol=0), (out - ref).abs().max()
    # the module's transform_matrix from the same parameters, in the same launch as the chain
    m, M, ap = A.affine_chain({k: v.cuda() for k, v in p.items()}, "cuda", x.shape[-2], x.shape[-1], with_matrix=True)
    Mref = torch.from_numpy(d[name + "__matrix"])
    assert torch.allclose(M.cpu()[applied], Mref[applied], atol=1e-5, rtol=1e-5)
    assert torch.equal(ap.cpu().bool(), applied)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gaussian_taps_dtype_entry_point_is_the_cast_round_trip(dtype):
    """km_gaussian_taps_dtype_fwd (one sigma per sample, the probability draw thresholded in the launch, taps rounded to the image dtype in the
    launch) against km_gaussian_taps_fwd + the torch ops it replaces: sigma.unsqueeze(-1).expand(-1, 2).contiguous(), batch_prob > 0.5, and
    filter2d's taps.to(image dtype) (kornia/filters/filter.py:126) widened back to float32 - bit for bit, and the blur that uses them."""
    import kornia_amd.augmentation as A
    from kornia_amd.filters.filter import filter2d_separable, filter2d_separable_taps

    g = torch.Generator().manual_seed(11)
    B = 37
    sigma = (0.1 + 1.9 * torch.rand(B, generator=g)).cuda()
    sigma[3] = 0.0  # (the sigma -> 0 limit: identity kernel)
    prob = (torch.rand(B, generator=g) < 0.6).float().cuda()
    for ks in ((5, 5), (3, 7)):
        ox, oy = A.gaussian_taps(sigma.unsqueeze(-1).expand(-1, 2), ks, apply=prob > 0.5)
        nx, ny = A.gaussian_taps(sigma, ks, batch_prob=prob, round_to=dtype)
        assert torch.equal(nx, ox.to(dtype).float()) and torch.equal(ny, oy.to(dtype).float())
        ox, oy = A.gaussian_taps(sigma.unsqueeze(-1).expand(-1, 2), ks)
        nx, ny = A.gaussian_taps(sigma, ks, round_to=dtype)
        assert torch.equal(nx, ox.to(dtype).float()) and torch.equal(ny, oy.to(dtype).float())
        s2 = torch.stack([sigma, sigma.flip(0)], dim=1)  # (sigma_y, sigma_x) per sample
        o2x, o2y = A.gaussian_taps(s2, ks)
        n2x, n2y = A.gaussian_taps(s2, ks, round_to=dtype)
        assert torch.equal(n2x, o2x.to(dtype).float()) and torch.equal(n2y, o2y.to(dtype).float())
    x = torch.rand(B, 3, 24, 32, generator=g).to(dtype).cuda()
    ox, oy = A.gaussian_taps(sigma.unsqueeze(-1).expand(-1, 2), (5, 5))
    nx, ny = A.gaussian_taps(sigma, (5, 5), round_to=dtype)
    assert torch.equal(filter2d_separable_taps(x, nx, ny, "reflect"), filter2d_separable(x, ox, oy, "reflect"))
    # the entry function: the draw rides in the taps (finite image: untouched samples come back bit for bit)
    out = A.random_gaussian_blur(x, {"sigma": sigma, "batch_prob": prob})
    ref = torch.where((prob > 0.5).view(-1, 1, 1, 1), filter2d_separable(x, ox, oy, "reflect"), x)
    assert torch.equal(out, ref)
    with pytest.raises(ValueError):
        A.gaussian_taps(sigma, (5, 5), apply=prob > 0.5, batch_prob=prob)
