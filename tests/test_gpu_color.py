"""GPU parity of the fused ColorJitter kernel (km_color_jitter_fwd) against fixtures from the real reference
(tests/golden/color_jitter.npz) and against oracle/color_ref.py on larger random cases."""
import itertools
import os
import sys

import pytest
import torch

from _util import golden

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def _t(d, k):
    return torch.from_numpy(d[k])


# fp32: elementwise chains agree to rounding; a pixel may sit on a clamp / sextant boundary, hence 5e-6 absolute
TOL = dict(atol=5e-6, rtol=0)


def test_single_stages_match_reference_fixture():
    import kornia_amd as K

    d = golden("color_jitter")
    x = _t(d, "x").cuda()
    E = K.enhance
    assert torch.allclose(E.adjust_brightness_accumulative(x, _t(d, "bf").cuda()).cpu(), _t(d, "brightness"), **TOL)
    assert torch.allclose(E.adjust_contrast_with_mean_subtraction(x, _t(d, "cf").cuda()).cpu(), _t(d, "contrast"), **TOL)
    assert torch.allclose(E.adjust_saturation_with_gray_subtraction(x, _t(d, "sf").cuda()).cpu(), _t(d, "saturation"), **TOL)
    assert torch.allclose(E.adjust_hue(x, (_t(d, "hf") * 2 * 3.141592653589793).cuda()).cpu(), _t(d, "hue"), **TOL)


@pytest.mark.parametrize("order", ["0123", "3210", "2031", "1302"])
def test_fused_sequence_matches_reference_fixture(order):
    import kornia_amd as K

    d = golden("color_jitter")
    out = K.enhance.color_jitter(_t(d, "x").cuda(), _t(d, "bf").cuda(), _t(d, "cf").cuda(), _t(d, "sf").cuda(), _t(d, "hf").cuda(),
                                 [int(c) for c in order])
    assert torch.allclose(out.cpu(), _t(d, "seq_" + order), atol=1e-5, rtol=0), (out.cpu() - _t(d, "seq_" + order)).abs().max()
    if order == "2031":
        assert torch.allclose(out.cpu(), _t(d, "module_2031"), atol=1e-5, rtol=0)


@pytest.mark.parametrize("shape", [(3, 3, 37, 53), (2, 3, 224, 224), (1, 3, 5, 7)])
def test_all_orders_vs_restatement(shape):
    import color_ref
    import kornia_amd as K

    g = torch.Generator().manual_seed(9)
    x = torch.rand(*shape, generator=g)
    B = shape[0]
    bf, cf, sf = (0.5 + torch.rand(B, generator=g) for _ in range(3))
    hf = (torch.rand(B, generator=g) - 0.5)
    for order in itertools.islice(itertools.permutations(range(4)), 0, 24, 5):
        ref = color_ref.color_jitter(x, bf, cf, sf, hf, order)
        out = K.enhance.color_jitter(x.cuda(), bf.cuda(), cf.cuda(), sf.cuda(), hf.cuda(), list(order))
        assert torch.allclose(out.cpu(), ref, atol=2e-5, rtol=0), (order, (out.cpu() - ref).abs().max())
    # subset of stages, float factors, (3,H,W) input
    ref = color_ref.adjust_saturation_with_gray_subtraction(color_ref.adjust_brightness_accumulative(x[0], 1.3), 0.4)
    out = K.enhance.color_jitter(x[0].cuda(), brightness_factor=1.3, saturation_factor=0.4)
    assert out.shape == x[0].shape and torch.allclose(out.cpu(), ref, **TOL)


def test_half_precision_and_errors():
    import color_ref
    import kornia_amd as K

    g = torch.Generator().manual_seed(2)
    x = torch.rand(4, 3, 64, 64, generator=g)
    f = [0.7 + 0.6 * torch.rand(4, generator=g) for _ in range(3)] + [(torch.rand(4, generator=g) - 0.5) * 0.2]
    ref = color_ref.color_jitter(x, *f, [0, 1, 2, 3])
    for dt in (torch.bfloat16, torch.float16):
        out = K.enhance.color_jitter(x.to(dt).cuda(), *[t.cuda() for t in f], [0, 1, 2, 3])
        assert out.dtype == dt and torch.allclose(out.float().cpu(), color_ref.color_jitter(x.to(dt).float(), *f, [0, 1, 2, 3]), atol=1e-2)
    assert torch.allclose(K.enhance.color_jitter(x.cuda(), *[t.cuda() for t in f]).cpu(), ref, atol=2e-5)
    with pytest.raises(ValueError):
        K.enhance.adjust_hue(torch.rand(2, 1, 8, 8, device="cuda"), 0.1)
    with pytest.raises(ValueError):
        K.enhance.color_jitter(x.cuda(), 1.0, 1.0, 1.0, 0.0, [0, 1, 1, 3])
    with pytest.raises(Exception):
        K.enhance.adjust_brightness_accumulative(x, 1.1)  # CPU tensor: no fallback


def test_identity_factors_at_full_size():
    """Neutral factors reproduce the image (brightness / contrast / saturation exactly, hue through HSV to rounding)."""
    import kornia_amd as K

    x = torch.rand(64, 3, 224, 224, device="cuda")
    assert torch.equal(K.enhance.color_jitter(x, 1.0, 1.0, 1.0, None), x)
    assert torch.allclose(K.enhance.adjust_hue(x, 0.0), x, atol=5e-6)


def test_stage_enable_flags_skip_on_device():
    """enable[k] == 0 skips every stage of kind k (the module's `(factor != neutral).any()` guards)."""
    import color_ref
    import kornia_amd as K

    g = torch.Generator().manual_seed(6)
    x = torch.rand(3, 3, 40, 44, generator=g)
    bf, cf, sf = (0.5 + torch.rand(3, generator=g) for _ in range(3))
    hf = (torch.rand(3, generator=g) - 0.5) * 0.3
    en = torch.tensor([1, 0, 1, 0], device="cuda")
    out = K.enhance.color_jitter(x.cuda(), bf.cuda(), cf.cuda(), sf.cuda(), hf.cuda(), [3, 1, 0, 2], enable=en)
    ref = color_ref.adjust_saturation_with_gray_subtraction(color_ref.adjust_brightness_accumulative(x, bf), sf)
    assert torch.allclose(out.cpu(), ref, atol=5e-6, rtol=0)
    # all-zero brightness factors with the guard off: the stage is skipped instead of blacking the image out
    en = torch.stack([(torch.zeros(3) != 0).any(), torch.tensor(True), torch.tensor(True), torch.tensor(True)]).cuda()
    out = K.enhance.color_jitter(x.cuda(), torch.zeros(3).cuda(), None, None, None, [0], enable=en)
    assert torch.equal(out.cpu(), x)


@pytest.mark.parametrize("order", ["0", "1", "2", "3", "0123", "3210", "2031", "13"])
def test_backward_matches_autograd_through_the_reference_ops(order):
    """km_color_jitter_bwd against autograd through oracle/color_ref.py (the reference's op sequence, pinned to its fixtures):
    gradient wrt the image and wrt the four per-image factor tensors.  Factors strong enough that every clamp is active somewhere;
    grey (r = g = b: amax / amin ties), black and saturated pixels included.  fp32: 2e-5 relative to the largest entry - the forward
    values agree to rounding, a pixel whose pre-clamp value or hue sextant sits within rounding of a boundary may take the other side
    in one of the two implementations, so up to 1e-4 of the pixels are allowed to differ."""
    import color_ref
    import kornia_amd as K

    g = torch.Generator().manual_seed(11 + len(order))
    B, H, W = 3, 40, 52
    x = torch.rand(B, 3, H, W, generator=g)
    x[:, :, :3, :] = x[:, :1, :3, :]          # grey rows
    x[:, :, 3, :8] = 0.0                      # black
    x[:, :, 4, :8] = 1.0                      # white
    x[:, 0, 5, :8] = 1.0                      # saturated red
    f = [0.5 + 1.0 * torch.rand(B, generator=g) for _ in range(3)] + [(torch.rand(B, generator=g) - 0.5) * 0.6]
    w = torch.randn(B, 3, H, W, generator=g)
    stages = [int(c) for c in order]
    xr = x.clone().requires_grad_()
    fr = [t.clone().requires_grad_() for t in f]
    (color_ref.color_jitter(xr, *fr, stages) * w).sum().backward()
    xd = x.cuda().requires_grad_()
    fd = [t.cuda().requires_grad_() for t in f]
    given = [fd[k] if k in stages else None for k in range(4)]
    out = K.enhance.color_jitter(xd, *given, stages)
    assert torch.allclose(out.detach().cpu(), color_ref.color_jitter(x, *f, stages), atol=2e-5)
    (out * w.cuda()).sum().backward()
    gx, gr = xd.grad.cpu(), xr.grad
    bad = ((gx - gr).abs() > 2e-5 * gr.abs().max()).float().mean().item()
    assert bad <= 1e-4, f"{bad:.2e} of the image-gradient entries differ"
    for k in stages:
        a, r = fd[k].grad.cpu(), fr[k].grad
        assert torch.allclose(a, r, rtol=2e-3, atol=2e-3 * r.abs().max().item()), (k, a, r)
    # the per-sample switch: a sample that is not jittered passes its gradient through
    xd2 = x.cuda().requires_grad_()
    mask = torch.tensor([True, False, True]).cuda()
    out2 = K.enhance.color_jitter(xd2, *[t.cuda() if k in stages else None for k, t in enumerate(f)], stages, apply=mask)
    (out2 * w.cuda()).sum().backward()
    assert torch.equal(xd2.grad[1].cpu(), w[1])
    assert torch.allclose(xd2.grad[0].cpu(), gx[0], atol=1e-6) and torch.allclose(xd2.grad[2].cpu(), gx[2], atol=1e-6)


def test_backward_half_precision_and_odd_sizes():
    """bf16 / fp16 storage and a plane size that is not a multiple of four (scalar path) against the fp32 kernel on the same inputs."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(5)
    for (H, W) in ((16, 24), (7, 9)):
        x = torch.rand(2, 3, H, W, generator=g)
        w = torch.randn(2, 3, H, W, generator=g)
        f = [0.8 + 0.4 * torch.rand(2, generator=g) for _ in range(3)] + [(torch.rand(2, generator=g) - 0.5) * 0.2]
        ref = None
        for dt, tol in ((torch.float32, 0.0), (torch.bfloat16, 6e-2), (torch.float16, 1e-2)):
            xq = x.to(dt).cuda().requires_grad_()
            out = K.enhance.color_jitter(xq, *[t.cuda() for t in f], [0, 2, 3, 1])
            (out.float() * w.to(dt).float().cuda()).sum().backward()
            if ref is None:
                ref = xq.grad.float().cpu()
            else:
                assert xq.grad.dtype == dt
                assert ((xq.grad.float().cpu() - ref).abs() > tol * ref.abs().max()).float().mean().item() < 0.02


def test_parameter_table_kernel_matches_the_torch_assembly():
    """km_color_params_fwd (factor table with hue in radians, the module's `(factor != neutral).any()` stage switches, batch_prob > 0.5)
    against the same quantities assembled with torch ops, and the augmentation entry built on it against the generic one."""
    import math
    import kornia_amd.augmentation as A
    from kornia_amd.enhance.adjust import color_jitter

    g = torch.Generator().manual_seed(9)
    B = 300  # more samples than one pass of the block
    x = torch.rand(B, 3, 8, 12, generator=g).cuda()
    for neutral in ((), (0,), (1, 3), (0, 1, 2, 3)):
        P = {"brightness_factor": 0.8 + 0.4 * torch.rand(B, generator=g), "contrast_factor": 0.8 + 0.4 * torch.rand(B, generator=g),
             "saturation_factor": 0.8 + 0.4 * torch.rand(B, generator=g), "hue_factor": (torch.rand(B, generator=g) - 0.5) * 0.2,
             "order": torch.tensor([2, 0, 3, 1]), "batch_prob": torch.rand(B, generator=g)}
        for k, key in enumerate(("brightness_factor", "contrast_factor", "saturation_factor", "hue_factor")):
            if k in neutral:
                P[key] = torch.full((B,), (0.0, 1.0, 1.0, 0.0)[k])
        dP = {k: (v.cuda() if k != "order" else v) for k, v in P.items()}
        got = A.color_jitter(x, dP)
        en = torch.stack([(P["brightness_factor"] != 0).any(), (P["contrast_factor"] != 1).any(), (P["saturation_factor"] != 1).any(), (P["hue_factor"] != 0).any()])
        ref = color_jitter(x, dP["brightness_factor"], dP["contrast_factor"], dP["saturation_factor"], dP["hue_factor"], [2, 0, 3, 1], enable=en.cuda(),
                           apply=(dP["batch_prob"] > 0.5))
        assert torch.equal(got, ref), neutral
