"""GPU: the fused masked warp loss (km_warp_masked_loss) and ImageRegistrator on the native kernels, against the oracle's
composition of restated warps and against fixtures produced by the real reference (tests/golden/registration.npz).
tests/test_emulated_kernels.py runs the same cases on the host build of the kernels.  The file sorts after the hot-path tests: the
loss kernel was restructured (two rows of loads in flight, RGB unrolled) after its last device run (profiles/r01_registration_pyramid_gpu_tests.log
is the run of the first version), so its current form is verified through the host build only."""
import pytest
import torch
import torch.nn.functional as F

from _util import golden as _golden_np

pytestmark = pytest.mark.gpu


def golden(name):
    return {k: torch.from_numpy(v) for k, v in _golden_np(name).items()}


def T():
    import kornia_amd as K

    return K.geometry.transform


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("kind,fn", [("l1", F.l1_loss), ("mse", F.mse_loss)])
def test_single_level_loss_vs_reference(kind, fn):
    d = golden("registration")
    reg = T().ImageRegistrator("homography", loss_fn=fn)
    src, dst = d["src"].cuda(), d["dst"].cuda()
    for k in range(4):
        H = d["H"][k : k + 1].cuda().requires_grad_(True)
        loss = reg.get_single_level_loss(src, dst, H)
        loss.backward()
        assert abs(loss.item() - d[f"{kind}_loss_{k}"].item()) < 2e-6
        # the gradient of a bilinear sampler jumps at integer positions: same bound as the warps
        assert _rel(H.grad.cpu(), d[f"{kind}_grad_{k}"]) < 2e-4, (k, H.grad.cpu(), d[f"{kind}_grad_{k}"])


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("kind", ["l1", "mse"])
@pytest.mark.parametrize("shape", [(1, 3, 40, 56), (3, 1, 33, 70), (2, 4, 64, 130)])
def test_fused_loss_vs_oracle(oracle, shape, kind, align):
    g = torch.Generator().manual_seed(shape[-1])
    B = shape[0]
    src, dst = torch.rand(*shape, generator=g), torch.rand(*shape, generator=g)
    H = torch.eye(3)[None].repeat(B, 1, 1) + 0.08 * (torch.rand(B, 3, 3, generator=g) - 0.5)
    for Hm in (H, H[:1]):
        ref, gref = oracle.masked_warp_loss(src, dst, Hm, kind, align)
        Hg = Hm.cuda().requires_grad_(True)
        loss = T().masked_warp_loss(src.cuda(), dst.cuda(), Hg, kind, align)
        loss.backward()
        assert abs(loss.item() - ref.item()) < 1e-6
        assert _rel(Hg.grad.cpu(), gref) < 5e-5


def test_fused_equals_composition_and_edge_cases():
    t = T()
    g = torch.Generator().manual_seed(1)
    src, dst = torch.rand(2, 3, 24, 40, generator=g).cuda(), torch.rand(2, 3, 24, 40, generator=g).cuda()
    H = (torch.eye(3)[None].repeat(2, 1, 1) + 0.05 * (torch.rand(2, 3, 3, generator=g) - 0.5)).cuda()
    # the composition the reference runs, on the native warps (a loss function the fused path does not know)
    reg = t.ImageRegistrator("homography", loss_fn=lambda a, b, reduction: F.l1_loss(a, b, reduction=reduction))
    Hc = H.clone().requires_grad_(True)
    comp = reg.get_single_level_loss(src, dst, Hc)
    comp.backward()
    Hf = H.clone().requires_grad_(True)
    fused = t.ImageRegistrator("homography").get_single_level_loss(src, dst, Hf)
    fused.backward()
    assert abs(fused.item() - comp.item()) < 1e-6 and _rel(Hf.grad, Hc.grad) < 2e-4
    # nothing selected: the mean of an empty selection is nan in the reference too
    far = torch.tensor([[[1.0, 0.0, 5.0], [0.0, 1.0, 5.0], [0.0, 0.0, 1.0]]]).cuda()
    assert torch.isnan(t.masked_warp_loss(src[:1], dst[:1], far))
    # half precision: the warped value and the mask go through the storage dtype like the unfused pipeline
    for dt in (torch.bfloat16, torch.float16):
        lo = t.masked_warp_loss(src.to(dt), dst.to(dt), H, "l1")
        assert lo.dtype == dt and abs(lo.float().item() - fused.item()) < 2e-2
    with pytest.raises(ValueError):
        t.masked_warp_loss(src, dst, H, "huber")
    with pytest.raises(ValueError):
        t.masked_warp_loss(src, dst[:, :2], H)
    with pytest.raises(RuntimeError):
        t.masked_warp_loss(src.clone().requires_grad_(True), dst, H)


def test_unmasked_loss_is_config5_step():
    """threshold=None: F.l1_loss / F.mse_loss of the plain homography_warp (BASELINE config 5: learned H, grad wrt H)."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(5)
    x, target = torch.rand(4, 3, 32, 48, generator=g).cuda(), torch.rand(4, 3, 32, 48, generator=g).cuda()
    H = (torch.eye(3)[None] + 0.05 * torch.randn(4, 3, 3, generator=g)).cuda()
    for kind, fn in (("l1", F.l1_loss), ("mse", F.mse_loss)):
        Hc = H.clone().requires_grad_(True)
        ref = fn(K.homography_warp(x, Hc, (32, 48)), target)
        ref.backward()
        Hf = H.clone().requires_grad_(True)
        out = T().masked_warp_loss(x, target, Hf, kind, threshold=None)
        out.backward()
        assert abs(out.item() - ref.item()) < 1e-6 and _rel(Hf.grad, Hc.grad) < 2e-4


def test_models_match_reference_tests():
    """tests/geometry/transform/test_image_registrator.py:34-74 (the Similarity / Homography smoke and scale cases)."""
    t = T()
    eye = torch.eye(3)[None]
    for r in (True, False):
        for sc in (True, False):
            for sh in (True, False):
                s = t.Similarity(r, sc, sh)
                assert torch.allclose(s(), eye, atol=1e-4) and torch.allclose(s.forward_inverse(), eye, atol=1e-4) and repr(s)
    sim = t.Similarity(True, True, True)
    sim.scale.data *= 0.5
    assert torch.allclose(sim(), torch.tensor([[[0.5, 0, 0], [0, 0.5, 0], [0, 0, 1.0]]]), atol=1e-4)
    assert torch.allclose(sim.forward_inverse(), torch.tensor([[[2.0, 0, 0], [0, 2, 0], [0, 0, 1.0]]]), atol=1e-4)
    h = t.Homography()
    assert torch.allclose(h(), eye, atol=1e-4) and torch.allclose(h.forward_inverse(), eye, atol=1e-4) and repr(h)
    for name in ("homography", "similarity", "translation", "scale", "rotation"):
        assert t.ImageRegistrator(name) is not None
    with pytest.raises(ValueError):
        t.ImageRegistrator("affine3d")
    with pytest.raises(ValueError):
        t.ImageRegistrator(t.Homography())


def test_registration_toy():
    """tests/geometry/transform/test_image_registrator.py:87-99, and the reference's own result on the same data."""
    d = golden("registration")
    t = T()
    IR = t.ImageRegistrator("Similarity", num_iterations=500, lr=3e-4, pyramid_levels=2).cuda()
    model, inter = IR.register(d["toy_src"].cuda(), d["toy_dst"].cuda(), output_intermediate_models=True)
    assert len(inter) == 2
    assert torch.allclose(model.detach().cpu(), d["toy_H"], atol=1e-3, rtol=1e-3)
    assert torch.allclose(model.detach().cpu(), d["toy_model"], atol=5e-4, rtol=0)
    out = IR.warp_src_into_dst(d["toy_src"].cuda())
    assert out.shape == d["toy_src"].shape and IR.warp_dst_inro_src(d["toy_dst"].cuda()).shape == d["toy_dst"].shape
    with pytest.raises(ValueError):
        t.ImageRegistrator("similarity").register(torch.rand(1, 1, 16, 16).cuda(), torch.rand(1, 1, 8, 8).cuda())
    reg = t.ImageRegistrator("similarity", allow_shape_mismatch=True, num_iterations=2, pyramid_levels=2).cuda()
    assert reg.register(torch.rand(1, 1, 20, 20).cuda(), torch.rand(1, 1, 16, 16).cuda()) is not None


def test_fused_loss_linearity_at_full_size():
    """config 5 size (B=64 of the 512 x 3 x 256 x 256): the count is an integer, identity on equal images gives 0 loss, the loss of
    a per-sample batch is the count-weighted mean of the per-sample losses."""
    t = T()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(64, 3, 256, 256, device="cuda", generator=g)
    y = torch.rand(64, 3, 256, 256, device="cuda", generator=g)
    eye = torch.eye(3, device="cuda")[None]
    assert t.masked_warp_loss(x, x, eye, align_corners=True).item() < 1e-6
    H = eye + 0.05 * (torch.rand(64, 3, 3, device="cuda", generator=g) - 0.5)
    whole = t.masked_warp_loss(x, y, H)
    parts = torch.stack([t.masked_warp_loss(x[i : i + 1], y[i : i + 1], H[i : i + 1]) for i in range(0, 64, 16)])
    assert abs(whole.item() - 0.33) < 0.05 and (parts - whole).abs().max().item() < 0.02


def test_tail_of_the_fused_loss_scales_sums_and_casts_like_autograd(oracle):
    """km_warp_masked_loss_finish / km_scale_f64 (the two launches that replace the torch ops behind the accumulators): an upstream gradient other than one,
    a float64 model, more images than the finishing workgroup has threads (its strided sums), a shared matrix at that batch."""
    t = T()
    g = torch.Generator().manual_seed(11)
    B = 300
    src, dst = torch.rand(B, 1, 12, 16, generator=g), torch.rand(B, 1, 12, 16, generator=g)
    H = torch.eye(3)[None].repeat(B, 1, 1) + 0.05 * (torch.rand(B, 3, 3, generator=g) - 0.5)
    for Hm in (H, H[:1]):
        ref, gref = oracle.masked_warp_loss(src, dst, Hm, "l1", False)
        for dt in (torch.float32, torch.float64):
            Hg = Hm.to(dt).cuda().requires_grad_(True)
            loss = t.masked_warp_loss(src.cuda(), dst.cuda(), Hg, "l1", False)
            (2.5 * loss).backward()
            assert Hg.grad.dtype == dt and Hg.grad.shape == Hm.shape
            assert abs(loss.item() - ref.item()) < 1e-6
            assert _rel(Hg.grad.cpu().float(), 2.5 * gref) < 5e-5
    # an upstream gradient in another dtype than float32 / float64 takes the torch ops: same numbers
    Hg = H[:4].cuda().requires_grad_(True)
    lo = t.masked_warp_loss(src[:4].cuda().half(), dst[:4].cuda().half(), Hg, "l1")
    lo.backward()
    Hf = H[:4].cuda().requires_grad_(True)
    t.masked_warp_loss(src[:4].cuda().half().float(), dst[:4].cuda().half().float(), Hf, "l1").backward()
    assert lo.dtype == torch.float16 and _rel(Hg.grad, Hf.grad) < 2e-2
    # the loss is a tensor of its own, not a view of a buffer of the launch: callers add to it in place (the reference's registrator tests do)
    Hg = H[:4].cuda().requires_grad_(True)
    loss = t.masked_warp_loss(src[:4].cuda(), dst[:4].cuda(), Hg, "l1")
    assert not loss._is_view() and loss.dim() == 0
    loss += 1.0
    loss.backward()
    assert torch.isfinite(Hg.grad).all()
