"""GPU: BASELINE.json configs 2-5 at their full spatial sizes through size-independent properties
(the oracle comparisons at sizes it finishes in seconds live in the other test_gpu_* files)."""
import pytest
import torch

from _util import flagship_homographies

pytestmark = pytest.mark.gpu


def test_config2_full_size_step_is_consistent():
    """256x3x512x512 fwd+bwd: finite, deterministic grad wrt the image (fixed-point accumulation is
    order-independent), and a sub-batch reproduces the corresponding slice bit for bit."""
    import kornia_amd as K

    gen = torch.Generator().manual_seed(0)
    M = flagship_homographies(256, 512, 512, 512, 512, gen).cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand(256, 3, 512, 512, device="cuda", generator=g)
    go = torch.rand(256, 3, 512, 512, device="cuda", generator=g)

    def step(xs, Ms, gs):
        xs = xs.detach().requires_grad_()
        Ms = Ms.detach().requires_grad_()
        y = K.gaussian_blur2d(K.warp_perspective(xs, Ms, (512, 512)), (5, 5), (1.5, 1.5))
        y.backward(gs)
        return y.detach(), xs.grad, Ms.grad

    y, gx, gM = step(x, M, go)
    assert torch.isfinite(y).all() and torch.isfinite(gx).all() and torch.isfinite(gM).all()
    y2, gx2, gM2 = step(x, M, go)
    assert torch.equal(y, y2) and torch.equal(gx, gx2)  # bit-reproducible
    def close(a, b):  # relative to the largest entry of each matrix (the entries of one matrix span five orders of magnitude)
        return ((a - b).abs().amax(dim=(1, 2)) <= 1e-6 * b.abs().amax(dim=(1, 2))).all()

    assert close(gM, gM2)    # fp64 atomics: order may differ in the last bits
    ys, gxs, gMs = step(x[5:9], M[5:9], go[5:9])
    assert torch.equal(ys, y[5:9]) and torch.equal(gxs, gx[5:9])
    # a sub-batch is dealt to the workgroups of the one-read backward in other runs of tiles: the fp32 partial sums a thread keeps over
    # the tiles of a run group differently (fp64 beyond the thread)
    assert close(gMs, gM[5:9])


def test_config4_sobel_and_bicubic_affine_1080p():
    import kornia_amd as K

    H, W = 1080, 1920
    ramp = torch.arange(W, device="cuda", dtype=torch.float32).view(1, 1, 1, W).expand(2, 1, H, W).contiguous()
    gr = K.SpatialGradient("sobel", 1, True)(ramp)
    assert gr.shape == (2, 1, 2, H, W)
    assert torch.allclose(gr[:, :, 0, :, 1:-1], torch.ones_like(gr[:, :, 0, :, 1:-1]), atol=1e-4)  # d/dx of x == 1
    assert gr[:, :, 1].abs().max().item() < 1e-4                                                    # d/dy of x == 0
    x = torch.rand(2, 1, H, W, device="cuda")
    eye = torch.eye(2, 3, device="cuda")[None].repeat(2, 1, 1)
    # fp32 sampling positions are exact only to ~3e-5 px at x ~ 1900; bicubic of a noise image moves ~1/px
    assert torch.allclose(K.warp_affine(x, eye, (H, W), mode="bicubic"), x, atol=5e-4)               # identity
    shift = eye.clone()
    shift[:, 0, 2], shift[:, 1, 2] = 3.0, -2.0                                                       # dst = src shifted (+3, -2) px
    y = K.warp_affine(x, shift, (H, W), mode="bicubic")
    assert torch.allclose(y[..., : H - 2, 3:], x[..., 2:, : W - 3], atol=5e-4)
    a = torch.rand(2, 1, H, W, device="cuda")
    lin = K.spatial_gradient(x + 3 * a) - (K.spatial_gradient(x) + 3 * K.spatial_gradient(a))
    assert lin.abs().max().item() < 1e-5


def test_config5_learned_homography_step():
    """homography_warp (normalised H, align_corners=False default) with grad wrt H on 64x3x256x256:
    identity H with align_corners=True reproduces the input; one Adam-free descent step lowers the loss."""
    import kornia_amd as K

    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(64, 3, 256, 256, device="cuda", generator=g)
    eye = torch.eye(3, device="cuda")[None].repeat(64, 1, 1)
    assert torch.allclose(K.homography_warp(x, eye, (256, 256), align_corners=True), x, atol=1e-5)
    smooth = K.gaussian_blur2d(x, (9, 9), (3.0, 3.0))
    H_true = eye + 0.01 * torch.randn(64, 3, 3, device="cuda", generator=g)
    target = K.homography_warp(smooth, H_true, (256, 256))
    H = eye.clone().requires_grad_()
    loss0 = torch.nn.functional.l1_loss(K.homography_warp(smooth, H, (256, 256)), target)
    loss0.backward()
    assert torch.isfinite(H.grad).all() and H.grad.abs().max() > 0
    with torch.no_grad():
        H1 = H - 1e-3 * H.grad / H.grad.abs().amax(dim=(1, 2), keepdim=True)
    loss1 = torch.nn.functional.l1_loss(K.homography_warp(smooth, H1, (256, 256)), target)
    assert loss1.item() < loss0.item()


def test_config3_bf16_ops_224():
    """bf16 legs of config 3 (the ops RandomAffine / RandomGaussianBlur call): bf16 result == fp32 result of
    the same bf16 inputs rounded to bf16 (<= 1e-2 by construction)."""
    import kornia_amd as K

    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.rand(32, 3, 224, 224, device="cuda", generator=g).bfloat16()
    A = torch.tensor([[[0.96, 0.26, -5.0], [-0.26, 0.96, 30.0]]], device="cuda").repeat(32, 1, 1)
    w16 = K.warp_affine(x, A, (224, 224), align_corners=False)
    w32 = K.warp_affine(x.float(), A, (224, 224), align_corners=False)
    assert w16.dtype == torch.bfloat16 and torch.equal(w16, w32.bfloat16())
    sig = torch.rand(32, 2, device="cuda", generator=g) * 1.9 + 0.1
    b16 = K.gaussian_blur2d(x, (5, 5), sig)
    b32 = K.gaussian_blur2d(x.float(), (5, 5), sig)
    assert (b16.float() - b32).abs().max().item() <= 1e-2


def test_config3_sequence_runs_under_half_a_millisecond():
    """BASELINE config 3's per-GPU share (256 bf16 images of 224 x 224: RandomAffine -> ColorJitter -> RandomGaussianBlur with device
    parameters, p = 1) as a HIP-graph replay: finite output of the right shape in < 0.5 ms (round 1 timed this composite at 22 ms -
    a tensor was formatted into an error message on every call; the best of 5 x 20 replays keeps a cold box out of the verdict)."""
    import kornia_amd as K
    import kornia_amd.augmentation as A

    B = 256
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 3, 224, 224, generator=g).bfloat16().cuda()
    Pa = {"translations": (torch.rand(B, 2, generator=g) - 0.5) * 44.8, "center": torch.full((B, 2), 111.5), "scale": (0.8 + 0.4 * torch.rand(B, 1, generator=g)).expand(B, 2).contiguous(),
          "angle": (torch.rand(B, generator=g) - 0.5) * 30, "shear_x": (torch.rand(B, generator=g) - 0.5) * 10, "shear_y": torch.zeros(B)}
    Pj = {"brightness_factor": 0.8 + 0.4 * torch.rand(B, generator=g), "contrast_factor": 0.8 + 0.4 * torch.rand(B, generator=g),
          "saturation_factor": 0.8 + 0.4 * torch.rand(B, generator=g), "hue_factor": (torch.rand(B, generator=g) - 0.5) * 0.2}
    Pb = {"sigma": 0.1 + 1.9 * torch.rand(B, generator=g)}
    Pa, Pj, Pb = ({k: v.cuda() for k, v in d.items()} for d in (Pa, Pj, Pb))

    def seq(xx, a, j, b):
        return A.random_gaussian_blur(A.color_jitter(A.random_affine(xx, a), j, [0, 2, 3, 1]), b)

    step = K.graph.capture(seq, x, Pa, Pj, Pb, no_grad=True)
    out = step.replay()
    torch.cuda.synchronize()
    assert out.shape == x.shape and out.dtype == torch.bfloat16 and torch.isfinite(out.float()).all()
    with torch.no_grad():
        assert torch.equal(out, seq(x, Pa, Pj, Pb))
    best = float("inf")
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            step.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    assert best < 0.5, f"config 3 sequence: {best:.3f} ms per 256 images"
