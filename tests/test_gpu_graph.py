"""The native ops are allocation-free, stream-ordered C calls, so a whole fwd(+bwd) step can be captured into a HIP graph
(torch.cuda.CUDAGraph) and replayed - the launch-bound regime of small batches (SURVEY config 1 sizes)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B=4, S=64, seed=0):
    from _util import flagship_homographies

    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, S, S, generator=g)
    M = flagship_homographies(B, S, S, S, S, g, jitter=2.0)
    return x.cuda(), M.cuda()


def test_forward_capture_and_replay():
    import kornia_amd as K

    x, M = _inputs()
    S = x.shape[-1]

    def step(a, m):
        return K.sobel(K.gaussian_blur2d(K.warp_perspective(a, m, (S, S)), (5, 5), (1.5, 1.5)))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(2):  # warm-up: library load, tap cache
            step(x, M)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        y = step(x, M)
    x2, M2 = _inputs(seed=1)
    x.copy_(x2)
    M.copy_(M2)
    graph.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = step(x2, M2)
    assert torch.equal(y, ref)


def test_training_step_capture_and_replay():
    import kornia_amd as K

    x, M = _inputs(B=2, S=48)
    S = x.shape[-1]
    xg, Mg = x.clone().requires_grad_(), M.clone().requires_grad_()
    go = torch.rand_like(x)

    def step():
        y = K.gaussian_blur2d(K.warp_perspective(xg, Mg, (S, S)), (5, 5), (1.5, 1.5))
        gx, gM = torch.autograd.grad(y, (xg, Mg), go)
        return y, gx, gM

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y, gx, gM = step()
    x2, M2 = _inputs(B=2, S=48, seed=3)
    with torch.no_grad():
        xg.copy_(x2)
        Mg.copy_(M2)
    graph.replay()
    torch.cuda.synchronize()
    y_r, gx_r, gM_r = step()
    assert torch.equal(y, y_r) and torch.equal(gx, gx_r)  # grad wrt the image is deterministic (integer accumulation)
    assert torch.allclose(gM, gM_r, rtol=1e-5, atol=1e-6)  # fp64 atomics: order-dependent in the last bits only


def test_capture_helper_config3_sequence_and_config5_step():
    """kornia_amd.graph.capture: BASELINE config 3's augmentation sequence (device parameters, bf16) and config 5's learned-homography
    step (forward + gradient wrt H) replayed from HIP graphs on new inputs give what the eager calls give."""
    import kornia_amd as K
    import kornia_amd.augmentation as A

    g = torch.Generator().manual_seed(5)
    B = 8

    def params(seed):
        gg = torch.Generator().manual_seed(seed)
        Pa = {"translations": (torch.rand(B, 2, generator=gg) - 0.5) * 20, "center": torch.full((B, 2), 55.5), "scale": (0.8 + 0.4 * torch.rand(B, 1, generator=gg)).expand(B, 2).contiguous(),
              "angle": (torch.rand(B, generator=gg) - 0.5) * 30, "shear_x": (torch.rand(B, generator=gg) - 0.5) * 10, "shear_y": torch.zeros(B),
              "batch_prob": (torch.rand(B, generator=gg) > 0.3).float()}
        Pj = {"brightness_factor": 0.8 + 0.4 * torch.rand(B, generator=gg), "contrast_factor": 0.8 + 0.4 * torch.rand(B, generator=gg),
              "saturation_factor": 0.8 + 0.4 * torch.rand(B, generator=gg), "hue_factor": (torch.rand(B, generator=gg) - 0.5) * 0.2}
        Pb = {"sigma": 0.1 + 1.9 * torch.rand(B, generator=gg)}
        return [{k: v.cuda() for k, v in d.items()} for d in (Pa, Pj, Pb)]

    order = [0, 2, 3, 1]

    def seq(x, Pa, Pj, Pb):
        return A.random_gaussian_blur(A.color_jitter(A.random_affine(x, Pa), Pj, order), Pb)

    x = torch.rand(B, 3, 112, 112, generator=g).bfloat16().cuda()
    step = K.graph.capture(seq, x, *params(1), no_grad=True)
    x2 = torch.rand(B, 3, 112, 112, generator=g).bfloat16().cuda()
    P2 = params(2)
    out = step(x2, *P2).clone()
    with torch.no_grad():
        assert torch.equal(out, seq(x2, *P2))

    xs = torch.rand(4, 3, 64, 64, generator=g).cuda()
    tgt = torch.rand(4, 3, 64, 64, generator=g).cuda()
    H = (torch.eye(3)[None] + 0.01 * torch.randn(4, 3, 3, generator=g)).cuda().requires_grad_()

    def train(a, h, t):
        (gh,) = torch.autograd.grad(torch.nn.functional.l1_loss(K.homography_warp(a, h, (64, 64)), t), h)
        return gh

    tstep = K.graph.capture(train, xs, H, tgt)
    H2 = (torch.eye(3)[None] + 0.01 * torch.randn(4, 3, 3, generator=g)).cuda().requires_grad_()
    got = tstep(xs, H2, tgt).clone()
    assert torch.allclose(got, train(xs, H2, tgt), rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        tstep(xs[:2], H2, tgt)
