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
