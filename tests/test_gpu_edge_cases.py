"""GPU: edge cases the reference's tests exercise - empty batches, 1-pixel extents, non-contiguous /
expanded inputs (tests/filters/test_filters.py:359-366), channels-last strides, mixed dtypes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_empty_batch():
    import kornia_amd as K

    x = torch.rand(0, 3, 8, 9, device="cuda")
    assert K.warp_perspective(x, torch.zeros(0, 3, 3, device="cuda"), (5, 6)).shape == (0, 3, 5, 6)
    assert K.warp_affine(x, torch.zeros(0, 2, 3, device="cuda"), (5, 6)).shape == (0, 3, 5, 6)
    assert K.gaussian_blur2d(x, (3, 3), (1.0, 1.0)).shape == (0, 3, 8, 9)
    assert K.filter2d(x, torch.ones(1, 3, 3, device="cuda")).shape == (0, 3, 8, 9)
    assert K.spatial_gradient(x).shape == (0, 3, 2, 8, 9)
    xg = torch.rand(0, 3, 8, 9, device="cuda", requires_grad=True)
    K.gaussian_blur2d(K.warp_perspective(xg, torch.zeros(0, 3, 3, device="cuda"), (8, 9)), (3, 3), (1.0, 1.0)).sum().backward()
    assert xg.grad.shape == xg.shape


def test_tiny_extents(oracle):
    import kornia_amd as K

    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 2, 5, 7, generator=g)
    M = torch.eye(3)[None].repeat(2, 1, 1)
    M[:, 0, 2] = 0.5
    for ds in [(1, 4), (3, 1), (1, 1), (2, 2)]:
        ref = oracle.warp_perspective(x, M, ds)
        out = K.warp_perspective(x.cuda(), M.cuda(), ds).cpu()
        assert out.shape == ref.shape
        both = torch.isfinite(ref) & torch.isfinite(out)
        assert torch.equal(torch.isfinite(ref), torch.isfinite(out))  # 1-pixel extents divide by (n-1) = 0 in the reference too
        assert torch.equal(out[both], ref[both])
    x1 = torch.rand(1, 1, 1, 9, generator=g)  # single-row source
    A = torch.tensor([[[1.0, 0.0, 0.5], [0.0, 1.0, 0.0]]])
    ref = oracle.warp_affine(x1, A, (1, 9), align_corners=False)
    out = K.warp_affine(x1.cuda(), A.cuda(), (1, 9), align_corners=False).cpu()
    both = torch.isfinite(ref) & torch.isfinite(out)
    assert torch.equal(out[both], ref[both])
    xs = torch.rand(1, 1, 3, 3, generator=g)
    assert torch.equal(K.gaussian_blur2d(xs.cuda(), (3, 3), (1.0, 1.0)).cpu(), oracle.gaussian_blur2d(xs, (3, 3), (1.0, 1.0)))
    assert torch.equal(K.spatial_gradient(xs.cuda()).cpu(), oracle.spatial_gradient(xs))
    one = torch.rand(1, 1, 1, 1, generator=g)
    assert torch.equal(K.spatial_gradient(one.cuda()).cpu(), oracle.spatial_gradient(one))
    assert torch.equal(K.filter2d(one.cuda(), torch.ones(1, 1, 1).cuda()).cpu(), one)


def test_non_contiguous_inputs(oracle):
    import kornia_amd as K

    g = torch.Generator().manual_seed(4)
    base = torch.rand(1, 1, 12, 14, generator=g)
    x = base.expand(2, 3, 12, 14)  # stride-0 view, as in the reference's test_noncontiguous
    k = torch.rand(1, 3, 3, generator=g)
    out = K.filter2d(x.cuda(), k.cuda())
    assert out.is_contiguous() and torch.equal(out.cpu(), oracle.filter2d(x.contiguous(), k))
    xt = torch.rand(2, 12, 14, 3, generator=g).permute(0, 3, 1, 2)  # channels-last strides
    M = torch.eye(3)[None].repeat(2, 1, 1)
    M[:, 1, 2] = 1.25
    assert torch.equal(K.warp_perspective(xt.cuda(), M.cuda(), (12, 14)).cpu(), oracle.warp_perspective(xt.contiguous(), M, (12, 14)))
    assert torch.equal(K.gaussian_blur2d(xt.cuda(), (5, 5), (1.5, 1.5)).cpu(), oracle.gaussian_blur2d(xt.contiguous(), (5, 5), (1.5, 1.5)))
    Mt = M.transpose(1, 2).contiguous().transpose(1, 2)  # non-contiguous matrix
    assert torch.equal(K.warp_perspective(xt.cuda(), Mt.cuda(), (12, 14)).cpu(), oracle.warp_perspective(xt.contiguous(), M, (12, 14)))
    go = torch.rand(2, 3, 12, 14, generator=g).cuda().permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)  # non-contiguous grad
    xg = xt.cuda().requires_grad_()
    K.gaussian_blur2d(xg, (5, 5), (1.5, 1.5)).backward(go)
    assert torch.allclose(xg.grad.cpu(), oracle.gaussian_blur2d_backward(go.cpu().contiguous(), xt.contiguous(), (5, 5), (1.5, 1.5)), atol=1e-5)


def test_mixed_dtypes_and_streams(oracle):
    import kornia_amd as K

    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 16, 18, generator=g)
    M = torch.eye(3)[None].repeat(2, 1, 1)
    M[:, 0, 2] = -1.5
    # float64 matrix with a float32 image: computed in the image's precision
    assert torch.equal(K.warp_perspective(x.cuda(), M.double().cuda(), (16, 18)).cpu(), oracle.warp_perspective(x, M, (16, 18)))
    # a side stream: launches follow torch's current stream
    s = torch.cuda.Stream()
    xc, Mc = x.cuda(), M.cuda()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        y = K.gaussian_blur2d(K.warp_perspective(xc, Mc, (16, 18)), (5, 5), (1.5, 1.5))
    s.synchronize()
    assert torch.equal(y.cpu(), oracle.gaussian_blur2d(oracle.warp_perspective(x, M, (16, 18)), (5, 5), (1.5, 1.5)))
    # float64 end to end
    y64 = K.gaussian_blur2d(K.warp_perspective(xc.double(), Mc.double(), (16, 18)), (5, 5), (1.5, 1.5))
    ref64 = oracle.gaussian_blur2d(oracle.warp_perspective(x.double(), M.double(), (16, 18)), (5, 5), (1.5, 1.5))
    assert y64.dtype == torch.float64 and torch.allclose(y64.cpu(), ref64, atol=1e-13)


def test_batch_traversal_direction_does_not_change_results():
    """km_set_traversal: consecutive launches of the streaming kernels walk the batch in alternating directions (a consumer starts on what
    its producer left in the Infinity Cache).  A launch policy only: forward, blur and the image gradient are bit-identical whatever the
    direction (the scatter accumulates integers), the matrix gradient - fp64 atomics in workgroup order - within 1e-12 relative."""
    import kornia_amd as K
    from kornia_amd import _native as N

    lib = N.lib()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(5, 3, 96, 130, generator=g).cuda()
    M = (torch.eye(3) + 0.02 * torch.randn(5, 3, 3, generator=g)).cuda()
    M[:, 2, :2] *= 0.01
    go = torch.rand(5, 3, 80, 112, generator=g).cuda()

    def run():
        xx, MM = x.clone().requires_grad_(), M.clone().requires_grad_()
        y = K.filters.gaussian_blur2d(K.geometry.transform.warp_perspective(xx, MM, (80, 112)), (5, 5), (1.5, 1.5))
        y.backward(go)
        return y.detach(), xx.grad, MM.grad

    prev = lib.km_set_traversal(1)  # fixed: every launch forward
    try:
        ref = run()
        assert lib.km_set_traversal(0) == 1  # alternating (the default): two runs start with opposite parities at some launch
        for _ in range(3):
            got = run()
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
            assert (got[2] - ref[2]).abs().max().item() <= 1e-6 * ref[2].abs().max().item()  # fp32 partial sums per thread over the tiles a workgroup walks: the grouping follows the direction
    finally:
        lib.km_set_traversal(prev)


def test_two_threads_two_streams_share_no_launch_state(oracle):
    """the C ABI is re-entrant - no launch state shared between streams.  Two host threads drive the hot step on two
    streams at once; the traversal parity of the streaming kernels is kept per (device, stream), so each thread's launches alternate
    on their own, and both get the results of a lone run (bit-identical forward and image gradient)."""
    import threading

    import kornia_amd as K

    g = torch.Generator().manual_seed(8)
    xs = [torch.rand(3, 3, 96, 130, generator=g) for _ in range(2)]
    Ms = [torch.eye(3) + 0.02 * torch.randn(3, 3, 3, generator=g) for _ in range(2)]
    for M in Ms:
        M[:, 2, :2] *= 0.01
    gos = [torch.rand(3, 3, 80, 112, generator=g) for _ in range(2)]

    def run(k, stream, out, reps):
        with torch.cuda.stream(stream):
            for _ in range(reps):
                x = xs[k].cuda().requires_grad_()
                M = Ms[k].cuda().requires_grad_()
                y = K.filters.gaussian_blur2d(K.geometry.transform.warp_perspective(x, M, (80, 112)), (5, 5), (1.5, 1.5))
                y.backward(gos[k].cuda())
            stream.synchronize()
            out[k] = (y.detach().cpu(), x.grad.cpu(), M.grad.cpu())

    lone = {}
    for k in range(2):
        run(k, torch.cuda.current_stream(), lone, 1)
    both = {}
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    threads = [threading.Thread(target=run, args=(k, streams[k], both, 5 + k)) for k in range(2)]  # 5 and 6 steps: the parities drift apart
    for t_ in threads:
        t_.start()
    for t_ in threads:
        t_.join()
    for k in range(2):
        assert torch.equal(both[k][0], lone[k][0]) and torch.equal(both[k][1], lone[k][1])
        assert (both[k][2] - lone[k][2]).abs().max().item() <= 1e-6 * lone[k][2].abs().max().item()
