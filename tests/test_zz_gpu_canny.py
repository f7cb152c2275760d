"""GPU: canny (kornia_amd/filters/canny.py) against the reference's outputs (tests/golden/canny.npz).  The front half runs on
km_filter2d_sep_fwd + km_spatial_gradient_fwd, the back half on km_canny_nms_fwd + km_canny_hysteresis_sweep (fp32 without
autograd; differentiable calls keep the tensor expressions).  Also run through the host build of the kernels
(tests/test_emulated_kernels.py)."""
import pytest
import torch

from _util import golden as _golden_np

pytestmark = pytest.mark.gpu


def golden(name):
    return {k: torch.from_numpy(v) for k, v in _golden_np(name).items()}


def _agree(a, b):
    """fraction of pixels on which two edge maps differ (a threshold crossing may flip on a 1-ulp magnitude difference)"""
    return (a != b).float().mean().item()


def test_canny_vs_reference():
    import kornia_amd as K

    d = golden("canny")
    x = d["x"].cuda()
    mag, edges = K.filters.canny(x)
    assert mag.shape == (2, 1, 48, 64) and edges.shape == (2, 1, 48, 64)
    assert _agree(mag.cpu() > 0, d["mag"] > 0) < 2e-3 and _agree(edges.cpu(), d["edges"]) < 2e-3
    keep = (mag.cpu() > 0) & (d["mag"] > 0)
    assert torch.allclose(mag.cpu()[keep], d["mag"][keep], atol=1e-5, rtol=0)
    mag, edges = K.filters.Canny(hysteresis=False)(x)
    assert _agree(edges.cpu(), d["edges_nohyst"]) < 2e-3 and set(edges.unique().tolist()) <= {0.0, 0.5, 1.0}
    mag, edges = K.filters.canny(x[:, :1], 0.05, 0.3, (3, 3), (0.8, 0.8))
    assert _agree(edges.cpu(), d["edges_gray_k3"]) < 2e-3
    keep = (mag.cpu() > 0) & (d["mag_gray_k3"] > 0)
    assert torch.allclose(mag.cpu()[keep], d["mag_gray_k3"][keep], atol=1e-5, rtol=0)
    assert d["edges"].sum() > 50  # the fixture has edges to find


def test_canny_argument_checks():
    import kornia_amd as K
    from kornia_amd.core.exceptions import BaseError

    x = torch.rand(1, 1, 8, 8).cuda()
    with pytest.raises(BaseError, match="Invalid input thresholds"):
        K.filters.canny(x, 0.3, 0.2)
    with pytest.raises(BaseError, match="Invalid low threshold"):
        K.filters.canny(x, 0.0, 0.2)
    with pytest.raises(BaseError, match="Invalid high threshold"):
        K.filters.Canny(0.1, 1.0)
    with pytest.raises(BaseError):
        K.filters.canny(x[0])
    assert repr(K.filters.Canny())


def test_canny_known_answer_cross():
    """The 5x5 cross of the reference's tests/filters/test_canny.py:85-137 (magnitude and edges, atol 1e-4), its cardinality and
    non-contiguous-input cases (:43-50, 77-83)."""
    import kornia_amd as K

    cross = torch.zeros(1, 1, 5, 5)
    cross[0, 0, 1:4, 2] = 1.0
    cross[0, 0, 2, 1:4] = 1.0
    ring = torch.tensor([[1.2458, 0.9672, 1.2458], [0.9672, 0.0, 0.9672], [1.2458, 0.9672, 1.2458]])
    want_mag, want_edges = torch.zeros(1, 1, 5, 5), torch.zeros(1, 1, 5, 5)
    want_mag[0, 0, 1:4, 1:4] = ring
    want_edges[0, 0, 1:4, 1:4] = (ring > 0).float()
    mag, edges = K.filters.canny(cross.cuda())
    assert torch.allclose(mag.cpu(), want_mag, atol=1e-4, rtol=1e-4) and torch.allclose(edges.cpu(), want_edges, atol=1e-4, rtol=1e-4)
    for batch in (1, 2):
        m, e = K.filters.canny(torch.rand(batch, 3, 4, 4).cuda())
        assert m.shape == (batch, 1, 4, 4) and e.shape == (batch, 1, 4, 4)
    nc = torch.rand(2, 3, 5, 5).cuda().expand(2, -1, -1, -1)[..., ::1]
    m, e = K.filters.canny(nc.transpose(2, 3))
    assert m.is_contiguous() and e.shape == (2, 1, 5, 5)


@pytest.mark.parametrize("shape,hyst", [((2, 3, 70, 130), True), ((1, 1, 64, 64), True), ((2, 1, 33, 200), False)])
def test_canny_native_tail_matches_the_tensor_expressions(shape, hyst):
    """km_canny_nms_fwd + km_canny_hysteresis_sweep against the differentiable composition of the same arithmetic (taken when the
    input requires grad): same magnitude where both keep a pixel, edge maps equal up to threshold crossings of 1-ulp differences, and the
    hysteresis fixed point - long weak chains that cross tile borders included - is the same set of pixels."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(7)
    x = torch.rand(*shape, generator=g)
    # smooth structure so that weak edges form chains: a few blurred blobs on top of the noise
    yy, xx = torch.meshgrid(torch.arange(shape[2], dtype=torch.float32), torch.arange(shape[3], dtype=torch.float32), indexing="ij")
    x = 0.15 * x + 0.85 * (torch.sin(xx / 9.0) * torch.cos(yy / 7.0))[None, None] * 0.5 + 0.4
    xd = x.cuda()
    mag_n, edges_n = K.filters.canny(xd, 0.05, 0.4, hysteresis=hyst)
    mag_c, edges_c = K.filters.canny(xd.clone().requires_grad_(), 0.05, 0.4, hysteresis=hyst)
    mag_c, edges_c = mag_c.detach(), edges_c.detach()
    assert mag_n.shape == mag_c.shape and edges_n.shape == edges_c.shape
    assert _agree(mag_n > 0, mag_c > 0) < 2e-3 and _agree(edges_n, edges_c) < 2e-3
    keep = (mag_n > 0) & (mag_c > 0)
    assert torch.allclose(mag_n[keep], mag_c[keep], atol=1e-6, rtol=0)
    assert set(edges_n.unique().tolist()) <= ({0.0, 1.0} if hyst else {0.0, 0.5, 1.0})
    if hyst:
        assert edges_n.sum() > 0


def test_hysteresis_sweeps_follow_a_chain_across_tiles():
    """km_canny_hysteresis_sweep on a hand-made state: a weak serpentine that crosses the 64 x 64 tile grid many times with ONE strong
    seed at its end must be promoted completely (several sweeps: a sweep carries a promotion only as far as tiles that were already
    consistent), an isolated weak segment must be dropped, and the result must equal the reference's pixel-by-pixel loop."""
    from kornia_amd import _native as N

    lib = N.lib()
    H, W = 150, 200
    state = torch.zeros(1, 1, H, W)
    # serpentine: horizontal runs every 6 rows joined at alternating ends
    rows = list(range(3, H - 3, 6))
    for k, r in enumerate(rows):
        state[0, 0, r, 5:W - 5] = 0.5
        if k + 1 < len(rows):
            c = W - 6 if k % 2 == 0 else 5
            state[0, 0, r:rows[k + 1] + 1, c] = 0.5
    state[0, 0, rows[-1], 5 if len(rows) % 2 == 0 else W - 6] = 1.0  # the seed, at the far end of the chain
    state[0, 0, 1, 20:40] = 0.5  # touches nothing strong (two rows above the first run)
    # the reference's loop (kornia/filters/canny.py:156-176) on the host
    ref = state.clone()
    while True:
        strong = ref == 1
        grown = torch.nn.functional.max_pool2d(strong.float(), 3, 1, 1) > 0
        new = torch.where((ref == 0.5) & grown, torch.ones_like(ref), ref)
        if torch.equal(new, ref):
            break
        ref = new
    want = (ref == 1).float()
    assert want.sum() > (W - 10) * len(rows)  # the whole chain

    st = state.cuda().contiguous()
    out = torch.empty_like(st)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    sweeps = 0
    while True:
        N.check(lib.km_canny_hysteresis_sweep(st.data_ptr(), out.data_ptr(), flag.data_ptr(), 1, H, W, N.stream_ptr(st.device)), "sweep")
        sweeps += 1
        if int(flag.item()) == 0:
            break
        flag.zero_()
        assert sweeps < 200
    assert torch.equal(out.cpu(), want)
    assert 2 <= sweeps <= 2 + len(rows) * 4  # more than one sweep (the chain crosses tiles), far fewer than one per pixel of its length
