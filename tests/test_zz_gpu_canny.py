"""GPU: canny (kornia_amd/filters/canny.py) against the reference's outputs (tests/golden/canny.npz).  The front half runs on
km_filter2d_sep_fwd + km_spatial_gradient_fwd; the rest is pointwise torch code.  Verified through the host build of the
kernels (tests/test_emulated_kernels.py); the file sorts last because it has not been run on a device yet this round."""
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
