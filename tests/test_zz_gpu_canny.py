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
