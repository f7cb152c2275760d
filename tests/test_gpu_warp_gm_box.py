"""GPU (and, through tests/test_emulated_kernels.py, the host build of the kernels): the gradient of the bilinear warps with respect to the
matrix ALONE (the image needs no gradient - a learned homography over fixed images, BASELINE config 5) in the box form
(csrc/km_warp_gm_box.hip: the source box of a 64 x 32 output region through LDS by LDS-DMA, fp32 storage, zeros padding, RGB / grey): against the
oracle's autograd restatement (fp32 and fp64), and against the gather kernel (launch policy warp_gm_algo = 3) on the same inputs.  The cases
walk every way through the kernel: regions at the image's ragged right / bottom edge, boxes cut by the source image's borders, rotations whose
box does not fit (the gather rows inside the same kernel), channel counts and row lengths the box form does not take (the gather kernel)."""
import math

import pytest
import torch

from _util import flagship_homographies, rotation_affines

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).abs().amax(dim=(-2, -1)) / b.double().abs().amax(dim=(-2, -1)).clamp_min(1e-30)).max().item()


def _grad_wrt_matrix(kind, x, M, go, ds, align, algo=0):
    import kornia_amd as K
    from kornia_amd import _native as N

    prev = N.lib().km_config_set(b"warp_gm_algo", algo)
    try:
        Mg = M.cuda().requires_grad_()
        xc = x.cuda()  # (no gradient wrt the image: the matrix-gradient-only launch)
        if kind == "perspective":
            out = K.warp_perspective(xc, Mg, ds, "bilinear", "zeros", align)
        elif kind == "affine":
            out = K.warp_affine(xc, Mg, ds, "bilinear", "zeros", align)
        else:
            out = K.homography_warp(xc, Mg, ds, "bilinear", "zeros", align)
        out.backward(go.cuda())
        return Mg.grad.cpu()
    finally:
        N.lib().km_config_set(b"warp_gm_algo", prev)


def _oracle_grad(oracle, kind, x, M, go, ds, align):
    f = {"perspective": oracle.warp_perspective_backward, "affine": oracle.warp_affine_backward, "homography": oracle.homography_warp_backward}[kind]
    _, g32 = f(go, x, M, ds, "bilinear", "zeros", align)
    _, g64 = f(go.double(), x.double(), M.double(), ds, "bilinear", "zeros", align)
    return g32, g64


def _matrices(kind, B, H, W, h, w, g, rot=None):
    if kind == "perspective":
        M = flagship_homographies(B, H, W, h, w, g, jitter=5.0)
        if rot is not None:  # a rotation about the image centre in front of the jittered quad map
            c, s = math.cos(rot), math.sin(rot)
            cx, cy = (W - 1) / 2, (H - 1) / 2
            R = torch.tensor([[c, -s, cx - c * cx + s * cy], [s, c, cy - s * cx - c * cy], [0.0, 0.0, 1.0]])
            M = M @ R
        return M
    if kind == "affine":
        return rotation_affines(B, H, W, g)
    return torch.eye(3)[None] + 0.02 * torch.randn(B, 3, 3, generator=g)


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("kind", ["perspective", "affine", "homography"])
@pytest.mark.parametrize("shape", [(2, 3, 96, 160, 96, 160), (3, 1, 70, 132, 50, 100), (2, 3, 40, 64, 90, 200), (1, 3, 130, 68, 33, 65)])
def test_box_form_against_the_oracle_and_the_gather_kernel(oracle, shape, kind, align):
    B, C, H, W, h, w = shape
    g = torch.Generator().manual_seed(21)
    x = torch.rand(B, C, H, W, generator=g)
    M = _matrices(kind, B, H, W, h, w, g)
    go = torch.rand(B, C, h, w, generator=g) - 0.3
    gm = _grad_wrt_matrix(kind, x, M, go, (h, w), align)
    g32, g64 = _oracle_grad(oracle, kind, x, M, go, (h, w), align)
    assert gm.shape == M.shape and torch.isfinite(gm).all()
    assert _rel(gm, g32) < 5e-5, f"vs fp32 oracle {_rel(gm, g32):.3e}"
    assert _rel(gm, g64) < 5e-2, f"vs fp64 oracle {_rel(gm, g64):.3e}"
    rows = _grad_wrt_matrix(kind, x, M, go, (h, w), align, algo=3)
    assert _rel(gm, rows) < 2e-5, f"box form vs gather kernel {_rel(gm, rows):.3e}"  # (the same terms, other partial sums)


@pytest.mark.parametrize("deg", [4.0, 12.0, 45.0, 90.0])
def test_rotations_whose_box_does_not_fit_take_the_gather_rows(oracle, deg):
    B, C, H, W = 2, 3, 128, 192
    g = torch.Generator().manual_seed(22)
    x = torch.rand(B, C, H, W, generator=g)
    M = _matrices("perspective", B, H, W, H, W, g, rot=math.radians(deg))
    go = torch.rand(B, C, H, W, generator=g)
    gm = _grad_wrt_matrix("perspective", x, M, go, (H, W), True)
    g32, _ = _oracle_grad(oracle, "perspective", x, M, go, (H, W), True)
    assert _rel(gm, g32) < 5e-5, f"{deg} degrees: {_rel(gm, g32):.3e}"


@pytest.mark.parametrize("shape", [(2, 2, 64, 96, 64, 96), (2, 3, 50, 70, 48, 66), (1, 4, 33, 36, 40, 44)])
def test_what_the_box_form_does_not_take_still_runs(oracle, shape):
    """Channel counts other than 1 / 3 and rows that are not whole 16-byte chunks: the gather kernel, same results."""
    B, C, H, W, h, w = shape
    g = torch.Generator().manual_seed(23)
    x = torch.rand(B, C, H, W, generator=g)
    M = _matrices("perspective", B, H, W, h, w, g)
    go = torch.rand(B, C, h, w, generator=g)
    gm = _grad_wrt_matrix("perspective", x, M, go, (h, w), True)
    g32, _ = _oracle_grad(oracle, "perspective", x, M, go, (h, w), True)
    assert _rel(gm, g32) < 5e-5


def test_shared_matrix_and_magnified_and_minified_maps(oracle):
    g = torch.Generator().manual_seed(24)
    x = torch.rand(4, 3, 64, 128, generator=g)
    go = torch.rand(4, 3, 96, 160, generator=g)
    import kornia_amd as K

    # one matrix for the whole batch: every image adds to the same nine accumulators
    A = rotation_affines(1, 64, 128, g)
    Ag = A.cuda().requires_grad_()
    K.warp_affine(x.cuda(), Ag, (96, 160)).backward(go.cuda())
    _, ref = oracle.warp_affine_backward(go, x, A, (96, 160), "bilinear", "zeros", True)
    assert _rel(Ag.grad.cpu(), ref) < 5e-5
    for sc in (0.4, 2.5):  # minification: the box of a region does not fit; magnification: a few source pixels per region
        S = torch.tensor([[[sc, 0.0, 3.0], [0.0, sc, -2.0]]]).repeat(4, 1, 1)
        Sg = S.cuda().requires_grad_()
        K.warp_affine(x.cuda(), Sg, (96, 160)).backward(go.cuda())
        _, ref = oracle.warp_affine_backward(go, x, S, (96, 160), "bilinear", "zeros", True)
        assert _rel(Sg.grad.cpu(), ref) < 5e-5, sc


def test_non_finite_gradients_and_positions_follow_the_gather_kernel():
    """inf / NaN in grad_out and a matrix that is not a map: the box form gives what the gather kernel gives, entry for entry in the NaN pattern."""
    g = torch.Generator().manual_seed(25)
    x = torch.rand(2, 3, 64, 128, generator=g)
    M = flagship_homographies(2, 64, 128, 64, 128, g, jitter=3.0)
    go = torch.rand(2, 3, 64, 128, generator=g)
    go[0, 1, 10, 20] = float("inf")
    a = _grad_wrt_matrix("perspective", x, M, go, (64, 128), True)
    b = _grad_wrt_matrix("perspective", x, M, go, (64, 128), True, algo=3)
    assert torch.equal(torch.isfinite(a), torch.isfinite(b)) and _rel(a[1], b[1]) < 2e-5
    Mn = M.clone()
    Mn[1, 0, 0] = float("nan")
    go = torch.rand(2, 3, 64, 128, generator=g)
    a = _grad_wrt_matrix("perspective", x, Mn, go, (64, 128), True)
    b = _grad_wrt_matrix("perspective", x, Mn, go, (64, 128), True, algo=3)
    assert torch.equal(torch.isnan(a), torch.isnan(b)) and _rel(a[0], b[0]) < 2e-5
