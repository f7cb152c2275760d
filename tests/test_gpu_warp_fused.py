"""GPU: the one-read backward of the bilinear warps (csrc/km_warp_bwd_fused.hip: km_warp2d_bwd_ws with a workspace) - both gradients
from one pass over grad_out - against the CPU oracle and against the two-launch form it replaces.  Also runs on the host build of the
kernels (tests/test_emulated_kernels.py)."""
import pytest
import torch

from _util import flagship_homographies, rotation_affines, smooth_image

pytestmark = pytest.mark.gpu


def _lib():
    from kornia_amd import _native as N

    return N.lib()


def _run(fn, x, M, go, fused, **kw):
    """(grad_x, grad_M) of fn(x, M, ...) through the public API with the one-read backward on / off"""
    lib = _lib()
    prev = lib.km_config_set(b"warp_bwd_fused", 1 if fused else 0)
    try:
        xg, Mg = x.cuda().requires_grad_(), M.cuda().requires_grad_()
        fn(xg, Mg, **kw).backward(go.cuda())
    finally:
        lib.km_config_set(b"warp_bwd_fused", prev)
    return xg.grad.cpu(), Mg.grad.cpu()


def _rel(a, b):
    return ((a.double() - b.double()).abs().amax(dim=(-2, -1)) / b.double().abs().amax(dim=(-2, -1)).clamp_min(1e-30)).max().item()


@pytest.mark.parametrize("C", [3, 1])
@pytest.mark.parametrize("pad", ["zeros", "fill"])
@pytest.mark.parametrize("shape", [(2, 128, 192, 128, 192), (3, 200, 130, 150, 170), (2, 70, 66, 90, 61)])
def test_both_gradients_from_one_read_match_the_oracle(oracle, shape, pad, C):
    """warp_perspective, several tiles per image, ragged right / bottom tiles: grad wrt the image <= 1e-5 of the oracle (the exact
    fixed-point scale makes it tighter than the two launches'), grad wrt the matrix to 5e-5 of the fp32 oracle that shares the kernel's
    sampling positions."""
    import kornia_amd as K

    if pad == "fill" and C != 3:
        pytest.skip("fill_value is RGB in the reference")
    B, H, W, h, w = shape
    g = torch.Generator().manual_seed(B * H + w)
    x = torch.rand(B, C, H, W, generator=g)
    M = flagship_homographies(B, H, W, h, w, g, jitter=6.0)
    go = torch.rand(B, C, h, w, generator=g) - 0.4
    kw = {}
    if pad == "fill":
        kw = dict(padding_mode="fill", fill_value=torch.tensor([0.2, 0.5, 0.7]))
    gx, gM = _run(lambda a, m: K.warp_perspective(a, m, (h, w), **kw), x, M, go, True)
    gx2, gM2 = _run(lambda a, m: K.warp_perspective(a, m, (h, w), **kw), x, M, go, False)
    gxo, gMo = oracle.warp_perspective_backward(go, x, M, (h, w), **kw)
    assert torch.allclose(gx, gxo, atol=1e-5, rtol=0), (gx - gxo).abs().max()
    assert _rel(gM, gMo) <= 5e-5
    # the two forms agree with each other (they differ by the rounding of the fixed-point scale and the order of the fp64 sums)
    assert torch.allclose(gx, gx2, atol=5e-6 * go.abs().max().item(), rtol=0)  # each within 2e-6 max|grad_out| of the exact sum
    assert _rel(gM, gM2) <= 5e-5


@pytest.mark.parametrize("C", [2, 4, 5, 6, 7])
def test_any_channel_count_runs_through_the_same_tiles(oracle, C):
    """C = 3 a + r: a groups of three through the RGB instantiation, r single channels through the grey one (one launch sequence each), all
    adding to the same matrix gradient.  The one-read path must be the one that ran (workspace > 0), results as for RGB."""
    import kornia_amd as K

    lib = _lib()
    B, H, W, h, w = 2, 130, 140, 100, 150
    assert lib.km_warp2d_bwd_workspace_bytes(B, C, H, W, h, w, 1, 0, 0) > 0
    g = torch.Generator().manual_seed(C)
    x = torch.rand(B, C, H, W, generator=g)
    M = flagship_homographies(B, H, W, h, w, g, jitter=5.0)
    go = torch.rand(B, C, h, w, generator=g) - 0.4
    gx, gM = _run(lambda a, m: K.warp_perspective(a, m, (h, w)), x, M, go, True)
    gx2, gM2 = _run(lambda a, m: K.warp_perspective(a, m, (h, w)), x, M, go, False)
    gxo, gMo = oracle.warp_perspective_backward(go, x, M, (h, w))
    assert torch.allclose(gx, gxo, atol=1e-5, rtol=0), (gx - gxo).abs().max()
    assert _rel(gM, gMo) <= 5e-5 and _rel(gM, gM2) <= 5e-5
    # 16-bit storage through the grouped form
    gx, gM = _run(lambda a, m: K.warp_perspective(a, m, (h, w)), x.half(), M, go.half(), True)
    gx2, gM2 = _run(lambda a, m: K.warp_perspective(a, m, (h, w)), x.half(), M, go.half(), False)
    assert _rel(gM, gM2) <= 2e-3 and torch.allclose(gx.float(), gx2.float(), atol=2e-3, rtol=0)


@pytest.mark.parametrize("case", ["shift_right_down", "rotate_12deg", "flagship", "far_left", "shrink"])
@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("pad", ["border", "reflection"])
def test_border_and_reflection_padding_through_the_tile_owners(oracle, pad, align, case):
    """padding_mode = border / reflection: every sampling position is brought into the image first (ATen's clip / reflect_coordinates), so
    the tiles along an edge also own the output pixels that map beyond it (border) and every tile the pixels that map into its mirror
    images (reflection) - their boxes are the bounding boxes of those copies (kmt_tile_box_padded).  Both gradients against the oracle; the
    cells many outside pixels pile onto hold large sums, so the bound on the image gradient is relative to its largest entry."""
    import kornia_amd as K

    lib = _lib()
    B, C, H, W, h, w = 2, 3, 150, 200, 150, 200
    assert lib.km_warp2d_bwd_workspace_bytes(B, C, H, W, h, w, 1, {"border": 1, "reflection": 2}[pad], 0) > 0
    g = torch.Generator().manual_seed(17)
    x = torch.rand(B, C, H, W, generator=g)
    go = torch.rand(B, C, h, w, generator=g) - 0.4
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    if case == "flagship":
        M = flagship_homographies(B, H, W, h, w, g, jitter=10.0)
        fn = lambda a, m: K.warp_perspective(a, m, (h, w), padding_mode=pad, align_corners=align)
        bwd = lambda: oracle.warp_perspective_backward(go, x, M, (h, w), padding_mode=pad, align_corners=align)
    else:
        import math

        if case == "shift_right_down":
            A = torch.tensor([[1.0, 0.0, 23.4], [0.0, 1.0, 17.8]])
        elif case == "far_left":
            A = torch.tensor([[1.0, 0.02, -70.0], [-0.01, 1.0, 3.0]])
        elif case == "shrink":
            A = torch.tensor([[0.8, 0.0, 0.2 * cx], [0.0, 0.85, 0.15 * cy]])  # the image shrinks around its centre: a frame of padding all around
        else:
            c_, s_ = math.cos(math.radians(12.0)), math.sin(math.radians(12.0))
            A = torch.tensor([[c_, s_, (1 - c_) * cx - s_ * cy], [-s_, c_, s_ * cx + (1 - c_) * cy]])
        M = A.repeat(B, 1, 1)
        M[1, :, 2] += torch.tensor([-9.0, 6.0])
        fn = lambda a, m: K.warp_affine(a, m, (h, w), padding_mode=pad, align_corners=align)
        bwd = lambda: oracle.warp_affine_backward(go, x, M, (h, w), padding_mode=pad, align_corners=align)
    gx, gM = _run(fn, x, M, go, True)
    gx2, gM2 = _run(fn, x, M, go, False)  # (the generic kernel: global atomics)
    gxo, gMo = bwd()
    scale = max(1.0, gxo.abs().max().item())
    assert torch.allclose(gx, gxo, atol=2e-5 * scale, rtol=0), ((gx - gxo).abs().max(), scale)
    assert torch.allclose(gx2, gxo, atol=2e-5 * scale, rtol=0)
    assert _rel(gM, gMo) <= 2e-4, _rel(gM, gMo)
    # the image gradient alone takes the same kernel (no matrix gradient committed)
    xg = x.cuda().requires_grad_()
    fn(xg, M.cuda()).backward(go.cuda())
    assert torch.allclose(xg.grad.cpu(), gxo, atol=2e-5 * scale, rtol=0)


def test_affine_and_homography_modes(oracle):
    """The other two coordinate generators (warp_affine with and without align_corners, homography_warp) through the one-read backward."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(5)
    B, C, H, W, h, w = 3, 3, 140, 200, 120, 180
    x = smooth_image(B, C, H, W) + 0.05 * torch.rand(B, C, H, W, generator=g)
    go = torch.rand(B, C, h, w, generator=g)
    A = rotation_affines(B, H, W, g)
    A[:, :, :2] *= 0.3 * torch.rand(B, 1, 1, generator=g) + 0.85  # keep the boxes of the tiles small enough for the persistent loop
    for align in (True, False):
        gx, gA = _run(lambda a, m: K.warp_affine(a, m, (h, w), align_corners=align), x, A, go, True)
        gxo, gAo = oracle.warp_affine_backward(go, x, A, (h, w), align_corners=align)
        assert torch.allclose(gx, gxo, atol=1e-5, rtol=0)
        assert _rel(gA, gAo) <= 5e-5
    Hn = torch.eye(3)[None] + 0.03 * torch.randn(B, 3, 3, generator=g)
    gx, gH = _run(lambda a, m: K.homography_warp(a, m, (h, w)), x, Hn, go, True)
    gx2, gH2 = _run(lambda a, m: K.homography_warp(a, m, (h, w)), x, Hn, go, False)
    assert torch.allclose(gx, gx2, atol=5e-6, rtol=0)
    assert _rel(gH, gH2) <= 5e-5


def test_more_tiles_than_workers():
    """Every workgroup of the persistent launch walks SEVERAL tiles: the register prefetch of tile k + 1 during tile k, the ring of tile
    records (more than 64 tiles per worker on the host build), runs that span images.  Against the two launches on the same inputs."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(9)
    # the device has 256 workers (one per CU): 24 x 64 tiles = 6 per worker; the host build has 3 workers: 5 x 63 tiles = 105 per worker
    B, H, W = (24, 512, 512) if _lib().km_device_info(None, 0) > 0 else (5, 400, 560)
    x = torch.rand(B, 3, H, W, generator=g)
    M = flagship_homographies(B, H, W, H, W, g, jitter=8.0)
    go = torch.rand(B, 3, H, W, generator=g)
    gx, gM = _run(lambda a, m: K.warp_perspective(a, m, (H, W)), x, M, go, True)
    gx2, gM2 = _run(lambda a, m: K.warp_perspective(a, m, (H, W)), x, M, go, False)
    assert torch.allclose(gx, gx2, atol=5e-6, rtol=0), (gx - gx2).abs().max()
    assert _rel(gM, gM2) <= 5e-5


def test_shared_matrix_accumulates_over_the_batch(oracle):
    """One (1,3,3) matrix for the whole batch: every image's share of the matrix gradient lands in the same nine accumulators."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(2)
    x = torch.rand(4, 3, 96, 150, generator=g)
    M = flagship_homographies(1, 96, 150, 96, 150, g, jitter=3.0)
    go = torch.rand(4, 3, 96, 150, generator=g)
    gx, gM = _run(lambda a, m: K.warp_perspective(a, m.expand(4, 3, 3), (96, 150)), x, M, go, True)
    gxo, gMo = oracle.warp_perspective_backward(go, x, M.expand(4, 3, 3), (96, 150))
    assert torch.allclose(gx, gxo, atol=1e-5, rtol=0)
    assert _rel(gM, gMo.sum(0, keepdim=True)) <= 5e-5


@pytest.mark.parametrize("case", ["minify", "magnify", "nonfinite", "vanishing"])
def test_tiles_of_the_general_launch(oracle, case):
    """Tiles the persistent loop leaves to the general launch: boxes that do not fit its registers (2.2x minification), magnification
    beyond the fixed-point head-room, NaN / inf in grad_out (IEEE propagation like the oracle's), the vanishing line inside the image."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(11)
    B, C = 2, 3
    if case == "minify":
        H, W, h, w = 192, 256, 80, 112
    elif case == "magnify":
        H, W, h, w = 24, 20, 200, 180
    else:
        H, W, h, w = 128, 128, 128, 128
    x = torch.rand(B, C, H, W, generator=g)
    M = flagship_homographies(B, H, W, h, w, g, jitter=2.0)
    if case == "vanishing":
        M = torch.eye(3).repeat(B, 1, 1)
        M[:, 2, 0] = 1.0 / 90.0   # the denominator vanishes on a line that crosses the image
        M[:, 2, 2] = -0.35
    go = torch.rand(B, C, h, w, generator=g) - 0.5
    if case == "nonfinite":
        go[0, 1, 40, 50] = float("nan")
        go[1, 2, 70, 3] = float("inf")
    gx, gM = _run(lambda a, m: K.warp_perspective(a, m, (h, w)), x, M, go, True)
    gx2, gM2 = _run(lambda a, m: K.warp_perspective(a, m, (h, w)), x, M, go, False)
    if case == "nonfinite":
        gxo, _ = oracle.warp_perspective_backward(go, x, M, (h, w))
        assert torch.equal(torch.isnan(gx), torch.isnan(gxo)) and torch.equal(torch.isinf(gx), torch.isinf(gxo))
        fin = torch.isfinite(gxo)
        assert torch.allclose(gx[fin], gxo[fin], atol=1e-5, rtol=0)
        assert torch.isnan(gM).any()
        return
    # against the ORACLE (the two-launch HIP form only as a cross-check).  Under magnification ~9 x 9 footprints add to one source pixel, so
    # the bound is relative to the largest entry (each contribution is within one fp32 ulp of max|grad_out|); at the vanishing line the
    # positions of a few pixels are +-inf / NaN in both, and what the oracle holds finite must agree
    gxo, gMo = oracle.warp_perspective_backward(go, x, M, (h, w))
    fin = torch.isfinite(gxo)
    assert torch.equal(torch.isfinite(gx), fin) and torch.equal(torch.isfinite(gx2), fin)
    scale = max(1.0, gxo[fin].abs().max().item())
    assert torch.allclose(gx[fin], gxo[fin], atol=1e-5 * scale, rtol=0), (gx[fin] - gxo[fin]).abs().max()
    assert torch.allclose(gx[fin], gx2[fin], atol=2e-5 * scale, rtol=0), (gx[fin] - gx2[fin]).abs().max()
    assert torch.equal(torch.isfinite(gM), torch.isfinite(gMo))
    if torch.isfinite(gMo).all():
        assert _rel(gM, gMo) <= 2e-4 and _rel(gM, gM2) <= 2e-4


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
@pytest.mark.parametrize("case", ["affine_45deg", "affine_wide", "perspective_shift"])
def test_non_finite_gradient_at_a_pixel_no_tile_visits(oracle, case, bad):
    """A pixel that samples entirely outside the source is in no tile's box, yet the reference multiplies its grad_out into the matrix
    gradient (zeros for the out-of-bounds taps times inf / NaN): the matrix gradient of THAT image is not finite, the others' are
    untouched, grad wrt the image stays finite.  One-read backward == two launches == oracle in which entries are finite."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(3)
    if case == "affine_45deg":
        B, C, H, W, h, w = 3, 3, 70, 65, 70, 65
        import math

        c_, s_ = math.cos(math.radians(45.0)), math.sin(math.radians(45.0))
        cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
        M = torch.tensor([[c_, s_, (1 - c_) * cx - s_ * cy], [-s_, c_, s_ * cx + (1 - c_) * cy]]).repeat(B, 1, 1)
        fn = lambda a, m: K.warp_affine(a, m, (h, w))
        bwd = lambda go_, x_, M_: oracle.warp_affine_backward(go_, x_, M_, (h, w))
        hit = (2, slice(None), h - 1, w - 1)  # a corner of the output: rotated out of the source
    elif case == "affine_wide":
        B, C, H, W, h, w = 2, 1, 3, 130, 9, 70
        M = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]).repeat(B, 1, 1)
        fn = lambda a, m: K.warp_affine(a, m, (h, w))
        bwd = lambda go_, x_, M_: oracle.warp_affine_backward(go_, x_, M_, (h, w))
        hit = (1, slice(None), h - 1, 5)      # rows 3.. of the output are below the 3-row source
    else:
        B, C, H, W, h, w = 2, 3, 96, 160, 96, 160
        M = flagship_homographies(B, H, W, h, w, g, jitter=3.0)
        M[:, 0, 2] += 40.0                     # 40 px to the right: the left columns of the output sample outside
        fn = lambda a, m: K.warp_perspective(a, m, (h, w))
        bwd = lambda go_, x_, M_: oracle.warp_perspective_backward(go_, x_, M_, (h, w))
        hit = (0, slice(None), 50, 2)
    x = torch.rand(B, C, H, W, generator=g)
    go = torch.rand(B, C, h, w, generator=g) - 0.5
    go[hit] = bad
    gx, gM = _run(fn, x, M, go, True)
    gx2, gM2 = _run(fn, x, M, go, False)
    gxo, gMo = bwd(go, x, M)
    b = hit[0]
    assert not torch.isfinite(gMo[b]).any() or not torch.isfinite(gMo[b]).all(), "the case must make the oracle's matrix gradient non-finite"
    for name, m_ in (("one-read", gM), ("two launches", gM2)):
        assert torch.equal(torch.isfinite(m_), torch.isfinite(gMo)), f"{name}: finite entries differ from the oracle's\n{m_}\n{gMo}"
    others = [i for i in range(B) if i != b]
    assert _rel(gM[others], gMo[others]) <= 5e-5
    assert torch.isfinite(gx).all() and torch.allclose(gx, gxo, atol=1e-5, rtol=0)  # the pixel touches no source pixel


@pytest.mark.parametrize("fused", [True, False])
def test_non_finite_gradient_outside_the_source_reference_fixture(fused):
    """The same against the reference's own result (tests/golden/nonfinite_outside.npz, oracle/make_golden.py)."""
    import kornia_amd as K
    from _util import golden

    d = {k: torch.from_numpy(v) for k, v in golden("nonfinite_outside").items()}
    for tag in ("nan", "inf"):
        gx, gM = _run(lambda a, m: K.warp_affine(a, m, (70, 65)), d["affine__x"], d["affine__M"], d[f"affine_{tag}__go"], fused)
        assert torch.equal(torch.isfinite(gM), torch.isfinite(d[f"affine_{tag}__gM"])), gM
        assert torch.isfinite(gx).all() and torch.allclose(gx, d[f"affine_{tag}__gx"], atol=1e-5)
    gx, gM = _run(lambda a, m: K.warp_perspective(a, m, (96, 160)), d["persp__x"], d["persp__M"], d["persp__go"], fused)
    assert torch.equal(torch.isfinite(gM), torch.isfinite(d["persp__gM"])), gM
    assert torch.isfinite(gx).all() and torch.allclose(gx, d["persp__gx"], atol=1e-5)


@pytest.mark.parametrize("pad", ["zeros", "fill"])
def test_source_tile_staging_does_not_depend_on_the_address_or_the_tile_shape(oracle, pad):
    """The persistent loop stages the next tile's source tile by LDS-DMA: 16-byte pieces for full-width tiles of a 16-byte aligned image,
    4-byte pieces (a tile row per wave instruction) for ragged tiles and for an image at an address that is only 4-byte aligned; pad == fill
    subtracts the fill in LDS once the tile has landed.  Same image at both kinds of address, widths with and without a ragged last tile
    column: the image gradient must be the same bits, the matrix gradient the same to the rounding of its fp64 sums - and both the oracle's."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(77)
    for (H, W) in ((128, 192), (96, 160), (70, 100)):   # full tiles only / a ragged tile column / ragged rows and columns
        B, C = 3, 3
        x = torch.rand(B, C, H, W, generator=g)
        M = flagship_homographies(B, H, W, H, W, g, jitter=3.0)
        go = torch.rand(B, C, H, W, generator=g) - 0.5
        kw = dict(padding_mode="fill", fill_value=torch.tensor([0.25, 0.5, 0.75])) if pad == "fill" else {}
        fn = lambda a, m: K.warp_perspective(a, m, (H, W), **kw)
        xd = x.cuda()
        off = torch.empty(x.numel() + 1, device="cuda")[1:].view_as(x)
        off.copy_(xd)
        assert off.data_ptr() % 16 != 0 and xd.data_ptr() % 16 == 0
        outs = []
        for src in (xd, off):
            xg, Mg = src.detach().requires_grad_(), M.cuda().requires_grad_()
            fn(xg, Mg).backward(go.cuda())
            outs.append((xg.grad.cpu(), Mg.grad.cpu()))
        assert torch.equal(outs[0][0], outs[1][0]), (H, W)
        # (zeros: the same bits; fill: the last bit of single entries moves with the address - 2e-7 of the entry, i.e. <= 1e-5 of the largest one)
        assert _rel(outs[0][1], outs[1][1]) <= (1e-6 if pad == "zeros" else 2e-5)
        fill = kw.get("fill_value")
        gxo, gMo = oracle.warp_perspective_backward(go, x, M, (H, W), "bilinear", pad, True, fill)
        assert torch.allclose(outs[0][0], gxo, atol=1e-5, rtol=0), (H, W)
        assert _rel(outs[0][1], gMo) <= 5e-5


def test_launch_policy_without_the_scan_has_the_accelerator_kernels_semantics():
    """km_config_set("warp_bwd_no_scan", 1) (KM_WARP_BWD_SCAN=0): the one-read backward skips the scan for non-finite gradients at output pixels
    that sample entirely outside the source - ATen's CUDA / HIP grid_sampler backward skips out-of-bounds taps, so there such a gradient leaves
    the matrix gradient FINITE (and equal to the gradient of the same input with that entry zeroed); with the default policy the result is
    the CPU reference's (the fixture above)."""
    import kornia_amd as K
    from _util import golden
    from kornia_amd import _native as N

    d = {k: torch.from_numpy(v) for k, v in golden("nonfinite_outside").items()}
    go = d["persp__go"].clone()
    bad = ~torch.isfinite(go)
    assert bad.any()
    fn = lambda a, m: K.warp_perspective(a, m, (96, 160))
    prev = N.lib().km_config_set(b"warp_bwd_no_scan", 1)
    try:
        gx, gM = _run(fn, d["persp__x"], d["persp__M"], go, True)
    finally:
        N.lib().km_config_set(b"warp_bwd_no_scan", prev)
    gx0, gM0 = _run(fn, d["persp__x"], d["persp__M"], torch.where(bad, torch.zeros_like(go), go), True)
    assert torch.isfinite(gM).all() and torch.isfinite(gx).all()
    assert _rel(gM, gM0) <= 1e-6 and torch.equal(gx, gx0)
    gx1, gM1 = _run(fn, d["persp__x"], d["persp__M"], go, True)  # the default policy again
    assert torch.equal(torch.isfinite(gM1), torch.isfinite(d["persp__gM"]))


@pytest.mark.parametrize("case", ["growing", "zero_top", "nonfinite_bottom", "shrinking"])
@pytest.mark.parametrize("angle", [0.0, 30.0])
def test_fixed_point_scale_follows_the_gradients_of_the_box(oracle, case, angle):
    """The fixed-point scale of a tile comes from the largest |grad_out| of its box - for a box walked in several passes (30 degrees) from
    the passes seen so far, made coarser (accumulators shifted) when a later pass brings larger gradients: gradients that grow by 2^20
    over 64 rows, boxes whose first rows are all zero (the scale must still be free when the first non-zero - and small - gradient
    arrives: a first pass over zeros once pinned it at 2^0), a NaN / inf in the last rows of a box, and the opposite slope."""
    import kornia_amd as K

    g = torch.Generator().manual_seed(23)
    B, C, H, W, h, w = 2, 3, 192, 192, 192, 192
    x = torch.rand(B, C, H, W, generator=g)
    if angle == 0.0:
        M = flagship_homographies(B, H, W, h, w, g, jitter=3.0)
    else:
        th = torch.deg2rad(torch.tensor(angle))
        c, sn = torch.cos(th).item(), torch.sin(th).item()
        R = torch.tensor([[c, -sn, (1 - c) * W / 2 + sn * H / 2], [sn, c, (1 - c) * H / 2 - sn * W / 2], [0.0, 0.0, 1.0]])
        M = R[None].repeat(B, 1, 1)
    go = torch.rand(B, C, h, w, generator=g) - 0.5
    rows = torch.arange(h, dtype=torch.float32)
    if case == "growing":
        go = go * torch.exp2(rows * (20.0 / 64.0))[None, None, :, None]
    elif case == "shrinking":
        go = go * torch.exp2(-rows * (20.0 / 64.0))[None, None, :, None]
    elif case == "zero_top":
        # (rotated: the first pass over the box of the centre tile - rows 52 to ~120 of 52 to 140 - sees zeros only)
        keep = ((rows % 64) >= 40) if angle == 0.0 else (rows >= 125)
        go = go * keep.float()[None, None, :, None] * 1e-6
    else:
        go[0, 1, 60, 50] = float("nan")
        go[1, 2, 125, 130] = float("inf")
    gx, gM = _run(lambda a, m: K.warp_perspective(a, m, (h, w)), x, M, go, True)
    gxo, gMo = oracle.warp_perspective_backward(go, x, M, (h, w))
    if case == "nonfinite_bottom":
        assert torch.equal(torch.isnan(gx), torch.isnan(gxo)) and torch.equal(torch.isinf(gx), torch.isinf(gxo))
        fin = torch.isfinite(gxo)
        assert torch.allclose(gx[fin], gxo[fin], atol=1e-5, rtol=0)
        assert torch.isnan(gM).any()
        return
    if case == "zero_top":
        # every non-zero gradient is ~1e-6: the error bound is relative to THAT, not to 1
        assert (gx - gxo).abs().max().item() <= 4e-6 * go.abs().max().item(), (gx - gxo).abs().max().item()
    else:
        # per 64-row band of the source: within 4e-6 of the largest gradient that can reach the band (the contract of the fixed point)
        reach = 24 if angle == 0.0 else 120
        for y0 in range(0, H, 64):
            band = slice(y0, y0 + 64)
            tol = 4e-6 * max(go[:, :, max(0, y0 - reach):y0 + 64 + reach].abs().max().item(), 1e-30)
            err = (gx[:, :, band] - gxo[:, :, band]).abs().max().item()
            assert err <= tol, (case, y0, err, tol)
    assert _rel(gM, gMo) <= 5e-5


@pytest.mark.parametrize("angle,scale", [(20.0, 1.0), (45.0, 1.0), (0.0, 1.35), (33.0, 0.8)])
def test_boxes_walked_in_several_passes(oracle, angle, scale):
    """Rotations and magnifications whose boxes exceed what a workgroup holds in registers (6144 pixels): the persistent loop walks them
    in passes, the accumulators rescaled when a later pass brings a larger |grad_out| (the gradient grows along the image here, so that
    it happens).  Against the oracle and the two launches."""
    import math

    import kornia_amd as K

    g = torch.Generator().manual_seed(int(angle) + 3)
    B, C, H, W = 2, 3, 200, 264
    x = torch.rand(B, C, H, W, generator=g)
    a = math.radians(angle)
    ca, sa = scale * math.cos(a), scale * math.sin(a)
    cx, cy = (W - 1) / 2, (H - 1) / 2
    M = torch.tensor([[ca, sa, (1 - ca) * cx - sa * cy], [-sa, ca, sa * cx + (1 - ca) * cy], [0.0, 0.0, 1.0]]).repeat(B, 1, 1)
    M[:, 2, 0] = 2e-5  # a little perspective
    ramp = torch.linspace(0.01, 30.0, H).view(1, 1, H, 1)  # later rows carry larger gradients: later passes raise the maximum
    go = (torch.rand(B, C, H, W, generator=g) - 0.5) * ramp
    gx, gM = _run(lambda a_, m: K.warp_perspective(a_, m, (H, W)), x, M, go, True)
    gx2, gM2 = _run(lambda a_, m: K.warp_perspective(a_, m, (H, W)), x, M, go, False)
    gxo, gMo = oracle.warp_perspective_backward(go, x, M, (H, W))
    tol = 2e-6 * go.abs().max().item() * (4.0 if scale > 1.2 else 1.0)  # (magnification: several output pixels per source pixel)
    assert (gx - gxo).abs().max().item() <= tol, (gx - gxo).abs().max()
    assert (gx2 - gxo).abs().max().item() <= 2 * tol
    assert _rel(gM, gMo) <= 5e-5 and _rel(gM, gM2) <= 5e-5


def test_workspace_contract():
    """km_warp2d_bwd_workspace_bytes is 0 for what the one-read backward does not cover; a short or missing workspace falls back to the
    two launches with the same results; the workspace is not needed after the call."""
    import kornia_amd as K
    from kornia_amd import _native as N

    lib = _lib()
    B, C, H, W = 2, 3, 130, 140
    assert lib.km_warp2d_bwd_workspace_bytes(B, C, H, W, H, W, 1, 0, 0) == 80 * B * 3 * 3  # one 80-byte record per 64 x 64 tile
    assert lib.km_warp2d_bwd_workspace_bytes(B, C, H, W, H, W, 2, 0, 0) == 0   # bicubic
    assert lib.km_warp2d_bwd_workspace_bytes(B, C, H, W, H, W, 1, 1, 0) == 80 * B * 3 * 3  # border padding: the same tile owners (fp32 storage)
    assert lib.km_warp2d_bwd_workspace_bytes(B, C, H, W, H, W, 1, 1, 2) == 0   # ... bf16 storage: the generic kernel
    assert lib.km_warp2d_bwd_workspace_bytes(B, 4, H, W, H, W, 1, 0, 0) == 80 * B * 3 * 3  # RGBA = one group of three + one single channel, the larger sequence
    assert lib.km_warp2d_bwd_workspace_bytes(B, 8, H, W, H, W, 1, 0, 0) == 2 * 80 * B * 3 * 3  # two groups of three (+ two singles)
    assert lib.km_warp2d_bwd_workspace_bytes(B, 4, H, W, H, W, 1, 3, 0) == 0   # fill values are RGB / grey
    assert lib.km_warp2d_bwd_workspace_bytes(B, C, H, W, H, W, 1, 0, 1) == 0   # fp64
    g = torch.Generator().manual_seed(4)
    x = torch.rand(B, C, H, W, generator=g).cuda()
    M = flagship_homographies(B, H, W, H, W, g, jitter=3.0)
    go = torch.rand(B, C, H, W, generator=g).cuda()
    m = K.geometry.conversions._ChainFunction.apply(M.cuda(), (H, W), (H, W), True).contiguous()
    stream = N.stream_ptr(x.device)
    need = int(lib.km_warp2d_bwd_workspace_bytes(B, C, H, W, H, W, 1, 0, 0))
    outs = []
    for ws_bytes in (need, need - 16, 0):
        ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8).cuda()
        gsrc = torch.empty(B, C, H, W).cuda()
        # (ABI 2: with a sufficient workspace the first launch zeroes the accumulators itself - hand it garbage there)
        gm = torch.full((B, 9), 123.0, dtype=torch.float64).cuda() if ws_bytes == need else torch.zeros(B, 9, dtype=torch.float64).cuda()
        N.check(lib.km_warp2d_bwd_ws(go.data_ptr(), x.data_ptr(), m.data_ptr(), gsrc.data_ptr(), gm.data_ptr(), B, C, H, W, H, W, B, 0, 1, 1, 0, 1, None, 0,
                                     ws.data_ptr() if ws_bytes else None, ws_bytes, stream), "bwd")
        ws.fill_(255)  # not read after the call
        torch.cuda.synchronize()
        outs.append((gsrc.cpu(), gm.cpu()))
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1])  # short workspace == no workspace (two launches)
    assert torch.allclose(outs[0][0], outs[2][0], atol=5e-6, rtol=0)
    assert _rel(outs[0][1].view(B, 3, 3), outs[2][1].view(B, 3, 3)) <= 5e-5
