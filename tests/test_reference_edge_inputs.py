"""Build-container only (needs /root/reference and ROCm's clang++): the public entry points on inputs the fixtures do not hold -
strided / channels-last / empty / 64-bit / 16-bit tensors, every error convention the reference's argument checks have, and
NON-FINITE SAMPLING COORDINATES (a singular matrix) - against the live reference, on the host build of the kernels (tests/emu).

The last group was, until round 6, the one place where the native path was KNOWN to differ from the reference (DESIGN.md section 2,
"Non-finite sampling coordinates"): a NaN / inf coordinate was simply outside the source (zeros / fill), where ATen's CPU sampler
multiplies its masked-out (zero) taps by NaN weights and returns NaN.  Kernels and oracle now follow the reference; the tests
below assert EQUALITY with the live reference, forward and backward."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "emu")]
import ref_shim  # noqa: E402

pytestmark = [
    pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present (GPU box)"),
    pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="host build of the kernels needs ROCm's clang++"),
]


@pytest.fixture(scope="module")
def env():
    K = ref_shim.import_reference()
    from mode import emulated_device

    import kornia_amd.filters as AF
    import kornia_amd.geometry.transform as AT

    with emulated_device():
        yield K, AT, AF


def _inputs():
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, 3, 17, 23, generator=g)
    M = torch.eye(3).repeat(2, 1, 1)
    M[:, 0, 2] = 1.5
    M[:, 1, 0] = 0.07
    M[:, 2, 0] = 1e-3
    return x, M


def _both(fr, fa):
    """(result, exception) of the reference call and of the native call"""
    out = []
    for f in (fr, fa):
        try:
            out.append((f(), None))
        except Exception as e:  # noqa: BLE001 - the exception type is what is compared
            out.append((None, e))
    return out


@pytest.mark.parametrize("layout", ["contiguous", "transposed", "channels_last", "sliced", "expanded_batch"])
def test_warps_and_blur_on_strided_inputs_are_bit_identical(env, layout):
    K, AT, AF = env
    x, M = _inputs()
    if layout == "transposed":
        x = torch.rand(2, 3, 23, 17, generator=torch.Generator().manual_seed(2)).transpose(2, 3)
    elif layout == "channels_last":
        x = x.contiguous(memory_format=torch.channels_last)
    elif layout == "sliced":
        x = torch.rand(2, 3, 34, 46, generator=torch.Generator().manual_seed(3))[:, :, ::2, ::2]
    elif layout == "expanded_batch":
        x = x[:1].expand(2, -1, -1, -1)
    xd, Md = x.cuda(), M.cuda()
    assert torch.equal(AT.warp_perspective(xd, Md, (11, 13)), K.geometry.warp_perspective(x, M, (11, 13)))
    assert torch.equal(AT.warp_affine(xd, Md[:, :2], (19, 9)), K.geometry.warp_affine(x, M[:, :2], (19, 9)))
    assert torch.equal(AF.gaussian_blur2d(xd, (5, 5), (1.5, 1.5)), K.filters.gaussian_blur2d(x, (5, 5), (1.5, 1.5)))
    assert torch.equal(AF.spatial_gradient(xd), K.filters.spatial_gradient(x))


def test_empty_batch(env):
    K, AT, AF = env
    x, M = _inputs()
    for fa, fr in (
        (lambda: AT.warp_perspective(x[:0].cuda(), M[:0].cuda(), (11, 13)), lambda: K.geometry.warp_perspective(x[:0], M[:0], (11, 13))),
        (lambda: AF.gaussian_blur2d(x[:0].cuda(), (5, 5), (1.5, 1.5)), lambda: K.filters.gaussian_blur2d(x[:0], (5, 5), (1.5, 1.5))),
        (lambda: AF.sobel(x[:0].cuda()), lambda: K.filters.sobel(x[:0])),
    ):
        a, r = fa(), fr()
        assert a.shape == r.shape and a.dtype == r.dtype and a.numel() == 0


def test_float64_is_bit_identical_and_float16_is_the_fp32_result_rounded(env):
    K, AT, AF = env
    x, M = _inputs()
    assert torch.equal(AT.warp_perspective(x.double().cuda(), M.double().cuda(), (11, 13)), K.geometry.warp_perspective(x.double(), M.double(), (11, 13)))
    assert torch.equal(AF.gaussian_blur2d(x.double().cuda(), (5, 5), (1.5, 1.5)), K.filters.gaussian_blur2d(x.double(), (5, 5), (1.5, 1.5)))
    # 16-bit storage: the kernels compute in fp32 and round once; the reference's CPU path rounds the grid and every
    # intermediate to fp16, so the native result is the closer of the two to the fp32 answer - checked as such, at BASELINE's 1e-2
    xh, Mh = x.half(), M.half()
    exact = K.geometry.warp_perspective(xh.float(), Mh.float(), (11, 13))
    native = AT.warp_perspective(xh.cuda(), Mh.cuda(), (11, 13))
    assert native.dtype == torch.float16
    assert torch.equal(native, exact.half())
    assert (K.geometry.warp_perspective(xh, Mh, (11, 13)).float() - exact).abs().max() <= 1e-2


@pytest.mark.parametrize(
    "case",
    ["src_3dim", "matrix_batch_1", "bad_mode", "bad_padding", "even_kernel", "image_smaller_than_the_border", "affine_given_3x3", "integer_image"],
)
def test_error_conventions_are_the_references(env, case):
    K, AT, AF = env
    x, M = _inputs()
    xd, Md = x.cuda(), M.cuda()
    calls = {
        "src_3dim": (lambda: K.geometry.warp_perspective(x[0], M, (11, 13)), lambda: AT.warp_perspective(xd[0], Md, (11, 13))),
        "matrix_batch_1": (lambda: K.geometry.warp_perspective(x, M[:1], (11, 13)), lambda: AT.warp_perspective(xd, Md[:1], (11, 13))),
        "bad_mode": (lambda: K.geometry.warp_perspective(x, M, (11, 13), mode="cubic"), lambda: AT.warp_perspective(xd, Md, (11, 13), mode="cubic")),
        "bad_padding": (lambda: K.geometry.warp_affine(x, M[:, :2], (11, 13), padding_mode="wrap"), lambda: AT.warp_affine(xd, Md[:, :2], (11, 13), padding_mode="wrap")),
        "even_kernel": (lambda: K.filters.gaussian_blur2d(x, (4, 4), (1.5, 1.5)), lambda: AF.gaussian_blur2d(xd, (4, 4), (1.5, 1.5))),
        "image_smaller_than_the_border": (
            lambda: K.filters.gaussian_blur2d(x[..., :2, :2], (5, 5), (1.5, 1.5)),
            lambda: AF.gaussian_blur2d(xd[..., :2, :2], (5, 5), (1.5, 1.5)),
        ),
        "affine_given_3x3": (lambda: K.geometry.warp_affine(x, M, (11, 13)), lambda: AT.warp_affine(xd, Md, (11, 13))),
        "integer_image": (lambda: K.geometry.warp_perspective((x * 255).byte(), M, (11, 13)), lambda: AT.warp_perspective((xd * 255).byte(), Md, (11, 13))),
    }
    (r, re), (a, ae) = _both(*calls[case])
    assert re is not None and ae is not None, (case, re, ae)
    if case == "integer_image":
        # the reference fails inside ATen ("grid_sampler_2d_cpu_kernel_impl" not implemented for 'Byte'); the native path refuses the dtype up front
        assert isinstance(re, (NotImplementedError, RuntimeError)) and isinstance(ae, TypeError)
    else:
        # by name: the package re-uses Kornia's exception classes only when Kornia was imported first (kornia_amd/core/exceptions.py)
        assert type(ae).__name__ == type(re).__name__ and isinstance(ae, Exception), (case, re, ae)


NONFINITE_MATRICES = {
    "all_zero": lambda: torch.zeros(2, 3, 3),  # the closed-form inverse divides by a zero determinant: every coordinate NaN
    "nan_entry": lambda: torch.eye(3).repeat(2, 1, 1).index_put((torch.tensor([0, 1]), torch.tensor([0, 0]), torch.tensor([2, 2])), torch.tensor(float("nan"))),
    "inf_entry": lambda: torch.eye(3).repeat(2, 1, 1).index_put((torch.tensor([0, 1]), torch.tensor([1, 1]), torch.tensor([1, 1])), torch.tensor(float("inf"))),
    # denominator 0.25 (x - 11): exactly zero on the centre column of a 23-wide output, finite elsewhere
    "zero_denominator_column": lambda: torch.linalg.inv(torch.tensor([[1.0, 0, 0], [0, 1, 0], [0.25, 0, -0.25 * 11]], dtype=torch.float64)).float().repeat(2, 1, 1),
}


def _same_nan_pattern_and_values(a, r, atol=0.0):
    assert torch.equal(a.isnan(), r.isnan()), (a.isnan() != r.isnan()).sum()
    fin = ~r.isnan()
    if atol:
        assert torch.allclose(a[fin], r[fin], atol=atol, rtol=0)
    else:
        assert torch.equal(a[fin], r[fin])


@pytest.mark.parametrize("which", sorted(NONFINITE_MATRICES))
def test_nonfinite_sampling_coordinates_are_the_references(env, which):
    """A singular (or NaN / inf carrying) matrix, or a projective denominator that vanishes on a column of the output, makes sampling
    coordinates NaN / inf.  The reference (ATen's CPU sampler) gathers the taps of such a pixel as zeros and still multiplies them by the NaN
    weights: NaN for bilinear (zeros and fill padding) and bicubic, 0 for nearest.  Rounds 1-5 of the native path (and of the oracle) treated
    the pixel as outside the source - zeros / the fill colour - which silently masked a diverged homography; since round 6 kernels and oracle
    follow the reference (csrc/km_sampler.h km_bilinear_masked, oracle/ko_impl.h): equality against the LIVE reference, pixel for pixel."""
    K, AT, AF = env
    import oracle as O

    x, M_ok = _inputs()
    M = NONFINITE_MATRICES[which]()
    size = (11, 23)
    ref = K.geometry.warp_perspective(x, M, size)
    assert ref.isnan().any()  # the case is one
    _same_nan_pattern_and_values(AT.warp_perspective(x.cuda(), M.cuda(), size), ref)
    _same_nan_pattern_and_values(O.warp_perspective(x, M, size), ref)
    fill = torch.tensor([0.1, 0.2, 0.3])
    _same_nan_pattern_and_values(AT.warp_perspective(x.cuda(), M.cuda(), size, padding_mode="fill", fill_value=fill),
                                 K.geometry.warp_perspective(x, M, size, padding_mode="fill", fill_value=fill))
    _same_nan_pattern_and_values(AT.warp_perspective(x.cuda(), M.cuda(), size, mode="bicubic"), K.geometry.warp_perspective(x, M, size, mode="bicubic"), atol=1e-6)
    _same_nan_pattern_and_values(AT.warp_perspective(x.cuda(), M.cuda(), size, mode="nearest"), K.geometry.warp_perspective(x, M, size, mode="nearest"))
    _same_nan_pattern_and_values(AT.warp_affine(x.cuda(), M[:, :2].cuda(), size), K.geometry.warp_affine(x, M[:, :2], size))
    # one bad sample does not touch its neighbour in the batch
    mixed = torch.stack([M[0], M_ok[1]])
    _same_nan_pattern_and_values(AT.warp_perspective(x.cuda(), mixed.cuda(), size), K.geometry.warp_perspective(x, mixed, size))


@pytest.mark.parametrize("which", sorted(NONFINITE_MATRICES))
def test_nonfinite_sampling_coordinates_gradients_are_the_references(env, which):
    """Backward of the same cases against the live reference: a pixel with a non-finite position scatters nothing into the image gradient
    (ATen's scatter is masked), and the matrix gradient of its sample is NaN in the reference's entries; the neighbour in the batch is
    untouched."""
    K, AT, AF = env
    x, M_ok = _inputs()
    mixed = torch.stack([NONFINITE_MATRICES[which]()[0], M_ok[1]])
    go = torch.rand(2, 3, 11, 23, generator=torch.Generator().manual_seed(5)) - 0.5
    xr, Mr = x.clone().requires_grad_(True), mixed.clone().requires_grad_(True)
    K.geometry.warp_perspective(xr, Mr, (11, 23)).backward(go)
    xa, Ma = x.cuda().requires_grad_(True), mixed.cuda().requires_grad_(True)
    AT.warp_perspective(xa, Ma, (11, 23)).backward(go.cuda())
    assert torch.isfinite(xr.grad).all() and torch.allclose(xa.grad, xr.grad, atol=1e-5, rtol=0)
    assert torch.equal(torch.isfinite(Ma.grad), torch.isfinite(Mr.grad)), (Ma.grad, Mr.grad)
    assert not torch.isfinite(Mr.grad[0]).any() and torch.isfinite(Mr.grad[1]).all()
    assert ((Ma.grad[1] - Mr.grad[1]).abs().max() / Mr.grad[1].abs().max()).item() < 5e-3


def _mirrored_pairs(K):
    """(reference callable, native callable) for every public name the package mirrors: the drop-in claim of SURVEY 8(b) is 'same
    names, argument meaning and error behaviour', so each pair must have the same parameter names, order, kinds and defaults."""
    import kornia_amd as A
    import kornia_amd.enhance as AE
    import kornia_amd.filters as AF
    import kornia_amd.geometry as AG
    import kornia_amd.geometry.transform as AT

    pairs = []
    for ref_mod, nat_mod, names in (
        (K.geometry.transform, AT, [n for n in AT.__all__ if n not in ("warp_affine_blur", "warp_perspective_blur", "masked_warp_loss", "resize_bilinear", "grid_sample")]),
        (K.filters, AF, [n for n in dir(AF) if not n.startswith("_") and n not in ("blur", "filter", "gaussian", "kernels", "canny", "sobel") or n in ("canny", "sobel")]),
        (K.geometry, AG, ["transform_points", "normalize_homography", "normal_transform_pixel", "convert_affinematrix_to_homography", "create_meshgrid",
                          "normalize_pixel_coordinates", "convert_points_from_homogeneous", "convert_points_to_homogeneous"]),
        (K.enhance, AE, ["adjust_hue"]),
        (K.enhance.adjust, AE, ["adjust_brightness_accumulative", "adjust_contrast_with_mean_subtraction", "adjust_saturation_with_gray_subtraction"]),
    ):
        for n in names:
            r, a = getattr(ref_mod, n, None), getattr(nat_mod, n, None)
            if r is None or a is None or isinstance(a, type(os)):
                continue
            if isinstance(a, type):
                pairs.append((f"{n}.__init__", r.__init__, a.__init__))
                if "forward" in a.__dict__ and hasattr(r, "forward"):
                    pairs.append((f"{n}.forward", r.forward, a.forward))
            elif callable(a):
                pairs.append((n, r, a))
    return pairs


def test_every_mirrored_name_has_the_references_signature():
    import inspect

    K = ref_shim.import_reference()
    pairs = _mirrored_pairs(K)
    assert len(pairs) >= 70, len(pairs)
    wrong = []
    for name, r, a in pairs:
        a = getattr(a, "__wrapped__", a) if not hasattr(a, "__signature__") else a
        pr, pa = inspect.signature(r).parameters, inspect.signature(a).parameters
        same = list(pr) == list(pa) and all(pr[k].kind == pa[k].kind and repr(pr[k].default) == repr(pa[k].default) for k in pr)
        if not same:
            wrong.append((name, str(inspect.signature(r)), str(inspect.signature(a))))
    assert not wrong, wrong
