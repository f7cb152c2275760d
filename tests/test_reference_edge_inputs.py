"""Build-container only (needs /root/reference and ROCm's clang++): the public entry points on inputs the fixtures do not hold -
strided / channels-last / empty / 64-bit / 16-bit tensors, every error convention the reference's argument checks have, and
NON-FINITE SAMPLING COORDINATES (a singular matrix) - against the live reference, on the host build of the kernels (tests/emu).

The last group pins the one place where the native path is KNOWN to differ from the reference ("Non-finite
sampling coordinates"): the kernels decide tap bounds in floating point, so a NaN / inf coordinate is simply outside the source
(zeros / fill; the oracle restates the same rule), where ATen's CPU sampler multiplies its masked-out (zero) taps by NaN weights
and returns NaN.  The tests below state both sides so the difference cannot move unnoticed."""
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
}


@pytest.mark.parametrize("which", sorted(NONFINITE_MATRICES))
def test_nonfinite_sampling_coordinates_known_difference(env, which):
    """A singular (or NaN-carrying) matrix makes every sampling coordinate NaN.  Bilinear, zeros padding - the hot path:
    the reference (ATen's CPU sampler) returns NaN everywhere; the native kernels, like the oracle, treat the pixel as outside
    the source and return the padding value.  Neither raises.  Finite matrices in the same batch are unaffected (second half)."""
    K, AT, AF = env
    import oracle as O

    x, M_ok = _inputs()
    M = NONFINITE_MATRICES[which]()
    ref = K.geometry.warp_perspective(x, M, (11, 13))
    nat = AT.warp_perspective(x.cuda(), M.cuda(), (11, 13))
    orc = O.warp_perspective(x, M, (11, 13))
    assert ref.isnan().all()  # the reference's side of the difference
    assert torch.equal(nat, torch.zeros_like(nat)) and torch.equal(orc, nat)  # the native side = the oracle's
    nat_fill = AT.warp_perspective(x.cuda(), M.cuda(), (11, 13), padding_mode="fill", fill_value=torch.tensor([0.1, 0.2, 0.3]))
    assert torch.equal(nat_fill, torch.tensor([0.1, 0.2, 0.3]).view(1, 3, 1, 1).expand_as(nat_fill))
    # bicubic has no bounds decision to make on the weights: NaN on both sides
    assert AT.warp_perspective(x.cuda(), M.cuda(), (11, 13), mode="bicubic").isnan().all()
    assert K.geometry.warp_perspective(x, M, (11, 13), mode="bicubic").isnan().all()
    # one bad sample does not touch its neighbour in the batch
    mixed = torch.stack([M[0], M_ok[1]])
    nat_mixed = AT.warp_perspective(x.cuda(), mixed.cuda(), (11, 13))
    assert torch.equal(nat_mixed[1], K.geometry.warp_perspective(x, M_ok, (11, 13))[1])
    assert torch.equal(nat_mixed[0], torch.zeros_like(nat_mixed[0]))


def test_nonfinite_coordinates_gradients_stay_finite_on_the_native_path(env):
    """Backward of the same case: the native image gradient is zero for a sample whose coordinates are NaN (no tap is inside),
    finite for its neighbour; the reference's is NaN for the bad sample (NaN weights times the upstream gradient)."""
    K, AT, AF = env
    x, M_ok = _inputs()
    mixed = torch.stack([torch.zeros(3, 3), M_ok[1]])
    xr = x.clone().requires_grad_(True)
    K.geometry.warp_perspective(xr, mixed, (11, 13)).sum().backward()
    xa = x.cuda().requires_grad_(True)
    AT.warp_perspective(xa, mixed.cuda(), (11, 13)).sum().backward()
    assert torch.equal(xa.grad[0], torch.zeros_like(xa.grad[0]))
    assert torch.allclose(xa.grad[1], xr.grad[1], atol=1e-5, rtol=0)
    assert not torch.isfinite(xr.grad[0]).all() or torch.equal(xr.grad[0], torch.zeros_like(xr.grad[0]))


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
