"""affwarp / crop2d helpers (rotate, translate, scale, shear, affine, crop_and_resize, center_crop, crop_by_boxes,
crop_by_transform_mat) against fixtures produced by the real reference (tests/golden/warp_callers.npz).

CPU: the helpers' host logic (matrix construction, broadcasting, reparametrisation, sizes) with the native warps swapped
for the oracle's CPU warps - test infrastructure standing in for the device op.  GPU: the real thing."""
import pytest
import torch

from _util import golden


def _t(d, k, dev="cpu"):
    return torch.from_numpy(d[k]).to(dev)


def _run_all(T, d, dev):
    x, ang, trn, scl, shr, A, boxes = (_t(d, k, dev) for k in ("x", "angle", "trans", "scale_in", "shear_in", "A", "boxes"))
    out = {}
    out["rotate"] = T.rotate(x, ang)
    out["rotate_center_nearest"] = T.rotate(x, ang, center=torch.tensor([[10.0, 5.0]], device=dev).expand(3, -1), mode="nearest", padding_mode="border")
    out["translate"] = T.translate(x, trn)
    out["scale"] = T.scale(x, scl)
    out["scale_iso"] = T.scale(x, torch.tensor([1.3], device=dev))
    out["shear"] = T.shear(x, shr)
    out["affine"] = T.affine(x, A)
    out["affine_unbatched"] = T.affine(x[0], A[:1])
    out["crop_and_resize"] = T.crop_and_resize(x, boxes, (9, 14))
    out["crop_and_resize_ac0"] = T.crop_and_resize(x, boxes, (9, 14), align_corners=False)
    out["center_crop"] = T.center_crop(x, (10, 16))
    out["center_crop_odd_nearest"] = T.center_crop(x, (7, 9), mode="nearest")
    dstb = torch.tensor([[[0.0, 0.0], [11.0, 0.0], [11.0, 7.0], [0.0, 7.0]]], device=dev).expand(3, -1, -1)
    out["crop_by_boxes"] = T.crop_by_boxes(x, boxes, dstb)
    out["crop_by_transform_mat"] = T.crop_by_transform_mat(x, T.get_perspective_transform(boxes, dstb), (8, 12))
    out["crop_by_transform_mat_affine"] = T.crop_by_transform_mat(x, A, (8, 12), align_corners=False)
    return out


def _compare(out, d):
    for k, v in out.items():
        ref = _t(d, k)
        assert v.shape == ref.shape, k
        # the matrices go through sin / cos and a few roundings that differ from the reference's matmul chain: positions
        # move by ~1e-6 px; nearest-mode outputs may flip a handful of pixels that sit exactly between two sources
        if "nearest" in k:
            assert (v.cpu() != ref).float().mean() < 0.02, k
        else:
            assert torch.allclose(v.cpu(), ref, atol=2e-5, rtol=1e-5), (k, (v.cpu() - ref).abs().max())


def test_host_logic_with_oracle_warps(oracle, monkeypatch):
    import kornia_amd as K
    from kornia_amd.geometry.transform import affwarp, crop2d

    def wa(src, M, dsize, mode="bilinear", padding_mode="zeros", align_corners=True, fill_value=None):
        return oracle.warp_affine(src.contiguous(), M, dsize, mode, padding_mode, align_corners)

    def wp(src, M, dsize, mode="bilinear", padding_mode="zeros", align_corners=True, fill_value=None):
        return oracle.warp_perspective(src.contiguous(), M, dsize, mode, padding_mode, align_corners)

    monkeypatch.setattr(affwarp, "warp_affine", wa)
    monkeypatch.setattr(crop2d, "warp_affine", wa)
    monkeypatch.setattr(crop2d, "warp_perspective", wp)
    d = golden("warp_callers")
    _compare(_run_all(K.geometry.transform, d, "cpu"), d)
    T = K.geometry.transform
    with pytest.raises(TypeError):
        T.rotate(torch.rand(1, 1, 4, 4), 30.0)
    with pytest.raises(ValueError):
        T.translate(torch.rand(4, 4), torch.zeros(1, 2))
    with pytest.raises(ValueError):
        T.center_crop(torch.rand(1, 1, 8, 8), 4)
    with pytest.raises(AssertionError):
        T.crop_and_resize(torch.rand(1, 8, 8), torch.zeros(1, 4, 2), (4, 4))


@pytest.mark.gpu
def test_on_device_against_reference_fixtures():
    import kornia_amd as K

    d = golden("warp_callers")
    _compare(_run_all(K.geometry.transform, d, "cuda"), d)


def _run_filter_callers(F, x):
    return {"box3": F.box_blur(x, 3), "box35_sep_replicate": F.box_blur(x, (3, 5), "replicate", separable=True),
            "box57_constant": F.box_blur(x, (5, 7), "constant"), "lap3": F.laplacian(x, 3),
            "lap5_unnorm_circular": F.Laplacian(5, "circular", normalized=False)(x), "unsharp": F.unsharp_mask(x, (5, 5), (1.5, 1.5))}


def test_filter_callers_host_logic(oracle, monkeypatch):
    """box_blur / laplacian: tap construction + dispatch, with the oracle's filter2d standing in for the device op."""
    import kornia_amd as K
    from kornia_amd.filters import blur

    monkeypatch.setattr(blur, "filter2d", lambda x, k, border="reflect": oracle.filter2d(x, k, border))
    monkeypatch.setattr(blur, "filter2d_separable", lambda x, kx, ky, border="reflect": oracle.filter2d_separable(x, kx, ky, border))
    import sys

    gmod = sys.modules["kornia_amd.filters.gaussian"]  # (the package attribute `gaussian` is the kernel function)
    monkeypatch.setattr(gmod, "gaussian_blur2d", lambda x, ks, sg, border="reflect", separable=True: oracle.gaussian_blur2d(x, ks, sg, border))
    d = golden("filter_callers")
    for k, v in _run_filter_callers(K.filters, _t(d, "x")).items():
        assert torch.equal(v, _t(d, k)), k
    assert torch.equal(K.filters.get_laplacian_kernel1d(5), torch.tensor([1.0, 1.0, -4.0, 1.0, 1.0]))
    assert K.filters.get_box_kernel2d((3, 5)).shape == (1, 3, 5)


@pytest.mark.gpu
def test_filter_callers_on_device():
    import kornia_amd as K

    d = golden("filter_callers")
    for k, v in _run_filter_callers(K.filters, _t(d, "x", "cuda")).items():
        if k == "unsharp":  # the lerp epilogue is PyTorch's device kernel (fused multiply-add contraction differs from the host's)
            assert torch.allclose(v.cpu(), _t(d, k), atol=1e-6, rtol=0), k
        else:
            assert torch.equal(v.cpu(), _t(d, k)), k
