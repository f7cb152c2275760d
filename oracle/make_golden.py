"""TEST INFRASTRUCTURE ONLY - generates tests/golden/*.npz by running the REAL reference
(kornia @ /root/reference, imported through oracle/ref_shim.py) on CPU in the build container.

    python oracle/make_golden.py            # rewrites every fixture (deterministic: seeded inputs)

Each .npz stores the inputs AND the reference's outputs / autograd gradients, so the tests that
consume them (tests/test_oracle_golden.py on CPU, tests/test_gpu_golden.py on the MI355X) need
neither the reference tree nor the same RNG.  Sizes are small on purpose (whole directory < 3 MB).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ref_shim  # noqa: E402

K = ref_shim.import_reference()
from _util import flagship_homographies, rotation_affines  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
MODES = ["bilinear", "nearest", "bicubic"]
PADS = ["zeros", "border", "reflection", "fill"]
BORDERS = ["constant", "reflect", "replicate", "circular"]


ONLY = set(sys.argv[1:])  # `python oracle/make_golden.py pyramid` rewrites just that fixture (inputs are still drawn in order)


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f"wrote {name}.npz ({os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024:.0f} KiB)")


def main():
    T = K.geometry.transform
    F = K.filters
    g = torch.Generator().manual_seed(2024)

    # ---- 3x3 chain ---------------------------------------------------------------------------
    from kornia.core.utils import _inverse_3x3_closed_form
    from kornia.geometry.conversions import normalize_homography

    M = flagship_homographies(32, 512, 512, 384, 640, g)
    A = normalize_homography(M, (512, 512), (384, 640))
    save("chain", M=M, A=A, m=_inverse_3x3_closed_form(A), src_size=[512, 512], dst_size=[384, 640])

    # ---- warps ---------------------------------------------------------------------------------
    B, C, H, W, h, w = 2, 3, 24, 32, 20, 28
    x = torch.rand(B, C, H, W, generator=g)
    Mp = flagship_homographies(B, H, W, h, w, g, jitter=3.0)
    Aa = rotation_affines(B, H, W, g)
    Hn = torch.eye(3)[None] + 0.05 * torch.randn(B, 3, 3, generator=g)
    go = torch.rand(B, C, h, w, generator=g)
    fill = torch.tensor([0.1, 0.5, 0.9])
    d = dict(x=x, Mp=Mp, Aa=Aa, Hn=Hn, go=go, fill=fill, dsize=[h, w])
    for mode in MODES:
        for pad in PADS:
            for ac in (True, False):
                tag = f"{mode}_{pad}_{int(ac)}"
                fv = fill if pad == "fill" else None
                xg, Mg = x.clone().requires_grad_(), Mp.clone().requires_grad_()
                y = T.warp_perspective(xg, Mg, (h, w), mode, pad, ac, fv)
                d["persp_" + tag] = y
                if mode != "nearest":
                    y.backward(go)
                    d["persp_gx_" + tag], d["persp_gM_" + tag] = xg.grad, Mg.grad
                xg, Ag = x.clone().requires_grad_(), Aa.clone().requires_grad_()
                y = T.warp_affine(xg, Ag, (h, w), mode, pad, ac, fv)
                d["affine_" + tag] = y
                if mode == "bilinear":
                    y.backward(go)
                    d["affine_gx_" + tag], d["affine_gM_" + tag] = xg.grad, Ag.grad
                if pad != "fill":
                    xg, Hg = x.clone().requires_grad_(), Hn.clone().requires_grad_()
                    y = T.homography_warp(xg, Hg, (h, w), mode, pad, ac)
                    d["homog_" + tag] = y
                    if mode == "bilinear":
                        y.backward(go)
                        d["homog_gx_" + tag], d["homog_gH_" + tag] = xg.grad, Hg.grad
    d["affine_shared"] = T.warp_affine(x, Aa[:1], (h, w))
    save("warps", **d)

    # fp64 gradients wrt the matrices for the bilinear cases (the loose-tolerance truth)
    d64 = {}
    for pad in PADS:
        for ac in (True, False):
            tag = f"bilinear_{pad}_{int(ac)}"
            fv = fill.double() if pad == "fill" else None
            xg, Mg = x.double().requires_grad_(), Mp.double().requires_grad_()
            T.warp_perspective(xg, Mg, (h, w), "bilinear", pad, ac, fv).backward(go.double())
            d64["persp_gM_" + tag] = Mg.grad
    save("warps_f64", **d64)

    # ---- known-answer literals of the reference's own tests (re-evaluated, not retyped) -------------
    lit = {}
    img = torch.arange(12.0).view(1, 1, 3, 4)
    aff = torch.eye(2, 3)[None].clone()
    aff[..., -1] += 1.0
    lit["affine_translation_in"], lit["affine_translation_M"] = img, aff
    lit["affine_translation_out"] = T.warp_affine(img, aff, (3, 4))  # tests/geometry/transform/test_imgwarp.py:251-264
    x46 = torch.arange(24.0).view(1, 1, 4, 6)
    H46 = torch.tensor([[[1.0, 0.0, 1.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]])
    lit["persp_4x6_in"], lit["persp_4x6_H"] = x46, H46
    lit["persp_4x6_out"] = T.warp_perspective(x46, H46, (3, 5), align_corners=True)  # test_imgwarp.py:416-440
    p45 = torch.arange(20.0).view(1, 1, 4, 5)
    lit["hw_4x5_in"] = p45
    lit["hw_4x5_out"] = T.HomographyWarper(4, 5)(p45, torch.eye(3)[None])  # test_homography_warper.py:181-221
    save("literals", **lit)

    # ---- filters --------------------------------------------------------------------------------
    xf = torch.rand(4, 3, 18, 24, generator=g)
    df = dict(x=xf)
    kshapes = [(1, 3, 3), (1, 5, 6), (1, 2, 2), (4, 3, 5), (2, 4, 3)]
    for ki, ks in enumerate(kshapes):
        k = torch.rand(*ks, generator=g)
        df[f"k{ki}"] = k
        for border in BORDERS:
            for padding in ("same", "valid"):
                for beh in ("corr", "conv"):
                    xg, kg = xf.clone().requires_grad_(), k.clone().requires_grad_()
                    y = F.filter2d(xg, kg, border, False, padding, beh)
                    tag = f"{ki}_{border}_{padding}_{beh}"
                    df["f2d_" + tag] = y
                    if beh == "corr":
                        gy = torch.rand(y.shape, generator=g)
                        y.backward(gy)
                        df["f2d_gy_" + tag], df["f2d_gx_" + tag], df["f2d_gk_" + tag] = gy, xg.grad, kg.grad
    save("filter2d", **df)

    dg = {}
    x1 = torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(0))  # BASELINE.json configs[0]
    dg["cfg1_x"] = x1
    dg["cfg1_y"] = F.GaussianBlur2d((5, 5), (1.5, 1.5))(x1)
    dg["cfg1_kernel"] = F.get_gaussian_kernel1d(5, 1.5)
    sig = torch.rand(4, 2, generator=g) + 0.5
    sig2 = torch.rand(2, 2, generator=g) + 0.5
    dg["x"], dg["sig"], dg["sig2"] = xf, sig, sig2
    for border in BORDERS:
        xg = xf.clone().requires_grad_()
        y = F.gaussian_blur2d(xg, (5, 5), (1.5, 1.5), border)
        gy = torch.rand(y.shape, generator=g)
        y.backward(gy)
        dg[f"g5_{border}"], dg[f"g5_gy_{border}"], dg[f"g5_gx_{border}"] = y, gy, xg.grad
        dg[f"g5ns_{border}"] = F.gaussian_blur2d(xf, (5, 5), (1.5, 1.5), border, separable=False)
        dg[f"g37_{border}"] = F.gaussian_blur2d(xf, (3, 7), sig, border)
        dg[f"g33b2_{border}"] = F.gaussian_blur2d(xf, (3, 3), sig2, border)
    for mode in ("sobel", "diff"):
        for order in (1, 2):
            for nrm in (True, False):
                xg = xf.clone().requires_grad_()
                y = F.spatial_gradient(xg, mode, order, nrm)
                gy = torch.rand(y.shape, generator=g)
                y.backward(gy)
                tag = f"{mode}_{order}_{int(nrm)}"
                dg["sg_" + tag], dg["sg_gy_" + tag], dg["sg_gx_" + tag] = y, gy, xg.grad
    dg["sobel"] = F.sobel(xf)
    save("gaussian_sobel", **dg)

    # ---- transform_points ------------------------------------------------------------------------
    from kornia.geometry.linalg import transform_points

    dp = {}
    for D in (2, 3):
        P = torch.rand(3, 200, D, generator=g) * 2 - 1
        Tm = torch.eye(D + 1)[None] + 0.1 * torch.randn(3, D + 1, D + 1, generator=g)
        dp[f"P{D}"], dp[f"T{D}"] = P, Tm
        dp[f"out{D}"] = transform_points(Tm, P)
        dp[f"out{D}_shared"] = transform_points(Tm[:1], P)
    save("transform_points", **dp)

    # ---- the headline pipeline, small ----------------------------------------------------------------
    xh = torch.rand(2, 3, 64, 64, generator=g)
    Mh = flagship_homographies(2, 64, 64, 64, 64, g, jitter=2.0)
    gh = torch.rand(2, 3, 64, 64, generator=g)
    xg, Mg = xh.clone().requires_grad_(), Mh.clone().requires_grad_()
    y = F.gaussian_blur2d(T.warp_perspective(xg, Mg, (64, 64)), (5, 5), (1.5, 1.5))
    y.backward(gh)
    save("headline_small", x=xh, M=Mh, go=gh, y=y, gx=xg.grad, gM=Mg.grad)


    # ---- matrix builders --------------------------------------------------------
    db = {}
    Bb = 16
    base = torch.tensor([[0.0, 0.0], [63.0, 0.0], [63.0, 47.0], [0.0, 47.0]])
    db["ps"] = base[None] + 4.0 * (torch.rand(Bb, 4, 2, generator=g) - 0.5)
    db["pd"] = base[None] + 12.0 * (torch.rand(Bb, 4, 2, generator=g) - 0.5)
    ps, pd = db["ps"].clone().requires_grad_(), db["pd"].clone().requires_grad_()
    Hq = T.get_perspective_transform(ps, pd)
    gH = torch.rand(Bb, 3, 3, generator=g)
    Hq.backward(gH)
    db["H"], db["gH"], db["g_ps"], db["g_pd"] = Hq, gH, ps.grad, pd.grad
    db["H64"] = T.get_perspective_transform(db["ps"].double(), db["pd"].double())
    db["center"] = torch.rand(Bb, 2, generator=g) * 64
    db["angle"] = (torch.rand(Bb, generator=g) - 0.5) * 360
    db["scale"] = 0.5 + torch.rand(Bb, 2, generator=g)
    db["trans"] = (torch.rand(Bb, 2, generator=g) - 0.5) * 20
    db["sx"] = (torch.rand(Bb, generator=g) - 0.5)
    db["sy"] = (torch.rand(Bb, generator=g) - 0.5)
    db["rot"] = T.get_rotation_matrix2d(db["center"], db["angle"], db["scale"])
    db["rot64"] = T.get_rotation_matrix2d(db["center"].double(), db["angle"].double(), db["scale"].double())
    db["aff"] = T.get_affine_matrix2d(db["trans"], db["center"], db["scale"], db["angle"])
    db["aff_shear"] = T.get_affine_matrix2d(db["trans"], db["center"], db["scale"], db["angle"], db["sx"], db["sy"])
    db["aff_sx"] = T.get_affine_matrix2d(db["trans"], db["center"], db["scale"], db["angle"], sx=db["sx"])
    db["shear"] = T.get_shear_matrix2d(db["center"], db["sx"], db["sy"])
    db["transl"] = T.get_translation_matrix2d(db["trans"])
    db["a2r"] = K.geometry.conversions.angle_to_rotation_matrix(db["angle"].reshape(4, 4))
    db["d2r"] = K.geometry.conversions.deg2rad(db["angle"])
    save("builders", **db)


    # ---- explicit-grid sampling: F.grid_sample / remap -----------------------------
    import torch.nn.functional as Fn

    dgs = {}
    xs = torch.rand(2, 3, 13, 20, generator=g)
    grid = torch.rand(2, 9, 12, 2, generator=g) * 2.6 - 1.3
    gos = torch.rand(2, 3, 9, 12, generator=g)
    dgs["x"], dgs["grid"], dgs["go"] = xs, grid, gos
    for mode in MODES:
        for pad in ("zeros", "border", "reflection"):
            for al in (False, True):
                xg, gg = xs.clone().requires_grad_(), grid.clone().requires_grad_()
                y = Fn.grid_sample(xg, gg, mode=mode, padding_mode=pad, align_corners=al)
                y.backward(gos)
                tag = f"{mode}_{pad}_{int(al)}"
                dgs["y_" + tag], dgs["gx_" + tag], dgs["gg_" + tag] = y, xg.grad, gg.grad
    mx = torch.rand(2, 9, 12, generator=g) * 21 - 1
    my = torch.rand(2, 9, 12, generator=g) * 14 - 1
    dgs["map_x"], dgs["map_y"] = mx, my
    dgs["remap"] = T.remap(xs, mx, my)
    dgs["remap_ac"] = T.remap(xs, mx, my, align_corners=True)
    dgs["remap_bcast_nearest"] = T.remap(xs, mx[:1], my[:1], mode="nearest", padding_mode="border")
    dgs["remap_norm"] = T.remap(xs, grid[..., 0], grid[..., 1], normalized_coordinates=True, padding_mode="reflection")
    save("grid_sample", **dgs)


    # ---- ColorJitter arithmetic: the reference's own functions, sequenced as apply_transform does ----
    from kornia.constants import pi as kpi
    from kornia.enhance.adjust import (
        adjust_brightness_accumulative,
        adjust_contrast_with_mean_subtraction,
        adjust_hue,
        adjust_saturation_with_gray_subtraction,
    )

    dc = {}
    xc = torch.rand(4, 3, 20, 28, generator=g)
    xc[0, :, :4, :4] = 0.5  # a grey patch: zero saturation / delta == 0 branch of rgb_to_hsv
    xc[1, :, 0, 0] = torch.tensor([1.0, 0.0, 0.0])
    bf = 0.6 + 0.8 * torch.rand(4, generator=g)
    cf = 0.6 + 0.8 * torch.rand(4, generator=g)
    sf = 0.6 + 0.8 * torch.rand(4, generator=g)
    hf = (torch.rand(4, generator=g) - 0.5) * 0.4  # turns
    dc.update(x=xc, bf=bf, cf=cf, sf=sf, hf=hf)
    dc["brightness"] = adjust_brightness_accumulative(xc, bf)
    dc["contrast"] = adjust_contrast_with_mean_subtraction(xc, cf)
    dc["saturation"] = adjust_saturation_with_gray_subtraction(xc, sf)
    dc["hue"] = adjust_hue(xc, hf * 2 * kpi)
    fns = [lambda im: adjust_brightness_accumulative(im, bf), lambda im: adjust_contrast_with_mean_subtraction(im, cf),
           lambda im: adjust_saturation_with_gray_subtraction(im, sf), lambda im: adjust_hue(im, hf * 2 * kpi)]
    for order in ([0, 1, 2, 3], [3, 2, 1, 0], [2, 0, 3, 1], [1, 3, 0, 2]):
        out = xc
        for i in order:
            out = fns[i](out)
        dc["seq_" + "".join(map(str, order))] = out
    # the module itself with replayed parameters (what config 3 runs)
    aug = K.augmentation.ColorJitter(0.2, 0.2, 0.2, 0.1, p=1.0)
    params = {"brightness_factor": bf, "contrast_factor": cf, "saturation_factor": sf, "hue_factor": hf,
              "order": torch.tensor([2, 0, 3, 1]), "batch_prob": torch.ones(4, dtype=torch.bool), "forward_input_shape": torch.tensor(xc.shape)}
    dc["module_2031"] = aug(xc, params=params)
    save("color_jitter", **dc)


    # ---- callers of the warps: affwarp.py / crop2d.py helpers --------------------------------
    dh = {}
    xh = torch.rand(3, 2, 21, 30, generator=g)
    ang = torch.tensor([12.0, -35.0, 80.0])
    trn = torch.tensor([[2.5, -1.0], [0.0, 3.25], [-4.0, 1.5]])
    scl = torch.tensor([[1.2, 0.8], [0.7, 0.7], [1.0, 1.5]])
    shr = torch.tensor([[0.2, 0.0], [0.0, -0.3], [0.15, 0.1]])
    Aff = rotation_affines(3, 21, 30, g)
    dh.update(x=xh, angle=ang, trans=trn, scale_in=scl, shear_in=shr, A=Aff)
    dh["rotate"] = T.rotate(xh, ang)
    dh["rotate_center_nearest"] = T.rotate(xh, ang, center=torch.tensor([[10.0, 5.0]]).expand(3, -1), mode="nearest", padding_mode="border")
    dh["translate"] = T.translate(xh, trn)
    dh["scale"] = T.scale(xh, scl)
    dh["scale_iso"] = T.scale(xh, torch.tensor([1.3]))
    dh["shear"] = T.shear(xh, shr)
    dh["affine"] = T.affine(xh, Aff)
    dh["affine_unbatched"] = T.affine(xh[0], Aff[:1])
    boxes = torch.tensor([[[3.0, 2.0], [20.0, 4.0], [22.0, 15.0], [2.0, 13.0]],
                          [[5.0, 5.0], [25.0, 5.0], [25.0, 18.0], [5.0, 18.0]],
                          [[0.0, 0.0], [29.0, 0.0], [29.0, 20.0], [0.0, 20.0]]])
    dh["boxes"] = boxes
    dh["crop_and_resize"] = T.crop_and_resize(xh, boxes, (9, 14))
    dh["crop_and_resize_ac0"] = T.crop_and_resize(xh, boxes, (9, 14), align_corners=False)
    dh["center_crop"] = T.center_crop(xh, (10, 16))
    dh["center_crop_odd_nearest"] = T.center_crop(xh, (7, 9), mode="nearest")
    dstb = torch.tensor([[[0.0, 0.0], [11.0, 0.0], [11.0, 7.0], [0.0, 7.0]]]).expand(3, -1, -1)
    dh["crop_by_boxes"] = T.crop_by_boxes(xh, boxes, dstb)
    Mc = T.get_perspective_transform(boxes, dstb)
    dh["crop_by_transform_mat"] = T.crop_by_transform_mat(xh, Mc, (8, 12))
    dh["crop_by_transform_mat_affine"] = T.crop_by_transform_mat(xh, Aff, (8, 12), align_corners=False)
    save("warp_callers", **dh)

    # ---- callers of filter2d: box_blur / laplacian -----------------------------------------------------------------
    db2 = {}
    xb2 = torch.rand(2, 3, 20, 28, generator=g)
    db2["x"] = xb2
    db2["box3"] = F.box_blur(xb2, 3)
    db2["box35_sep_replicate"] = F.box_blur(xb2, (3, 5), "replicate", separable=True)
    db2["box57_constant"] = F.box_blur(xb2, (5, 7), "constant")
    db2["lap3"] = F.laplacian(xb2, 3)
    db2["lap5_unnorm_circular"] = F.laplacian(xb2, 5, "circular", normalized=False)
    db2["unsharp"] = F.unsharp_mask(xb2, (5, 5), (1.5, 1.5))
    save("filter_callers", **db2)

    # ---- pyramid: pyrdown / pyrup / build_pyramid (geometry/transform/pyramid.py:409-560) ------------------------------
    dpy = {}
    for tag, shape in (("even", (2, 3, 32, 48)), ("odd", (1, 2, 37, 29)), ("tiny", (1, 1, 6, 7))):
        xp = torch.rand(*shape, generator=g)
        dpy["x_" + tag] = xp
        for border in BORDERS:
            for ac in (False, True):
                dpy[f"down_{tag}_{border}_{int(ac)}"] = T.pyrdown(xp, border, ac)
                if tag != "even":
                    dpy[f"up_{tag}_{border}_{int(ac)}"] = T.pyrup(xp, border, ac)
        dpy["down_" + tag + "_factor1p5"] = T.pyrdown(xp, "reflect", False, 1.5)
        dpy["down_" + tag + "_factor3"] = T.pyrdown(xp, "replicate", True, 3.0)
    xg = dpy["x_even"].clone().requires_grad_(True)
    wgt = torch.rand(2, 3, 16, 24, generator=g)
    (T.pyrdown(xg) * wgt).sum().backward()
    dpy["down_even_w"] = wgt
    dpy["down_even_gx"] = xg.grad
    for i, lvl in enumerate(T.build_pyramid(dpy["x_even"], 4)):
        dpy[f"pyr_even_{i}"] = lvl
    xl = torch.rand(1, 2, 32, 64, generator=g)
    dpy["x_lap"] = xl
    for i, lvl in enumerate(T.build_laplacian_pyramid(xl, 3)):
        dpy[f"lap_{i}"] = lvl
    dpy["lit_pyrdown_in"] = torch.arange(16, dtype=torch.float32).reshape(1, 1, 4, 4)
    dpy["lit_pyrdown_out"] = T.pyrdown(dpy["lit_pyrdown_in"], align_corners=True)
    dpy["lit_pyrup_in"] = torch.arange(4, dtype=torch.float32).reshape(1, 1, 2, 2)
    dpy["lit_pyrup_out"] = T.pyrup(dpy["lit_pyrup_in"], align_corners=True)
    save("pyramid", **dpy)

    # ---- ImageRegistrator: one level's masked loss + its gradient wrt the model, and the toy registration ---------------
    # (geometry/transform/image_registrator.py:225-245; tests/geometry/transform/test_image_registrator.py:87-99)
    import torch.nn.functional as TF

    dr = {}
    xs = torch.rand(1, 3, 40, 56, generator=g)
    xd = torch.rand(1, 3, 40, 56, generator=g)
    dr["src"], dr["dst"] = xs, xd
    Hs = torch.eye(3)[None].repeat(4, 1, 1)
    Hs[1] = torch.tensor([[1.05, 0.02, 0.01], [-0.03, 0.97, -0.02], [0.0, 0.0, 1.0]])
    Hs[2] = torch.tensor([[0.9, -0.15, 0.2], [0.12, 1.1, -0.1], [0.05, -0.04, 1.0]])
    Hs[3] = torch.tensor([[0.6, 0.5, 0.4], [-0.5, 0.6, 0.3], [0.0, 0.0, 1.0]])  # large rotation: a good part falls outside
    dr["H"] = Hs
    for name, fn in (("l1", TF.l1_loss), ("mse", TF.mse_loss)):
        reg = T.ImageRegistrator("homography", loss_fn=fn)
        for k in range(4):
            Hk = Hs[k : k + 1].clone().requires_grad_(True)
            loss = reg.get_single_level_loss(xs, xd, Hk)
            loss.backward()
            dr[f"{name}_loss_{k}"] = loss.detach()
            dr[f"{name}_grad_{k}"] = Hk.grad
    homography = torch.eye(3)[None]
    homography[..., 0, 0] = 1.05
    homography[..., 1, 1] = 1.05
    homography[..., 0, 2] = 0.01
    toy_src = torch.rand(1, 3, 16, 18, generator=g)
    toy_dst = K.geometry.homography_warp(toy_src, homography, (16, 18), align_corners=False)
    torch.manual_seed(0)
    IR = T.ImageRegistrator("Similarity", num_iterations=500, lr=3e-4, pyramid_levels=2)
    model, inter = IR.register(toy_src, toy_dst, output_intermediate_models=True)
    dr["toy_src"], dr["toy_dst"], dr["toy_H"], dr["toy_model"] = toy_src, toy_dst, homography, model.detach()
    dr["toy_inter0"], dr["toy_inter1"] = inter[0], inter[1]
    save("registration", **dr)

    # ---- ScalePyramid (geometry/transform/pyramid.py:151-400) ----------------------------------------------------------
    dsp = {}
    xsp = torch.rand(1, 1, 64, 80, generator=g)
    dsp["x"] = xsp
    for tag, kw in (("default", {}), ("double", {"double_image": True, "n_levels": 2, "extra_levels": 2, "min_size": 20}), ("small_sigma", {"init_sigma": 0.4, "n_levels": 2, "min_size": 10})):
        pyr, sig, pd = T.ScalePyramid(**kw)(xsp)
        dsp[tag + "_n"] = len(pyr)
        for o, (a, b, c) in enumerate(zip(pyr, sig, pd)):
            dsp[f"{tag}_pyr_{o}"], dsp[f"{tag}_sig_{o}"], dsp[f"{tag}_pd_{o}"] = a, b, c
    save("scale_pyramid", **dsp)

    # ---- canny (filters/canny.py:32-161) --------------------------------------------------------------------------------
    dcn = {}
    yy, xx = torch.meshgrid(torch.arange(48.0), torch.arange(64.0), indexing="ij")
    shapes = ((xx - 30) ** 2 + (yy - 22) ** 2 < 15**2).float() * 0.7 + ((xx > 40) & (yy > 30)).float() * 0.3
    xc = (shapes[None, None].repeat(2, 3, 1, 1) * torch.tensor([1.0, 0.8, 0.6]).view(1, 3, 1, 1) + 0.08 * torch.rand(2, 3, 48, 64, generator=g)).clamp(0, 1)
    dcn["x"] = xc
    dcn["mag"], dcn["edges"] = F.canny(xc)
    dcn["mag_nohyst"], dcn["edges_nohyst"] = F.canny(xc, hysteresis=False)
    dcn["mag_gray_k3"], dcn["edges_gray_k3"] = F.canny(xc[:, :1], 0.05, 0.3, (3, 3), (0.8, 0.8))
    save("canny", **dcn)

    # ---- BASELINE config 3 at its own spatial size: AugmentationSequential(RandomAffine, ColorJitter, RandomGaussianBlur) on 224x224 -------
    # (parity by parameter replay; the GPU box has no reference, so the sampled parameters and the reference's
    #  fp32 output travel as a fixture.  The image holds bf16-representable values so that the bf16 leg starts from the same pixels.)
    A = K.augmentation
    torch.manual_seed(7)
    aug = A.AugmentationSequential(
        A.RandomAffine(degrees=15.0, translate=(0.1, 0.1), scale=(0.8, 1.2), shear=5.0, p=1.0),
        A.ColorJitter(0.2, 0.2, 0.2, 0.1, p=1.0),
        A.RandomGaussianBlur((5, 5), (0.1, 2.0), p=1.0),
    )
    x3 = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(33)).bfloat16()
    out3 = aug(x3.float())
    d3 = {"x_bf16_bits": x3.view(torch.int16).numpy().view(np.uint16), "out": out3}
    names = ("affine", "jitter", "blur")
    for name, p3 in zip(names, aug._params):
        for k, v in p3.data.items():
            if isinstance(v, torch.Tensor):
                d3[f"{name}__{k}"] = v
    d3["affine__matrix"] = aug[0].transform_matrix
    # stage outputs of the first image (the pipeline re-run module by module with the same parameters) for localising a mismatch
    st = x3.float()
    for name, mod, p3 in zip(names, aug, aug._params):
        st = mod(st, params=p3.data)
        d3[f"stage_{name}_img0"] = st[0].clone()
    assert torch.equal(st, out3)
    save("config3", **d3)

    # ---- the geometric augmentations with p < 1 (kornia/augmentation/_2d/geometric/{affine,perspective}.py, base.py:348-393): the sampled
    #      parameters - batch_prob included - and the reference's output, for the replay of kornia_amd.augmentation on the device ----------------
    torch.manual_seed(19)
    xg = torch.rand(6, 3, 48, 64, generator=torch.Generator().manual_seed(91))
    dg = {"x": xg}
    for name, mod in (("perspective", A.RandomPerspective(0.4, p=0.6)), ("perspective_nearest_align", A.RandomPerspective(0.3, resample="nearest", align_corners=True, p=0.6)),
                      ("affine", A.RandomAffine(degrees=25.0, translate=(0.1, 0.2), scale=(0.7, 1.3), shear=8.0, p=0.6)),
                      ("affine_border", A.RandomAffine(degrees=10.0, padding_mode="border", align_corners=True, p=0.6))):
        for attempt in range(20):  # a draw that mixes transformed and untouched samples
            out = mod(xg)
            bp = mod._params["batch_prob"] > 0.5
            if 0 < int(bp.sum()) < xg.shape[0]:
                break
        dg[f"{name}__out"] = out
        dg[f"{name}__matrix"] = mod.transform_matrix
        for k, v in mod._params.items():
            if isinstance(v, torch.Tensor):
                dg[f"{name}__{k}"] = v
    save("geometric_aug", **dg)

    # ---- inf / NaN in grad_out at a pixel that samples ENTIRELY outside the source (own generator: appended in round 4) -------------------
    # ATen's CPU backward multiplies the zeros it gathered for out-of-bounds taps by grad_out, so the matrix gradient of that image is NaN
    # while grad wrt the image stays finite; the one-read backward of the HIP path visits no such pixel and must find it by other means.
    import math

    g2 = torch.Generator().manual_seed(404)
    dn = {}
    Bn, Cn, Hn_, Wn = 3, 3, 70, 65
    c_, s_ = math.cos(math.radians(45.0)), math.sin(math.radians(45.0))
    cx, cy = (Wn - 1) / 2.0, (Hn_ - 1) / 2.0
    Mn = torch.tensor([[c_, s_, (1 - c_) * cx - s_ * cy], [-s_, c_, s_ * cx + (1 - c_) * cy]]).repeat(Bn, 1, 1)
    xn = torch.rand(Bn, Cn, Hn_, Wn, generator=g2)
    for tag, bad in (("nan", float("nan")), ("inf", float("inf"))):
        gon = torch.rand(Bn, Cn, Hn_, Wn, generator=g2) - 0.5
        gon[2, :, Hn_ - 1, Wn - 1] = bad  # a corner of the output: rotated out of the source
        xr, Mr = xn.clone().requires_grad_(), Mn.clone().requires_grad_()
        T.warp_affine(xr, Mr, (Hn_, Wn)).backward(gon)
        dn[f"affine_{tag}__go"], dn[f"affine_{tag}__gx"], dn[f"affine_{tag}__gM"] = gon, xr.grad, Mr.grad
    dn["affine__x"], dn["affine__M"] = xn, Mn
    Bp, Hp, Wp = 2, 96, 160
    Mpn = flagship_homographies(Bp, Hp, Wp, Hp, Wp, g2, jitter=3.0)
    Mpn[:, 0, 2] += 40.0  # 40 px to the right: the left columns of the output sample outside
    xp = torch.rand(Bp, 3, Hp, Wp, generator=g2)
    gop = torch.rand(Bp, 3, Hp, Wp, generator=g2) - 0.5
    gop[0, :, 50, 2] = float("inf")
    xr, Mr = xp.clone().requires_grad_(), Mpn.clone().requires_grad_()
    T.warp_perspective(xr, Mr, (Hp, Wp)).backward(gop)
    dn.update(persp__x=xp, persp__M=Mpn, persp__go=gop, persp__gx=xr.grad, persp__gM=Mr.grad)
    save("nonfinite_outside", **dn)

    # ---- NON-FINITE SAMPLING COORDINATES (own generator: appended in round 6) ---------------------------------------------------------------
    # A singular matrix (zero determinant into the closed-form inverse), a NaN / inf entry, or a projective denominator of exactly zero make the
    # sampling coordinates NaN / +-inf.  ATen's CPU sampler (bilinear, zeros padding) masks such taps out and still multiplies the zeros it
    # gathers by the NaN weights: NaN forward, nothing scattered into grad wrt the image, NaN into the grid - hence the matrix - gradient.
    # Bicubic: NaN through the coefficients.  Nearest: the converted index is out of bounds -> 0.  (border / reflection with such coordinates
    # convert a non-finite float to an integer - the platform's conversion, and ATen's CPU backward reads out of bounds there: not a fixture.)
    g3 = torch.Generator().manual_seed(606)
    Bq, Cq, Hq, Wq = 5, 3, 21, 33
    xq = torch.rand(Bq, Cq, Hq, Wq, generator=g3)
    goq = torch.rand(Bq, Cq, Hq, Wq, generator=g3) - 0.5
    ok3 = flagship_homographies(Bq, Hq, Wq, Hq, Wq, g3, jitter=2.0)
    nan, inf = float("nan"), float("inf")
    Mq = ok3.clone()
    Mq[0] = 0.0                                   # singular: every coordinate NaN
    Mq[1, 0, 2] = nan                             # a NaN entry
    Mq[2, 1, 1] = inf                             # an inf entry
    a_ = 0.0625                                   # denominator a (x - (W - 1) / 2): exactly zero on the centre column of the output, finite elsewhere
    Mq[3] = torch.linalg.inv(torch.tensor([[1.0, 0, 0], [0, 1, 0], [a_, 0, -a_ * (Wq - 1) / 2]], dtype=torch.float64)).float()
    Aq = rotation_affines(Bq, Hq, Wq, g3)
    Aq[0] = 0.0
    Aq[1, 0, 2] = nan
    Aq[2, 1, 1] = inf
    Aq[3, 0, 0] = 3e38                            # overflows to inf in the coordinate
    Hn = torch.eye(3).repeat(Bq, 1, 1) + 0.02 * torch.randn(Bq, 3, 3, generator=g3)
    Hn[0] = 0.0                                   # Z = 0: transform_points' guard -> s = 1, coordinates 0 (finite)
    Hn[1, 0, 2] = nan
    Hn[2, 2, 2] = inf
    Hn[3, 0, 0] = 3e38
    dq = {"x": xq, "go": goq, "persp__M": Mq, "affine__M": Aq, "homography__M": Hn, "fill": torch.tensor([0.1, 0.2, 0.3])}
    for api, fn, Min in (("persp", T.warp_perspective, Mq), ("affine", T.warp_affine, Aq), ("homography", T.homography_warp, Hn)):
        for mode, pad in (("bilinear", "zeros"), ("bilinear", "fill"), ("bicubic", "zeros"), ("nearest", "zeros")):
            if api == "homography" and pad == "fill":
                continue
            xr, Mr = xq.clone().requires_grad_(), Min.clone().requires_grad_()
            kw = dict(mode=mode, padding_mode=pad)
            if pad == "fill":
                kw["fill_value"] = dq["fill"]
            out = fn(xr, Mr, (Hq, Wq), **kw)
            out.backward(goq)
            key = f"{api}__{mode}_{pad}"
            dq[key + "__out"], dq[key + "__gx"] = out.detach(), xr.grad
            dq[key + "__gM"] = Mr.grad if Mr.grad is not None else torch.zeros_like(Min)
    save("nonfinite_coords", **dq)

    # ---- SEEDED augmentation pipelines (own generator: appended in round 6) -----------------------------------------------------------------
    # `torch.manual_seed(s); AugmentationSequential(RandomAffine, ColorJitter, RandomGaussianBlur)(x)` - BASELINE config 3 as it is written,
    # sampling included: the parameters Kornia's generators draw for the seed and its fp32 output, for kornia_amd.augmentation's modules (which
    # draw from the same generator in the same order) on a machine without Kornia.
    from kornia.augmentation import AugmentationSequential, ColorJitter, RandomAffine, RandomGaussianBlur

    pipelines = {
        "config3": lambda: AugmentationSequential(RandomAffine(degrees=15.0, translate=(0.1, 0.1), scale=(0.8, 1.2), shear=5.0, p=1.0),
                                                  ColorJitter(0.2, 0.2, 0.2, 0.1, p=1.0), RandomGaussianBlur((5, 5), (0.1, 2.0), p=1.0)),
        "with_probabilities": lambda: AugmentationSequential(RandomAffine(degrees=(-30.0, 10.0), scale=(0.7, 1.1, 0.9, 1.3), shear=(-4.0, 4.0, -2.0, 6.0), padding_mode="border", p=0.6),
                                                             ColorJitter(0.3, (0.5, 1.5), 0.1, (-0.05, 0.2), p=0.7), RandomGaussianBlur((3, 5), (0.2, 1.5), border_type="replicate", p=0.5)),
        "same_on_batch": lambda: AugmentationSequential(RandomAffine(degrees=20.0, translate=(0.2, 0.05), p=0.8), ColorJitter(0.1, 0.1, 0.1, 0.1),
                                                        RandomGaussianBlur((5, 5), (0.5, 1.0), p=1.0), same_on_batch=True),
    }
    da = {"x": torch.rand(5, 3, 40, 56, generator=torch.Generator().manual_seed(808))}
    for pname, make in pipelines.items():
        for seed in (3, 11):
            torch.manual_seed(seed)
            aug = make()
            out = aug(da["x"])
            key = f"{pname}__seed{seed}"
            da[key + "__out"] = out
            da[key + "__rng_after"] = torch.get_rng_state()[:64].clone()  # (the head of the Mersenne state: the generator stands where Kornia left it)
            for item in aug._params:
                for k, v in item.data.items():
                    if isinstance(v, torch.Tensor):
                        da[f"{key}__{item.name}__{k}"] = v
    save("aug_modules", **da)


if __name__ == "__main__":
    torch.set_num_threads(4)
    main()
