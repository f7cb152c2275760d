"""Drop-in activation for an installed Kornia: ``kornia_amd.patch()`` rebinds Kornia's hot-path
functions - in every ``kornia.*`` module that imported them by value (augmentation, affwarp, crop2d,
pyramid, tracking, contrib ... ~40 modules) - to dispatchers that run the native
gfx950 kernels for HIP tensors and Kornia's own implementation otherwise.

The fallback inside a dispatcher is *Kornia's* function (CPU tensors, float64 where unsupported,
tracing / torch.compile / export, exotic arguments): it is not part of the kornia_amd product path,
which itself never falls back.  ``unpatch()`` restores the originals; each dispatcher keeps the
original reachable as ``__wrapped__`` (``torch.jit.script`` users must script that one).
"""
from __future__ import annotations

import functools
import importlib
import sys
import threading
import weakref
from typing import Callable

import torch

from . import _native as _N
from . import enhance as _e
from . import filters as _f
from . import geometry as _g

_NATIVE = {
    "kornia.geometry.transform.imgwarp": {
        "warp_perspective": _g.warp_perspective,
        "warp_affine": _g.warp_affine,
        "homography_warp": _g.homography_warp,
        "warp_grid": _g.warp_grid,
        "remap": _g.remap,
        "get_affine_matrix2d": _g.get_affine_matrix2d,  # RandomAffine.compute_transformation: one launch instead of ~45
        "get_perspective_transform": _g.get_perspective_transform,  # RandomPerspective / crops: one launch instead of ~40
    },
    # pyrdown as one fused launch (blur + decimation), pyrup on the native resize + 5x5 filter
    "kornia.geometry.transform.pyramid": {"pyrdown": _g.transform.pyrdown, "pyrup": _g.transform.pyrup},
    "kornia.geometry.linalg": {"transform_points": _g.transform_points},
    "kornia.geometry.conversions": {"normalize_homography": _g.normalize_homography},
    "kornia.filters.filter": {"filter2d": _f.filter2d, "filter2d_separable": _f.filter2d_separable},
    "kornia.filters.gaussian": {"gaussian_blur2d": _f.gaussian_blur2d},
    "kornia.filters.sobel": {"spatial_gradient": _f.spatial_gradient, "sobel": _f.sobel},
    # blur + Sobel were already native through the two entries above; this one also replaces the ~25 elementwise / fixed-kernel
    # convolution launches and the host-synchronised hysteresis loop behind them (km_canny_nms_fwd, km_canny_hysteresis_sweep)
    "kornia.filters.canny": {"canny": _f.canny},
    # fused colour kernels (the ColorJitter leg)
    "kornia.enhance.adjust": {
        "adjust_brightness_accumulative": _e.adjust_brightness_accumulative,
        "adjust_contrast_with_mean_subtraction": _e.adjust_contrast_with_mean_subtraction,
        "adjust_saturation_with_gray_subtraction": _e.adjust_saturation_with_gray_subtraction,
        "adjust_hue": _e.adjust_hue,
    },
}
_COLOR_OPS = {id(f) for f in _NATIVE["kornia.enhance.adjust"].values()}
_COLOR_DTYPES = (torch.float32, torch.bfloat16, torch.float16)
_SUPPORTED = (torch.float32, torch.float64, torch.bfloat16, torch.float16)
_patched: dict = {}  # id(original) -> (original, dispatcher)
_patched_methods: list = []  # (class, attribute name, original function)


def _use_native(args, kwargs) -> bool:
    if torch.jit.is_tracing() or torch.jit.is_scripting() or torch.compiler.is_compiling():
        return False
    tensors = [a for a in list(args) + list(kwargs.values()) if isinstance(a, torch.Tensor)]
    if not tensors:
        return False
    return all(_N.on_device(t) and t.dtype in _SUPPORTED for t in tensors if t.is_floating_point()) and any(_N.on_device(t) for t in tensors)


def _use_native_color(args, kwargs) -> bool:
    """The fused colour kernels: RGB float32/bfloat16/float16 images, default options (differentiable: km_color_jitter_bwd)."""
    if not _use_native(args, kwargs) or kwargs.get("clip_output", True) is not True:
        return False
    image = args[0] if args else kwargs.get("image")
    if not isinstance(image, torch.Tensor) or image.dim() < 3 or image.shape[-3] != 3 or image.dtype not in _COLOR_DTYPES:
        return False
    return True


def _dispatcher(original: Callable, native: Callable) -> Callable:
    accept = _use_native_color if id(native) in _COLOR_OPS else _use_native

    @functools.wraps(original)
    def wrapper(*args, **kwargs):
        if accept(args, kwargs):
            return native(*args, **kwargs)
        return original(*args, **kwargs)

    wrapper.__wrapped__ = original
    wrapper.__kornia_amd_native__ = native
    return wrapper


def _color_jitter_apply(original: Callable) -> Callable:
    """ColorJitter.apply_transform (kornia/augmentation/_2d/intensity/color_jitter.py:126-159) as ONE fused pass.

    The module's per-stage ``torch.where((factor != neutral).any(), fn(img), img)`` guards become a (4,) device flag
    vector read by the kernel, so nothing synchronises beyond what the reference's own loop over ``order`` does."""

    from . import augmentation as _aug

    @functools.wraps(original)
    def apply_transform(self, input, params, flags, transform=None):
        keys = ("brightness_factor", "contrast_factor", "saturation_factor", "hue_factor")
        ok = (
            isinstance(input, torch.Tensor) and _N.on_device(input) and input.dim() == 4 and input.shape[1] == 3
            and input.dtype in _COLOR_DTYPES
            and all(isinstance(params.get(k), torch.Tensor) for k in keys)
            and not (torch.jit.is_tracing() or torch.jit.is_scripting() or torch.compiler.is_compiling())
        )
        order = None
        if ok:
            order = self._fixed_order if getattr(self, "_fixed_order", None) is not None else params["order"].tolist()
            order = [int(i) for i in order]
            # the fused kernel applies each stage at most once per pass and takes ONE contrast mean; the reference accepts any
            # sequence of ids in 0..3, so anything else stays on its own implementation
            ok = len(order) <= 4 and all(0 <= i <= 3 for i in order) and len(set(order)) == len(order)
        if not ok:
            return original(self, input, params, flags, transform)
        # factor table, stage switches ((factor != neutral).any() per stage) and the per-sample probability switch of transform_inputs
        # (augmentation/base.py:380-393) come out of ONE launch (km_color_params_fwd), and the switch rides inside the colour kernel: a sample
        # whose draw failed is passed through untouched, and the torch.where pass that follows finds nothing to do (_blend_by_prob below).
        # No torch op of this hook's own touches the device.
        switched = (_switch_allowed() and not (getattr(self, "p", 1.0) == 1.0 and getattr(self, "p_batch", 1.0) == 1.0)
                    and not (torch.is_grad_enabled() and input.requires_grad))
        bp = params.get("batch_prob") if switched else None
        switched = isinstance(bp, torch.Tensor) and bp.numel() == input.shape[0]
        sub = {k: params[k] for k in keys}
        if switched:
            sub["batch_prob"] = bp
        out = _aug.color_jitter(input, sub, order)
        if switched:
            out._kornia_amd_blended_with = weakref.ref(input)
        return out

    apply_transform.__wrapped__ = original
    return apply_transform


def _registrator_level_loss(original: Callable) -> Callable:
    """ImageRegistrator.get_single_level_loss (kornia/geometry/transform/image_registrator.py:225-245) as ONE launch:
    both warps, the elementwise loss, the mask, the selection and the mean - and their backward - fused (km_warp_masked_loss)."""
    import torch.nn.functional as F

    from .geometry.transform.image_registrator import masked_warp_loss

    @functools.wraps(original)
    def get_single_level_loss(self, img_src, img_dst, transform_model):
        import kornia.geometry.transform.homography_warper as hw_mod

        kind = "l1" if self.loss_fn is F.l1_loss else ("mse" if self.loss_fn is F.mse_loss else None)
        ok = (
            kind is not None and self.warper is hw_mod.HomographyWarper
            and isinstance(img_src, torch.Tensor) and isinstance(img_dst, torch.Tensor) and _N.on_device(img_src) and _N.on_device(img_dst)
            and img_src.dim() == 4 and img_src.shape == img_dst.shape and img_src.shape[-1] >= 2 and img_src.dtype in _COLOR_DTYPES
            and img_src.dtype == img_dst.dtype and isinstance(transform_model, torch.Tensor) and transform_model.dim() == 3
            and transform_model.shape[-2:] == (3, 3) and transform_model.shape[0] in (1, img_src.shape[0])
            and not (torch.is_grad_enabled() and (img_src.requires_grad or img_dst.requires_grad))
            and not (torch.jit.is_tracing() or torch.jit.is_scripting() or torch.compiler.is_compiling())
        )
        if not ok:
            return original(self, img_src, img_dst, transform_model)
        return masked_warp_loss(img_src, img_dst, transform_model, kind)

    get_single_level_loss.__wrapped__ = original
    return get_single_level_loss


_tls = threading.local()  # .blend_follows > 0: the caller is transform_inputs, whose _blend_by_prob call comes right after apply_transform


def _transform_inputs(original: Callable) -> Callable:
    """_AugmentationBase.transform_inputs (kornia/augmentation/base.py:363-400) unchanged, except that the apply_transform hooks below
    know they were called from it: only there is ``params['batch_prob']`` the draw of THIS call and is the result blended afterwards, so
    only there may the per-sample switch ride inside the op's launch.  A direct call of ``apply_transform`` or ``inverse_transform``
    (which hands over ``self._params``) transforms every sample it is given, as in the reference."""

    @functools.wraps(original)
    def transform_inputs(self, *args, **kwargs):
        _tls.blend_follows = getattr(_tls, "blend_follows", 0) + 1
        try:
            return original(self, *args, **kwargs)
        finally:
            _tls.blend_follows -= 1

    transform_inputs.__wrapped__ = original
    return transform_inputs


def _switch_allowed() -> bool:
    return getattr(_tls, "blend_follows", 0) > 0


def _blend_by_prob(original: Callable) -> Callable:
    """_AugmentationBase._blend_by_prob (kornia/augmentation/base.py:348-361): the per-sample probability blend of every augmentation,
    ``torch.where(to_apply, transformed, input)``, as one native pass that reads only the side it keeps (km_select_samples_fwd)."""
    from .augmentation import select_samples

    def blend(transformed, not_transformed, to_apply):
        # a geometric augmentation whose apply step already carried the per-sample switch inside its warp launch (_geometric_apply):
        # the samples that are not transformed were copied there, nothing is left to select
        tag = getattr(transformed, "_kornia_amd_blended_with", None)
        if tag is not None and tag() is not_transformed:
            return transformed
        ok = (
            isinstance(transformed, torch.Tensor) and isinstance(not_transformed, torch.Tensor) and isinstance(to_apply, torch.Tensor)
            and _N.on_device(transformed) and _N.on_device(not_transformed) and transformed.shape == not_transformed.shape
            and transformed.dim() >= 2 and to_apply.dim() == 1 and transformed.shape[0] == to_apply.shape[0]
            and transformed.dtype == not_transformed.dtype and transformed.dtype in _COLOR_DTYPES
            and not (torch.is_grad_enabled() and (transformed.requires_grad or not_transformed.requires_grad))
            and not (torch.jit.is_tracing() or torch.jit.is_scripting() or torch.compiler.is_compiling())
        )
        if not ok:
            return original(transformed, not_transformed, to_apply)
        return select_samples(transformed, not_transformed, to_apply.to(transformed.device))

    blend.__wrapped__ = original
    return staticmethod(blend)


def _gaussian_blur_apply(original: Callable) -> Callable:
    """RandomGaussianBlur.apply_transform (kornia/augmentation/_2d/intensity/gaussian_blur.py:95-114): the instance holds the reference's
    gaussian_blur2d captured at construction (:93), so the method itself is replaced - per-sample taps in one launch
    (km_gaussian_taps_fwd) and the fused separable blur, nothing synchronises on the sampled sigmas."""
    from .augmentation import gaussian_taps
    from .filters.filter import filter2d_separable_taps

    @functools.wraps(original)
    def apply_transform(self, input, params, flags, transform=None):
        sigma = params.get("sigma") if hasattr(params, "get") else None
        ok = (
            isinstance(input, torch.Tensor) and _N.on_device(input) and input.dim() == 4 and input.dtype in _COLOR_DTYPES
            and isinstance(sigma, torch.Tensor) and sigma.dim() == 1 and flags.get("separable", True)
            and not (torch.is_grad_enabled() and (input.requires_grad or sigma.requires_grad))
            and not (torch.jit.is_tracing() or torch.jit.is_scripting() or torch.compiler.is_compiling())
        )
        if not ok:
            return original(self, input, params, flags, transform)
        s1 = sigma.to(device=input.device, dtype=torch.float32)
        if getattr(self, "same_on_batch", False):
            s1 = s1[:1]
        # (one sigma per sample, the taps rounded to the image dtype inside the launch: no expanded copy, no cast round trips)
        taps_x, taps_y = gaussian_taps(s1, flags["kernel_size"], round_to=input.dtype)
        return filter2d_separable_taps(input, taps_x, taps_y, flags["border_type"].name.lower())

    apply_transform.__wrapped__ = original
    return apply_transform


def _affine_compute(original: Callable) -> Callable:
    """RandomAffine.compute_transformation (_2d/geometric/affine.py:125-141) with the normalise / invert chain of the warp that follows it
    folded into the same launch (``km_affine_params_chain_fwd``): the six sampled parameter tensors -> the (B,3,3) pixel matrix the module
    keeps as its ``transform_matrix`` AND the (B,9) matrix the warp kernel reads (the warp's prologue).  The second one
    is parked on the module, keyed by the identity of the sampled ``angle`` tensor; ``apply_transform`` (below) takes it when it is called
    with the same parameters for an image of the same size - two launches for the whole augmentation instead of three."""
    from . import augmentation as _aug

    keys = ("translations", "center", "scale", "angle", "shear_x", "shear_y")

    @functools.wraps(original)
    def compute_transformation(self, input, params, flags):
        ok = (
            isinstance(input, torch.Tensor) and _N.on_device(input) and input.dim() == 4 and input.dtype in _COLOR_DTYPES
            and all(isinstance(params.get(k), torch.Tensor) for k in keys)
            and not (torch.is_grad_enabled() and any(params[k].requires_grad for k in keys))
            and not (torch.jit.is_tracing() or torch.jit.is_scripting() or torch.compiler.is_compiling())
        )
        if ok:
            B = input.shape[0]
            ok = all(params[k].shape[0] == B for k in keys) and tuple(params["translations"].shape) == (B, 2) and tuple(params["scale"].shape) == (B, 2)
        self._kornia_amd_chain = None
        if not ok:
            return original(self, input, params, flags)
        height, width = int(input.shape[-2]), int(input.shape[-1])
        chain_params = {k: params[k] for k in keys}
        bp = params.get("batch_prob")
        if isinstance(bp, torch.Tensor) and bp.numel() == B:
            chain_params["batch_prob"] = bp  # (thresholded inside the same launch: the uint8 switch the warp reads)
        m, M, flags_u8 = _aug.affine_chain(chain_params, input.device, height, width, with_matrix=True)
        # parked only for the warp that transform_inputs runs next on the same image: an image that requires a gradient goes through the
        # module's own apply_transform (below), which would leave the chain behind for a later, unrelated call (inverse_transform hands
        # over the same parameter tensors with the INVERSE matrix)
        if not (torch.is_grad_enabled() and input.requires_grad):
            self._kornia_amd_chain = (weakref.ref(params["angle"]), m, height, width, flags_u8)
        return M.to(input.dtype)

    compute_transformation.__wrapped__ = original
    return compute_transformation


def _geometric_apply(original: Callable, kind: str) -> Callable:
    """RandomAffine.apply_transform (kornia/augmentation/_2d/geometric/affine.py:143-162) and RandomPerspective.apply_transform
    (_2d/geometric/perspective.py:100-115) with the per-sample probability switch of ``transform_inputs`` (augmentation/base.py:380-393)
    folded INTO the warp's launch: the matrix chain (normalise, invert) is one launch, the warp with the switch another
    (``km_warp2d_fwd_masked`` copies the samples whose ``batch_prob`` draw failed), and the ``torch.where`` pass that follows in the
    reference finds nothing to do (``_blend_by_prob`` above recognises the result).  The matrix is the one the module hands over
    (``transform``: what ``compute_transformation`` - already one native launch - produced, or the caller's own in
    ``inverse_transform``), never re-derived from the parameters.  Anything unusual falls through to the module's own method."""
    from .geometry.transform.imgwarp import COORD_AFFINE, COORD_PERSPECTIVE, _warp, _warp_affine_from_chain

    @functools.wraps(original)
    def apply_transform(self, input, params, flags, transform=None):
        # the matrix parked by compute_transformation (_affine_compute) serves exactly ONE call - this one, whatever path it takes:
        # taken off the module before any fall-through, so that no later call (inverse_transform, a direct apply_transform) finds it
        chain = getattr(self, "_kornia_amd_chain", None)
        if chain is not None:
            self._kornia_amd_chain = None
        ok = (
            isinstance(input, torch.Tensor) and isinstance(transform, torch.Tensor) and _N.on_device(input) and _N.on_device(transform)
            and input.dim() == 4 and input.dtype in _COLOR_DTYPES and transform.dim() == 3 and tuple(transform.shape[-2:]) == (3, 3)
            and transform.shape[0] == input.shape[0] and transform.dtype in _SUPPORTED
            and not (torch.is_grad_enabled() and (input.requires_grad or transform.requires_grad))
            and not (torch.jit.is_tracing() or torch.jit.is_scripting() or torch.compiler.is_compiling())
        )
        if not ok:
            return original(self, input, params, flags, transform)
        B, _, height, width = input.shape
        bp = None
        if _switch_allowed() and not (getattr(self, "p", 1.0) == 1.0 and getattr(self, "p_batch", 1.0) == 1.0):
            bp = params.get("batch_prob") if hasattr(params, "get") else None
            if not (isinstance(bp, torch.Tensor) and bp.numel() == B):  # (inverse_inputs may call with a subset of the batch: no switch then)
                bp = None

        def switch():  # the per-sample switch as a (B,) bool tensor (two small torch launches: only where no parameter launch made it already)
            return None if bp is None else torch.atleast_1d(bp.to(input.device) > 0.5)

        apply = None
        mode = flags["resample"].name.lower()
        if kind == "rotation":
            # RandomRotation.apply_transform (_2d/geometric/rotation.py:112-124): affine(input, transform[..., :2, :3], resample, "zeros",
            # align_corners) = warp_affine to the input's own size (affwarp.py:136-193)
            apply = switch()
            out = _warp(input, transform[:, :2, :].contiguous(), (height, width), COORD_AFFINE, 1, mode, "zeros", flags["align_corners"], None, apply)
        elif kind == "affine":
            padding_mode = flags["padding_mode"].name.lower()
            fill_value = flags.get("fill_value")
            if padding_mode == "fill" and fill_value is None:
                fill_value = torch.zeros(input.shape[1], device=input.device, dtype=input.dtype)
            angle = params.get("angle") if hasattr(params, "get") else None
            # (only inside transform_inputs - the forward call that compute_transformation preceded; inverse_transform and direct calls
            # bring their own `transform`, which is the one to warp with)
            if (chain is not None and _switch_allowed() and angle is not None and chain[0]() is angle and chain[2:4] == (height, width)
                    and chain[1].shape[0] == B):
                # the matrix the warp reads came out of compute_transformation's own launch (_affine_compute): straight to the sampler.
                # (16-bit images: the reference forms the pixel matrix IN the image dtype; the sampler here reads the float32 one - nearer to
                # the float32 reference the 1e-2 of BASELINE.json is measured against, and no cast / chain launches between the two kernels)
                # (the per-sample switch as the parameter launch thresholded it - the same draw: no torch op of this hook's own on the device)
                apply = None if bp is None else (chain[4] if chain[4] is not None else switch())
                out = _warp_affine_from_chain(input, chain[1], mode, padding_mode, flags["align_corners"], fill_value, apply)
            else:
                apply = switch()
                out = _warp(input, transform[:, :2, :], (height, width), COORD_AFFINE, 1, mode, padding_mode, flags["align_corners"], fill_value, apply)
        else:
            apply = switch()
            out = _warp(input, transform, (height, width), COORD_PERSPECTIVE, 1, mode, "zeros", flags["align_corners"], torch.zeros(3), apply)
        if apply is not None:
            out._kornia_amd_blended_with = weakref.ref(input)
        return out

    apply_transform.__wrapped__ = original
    return apply_transform


def patch() -> int:
    """Activate the native path inside Kornia. Returns the number of rebound module attributes.

    All or nothing: if the installed Kornia lacks one of the modules / attributes the hooks bind to (another release's layout), everything
    rebound so far is restored before the error propagates - no half-patched library, and a later ``patch()`` starts from scratch."""
    if _patched:
        return 0
    import kornia  # noqa: F401 - must be importable

    try:
        return _patch_all()
    except BaseException:
        _patched.setdefault(0, (None, None))  # (unpatch() returns early on an empty table: the method hooks are undone below either way)
        unpatch()
        raise


def _patch_all() -> int:
    for mod_name, table in _NATIVE.items():
        mod = importlib.import_module(mod_name)
        for name, native in table.items():
            original = getattr(mod, name)
            _patched[id(original)] = (original, _dispatcher(original, native))
    count = 0
    for mod_name, mod in list(sys.modules.items()):
        if mod is None or not (mod_name == "kornia" or mod_name.startswith("kornia.")):
            continue
        for attr, value in list(vars(mod).items()):
            hit = _patched.get(id(value))
            if hit is not None and value is hit[0]:
                setattr(mod, attr, hit[1])
                count += 1
    # ColorJitter instances bind the four adjust functions at construction and wrap each in a torch.where:
    # replace the method so that the whole sequence is one kernel
    cj_mod = importlib.import_module("kornia.augmentation._2d.intensity.color_jitter")
    original = cj_mod.ColorJitter.apply_transform
    cj_mod.ColorJitter.apply_transform = _color_jitter_apply(original)
    _patched_methods.append((cj_mod.ColorJitter, "apply_transform", original))
    ir_mod = importlib.import_module("kornia.geometry.transform.image_registrator")
    original = ir_mod.ImageRegistrator.get_single_level_loss
    ir_mod.ImageRegistrator.get_single_level_loss = _registrator_level_loss(original)
    _patched_methods.append((ir_mod.ImageRegistrator, "get_single_level_loss", original))
    # the augmentation layer: the probability blend of every augmentation and RandomGaussianBlur's apply step
    base_mod = importlib.import_module("kornia.augmentation.base")
    original = base_mod._AugmentationBase.transform_inputs
    base_mod._AugmentationBase.transform_inputs = _transform_inputs(original)
    _patched_methods.append((base_mod._AugmentationBase, "transform_inputs", original))
    original = base_mod._AugmentationBase.__dict__["_blend_by_prob"]  # the staticmethod object itself
    base_mod._AugmentationBase._blend_by_prob = _blend_by_prob(original.__func__)
    _patched_methods.append((base_mod._AugmentationBase, "_blend_by_prob", original))
    gb_mod = importlib.import_module("kornia.augmentation._2d.intensity.gaussian_blur")
    original = gb_mod.RandomGaussianBlur.apply_transform
    gb_mod.RandomGaussianBlur.apply_transform = _gaussian_blur_apply(original)
    _patched_methods.append((gb_mod.RandomGaussianBlur, "apply_transform", original))
    # the geometric leg of the same layer: the probability switch rides in the warp's own launch
    # (RandomShear / RandomTranslate apply exactly like RandomAffine - _2d/geometric/shear.py:113-131, translate.py:102-120 -, RandomRotation
    # through `affine` with zeros padding)
    aff_mod = importlib.import_module("kornia.augmentation._2d.geometric.affine")
    original = aff_mod.RandomAffine.compute_transformation
    aff_mod.RandomAffine.compute_transformation = _affine_compute(original)
    _patched_methods.append((aff_mod.RandomAffine, "compute_transformation", original))
    for mod_name, cls_name, kind in (("kornia.augmentation._2d.geometric.affine", "RandomAffine", "affine"),
                                     ("kornia.augmentation._2d.geometric.perspective", "RandomPerspective", "perspective"),
                                     ("kornia.augmentation._2d.geometric.shear", "RandomShear", "affine"),
                                     ("kornia.augmentation._2d.geometric.translate", "RandomTranslate", "affine"),
                                     ("kornia.augmentation._2d.geometric.rotation", "RandomRotation", "rotation")):
        cls = getattr(importlib.import_module(mod_name), cls_name)
        original = cls.apply_transform
        cls.apply_transform = _geometric_apply(original, kind)
        _patched_methods.append((cls, "apply_transform", original))
    return count + 11


def unpatch() -> int:
    """Restore Kornia's own functions."""
    if not _patched:
        return 0
    by_wrapper = {id(w): o for o, w in _patched.values() if w is not None}
    count = 0
    for mod_name, mod in list(sys.modules.items()):
        if mod is None or not (mod_name == "kornia" or mod_name.startswith("kornia.")):
            continue
        for attr, value in list(vars(mod).items()):
            orig = by_wrapper.get(id(value))
            if orig is not None:
                setattr(mod, attr, orig)
                count += 1
    for cls, name, original in _patched_methods:
        setattr(cls, name, original)
        count += 1
    _patched_methods.clear()
    _patched.clear()
    return count


def is_patched() -> bool:
    return bool(_patched)
