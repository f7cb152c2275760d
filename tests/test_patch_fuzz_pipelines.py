"""Build-container only (needs /root/reference and ROCm's clang++): SEEDED random augmentation pipelines, unpatched against patched.

`torch.manual_seed(s); AugmentationSequential(...)(x)` must give the same images with and without `kornia_amd.patch()`: the hooks may
not consume the generator differently from the modules they replace (the reference samples its parameters on the host),
may not disturb `random_apply`'s choice, the per-sample probability mask, `same_on_batch`, `keepdim`, or what `inverse()` replays.
Random chains of 1-4 of the hooked modules (+ a flip, which is not hooked) with random arguments; time-bounded (`KM_SWEEP_SECONDS`).

Recorded long run (round 5, 270 s): 9 700 pipelines, none differing by more than 1.3e-4 (chains of RandomPerspective: the native
`get_perspective_transform` and torch's solver differ in the last bit, SURVEY 8(a20) / tests/run_reference_tests_on_native.py KNOWN).
Left out because one flipped pixel is a difference of 1: `nearest` resampling under RandomPerspective, and masks (warped with
`nearest` by the container) - 4 such flips in those 9 700."""
import os
import random
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "emu")]
import ref_shim  # noqa: E402

pytestmark = [
    pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present (GPU box)"),
    pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"), reason="host build of the kernels needs ROCm's clang++"),
]

# In the suite every test runs a FIXED number of seeded pipelines (the same ones on every host); KM_SWEEP_SECONDS turns a test into a
# time-bounded sweep that keeps drawing until the time is up (the recorded long runs).
SECONDS = float(os.environ.get("KM_SWEEP_SECONDS", "0"))
FIXED = 60


def _more(t0, n, fixed=FIXED):
    return (time.time() - t0 < SECONDS) if SECONDS > 0 else (n < fixed)


def _pipeline_factory(A, rng):
    """draws a pipeline description once; the returned callable builds a fresh instance of it"""
    u, ch = rng.uniform, rng.choice
    pool = [
        lambda: (A.RandomAffine, (u(0, 40), (u(0, 0.2),) * 2 if rng.random() < 0.7 else None, (u(0.6, 1), u(1, 1.5)) if rng.random() < 0.7 else None, u(0, 10) if rng.random() < 0.5 else None),
                 dict(resample=ch(["bilinear", "nearest", "bicubic"]), padding_mode=ch(["zeros", "border", "reflection"]), p=ch([0.0, 0.3, 0.7, 1.0]), same_on_batch=rng.random() < 0.2, keepdim=rng.random() < 0.3)),
        lambda: (A.RandomPerspective, (u(0.1, 0.6),), dict(p=ch([0.0, 0.5, 1.0]), same_on_batch=rng.random() < 0.2)),
        lambda: (A.RandomRotation, (u(5, 90),), dict(p=ch([0.4, 1.0]), same_on_batch=rng.random() < 0.2)),
        lambda: (A.RandomShear, (tuple(sorted((u(-10, 10), u(-10, 10)))),), dict(p=ch([0.4, 1.0]))),
        lambda: (A.RandomTranslate, (tuple(sorted((u(0, 0.3), u(0, 0.3)))),), dict(p=ch([0.4, 1.0]))),
        lambda: (A.ColorJitter, (u(0, 0.4), u(0, 0.4), u(0, 0.4), u(0, 0.3)), dict(p=ch([0.0, 0.5, 1.0]), same_on_batch=rng.random() < 0.2)),
        lambda: (A.RandomGaussianBlur, ((ch([3, 5, 7]), ch([3, 5])), (0.1, u(0.5, 3))),
                 dict(p=ch([0.0, 0.5, 1.0]), separable=rng.random() < 0.8, border_type=ch(["reflect", "replicate", "circular", "constant"]))),
        lambda: (A.RandomHorizontalFlip, (), dict(p=0.5)),
    ]  # fmt: skip
    spec = [ch(pool)() for _ in range(rng.randint(1, 4))]
    random_apply = ch([False, False, False, 1, 2])
    if random_apply is not False and random_apply > len(spec):
        random_apply = False
    return lambda: A.AugmentationSequential(*[cls(*a, **kw) for cls, a, kw in spec], data_keys=["input"], random_apply=random_apply)


def _close(r, a, has_nearest, what):
    """5e-4 everywhere - except that a pipeline with a `nearest` resampling may flip single pixels (a position an ulp apart rounds to the other
    neighbour: a difference of up to 1 at that pixel, and at the pixels a later blur spreads it over): there at most 0.5 % of the pixels may
    differ by more."""
    d = (r - a).abs()
    if has_nearest:
        assert (d > 5e-4).float().mean().item() <= 5e-3, (what, (d > 5e-4).float().mean().item())
    else:
        assert d.max().item() <= 5e-4, (what, d.max().item())


def test_seeded_pipelines_reproduce_under_patch():
    K = ref_shim.import_reference()
    from mode import emulated_device

    import kornia_amd.kornia_patch as P

    rng = random.Random(20250923)
    t0, n = time.time(), 0
    while _more(t0, n):
        make = _pipeline_factory(K.augmentation, rng)
        seed, with_inverse = rng.randint(0, 10**6), rng.random() < 0.5
        x = torch.rand(rng.randint(1, 5), 3, rng.randint(12, 48), rng.randint(12, 48), generator=torch.Generator().manual_seed(seed))
        torch.manual_seed(seed)
        aug = make()
        ref = [aug(x)]
        if with_inverse:
            ref.append(aug.inverse(ref[0]))
        with emulated_device():
            assert P.patch() > 0
            try:
                torch.manual_seed(seed)
                aug2 = make()
                out = [aug2(x.cuda())]
                if with_inverse:
                    out.append(aug2.inverse(out[0]))
            finally:
                P.unpatch()
        names = [type(m).__name__ for m in aug.children()]
        for r, a in zip(ref, out):
            assert r.shape == a.shape and r.dtype == a.dtype, (n, seed, names)
            _close(r, a, any(getattr(m, "flags", {}).get("resample", None) is not None and "NEAREST" in str(m.flags["resample"]).upper() for m in aug.children()), (n, seed, names))
        n += 1
    assert n >= 25


def test_seeded_pipelines_gradient_then_replay_then_inverse_under_patch():
    """The call sequences in which hooks could leave state behind (the advisor's round-4 finding was one): a grad-enabled forward with
    the gradient taken, then a no_grad replay of the same parameters, then `inverse()` - image, image gradient, replay and inverse all
    against the unpatched modules.  Recorded: 1 900 pipelines in 120 s, worst 8.3e-5."""
    K = ref_shim.import_reference()
    from mode import emulated_device

    import kornia_amd.kornia_patch as P

    def run(make, x, go_seed, seed, device):
        torch.manual_seed(seed)
        aug = make()
        xg = x.to(device).requires_grad_(True) if device else x.clone().requires_grad_(True)
        y = aug(xg)
        go = torch.rand(y.shape, generator=torch.Generator().manual_seed(go_seed))
        (y * (go.to(device) if device else go)).sum().backward()
        with torch.no_grad():
            y2 = aug(xg.detach(), params=aug._params)
            inv = aug.inverse(y2)
        return [y.detach(), xg.grad / max(1.0, xg.grad.abs().max().item()), y2, inv], [type(m).__name__ for m in aug.children()]

    rng = random.Random(777)
    t0, n = time.time(), 0
    while _more(t0, n, 40):
        make, seed = _pipeline_factory(K.augmentation, rng), rng.randint(0, 10**6)
        x = torch.rand(rng.randint(1, 4), 3, rng.randint(12, 40), rng.randint(12, 40), generator=torch.Generator().manual_seed(seed))
        ref, names = run(make, x, seed + 2, seed, None)
        with emulated_device():
            assert P.patch() > 0
            try:
                out, _ = run(make, x, seed + 2, seed, "cuda")
            finally:
                P.unpatch()
        for what, r, a in zip(("image", "image gradient", "replay", "inverse"), ref, out):
            assert r.shape == a.shape, (n, seed, names, what)
            assert (r - a).abs().max().item() <= 5e-4, (n, seed, names, what, (r - a).abs().max().item())
        n += 1
    assert n >= 20
