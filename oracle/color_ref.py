"""TEST INFRASTRUCTURE ONLY - CPU restatement (torch elementwise ops, fp32) of the reference's ColorJitter arithmetic:
kornia/enhance/adjust.py:80-134 (saturation), :179-254 (hue), :414-469 (contrast), :542-593 (brightness),
kornia/color/hsv.py:27-131, kornia/color/gray.py:60-92, sequenced as
kornia/augmentation/_2d/intensity/color_jitter.py:126-159 does.  Pinned against tests/golden/color_jitter.npz
(produced by the real reference).  Never imported by the product path."""
from __future__ import annotations

import math

import torch


def _bc(f: torch.Tensor, image: torch.Tensor) -> torch.Tensor:
    f = torch.as_tensor(f, dtype=image.dtype)
    while f.dim() != image.dim():
        f = f[..., None]
    return f


def rgb_to_grayscale(image):
    r, g, b = image.unbind(dim=-3)
    w = torch.tensor([0.299, 0.587, 0.114], dtype=image.dtype)
    out = r * w[0]
    out = torch.addcmul(out, g, w[1])
    out = torch.addcmul(out, b, w[2])
    return out.unsqueeze(-3)


def adjust_brightness_accumulative(image, factor):
    return (image * _bc(factor, image)).clamp(min=0.0, max=1.0)


def adjust_contrast_with_mean_subtraction(image, factor):
    f = _bc(factor, image)
    mean = rgb_to_grayscale(image).mean((-2, -1), True)
    return (image * f + mean * (1 - f)).clamp(min=0.0, max=1.0)


def adjust_saturation_with_gray_subtraction(image, factor):
    f = _bc(factor, image)
    return torch.clamp((1 - f) * rgb_to_grayscale(image) + f * image, 0.0, 1.0)


def rgb_to_hsv(image, eps=1e-8):
    mx, mn = image.amax(-3), image.amin(-3)
    deltac = mx - mn
    v = mx
    s = deltac / (mx + eps)
    deltac = torch.where(deltac == 0, torch.ones_like(deltac), deltac)
    rc, gc, bc = torch.unbind(mx.unsqueeze(-3) - image, dim=-3)
    h1 = bc - gc
    h2 = (rc - bc) + 2.0 * deltac
    h3 = (gc - rc) + 4.0 * deltac
    r, g, b = torch.unbind(image, dim=-3)
    h = torch.where((r >= g) & (r >= b), h1, torch.where(g >= b, h2, h3))
    h = h / deltac
    h = (h / 6.0) % 1.0
    h = 2.0 * math.pi * h
    return torch.stack((h, s, v), dim=-3)


def hsv_to_rgb(image):
    h = image[..., 0, :, :] / (2 * math.pi)
    s = image[..., 1, :, :]
    v = image[..., 2, :, :]
    hi = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - hi
    p = v * (1.0 - s)
    q = v * (1.0 - f * s)
    t = v * (1.0 - (1.0 - f) * s)
    hi = hi.long().clamp_(0, 5)
    m0, m1, m2, m3, m4 = (hi == k for k in range(5))
    r = torch.where(m0, v, torch.where(m1, q, torch.where(m2, p, torch.where(m3, p, torch.where(m4, t, v)))))
    g = torch.where(m0, t, torch.where(m1, v, torch.where(m2, v, torch.where(m3, q, torch.where(m4, p, p)))))
    b = torch.where(m0, p, torch.where(m1, p, torch.where(m2, t, torch.where(m3, v, torch.where(m4, v, q)))))
    return torch.stack((r, g, b), dim=-3)


def adjust_hue(image, factor):
    hsv = rgb_to_hsv(image)
    h, s, v = torch.chunk(hsv, chunks=3, dim=-3)
    h = torch.fmod(h + _bc(factor, hsv), 2 * math.pi)
    return hsv_to_rgb(torch.cat([h, s, v], dim=-3))


def color_jitter(image, brightness_factor, contrast_factor, saturation_factor, hue_factor, order):
    """hue_factor in turns (x 2 pi applied here, as color_jitter.py:147 does)."""
    fns = [
        lambda im: adjust_brightness_accumulative(im, brightness_factor),
        lambda im: adjust_contrast_with_mean_subtraction(im, contrast_factor),
        lambda im: adjust_saturation_with_gray_subtraction(im, saturation_factor),
        lambda im: adjust_hue(im, hue_factor * 2 * math.pi),
    ]
    out = image
    for i in order:
        out = fns[int(i)](out)
    return out
