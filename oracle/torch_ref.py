"""TEST INFRASTRUCTURE ONLY - the reference's hot path restated as the PyTorch op sequence it executes.

Kornia is pure Python over ATen: on CPU its ``warp_perspective`` + ``gaussian_blur2d`` IS this
sequence of torch ops (``F.grid_sample``, ``F.pad``, ``F.conv2d`` ...).  This module restates that
sequence (citing the reference lines) so that the GPU box - where ``/root/reference`` does not
exist but PyTorch does - can (a) time "the reference's CPU path" on its host cores for
``bench.py``'s ``cpu_baseline`` leg and (b) cross-check the C oracle at sizes with no committed
fixture.  It is validated against the real reference in ``tests/test_oracle_golden.py`` (against the fixtures the real
reference wrote, ``oracle/make_golden.py``).

Never imported by the product path.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _inv3(m: torch.Tensor) -> torch.Tensor:
    # kornia/core/utils.py:159-166
    a, b, c = m[..., :, 0], m[..., :, 1], m[..., :, 2]
    r0 = torch.linalg.cross(b, c, dim=-1)
    r1 = torch.linalg.cross(c, a, dim=-1)
    r2 = torch.linalg.cross(a, b, dim=-1)
    det = (a * r0).sum(-1)
    return torch.stack([r0, r1, r2], dim=-2) / det[..., None, None]


def _npix(h: int, w: int) -> torch.Tensor:
    # kornia/geometry/conversions.py:1750-1763
    wd = 1e-14 if w == 1 else w - 1.0
    hd = 1e-14 if h == 1 else h - 1.0
    return torch.tensor([[2.0 / wd, 0.0, -1.0], [0.0, 2.0 / hd, -1.0], [0.0, 0.0, 1.0]]).unsqueeze(0)


def normalize_homography(M, src_size, dst_size):
    # kornia/geometry/conversions.py:1713-1725
    ns = _npix(*src_size).to(M)
    nd = _npix(*dst_size).to(M)
    return nd @ (M @ _inv3(ns))


def _meshgrid(h, w, device=None, dtype=None):
    # kornia/geometry/grid.py:65-80
    xs = torch.linspace(0, w - 1, w, device=device, dtype=dtype)
    ys = torch.linspace(0, h - 1, h, device=device, dtype=dtype)
    xs = (xs / (w - 1) - 0.5) * 2
    ys = (ys / (h - 1) - 0.5) * 2
    g = torch.stack(torch.meshgrid([xs, ys], indexing="ij"), dim=-1)
    return g.permute(1, 0, 2).unsqueeze(0)


def warp_perspective(src, M, dsize, mode="bilinear", padding_mode="zeros", align_corners=True):
    # kornia/geometry/transform/imgwarp.py:143-174 (eager branch)
    B, _, H, W = src.shape
    h, w = dsize
    m = _inv3(normalize_homography(M, (H, W), (h, w)))
    grid = _meshgrid(h, w, device=src.device).to(src.dtype)
    gx0, gy0 = grid[..., 0], grid[..., 1]
    den = m[:, 2, 0, None, None] * gx0 + m[:, 2, 1, None, None] * gy0 + m[:, 2, 2, None, None]
    gx = (m[:, 0, 0, None, None] * gx0 + m[:, 0, 1, None, None] * gy0 + m[:, 0, 2, None, None]) / den
    gy = (m[:, 1, 0, None, None] * gx0 + m[:, 1, 1, None, None] * gy0 + m[:, 1, 2, None, None]) / den
    return F.grid_sample(src, torch.stack([gx, gy], dim=-1), align_corners=align_corners, mode=mode, padding_mode=padding_mode)


def _gauss1d(k: int, sigma: torch.Tensor) -> torch.Tensor:
    # kornia/filters/kernels.py:113-120
    x = (torch.arange(k, device=sigma.device, dtype=sigma.dtype) - float(k // 2)).expand(sigma.shape[0], -1)
    if k % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * sigma.pow(2.0)))
    return g / g.sum(-1, keepdim=True)


def _filter2d(x, kernel, border="reflect"):
    # kornia/filters/filter.py:122-150 ('same', 'corr')
    b, c, h, w = x.shape
    k = kernel[:, None].to(x).expand(-1, c, -1, -1)
    kh, kw = k.shape[-2:]
    pads = [(kw - 1) // 2, kw - 1 - (kw - 1) // 2, (kh - 1) // 2, kh - 1 - (kh - 1) // 2]
    x = F.pad(x, pads, mode=border)
    k = k.reshape(-1, 1, kh, kw)
    x = x.view(-1, k.size(0), x.size(-2), x.size(-1))
    return F.conv2d(x, k, groups=k.size(0), padding=0, stride=1).view(b, c, h, w)


def gaussian_blur2d(x, kernel_size, sigma, border="reflect"):
    # kornia/filters/gaussian.py:95-115 (separable branch)
    s = torch.tensor([sigma], device=x.device, dtype=x.dtype)
    ky, kx = kernel_size
    kernel_x = _gauss1d(kx, s[:, 1].view(1, 1))
    kernel_y = _gauss1d(ky, s[:, 0].view(1, 1))
    out_x = _filter2d(x, kernel_x[..., None, :], border)
    return _filter2d(out_x, kernel_y[..., None], border)


def headline_step(x, M, grad_out, dsize=(512, 512)):
    """One fwd+bwd of BASELINE.json's metric: gaussian_blur2d(warp_perspective(x, M)), grads wrt x and M."""
    x = x.detach().requires_grad_()
    M = M.detach().requires_grad_()
    y = gaussian_blur2d(warp_perspective(x, M, dsize), (5, 5), (1.5, 1.5))
    y.backward(grad_out)
    return y.detach(), x.grad, M.grad
