"""Shared test helpers: seeded inputs in the shape of BASELINE.json's configs."""
from __future__ import annotations

import torch


def perspective_from_quads(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """(B,4,2) x (B,4,2) -> (B,3,3) homographies mapping src->dst (8x8 DLT solve in float64, H[2,2]=1).
    Stand-in for kornia.geometry.get_perspective_transform so tests/bench need no reference import."""
    B = src.shape[0]
    s, d = src.double(), dst.double()
    A = torch.zeros(B, 8, 8, dtype=torch.float64)
    b = torch.zeros(B, 8, dtype=torch.float64)
    for k in range(4):
        x, y, u, v = s[:, k, 0], s[:, k, 1], d[:, k, 0], d[:, k, 1]
        A[:, 2 * k, 0], A[:, 2 * k, 1], A[:, 2 * k, 2] = x, y, 1.0
        A[:, 2 * k, 6], A[:, 2 * k, 7] = -u * x, -u * y
        A[:, 2 * k + 1, 3], A[:, 2 * k + 1, 4], A[:, 2 * k + 1, 5] = x, y, 1.0
        A[:, 2 * k + 1, 6], A[:, 2 * k + 1, 7] = -v * x, -v * y
        b[:, 2 * k], b[:, 2 * k + 1] = u, v
    h = torch.linalg.solve(A, b)
    return torch.cat([h, torch.ones(B, 1, dtype=torch.float64)], dim=1).view(B, 3, 3).float()


def flagship_homographies(B: int, H: int, W: int, h: int, w: int, gen: torch.Generator, jitter: float = 8.0) -> torch.Tensor:
    """The reference benchmark's recipe (benchmarks/geometry/flagship.py:89-107): image quad -> quad + jitter*randn."""
    src = torch.tensor([[0.0, 0.0], [W - 1.0, 0.0], [W - 1.0, H - 1.0], [0.0, H - 1.0]]).expand(B, 4, 2)
    dst = torch.tensor([[0.0, 0.0], [w - 1.0, 0.0], [w - 1.0, h - 1.0], [0.0, h - 1.0]]).expand(B, 4, 2)
    return perspective_from_quads(src, dst + jitter * torch.randn(B, 4, 2, generator=gen))


def rotation_affines(B: int, H: int, W: int, gen: torch.Generator) -> torch.Tensor:
    """(B,2,3) rotation+scale+translation about the image centre (get_rotation_matrix2d-like)."""
    ang = (torch.rand(B, generator=gen) - 0.5) * 1.5
    sc = 0.8 + 0.4 * torch.rand(B, generator=gen)
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0
    a, b = sc * torch.cos(ang), sc * torch.sin(ang)
    tx = (1 - a) * cx - b * cy + 3.0 * torch.randn(B, generator=gen)
    ty = b * cx + (1 - a) * cy + 3.0 * torch.randn(B, generator=gen)
    return torch.stack([torch.stack([a, b, tx], -1), torch.stack([-b, a, ty], -1)], 1)


def smooth_image(B: int, C: int, H: int, W: int) -> torch.Tensor:
    v, u = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    img = 0.5 + 0.5 * torch.sin(6 * torch.pi * u) * torch.cos(4 * torch.pi * v)
    return img.expand(B, C, H, W).contiguous()


def golden(name: str):
    """Load tests/golden/<name>.npz (reference-generated fixture) as a dict of numpy arrays."""
    import os

    import numpy as np

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}
