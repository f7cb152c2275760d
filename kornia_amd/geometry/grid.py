"""create_meshgrid (kornia/geometry/grid.py:24-80).

The warp kernels generate these coordinates in registers; this host version exists for API
completeness (``HomographyWarper.grid``, ``warp_grid`` callers) and is plain tensor construction.
"""
from __future__ import annotations

from typing import Optional

import torch

__all__ = ["create_meshgrid"]


def create_meshgrid(
    height: int,
    width: int,
    normalized_coordinates: bool = True,
    device: Optional[torch.device] = None,
    dtype: Optional[torch.dtype] = None,
) -> torch.Tensor:
    """(1,H,W,2) grid of (x, y) coordinates, in [-1, 1] when ``normalized_coordinates``."""
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    if normalized_coordinates:
        xs = (xs / (width - 1) - 0.5) * 2
        ys = (ys / (height - 1) - 0.5) * 2
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([gx, gy], dim=-1).unsqueeze(0)
