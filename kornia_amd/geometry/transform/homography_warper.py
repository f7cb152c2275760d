"""HomographyWarper module (kornia/geometry/transform/homography_warper.py:77-195)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from ..grid import create_meshgrid
from .imgwarp import homography_warp, warp_grid

__all__ = ["HomographyWarper"]


class HomographyWarper(nn.Module):
    r"""Warp (N,C,H,W) tensors by destination->source homographies, :math:`X_{src} = H_{src}^{dst} X_{dst}`.

    Args mirror the reference: ``height``/``width`` of the destination, ``mode``, ``padding_mode``,
    ``normalized_coordinates`` (default True), ``align_corners`` (default **False**).

    ``precompute_warp_grid(H)`` keeps the homography: the native warp regenerates the grid in
    registers, so a later ``forward(patch)`` costs one launch and reads no (N,H,W,2) grid from HBM.
    ``_warped_grid`` is still materialised lazily for code that inspects it.
    """

    def __init__(
        self,
        height: int,
        width: int,
        mode: str = "bilinear",
        padding_mode: str = "zeros",
        normalized_coordinates: bool = True,
        align_corners: bool = False,
    ) -> None:
        super().__init__()
        self.height = height
        self.width = width
        self.mode = mode
        self.padding_mode = padding_mode
        self.normalized_coordinates = normalized_coordinates
        self.align_corners = align_corners
        self.grid = create_meshgrid(height, width, normalized_coordinates=normalized_coordinates)
        self._precomputed_homography: Optional[torch.Tensor] = None
        self._warped_grid_cache: Optional[torch.Tensor] = None

    @property
    def _warped_grid(self) -> Optional[torch.Tensor]:
        if self._precomputed_homography is None:
            return None
        if self._warped_grid_cache is None:
            H = self._precomputed_homography
            self._warped_grid_cache = warp_grid(self.grid.to(H.device), H)
        return self._warped_grid_cache

    def precompute_warp_grid(self, src_homo_dst: torch.Tensor) -> None:
        """Remember the homography/ies ((1,3,3), (N,3,3) or (N,1,3,3)) for later ``forward(patch)``."""
        # a snapshot, like the reference's warp_grid(self.grid, src_homo_dst) at this point (homography_warper.py:134):
        # later in-place updates of ``src_homo_dst`` (an optimiser step on a Parameter) must not change forward(patch).
        # clone() keeps the autograd link, so a differentiable precompute stays differentiable.
        self._precomputed_homography = src_homo_dst.reshape(-1, 3, 3).clone()
        self._warped_grid_cache = None

    def forward(self, patch_src: torch.Tensor, src_homo_dst: Optional[torch.Tensor] = None) -> torch.Tensor:
        if src_homo_dst is None:
            H = self._precomputed_homography
            if H is None:
                raise RuntimeError(
                    "Unknown warping. If homographies are not provided they must be preset"
                    " using the method: precompute_warp_grid()."
                )
            if not H.device == patch_src.device:
                raise TypeError(
                    "Patch and warped grid must be on the same device. Got"
                    f" patch.device: {patch_src.device} warped_grid.device: {H.device}. Whether recall"
                    " precompute_warp_grid() with the correct device for the homograhy"
                    " or change the patch device."
                )
            src_homo_dst = H
        return homography_warp(
            patch_src,
            src_homo_dst,
            (self.height, self.width),
            mode=self.mode,
            padding_mode=self.padding_mode,
            align_corners=self.align_corners,
            normalized_coordinates=self.normalized_coordinates,
        )
