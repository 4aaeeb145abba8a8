"""Synthetic volume pairs for benchmarks and tests (no dataset ships with the reference).

uniform_pair : x, y ~ U[0,1)                                  (BASELINE configs 1-3)
oasis_like_pair : a smooth "anatomy" inside an ellipsoidal head mask with zero background, and a moving image that is the
                  fixed one deformed by a smooth random displacement of a few voxels (BASELINE configs 4-5, "OASIS-style")
Both are generated on the CPU generator (reproducible across devices) and finished on the GPU with the HIP resampling / warp
operators of the hot path."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from . import ops


def uniform_pair(size: Sequence[int], batch: int, seed: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 1, *size, generator=g).to(device)
    y = torch.rand(batch, 1, *size, generator=g).to(device)
    return x, y


def oasis_like_pair(size: Sequence[int], batch: int, seed: int, device, max_disp: float = 3.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """fixed y = smooth texture (U[0,1) on a size/8 lattice, tri-linearly up-sampled) x ellipsoid mask (semi-axes 0.4 * extent);
    moving x = y resampled through a smooth random displacement (U[-max_disp, max_disp] voxels on a size/16 lattice, up-sampled);
    both clipped to [0, 1]."""
    size = [int(s) for s in size]
    g = torch.Generator().manual_seed(seed)
    coarse = torch.rand(batch, 1, *[max(s // 8, 2) for s in size], generator=g).to(device)
    base = ops.resize_trilinear(coarse, size)
    axes = [torch.linspace(-0.5, 0.5, s, device=device) for s in size]
    zz, yy, xx = torch.meshgrid(*axes, indexing="ij")
    mask = ((zz / 0.4) ** 2 + (yy / 0.4) ** 2 + (xx / 0.4) ** 2 <= 1.0).float()[None, None]
    y = (base * mask).clamp_(0.0, 1.0).contiguous()
    lattice = (torch.rand(batch, 3, *[max(s // 16, 2) for s in size], generator=g) * 2 - 1).to(device) * max_disp
    field = ops.resize_trilinear(lattice, size)
    # displacement in voxels -> the SpatialTransformer's convention: the field is added to the voxel grid (network_blocks.py:101-121)
    x = ops.warp3d(field.contiguous(), y).clamp_(0.0, 1.0).contiguous()
    return x, y
