"""Evaluation metrics of the reference's harness on the device.

`Evaluate.rmse` / `.dsc` / `.jdet` / `.warp_landmarks` (evaluate.py:315-335, 410-423) and the "JDetLeq0" expression (evaluate.py:1441-1446)
are methods / inline code of the unchanged caller; these functions compute the same scalars with HIP kernels (one streaming pass and a
device-side finish, no intermediate tensors) so that `Evaluate` can call them instead of its torch expressions."""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from .network_blocks import ResizeTransform


def rmse(input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """root mean squared error between two images (evaluate.py:315-319)"""
    return ops.rmse(input, target)


def dsc(input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """dice similarity coefficient between two segmentation maps (evaluate.py:321-327)"""
    return ops.dsc(input, target)


def jdet(df: torch.Tensor) -> torch.Tensor:
    """Jacobian determinant of a displacement field (evaluate.py:329-335 -> src.losses.jacobian_det)"""
    return ops.jacobian_det(df, True)


def jdet_leq0_percent(df: torch.Tensor) -> torch.Tensor:
    """percentage of voxels whose Jacobian determinant is <= 0 (evaluate.py:1441-1446): folding / non-diffeomorphic voxels"""
    return ops.percent_leq0(ops.jacobian_det(df, True))


def warp_landmarks(lm: torch.Tensor, df: torch.Tensor) -> torch.Tensor:
    """landmarks moved by a displacement field sampled at the (truncated) landmark positions (evaluate.py:410-423)"""
    return ops.warp_landmarks(lm, df)


def resize_dfs(dfs: Dict[int, torch.Tensor], target_size=None) -> Dict[int, torch.Tensor]:
    """every level's field resized (and rescaled) to the size of level 0 or to `target_size` - the evident intent of the reference's
    src/components/utils.py:4-13, which cannot run as written (`range(dfs.keys())`, the batch dimension used as a size)"""
    out = {}
    for l, d in dfs.items():
        tgt = dfs[0].shape[2] if target_size is None else target_size[0]
        out[l] = ResizeTransform(vel_resize=1 / (tgt / d.shape[2]), ndims=d.dim() - 2)(d)
    return out
