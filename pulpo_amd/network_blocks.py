"""Network blocks of PULPo with the reference's class names, constructor signatures and state-dict keys
(reference: src/network_blocks.py), every forward running hand-written HIP kernels through pulpo_amd.ops.

The nn.Conv3d / nn.BatchNorm3d children exist only as parameter + buffer containers so that checkpoints written by the
reference load unchanged ('_op.0.weight', '_op.1.running_mean', ...); they are never called.
2-D inputs (train.py --ndims 2) get nn.Conv2d / nn.BatchNorm2d containers, as in the reference, and run as depth-1 volumes through the
same kernels (pulpo_amd.ops lifts (B,C,H,W) tensors; see the "2-D mode" block there).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import ops


def gauss_sampler(mu: torch.Tensor, sigma: torch.Tensor, var: Optional[int] = 1) -> torch.Tensor:
    """z = mu + sigma * eps, eps ~ N(0, var^2) drawn in fp32 (reference src/network_blocks.py:7-8).
    PULPoEncoder recognises this function and fuses it into the mu/sigma head kernel."""
    return mu + sigma * (var * torch.randn_like(sigma, dtype=torch.float32))


class FixedNoiseSampler:
    """Sampler with injected noise (tests / reproducible inference): z = mu + sigma * eps with a given eps tensor."""

    def __init__(self, eps: torch.Tensor):
        self.fixed_eps = eps

    def __call__(self, mu, sigma):
        return mu + sigma * self.fixed_eps


def _ndims(input_size: Sequence[int], who: str) -> int:
    nd = len(input_size)
    if nd not in (2, 3):
        raise NotImplementedError(f"{who}: volumes (ndims 3) or slices (ndims 2) expected, got ndims={nd}")
    return nd


class ConvUnit(nn.Module):
    """Conv3d(3x3x3, pad 1) -> BatchNorm3d -> LeakyReLU(0.2): one fused HIP pipeline (src/network_blocks.py:11-29)"""

    def __init__(self, input_size: Sequence[int], in_channels: int, out_channels: int = None) -> None:
        super().__init__()
        nd = _ndims(input_size, "ConvUnit")
        out_channels = out_channels or in_channels
        Conv, BatchNorm = (nn.Conv3d, nn.BatchNorm3d) if nd == 3 else (nn.Conv2d, nn.BatchNorm2d)
        self._op = nn.Sequential(
            Conv(in_channels, out_channels, kernel_size=3, padding=1),
            BatchNorm(out_channels),
            nn.LeakyReLU(negative_slope=0.2, inplace=True),
        )

    def forward(self, x: torch.Tensor, pool_after: bool = False, out=None, pool_only: bool = False, blocked_out: bool = False):
        """out: (buffer, first channel), pool_only: the caller reads only AvgPool(result) -> (result or None, pooled or None); blocked_out: the result in the
        channel-blocked form (C / 8, B, D, H, W, 8) for the next unit of a ConvSequence; see ops.conv_bn_lrelu"""
        conv, bn = self._op[0], self._op[1]
        use_batch_stats = self.training or bn.running_mean is None
        # running statistics and num_batches_tracked are updated inside the BatchNorm finalize kernel
        return ops.conv_bn_lrelu(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                 training=use_batch_stats, momentum=bn.momentum, eps=bn.eps,
                                 num_batches_tracked=bn.num_batches_tracked if self.training else None, pool_after=pool_after, out=out,
                                 pool_only=pool_only, blocked_out=blocked_out)


class ConvSequence(nn.Module):
    """`depth` ConvUnits, the first one changes the channel count (src/network_blocks.py:32-46)"""

    def __init__(self, input_size: Sequence[int], in_channels: int, out_channels: int, depth: int) -> None:
        super().__init__()
        units = [ConvUnit(input_size, in_channels, out_channels)]
        units += [ConvUnit(input_size, out_channels) for _ in range(depth - 1)]
        self._op = nn.Sequential(*units)

    def forward(self, x: torch.Tensor, pool_after: bool = False, out=None, pool_only: bool = False):
        """pool_after: the caller pools the result next (DownPath) - the last unit then writes AvgPool(result) along with it;
        out: (buffer, first channel) - the last unit writes its result into that channel range of a wider buffer (ops.conv_bn_lrelu);
        pool_only (with pool_after): the caller reads ONLY the pooled result -> returns (result or None, pooled or None)"""
        if not pool_after and out is None and not ops.BLOCKED_Z:
            return self._op(x)                       # (the reference's own call: hooks on the Sequential fire)
        n = len(self._op)
        for k, unit in enumerate(self._op):
            last = k + 1 == n
            if last and pool_only:
                return unit(x, pool_after=pool_after, out=out, pool_only=True)
            # an activation between two units has no reader but the next unit's convolution and weight gradient: where those run the F(2x2x2,3x3x3)
            # kernels it travels in the channel-blocked layout (ops.blocked_z_wanted; a six-dimensional tensor, also in the unit's forward hooks)
            blk = (not last) and x.dim() in (5, 6) and ops.blocked_z_wanted(x, unit._op[0].weight, self._op[k + 1]._op[0].weight, unit.training and self._op[k + 1].training)
            x = unit(x, pool_after=pool_after and last, out=out if last else None, blocked_out=blk)
        return x


class MuSigmaBlock(nn.Module):
    """two 1x1x1 convolutions C -> zdim; sigma = softplus (src/network_blocks.py:49-60).  zdim == ndims (the reference model's setting,
    models.py:88) is one launch of the fused head kernel; any other zdim runs the same kernel over groups of three latent channels."""

    def __init__(self, input_size: Sequence[int], in_channels: int, zdim: int) -> None:
        super().__init__()
        nd = _ndims(input_size, "MuSigmaBlock")
        if zdim < 1:
            raise ValueError("MuSigmaBlock: zdim >= 1 expected")
        self.ndims, self.zdim = nd, zdim
        Conv = nn.Conv3d if nd == 3 else nn.Conv2d
        self._conv_mu = Conv(in_channels, zdim, kernel_size=1)
        self._conv_sigma = nn.Sequential(Conv(in_channels, zdim, kernel_size=1), nn.Softplus())

    def sample(self, x: torch.Tensor, eps: Optional[torch.Tensor]):
        """fused head: (mu, sigma, z = mu + sigma*eps); eps None -> z = mu"""
        cs = self._conv_sigma[0]
        return ops.mu_sigma_sample(x, self._conv_mu.weight, self._conv_mu.bias, cs.weight, cs.bias, eps)

    def forward(self, x: torch.Tensor):
        mu, sigma, _ = self.sample(x, None)
        return [mu, sigma]


class VelocityField(nn.Module):
    """latent sample -> stationary velocity field (src/network_blocks.py:63-85)"""

    def __init__(self, input_size: Sequence[int], zdim: int, max_channels: int, depth: int) -> None:
        super().__init__()
        nd = _ndims(input_size, "VelocityField")
        Conv = nn.Conv3d if nd == 3 else nn.Conv2d
        self.depth = depth
        if depth == 1:
            layers = [Conv(zdim, nd, kernel_size=3)]               # unpadded in the reference (src/network_blocks.py:75)
        elif depth == 0:
            layers = [nn.Identity()]
        else:
            layers = [ConvUnit(input_size, zdim, max_channels)]
            layers += [ConvUnit(input_size, max_channels, max_channels) for _ in range(depth - 2)]
            layers += [Conv(max_channels, nd, kernel_size=1)]
        self._op = nn.Sequential(*layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.depth == 0:
            return x
        if self.depth == 1:
            conv = self._op[0]
            full = ops.conv3d_k3(x, conv.weight, conv.bias)         # 'valid' conv = interior of the zero-padded one
            return (full[:, :, 1:-1, 1:-1, 1:-1] if full.dim() == 5 else full[:, :, 1:-1, 1:-1]).contiguous()
        for unit in list(self._op)[:-1]:
            x = unit(x)
        last = self._op[-1]
        return ops.conv1x1_to3(x, last.weight, last.bias)


class SpatialTransformer(nn.Module):
    """warp `moving_image` by the displacement field `df` (src/network_blocks.py:88-121).

    The sampling grid is implicit in the kernel; the persistent `grid` buffer is kept only because reference
    checkpoints contain it (src/network_blocks.py:99).  `mode` is stored and ignored, as in the reference (:92,:120)."""

    def __init__(self, size, mode="bilinear"):
        super().__init__()
        self.size = size
        self.mode = mode
        _ndims(size, "SpatialTransformer")
        axes = [torch.arange(0, int(s)) for s in size]
        grid = torch.stack(torch.meshgrid(axes, indexing="ij")).unsqueeze(0).to(torch.float32)
        self.register_buffer("grid", grid, persistent=True)

    def forward(self, df: torch.Tensor, moving_image: torch.Tensor) -> torch.Tensor:
        if tuple(df.shape[2:]) != tuple(int(s) for s in self.size):
            raise ValueError(f"SpatialTransformer built for grid {tuple(self.size)} got a field of size {tuple(df.shape[2:])}")
        return ops.warp3d(df, moving_image)


class ResizeTransform(nn.Module):
    """resize a displacement field and rescale its magnitude by the same factor (src/network_blocks.py:124-150)"""

    def __init__(self, vel_resize, ndims):
        super().__init__()
        if ndims not in (2, 3):
            raise NotImplementedError("ResizeTransform: ndims 2 or 3 expected")
        self.factor = 1.0 / vel_resize
        self.mode = "trilinear" if ndims == 3 else "bilinear"

    def out_size(self, x: torch.Tensor):
        return [int(s * self.factor) for s in x.shape[2:]]     # floor(in * scale_factor), as F.interpolate

    def forward(self, x: torch.Tensor, add: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`add` (optional) is summed into the result inside the kernel (the DFAdder of the decoder)"""
        if self.factor == 1:
            return x if add is None else x + add
        # scaling commutes with the (linear) interpolation: one kernel does both orders of the reference (factor < 1: resize, then scale;
        # factor > 1: scale, then resize - network_blocks.py:138-147); coordinates are mapped with 1 / factor, as F.interpolate(scale_factor=)
        return ops.resize_trilinear(x, self.out_size(x), self.factor, add, scale_factor=self.factor)


class DFAdder(nn.Module):
    def forward(self, df1, df2):
        return df1 + df2


class VecInt(nn.Module):
    """scaling and squaring integration of a stationary velocity field (src/network_blocks.py:160-177)"""

    def __init__(self, inshape, nsteps):
        super().__init__()
        assert nsteps >= 0, "nsteps should be >= 0, found: %d" % nsteps
        self.nsteps = nsteps
        self.scale = 1.0 / (2 ** self.nsteps)
        self.transformer = SpatialTransformer(inshape)      # holds the checkpointed grid buffer

    def forward(self, vec: torch.Tensor) -> torch.Tensor:
        return ops.vecint(vec, self.nsteps)
