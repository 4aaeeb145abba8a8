"""PULPo's hierarchical conditional VAE components on the HIP operator set.

Same classes, constructor signatures, return structures and state-dict keys as the reference's
src/components/pulpo.py (DownPath :9-62, Autoencoder :65-215, PULPoEncoder :219-263, SVFDecoder :265-319,
PULPoPrior :323-341); the data flow is re-organised around fused kernels:
  * the six feedback tensors are up-sampled x2 and concatenated by ONE gather kernel (channels-last, 16 channels);
  * mu / sigma / sample come out of one head kernel;
  * ResizeTransform(1/2) + DFAdder is one kernel; VecInt is seven self-warp launches on L2-resident fields.
Multi-channel activations are channels-last (torch.channels_last_3d) so the MFMA convolutions read whole channel
vectors; the tensors handed back to the caller have the reference's logical (B, C, D, H, W) shapes.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..network_blocks import (ConvSequence, DFAdder, FixedNoiseSampler, MuSigmaBlock, ResizeTransform, SpatialTransformer, VecInt,
                              VelocityField, gauss_sampler)
from ..utils import ModuleIntDict

FIELD_ITEMS = ("velocity_fields", "individual_dfs", "combined_dfs", "final_dfs")


def _channel_plan(total_levels: int, n0: int) -> Dict[int, int]:
    """channels per pyramid level: n0 * (1, 2, 4, 6, 6, ...)   (reference pulpo.py:26-28)"""
    mult = [1, 2, 4] + [6] * max(total_levels - 3, 0)
    return {k: n0 * m for k, m in enumerate(mult)}


def _halved(size: Sequence[int]) -> List[int]:
    return [(int(s) + 1) // 2 for s in size]


class DownPath(nn.Module):
    """encoder pyramid: cat(x, y) -> ConvSequence -> [avg-pool -> ConvSequence] * (T-1)"""

    def __init__(self, total_levels: int, latent_levels: int, input_size: Sequence[int], input_channels: int = 2, n0: int = 32) -> None:
        super().__init__()
        self.total_levels = total_levels
        self.latent_levels = latent_levels
        self.lk_offset = total_levels - latent_levels
        self.input_size = input_size
        chans = _channel_plan(total_levels, n0)
        AvgPool = nn.AvgPool3d if len(input_size) == 3 else nn.AvgPool2d
        self.downsample = AvgPool(kernel_size=2, stride=2, padding=0, ceil_mode=True)        # parameter-free; HIP kernel is used
        self.down_blocks = ModuleIntDict()
        for k in range(total_levels):
            cin = input_channels if k == 0 else chans[k - 1]
            self.down_blocks[k] = ConvSequence(input_size=input_size, in_channels=cin, out_channels=chans[k], depth=3)

    def _room(self, k: int, like: torch.Tensor):
        """(buffer, first channel) for level k's activation when a consumer has announced that it will concatenate `_pulpo_skip_room[k]`
        channels IN FRONT of it (PULPoEncoder: torch.cat([feedback, down_activation], 1)): the activation is then produced as the tail of
        that concatenation's buffer and the cat costs nothing (ops.cat_channels).  Set by src.models.PULPo; absent -> plain tensors."""
        room = getattr(self, "_pulpo_skip_room", None)
        if not room or k not in room or like.dim() != 5:
            return None
        cout = self.down_blocks[k]._op[-1]._op[0].out_channels
        B, _, D, H, W = like.shape
        return ops.new_cl(B, int(room[k]) + cout, D, H, W, like.device, ops.act_dtype()), int(room[k])

    def forward(self, x: torch.Tensor, y: torch.Tensor, _needed=None) -> Dict[int, torch.Tensor]:
        """_needed (private, set by src.models.PULPo): the levels whose activation the caller will read.  The reference's Autoencoder never looks at
        the levels above its first latent level (down_activations[k], k < lk_offset: pulpo.py:183-207); for those only the POOLED activation has
        a reader, and with `_needed` given they are neither written (524 MB at 160^3 x 32 channels) nor returned.  Default: every level."""
        h = torch.cat([x, y], dim=1)            # two planar volumes side by side; read in place by the first conv
        acts: Dict[int, torch.Tensor] = {}
        cur = h
        for k in range(self.total_levels):
            pool_next = k + 1 < self.total_levels
            only = bool(_needed is not None and k not in _needed and pool_next and h.dim() == 5)
            if only:
                z, pooled = self.down_blocks[k](cur, pool_after=True, pool_only=True)
                if z is None:                    # (the fused pass took it: the un-pooled tensor does not exist)
                    cur = pooled
                    continue
                acts[k] = z
            else:
                acts[k] = self.down_blocks[k](cur, pool_after=pool_next, out=self._room(k, cur))
            if pool_next:
                # (the activation is pooled AND handed out as a skip connection: one operator, so that its two gradients meet in one kernel)
                acts[k], cur = ops.avg_pool2_skip(acts[k])
        return acts


class PULPoEncoder(nn.Module):
    """posterior head of one latent level: [merge feedback with the encoder activation] -> (mu, sigma) -> sample"""

    def __init__(self, sampler, num_channels: int, zdim: int, input_size: Sequence[int], n0: int = 32) -> None:
        super().__init__()
        self.sampler = sampler
        self.num_channels = num_channels
        self.zdim = zdim
        # constructed on every level like the reference does (pulpo.py:235-240), unused on the coarsest one
        self.sample_merge_block = ConvSequence(input_size=input_size, in_channels=num_channels + n0 * zdim, out_channels=num_channels, depth=2)
        self.mu_sigma = MuSigmaBlock(input_size=input_size, in_channels=num_channels, zdim=zdim)

    def forward(self, down_activation: torch.Tensor, feedback: Optional[torch.Tensor] = None):
        h = down_activation
        if feedback is not None:
            h = self.sample_merge_block(ops.cat_channels(feedback, down_activation))     # (the buffer itself where both were produced into it)
        sampler = self.sampler
        if sampler is gauss_sampler:                         # fused: noise drawn once, sample formed in the head kernel
            eps = torch.randn((h.shape[0], self.zdim) + tuple(h.shape[2:]), device=h.device, dtype=torch.float32)
            return self.mu_sigma.sample(h, eps)
        if isinstance(sampler, FixedNoiseSampler):
            return self.mu_sigma.sample(h, sampler.fixed_eps)
        mu, sigma, _ = self.mu_sigma.sample(h, None)         # user-supplied sampler: the reference's seam (pulpo.py:231,261)
        return mu, sigma, sampler(mu, sigma)


class SVFDecoder(nn.Module):
    """z -> velocity field -> (+ up-scaled coarser field) -> scaling & squaring -> [resize] -> warp"""

    def __init__(self, zdim: int, insize: Sequence[int], outsize: Sequence[int], df_resolution: str, n0: int = 32, cp_depth: int = 3) -> None:
        super().__init__()
        self.zdim = zdim
        self.insize = insize
        self.outsize = outsize
        self.cp_depth = cp_depth
        self.velocity_field = VelocityField(input_size=insize, zdim=zdim, max_channels=n0, depth=cp_depth)
        self.vel_resize_level = 1 / 2
        self.resizer_level = ResizeTransform(self.vel_resize_level, ndims=len(insize))
        self.vel_resize_output = 1 / (outsize[0] / insize[0])           # dim-0 ratio only, as the reference (pulpo.py:290)
        self.resizer_output = ResizeTransform(self.vel_resize_output, ndims=len(insize))
        self.combine_deformation_field = DFAdder()
        self.integrate = VecInt(insize, nsteps=7)
        self.spatial_transform = SpatialTransformer(outsize)

    def forward(self, z: torch.Tensor, input_image: torch.Tensor, combined_df: Optional[torch.Tensor] = None):
        individual_df = self.velocity_field(z)
        if combined_df is None:
            combined = individual_df
        else:
            combined = self.resizer_level(combined_df, add=individual_df)        # 2*up(combined_{l+1}) + individual, one kernel
        integrated = self.resizer_output(self.integrate(combined))
        warped = self.spatial_transform(integrated, input_image)
        # the reference hands back the individual field twice (pulpo.py:319)
        return individual_df, individual_df, combined, integrated, warped


class Autoencoder(nn.Module):
    """coarse-to-fine latent hierarchy with feedback of every coarser level's variables"""

    def __init__(self, sampler, decoder: str, total_levels: int, latent_levels: int, zdim: int, input_size: Sequence[int],
                 feedback: List[str], df_resolution: str, n0: int = 32, cp_depth: int = 3) -> None:
        super().__init__()
        self.sampler = sampler
        self.total_levels = total_levels
        self.latent_levels = latent_levels
        self.lk_offset = total_levels - latent_levels
        self.input_size = input_size
        self.feedback = feedback
        self.df_resolution = df_resolution
        self.cp_depth = cp_depth
        ndims = len(input_size)
        if ndims not in (2, 3):
            raise NotImplementedError("Autoencoder: volumes (ndims 3) or slices (ndims 2) expected")
        if df_resolution not in ("level_res", "full_res"):
            raise ValueError(f"df_resolution is {df_resolution}. Not a known option.")

        self.level_sizes = {0: [int(d) for d in input_size]}
        for k in range(total_levels - 1):
            self.level_sizes[k + 1] = _halved(self.level_sizes[k])
        chans = _channel_plan(total_levels, n0)

        fb_channels = 0
        for item in self.feedback:
            if item == "samples":
                fb_channels += zdim
            elif item == "transformed":
                fb_channels += 1
            elif item in FIELD_ITEMS or item == "control_points":        # control_points: old name of velocity_fields
                fb_channels += ndims
            else:
                raise ValueError(f"Feedback list contains {item}. Not a known option.")

        self.up_blocks = ModuleIntDict()
        for k in range(self.lk_offset, total_levels - 1):
            self.up_blocks[k] = ConvSequence(input_size=input_size, in_channels=fb_channels, out_channels=n0 * zdim, depth=2)

        self.encoders = ModuleIntDict()
        for l in range(latent_levels):
            k = self.lk_offset + l
            self.encoders[l] = PULPoEncoder(sampler=sampler, num_channels=chans[k], zdim=zdim, input_size=self.level_sizes[k], n0=n0)

        if decoder != "SVF":
            raise ValueError(f"Decoder is {decoder}. Not a known option.")
        self.decoder = SVFDecoder
        self.decoders = ModuleIntDict()
        for l in range(latent_levels):
            k = self.lk_offset + l
            full = df_resolution == "full_res"
            self.decoders[l] = SVFDecoder(zdim=zdim, insize=self.level_sizes[k], outsize=input_size if (l == 0 or full) else self.level_sizes[k],
                                          df_resolution=df_resolution, n0=n0, cp_depth=cp_depth)
        self.mode = "trilinear" if ndims == 3 else "bilinear"

    # ------------------------------------------------------------------------------------------------------------
    def _gather_feedback(self, store: Dict[str, Dict[int, torch.Tensor]], level: int, size) -> torch.Tensor:
        srcs = []
        for item in self.feedback:
            name = "velocity_fields" if item == "control_points" else item
            if name not in store:
                raise ValueError(f"Feedback list contains {item}. Not a known option.")
            srcs.append(store[name][level])
        exact_x2 = all(all(int(o) == 2 * int(i) for o, i in zip(size, s.shape[2:])) for s in srcs)
        if exact_x2 and srcs[0].dim() == 5 and sum(s.shape[1] for s in srcs) <= 16 and len(srcs) <= 8:
            return ops.feedback_up2(srcs)
        # ragged pyramid (a size not divisible by 2^(T-1)) or df_resolution='full_res' (final_dfs / transformed arrive at full
        # resolution and are down-sampled): generic resize per tensor, then concatenate
        fmt = torch.channels_last_3d if srcs[0].dim() == 5 else torch.channels_last
        return torch.cat([ops.resize_trilinear(s, size) for s in srcs], dim=1).contiguous(memory_format=fmt)

    def forward(self, x: torch.Tensor, down_activations, deterministic: bool = False):
        L, o = self.latent_levels, self.lk_offset
        # moving image on every latent level (pulpo.py:171-179): level 0 keeps the full-resolution image
        if self.df_resolution == "full_res":
            level_x = {l: x for l in range(L)}
        else:
            level_x = {0: x}
            for _ in range(o):
                level_x[0] = ops.avg_pool2(level_x[0])
            for l in range(1, L):
                level_x[l] = ops.avg_pool2(level_x[l - 1])
            level_x[0] = x

        names = ("mus", "sigmas", "samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed")
        store: Dict[str, Dict[int, torch.Tensor]] = {n: {} for n in names}
        for l in reversed(range(L)):
            k = l + o
            if l == L - 1:
                mu, sigma, z = self.encoders[l](down_activations[k])
                coarser = None
            else:
                fb = self._gather_feedback(store, l + 1, down_activations[k].shape[2:])
                tag = getattr(down_activations[k], "_pulpo_cat", None)       # DownPath left room in front of its activation: write the feedback path there
                if tag is not None:
                    # (a second differentiable pass over the SAME down_activations - several samples from one DownPath call - must not overwrite
                    #  the head the first pass's backward still reads: it takes plain tensors and torch.cat)
                    if getattr(tag[0], "_pulpo_head_taken", False) and torch.is_grad_enabled():
                        tag = None
                    else:
                        tag[0]._pulpo_head_taken = True
                mu, sigma, z = self.encoders[l](down_activations[k], feedback=self.up_blocks[k](fb, out=(tag[0], 0) if tag is not None else None))
                coarser = store["combined_dfs"][l + 1]
            store["mus"][l], store["sigmas"][l], store["samples"][l] = mu, sigma, z
            outs = self.decoders[l](mu if deterministic else z, level_x[l], combined_df=coarser)
            for name, t in zip(names[3:], outs):
                store[name][l] = t
        return tuple(store[n] for n in names)


class PULPoPrior(nn.Module):
    """standard-normal prior on every level (reference pulpo.py:323-341).  The tensors carry a marker so that the KL
    kernel can skip reading them.

    What is handed out are zero-stride views (`expand`) of one cached 0 and one cached 1 per device: no fill kernel per level and step,
    nothing to go stale, and an in-place write by a caller raises (torch refuses to write through an expanded view) instead of silently
    corrupting every later step's prior."""

    def __init__(self) -> None:
        super().__init__()
        self._constants = {}

    def _apply(self, fn, *args, **kwargs):
        self._constants = {}                  # .to() / .cuda(): cached scalars of the old device are dropped
        return super()._apply(fn, *args, **kwargs)

    def __deepcopy__(self, memo):
        return PULPoPrior()

    def forward(self, posterior_mus: Dict[int, torch.Tensor], posterior_sigmas: Dict[int, torch.Tensor]):
        prior_mus, prior_sigmas = {}, {}
        for l in posterior_mus.keys():
            dev = posterior_mus[l].device
            cached = self._constants.get(dev)
            if cached is None:
                cached = self._constants[dev] = (torch.zeros((), device=dev, dtype=torch.float32), torch.ones((), device=dev, dtype=torch.float32))
            mu0, sigma1 = cached[0].expand(posterior_mus[l].shape), cached[1].expand(posterior_sigmas[l].shape)
            mu0._pulpo_std_normal = True
            sigma1._pulpo_std_normal = True
            prior_mus[l], prior_sigmas[l] = mu0, sigma1
        return prior_mus, prior_sigmas
