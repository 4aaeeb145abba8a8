"""PULPo LightningModule on the MI355X hot path (drop-in for the reference's src/models.py:24-400).

Same constructor signature, hyper-parameter names, sub-module names (downpath / autoencoder / prior -> identical
state-dict keys), step / predict API and logged metric names as the reference.  Conscious deviations, all outside the
arithmetic:
  * torch.autograd.set_detect_anomaly(True) (reference models.py:50) is NOT switched on: it is a global debug mode.
  * the NaN guard (reference models.py:188-194) looks at the previous step's regularisation terms instead of
    synchronising the stream once per level inside the current step; it trips one step later.
  * image logging (reference models.py:258-308) needs torchvision + a TensorBoard logger and is skipped without them.
"""
from __future__ import annotations

import math
import os
from abc import ABC
from typing import Dict, List, Optional

import torch

from ._lightning import LightningModule
from .components.pulpo import Autoencoder, DownPath, PULPoEncoder, PULPoPrior, SVFDecoder  # noqa: F401  (re-exported like the reference)
from .losses import (HierarchicalKLLoss, HierarchicalReconstructionLoss, HierarchicalRegularization, JDetStd, KL_nondiagonal,
                     KL_two_gauss_with_diag_cov, L2_reg)
from .network_blocks import DFAdder, ResizeTransform, SpatialTransformer, VecInt, gauss_sampler  # noqa: F401
from . import ops


class _DoneEvent:
    def query(self) -> bool:
        return True


class PULPo(ABC, LightningModule):

    def __init__(
        self,
        total_levels: int,
        latent_levels: int,
        beta: float,
        input_size: List[int],
        lr: float = 1e-4,
        recon_loss: list = ["ncc"],
        dice_factor: int = 1,
        similarity_pyramid: bool = False,
        lamb: float = 0.025,
        gamma: float = 0.05,
        regularizer: str = "L2",
        image_logging_frequency: int = 1000,
        feedback: list = ["samples", "velocity_field", "individual_dfs", "combined_dfs", "final_dfs", "transformed"],
        df_resolution: str = "level_res",
        n0: int = 32,
        segs: bool = False,
        lms: bool = False,
        mask: bool = False,
        nondiagonal: bool = False,
        cp_depth: int = 3,
    ) -> None:
        super().__init__()
        self.validation_counter = 0
        self.save_hyperparameters()
        self.segs, self.lms, self.mask = segs, lms, mask
        self.latent_levels = latent_levels
        self.total_levels = total_levels
        self.lk_offset = total_levels - latent_levels
        self.beta = beta
        self.df_resolution = df_resolution
        self.recon_loss = recon_loss
        self.input_size = input_size
        self.ndims = len(input_size)
        self.cp_depth = cp_depth
        # floor division here, ceil in Autoencoder.level_sizes: reference quirk (models.py:69 vs pulpo.py:95)
        self.level_sizes = {l: torch.tensor(self.input_size) // (2 ** (l + self.lk_offset)) for l in range(latent_levels)}

        self.df_combiner = DFAdder()
        self.prior = PULPoPrior()
        self.downpath = DownPath(total_levels=total_levels, latent_levels=latent_levels, input_size=input_size, input_channels=2, n0=n0)
        # NB: the shipped default feedback list holds 'velocity_field', which Autoencoder rejects exactly like the
        # reference does; pass 'velocity_fields'.
        self.autoencoder = Autoencoder(sampler=gauss_sampler, decoder="SVF", total_levels=total_levels, latent_levels=latent_levels,
                                       zdim=self.ndims, input_size=input_size, feedback=self.hparams.feedback,
                                       df_resolution=self.hparams.df_resolution, n0=n0, cp_depth=cp_depth)

        # the encoders concatenate the feedback path's output IN FRONT of the DownPath activation of their level (components/pulpo.py): DownPath
        # produces those activations as the tail of the concatenation's buffer, the feedback path writes the head (ops.cat_channels).
        # PULPO_PREWRITTEN_CAT=0: plain torch.cat (A/B switch).
        if os.environ.get("PULPO_PREWRITTEN_CAT", "1") != "0" and self.ndims == 3:
            self.downpath._pulpo_skip_room = {int(k): int(blk._op[-1]._op[0].out_channels) for k, blk in self.autoencoder.up_blocks.items()}

        # the Autoencoder reads down_activations[k] for k >= lk_offset only (components/pulpo.py): the levels above are not materialised
        self._needed_levels = frozenset(range(self.lk_offset, total_levels)) if os.environ.get("PULPO_SKIP_UNUSED_ACTIVATIONS", "1") != "0" else None

        if self.hparams.regularizer == "jdet":
            regularization_loss = JDetStd
        elif self.hparams.regularizer == "L2":
            regularization_loss = L2_reg
        else:
            raise ValueError(f"Hyperparameter regularizer is {self.hparams.regularizer}. Not a known option.")

        # NCC window per level and the loss-magnitude equalisation weights (reference models.py:104-123)
        window_size = {l: 1 + 2 * (latent_levels - l) for l in range(latent_levels)}
        if latent_levels == 1:
            window_size = {0: 9}
        scale = {l: (2.0 ** self.ndims) ** l for l in range(latent_levels)}
        kl_w = dict(scale)
        if df_resolution == "full_res":
            rec_w = {l: 1.0 for l in range(latent_levels)}
            reg_w = {l: 1.0 for l in range(latent_levels)}
        else:
            rec_w, reg_w = dict(scale), dict(scale)
            full_to_level0 = 2 ** (self.ndims * self.lk_offset)      # level 0 is evaluated at full resolution
            rec_w[0] = scale[0] / full_to_level0
            reg_w[0] = scale[0] / full_to_level0
        rec_w[0] *= 4
        self.window_size = window_size

        kl_fn = KL_nondiagonal if nondiagonal else KL_two_gauss_with_diag_cov
        self.hierarchical_kl_loss = HierarchicalKLLoss(KL_divergence=kl_fn, weight_dict=kl_w, similarity_pyramid=similarity_pyramid,
                                                       level_sizes=self.level_sizes)
        self.hierarchical_recon_loss = HierarchicalReconstructionLoss(recon_loss=recon_loss, weight_dict=rec_w,
                                                                      similarity_pyramid=similarity_pyramid, window_size=window_size,
                                                                      ndims=self.ndims)
        self.hierarchical_regularization = HierarchicalRegularization(regularizer=regularization_loss, weight_dict=reg_w,
                                                                      similarity_pyramid=similarity_pyramid)
        self._pending_nan_probe = None          # (pinned flag, event) of an earlier step's NaN test

    # ------------------------------------------------------------------------------------------------ steps
    def _forward_and_losses(self, x, y, seg_x=None, seg_y=None):
        acts = self.downpath(x, y, _needed=self._needed_levels)
        outs = self.autoencoder(x, acts)
        mus, sigmas, samples, velocity_fields, individual_dfs, combined_dfs, final_dfs, y_hat = outs
        prior_mus, prior_sigmas = self.prior(mus, sigmas)
        if "dice" in self.hparams.recon_loss:
            y_hat_seg = self.transform_segmentation(final_dfs, seg_x)
        else:
            y_hat_seg = {k: None for k in final_dfs}
        kl, kl_levels = self.hierarchical_kl_loss(prior_mus, prior_sigmas, mus, sigmas, scale=self.beta)       # (kl * beta, beta * levels: models.py:161-162)
        rec, rec_levels = self.hierarchical_recon_loss(y_hat, y, y_hat_seg, seg_y, gamma=self.hparams.gamma, dice_factor=self.hparams.dice_factor)
        reg, reg_levels = self.hierarchical_regularization(final_dfs, lamb=self.hparams.lamb)
        total = kl + rec + reg
        return outs, (prior_mus, prior_sigmas), (total, kl, rec, reg), (kl_levels, rec_levels, reg_levels)

    def _logging_active(self) -> bool:
        """is anybody going to read what log_dict receives?  (a Lightning trainer, or a logger attached to the stand-alone base)"""
        if getattr(self, "logger", None) is not None:
            return True
        tr = getattr(self, "_trainer", None)                       # pl.LightningModule keeps the attached trainer here
        return tr is not None

    def _log_levels(self, stage: str, mus, sigmas, priors, levels, **kw):
        if not self._logging_active():
            return                    # the 16 per-level mean reductions are only computed when they have a consumer
        kl_l, rec_l, reg_l = levels
        for level in kl_l.keys():
            with torch.no_grad():
                stats = {
                    f"{stage}_distribution_levels/mean_prior_mu_{level}": torch.mean(priors[0][level]),
                    f"{stage}_distribution_levels/mean_prior_sigma_{level}": torch.mean(priors[1][level]),
                    f"{stage}_distribution_levels/mean_posterior_mu_{level}": torch.mean(mus[level]),
                    f"{stage}_distribution_levels/mean_posterior_sigma_{level}": torch.mean(sigmas[level]),
                }
            self.log_dict({
                f"{stage}_levels/kl loss level {level}": kl_l[level],
                f"{stage}_levels/recon loss level {level}": rec_l[level],
                f"{stage}_levels/regularization loss level {level}": reg_l[level],
                **stats}, **kw)

    def _check_previous_step_for_nan(self):
        """the reference tests the regularisation terms for NaN inside the step with a host synchronisation (models.py:188-192); here the
        flag of step n travels to pinned host memory asynchronously and is looked at when its copy has completed (normally at step n+1),
        so the host never waits for the GPU"""
        pending = self._pending_nan_probe
        if pending is None:
            return
        flag, event = pending
        if not event.query():
            return                    # not there yet: look again next step
        self._pending_nan_probe = None
        if bool(flag.item()):
            print("NAN IN REGULARIZATION LOSS")
            torch.save(self.state_dict(), "nan_state_dict.pt")
            self.trainer.should_stop = True

    def _arm_nan_probe(self, reg_levels) -> None:
        if self._pending_nan_probe is not None:
            return                    # the previous flag has not been consumed yet
        bad = torch.isnan(torch.stack([v.detach() for v in reg_levels.values()])).any()
        if bad.is_cuda:
            flag = torch.empty((), dtype=torch.bool, pin_memory=True)
            flag.copy_(bad, non_blocking=True)
            event = torch.cuda.Event()
            event.record()
        else:
            flag, event = bad, _DoneEvent()
        self._pending_nan_probe = (flag, event)

    def training_step(self, batch, batch_idx):
        x, y, seg_x, seg_y, lm1, lm2, mask1, mask2 = batch
        in_graph = getattr(self, "_pulpo_in_graph", False)      # dp.DataParallelStepper(graph=True) is capturing this call into a HIP graph:
        if not in_graph:                                        # the host-side NaN probe (pinned memory, events) stays outside, driven by the stepper
            self._check_previous_step_for_nan()
        eng = self._engine()
        if eng is not None and torch.is_grad_enabled():
            eng.arm(not self._ddp_wrapped())          # multi-rank: this forward pass registers the triggers of the bucketed gradient exchange
        outs, priors, (total, kl, rec, reg), levels = self._forward_and_losses(x, y, seg_x, seg_y)
        self.log_dict({"train/kl_loss": kl, "train/reconstruction_loss": rec, "train/regularization_loss": reg, "train/total_loss": total},
                      on_step=True, on_epoch=True, prog_bar=True)
        self._log_levels("train", outs[0], outs[1], priors, levels, on_step=False, on_epoch=True)
        if in_graph:
            self._pulpo_last_reg_levels = {k: v.detach() for k, v in levels[2].items()}
        else:
            self._arm_nan_probe(levels[2])
        return total

    def validation_step(self, batch, batch_idx):
        if batch_idx == self.trainer.num_val_batches[0] - 1:
            self.validation_counter += 1
        x, y, seg_x, seg_y, lm1, lm2, mask1, mask2 = batch
        outs, priors, (total, kl, rec, reg), levels = self._forward_and_losses(x, y, seg_x, seg_y)
        self.log_dict({"val/kl_loss": kl, "val/reconstruction_loss": rec, "val/regularization_loss": reg, "val/total_loss": total}, on_epoch=True)
        self._log_levels("val", outs[0], outs[1], priors, levels, on_epoch=True)
        return total

    # ------------------------------------------------------------------------------------------------ inference
    def predict_output_samples(self, x: torch.Tensor, y: torch.Tensor, N: int = 1):
        """N stochastic forward passes stacked on the batch axis -> ({l: (B,N,1,...)}, {l: (B,N,3,...)})"""
        bs = x.shape[0]
        xb, yb = torch.cat([x] * N, dim=0), torch.cat([y] * N, dim=0)          # rank-agnostic (3-D volumes and 2-D slices), models.py:314-315
        outs = self.autoencoder(xb, self.downpath(xb, yb))
        individual_dfs, outputs = outs[4], outs[7]
        fold = lambda t: t.view([N, bs] + list(t.shape[1:])).transpose(0, 1)
        return {k: fold(v) for k, v in outputs.items()}, {k: fold(v) for k, v in individual_dfs.items()}

    def predict(self, x: torch.Tensor, y: torch.Tensor, N: int = 1):
        _, individual_dfs = self.predict_output_samples(x, y, N)
        avg_dfs = {k: v.mean(dim=1) for k, v in individual_dfs.items()}
        _, avg_final = self.combine_dfs(avg_dfs)
        # every level warps the FULL-resolution moving image (grid smaller than the image for l >= 1, models.py:330)
        avg_outputs = {k: self.autoencoder.decoders[k].spatial_transform(avg_final[k], x) for k in avg_final}
        return avg_outputs, avg_dfs

    def predict_deterministic(self, x: torch.Tensor, y: torch.Tensor):
        outs = self.autoencoder(x, self.downpath(x, y, _needed=self._needed_levels), deterministic=True)
        return outs[7], outs[4]

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        return self.autoencoder(x, self.downpath(x, y, _needed=self._needed_levels))[7][0]

    # ------------------------------------------------------------------------------------------------ helpers
    def combine_dfs(self, individual_dfs: Dict[int, torch.Tensor]):
        """individual level fields -> (combined, integrated + resized) fields (reference models.py:349-368)"""
        combined, final = {}, {}
        for l in reversed(range(self.latent_levels)):
            if l + 1 in combined:
                ratio = individual_dfs[l].shape[2] / individual_dfs[l + 1].shape[2]
                combined[l] = ResizeTransform(vel_resize=1 / ratio, ndims=self.ndims)(combined[l + 1], add=individual_dfs[l])
            else:
                combined[l] = individual_dfs[l]
        for l in reversed(range(self.latent_levels)):
            integrated = ops.vecint(combined[l], 7)
            target = self.input_size if (l == 0 or self.hparams.df_resolution == "full_res") else combined[l].shape[2:]
            final[l] = ResizeTransform(vel_resize=1 / (target[0] / integrated.shape[2]), ndims=self.ndims)(integrated)
        return combined, final

    def transform_segmentation(self, dfs: Dict[int, torch.Tensor], seg: torch.Tensor):
        """warp the (pooled) segmentation maps with each level's field (reference models.py:370-388)"""
        if self.df_resolution == "full_res":        # every level's field is full resolution: every level warps the full map (models.py:375-376)
            level_seg = {l: seg for l in range(self.latent_levels)}
        else:
            level_seg = {0: seg}
            for _ in range(self.lk_offset):
                level_seg[0] = ops.avg_pool2(level_seg[0])
            for l in range(1, self.latent_levels):
                level_seg[l] = ops.avg_pool2(level_seg[l - 1])
            level_seg[0] = seg
        return {k: self.autoencoder.decoders[k].spatial_transform(dfs[k], level_seg[k]) for k in dfs}

    # ------------------------------------------------------------------------------------------------ optimizer + Lightning's backward hooks
    # The reference leaves the step to Lightning: `return total_loss` (models.py:196) -> Lightning's closure runs optimizer_zero_grad,
    # on_before_backward, backward, on_after_backward and hands the closure to `optimizer.step` of what configure_optimizers returned
    # (models.py:398-400; pytorch_lightning 1.8 loops/optimization/optimizer_loop.py).  The overrides below make that loop run the step
    # bench.py times: parameter gradients straight into the flat arena, weight gradients on the side stream, one finishing launch, fused
    # Adam, in-place re-pack of the weights - the same pieces `dp.DataParallelStepper.step` strings together.  PULPO_LIGHTNING_FAST=0
    # switches all of it off (plain autograd + torch.optim.Adam).
    def configure_optimizers(self):
        first = next(self.parameters())
        if not first.is_cuda or os.environ.get("PULPO_LIGHTNING_FAST", "1") == "0":
            self._pulpo_optimizer = None
            if not first.is_cuda:
                # a strategy that builds its optimizers BEFORE it moves the model to the device would end up here for good: say so instead of
                # silently handing it the slow loop (plain autograd + torch.optim.Adam, every operator still a HIP kernel)
                import warnings
                warnings.warn("pulpo_amd: configure_optimizers() was called while the parameters are not on a GPU - returning plain torch.optim.Adam "
                              "(no flat arena, no fused update, no direct-to-arena gradients).  Move the model to the device first and call "
                              "configure_optimizers() again for the fast step.", RuntimeWarning, stacklevel=2)
            return torch.optim.Adam(self.parameters(), lr=self.hparams.lr)
        from .dp import ArenaAdam
        opt = ArenaAdam(self, lr=float(self.hparams.lr))
        self._pulpo_optimizer = opt                # (a plain attribute: neither a sub-module nor part of the state dict)
        return opt

    def _engine(self):
        opt = getattr(self, "_pulpo_optimizer", None)
        return None if opt is None else opt.engine

    def _ddp_wrapped(self) -> bool:
        """is a DistributedDataParallel wrapper (Lightning's ddp strategies) reducing the gradients?  It needs them from autograd's
        AccumulateGrad nodes, so the direct-to-arena path is off and the optimizer neither exchanges nor rescales them."""
        tr = getattr(self, "_trainer", None)           # (pytorch_lightning >= 1.8: `trainer` is a property that RAISES while no Trainer is attached)
        if tr is None:
            try:
                tr = getattr(self, "trainer", None)
            except RuntimeError:
                tr = None                              # no Trainer: nothing wraps the module
        wrapped = getattr(getattr(tr, "strategy", None), "model", None)
        return isinstance(wrapped, torch.nn.parallel.DistributedDataParallel)

    def optimizer_zero_grad(self, epoch, batch_idx, optimizer, *args, **kwargs):
        optimizer.zero_grad()                      # (ArenaAdam: one fill, the gradient views stay attached)

    def backward(self, loss, *args, **kwargs):
        eng = self._engine()
        if eng is None:
            return loss.backward(*[a for a in args if torch.is_tensor(a)], **kwargs)      # (PL 1.x passes (optimizer, optimizer_idx) first)
        ddp = self._ddp_wrapped()
        self._pulpo_optimizer.reduced_elsewhere = ddp
        eng.backward(loss, direct=not ddp)

