"""Loss stack of PULPo on HIP kernels, with the reference's public names (src/losses.py).

On the hot path (BASELINE configs: recon_loss=['ncc'], regularizer='L2', diagonal KL):
    KL_two_gauss_with_diag_cov (:47-76), NCC_loss (:85-135), L2_reg (:208-222) and the three Hierarchical*
    wrappers (:225-355).
Alternative hyper-parameters / evaluation metrics, also on HIP kernels (metrics.hip):
    L2_loss (:79-83, `--recon_loss mse`), Soft_dice_loss (:137-145, `--recon_loss dice`), jacobian_det (:172-199),
    JDetStd (:202-204, `--regularizer jdet`) - the 3-D forms.
KL_nondiagonal (:8-44, `--nondiagonal`) likewise.  2-D inputs (train.py --ndims 2) run the reference's 2-D forms through the same
kernels as depth-1 volumes.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from . import ops


def _off_path(name: str):
    raise NotImplementedError(
        f"{name} is outside the MI355X hot path built so far (SURVEY.md §8(f)); the BASELINE configurations "
        "use recon_loss=['ncc'], regularizer='L2' and the diagonal KL.")


class KL_nondiagonal:
    """KL against a prior with a non-diagonal (graph-Laplacian) precision, reference losses.py:8-44: the degree matrix is
    computed inside the kernel instead of being materialised."""

    def __init__(self, inshape, prior_lambda=20) -> None:
        self.prior_lambda = prior_lambda
        self.inshape = [int(s) for s in inshape]
        self.ndims = len(self.inshape)
        if self.ndims not in (2, 3):
            _off_path("KL_nondiagonal (ndims %d)" % self.ndims)

    def loss(self, prior_mean, prior_sigma, flow_mean, flow_sigma):
        return ops.kl_nondiagonal(flow_mean, flow_sigma, self.prior_lambda)


def L2_loss(input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """spatial sum of squared differences, mean over batch and channels (reference losses.py:79-83)"""
    return ops.l2_loss(input, target)


def Soft_dice_loss(input: torch.Tensor, target: torch.Tensor, dice_factor=1) -> torch.Tensor:
    """mean over (batch, channel) of (1 - soft dice), times voxels / dice_factor (reference losses.py:137-145)"""
    return ops.soft_dice_loss(input, target, dice_factor)


def jacobian_det(deformation_field: torch.Tensor, lamb=None, normalize=True) -> torch.Tensor:
    """(B,3,D,H,W) -> (B,D,H,W) Jacobian determinant (reference losses.py:172-199); the 2-D form has no HIP path"""
    if deformation_field.dim() not in (4, 5):
        _off_path("jacobian_det")
    return ops.jacobian_det(deformation_field, normalize)


def JDetStd(deformation_field: torch.Tensor, lamb=0, normalize=True) -> torch.Tensor:
    """lamb * std of the Jacobian determinant (reference losses.py:202-204), differentiable"""
    if deformation_field.dim() not in (4, 5):
        _off_path("JDetStd")
    return ops.jdet_std(deformation_field, lamb, normalize)


def _is_std_normal(mu1, sigma1) -> bool:
    return getattr(mu1, "_pulpo_std_normal", False) and getattr(sigma1, "_pulpo_std_normal", False)


def KL_two_gauss_with_diag_cov(mu0: torch.Tensor, sigma0: torch.Tensor, mu1: torch.Tensor, sigma1: torch.Tensor, eps: float = 1e-10) -> torch.Tensor:
    """KL[p0 || p1] for diagonal Gaussians: sum over features, mean over the batch.  eps is fixed at the reference's 1e-10."""
    if eps != 1e-10:
        raise NotImplementedError("KL_two_gauss_with_diag_cov: the HIP kernel uses the reference's eps = 1e-10")
    if _is_std_normal(mu1, sigma1):
        return ops.kl_diag(mu0, sigma0, None, None)
    return ops.kl_diag(mu0, sigma0, mu1, sigma1)


def NCC_loss(y_pred: torch.Tensor, y_true: torch.Tensor, win_size: int = 9, gamma: float = 0.05) -> torch.Tensor:
    """-gamma * sum_voxels mean_batch(local squared NCC) with a win_size^3 zero-padded window"""
    if y_pred.dim() not in (4, 5):
        raise NotImplementedError("NCC_loss: volumes (B,1,D,H,W) or slices (B,1,H,W) expected")
    return ops.ncc_loss(y_pred, y_true, win_size, gamma)


def L2_reg(deformation_field: torch.Tensor, lamb=0) -> torch.Tensor:
    """lamb * H*W*D * mean of squared forward differences over the [1:,1:,1:] block"""
    if deformation_field.dim() not in (4, 5):
        raise NotImplementedError("L2_reg: fields (B,3,D,H,W) or (B,2,H,W) expected")
    return ops.l2_reg(deformation_field, lamb)


def _apply_pyramid(weight_dict: Dict[int, float], similarity_pyramid: bool) -> Dict[int, float]:
    if similarity_pyramid:                      # in place, like the reference (losses.py:238-240)
        for l in weight_dict.keys():
            weight_dict[l] = weight_dict[l] / 2 ** l
    return weight_dict


def _weighted(weight_dict: Dict[int, float], terms: Dict[int, torch.Tensor], scale=None):
    """(sum_l w_l * term_l, {l: w_l * term_l}) - losses.py:262-276 / 305-325 / 343-355.  Device scalars: one kernel for the whole sum
    (ops.weighted_sum, bit-identical to the reference's per-level mul / add chain); anything else: that chain itself."""
    levels = list(weight_dict.keys())
    vals = [terms[l] for l in levels]
    if all(isinstance(v, torch.Tensor) and v.is_cuda and v.numel() == 1 and v.dtype == torch.float32 for v in vals):
        total, per_level = ops.weighted_sum(vals, [weight_dict[l] for l in levels], scale)
        return total, dict(zip(levels, per_level))
    total, all_levels = 0.0, {}
    for l in levels:
        all_levels[l] = weight_dict[l] * terms[l]
        total = total + all_levels[l]
    if scale is not None:
        total = total * scale
        all_levels = {l: scale * v for l, v in all_levels.items()}
    return total, all_levels


class HierarchicalKLLoss(nn.Module):
    """sum_l w_l * KL_l; also returns the per-level terms (losses.py:225-276)"""

    def __init__(self, KL_divergence, weight_dict: Dict[int, float], similarity_pyramid: bool, level_sizes: Dict[int, torch.Tensor] = None) -> None:
        super().__init__()
        self.weight_dict = _apply_pyramid(weight_dict, similarity_pyramid)
        self.KL_divergence = KL_divergence
        if KL_divergence == KL_nondiagonal:          # one instance per level (reference losses.py:242-243)
            self.KL_divergence = {key: KL_nondiagonal(inshape=level_sizes[key]).loss for key in level_sizes.keys()}

    def forward(self, prior_mus, prior_sigmas, posterior_mus, posterior_sigmas, scale=None):
        """scale (not in the reference's signature; models.py applies `* beta` to the results, :161-162): folded into the same launch"""
        assert self.weight_dict.keys() == prior_mus.keys()
        assert prior_mus.keys() == prior_sigmas.keys() == posterior_mus.keys() == posterior_sigmas.keys()
        terms = {}
        for l in self.weight_dict:
            if isinstance(self.KL_divergence, dict):   # argument order of the reference for the non-diagonal class (losses.py:267-269)
                terms[l] = self.KL_divergence[l](prior_mus[l], prior_sigmas[l], posterior_mus[l], posterior_sigmas[l])
            else:
                terms[l] = self.KL_divergence(posterior_mus[l], posterior_sigmas[l], prior_mus[l], prior_sigmas[l])
        return _weighted(self.weight_dict, terms, scale)


class HierarchicalReconstructionLoss(nn.Module):
    """sum_l w_l * recon(y_hat_l, y resized to level l) / len(recon_loss)   (losses.py:279-325)"""

    def __init__(self, recon_loss: List[str], weight_dict: Dict[int, float], similarity_pyramid: bool, ndims: int, window_size: Dict[int, float]) -> None:
        super().__init__()
        self.recon_loss = recon_loss
        self.weight_dict = _apply_pyramid(weight_dict, similarity_pyramid)
        self.window_size = window_size
        self.ndims = ndims
        self.mode = "trilinear" if ndims == 3 else "bilinear"

    def forward(self, y_hat, y, y_hat_seg=None, seg_y=None, gamma: float = 0.05, dice_factor: int = 1):
        single = len(self.recon_loss) == 1 and self.recon_loss[0] in ("mse", "ncc", "dice")
        loss = 0.0
        all_levels, terms = {}, {}
        for l, w in self.weight_dict.items():
            size = y_hat[l].shape[2:]
            # F.interpolate(y, size) of the reference; the identity resize at full resolution is skipped
            y_target = y if tuple(size) == tuple(y.shape[2:]) else ops.resize_trilinear(y, size)
            if single and self.recon_loss[0] != "dice":
                # one term per level (the default, ["ncc"]): weighting and summation of all levels in one launch; x / 1 is x
                terms[l] = (NCC_loss(y_hat[l], y_target, gamma=gamma, win_size=self.window_size[l]) if self.recon_loss[0] == "ncc"
                            else L2_loss(y_hat[l], y_target))
                continue
            term = 0.0
            if "mse" in self.recon_loss:
                term = term + w * L2_loss(y_hat[l], y_target)
            if "ncc" in self.recon_loss:
                term = term + w * NCC_loss(y_hat[l], y_target, gamma=gamma, win_size=self.window_size[l])
            if "dice" in self.recon_loss:
                seg_size = y_hat_seg[l].shape[2:]
                seg_target = seg_y if tuple(seg_size) == tuple(seg_y.shape[2:]) else ops.resize_trilinear(seg_y, seg_size)
                term = term + w * Soft_dice_loss(y_hat_seg[l], seg_target, dice_factor=dice_factor)
            all_levels[l] = term / len(self.recon_loss)
            loss = loss + all_levels[l]
        if terms:
            return _weighted(self.weight_dict, terms)
        return loss, all_levels


class HierarchicalRegularization(nn.Module):
    """sum_l w_l * regularizer(df_l, lamb)   (losses.py:327-355)"""

    def __init__(self, regularizer, weight_dict: Dict[int, float], similarity_pyramid: bool) -> None:
        super().__init__()
        self.regularizer = regularizer
        self.weight_dict = _apply_pyramid(weight_dict, similarity_pyramid)

    def forward(self, dfs: Dict[int, torch.Tensor], lamb: float = 0):
        assert self.weight_dict.keys() == dfs.keys()
        return _weighted(self.weight_dict, {l: self.regularizer(dfs[l], lamb) for l in self.weight_dict})
