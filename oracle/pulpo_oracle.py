"""CPU oracle for the PULPo registration hot path.  TEST INFRASTRUCTURE ONLY.

A functional, state-dict-driven restatement (torch fp32 on the CPU) of the path
BASELINE.json:north_star names, written from the reference's behaviour:

    /root/reference/src/network_blocks.py   (ConvUnit, MuSigmaBlock, VelocityField, SpatialTransformer,
                                             ResizeTransform, VecInt, gauss_sampler)
    /root/reference/src/components/pulpo.py (DownPath, Autoencoder, PULPoEncoder, SVFDecoder, PULPoPrior)
    /root/reference/src/losses.py           (KL_two_gauss_with_diag_cov, NCC_loss, L2_reg, Hierarchical*)
    /root/reference/src/models.py           (loss-weight dictionaries, training_step arithmetic)

Each function cites the reference lines it follows.  It consumes the reference's
own state-dict key names, so loading a reference checkpoint into it is the
checkpoint-compatibility test.

PINNING: tests/test_oracle_golden.py checks every function here against the
fixtures under tests/golden/, which tests/golden/make_golden.py produced by
importing the real reference modules in the build container.

WHO MAY IMPORT THIS: tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, as the checker / reported CPU baseline.  Nothing under
pulpo_amd/ imports it; the product path has no CPU fallback.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

FEEDBACK_DEFAULT = ("samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed")


# =============================================================================== configuration
@dataclass
class Cfg:
    total_levels: int
    latent_levels: int
    input_size: Sequence[int]
    n0: int = 32
    feedback: Sequence[str] = FEEDBACK_DEFAULT
    cp_depth: int = 3
    zdim: int = 3
    beta: float = 0.1
    gamma: float = 0.05
    lamb: float = 0.025
    df_resolution: str = "level_res"

    @property
    def offset(self) -> int:                       # lk_offset, pulpo.py:22,86
        return self.total_levels - self.latent_levels

    def channels(self) -> List[int]:               # pulpo.py:26-28
        mult = [1, 2, 4] + [6] * max(self.total_levels - 3, 0)
        return [self.n0 * m for m in mult[: self.total_levels]]

    def level_sizes(self) -> List[List[int]]:      # pulpo.py:93-96 (ceil halving)
        sizes = [list(self.input_size)]
        for _ in range(self.total_levels - 1):
            sizes.append([(s + 1) // 2 for s in sizes[-1]])
        return sizes


def weight_tables(cfg: Cfg) -> Tuple[Dict[int, int], Dict[int, float], Dict[int, float], Dict[int, float]]:
    """NCC window sizes and the three per-level loss-weight dictionaries (models.py:104-123,
    df_resolution='level_res', similarity_pyramid=False)."""
    L, nd, o = cfg.latent_levels, len(cfg.input_size), cfg.offset
    window = {l: 1 + 2 * (L - l) for l in range(L)}
    if L == 1:
        window = {0: 9}
    scale = {l: float((2.0 ** nd) ** l) for l in range(L)}
    kl_w, rec_w, reg_w = dict(scale), dict(scale), dict(scale)
    if cfg.df_resolution == "full_res":               # every level is evaluated at full resolution (models.py:112-115)
        rec_w = {l: 1.0 for l in range(L)}
        reg_w = {l: 1.0 for l in range(L)}
    else:
        rec_w[0] = scale[0] / (2 ** (nd * o))
        reg_w[0] = scale[0] / (2 ** (nd * o))
    rec_w[0] *= 4
    return window, kl_w, rec_w, reg_w


# =============================================================================== elementary operators
# bf16-operand mode (BASELINE configs 4-5).  The reference has no reduced-precision mode, so this is a
# DEFINITION, not a restatement - "parity unpinned" against the reference: each 3x3x3 convolution product (forward, data
# gradient, weight gradient) takes its two operands rounded to bf16 (round-to-nearest-even) and accumulates in fp32;
# GEMMs with <= 4 reduction channels (the image / latent inputs) stay fp32.  Everything else is fp32 as in the reference.
CONV_PRECISION = "fp32"
# bf16 ACTIVATION STORAGE (configs 4-5, on top of the bf16 operands; pulpo_amd.ops.ACT_BF16).  Also a definition ("parity unpinned"): the
# multi-channel activation tensors are ROUNDED to bf16 where the product path stores them - a ConvUnit's pre-norm output y when the
# bf16-operand kernel produces it (more than 4 reduction channels), every ConvUnit output z, pooled feature maps, the concatenated
# up-sampled feedback - and so are the gradients that arrive at those tensors in the backward pass; all arithmetic in between is fp32
# (BatchNorm statistics are those of y AS ROUNDED).  Images, latents, fields, losses, parameters and their gradients are never rounded.
ACT_PRECISION = "fp32"


def _rb(t: Tensor) -> Tensor:
    return t.bfloat16().to(t.dtype)


class _StoreBf16(torch.autograd.Function):
    """a tensor that lives in HBM as bf16: the value is rounded on the way forward, its gradient on the way back"""

    @staticmethod
    def forward(ctx, x):
        return _rb(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


def stored(t: Tensor) -> Tensor:
    """identity in fp32 storage; round-trip through bf16 (value and gradient) under ACT_PRECISION == 'bf16'"""
    return _StoreBf16.apply(t) if (ACT_PRECISION == "bf16" and CONV_PRECISION == "bf16") else t


class _ConvBf16Operands(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        xin, win = (_rb(x), _rb(w)) if w.shape[1] > 4 else (x, w)
        return F.conv3d(xin, win, b, stride=1, padding=1)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyr = _rb(dy)
        gx = gw = None
        if ctx.needs_input_grad[0]:                      # data gradient: K = Cout channels
            gx = torch.nn.grad.conv3d_input(x.shape, _rb(w) if w.shape[0] > 4 else w, dyr if w.shape[0] > 4 else dy, stride=1, padding=1)
        if ctx.needs_input_grad[1]:                      # weight gradient: bf16 operands when Cin > 4
            big = w.shape[1] > 4
            gw = torch.nn.grad.conv3d_weight(_rb(x) if big else x, w.shape, dyr if big else dy, stride=1, padding=1)
        gb = dy.sum(dim=(0, 2, 3, 4)) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def conv3_k3(h: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """the padded 3x3x3 convolution of ConvUnit in the active operand precision"""
    if CONV_PRECISION == "bf16":
        return _ConvBf16Operands.apply(h, w, b)
    return F.conv3d(h, w, b, stride=1, padding=1)


def conv_unit(h: Tensor, sd: Dict[str, Tensor], prefix: str, training: bool) -> Tensor:
    """Conv3d(k3,p1,bias) -> BatchNorm3d(eps 1e-5, momentum 0.1) -> LeakyReLU(0.2)   network_blocks.py:22-26.
    `prefix` names the ConvUnit ('..._op.0'); its children are '_op.0' (conv) and '_op.1' (bn).
    In training mode the running statistics in `sd` are updated in place, as nn.BatchNorm3d does."""
    w, b = sd[prefix + "._op.0.weight"], sd[prefix + "._op.0.bias"]
    h = conv3_k3(h, w, b)
    if w.shape[1] > 4:
        h = stored(h)                    # the pre-norm tensor y (fp32 behind the exact-fp32 kernel of the <= 4-channel input layers)
    rm, rv = sd[prefix + "._op.1.running_mean"], sd[prefix + "._op.1.running_var"]
    if training:
        nbt = sd.get(prefix + "._op.1.num_batches_tracked")
        if nbt is not None:
            nbt += 1
    h = F.batch_norm(h, rm, rv, sd[prefix + "._op.1.weight"], sd[prefix + "._op.1.bias"], training=training, momentum=0.1, eps=1e-5)
    return stored(F.leaky_relu(h, 0.2))


def conv_sequence(h: Tensor, sd, prefix: str, depth: int, training: bool) -> Tensor:
    """ConvSequence = `depth` ConvUnits, keys '<prefix>._op.<i>'   network_blocks.py:41-43"""
    for i in range(depth):
        h = conv_unit(h, sd, f"{prefix}._op.{i}", training)
    return h


def mu_sigma(h: Tensor, sd, prefix: str) -> Tuple[Tensor, Tensor]:
    """two 1x1x1 convs; sigma = softplus   network_blocks.py:54-60"""
    mu = F.conv3d(h, sd[prefix + "._conv_mu.weight"], sd[prefix + "._conv_mu.bias"])
    sg = F.softplus(F.conv3d(h, sd[prefix + "._conv_sigma.0.weight"], sd[prefix + "._conv_sigma.0.bias"]))
    return mu, sg


def velocity_field(z: Tensor, sd, prefix: str, depth: int, training: bool) -> Tensor:
    """depth>=2: (depth-1) ConvUnits then a 1x1x1 conv; depth 1: bare unpadded 3x3x3 conv; depth 0: identity.
    network_blocks.py:74-82"""
    if depth == 0:
        return z
    if depth == 1:
        return F.conv3d(z, sd[prefix + "._op.0.weight"], sd[prefix + "._op.0.bias"])
    h = z
    for i in range(depth - 1):
        h = conv_unit(h, sd, f"{prefix}._op.{i}", training)
    return F.conv3d(h, sd[f"{prefix}._op.{depth - 1}.weight"], sd[f"{prefix}._op.{depth - 1}.bias"])


def identity_grid(size: Sequence[int], dtype=torch.float32) -> Tensor:
    """(1,3,D,H,W) voxel-index grid, 'ij' order   network_blocks.py:94-98"""
    axes = [torch.arange(s, dtype=dtype) for s in size]
    return torch.stack(torch.meshgrid(*axes, indexing="ij")).unsqueeze(0)


def warp(df: Tensor, img: Tensor) -> Tensor:
    """SpatialTransformer.forward   network_blocks.py:101-121.
    Grid size = df's spatial size; the image may have a different size (models.py:330).
    Normalisation uses (S-1) (align_corners=True convention) but sampling is align_corners=False."""
    size = df.shape[2:]
    loc = identity_grid(size, df.dtype).to(df.device) + df
    comps = [2 * (loc[:, i] / (size[i] - 1) - 0.5) for i in range(3)]
    grid = torch.stack(comps[::-1], dim=-1)          # channels last, xyz order (network_blocks.py:116-117)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=False)


def warp_explicit(df: Tensor, img: Tensor) -> Tensor:
    """The same operator written as an explicit 8-corner gather: this is the arithmetic the HIP kernel
    performs.  Sample coordinate along dim i:
        n_i = 2*((p_i + d_i)/(Sg_i - 1) - 0.5);  c_i = ((n_i + 1)*Si_i - 1)/2, clamped to [0, Si_i-1]."""
    B, _, *Sg = df.shape
    Si = img.shape[2:]
    loc = identity_grid(Sg, df.dtype) + df
    idx0, idx1, frac = [], [], []
    for i in range(3):
        n = 2 * (loc[:, i] / (Sg[i] - 1) - 0.5)
        c = (((n + 1) * Si[i] - 1) / 2).clamp(0, Si[i] - 1)
        f0 = c.floor()
        idx0.append(f0.long())
        idx1.append((f0.long() + 1).clamp(max=Si[i] - 1))
        frac.append(c - f0)
    C = img.shape[1]
    flat = img.reshape(B, C, -1)
    out = torch.zeros(B, C, *Sg, dtype=img.dtype)
    for cz in (0, 1):
        for cy in (0, 1):
            for cx in (0, 1):
                iz = idx1[0] if cz else idx0[0]
                iy = idx1[1] if cy else idx0[1]
                ix = idx1[2] if cx else idx0[2]
                wz = frac[0] if cz else 1 - frac[0]
                wy = frac[1] if cy else 1 - frac[1]
                wx = frac[2] if cx else 1 - frac[2]
                lin = ((iz * Si[1] + iy) * Si[2] + ix).reshape(B, 1, -1).expand(B, C, -1)
                out += (wz * wy * wx).unsqueeze(1) * flat.gather(2, lin).reshape(B, C, *Sg)
    return out


def vecint(v: Tensor, nsteps: int = 7) -> Tensor:
    """scaling and squaring   network_blocks.py:173-177"""
    v = v * (1.0 / (2 ** nsteps))
    for _ in range(nsteps):
        v = v + warp(v, v)
    return v


def resize_field(x: Tensor, vel_resize: float) -> Tensor:
    """ResizeTransform   network_blocks.py:129-150 (factor = 1/vel_resize)"""
    factor = 1.0 / vel_resize
    if factor < 1:
        x = F.interpolate(x, align_corners=False, scale_factor=factor, mode="trilinear")
        x = factor * x
    elif factor > 1:
        x = factor * x
        x = F.interpolate(x, align_corners=False, scale_factor=factor, mode="trilinear")
    return x


def pool2(x: Tensor) -> Tensor:
    """AvgPool3d(2, 2, ceil_mode=True)   pulpo.py:33,174"""
    return F.avg_pool3d(x, kernel_size=2, stride=2, padding=0, ceil_mode=True)


def resize_to(x: Tensor, size) -> Tensor:
    """F.interpolate(size=..., trilinear, align_corners=False)   pulpo.py:202, losses.py:313"""
    return F.interpolate(x, size=tuple(size), mode="trilinear", align_corners=False)


# =============================================================================== losses
def kl_diag(mu0: Tensor, sigma0: Tensor, mu1: Optional[Tensor] = None, sigma1: Optional[Tensor] = None, eps: float = 1e-10) -> Tensor:
    """KL[N(mu0,sigma0^2) || N(mu1,sigma1^2)], summed over features, mean over batch   losses.py:47-76.
    Default second argument is the N(0,1) prior (pulpo.py:330-341)."""
    if mu1 is None:
        mu1 = torch.zeros_like(mu0)
    if sigma1 is None:
        sigma1 = torch.ones_like(sigma0)
    s0 = sigma0.flatten(1) ** 2
    s1 = sigma1.flatten(1) ** 2
    term = (s0 + (mu1.flatten(1) - mu0.flatten(1)) ** 2) / (s1 + eps) + torch.log(s1 + eps) - torch.log(s0 + eps) - 1
    return (0.5 * term.sum(dim=1)).mean()


def box_sum(v: Tensor, w: int) -> Tensor:
    """zero-padded w^3 box sum as the reference computes it: a dense ones-kernel conv   losses.py:99-122"""
    k = torch.ones(1, 1, w, w, w, dtype=v.dtype, device=v.device)
    return F.conv3d(v, k, stride=1, padding=w // 2)


def ncc(pred: Tensor, true: Tensor, win: int = 9, gamma: float = 0.05) -> Tensor:
    """local normalised cross-correlation   losses.py:85-135"""
    I, J = true, pred
    Is, Js = box_sum(I, win), box_sum(J, win)
    I2s, J2s, IJs = box_sum(I * I, win), box_sum(J * J, win), box_sum(I * J, win)
    n = float(win ** 3)
    uI, uJ = Is / n, Js / n
    cross = IJs - uJ * Is - uI * Js + uI * uJ * n
    Iv = I2s - 2 * uI * Is + uI * uI * n
    Jv = J2s - 2 * uJ * Js + uJ * uJ * n
    cc = cross * cross / (Iv * Jv + 1e-8)
    return -torch.sum(cc.mean(dim=0)) * gamma


def ncc_grad_closed_form(pred: Tensor, true: Tensor, win: int, gamma: float) -> Tensor:
    """d ncc / d pred in closed form (SURVEY.md §8 'Backward semantics'): what the HIP backward evaluates.
    With box sums S_*, n = w^3, cross = S_IJ - S_I S_J/n, Iv = S_II - S_I^2/n, Jv = S_JJ - S_J^2/n, D = Iv Jv + 1e-8:
      dL/dJ = -(gamma/B) [ box(-2 cross S_I/(n D) + 2 cross^2 Iv S_J/(n D^2)) + 2 J box(-cross^2 Iv / D^2) + I box(2 cross / D) ]"""
    I, J = true.double(), pred.double()
    n = float(win ** 3)
    SI, SJ = box_sum(I, win), box_sum(J, win)
    SII, SJJ, SIJ = box_sum(I * I, win), box_sum(J * J, win), box_sum(I * J, win)
    cross = SIJ - SI * SJ / n
    Iv = SII - SI * SI / n
    Jv = SJJ - SJ * SJ / n
    D = Iv * Jv + 1e-8
    a = -2 * cross * SI / (n * D) + 2 * cross * cross * Iv * SJ / (n * D * D)
    b = -cross * cross * Iv / (D * D)
    c = 2 * cross / D
    g = box_sum(a, win) + 2 * J * box_sum(b, win) + I * box_sum(c, win)
    return (-(gamma / pred.shape[0]) * g).to(pred.dtype)


def l2_reg(df: Tensor, lamb: float) -> Tensor:
    """squared forward differences on the [1:,1:,1:] block, mean, times lamb*H*W*D   losses.py:217-222"""
    H, W, D = df.shape[-3:]
    c = df[:, :, 1:, 1:, 1:]
    d = (c - df[:, :, :-1, 1:, 1:]) ** 2 + (c - df[:, :, 1:, :-1, 1:]) ** 2 + (c - df[:, :, 1:, 1:, :-1]) ** 2
    return d.mean() * lamb * H * W * D


# =============================================================================== alternative losses / evaluation metrics
def l2_loss(inp: Tensor, tgt: Tensor) -> Tensor:
    """spatial sum of squared differences, mean over batch and channels   losses.py:79-83"""
    return ((inp - tgt) ** 2).flatten(2).sum(dim=2).mean()


def soft_dice(inp: Tensor, tgt: Tensor, dice_factor: float = 1) -> Tensor:
    """mean_(b,c) (1 - (2<t,i> + eps)/(|t|^2 + |i|^2 + eps)) * voxels / dice_factor, eps 1e-6   losses.py:137-145"""
    eps = 1e-6
    num = 2.0 * (tgt * inp).flatten(2).sum(dim=2) + eps
    den = (tgt ** 2).flatten(2).sum(dim=2) + (inp ** 2).flatten(2).sum(dim=2) + eps
    return (1 - num / den).mean() * float(math.prod(inp.shape[2:])) / dice_factor


def jacobian_det(df: Tensor, normalize: bool = True) -> Tensor:
    """determinant of (I + grad u) with central differences on a replicate-padded field   losses.py:172-199.
    Reference quirks kept: with normalize the channels are first scaled by 2/S_i; the field is then channel-FLIPPED and its
    (flipped) channel c scaled by ((D-1, H-1, W-1)[c] - 1)/2."""
    B, _, D, H, W = df.shape
    S = (D, H, W)
    u = torch.stack([df[:, i] * 2 / S[i] for i in range(3)], dim=1) if normalize else df
    scale = torch.tensor([(D - 1 - 1) / 2, (H - 1 - 1) / 2, (W - 1 - 1) / 2], dtype=df.dtype).view(1, 3, 1, 1, 1)
    uf = u.flip(1) * scale
    J = [[None] * 3 for _ in range(3)]
    for a in range(3):                                   # derivative axis: 0 = D, 1 = H, 2 = W
        idx = torch.arange(S[a])
        plus = uf.index_select(2 + a, (idx + 1).clamp(max=S[a] - 1))
        minus = uf.index_select(2 + a, (idx - 1).clamp(min=0))
        g = 0.5 * (plus - minus)
        for c in range(3):
            J[a][c] = g[:, c] + (1.0 if a == c else 0.0)
    return (J[0][0] * (J[1][1] * J[2][2] - J[2][1] * J[1][2]) - J[0][1] * (J[1][0] * J[2][2] - J[2][0] * J[1][2])
            + J[0][2] * (J[1][0] * J[2][1] - J[2][0] * J[1][1]))


def jdet_std(df: Tensor, lamb: float = 0.0, normalize: bool = True) -> Tensor:
    """lamb * (unbiased) standard deviation of the Jacobian determinant   losses.py:202-204"""
    return lamb * jacobian_det(df, normalize).std()


def kl_nondiagonal(mu: Tensor, sigma: Tensor, prior_lambda: float = 20.0) -> Tensor:
    """KL_nondiagonal.loss(prior_mu, prior_sigma, mu, sigma)   losses.py:8-44 (the prior arguments are unused there).
    D = (number of in-volume voxels of the 3x3x3 neighbourhood) - 1; precision term = mean squared forward differences."""
    S = mu.shape[2:]
    nd = len(S)
    D = box_sum(torch.ones(1, 1, *S, dtype=mu.dtype), 3) - 1
    s2 = sigma ** 2
    sigma_term = prior_lambda * D * s2 - torch.log(s2)
    sm = 0
    for a in range(nd):
        d = mu.narrow(2 + a, 1, S[a] - 1) - mu.narrow(2 + a, 0, S[a] - 1)
        sm = sm + (d * d).mean()
    precision = 0.5 * sm / nd
    return (sigma_term.mean() + (prior_lambda / 2) * precision) * nd * 0.5 * float(math.prod(S))


def rmse(inp: Tensor, tgt: Tensor) -> Tensor:
    """Evaluate.rmse   evaluate.py:315-319"""
    return torch.sqrt(((inp - tgt) ** 2).mean())


def dsc(inp: Tensor, tgt: Tensor) -> Tensor:
    """Evaluate.dsc   evaluate.py:321-327 (means over the spatial dims, then over batch and channel)"""
    dims = list(range(2, inp.dim()))
    return (((2.0 * tgt * inp).mean(dim=dims) + 1e-6) / ((tgt ** 2).mean(dim=dims) + (inp ** 2).mean(dim=dims) + 1e-6)).mean()


def jdet_leq0_percent(df: Tensor) -> Tensor:
    """the 'JDetLeq0' metric   evaluate.py:1441-1446"""
    jd = jacobian_det(df)
    return (jd <= 0).sum() / jd.numel() * 100


def warp_landmarks(lm: Tensor, df: Tensor) -> Tensor:
    """Evaluate.warp_landmarks   evaluate.py:410-423 (= src/components/utils.py:15-25)"""
    i = lm.long()
    return i - df[:, :, i[0, :, 0], i[0, :, 1], i[0, :, 2]].transpose(-2, -1)


def transform_segmentation(sd, cfg: "Cfg", final_dfs: Dict[int, Tensor], seg: Tensor) -> Dict[int, Tensor]:
    """PULPo.transform_segmentation   models.py:370-388: the full-resolution map on every level for df_resolution 'full_res', else the
    avg-pool chain (level 0 keeps the full map)"""
    L, o = cfg.latent_levels, cfg.total_levels - cfg.latent_levels
    if cfg.df_resolution == "full_res":
        level_seg = {l: seg for l in range(L)}
    else:
        level_seg = {0: seg}
        for _ in range(o):
            level_seg[0] = pool2(level_seg[0])
        for l in range(1, L):
            level_seg[l] = pool2(level_seg[l - 1])
        level_seg[0] = seg
    return {k: warp(final_dfs[k], level_seg[k]) for k in final_dfs}


def recon_ncc_dice(outs, y: Tensor, yhat_seg: Dict[int, Tensor], seg_y: Tensor, cfg: "Cfg", dice_factor: float = 1):
    """HierarchicalReconstructionLoss with recon_loss = ['ncc', 'dice']   losses.py:301-325"""
    window, _, rec_w, _ = weight_tables(cfg)
    yhat = outs[7]
    rec_l = {}
    for l, w in rec_w.items():
        t = w * ncc(yhat[l], resize_to(y, yhat[l].shape[2:]), window[l], cfg.gamma)
        t = t + w * soft_dice(yhat_seg[l], resize_to(seg_y, yhat_seg[l].shape[2:]), dice_factor)
        rec_l[l] = t / 2
    return sum(rec_l.values()), rec_l


# =============================================================================== network
OUT_NAMES = ("mus", "sigmas", "samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed")


def down_path(sd, cfg: Cfg, x: Tensor, y: Tensor, training: bool, prefix: str = "downpath") -> Dict[int, Tensor]:
    """DownPath.forward   pulpo.py:47-62"""
    h = torch.cat([x, y], dim=1)
    acts = {0: conv_sequence(h, sd, f"{prefix}.down_blocks.0", 3, training)}
    for k in range(1, cfg.total_levels):
        acts[k] = conv_sequence(stored(pool2(acts[k - 1])), sd, f"{prefix}.down_blocks.{k}", 3, training)
    return acts


def svf_decoder(sd, cfg: Cfg, prefix: str, z: Tensor, image: Tensor, combined_below: Optional[Tensor], out_ratio: float,
                training: bool):
    """SVFDecoder.forward   pulpo.py:301-319.  out_ratio = outsize[0]/insize[0] (pulpo.py:290)."""
    ind = velocity_field(z, sd, prefix + ".velocity_field", cfg.cp_depth, training)
    comb = ind if combined_below is None else resize_field(combined_below, 0.5) + ind
    integ = vecint(comb, 7)
    integ = resize_field(integ, 1.0 / out_ratio)
    warped = warp(integ, image)
    return ind, ind, comb, integ, warped


def autoencoder(sd, cfg: Cfg, x: Tensor, acts: Dict[int, Tensor], eps: Optional[Dict[int, Tensor]], training: bool,
                deterministic: bool = False, prefix: str = "autoencoder"):
    """Autoencoder.forward   pulpo.py:160-215.  eps[l] is the injected standard-normal noise of level l
    (the reference draws it in gauss_sampler, network_blocks.py:7-8)."""
    L, o = cfg.latent_levels, cfg.offset
    sizes = cfg.level_sizes()
    # level_x: pulpo.py:171-179
    full = cfg.df_resolution == "full_res"
    if full:                                          # pulpo.py:168-169
        lx = {l: x for l in range(L)}
    else:
        lx = {0: x}
        for _ in range(o):
            lx[0] = pool2(lx[0])
        for l in range(1, L):
            lx[l] = pool2(lx[l - 1])
        lx[0] = x
    out = {n: {} for n in OUT_NAMES}
    for l in reversed(range(L)):
        k = l + o
        enc = f"{prefix}.encoders.{l}"
        if l == L - 1:
            h = acts[k]
        else:
            fb = []
            for item in cfg.feedback:
                if item == "control_points":
                    item = "velocity_fields"
                if item not in OUT_NAMES:
                    raise ValueError(f"Feedback list contains {item}. Not a known option.")
                fb.append(resize_to(out[item][l + 1], acts[k].shape[2:]))
            up = conv_sequence(stored(torch.cat(fb, dim=1)), sd, f"{prefix}.up_blocks.{k}", 2, training)
            h = conv_sequence(torch.cat([up, acts[k]], dim=1), sd, enc + ".sample_merge_block", 2, training)
        mu, sg = mu_sigma(h, sd, enc + ".mu_sigma")
        if eps is None:
            z = mu + sg * torch.randn_like(sg, dtype=torch.float32)
        else:
            z = mu + sg * eps[l]
        out_ratio = (cfg.input_size[0] / sizes[k][0]) if (l == 0 or full) else 1.0          # pulpo.py:146,290
        res = svf_decoder(sd, cfg, f"{prefix}.decoders.{l}", mu if deterministic else z, lx[l],
                          None if l == L - 1 else out["combined_dfs"][l + 1], out_ratio, training)
        out["mus"][l], out["sigmas"][l], out["samples"][l] = mu, sg, z
        for nme, t in zip(OUT_NAMES[3:], res):
            out[nme][l] = t
    return tuple(out[n] for n in OUT_NAMES)


def forward(sd, cfg: Cfg, x: Tensor, y: Tensor, eps=None, training: bool = True, deterministic: bool = False):
    """downpath + autoencoder (models.py:138-139)"""
    return autoencoder(sd, cfg, x, down_path(sd, cfg, x, y, training), eps, training, deterministic)


def losses(outs, y: Tensor, cfg: Cfg):
    """the loss block of training_step   models.py:151-164 with losses.py:246-355"""
    mus, sigmas, _, _, _, _, final_dfs, yhat = outs
    window, kl_w, rec_w, reg_w = weight_tables(cfg)
    kl_l = {l: cfg.beta * (w * kl_diag(mus[l], sigmas[l])) for l, w in kl_w.items()}
    kl = sum(w * kl_diag(mus[l], sigmas[l]) for l, w in kl_w.items()) * cfg.beta
    rec_l = {l: (w * ncc(yhat[l], resize_to(y, yhat[l].shape[2:]), window[l], cfg.gamma)) / 1 for l, w in rec_w.items()}
    rec = sum(rec_l.values())
    reg_l = {l: w * l2_reg(final_dfs[l], cfg.lamb) for l, w in reg_w.items()}
    reg = sum(reg_l.values())
    return kl + rec + reg, kl, rec, reg, kl_l, rec_l, reg_l


def combine_dfs(individual: Dict[int, Tensor], cfg: Cfg):
    """PULPo.combine_dfs   models.py:349-368"""
    L = cfg.latent_levels
    comb, fin = {}, {}
    for l in reversed(range(L)):
        if l + 1 in comb:
            ratio = individual[l].shape[2] / individual[l + 1].shape[2]
            comb[l] = individual[l] + resize_field(comb[l + 1], 1.0 / ratio)
        else:
            comb[l] = individual[l]
    for l in reversed(range(L)):
        f = vecint(comb[l], 7)
        tgt = cfg.input_size if (l == 0 or cfg.df_resolution == "full_res") else comb[l].shape[2:]
        fin[l] = resize_field(f, 1.0 / (tgt[0] / f.shape[2]))
    return comb, fin


# =============================================================================== state dict helpers
def init_state_dict(cfg: Cfg, seed: int = 0) -> Dict[str, Tensor]:
    """A reference-shaped state dict with PyTorch-default initialisation (kaiming-uniform(a=sqrt 5) conv weights,
    U(-1/sqrt(fan_in), +) biases, BN gamma 1 / beta 0 / mean 0 / var 1).  Key inventory follows
    tests/golden/state_keys.txt; persistent SpatialTransformer.grid buffers are included (network_blocks.py:99)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def conv(name, cin, cout, k):
        fan_in = cin * k ** 3
        bound = 1.0 / math.sqrt(fan_in)
        sd[name + ".weight"] = (torch.rand(cout, cin, k, k, k, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def unit(name, cin, cout):
        conv(name + "._op.0", cin, cout, 3)
        sd[name + "._op.1.weight"] = torch.ones(cout)
        sd[name + "._op.1.bias"] = torch.zeros(cout)
        sd[name + "._op.1.running_mean"] = torch.zeros(cout)
        sd[name + "._op.1.running_var"] = torch.ones(cout)
        sd[name + "._op.1.num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    def seq(name, cin, cout, depth):
        for i in range(depth):
            unit(f"{name}._op.{i}", cin if i == 0 else cout, cout)

    ch, sizes, o, L, T = cfg.channels(), cfg.level_sizes(), cfg.offset, cfg.latent_levels, cfg.total_levels
    for k in range(T):
        seq(f"downpath.down_blocks.{k}", 2 if k == 0 else ch[k - 1], ch[k], 3)
    fbc = sum({"samples": cfg.zdim, "transformed": 1}.get(i, 3) for i in cfg.feedback)
    for k in range(o, T - 1):
        seq(f"autoencoder.up_blocks.{k}", fbc, cfg.n0 * cfg.zdim, 2)
    for l in range(L):
        k = l + o
        seq(f"autoencoder.encoders.{l}.sample_merge_block", ch[k] + cfg.n0 * cfg.zdim, ch[k], 2)
        conv(f"autoencoder.encoders.{l}.mu_sigma._conv_mu", ch[k], cfg.zdim, 1)
        conv(f"autoencoder.encoders.{l}.mu_sigma._conv_sigma.0", ch[k], cfg.zdim, 1)
    for l in range(L):
        k = l + o
        d = f"autoencoder.decoders.{l}"
        if cfg.cp_depth == 1:                          # a bare (unpadded) 3x3x3 convolution, network_blocks.py:74-75
            conv(d + ".velocity_field._op.0", cfg.zdim, 3, 3)
        elif cfg.cp_depth >= 2:                        # (depth 0: nn.Identity, no parameters - network_blocks.py:76-77)
            unit(d + ".velocity_field._op.0", cfg.zdim, cfg.n0)
            for i in range(1, cfg.cp_depth - 1):
                unit(d + f".velocity_field._op.{i}", cfg.n0, cfg.n0)
            conv(d + f".velocity_field._op.{cfg.cp_depth - 1}", cfg.n0, 3, 1)
        sd[d + ".integrate.transformer.grid"] = identity_grid(sizes[k])
        sd[d + ".spatial_transform.grid"] = identity_grid(cfg.input_size if (l == 0 or cfg.df_resolution == "full_res") else sizes[k])
    return sd


def clone_sd(sd, requires_grad: bool = False) -> Dict[str, Tensor]:
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and not ("running_" in k or k.endswith(".grid")):
            t.requires_grad_(True)
        out[k] = t
    return out


def train_step(sd, cfg: Cfg, x: Tensor, y: Tensor, eps):
    """forward + losses + backward; returns (loss tuple, grads by state-dict key, outputs).  `sd` must hold leaf
    tensors with requires_grad (clone_sd(..., True)); BN running stats in `sd` are updated."""
    outs = forward(sd, cfg, x, y, eps, training=True)
    ls = losses(outs, y, cfg)
    params = {k: v for k, v in sd.items() if v.requires_grad}
    gl = torch.autograd.grad(ls[0], list(params.values()), allow_unused=True)
    return ls, {k: g for k, g in zip(params, gl)}, outs


# =============================================================================== Monte-Carlo uncertainty maps
def mc_std_map(stack: Tensor, scale: Optional[Tensor] = None) -> Tensor:
    """(N, C, D, H, W) samples -> (D, H, W): torch.mean(torch.std(stack * scale, axis=0), axis=0)   evaluate.py:243-251
    (torch.std is the unbiased estimator; `scale` is the warped mask of evaluate.py:249, broadcast over N and C)"""
    if scale is not None:
        stack = stack * scale
    return torch.mean(torch.std(stack, axis=0), axis=0)
