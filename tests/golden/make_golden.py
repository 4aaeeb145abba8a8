#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference); the reference never
travels to the GPU box, these .npz fixtures do.  Everything is seeded, fp32, CPU.

The reference modules that import cleanly here are used unmodified:
  src.network_blocks, src.components.pulpo, src.losses   
src.models needs pytorch_lightning/torchvision, which the image lacks.  The step
fixtures assemble the training step from the component modules exactly as
models.py:134-164 does (DownPath -> Autoencoder -> Prior -> three Hierarchical*
losses -> beta*KL + recon + reg), with the loss-weight dictionaries of
models.py:104-123 evaluated in this script.  The `models` mode (round 3) imports
src/models.py ITSELF behind two plumbing-only stand-in modules (a LightningModule
that is an nn.Module with save_hyperparameters / log_dict, no-op make_grid /
flow_to_image - SURVEY Appendix B) and stores what the class's own methods return:
models_api_{level,full}_res_*.npz.

usage:  python tests/golden/make_golden.py   (writes next to itself)
"""
import os
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
# make sure `src` resolves to the reference, never to this repo's drop-in shim
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.abspath(os.path.join(HERE, "..", ".."))]
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn.functional as F

import src.network_blocks as nb          # noqa: E402
import src.components.pulpo as cp        # noqa: E402
import src.losses as ls                  # noqa: E402

assert os.path.abspath(nb.__file__).startswith(REF), nb.__file__
torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %7.1f KiB  (%d arrays)" % (name + ".npz", os.path.getsize(path) / 1024, len(arrs)))


# --------------------------------------------------------------------------- 1. warp3d
def gen_warp():
    g = torch.Generator().manual_seed(101)
    out = {}
    S = [6, 8, 10]
    st = nb.SpatialTransformer(S)
    # (a) random field, 1-channel image, upstream = ones and random
    df = (torch.randn(2, 3, *S, generator=g) * 1.5).requires_grad_(True)
    img = torch.rand(2, 1, *S, generator=g).requires_grad_(True)
    up = torch.randn(2, 1, *S, generator=g)
    o = st(df, img)
    gdf1, gimg1 = torch.autograd.grad(o.sum(), [df, img], retain_graph=True)
    gdf2, gimg2 = torch.autograd.grad((o * up).sum(), [df, img])
    out.update(a_df=npy(df), a_img=npy(img), a_up=npy(up), a_out=npy(o), a_gdf_ones=npy(gdf1), a_gimg_ones=npy(gimg1),
               a_gdf_rand=npy(gdf2), a_gimg_rand=npy(gimg2))
    # (b) zero field is NOT the identity (Appendix A.2)
    img = torch.rand(1, 1, *S, generator=g)
    out.update(b_img=npy(img), b_out=npy(st(torch.zeros(1, 3, *S), img)))
    # (c) 3-channel image (the VecInt use: image == field)
    df = (torch.randn(1, 3, *S, generator=g) * 0.8).requires_grad_(True)
    up = torch.randn(1, 3, *S, generator=g)
    o = st(df, df)
    gdf, = torch.autograd.grad((o * up).sum(), [df])
    out.update(c_df=npy(df), c_up=npy(up), c_out=npy(o), c_gdf=npy(gdf))
    # (d) image larger than the grid (models.py:330: full-res x warped by a level-res field)
    Sg = [4, 5, 6]
    stg = nb.SpatialTransformer(Sg)
    df = torch.randn(1, 3, *Sg, generator=g) * 1.2
    img = torch.rand(1, 1, 8, 10, 12, generator=g)
    out.update(d_df=npy(df), d_img=npy(img), d_out=npy(stg(df, img)))
    save("warp3d", **out)


# --------------------------------------------------------------------------- 2. vecint
def gen_vecint():
    g = torch.Generator().manual_seed(102)
    vi = nb.VecInt([8, 8, 8], nsteps=7)
    v = (torch.randn(1, 3, 8, 8, 8, generator=g) * 2.0).requires_grad_(True)
    up = torch.randn(1, 3, 8, 8, 8, generator=g)
    o = vi(v)
    gv, = torch.autograd.grad((o * up).sum(), [v])
    vi2 = nb.VecInt([5, 6, 7], nsteps=7)
    v2 = (torch.randn(2, 3, 5, 6, 7, generator=g) * 1.0).requires_grad_(True)
    up2 = torch.randn(2, 3, 5, 6, 7, generator=g)
    o2 = vi2(v2)
    gv2, = torch.autograd.grad((o2 * up2).sum(), [v2])
    save("vecint", v=npy(v), up=npy(up), out=npy(o), gv=npy(gv), v2=npy(v2), up2=npy(up2), out2=npy(o2), gv2=npy(gv2))


# --------------------------------------------------------------------------- 3-5. resize / pool / interpolate
def gen_resample():
    g = torch.Generator().manual_seed(103)
    out = {}
    # ResizeTransform(1/2): x2 up with x2 magnitude (network_blocks.py:144-147)
    rt = nb.ResizeTransform(0.5, 3)
    x = torch.randn(1, 3, 4, 5, 6, generator=g).requires_grad_(True)
    up = torch.randn(1, 3, 8, 10, 12, generator=g)
    o = rt(x)
    gx, = torch.autograd.grad((o * up).sum(), [x])
    out.update(rt_x=npy(x), rt_up=npy(up), rt_out=npy(o), rt_gx=npy(gx))
    # avg_pool3d k2 s2 ceil_mode on odd and even sizes (pulpo.py:33,174)
    for tag, shp in (("odd", (2, 3, 5, 6, 7)), ("even", (1, 4, 6, 8, 4))):
        x = torch.randn(*shp, generator=g).requires_grad_(True)
        o = F.avg_pool3d(x, kernel_size=2, stride=2, padding=0, ceil_mode=True)
        up = torch.randn(*o.shape, generator=g)
        gx, = torch.autograd.grad((o * up).sum(), [x])
        out.update({f"pool_{tag}_x": npy(x), f"pool_{tag}_up": npy(up), f"pool_{tag}_out": npy(o), f"pool_{tag}_gx": npy(gx)})
    # F.interpolate(size=...) x2 up (feedback, pulpo.py:202) and 1/2, 1/4, 1/8 down (y_target, losses.py:313)
    x = torch.randn(2, 3, 4, 6, 5, generator=g).requires_grad_(True)
    o = F.interpolate(x, size=(8, 12, 10), mode="trilinear", align_corners=False)
    up = torch.randn(*o.shape, generator=g)
    gx, = torch.autograd.grad((o * up).sum(), [x])
    out.update(up2_x=npy(x), up2_up=npy(up), up2_out=npy(o), up2_gx=npy(gx))
    y = torch.rand(1, 1, 16, 16, 24, generator=g)
    out["dn_y"] = npy(y)
    for f in (1, 2, 4, 8):
        out[f"dn_out{f}"] = npy(F.interpolate(y, size=(16 // f, 16 // f, 24 // f), mode="trilinear", align_corners=False))
    # non-integer ratio (generic size= path)
    x = torch.randn(1, 2, 5, 7, 6, generator=g)
    out.update(gen_x=npy(x), gen_out=npy(F.interpolate(x, size=(8, 9, 11), mode="trilinear", align_corners=False)))
    save("resample", **out)


# --------------------------------------------------------------------------- 6. ConvUnit
def gen_convunit():
    torch.manual_seed(104)
    g = torch.Generator().manual_seed(104)
    out = {}
    for tag, (cin, cout, S, B) in {"a": (3, 4, (6, 6, 6), 2), "b": (5, 7, (4, 6, 5), 1)}.items():
        cu = nb.ConvUnit(list(S), cin, cout)
        with torch.no_grad():  # non-trivial affine + running stats
            cu._op[1].weight.copy_(torch.rand(cout, generator=g) + 0.5)
            cu._op[1].bias.copy_(torch.randn(cout, generator=g) * 0.3)
        sd0 = {k: npy(v) for k, v in cu.state_dict().items()}
        x = torch.randn(B, cin, *S, generator=g).requires_grad_(True)
        up = torch.randn(B, cout, *S, generator=g)
        cu.train()
        o = cu(x)
        params = [cu._op[0].weight, cu._op[0].bias, cu._op[1].weight, cu._op[1].bias]
        grads = torch.autograd.grad((o * up).sum(), [x] + params)
        sd1 = {k: npy(v) for k, v in cu.state_dict().items()}
        cu.eval()
        oe = cu(x)
        out.update({f"{tag}_x": npy(x), f"{tag}_up": npy(up), f"{tag}_out_train": npy(o), f"{tag}_out_eval": npy(oe),
                    f"{tag}_gx": npy(grads[0]), f"{tag}_gw": npy(grads[1]), f"{tag}_gb": npy(grads[2]),
                    f"{tag}_ggamma": npy(grads[3]), f"{tag}_gbeta": npy(grads[4])})
        out.update({f"{tag}_sd0.{k}": v for k, v in sd0.items()})
        out.update({f"{tag}_sd1.{k}": v for k, v in sd1.items()})
    save("convunit", **out)


# --------------------------------------------------------------------------- 7. MuSigma + sampler
def gen_musigma():
    torch.manual_seed(105)
    g = torch.Generator().manual_seed(105)
    ms = nb.MuSigmaBlock([4, 5, 6], 6, 3)
    x = (torch.randn(2, 6, 4, 5, 6, generator=g) * 3).requires_grad_(True)
    eps = torch.randn(2, 3, 4, 5, 6, generator=g)
    mu, sigma = ms(x)
    z = mu + sigma * eps                      # gauss_sampler with injected noise (network_blocks.py:7-8)
    up = torch.randn(2, 3, 4, 5, 6, generator=g)
    params = list(ms.parameters())
    grads = torch.autograd.grad((z * up).sum() + (mu * mu).sum() + sigma.sum(), [x] + params)
    out = {"x": npy(x), "eps": npy(eps), "up": npy(up), "mu": npy(mu), "sigma": npy(sigma), "z": npy(z), "gx": npy(grads[0])}
    out.update({"sd." + k: npy(v) for k, v in ms.state_dict().items()})
    for (n, _), gr in zip(ms.named_parameters(), grads[1:]):
        out["g." + n] = npy(gr)
    # VelocityField depth 3 (3->n0->n0 ConvUnits + 1x1x1), eval mode only here (train mode is covered by the full step)
    vf = nb.VelocityField([4, 5, 6], 3, 8, 3)
    vf.eval()
    zz = torch.randn(1, 3, 4, 5, 6, generator=g)
    out.update({"vf_z": npy(zz), "vf_out": npy(vf(zz))})
    out.update({"vf_sd." + k: npy(v) for k, v in vf.state_dict().items()})
    save("musigma", **out)


# --------------------------------------------------------------------------- 8-9. losses
def smooth_volume(g, S, B=1):
    """smooth blob with zero background, like a skull-stripped scan"""
    lo = torch.rand(B, 1, *(max(2, s // 4) for s in S), generator=g)
    v = F.interpolate(lo, size=S, mode="trilinear", align_corners=False)
    zz, yy, xx = torch.meshgrid(*[torch.linspace(-1, 1, s) for s in S], indexing="ij")
    mask = ((zz ** 2 + yy ** 2 + xx ** 2) < 0.8).float()
    return v * mask


def gen_losses():
    g = torch.Generator().manual_seed(106)
    out = {}
    for w, S, B in ((3, (12, 12, 12), 2), (5, (12, 14, 13), 1), (7, (14, 12, 16), 1), (9, (16, 16, 16), 2), (11, (16, 14, 12), 1)):
        for kind in ("rand", "smooth"):
            if kind == "rand":
                yt = torch.rand(B, 1, *S, generator=g)
                yp = torch.rand(B, 1, *S, generator=g).requires_grad_(True)
            else:
                yt = smooth_volume(g, S, B)
                yp = (smooth_volume(g, S, B) * 0.7 + 0.3 * yt).requires_grad_(True)
            loss = ls.NCC_loss(yp, yt, win_size=w, gamma=0.05)
            gp, = torch.autograd.grad(loss, [yp])
            out.update({f"ncc{w}_{kind}_true": npy(yt), f"ncc{w}_{kind}_pred": npy(yp), f"ncc{w}_{kind}_loss": npy(loss),
                        f"ncc{w}_{kind}_gpred": npy(gp)})
    # KL[posterior || N(0,1)] (losses.py:47-76 called as in losses.py:271-273)
    mu = torch.randn(2, 3, 4, 5, 6, generator=g).requires_grad_(True)
    sg = (F.softplus(torch.randn(2, 3, 4, 5, 6, generator=g))).requires_grad_(True)
    kl = ls.KL_two_gauss_with_diag_cov(mu, sg, torch.zeros_like(mu), torch.ones_like(sg))
    gmu, gsg = torch.autograd.grad(kl, [mu, sg])
    out.update(kl_mu=npy(mu), kl_sigma=npy(sg), kl_loss=npy(kl), kl_gmu=npy(gmu), kl_gsigma=npy(gsg))
    # general KL between two diagonal Gaussians
    mu1 = torch.randn(2, 3, 4, 5, 6, generator=g)
    sg1 = F.softplus(torch.randn(2, 3, 4, 5, 6, generator=g)) + 0.1
    out.update(kl2_mu1=npy(mu1), kl2_sigma1=npy(sg1), kl2_loss=npy(ls.KL_two_gauss_with_diag_cov(mu, sg, mu1, sg1)))
    # L2_reg (losses.py:208-222)
    df = torch.randn(2, 3, 5, 6, 7, generator=g).requires_grad_(True)
    r = ls.L2_reg(df, 0.025)
    gdf, = torch.autograd.grad(r, [df])
    out.update(reg_df=npy(df), reg_loss=npy(r), reg_gdf=npy(gdf))
    save("losses", **out)


# --------------------------------------------------------------------------- 10. full training step
FEEDBACK = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]


def weight_dicts(T, L, ndims=3):
    """models.py:104-123 (df_resolution == 'level_res', similarity_pyramid False)"""
    o = T - L
    window = {l: 1 + 2 * (L - l) for l in range(L)}
    if L == 1:
        window = {0: 9}
    scale = {l: (2.0 ** ndims) ** l for l in range(L)}
    kl_w = dict(scale)
    rec_w = dict(scale)
    reg_w = dict(scale)
    rec_w[0] = scale[0] / (2 ** (ndims * o))
    reg_w[0] = scale[0] / (2 ** (ndims * o))
    rec_w[0] *= 4
    return window, kl_w, rec_w, reg_w


class Step:
    """The reference's training_step assembled from its own component modules (no Lightning)."""

    def __init__(self, T, L, size, n0, seed, df_resolution="level_res", cp_depth=3):
        torch.manual_seed(seed)
        self.T, self.L = T, L
        nd = len(size)                           # 3 (volumes) or 2 (slices, train.py --ndims 2): zdim = ndims (models.py:88)
        self.down = cp.DownPath(T, L, list(size), 2, n0)
        self.ae = cp.Autoencoder(nb.gauss_sampler, "SVF", T, L, nd, list(size), list(FEEDBACK), df_resolution, n0, cp_depth)
        self.prior = cp.PULPoPrior()
        window, kl_w, rec_w, reg_w = weight_dicts(T, L, ndims=nd)
        if df_resolution == "full_res":          # models.py:112-115,123
            rec_w = {l: 1.0 for l in range(L)}
            reg_w = {l: 1.0 for l in range(L)}
            rec_w[0] *= 4
        self.kl = ls.HierarchicalKLLoss(ls.KL_two_gauss_with_diag_cov, kl_w, False, None)
        self.rec = ls.HierarchicalReconstructionLoss(["ncc"], rec_w, False, nd, window)
        self.reg = ls.HierarchicalRegularization(ls.L2_reg, reg_w, False)
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():  # make BN affine + running stats non-trivial so eval mode is a real test
            for m in list(self.down.modules()) + list(self.ae.modules()):
                if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.BatchNorm2d)):
                    m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                    m.running_mean.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                    m.running_var.copy_(torch.rand(m.bias.shape, generator=g) * 0.5 + 0.75)

    def state_dict(self):
        sd = {"downpath." + k: v for k, v in self.down.state_dict().items()}
        sd.update({"autoencoder." + k: v for k, v in self.ae.state_dict().items()})
        return sd

    def named_parameters(self):
        for k, v in self.down.named_parameters():
            yield "downpath." + k, v
        for k, v in self.ae.named_parameters():
            yield "autoencoder." + k, v

    def set_eps(self, eps):
        for l in range(self.L):
            self.ae.encoders[l].sampler = (lambda mu, sigma, e=eps[l]: mu + sigma * e)

    def forward(self, x, y, deterministic=False):
        acts = self.down(x, y)
        return self.ae(x, acts, deterministic=deterministic)

    def losses(self, outs, y, beta=0.1, gamma=0.05, lamb=0.025):
        mus, sigmas, samples, vfs, ind, comb, fin, yhat = outs
        pm, ps = self.prior(mus, sigmas)
        kl, kl_l = self.kl(pm, ps, mus, sigmas)
        kl = kl * beta
        kl_l = {n: beta * v for n, v in kl_l.items()}
        rec, rec_l = self.rec(yhat, y, {k: None for k in fin}, None, gamma=gamma, dice_factor=1)
        reg, reg_l = self.reg(fin, lamb=lamb)
        return kl + rec + reg, kl, rec, reg, kl_l, rec_l, reg_l


OUT_NAMES = ["mus", "sigmas", "samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]


def gen_step(name, T, L, size, n0, B, seed, with_grads=True, smooth=False, df_resolution="level_res", cp_depth=3):
    st = Step(T, L, size, n0, seed, df_resolution, cp_depth)
    g = torch.Generator().manual_seed(seed + 2)
    if smooth:
        y = smooth_volume(g, tuple(size), B)
        x = (0.6 * smooth_volume(g, tuple(size), B) + 0.4 * y)
    else:
        x = torch.rand(B, 1, *size, generator=g)
        y = torch.rand(B, 1, *size, generator=g)
    o = T - L
    lvl = lambda l: [s // (2 ** (l + o)) for s in size]
    eps = {l: torch.randn(B, len(size), *lvl(l), generator=g) for l in range(L)}
    st.set_eps(eps)
    out = {"cfg": np.array([T, L, n0, B] + list(size), dtype=np.int64), "x": npy(x), "y": npy(y)}
    out.update({f"eps.{l}": npy(e) for l, e in eps.items()})
    # the persistent SpatialTransformer.grid buffers are pure aranges (network_blocks.py:94-99): inventoried in
    # state_keys.txt, not stored
    out.update({"sd0." + k: npy(v) for k, v in st.state_dict().items() if not k.endswith(".grid")})
    # ---- training mode forward + backward
    st.down.train(); st.ae.train()
    outs = st.forward(x, y)
    total, kl, rec, reg, kl_l, rec_l, reg_l = st.losses(outs, y)
    for nme, d in zip(OUT_NAMES, outs):
        out.update({f"train.{nme}.{l}": npy(v) for l, v in d.items()})
    out.update({"train.total": npy(total), "train.kl": npy(kl), "train.rec": npy(rec), "train.reg": npy(reg)})
    out.update({f"train.kl_l.{l}": npy(v) for l, v in kl_l.items()})
    out.update({f"train.rec_l.{l}": npy(v) for l, v in rec_l.items()})
    out.update({f"train.reg_l.{l}": npy(v) for l, v in reg_l.items()})
    if with_grads:
        total.backward()
        for k, p in st.named_parameters():
            if p.grad is not None:
                out["grad." + k] = npy(p.grad)
            else:
                out["nograd." + k] = np.zeros(1, dtype=np.float32)
    # buffers after the training forward (BN running stats, num_batches_tracked)
    for k, v in st.state_dict().items():
        if "running_" in k or "num_batches" in k:
            out["sd1." + k] = npy(v)
    # ---- eval mode: stochastic (same eps), deterministic, combine_dfs inputs
    st.down.eval(); st.ae.eval()
    with torch.no_grad():
        outs_e = st.forward(x, y)
        tot_e = st.losses(outs_e, y)
        outs_d = st.forward(x, y, deterministic=True)
    for nme, d in zip(OUT_NAMES, outs_e):
        out.update({f"eval.{nme}.{l}": npy(v) for l, v in d.items()})
    out.update({"eval.total": npy(tot_e[0]), "eval.kl": npy(tot_e[1]), "eval.rec": npy(tot_e[2]), "eval.reg": npy(tot_e[3])})
    for nme in ("individual_dfs", "final_dfs", "transformed"):
        d = outs_d[OUT_NAMES.index(nme)]
        out.update({f"det.{nme}.{l}": npy(v) for l, v in d.items()})
    save(name, **out)
    return float(total)


# --------------------------------------------------------------------------- 11b. --recon_loss ncc dice with segmentations
def gen_step_dice(name, df_resolution, T=3, L=2, size=(16, 16, 16), n0=2, B=1, seed=160):
    """training step with recon_loss = ["ncc", "dice"] (train.py:29, --segs): the segmentation maps are warped per level exactly as
    models.py:370-388 (transform_segmentation) does - full-resolution map on EVERY level when df_resolution == "full_res", the avg-pool
    chain otherwise - with the reference's own SpatialTransformer instances, then fed to the reference's HierarchicalReconstructionLoss."""
    st = Step(T, L, list(size), n0, seed, df_resolution)
    st.rec = ls.HierarchicalReconstructionLoss(["ncc", "dice"], dict(st.rec.weight_dict), False, 3, st.rec.window_size)
    g = torch.Generator().manual_seed(seed + 2)
    y = smooth_volume(g, tuple(size), B)
    x = 0.6 * smooth_volume(g, tuple(size), B) + 0.4 * y
    # soft 3-label "segmentations" (3 channels, like one-hot label maps after interpolation)
    seg_x = torch.softmax(4 * torch.cat([smooth_volume(g, tuple(size), B) for _ in range(3)], dim=1), dim=1)
    seg_y = torch.softmax(4 * torch.cat([smooth_volume(g, tuple(size), B) for _ in range(3)], dim=1), dim=1)
    o = T - L
    eps = {l: torch.randn(B, 3, *[s // 2 ** (l + o) for s in size], generator=g) for l in range(L)}
    st.set_eps(eps)
    out = {"cfg": np.array([T, L, n0, B] + list(size), dtype=np.int64), "x": npy(x), "y": npy(y), "seg_x": npy(seg_x), "seg_y": npy(seg_y)}
    out.update({f"eps.{l}": npy(e) for l, e in eps.items()})
    out.update({"sd0." + k: npy(v) for k, v in st.state_dict().items() if not k.endswith(".grid")})
    st.down.train(); st.ae.train()
    outs = st.forward(x, y)
    mus, sigmas, samples, vfs, ind, comb, fin, yhat = outs
    # models.py:370-388
    if df_resolution == "full_res":
        level_seg = {l: seg_x for l in range(L)}
    else:
        level_seg = {0: seg_x}
        for _ in range(o):
            level_seg[0] = F.avg_pool3d(level_seg[0], kernel_size=2, stride=2, padding=0, ceil_mode=True)
        for l in range(1, L):
            level_seg[l] = F.avg_pool3d(level_seg[l - 1], kernel_size=2, stride=2, padding=0, ceil_mode=True)
        level_seg[0] = seg_x
    yhat_seg = {k: st.ae.decoders[k].spatial_transform(fin[k], level_seg[k]) for k in fin}
    pm, ps = st.prior(mus, sigmas)
    kl, _ = st.kl(pm, ps, mus, sigmas)
    kl = kl * 0.1
    rec, rec_l = st.rec(yhat, y, yhat_seg, seg_y, gamma=0.05, dice_factor=1)
    reg, _ = st.reg(fin, lamb=0.025)
    total = kl + rec + reg
    out.update({f"train.y_hat_seg.{l}": npy(v) for l, v in yhat_seg.items()})
    out.update({f"train.rec_l.{l}": npy(v) for l, v in rec_l.items()})
    out.update({"train.total": npy(total), "train.kl": npy(kl), "train.rec": npy(rec), "train.reg": npy(reg)})
    total.backward()
    for k, p_ in st.named_parameters():
        if p_.grad is not None:
            out["grad." + k] = npy(p_.grad)
    save(name, **out)
    return float(total)


# --------------------------------------------------------------------------- 12. alternative losses / evaluation metrics
def gen_metrics():
    g = torch.Generator().manual_seed(130)
    out = {}
    # L2_loss (losses.py:79-83): mean over (B,C) of the spatial sum of squared differences
    a = torch.rand(2, 1, 5, 6, 7, generator=g).requires_grad_(True)
    b = torch.rand(2, 1, 5, 6, 7, generator=g)
    l = ls.L2_loss(a, b)
    ga, = torch.autograd.grad(l, [a])
    out.update(l2_in=npy(a), l2_tgt=npy(b), l2_loss=npy(l), l2_gin=npy(ga))
    # Soft_dice_loss (losses.py:137-145) on soft "segmentations" with 3 channels
    a = torch.rand(2, 3, 6, 5, 4, generator=g).requires_grad_(True)
    b = (torch.rand(2, 3, 6, 5, 4, generator=g) > 0.6).float()
    for df_ in (1, 4):
        l = ls.Soft_dice_loss(a, b, dice_factor=df_)
        ga, = torch.autograd.grad(l, [a])
        out.update({f"dice{df_}_loss": npy(l), f"dice{df_}_gin": npy(ga)})
    out.update(dice_in=npy(a), dice_tgt=npy(b))
    # jacobian_det / JDetStd (losses.py:147-204), 3-D, normalize True and False
    d = (torch.randn(2, 3, 6, 7, 8, generator=g) * 0.8).requires_grad_(True)
    for norm in (True, False):
        jd = ls.jacobian_det(d, normalize=norm)
        out[f"jdet_norm{int(norm)}"] = npy(jd)
        s_ = ls.JDetStd(d, lamb=0.3, normalize=norm)
        gd, = torch.autograd.grad(s_, [d])
        out.update({f"jstd_norm{int(norm)}": npy(s_), f"jstd_gd_norm{int(norm)}": npy(gd)})
    out["jdet_df"] = npy(d)
    # KL_nondiagonal (losses.py:8-44) as HierarchicalKLLoss calls it: loss(prior_mu, prior_sigma, posterior_mu, posterior_sigma)
    shape = (5, 6, 7)
    kln = ls.KL_nondiagonal(inshape=torch.tensor(shape), prior_lambda=20)
    kln.D = kln.D.cpu()
    mu = torch.randn(2, 3, *shape, generator=g).requires_grad_(True)
    sg = (F.softplus(torch.randn(2, 3, *shape, generator=g)) + 0.05).requires_grad_(True)
    l = kln.loss(torch.zeros_like(mu), torch.ones_like(sg), mu, sg)
    gm, gs = torch.autograd.grad(l, [mu, sg])
    out.update(kln_mu=npy(mu), kln_sigma=npy(sg), kln_loss=npy(l), kln_gmu=npy(gm), kln_gsigma=npy(gs), kln_D=npy(kln.D))
    save("metrics", **out)


def gen_evalmetrics():
    """evaluation scalars of the harness.  warp_landmarks comes from the REAL reference module
    (src/components/utils.py:15-25, identical to Evaluate.warp_landmarks evaluate.py:410-423); rmse / dsc / JDetLeq0 are methods or inline
    code of evaluate.py (not importable here: h5py, seaborn, torchvision are absent), so their arithmetic is restated verbatim:
        rmse      evaluate.py:315-319   sqrt(torch.nn.MSELoss()(input, target))
        dsc       evaluate.py:321-327   ((2*t*i).mean(spatial) + 1e-6) / ((t**2).mean(spatial) + (i**2).mean(spatial) + 1e-6), .mean()
        JDetLeq0  evaluate.py:1441-1446 (sum(jdet <= 0) / prod(jdet.squeeze().size())) * 100 with the reference's jacobian_det"""
    import src.components.utils as cu
    g = torch.Generator().manual_seed(170)
    out = {}
    a = torch.rand(2, 1, 9, 10, 11, generator=g)
    b = torch.rand(2, 1, 9, 10, 11, generator=g)
    out.update(rmse_a=npy(a), rmse_b=npy(b), rmse=npy(torch.sqrt(torch.nn.MSELoss()(a, b))))
    i = torch.softmax(3 * torch.randn(2, 4, 7, 8, 9, generator=g), dim=1)
    t = torch.softmax(3 * torch.randn(2, 4, 7, 8, 9, generator=g), dim=1)
    sumdims = [2, 3, 4]
    d = (((2. * t * i).mean(dim=sumdims) + 1e-6) / ((t ** 2).mean(dim=sumdims) + (i ** 2).mean(dim=sumdims) + 1e-6)).mean()
    out.update(dsc_in=npy(i), dsc_tgt=npy(t), dsc=npy(d))
    df = torch.randn(1, 3, 10, 12, 14, generator=g) * 4.0            # large enough for folding voxels
    jd = ls.jacobian_det(df)
    pct = (torch.sum(jd <= 0) / torch.prod(torch.tensor(jd.squeeze().size()))) * 100
    assert 1.0 < float(pct) < 99.0
    out.update(leq_df=npy(df), leq_jdet=npy(jd), leq_pct=npy(pct))
    # landmarks: float coordinates (the datasets store them as floats; .long() truncates), 3 samples of a field
    dfl = torch.randn(3, 3, 10, 12, 14, generator=g) * 2.0
    lm = torch.stack([torch.rand(1, 17, generator=g) * 9.99, torch.rand(1, 17, generator=g) * 11.99, torch.rand(1, 17, generator=g) * 13.99], dim=-1)
    lm[0, 0] = torch.tensor([-1.0, -2.0, -3.0])                       # negative indices wrap in the reference's tensor indexing
    out.update(lm=npy(lm), lm_df=npy(dfl), lm_out=npy(cu.warp_landmarks(lm, dfl).float()))
    save("evalmetrics", **out)


def gen_init_tables():
    """models.py:104-123 evaluated for the (T, L) pairs SURVEY.md §8(c).11 lists"""
    out = {}
    for T, L in ((3, 2), (4, 3), (5, 4), (6, 5), (5, 5), (1, 1)):
        window, kl_w, rec_w, reg_w = weight_dicts(T, L)
        out[f"T{T}L{L}"] = np.array([[window[l], kl_w[l], rec_w[l], reg_w[l]] for l in range(L)], dtype=np.float64)
    save("init_tables", **out)


def gen_state_keys():
    """state-dict key + shape inventory at the BASELINE configs (for checkpoint compatibility tests)"""
    lines = []
    for T, L, size, n0 in ((3, 2, (32, 32, 32), 32), (5, 4, (32, 32, 32), 32)):
        torch.manual_seed(0)
        down = cp.DownPath(T, L, list(size), 2, n0)
        ae = cp.Autoencoder(nb.gauss_sampler, "SVF", T, L, 3, list(size), list(FEEDBACK), "level_res", n0, 3)
        sd = {"downpath." + k: v for k, v in down.state_dict().items()}
        sd.update({"autoencoder." + k: v for k, v in ae.state_dict().items()})
        lines.append(f"# T={T} L={L} size={size} n0={n0}")
        for k, v in sd.items():
            lines.append(f"{T}/{L} {k} {tuple(v.shape)} {str(v.dtype).replace('torch.', '')}")
    with open(os.path.join(HERE, "state_keys.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote state_keys.txt", len(lines), "lines")
    # the shipped default feedback list does not construct (Appendix A.1)
    try:
        cp.Autoencoder(nb.gauss_sampler, "SVF", 3, 2, 3, [16, 16, 16],
                       ["samples", "velocity_field", "individual_dfs", "combined_dfs", "final_dfs", "transformed"], "level_res", 4, 3)
        raise SystemExit("expected ValueError")
    except ValueError as e:
        print("default feedback list ->", e)



# --------------------------------------------------------------------------- 13. 2-D mode (train.py --ndims 2): operators and a full step
def gen_2d():
    g = torch.Generator().manual_seed(77)
    out = {}
    # SpatialTransformer 2-D: random field, zero field (not the identity), image larger than the grid; VecInt
    H, W = 10, 14
    st = nb.SpatialTransformer([H, W])
    df = (torch.randn(2, 2, H, W, generator=g) * 1.5).requires_grad_(True)
    img = torch.rand(2, 3, H, W, generator=g).requires_grad_(True)
    up = torch.randn(2, 3, H, W, generator=g)
    o = st(df.clone(), img)
    gd, gi = torch.autograd.grad((o * up).sum(), [df, img])
    out.update({"w_df": npy(df), "w_img": npy(img), "w_up": npy(up), "w_out": npy(o), "w_gdf": npy(gd), "w_gimg": npy(gi)})
    out["w_zero"] = npy(st(torch.zeros(1, 2, H, W), img[:1].detach()))
    big = torch.rand(1, 1, 2 * H, 2 * W, generator=g)
    out.update({"w_big": npy(big), "w_big_out": npy(st(df[:1].detach().clone(), big))})
    vi = nb.VecInt([H, W], 7)
    v = (torch.randn(1, 2, H, W, generator=g) * 2).requires_grad_(True)
    vo = vi(v)
    gv, = torch.autograd.grad((vo * up[:1, :2]).sum(), [v])
    out.update({"vi_in": npy(v), "vi_out": npy(vo), "vi_g": npy(gv)})
    # pooling / interpolation / ResizeTransform
    x = torch.rand(2, 5, 9, 12, generator=g)
    out.update({"r_x": npy(x), "r_pool": npy(F.avg_pool2d(x, 2, 2, 0, ceil_mode=True)), "r_up": npy(F.interpolate(x, size=[18, 24], mode="bilinear", align_corners=False)),
                "r_down": npy(F.interpolate(x, size=[5, 7], mode="bilinear", align_corners=False))})
    f2 = torch.randn(1, 2, 6, 8, generator=g)
    out.update({"r_f": npy(f2), "r_rt_up": npy(nb.ResizeTransform(0.5, 2)(f2)), "r_rt_down": npy(nb.ResizeTransform(2.0, 2)(f2))})
    # ConvUnit 2-D (train + eval), MuSigmaBlock 2-D
    torch.manual_seed(5)
    cu = nb.ConvUnit([12, 10], 6, 10)
    xc = torch.randn(2, 6, 12, 10, generator=g).requires_grad_(True)
    upc = torch.randn(2, 10, 12, 10, generator=g)
    out.update({"cu_sd0." + k: npy(v) for k, v in cu.state_dict().items()})
    cu.train()
    oc = cu(xc)
    grads = torch.autograd.grad((oc * upc).sum(), [xc] + list(cu.parameters()))
    out.update({"cu_x": npy(xc), "cu_up": npy(upc), "cu_out": npy(oc), "cu_gx": npy(grads[0])})
    out.update({"cu_g." + k: npy(gv_) for (k, _), gv_ in zip(cu.named_parameters(), grads[1:])})
    out.update({"cu_sd1." + k: npy(v) for k, v in cu.state_dict().items()})
    cu.eval()
    out["cu_out_eval"] = npy(cu(xc.detach()))
    ms = nb.MuSigmaBlock([12, 10], 6, 2)
    mu, sg = ms(xc.detach())
    out.update({"ms_sd." + k: npy(v) for k, v in ms.state_dict().items()})
    out.update({"ms_mu": npy(mu), "ms_sigma": npy(sg)})
    # losses: NCC (w = 3, 5, 7), L2_reg, jacobian_det / JDetStd, KL_nondiagonal, all 2-D
    a = smooth_volume(g, (20, 24), 2)[:, :, 0] if False else torch.rand(2, 1, 20, 24, generator=g)
    b = (0.5 * a + 0.5 * torch.rand(2, 1, 20, 24, generator=g)).requires_grad_(True)
    out.update({"l_a": npy(a), "l_b": npy(b)})
    for w in (3, 5, 7):
        l = ls.NCC_loss(b, a, win_size=w, gamma=0.05)
        gb, = torch.autograd.grad(l, [b])
        out.update({f"l_ncc{w}": npy(l), f"l_ncc{w}_g": npy(gb)})
    fld = (torch.randn(2, 2, 20, 24, generator=g) * 2).requires_grad_(True)
    l2 = ls.L2_reg(fld, lamb=0.025)
    g2, = torch.autograd.grad(l2, [fld])
    jd = ls.jacobian_det(fld.detach())
    js = ls.JDetStd(fld, lamb=0.7)
    gj, = torch.autograd.grad(js, [fld])
    out.update({"l_fld": npy(fld), "l_l2": npy(l2), "l_l2_g": npy(g2), "l_jdet": npy(jd), "l_jstd": npy(js), "l_jstd_g": npy(gj)})
    mu2 = torch.randn(2, 2, 12, 10, generator=g).requires_grad_(True)
    sg2 = (torch.rand(2, 2, 12, 10, generator=g) + 0.3).requires_grad_(True)
    kn = ls.KL_nondiagonal([12, 10]).loss(None, None, mu2, sg2)
    gk = torch.autograd.grad(kn, [mu2, sg2])
    out.update({"l_mu": npy(mu2), "l_sg": npy(sg2), "l_klnd": npy(kn), "l_klnd_gmu": npy(gk[0]), "l_klnd_gsg": npy(gk[1])})
    save("ops2d", **out)
    t = gen_step("step2d_T3L2_n4_32x24", T=3, L=2, size=[32, 24], n0=4, B=2, seed=210)
    print("   total loss", t)

# --------------------------------------------------------------------------- 15. the reference's own PULPo class (models.py), API level
def _import_reference_models():
    """import /root/reference/src/models.py itself.  It needs pytorch_lightning and torchvision, which this image lacks: two stand-in
    modules are registered for the duration of the import - PLUMBING ONLY: a LightningModule that is an nn.Module
    with save_hyperparameters / log_dict / log / a trainer handle, and no-op make_grid / flow_to_image.  No arithmetic of the path lives in
    them; everything computed below is computed by the reference's own class."""
    import inspect
    import types

    class _HParams(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    class LightningModule(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.logged = {}
            self.trainer = types.SimpleNamespace(should_stop=False, num_val_batches=[1], global_step=0)

        def save_hyperparameters(self):
            frame = inspect.currentframe().f_back
            info = inspect.getargvalues(frame)
            self.hparams = _HParams({k: info.locals[k] for k in info.args if k != "self"})

        def log_dict(self, d, **kw):
            self.logged.update({k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in d.items()})

        def log(self, k, v, **kw):
            self.logged[k] = v

    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = LightningModule
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda imgs, **kw: imgs
    tvu.flow_to_image = lambda flow: flow
    tv.utils = tvu
    sys.modules.setdefault("pytorch_lightning", pl)
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.utils", tvu)
    import src.models as rm
    assert os.path.abspath(rm.__file__).startswith(REF), rm.__file__
    return rm


def gen_models_api(df_resolution, T=3, L=2, size=(16, 16, 16), n0=2, seed=170):
    """The class the callers use (train.py:82-87, evaluate.py:190-273) driven through its own methods: training_step (recon_loss ncc + dice, so
    that transform_segmentation runs inside it), predict_output_samples(N=2), predict(N=2), predict_deterministic, forward, combine_dfs,
    transform_segmentation - models.py:134-196, 312-388.  Noise: every encoder's sampler replaced by mu + sigma * eps_l (the seam of
    pulpo.py:231,261)."""
    rm = _import_reference_models()
    import contextlib
    import io
    torch.manual_seed(seed)
    fb = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
    with contextlib.redirect_stdout(io.StringIO()):
        model = rm.PULPo(T, L, 0.1, list(size), feedback=fb, n0=n0, df_resolution=df_resolution, recon_loss=["ncc", "dice"], segs=True)
    torch.autograd.set_detect_anomaly(False)
    g = torch.Generator().manual_seed(seed + 1)
    B = 2
    y = smooth_volume(g, tuple(size), B)
    x = 0.6 * smooth_volume(g, tuple(size), B) + 0.4 * y
    seg_x = torch.softmax(4 * torch.cat([smooth_volume(g, tuple(size), B) for _ in range(3)], dim=1), dim=1)
    seg_y = torch.softmax(4 * torch.cat([smooth_volume(g, tuple(size), B) for _ in range(3)], dim=1), dim=1)
    o = T - L
    eps = {l: torch.randn(B, 3, *[s // 2 ** (l + o) for s in size], generator=g) for l in range(L)}
    def set_eps(nb_):              # the first nb_ batch rows of the noise (N = 2 samples of one pair use both rows, a single pair the first)
        for l in range(L):
            model.autoencoder.encoders[l].sampler = (lambda mu, sigma, e=eps[l][:nb_]: mu + sigma * e)
    set_eps(2)
    out = {"cfg": np.array([T, L, n0, B] + list(size), dtype=np.int64), "x": npy(x), "y": npy(y), "seg_x": npy(seg_x), "seg_y": npy(seg_y)}
    out.update({f"eps.{l}": npy(e) for l, e in eps.items()})
    out.update({"sd0." + k: npy(v) for k, v in model.state_dict().items() if not k.endswith(".grid")})
    out["hparams.window_size"] = np.array([model.hierarchical_recon_loss.window_size[l] for l in range(L)], dtype=np.int64)
    out["hparams.kl_w"] = np.array([model.hierarchical_kl_loss.weight_dict[l] for l in range(L)])
    out["hparams.rec_w"] = np.array([model.hierarchical_recon_loss.weight_dict[l] for l in range(L)])
    out["hparams.reg_w"] = np.array([model.hierarchical_regularization.weight_dict[l] for l in range(L)])

    # ---- inference API, eval mode (evaluate.py:100 puts the model in eval mode): one pair, N = 2 samples -> the batch of 2 noise tensors
    model.eval()
    x1, y1, s1 = x[:1], y[:1], seg_x[:1]
    with torch.no_grad():
        outs_s, dfs_s = model.predict_output_samples(x1, y1, N=2)
        avg_out, avg_dfs = model.predict(x1, y1, N=2)
        comb, fin = model.combine_dfs(avg_dfs)
        tseg = model.transform_segmentation(fin, s1)
        set_eps(1)
        det_out, det_dfs = model.predict_deterministic(x1, y1)
        set_eps(2)
        fwd = model.forward(x, y)
    for l in range(L):
        out[f"samples.outputs.{l}"] = npy(outs_s[l]); out[f"samples.individual_dfs.{l}"] = npy(dfs_s[l])
        out[f"predict.outputs.{l}"] = npy(avg_out[l]); out[f"predict.avg_dfs.{l}"] = npy(avg_dfs[l])
        out[f"combine.combined.{l}"] = npy(comb[l]); out[f"combine.final.{l}"] = npy(fin[l])
        out[f"transform_segmentation.{l}"] = npy(tseg[l])
        out[f"deterministic.outputs.{l}"] = npy(det_out[l]); out[f"deterministic.individual_dfs.{l}"] = npy(det_dfs[l])
    out["forward"] = npy(fwd)

    # ---- training_step (train mode; the batch is the 8-tuple of the reference's datasets)
    model.train()
    total = model.training_step((x, y, seg_x, seg_y, None, None, None, None), 0)
    out["train.total"] = npy(total)
    for k in ("kl_loss", "reconstruction_loss", "regularization_loss", "total_loss"):
        out["train.logged." + k] = npy(model.logged["train/" + k])
    for l in range(L):
        for k in ("kl loss level", "recon loss level", "regularization loss level"):
            out[f"train.logged.{k.replace(' ', '_')}.{l}"] = npy(model.logged[f"train_levels/{k} {l}"])
    total.backward()
    for k, p_ in model.named_parameters():
        if p_.grad is not None:
            out["grad." + k] = npy(p_.grad)
    opt = model.configure_optimizers()
    out["optimizer.lr"] = np.array(opt.param_groups[0]["lr"])
    out["optimizer.betas"] = np.array(opt.param_groups[0]["betas"])
    save(f"models_api_{df_resolution}_T{T}L{L}_n{n0}_16", **out)
    return float(total)


# --------------------------------------------------------------------------- round 4: the corners of the operator surface
def gen_round4():
    """VelocityField depth 0 / 1 (network_blocks.py:70-79; depth 1 is an UNPADDED 3x3x3 convolution), ResizeTransform with factor < 1 and
    with sizes where `scale_factor` and the size ratio disagree (network_blocks.py:138-149: F.interpolate(scale_factor=...) maps
    coordinates with 1 / scale_factor, not with in / out), MuSigmaBlock with zdim != ndims (network_blocks.py:49-60), a VelocityField
    fed by such a latent."""
    g = torch.Generator().manual_seed(180)
    out = {}

    def with_grads(tag, module, x, extra_inputs=()):
        x = x.clone().requires_grad_(True)
        y = module(x)
        up = torch.randn(y.shape, generator=g)
        params = list(module.parameters())
        grads = torch.autograd.grad((y * up).sum(), [x] + params)
        out.update({f"{tag}.x": npy(x), f"{tag}.up": npy(up), f"{tag}.y": npy(y), f"{tag}.gx": npy(grads[0])})
        out.update({f"{tag}.sd.{k}": npy(v) for k, v in module.state_dict().items()})
        for (n, _), gr in zip(module.named_parameters(), grads[1:]):
            out[f"{tag}.g.{n}"] = npy(gr)

    torch.manual_seed(181)
    with_grads("vf0", nb.VelocityField([9, 8, 10], 3, 8, 0), torch.randn(2, 3, 9, 8, 10, generator=g))
    with_grads("vf1", nb.VelocityField([9, 8, 10], 3, 8, 1), torch.randn(2, 3, 9, 8, 10, generator=g))
    with_grads("vf1_2d", nb.VelocityField([9, 8], 2, 8, 1), torch.randn(2, 2, 9, 8, generator=g))
    with_grads("rs_half_even", nb.ResizeTransform(2, 3), torch.randn(1, 3, 12, 10, 8, generator=g))
    with_grads("rs_half_odd", nb.ResizeTransform(2, 3), torch.randn(2, 3, 9, 7, 5, generator=g))
    with_grads("rs_x1p5", nb.ResizeTransform(1 / 1.5, 3), torch.randn(1, 3, 5, 6, 7, generator=g))
    with_grads("rs_x0p625", nb.ResizeTransform(1.6, 3), torch.randn(1, 3, 16, 8, 11, generator=g))
    with_grads("rs_half_2d", nb.ResizeTransform(2, 2), torch.randn(1, 2, 9, 12, generator=g))
    for zdim in (5, 1):
        ms = nb.MuSigmaBlock([4, 5, 6], 8, zdim)
        x = (torch.randn(2, 8, 4, 5, 6, generator=g) * 2).requires_grad_(True)
        eps = torch.randn(2, zdim, 4, 5, 6, generator=g)
        mu, sigma = ms(x)
        z = mu + sigma * eps
        up = torch.randn(2, zdim, 4, 5, 6, generator=g)
        grads = torch.autograd.grad((z * up).sum() + (mu * mu).sum() + sigma.sum(), [x] + list(ms.parameters()))
        t = f"ms{zdim}"
        out.update({f"{t}.x": npy(x), f"{t}.eps": npy(eps), f"{t}.up": npy(up), f"{t}.mu": npy(mu), f"{t}.sigma": npy(sigma), f"{t}.z": npy(z),
                    f"{t}.gx": npy(grads[0])})
        out.update({f"{t}.sd.{k}": npy(v) for k, v in ms.state_dict().items()})
        for (n, _), gr in zip(ms.named_parameters(), grads[1:]):
            out[f"{t}.g.{n}"] = npy(gr)
    vf = nb.VelocityField([4, 5, 6], 5, 8, 3)
    vf.eval()
    zz = torch.randn(1, 5, 4, 5, 6, generator=g)
    out.update({"vf_z5.z": npy(zz), "vf_z5.y": npy(vf(zz))})
    out.update({"vf_z5.sd." + k: npy(v) for k, v in vf.state_dict().items()})
    save("blocks_r4", **out)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "round4":
        gen_round4()
        # cp_depth = 0 (network_blocks.py:73-74: the latent sample IS the velocity field) through the whole step.  cp_depth = 1 cannot run
        # in the reference's own model: the unpadded convolution shrinks the field by 2 voxels per axis and VecInt's grid no longer fits
        t = gen_step("step_cp0_T3L2_n2_16", T=3, L=2, size=[16, 16, 16], n0=2, B=1, seed=190, cp_depth=0)
        print("   total loss", t)
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "2d":
        gen_2d()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "evalmetrics":
        gen_evalmetrics()
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "models":
        for res in ("level_res", "full_res"):
            print("   training_step loss", gen_models_api(res))
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dice":
        for res in ("full_res", "level_res"):
            print("   total loss", gen_step_dice(f"step_dice_{res}_T3L2_n2_16", res))
        raise SystemExit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "step32":
        # a step large enough (32^3) for the Winograd forward / data-gradient and weight-gradient kernels to be the ones compared
        # with the real reference (the 16^3 cases below run the direct kernels)
        t = gen_step("step_T3L2_n8_32", T=3, L=2, size=[32, 32, 32], n0=8, B=1, seed=150, smooth=True)
        print("   total loss", t)
        raise SystemExit(0)
    gen_warp()
    gen_vecint()
    gen_resample()
    gen_convunit()
    gen_musigma()
    gen_losses()
    t = gen_step("step_T3L2_n4_16", T=3, L=2, size=[16, 16, 16], n0=4, B=2, seed=110)
    print("   total loss", t)
    t = gen_step("step_T4L3_n2_16x24x16", T=4, L=3, size=[16, 24, 16], n0=2, B=1, seed=120, smooth=True)
    print("   total loss", t)
    t = gen_step("step_fullres_T3L2_n2_16", T=3, L=2, size=[16, 16, 16], n0=2, B=1, seed=140, df_resolution="full_res")
    print("   total loss", t)
    gen_metrics()
    gen_init_tables()
    gen_state_keys()
