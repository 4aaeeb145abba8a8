"""2-D mode (train.py --ndims 2: slices instead of volumes) against golden vectors generated from the REAL reference
(tests/golden/make_golden.py 2d -> ops2d.npz, step2d_T3L2_n4_32x24.npz).  Slices run as depth-1 volumes through the same HIP kernels
(pulpo_amd/ops.py, "2-D mode"); the kernels with ndims-dependent arithmetic switch to the reference's 2-D form when the depth is 1.
The CPU oracle is 3-D only, so these fixtures are the pin for this mode.  Tolerances as in the 3-D tests."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy
FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
OUT = ("mus", "sigmas", "samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed")


@pytest.fixture(scope="module")
def api():
    assert torch.cuda.is_available()
    import src.models as models
    import src.network_blocks as nb
    import src.losses as losses
    from pulpo_amd._lib import lib
    lib.load()
    return models, nb, losses


def close(a, b, atol=1e-5, rtol=1e-5):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), atol=atol, rtol=rtol)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_warp_vecint_2d(api, golden):
    _, nb, _ = api
    g = golden("ops2d")
    H, W = g["w_df"].shape[2:]
    st = nb.SpatialTransformer([H, W]).cuda()
    df, img = T(g["w_df"]).cuda().requires_grad_(True), T(g["w_img"]).cuda().requires_grad_(True)
    out = st(df, img)
    close(out, g["w_out"], atol=2e-6)
    gd, gi = torch.autograd.grad((out * T(g["w_up"]).cuda()).sum(), [df, img])
    close(gd, g["w_gdf"], atol=1e-5)
    close(gi, g["w_gimg"], atol=1e-5)
    close(st(torch.zeros(1, 2, H, W, device="cuda"), img[:1].detach()), g["w_zero"], atol=2e-6)        # zero field != identity
    close(st(df[:1].detach(), T(g["w_big"]).cuda()), g["w_big_out"], atol=2e-6)                         # image larger than the grid
    vi = nb.VecInt([H, W], 7).cuda()
    v = T(g["vi_in"]).cuda().requires_grad_(True)
    vo = vi(v)
    close(vo, g["vi_out"], atol=1e-5)
    gv, = torch.autograd.grad((vo * T(g["w_up"]).cuda()[:1, :2]).sum(), [v])
    assert rel_l2(gv, g["vi_g"]) < 1e-4


def test_resample_2d(api, golden):
    from pulpo_amd import ops
    _, nb, _ = api
    g = golden("ops2d")
    x = T(g["r_x"]).cuda()
    close(ops.avg_pool2(x), g["r_pool"], atol=1e-6)
    close(ops.resize_trilinear(x, [18, 24]), g["r_up"], atol=1e-6)
    close(ops.resize_trilinear(x, [5, 7]), g["r_down"], atol=1e-6)
    f = T(g["r_f"]).cuda()
    close(nb.ResizeTransform(0.5, 2)(f), g["r_rt_up"], atol=1e-5)
    close(nb.ResizeTransform(2.0, 2)(f), g["r_rt_down"], atol=1e-5)


def test_conv_unit_and_heads_2d(api, golden):
    _, nb, _ = api
    g = golden("ops2d")
    cu = nb.ConvUnit([12, 10], 6, 10)
    cu.load_state_dict({k[7:]: T(v.copy()) for k, v in g.items() if k.startswith("cu_sd0.")})
    cu = cu.cuda().train()
    x = T(g["cu_x"]).cuda().requires_grad_(True)
    out = cu(x)
    close(out, g["cu_out"], atol=2e-5)
    grads = torch.autograd.grad((out * T(g["cu_up"]).cuda()).sum(), [x] + list(cu.parameters()))
    assert rel_l2(grads[0], g["cu_gx"]) < 1e-4
    for (k, _), gv in zip(cu.named_parameters(), grads[1:]):
        ref = g["cu_g." + k]
        if k.endswith("_op.0.bias"):
            assert float(gv.abs().max()) <= 1e-3 * max(1e-3, float(np.abs(g["cu_g._op.0.weight"]).max()))     # true gradient 0 (BatchNorm follows)
        else:
            assert rel_l2(gv, ref) < 1e-4, k
    for k, v in cu.state_dict().items():
        if "running" in k or "num_batches" in k:
            close(v, g["cu_sd1." + k], atol=1e-6)
    cu.eval()
    close(cu(x.detach()), g["cu_out_eval"], atol=2e-5)
    ms = nb.MuSigmaBlock([12, 10], 6, 2)
    ms.load_state_dict({k[6:]: T(v.copy()) for k, v in g.items() if k.startswith("ms_sd.")})
    mu, sg = ms.cuda()(x.detach())
    close(mu, g["ms_mu"], atol=1e-5)
    close(sg, g["ms_sigma"], atol=1e-5)


def test_losses_2d(api, golden):
    _, _, L = api
    g = golden("ops2d")
    a = T(g["l_a"]).cuda()
    for w in (3, 5, 7):
        b = T(g["l_b"]).cuda().requires_grad_(True)
        l = L.NCC_loss(b, a, win_size=w, gamma=0.05)
        np.testing.assert_allclose(float(l), float(g[f"l_ncc{w}"]), rtol=1e-4)
        gb, = torch.autograd.grad(l, [b])
        assert rel_l2(gb, g[f"l_ncc{w}_g"]) < 2e-3, w
    f = T(g["l_fld"]).cuda().requires_grad_(True)
    l2 = L.L2_reg(f, lamb=0.025)
    np.testing.assert_allclose(float(l2), float(g["l_l2"]), rtol=1e-5)
    g2, = torch.autograd.grad(l2, [f])
    assert rel_l2(g2, g["l_l2_g"]) < 1e-5
    close(L.jacobian_det(f.detach()), g["l_jdet"], atol=1e-5)
    js = L.JDetStd(f, lamb=0.7)
    np.testing.assert_allclose(float(js), float(g["l_jstd"]), rtol=1e-4)
    gj, = torch.autograd.grad(js, [f])
    assert rel_l2(gj, g["l_jstd_g"]) < 1e-4
    mu, sg = T(g["l_mu"]).cuda().requires_grad_(True), T(g["l_sg"]).cuda().requires_grad_(True)
    kn = L.KL_nondiagonal([12, 10]).loss(None, None, mu, sg)
    np.testing.assert_allclose(float(kn), float(g["l_klnd"]), rtol=1e-5)
    gm, gs = torch.autograd.grad(kn, [mu, sg])
    assert rel_l2(gm, g["l_klnd_gmu"]) < 1e-5 and rel_l2(gs, g["l_klnd_gsg"]) < 1e-5


def test_training_step_2d_matches_reference_golden(api, golden):
    models, nb, _ = api
    g = golden("step2d_T3L2_n4_32x24")
    Tl, L, n0, B, *size = [int(v) for v in g["cfg"]]
    assert len(size) == 2
    model = models.PULPo(Tl, L, 0.1, size, feedback=FB, n0=n0)
    sd = model.state_dict()
    loaded = 0
    for k, v in g.items():
        if k.startswith("sd0."):
            assert k[4:] in sd and tuple(sd[k[4:]].shape) == v.shape, k
            sd[k[4:]] = T(v.copy())
            loaded += 1
    assert loaded > 50
    model.load_state_dict(sd)
    model = model.cuda().train()
    for l in range(L):
        model.autoencoder.encoders[l].sampler = nb.FixedNoiseSampler(T(g[f"eps.{l}"]).cuda())
    x, y = T(g["x"]).cuda(), T(g["y"]).cuda()
    outs, _, (total, kl, rec, reg), levels = model._forward_and_losses(x, y)
    for name, d in zip(OUT, outs):
        for l, v in d.items():
            ref = g[f"train.{name}.{l}"]
            assert tuple(v.shape) == ref.shape, (name, l, v.shape, ref.shape)
            err = np.abs(v.detach().cpu().numpy() - ref).max()
            assert err <= 1e-4 * max(1.0, np.abs(ref).max()), (name, l, err)
    for key, val in zip(("total", "kl", "rec", "reg"), (total, kl, rec, reg)):
        np.testing.assert_allclose(float(val), float(g["train." + key]), rtol=1e-4)
    total.backward()
    checked = 0
    for k, p in model.named_parameters():
        if "grad." + k in g:
            ref = g["grad." + k]
            if k.endswith("_op.0.bias") and "velocity_field._op.2" not in k:
                wref = np.abs(g["grad." + k[:-4] + "weight"]).max()
                assert np.abs(p.grad.cpu().numpy()).max() <= 1e-3 * max(wref, 1e-3), k
                continue
            assert rel_l2(p.grad, ref) < 1e-3, (k, rel_l2(p.grad, ref))
            checked += 1
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    assert checked > 40
    sd1 = model.state_dict()
    for k, v in g.items():
        if k.startswith("sd1."):
            np.testing.assert_allclose(sd1[k[4:]].cpu().numpy(), v, atol=1e-5, rtol=1e-5)
    model.eval()
    with torch.no_grad():
        outs_e, _, (tot_e, *_), _ = model._forward_and_losses(x, y)
        np.testing.assert_allclose(float(tot_e), float(g["eval.total"]), rtol=1e-4)
        det_out, det_ind = model.predict_deterministic(x, y)
        for l in det_out:
            np.testing.assert_allclose(det_out[l].cpu().numpy(), g[f"det.transformed.{l}"], atol=1e-4)
            np.testing.assert_allclose(det_ind[l].cpu().numpy(), g[f"det.individual_dfs.{l}"], atol=1e-4)
        # predict_output_samples / predict on slices (reference models.py:312-331 stacks with torch.vstack, rank-agnostic): N copies on
        # the batch axis with the SAME injected noise -> every sample equals the single stochastic eval forward
        for l in range(L):
            e = model.autoencoder.encoders[l].sampler.fixed_eps
            model.autoencoder.encoders[l].sampler = nb.FixedNoiseSampler(e.repeat(2, 1, 1, 1))
        o_s, d_s = model.predict_output_samples(x, y, N=2)
        for l in o_s:
            assert tuple(o_s[l].shape[:3]) == (B, 2, 1) and tuple(d_s[l].shape[:3]) == (B, 2, 2) and o_s[l].dim() == 5
            np.testing.assert_allclose(o_s[l][:, 0].cpu().numpy(), g[f"eval.transformed.{l}"], atol=1e-4)
            np.testing.assert_allclose(d_s[l][:, 1].cpu().numpy(), g[f"eval.individual_dfs.{l}"], atol=1e-4)
        avg_out, avg_dfs = model.predict(x, y, N=2)
        for l in avg_dfs:
            np.testing.assert_allclose(avg_dfs[l].cpu().numpy(), g[f"eval.individual_dfs.{l}"], atol=1e-4)
            want = tuple(size) if l == 0 else tuple(avg_dfs[l].shape[2:])          # level 0's field is resized to full resolution (models.py:364-366)
            assert tuple(avg_out[l].shape) == (B, 1) + want and bool(torch.isfinite(avg_out[l]).all())
        np.testing.assert_allclose(avg_out[0].cpu().numpy(), g["eval.transformed.0"], atol=1e-4)
