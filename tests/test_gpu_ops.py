"""GPU parity tests: every HIP operator (through the C ABI, via pulpo_amd.ops) against
  (1) the golden vectors produced by the real reference (tests/golden), and
  (2) the CPU oracle (oracle/pulpo_oracle.py) on seeded random inputs, including ragged / odd sizes.
Stated fp32 tolerances: fields / warped volumes atol 1e-4, loss terms rtol 1e-4,
gradients relative-L2 <= 1e-3 (<= 32^3 cases)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pulpo_oracle as O

pytestmark = pytest.mark.gpu

T = torch.from_numpy


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    from pulpo_amd import ops as _ops
    from pulpo_amd._lib import lib
    lib.load()
    return _ops


def dev(a):
    t = T(a) if isinstance(a, np.ndarray) else a
    return t.to("cuda")


def close(a, b, atol=1e-5, rtol=1e-5):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


def rel_l2(a, b):
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))


# ================================================================================================ warp / vecint
def test_warp_golden(ops, golden):
    g = golden("warp3d")
    df, img = dev(g["a_df"]).requires_grad_(True), dev(g["a_img"]).requires_grad_(True)
    out = ops.warp3d(df, img)
    close(out, g["a_out"], atol=2e-6)
    gdf, gimg = torch.autograd.grad((out * dev(g["a_up"])).sum(), [df, img])
    close(gdf, g["a_gdf_rand"], atol=1e-5)
    close(gimg, g["a_gimg_rand"], atol=1e-5)
    close(ops.warp3d(torch.zeros(1, 3, 6, 8, 10, device="cuda"), dev(g["b_img"])), g["b_out"], atol=2e-6)   # zero field != identity
    close(ops.warp3d(dev(g["c_df"]), dev(g["c_df"])), g["c_out"], atol=3e-6)
    close(ops.warp3d(dev(g["d_df"]), dev(g["d_img"])), g["d_out"], atol=2e-6)                                # image larger than grid
    # self-warp gradient (field is also the image)
    v = dev(g["c_df"]).requires_grad_(True)
    gv, = torch.autograd.grad((ops.warp3d(v, v) * dev(g["c_up"])).sum(), [v])
    close(gv, g["c_gdf"], atol=1e-5)


def test_warp_large_random_vs_oracle(ops):
    gen = torch.Generator().manual_seed(5)
    df = torch.randn(2, 3, 17, 20, 33, generator=gen) * 3
    img = torch.rand(2, 1, 17, 20, 33, generator=gen)
    close(ops.warp3d(df.cuda(), img.cuda()), O.warp(df, img), atol=1e-5)
    # constant image is a fixed point whatever the field (size-independent property)
    c = torch.full((1, 1, 40, 40, 40), 0.37, device="cuda")
    out = ops.warp3d(torch.randn(1, 3, 40, 40, 40, device="cuda") * 5, c)
    assert float((out - 0.37).abs().max()) < 1e-6


def test_vecint_golden(ops, golden):
    g = golden("vecint")
    for s in ("", "2"):
        v = dev(g["v" + s]).requires_grad_(True)
        out = ops.vecint(v, 7)
        close(out, g["out" + s], atol=1e-5)
        gv, = torch.autograd.grad((out * dev(g["up" + s])).sum(), [v])
        assert rel_l2(gv, g["gv" + s]) < 1e-4


@pytest.mark.parametrize("size,amp", [((24, 20, 28), 1.0), ((16, 32, 18), 12.0), ((40, 40, 40), 3.0), ((10, 10, 10), 1.5), ((8, 13, 9), 4.0), ((6, 7, 5), 1.0)])
def test_vecint_backward_lds_tiled_scatter_vs_oracle(ops, size, amp):
    """fields of 8^3 and up (round 5; 16^3 before - the last case stays on the plain scatter) run the backward squaring steps with the scatter collected in LDS boxes (vecint_bwd_tile_kernel): small
    displacements stay inside a tile's box, large ones (amp 12: several voxels per step at the end) take the memory-atomic fallback; the
    identity path rides in the box (no copy per step: the step buffers are zeroed by one fill);
    both against autograd through the oracle's VecInt in float64"""
    gen = torch.Generator().manual_seed(int(amp * 10) + size[0])
    v = torch.randn(2, 3, *size, generator=gen) * amp
    up = torch.randn(2, 3, *size, generator=gen)
    vg = v.cuda().requires_grad_(True)
    out = ops.vecint(vg, 7)
    gv, = torch.autograd.grad((out * up.cuda()).sum(), [vg])
    vr = v.double().requires_grad_(True)
    ref = O.vecint(vr, 7)
    gr, = torch.autograd.grad((ref * up.double()).sum(), [vr])
    assert rel_l2(out, ref) < 1e-5
    # a rough field puts some sample coordinates within rounding of a cell boundary, where floor() - and with it the gradient - jumps: the
    # fp32 CPU evaluation of the same operator sets the scale of that effect
    v32 = v.clone().requires_grad_(True)
    g32, = torch.autograd.grad((O.vecint(v32, 7) * up).sum(), [v32])
    assert rel_l2(gv, gr) < max(1e-4, 3.0 * rel_l2(g32, gr))


@pytest.mark.parametrize("B,size", [(1, (20, 20, 20)), (2, (10, 10, 10)), (1, (12, 14, 10)), (1, (6, 7, 5)), (2, (16, 24, 20))])
def test_vecint_forward_fused_in_lds_equals_the_step_by_step_kernels(ops, B, size):
    """fields of <= 8192 voxels (the 20^3 and 10^3 pyramid levels and BASELINE config 5's coarse levels) run all seven squaring steps in ONE
    launch with the field resident in LDS (vecint_fwd_lds_kernel; network_blocks.py:160-177): compared with the operator evaluated step by
    step through the warp kernel (v / 2^7, then seven times v + warp(v, v)) - the same arithmetic, to fp32 rounding at worst - and with the
    float64 oracle; the backward pass (which reads the intermediate fields the fused kernel saved) against autograd through the oracle"""
    gen = torch.Generator().manual_seed(B * 100 + size[0])
    v = torch.randn(B, 3, *size, generator=gen) * 2.0
    up = torch.randn(B, 3, *size, generator=gen)
    vg = v.cuda().requires_grad_(True)
    out = ops.vecint(vg, 7)
    gv, = torch.autograd.grad((out * up.cuda()).sum(), [vg])
    with torch.no_grad():
        w = v.cuda() * (1.0 / 128.0)
        for _ in range(7):
            w = w + ops.warp3d(w, w)
    assert float((out - w).abs().max()) <= 1e-5 * max(1.0, float(w.abs().max()))      # (measured 1e-6: contraction / add order of seven squarings)
    vr = v.double().requires_grad_(True)
    ref = O.vecint(vr, 7)
    gr, = torch.autograd.grad((ref * up.double()).sum(), [vr])
    assert rel_l2(out, ref) < 1e-5
    v32 = v.clone().requires_grad_(True)
    g32, = torch.autograd.grad((O.vecint(v32, 7) * up).sum(), [v32])
    assert rel_l2(gv, gr) < max(1e-4, 3.0 * rel_l2(g32, gr))


# ================================================================================================ resampling
@pytest.mark.parametrize("size,C", [((8, 12, 10), 16), ((7, 9, 5), 8), ((6, 6, 6), 1)])
def test_avg_pool_with_skip_gradient_meets_the_pooled_one_in_one_kernel(ops, size, C):
    """DownPath hands an activation to the next level (pooled) and to the decoder (skip connection): ops.avg_pool2_skip returns both and its
    backward pass forms skip gradient + pooling backward in one pass - here with the skip gradient a channel slice of a concatenation's
    gradient, as in the model (components/pulpo.py:58, 77).  Same values as the separate operators (the sum is one fp32 addition either
    way: bit-identical); odd sizes (ceil-mode windows) and the single-channel case, which takes the separate operators"""
    gen = torch.Generator().manual_seed(C + size[0])
    x = torch.randn(2, C, *size, generator=gen).cuda()
    x = x.contiguous(memory_format=torch.channels_last_3d) if C > 1 else x
    other = torch.randn(2, 4, *size, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    wcat = torch.randn(2, C + 4, *size, generator=gen).cuda()
    psize = [(s + 1) // 2 for s in size]
    wp = torch.randn(2, C, *psize, generator=gen).cuda()
    res = []
    for fused in (False, True):
        xg = x.clone().requires_grad_(True)
        h = xg * 1.0
        if fused:
            skip, pooled = ops.avg_pool2_skip(h)
        else:
            skip, pooled = h, ops.avg_pool2(h)
        loss = (torch.cat([other, skip], dim=1) * wcat).sum() + (pooled * wp).sum()
        g, = torch.autograd.grad(loss, [xg])
        res.append((pooled.detach(), g))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])
    ref = F.avg_pool3d(x.cpu().double(), 2, 2, ceil_mode=True)
    assert rel_l2(res[1][0], ref) < 1e-6


@pytest.mark.parametrize("size,cin,cout", [((16, 16, 16), 8, 16), ((10, 9, 7), 4, 8)])
def test_last_unit_of_an_encoder_level_writes_the_pooled_tensor_too(ops, size, cin, cout):
    """DownPath pools the output of every level's ConvSequence (components/pulpo.py:58): the last unit's BatchNorm / LeakyReLU pass writes
    AvgPool(z) along with z (pulpo_bn_lrelu_apply_pool2) and ops.avg_pool2_skip picks it up instead of reading z again.  Same z, same
    pooled tensor, same gradients as the separate passes (same expressions, same summation order); even and odd (ceil-mode) sizes"""
    from pulpo_amd.network_blocks import ConvSequence
    x = torch.randn(1, cin, *size, generator=torch.Generator().manual_seed(3)).cuda()
    res = []
    for fused in (False, True):
        torch.manual_seed(11)
        seq = ConvSequence(list(size), cin, cout, 2).cuda().train()
        xg = x.clone().requires_grad_(True)
        z = seq(xg, pool_after=fused)
        assert hasattr(z, "_pulpo_pooled") == fused
        skip, pooled = ops.avg_pool2_skip(z)
        loss = (skip * skip).sum() + (pooled * torch.arange(pooled.numel(), device="cuda").view_as(pooled).float().sin()).sum()
        grads = torch.autograd.grad(loss, [xg] + list(seq.parameters()))
        res.append((z.detach(), pooled.detach(), grads))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert rel_l2(res[1][2][0], res[0][2][0]) < 1e-6
    ref = F.avg_pool3d(res[0][0].cpu().double(), 2, 2, ceil_mode=True)
    assert rel_l2(res[1][1], ref) < 1e-6
    # the backward pass of avg_pool2_skip also delivers the last unit's BatchNorm-backward sums (pulpo_avgpool2_bwd_bnred): same partial
    # rows as the separate reduction pass, hence the same gradients for every parameter
    ops.BN_REDUCE_IN_DGRAD = False
    try:
        torch.manual_seed(11)
        seq = ConvSequence(list(size), cin, cout, 2).cuda().train()
        xg = x.clone().requires_grad_(True)
        skip, pooled = ops.avg_pool2_skip(seq(xg, pool_after=True))
        loss = (skip * skip).sum() + (pooled * torch.arange(pooled.numel(), device="cuda").view_as(pooled).float().sin()).sum()
        plain = torch.autograd.grad(loss, [xg] + list(seq.parameters()))
    finally:
        ops.BN_REDUCE_IN_DGRAD = True
    scale = max(float(g.abs().max()) for g in plain)
    for a_, b_ in zip(res[1][2], plain):
        assert float((a_ - b_).abs().max()) <= 2e-5 * scale


def test_resample_golden(ops, golden):
    g = golden("resample")
    x = dev(g["rt_x"]).requires_grad_(True)
    o = ops.resize_trilinear(x, (8, 10, 12), 2.0)                      # ResizeTransform(1/2)
    close(o, g["rt_out"], atol=1e-6)
    close(torch.autograd.grad((o * dev(g["rt_up"])).sum(), [x])[0], g["rt_gx"], atol=1e-5)
    x = dev(g["up2_x"]).requires_grad_(True)
    o = ops.resize_trilinear(x, (8, 12, 10))
    close(o, g["up2_out"], atol=1e-6)
    close(torch.autograd.grad((o * dev(g["up2_up"])).sum(), [x])[0], g["up2_gx"], atol=1e-5)
    y = dev(g["dn_y"])
    for f in (1, 2, 4, 8):
        close(ops.resize_trilinear(y, (16 // f, 16 // f, 24 // f)), g[f"dn_out{f}"], atol=1e-6)
    close(ops.resize_trilinear(dev(g["gen_x"]), (8, 9, 11)), g["gen_out"], atol=1e-6)      # non-integer ratio
    # generic (atomic) backward path on the non-integer ratio
    xg = dev(g["gen_x"]).requires_grad_(True)
    up = torch.randn(1, 2, 8, 9, 11)
    gx, = torch.autograd.grad((ops.resize_trilinear(xg, (8, 9, 11)) * up.cuda()).sum(), [xg])
    xc = T(g["gen_x"]).requires_grad_(True)
    gref, = torch.autograd.grad((O.resize_to(xc, (8, 9, 11)) * up).sum(), [xc])
    close(gx, gref, atol=1e-5)


@pytest.mark.parametrize("tag", ["odd", "even"])
def test_avgpool_golden(ops, golden, tag):
    g = golden("resample")
    x = dev(g[f"pool_{tag}_x"]).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    o = ops.avg_pool2(x)
    close(o, g[f"pool_{tag}_out"], atol=1e-6)
    close(torch.autograd.grad((o * dev(g[f"pool_{tag}_up"])).sum(), [x])[0], g[f"pool_{tag}_gx"], atol=1e-6)


def test_feedback_up2_vs_oracle(ops):
    gen = torch.Generator().manual_seed(7)
    srcs = [torch.randn(2, c, 3, 5, 4, generator=gen) for c in (3, 3, 3, 3, 3, 1)]
    up = torch.randn(2, 16, 6, 10, 8, generator=gen)
    cs = [s.clone().requires_grad_(True) for s in srcs]
    ref = torch.cat([O.resize_to(s, (6, 10, 8)) for s in cs], dim=1)
    gref = torch.autograd.grad((ref * up).sum(), cs)
    ds = [s.cuda().requires_grad_(True) for s in srcs]
    out = ops.feedback_up2(ds)
    close(out, ref, atol=1e-6)
    gout = torch.autograd.grad((out * up.cuda()).sum(), ds)
    for a, b in zip(gout, gref):
        close(a, b, atol=1e-5)


# ================================================================================================ ConvUnit
@pytest.mark.parametrize("tag", ["a", "b"])
def test_conv_unit_golden(ops, golden, tag):
    g = golden("convunit")
    sd = {k[len(tag) + 5:]: dev(v.copy()) for k, v in g.items() if k.startswith(tag + "_sd0.")}
    w, b = sd["_op.0.weight"].requires_grad_(True), sd["_op.0.bias"].requires_grad_(True)
    gam, bet = sd["_op.1.weight"].requires_grad_(True), sd["_op.1.bias"].requires_grad_(True)
    rm, rv = sd["_op.1.running_mean"], sd["_op.1.running_var"]
    x = dev(g[tag + "_x"]).requires_grad_(True)
    out = ops.conv_bn_lrelu(x, w, b, gam, bet, rm, rv, training=True)
    close(out, g[tag + "_out_train"], atol=2e-5)
    grads = torch.autograd.grad((out * dev(g[tag + "_up"])).sum(), [x, w, b, gam, bet])
    for got, key in zip(grads, ("gx", "gw", "gb", "ggamma", "gbeta")):
        ref = g[f"{tag}_{key}"]
        if key == "gb":          # bias feeding BatchNorm: true gradient is 0, the reference holds rounding noise
            assert np.abs(got.cpu().numpy()).max() <= 1e-4 * max(1.0, np.abs(g[f"{tag}_gw"]).max())
            continue
        assert rel_l2(got, ref) < 1e-4, key
    close(rm, g[f"{tag}_sd1._op.1.running_mean"], atol=1e-6)
    close(rv, g[f"{tag}_sd1._op.1.running_var"], atol=1e-6)
    oe = ops.conv_bn_lrelu(x, w, b, gam, bet, rm, rv, training=False)
    close(oe, g[tag + "_out_eval"], atol=2e-5)


@pytest.mark.parametrize("size,cin,cout", [((10, 10, 10), 192, 192), ((20, 20, 20), 288, 192), ((32, 32, 32), 32, 64)])
def test_conv_unit_eval_mode_fused_store_equals_the_separate_passes(ops, size, cin, cout):
    """inference (no gradient): the convolution's store applies the eval-mode BatchNorm + LeakyReLU (one kernel per ConvUnit) - on the
    split-K shapes (10^3, 20^3 with >= 192 reduction channels) the ordered reduction of the partial slabs does it.  Same values as the
    path that keeps the pre-norm tensor for a backward pass (convolution, then bn_lrelu_apply)."""
    gen = torch.Generator().manual_seed(cin + size[0])
    x = torch.randn(1, cin, *size, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=gen) / (27 * cin) ** 0.5).cuda()
    b, gam, bet = torch.randn(cout, generator=gen).cuda(), torch.rand(cout, generator=gen).cuda() + 0.5, torch.randn(cout, generator=gen).cuda()
    rm, rv = torch.randn(cout, generator=gen).cuda() * 0.1, torch.rand(cout, generator=gen).cuda() + 0.5
    with torch.no_grad():
        fused = ops.conv_bn_lrelu(x, w, b, gam, bet, rm, rv, training=False)
    xg = x.clone().requires_grad_(True)
    plain = ops.conv_bn_lrelu(xg, w, b, gam, bet, rm, rv, training=False)
    assert float((fused - plain.detach()).abs().max()) <= 2e-6 * float(plain.detach().abs().max())
    ref = F.leaky_relu(F.batch_norm(F.conv3d(x.cpu().double(), w.cpu().double(), b.cpu().double(), padding=1), rm.cpu().double(), rv.cpu().double(),
                                    gam.cpu().double(), bet.cpu().double(), False, 0.1, 1e-5), 0.2)
    assert rel_l2(fused, ref) < 3e-6


CONV_CASES = [
    # B, Cin, Cout, size, channels-last input?
    (1, 2, 32, (9, 11, 13), False),
    (2, 3, 32, (8, 8, 8), False),
    (1, 16, 96, (6, 10, 9), True),
    (1, 32, 32, (16, 16, 16), True),
    (2, 32, 64, (5, 9, 17), True),
    (1, 64, 64, (8, 8, 16), True),
    (1, 160, 64, (4, 8, 8), True),
    (1, 20, 12, (7, 6, 5), True),       # odd channel counts (generic path)
    (1, 192, 192, (3, 5, 4), True),
    (1, 32, 3, (6, 7, 8), True),
    # whole 4x8x8 tiles from 64^3 up with <= 4 input channels: the persistent kernel of the input layers (planar and channels-last operands, two / three cout tiles)
    (1, 2, 32, (64, 64, 64), False),
    (2, 3, 64, (64, 72, 64), False),
    (1, 4, 96, (64, 64, 72), True),
]


@pytest.mark.parametrize("B,Cin,Cout,size,cl", CONV_CASES)
def test_conv3d_vs_oracle(ops, B, Cin, Cout, size, cl):
    gen = torch.Generator().manual_seed(B * 1000 + Cin * 10 + Cout)
    x = torch.randn(B, Cin, *size, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=gen) / (27 * Cin) ** 0.5
    b = torch.randn(Cout, generator=gen)
    up = torch.randn(B, Cout, *size, generator=gen)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv3d(xr.double(), wr.double(), br.double(), padding=1)
    gref = torch.autograd.grad((ref * up.double()).sum(), [xr, wr, br])
    xd = x.cuda()
    if cl:
        xd = xd.contiguous(memory_format=torch.channels_last_3d)
    xd.requires_grad_(True)
    wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    out = ops.conv3d_k3(xd, wd, bd)
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < 2e-6
    gx, gw, gb = torch.autograd.grad((out * up.cuda()).sum(), [xd, wd, bd])
    assert rel_l2(gx, gref[0]) < 2e-6
    assert rel_l2(gw, gref[1]) < 1e-5
    assert rel_l2(gb, gref[2]) < 1e-5


@pytest.mark.parametrize("Cin,Cout,size", [(2, 32, (64, 64, 64)), (3, 64, (64, 72, 64)), (32, 48, (32, 32, 32)), (48, 16, (32, 32, 64))])
def test_narrow_input_unit_training_pass_vs_oracle(ops, Cin, Cout, size):
    """a training-mode ConvUnit on a 2- / 3-channel image at a size the persistent input-layer kernel takes (whole 4x8x8 tiles from 64^3 up):
    output, running statistics (the kernel's per-tile partial sums, flushed a tile late) and parameter gradients against the oracle in double.
    Also: 48 and 16 output channels through the F(2x2x2,3x3x3) kernel - cout tiles of 32 that are half empty (statistics and stores masked)."""
    if Cin > 4:
        assert ops.lib.query("pulpo_conv3d_k3_algo", 2, *size, Cin, Cout) == 3
    import src.network_blocks as nb
    gen = torch.Generator().manual_seed(7 * Cin + Cout)
    torch.manual_seed(11)
    unit = nb.ConvUnit(list(size), Cin, Cout)
    with torch.no_grad():
        unit._op[1].weight.copy_(torch.rand(Cout, generator=gen) * 0.5 + 0.75)
        unit._op[1].bias.copy_(torch.randn(Cout, generator=gen) * 0.1)
    sd = {"u." + k: v.detach().clone() for k, v in unit.state_dict().items()}
    x = torch.randn(2, Cin, *size, generator=gen)
    up = torch.randn(2, Cout, *size, generator=gen)
    unit = unit.cuda().train()
    names = ["u._op.0.weight", "u._op.0.bias", "u._op.1.weight", "u._op.1.bias"]

    def oracle(dtype):
        sdr = {k: (v.to(dtype).requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.to(dtype) if v.is_floating_point() else v.clone()))
               for k, v in sd.items()}
        zr = O.conv_unit(x.to(dtype), sdr, "u", training=True)
        return zr, torch.autograd.grad((zr * up.to(dtype)).sum(), [sdr[n] for n in names]), sdr

    zr, gr, sdr = oracle(torch.float64)
    z32, g32, _ = oracle(torch.float32)
    z = unit(x.cuda())
    g = torch.autograd.grad((z * up.cuda()).sum(), [unit._op[0].weight, unit._op[0].bias, unit._op[1].weight, unit._op[1].bias])
    assert rel_l2(z, zr) < 3e-6
    close(unit._op[1].running_mean, sdr["u._op.1.running_mean"].float(), atol=1e-6, rtol=1e-5)
    close(unit._op[1].running_var, sdr["u._op.1.running_var"].float(), atol=1e-6, rtol=1e-5)
    # gradients behind a training-mode BatchNorm cancel heavily (the conv's weight / bias gradients are sums of a zero-mean tensor): the bound is
    # the fp32 oracle's own distance from the double evaluation
    for got, want, o32, nme in zip(g, gr, g32, names):
        if nme.endswith("0.bias"):
            assert float(got.abs().max()) <= 1e-3 * float(g[2].abs().max()), nme      # (exactly zero in exact arithmetic)
            continue
        bound = 3.0 * rel_l2(o32, want) + 1e-5
        assert rel_l2(got, want) <= bound, (nme, rel_l2(got, want), bound)


@pytest.mark.parametrize("B,Cin,Cout,size", [(4, 64, 128, (16, 16, 16)), (3, 32, 192, (16, 16, 16)), (2, 32, 256, (16, 16, 16))])
def test_training_statistics_on_small_volumes_with_many_work_items(ops, B, Cin, Cout, size):
    """whole 4x8x8 tiles, >= 256 (tile, cout tile) work items, but fewer than 20^3 voxels: pulpo_conv3d_k3_stat_tiles() counts 2-deep tiles there
    while the F(2x2x2,3x3x3) kernel writes one statistics row per 4-deep tile (round-4 advisor finding: half of the statistics buffer stayed
    uninitialised and bn_fwd_finalize summed it).  The policy must not pick that kernel for such a volume, its entry point must refuse the call,
    and the training-mode unit's output and running statistics must match the oracle in double."""
    assert ops.lib.query("pulpo_conv3d_k3_algo", B, *size, Cin, Cout) != 3
    import src.network_blocks as nb
    gen = torch.Generator().manual_seed(3 * Cin + Cout + B)
    torch.manual_seed(5)
    unit = nb.ConvUnit(list(size), Cin, Cout)
    sd = {"u." + k: v.detach().clone().double() if v.is_floating_point() else v.clone() for k, v in unit.state_dict().items()}
    x = torch.randn(B, Cin, *size, generator=gen)
    zr = O.conv_unit(x.double(), sd, "u", training=True)
    unit = unit.cuda().train()
    z = unit(x.cuda())
    assert rel_l2(z, zr) < 3e-6
    close(unit._op[1].running_mean, sd["u._op.1.running_mean"].float(), atol=1e-6, rtol=1e-5)
    close(unit._op[1].running_var, sd["u._op.1.running_var"].float(), atol=1e-6, rtol=1e-5)
    # the entry point itself refuses statistics rows it cannot count
    import ctypes
    xc = x.cuda().contiguous(memory_format=torch.channels_last_3d)
    y = ops.new_cl(B, Cout, *size, "cuda")
    wp = torch.empty(ops.lib.query("pulpo_conv3d_k3_packed_wino3_floats", Cin, Cout), device="cuda")
    stats = torch.empty(ops.lib.query("pulpo_conv3d_k3_stat_tiles", B, *size) * 2 * Cout, device="cuda")
    with pytest.raises(Exception, match="statistics rows"):
        ops.lib.call("pulpo_conv3d_k3_fwd_wino3", ctypes.c_void_p(xc.data_ptr()), *ops.grid_strides(xc), ctypes.c_void_p(wp.data_ptr()), None, None, 0.2,
                     ctypes.c_void_p(y.data_ptr()), *ops.grid_strides(y), ctypes.c_void_p(stats.data_ptr()), B, *size, Cin, Cout, ops._stream())


@pytest.mark.parametrize("B,Cin,Cout,size,bf16", [(1, 2, 32, (64, 64, 64), False), (2, 2, 32, (9, 11, 13), False), (1, 1, 36, (13, 8, 9), False), (1, 2, 64, (20, 24, 24), True)])
def test_input_layer_weight_gradient_with_the_batchnorm_backward_fused(ops, B, Cin, Cout, size, bf16):
    """the image-pair ConvUnit (nobody needs its data gradient): the weight-gradient kernel forms dy = BatchNorm / LeakyReLU backward of dz per element
    while staging (pulpo_conv3d_k3_wgrad_bn) instead of a pass that writes dy - same parameter gradients as the separate passes (fp32 rounding of
    a different summation order only) and as the oracle in double; also with bf16-stored dz (ACT_BF16: y of this layer stays fp32)"""
    import src.network_blocks as nb
    gen = torch.Generator().manual_seed(17 * Cin + Cout)
    torch.manual_seed(3)
    unit = nb.ConvUnit(list(size), Cin, Cout)
    with torch.no_grad():
        unit._op[1].weight.copy_(torch.rand(Cout, generator=gen) * 0.5 + 0.75)
        unit._op[1].bias.copy_(torch.randn(Cout, generator=gen) * 0.1)
    sd = {"u." + k: v.detach().clone() for k, v in unit.state_dict().items()}
    x = torch.randn(B, Cin, *size, generator=gen)
    up = torch.randn(B, Cout, *size, generator=gen)
    unit = unit.cuda().train()
    ps = [unit._op[0].weight, unit._op[0].bias, unit._op[1].weight, unit._op[1].bias]
    names = ["u._op.0.weight", "u._op.0.bias", "u._op.1.weight", "u._op.1.bias"]
    state = {k: v.clone() for k, v in unit.state_dict().items()}

    def run(fused):
        unit.load_state_dict(state)
        ops.FUSE_INPUT_WGRAD = fused
        if bf16:
            ops.set_conv_precision("bf16", activations="bf16")
        try:
            z = unit(x.cuda())
            return z, torch.autograd.grad((z.float() * up.cuda()).sum(), ps)
        finally:
            ops.FUSE_INPUT_WGRAD = True
            ops.set_conv_precision("fp32")

    z1, g1 = run(True)
    z0, g0 = run(False)
    assert torch.equal(z1, z0)
    for a, b, nme in zip(g1, g0, names):
        scale = float(g0[0].abs().max()) if nme.endswith("0.bias") else None       # (true gradient zero: compared on the weight gradient's scale)
        if scale is not None:
            assert float((a - b).abs().max()) <= 1e-4 * scale, nme
        else:
            assert rel_l2(a, b) < (2e-3 if bf16 else 2e-5), (nme, rel_l2(a, b))
    if not bf16:
        sdr = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
               for k, v in sd.items()}
        zr = O.conv_unit(x.double(), sdr, "u", training=True)
        gr = torch.autograd.grad((zr * up.double()).sum(), [sdr[n] for n in names])
        sd32 = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
        g32 = torch.autograd.grad((O.conv_unit(x, sd32, "u", training=True) * up).sum(), [sd32[n] for n in names])
        for got, want, o32, nme in zip(g1, gr, g32, names):
            if nme.endswith("0.bias"):
                continue
            assert rel_l2(got, want) <= 3.0 * rel_l2(o32, want) + 1e-5, (nme, rel_l2(got, want))


@pytest.mark.parametrize("B,Cin,Cout,size", [(1, 32, 32, (64, 64, 64)), (2, 16, 96, (10, 20, 28)), (1, 64, 32, (6, 12, 17)), (1, 8, 8, (4, 16, 16)), (1, 40, 24, (12, 10, 9)),
                                             (1, 32, 64, (9, 16, 16)), (3, 96, 32, (2, 24, 24))])
def test_weight_gradient_winograd_in_all_three_axes_vs_oracle(ops, B, Cin, Cout, size):
    """the weight gradient of channels-last operands: F(2x2x2,3x3x3) (conv3d_k3_wgrad_w3x: plane pairs, even depths - ragged rows and columns, depths
    that are not multiples of four, several batch elements and column segments per workgroup) and, for the odd and the two-plane depths, the (y, x)
    kernel it falls back to; against the double-precision gradient"""
    gen = torch.Generator().manual_seed(B * 100 + Cin + Cout)
    x = torch.randn(B, Cin, *size, generator=gen)
    dy = torch.randn(B, Cout, *size, generator=gen)
    w = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad((F.conv3d(x.double(), w, padding=1) * dy.double()).sum(), [w])
    xd = x.cuda().contiguous(memory_format=torch.channels_last_3d)
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last_3d)
    D, H, W = size
    algo = ops.lib.query("pulpo_conv3d_k3_wgrad_algo", B, D, H, W, Cin, Cout, 1)
    assert D * H * W >= 1000 and algo == (3 if D % 2 == 0 and D >= 4 else 2), algo
    got = ops._wgrad_raw(xd, dyd, Cin, Cout)
    assert rel_l2(got, ref) < 3e-6, rel_l2(got, ref)


@pytest.mark.parametrize("B,Cin,Cout,size", [(1, 2, 32, (9, 11, 13)), (2, 3, 32, (20, 17, 33)), (1, 4, 48, (8, 24, 16)), (1, 1, 36, (13, 8, 9))])
def test_weight_gradient_of_the_narrow_input_layers(ops, B, Cin, Cout, size):
    """<= 4 input channels with a channels-last output gradient (what a ConvUnit's backward hands over) take their own kernel
    (conv3d_k3_wgrad_smallc: waves split the voxels, every wave all row tiles); ragged volumes, two cout tiles, planar x"""
    gen = torch.Generator().manual_seed(B * 100 + Cin * 10 + Cout)
    x = torch.randn(B, Cin, *size, generator=gen)
    dy = torch.randn(B, Cout, *size, generator=gen)
    w = torch.zeros(Cout, Cin, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    ref, = torch.autograd.grad((F.conv3d(x.double(), w, padding=1) * dy.double()).sum(), [w])
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last_3d)
    got = ops._wgrad_raw(x.cuda(), dyd, Cin, Cout)
    assert rel_l2(got, ref) < 1e-5


@pytest.mark.parametrize("B,Cin,Cout,size,cl", [c for c in CONV_CASES if c[1] > 4] + [(1, 32, 64, (16, 64, 64), True), (1, 96, 96, (64, 64, 64), True)])
def test_conv3d_bf16_operands_vs_oracle(ops, B, Cin, Cout, size, cl):
    """bf16-operand mode (BASELINE configs 4-5): the kernel against the oracle's definition (operands rounded to bf16, exact
    products, wide accumulation) - bf16 x bf16 products are exact in fp32, so the fp32 tolerances of the fp32 test apply.
    Also bounds the distance to the fp32 convolution (stated: relative L2 <= 1e-2 for N(0,1) data)."""
    gen = torch.Generator().manual_seed(B * 1000 + Cin * 10 + Cout)
    x = torch.randn(B, Cin, *size, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=gen) / (27 * Cin) ** 0.5
    b = torch.randn(Cout, generator=gen)
    up = torch.randn(B, Cout, *size, generator=gen)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    O.CONV_PRECISION = "bf16"
    ops.set_conv_precision("bf16")
    try:
        ref = O.conv3_k3(xr, wr, br)
        gref = torch.autograd.grad((ref * up.double()).sum(), [xr, wr, br])
        xd = x.cuda().contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        out = ops.conv3d_k3(xd, wd, bd)
        gx, gw, gb = torch.autograd.grad((out * up.cuda()).sum(), [xd, wd, bd])
    finally:
        O.CONV_PRECISION = "fp32"
        ops.set_conv_precision("fp32")
    assert rel_l2(out, ref) < 2e-6
    assert rel_l2(gx, gref[0]) < 2e-6
    assert rel_l2(gw, gref[1]) < 1e-5
    assert rel_l2(gb, gref[2]) < 1e-5
    full = F.conv3d(x.double(), w.double(), b.double(), padding=1)
    assert rel_l2(out, full) < 1e-2


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("Cin,Cout,size", [(32, 64, (8, 16, 16)), (192, 192, (5, 6, 4)), (3, 32, (9, 8, 8)), (16, 12, (7, 6, 5)), (2, 32, (64, 64, 64))])
def test_eval_mode_fused_epilogue_equals_separate_passes(ops, precision, Cin, Cout, size):
    """inference runs conv + folded BatchNorm + LeakyReLU as one kernel (also through the split-K reduce of small volumes);
    with a gradient requested the pre-norm tensor is kept and the normalisation is a separate pass: same bits either way,
    and the eval-mode gradient matches the oracle."""
    gen = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(2, Cin, *size, generator=gen)
    sd = {"u._op.0.weight": torch.randn(Cout, Cin, 3, 3, 3, generator=gen) / (27 * Cin) ** 0.5, "u._op.0.bias": torch.randn(Cout, generator=gen),
          "u._op.1.weight": torch.rand(Cout, generator=gen) + 0.5, "u._op.1.bias": torch.randn(Cout, generator=gen),
          "u._op.1.running_mean": torch.randn(Cout, generator=gen) * 0.1, "u._op.1.running_var": torch.rand(Cout, generator=gen) + 0.5}
    d = {k: v.cuda() for k, v in sd.items()}
    args = (d["u._op.0.weight"], d["u._op.0.bias"], d["u._op.1.weight"], d["u._op.1.bias"], d["u._op.1.running_mean"], d["u._op.1.running_var"])
    ops.set_conv_precision(precision)
    O.CONV_PRECISION = precision
    try:
        xg = x.cuda().requires_grad_(True)
        sep = ops.conv_bn_lrelu(xg, *args, training=False)
        fused = ops.conv_bn_lrelu(x.cuda(), *args, training=False)
        assert torch.equal(sep.detach(), fused)
        xr = x.double().requires_grad_(True)
        ref = O.conv_unit(xr, {k: v.double() for k, v in sd.items()}, "u", training=False)
        assert rel_l2(fused, ref) < 5e-6
        up = torch.randn(ref.shape, generator=gen)
        gx, = torch.autograd.grad((sep * up.cuda()).sum(), [xg])
        gr, = torch.autograd.grad((ref * up.double()).sum(), [xr])
        # bf16 mode: the gradient operand dy is itself rounded to bf16 - fp32-vs-fp64 differences in dy flip a few roundings
        assert rel_l2(gx, gr) < (5e-6 if precision == "fp32" else 2e-4)
    finally:
        ops.set_conv_precision("fp32")
        O.CONV_PRECISION = "fp32"


@pytest.mark.parametrize("B,Cin,Cout,size", [(1, 32, 32, (64, 64, 64)), (1, 64, 64, (64, 64, 64)), (2, 16, 96, (64, 56, 80)), (1, 20, 12, (64, 64, 70)),
                                            (1, 96, 32, (68, 61, 67)), (2, 192, 192, (20, 20, 20)), (1, 8, 40, (24, 20, 17)),
                                            (1, 288, 192, (20, 20, 20)), (1, 128, 192, (20, 20, 20)), (1, 192, 192, (10, 10, 10)),
                                            (2, 96, 64, (10, 12, 9)), (1, 96, 96, (40, 40, 40)), (1, 160, 64, (32, 40, 48)), (2, 64, 128, (24, 32, 32)),
                                            (1, 72, 32, (44, 64, 40)), (1, 24, 64, (32, 32, 32))])
def test_conv3d_winograd_kernel_vs_oracle(ops, B, Cin, Cout, size):
    """volumes of >= 20^3 voxels with depth % 4 == 0 take the F(2x2,3x3) Winograd kernels for forward and data gradient
    (pulpo_conv3d_k3_algo = 2: the pipelined conv3d_k3_wino2p_mfma for operands with a multiple of 8 channels, the round-2
    conv3d_k3_wino2_mfma otherwise - Cin = 20 forward, Cout = 12 data gradient here) and for the weight gradient: same fp32 tolerance
    against the fp64 convolution as the direct kernels; ragged H / W (odd sizes: half-filled blocks) and the 20^3 level's 128 / 192 / 288
    channel layers (split-K work items where the tiles are few) included, and the 10^3 level, whose depth is not a multiple of 4
    (ragged depth tile: pipelined kernel with split-K only)"""
    from pulpo_amd._lib import lib
    # F(2x2x2,3x3x3) (algo 3, conv3d_k3_wino3_mfma) where the volume is whole 4x8x8 tiles, the GEMM has >= 16 reduction channels (a
    # multiple of 8) and a multiple of 4 (>= 16) output channels - cout tiles of 32, the last one possibly partly empty - and there are >= 256 work items - forward and / or data gradient of the last
    # five cases, of the two 64^3 cases and the data gradient of 16 -> 96; F(2x2,3x3) (algo 2) everywhere else
    whole = size[0] % 4 == 0 and size[1] % 8 == 0 and size[2] % 8 == 0
    items = B * (size[0] // 4) * (size[1] // 8) * (size[2] // 8)
    expect = lambda K, N: 3 if (whole and K >= 16 and K % 8 == 0 and N % 4 == 0 and (N % 32 == 0 or N >= 16) and items * ((N + 31) // 32) >= 256) else 2
    assert lib.query("pulpo_conv3d_k3_algo", B, *size, Cin, Cout) == expect(Cin, Cout)
    assert lib.query("pulpo_conv3d_k3_algo", B, *size, Cout, Cin) == expect(Cout, Cin)
    if (Cin, Cout) in ((64, 64), (96, 96), (160, 64), (64, 128)):
        assert expect(Cin, Cout) == 3 and expect(Cout, Cin) == 3
    if (Cin, Cout) == (72, 32):
        assert expect(Cin, Cout) == 3 and expect(Cout, Cin) == 3          # (data gradient: 72 output channels = three cout tiles, the last a quarter full)
    if (Cin, Cout) == (16, 96):
        assert expect(Cin, Cout) == 3 and expect(Cout, Cin) == 3          # (16 reduction channels = two chunks; 16 output channels = half a cout tile)
    assert lib.query("pulpo_conv3d_k3_wino2_pipelined", *size, Cin, Cin) == int(Cin % 8 == 0)
    if (B, Cin, size) == (1, 288, (20, 20, 20)):                     # few tiles, many reduction channels: three split-K work items per tile
        assert lib.query("pulpo_conv3d_k3_fwd_wino2_scratch_floats", B, *size, Cin, Cout) == 3 * B * 8000 * Cout
    gen = torch.Generator().manual_seed(B * 1000 + Cin * 10 + Cout)
    x = torch.randn(B, Cin, *size, generator=gen)
    w = torch.randn(Cout, Cin, 3, 3, 3, generator=gen) / (27 * Cin) ** 0.5
    b = torch.randn(Cout, generator=gen)
    up = torch.randn(B, Cout, *size, generator=gen)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv3d(xr.double(), wr.double(), br.double(), padding=1)
    gref = torch.autograd.grad((ref * up.double()).sum(), [xr, wr, br])
    xd = x.cuda().contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    out = ops.conv3d_k3(xd, wd, bd)
    gx, gw, gb = torch.autograd.grad((out * up.cuda()).sum(), [xd, wd, bd])
    assert rel_l2(out, ref) < 2e-6
    assert rel_l2(gx, gref[0]) < 2e-6
    assert rel_l2(gw, gref[1]) < 1e-5
    assert rel_l2(gb, gref[2]) < 1e-5


def test_packed_weight_cache_follows_weight_updates(ops):
    """the GEMM-ordered weight packs are cached per weight version: an in-place update through torch (optimizer) or behind its back (the
    fused Adam kernel writes through raw pointers and calls ops.invalidate_weight_packs) must be seen by the next convolution"""
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(1, 16, 32, 32, 32, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(16, 16, 3, 3, 3, generator=gen) / 20).cuda()
    y0 = ops.conv3d_k3(x, w)
    assert torch.equal(ops.conv3d_k3(x, w), y0)                      # cached pack: same result
    w.mul_(2.0)                                                      # torch in-place: version counter moves
    assert rel_l2(ops.conv3d_k3(x, w), 2.0 * y0) < 1e-6
    p = w.clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    y1 = ops.conv3d_k3(x, p)
    ops.adam_step(p.view(-1), torch.ones_like(p).view(-1), m.view(-1), v.view(-1), lr=0.1, step=1)       # raw-pointer update
    y2 = ops.conv3d_k3(x, p)
    ref = F.conv3d(x.cpu().double(), p.cpu().double(), padding=1)
    assert rel_l2(y2, ref) < 2e-6 and rel_l2(y1, ref) > 1e-2
    # the same through the bf16-operand packs (rewritten in place by the same launch)
    ops.set_conv_precision("bf16")
    try:
        q = w.clone()
        z1 = ops.conv3d_k3(x, q)
        ops.adam_step(q.view(-1), torch.ones_like(q).view(-1), torch.zeros_like(q).view(-1), torch.zeros_like(q).view(-1), lr=0.1, step=1)
        z2 = ops.conv3d_k3(x, q)
        refq = F.conv3d(x.cpu().double(), q.cpu().double(), padding=1)
        assert rel_l2(z2, refq) < 1e-2 and rel_l2(z1, refq) > 1e-1
    finally:
        ops.set_conv_precision("fp32")


def test_every_live_weight_pack_follows_a_raw_pointer_update(ops):
    """One weight used at TWO volume shapes (a training batch and a validation batch, predict_output_samples with N > 1) has two live packs
    per orientation and precision; all of them must be rewritten by the in-place refresh behind the fused Adam kernel.  (Round 2 keyed
    the bf16 packs of a weight without the shape: the second shape's pack replaced the first's registry entry, and the first shape kept
    convolving with pre-update weights.)  Also: packs of temporaries (non-leaf weights, e.g. the 2-D mode's lifted 3x3 kernels) are not
    cached or registered, so the registry does not grow with repeated forwards."""
    gen = torch.Generator().manual_seed(6)
    xa = torch.randn(1, 16, 24, 24, 24, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    xb = torch.randn(2, 16, 16, 16, 16, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    w0 = (torch.randn(16, 16, 3, 3, 3, generator=gen) / 20).cuda()
    for precision, tol_new, tol_old in (("fp32", 2e-6, 1e-2), ("bf16", 1e-2, 1e-1)):
        ops.set_conv_precision(precision)
        try:
            p = w0.clone()
            ya0, yb0 = ops.conv3d_k3(xa, p), ops.conv3d_k3(xb, p)              # two shapes -> two packs of the same weight
            ops.adam_step(p.view(-1), torch.ones_like(p).view(-1), torch.zeros_like(p).view(-1), torch.zeros_like(p).view(-1), lr=0.1, step=1)
            ya1, yb1 = ops.conv3d_k3(xa, p), ops.conv3d_k3(xb, p)
            for x_, y_old, y_new in ((xa, ya0, ya1), (xb, yb0, yb1)):
                ref = F.conv3d(x_.cpu().double(), p.cpu().double(), padding=1)
                assert rel_l2(y_new, ref) < tol_new, (precision, tuple(x_.shape), rel_l2(y_new, ref))
                assert rel_l2(y_old, ref) > tol_old
        finally:
            ops.set_conv_precision("fp32")
    # temporaries: a derived (non-leaf) weight is packed per call and never enters the registry
    base = torch.nn.Parameter(w0.clone())
    n0 = len(ops._PACK_REGISTRY)
    for _ in range(5):
        derived = base * 1.0
        assert not derived.is_leaf
        ops.conv3d_k3(xa, derived)
    assert len(ops._PACK_REGISTRY) == n0
    x2 = torch.randn(2, 4, 24, 20, generator=gen).cuda()                        # 2-D mode: the 3x3 weight is lifted into a temporary per call
    w2 = torch.nn.Parameter((torch.randn(8, 4, 3, 3, generator=gen) / 6).cuda())
    bn = torch.nn.BatchNorm2d(8).cuda()
    for _ in range(5):
        ops.conv_bn_lrelu(x2, w2, torch.zeros(8, device="cuda"), bn.weight, bn.bias, bn.running_mean, bn.running_var, training=True).sum().backward()
    assert len(ops._PACK_REGISTRY) == n0


@pytest.mark.parametrize("chans,size", [((16, 32, 64, 32), (24, 32, 40)), ((32, 32, 32), (32, 32, 32)), ((8, 96, 96), (20, 24, 24)),
                                        ((64, 64, 96, 64), (32, 32, 32))])
def test_bn_backward_reduction_inside_the_data_gradient_kernel(ops, chans, size):
    """In a ConvSequence the data-gradient convolution of unit u also delivers the BatchNorm-backward sums of unit u-1 (one HBM pass over
    dz and y less per unit).  Checked: the fused path IS taken on every inner link of these shapes, its gradients equal those of the
    separate-pass path to fp32 rounding, both match the float64 oracle, and a unit whose output has a second consumer (autograd sums two
    gradients) falls back to the separate pass with the right result."""
    gen = torch.Generator().manual_seed(sum(chans))
    x = torch.randn(1, chans[0], *size, generator=gen)
    sd = {}
    for u in range(len(chans) - 1):
        ci, co = chans[u], chans[u + 1]
        sd.update({f"u{u}._op.0.weight": torch.randn(co, ci, 3, 3, 3, generator=gen) / (27 * ci) ** 0.5, f"u{u}._op.0.bias": torch.randn(co, generator=gen),
                   f"u{u}._op.1.weight": torch.rand(co, generator=gen) + 0.5, f"u{u}._op.1.bias": torch.randn(co, generator=gen)})
    up = torch.randn(1, chans[-1], *size, generator=gen)
    side = torch.randn(1, chans[1], *size, generator=gen)

    default_switch = ops.BN_REDUCE_IN_DGRAD

    def run(fused, second_consumer):
        ops.BN_REDUCE_IN_DGRAD = fused
        d = {k: v.cuda().requires_grad_(True) for k, v in sd.items()}
        xg = x.cuda().requires_grad_(True)
        h = xg
        taken = []
        orig = ops._take_bn_tile_parts
        def spy(*a):
            r = orig(*a)
            taken.append(r is not None)
            return r
        ops._take_bn_tile_parts = spy
        try:
            extra = 0.0
            for u in range(len(chans) - 1):
                co = chans[u + 1]
                h = ops.conv_bn_lrelu(h, d[f"u{u}._op.0.weight"], d[f"u{u}._op.0.bias"], d[f"u{u}._op.1.weight"], d[f"u{u}._op.1.bias"],
                                      torch.zeros(co, device="cuda"), torch.ones(co, device="cuda"), training=True)
                if second_consumer and u == 0:
                    extra = (h * side.cuda()).sum()
            loss = (h * up.cuda()).sum() + extra
            grads = torch.autograd.grad(loss, [xg] + [d[k] for k in sorted(sd)])
        finally:
            ops._take_bn_tile_parts = orig
            ops.BN_REDUCE_IN_DGRAD = default_switch
        return [g.cpu() for g in grads], taken

    def oracle(second_consumer):
        d = {k: v.double().requires_grad_(True) for k, v in sd.items()}
        xr = x.double().requires_grad_(True)
        h, extra = xr, 0.0
        for u in range(len(chans) - 1):
            d[f"u{u}._op.1.running_mean"] = torch.zeros(chans[u + 1], dtype=torch.float64)
            d[f"u{u}._op.1.running_var"] = torch.ones(chans[u + 1], dtype=torch.float64)
            h = O.conv_unit(h, d, f"u{u}", training=True)
            if second_consumer and u == 0:
                extra = (h * side.double()).sum()
        return torch.autograd.grad((h * up.double()).sum() + extra, [xr] + [d[k] for k in sorted(sd)])

    nlink = len(chans) - 2
    for second in (False, True):
        gf, taken_f = run(True, second)
        gs, taken_s = run(False, second)
        gr = oracle(second)
        assert not any(taken_s)
        # units are walked last to first; unit u < last is offered the sums of its consumer, unit 0 with a second consumer must refuse them
        want = [False] + [True] * nlink
        if second:
            want[-1] = False
        assert taken_f == want, (taken_f, want)
        for a, b, r in zip(gf, gs, gr):
            if r.dim() == 1 and r.numel() in chans[1:] and float(r.abs().max()) < 1e-6 * float(gr[0].abs().max()):
                continue                     # conv bias in front of a training-mode BatchNorm: exact gradient 0, computed value is rounding noise
            assert rel_l2(a, b) < 2e-5
            assert rel_l2(a, r) < 2e-3 and rel_l2(b, r) < 2e-3      # (fp32 vs float64 through three units: LeakyReLU branch flips near 0)


def test_conv_linearity_at_full_channel_width(ops):
    """size-independent property at a BASELINE layer shape (32->32 @ 48^3): conv(a*x1 + x2) = a*conv(x1) + conv(x2) (no bias)"""
    gen = torch.Generator().manual_seed(3)
    x1 = torch.randn(1, 32, 48, 48, 48, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    x2 = torch.randn(1, 32, 48, 48, 48, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(32, 32, 3, 3, 3, generator=gen) / 30).cuda()
    lhs = ops.conv3d_k3(2.5 * x1 + x2, w)
    rhs = 2.5 * ops.conv3d_k3(x1, w) + ops.conv3d_k3(x2, w)
    assert rel_l2(lhs, rhs) < 1e-6


# ================================================================================================ heads
def test_mu_sigma_golden(ops, golden):
    g = golden("musigma")
    sd = {k[3:]: dev(v.copy()).requires_grad_(True) for k, v in g.items() if k.startswith("sd.")}
    x = dev(g["x"]).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    mu, sg, z = ops.mu_sigma_sample(x, sd["_conv_mu.weight"], sd["_conv_mu.bias"], sd["_conv_sigma.0.weight"], sd["_conv_sigma.0.bias"],
                                    dev(g["eps"]))
    close(mu, g["mu"], atol=1e-5); close(sg, g["sigma"], atol=1e-5); close(z, g["z"], atol=1e-5)
    names = [k[2:] for k in g if k.startswith("g.")]
    grads = torch.autograd.grad((z * dev(g["up"])).sum() + (mu * mu).sum() + sg.sum(), [x] + [sd[n] for n in names])
    assert rel_l2(grads[0], g["gx"]) < 1e-5
    for got, n in zip(grads[1:], names):
        assert rel_l2(got.reshape(-1), g["g." + n].reshape(-1)) < 1e-5, n


def test_velocity_field_eval_golden(ops, golden):
    g = golden("musigma")
    sd = {k[6:]: dev(v.copy()) for k, v in g.items() if k.startswith("vf_sd.")}
    h = dev(g["vf_z"])
    for i in (0, 1):
        p = f"_op.{i}._op."
        h = ops.conv_bn_lrelu(h, sd[p + "0.weight"], sd[p + "0.bias"], sd[p + "1.weight"], sd[p + "1.bias"], sd[p + "1.running_mean"],
                              sd[p + "1.running_var"], training=False)
    out = ops.conv1x1_to3(h, sd["_op.2.weight"], sd["_op.2.bias"])
    close(out, g["vf_out"], atol=2e-5)


# ================================================================================================ losses
@pytest.mark.parametrize("w", [3, 5, 7, 9, 11])
@pytest.mark.parametrize("kind", ["rand", "smooth"])
def test_ncc_golden(ops, golden, w, kind):
    g = golden("losses")
    pred = dev(g[f"ncc{w}_{kind}_pred"]).requires_grad_(True)
    true = dev(g[f"ncc{w}_{kind}_true"])
    loss = ops.ncc_loss(pred, true, w, 0.05)
    close(loss, g[f"ncc{w}_{kind}_loss"], rtol=1e-4, atol=1e-6)
    gp, = torch.autograd.grad(loss * 1.7, [pred])
    ref = g[f"ncc{w}_{kind}_gpred"] * 1.7
    # where both window variances vanish (zero background) the fp32 reference gradient is itself only ~1e-3 accurate
    assert np.abs(gp.cpu().numpy() - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7


def test_ncc_wide_row_vs_oracle(ops):
    """rows longer than one wavefront segment (W = 150 > 64 - 2*pad) and B = 2"""
    gen = torch.Generator().manual_seed(11)
    t, p = torch.rand(2, 1, 6, 7, 150, generator=gen), torch.rand(2, 1, 6, 7, 150, generator=gen)
    close(ops.ncc_loss(p.cuda(), t.cuda(), 9, 0.05), O.ncc(p, t, 9, 0.05), rtol=1e-4)


def test_kl_l2reg_golden(ops, golden):
    g = golden("losses")
    mu, sg = dev(g["kl_mu"]).requires_grad_(True), dev(g["kl_sigma"]).requires_grad_(True)
    kl = ops.kl_std_normal(mu, sg)
    close(kl, g["kl_loss"], rtol=1e-5)
    gm, gs = torch.autograd.grad(kl, [mu, sg])
    close(gm, g["kl_gmu"], atol=1e-6); close(gs, g["kl_gsigma"], atol=1e-5, rtol=1e-5)
    df = dev(g["reg_df"]).requires_grad_(True)
    r = ops.l2_reg(df, 0.025)
    close(r, g["reg_loss"], rtol=1e-5)
    close(torch.autograd.grad(r, [df])[0], g["reg_gdf"], atol=1e-7, rtol=1e-4)


# ================================================================================================ optimizer
def test_adam_matches_torch(ops):
    gen = torch.Generator().manual_seed(2)
    p0 = torch.randn(1003, generator=gen)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = p0.cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        gr = torch.randn(1003, generator=gen)
        ref.grad = gr.clone()
        opt.step()
        ops.adam_step(p, gr.cuda(), m, v, 1e-3, step)
    close(p, ref, atol=1e-6)


def test_cpu_tensor_is_refused(ops):
    from pulpo_amd._lib import PulpoHipError
    with pytest.raises(PulpoHipError):
        ops.warp3d(torch.zeros(1, 3, 4, 4, 4), torch.zeros(1, 1, 4, 4, 4))


# ================================================================================================ alternative losses / metrics
def test_alternative_losses_and_metrics_golden(ops, golden):
    g = golden("metrics")
    a = dev(g["l2_in"]).requires_grad_(True)
    l = ops.l2_loss(a, dev(g["l2_tgt"]))
    close(l, g["l2_loss"], rtol=1e-5)
    close(torch.autograd.grad(l * 0.7, [a])[0], 0.7 * g["l2_gin"], atol=1e-7, rtol=1e-5)
    a = dev(g["dice_in"]).requires_grad_(True)
    for df_ in (1, 4):
        l = ops.soft_dice_loss(a, dev(g["dice_tgt"]), df_)
        close(l, g[f"dice{df_}_loss"], rtol=1e-5)
        close(torch.autograd.grad(l, [a])[0], g[f"dice{df_}_gin"], atol=1e-6, rtol=1e-4)
    d = dev(g["jdet_df"]).requires_grad_(True)
    for norm in (1, 0):
        close(ops.jacobian_det(d, bool(norm)), g[f"jdet_norm{norm}"], atol=1e-5, rtol=1e-5)
        s_ = ops.jdet_std(d, 0.3, bool(norm))
        close(s_, g[f"jstd_norm{norm}"], rtol=1e-4)
        close(torch.autograd.grad(s_, [d])[0], g[f"jstd_gd_norm{norm}"], atol=1e-6, rtol=1e-3)


def test_eval_metrics_golden(ops, golden):
    """rmse / dsc / % |J| <= 0 / landmark warp of the evaluation harness (evaluate.py:315-327, 410-423, 1441-1446) on the device against
    goldens made with the reference's warp_landmarks / jacobian_det and the harness's expressions (make_golden.py evalmetrics)"""
    from pulpo_amd import eval_metrics as M
    import src.components.utils as shim_utils
    g = golden("evalmetrics")
    close(M.rmse(dev(g["rmse_a"]), dev(g["rmse_b"])), g["rmse"], rtol=1e-6)
    close(M.dsc(dev(g["dsc_in"]), dev(g["dsc_tgt"])), g["dsc"], rtol=1e-6)
    df = dev(g["leq_df"])
    close(M.jdet(df), g["leq_jdet"], atol=1e-4, rtol=1e-5)
    # a determinant within rounding of 0 may fall on either side: allow the count to differ by the number of such voxels
    near = float((np.abs(g["leq_jdet"]) < 1e-4).sum()) * 100.0 / g["leq_jdet"].size
    assert abs(float(M.jdet_leq0_percent(df)) - float(g["leq_pct"])) <= near + 1e-4
    close(ops.percent_leq0(dev(g["leq_jdet"])), g["leq_pct"], rtol=1e-6)
    out = shim_utils.warp_landmarks(dev(g["lm"]), dev(g["lm_df"]))
    assert tuple(out.shape) == g["lm_out"].shape
    close(out, g["lm_out"], atol=1e-6)
    out_cpu_lm = M.warp_landmarks(T(g["lm"]), dev(g["lm_df"]))          # landmarks may arrive on the host (dataset tensors)
    close(out_cpu_lm, g["lm_out"], atol=1e-6)
    bad = T(g["lm"]).clone()
    bad[0, 3, 1] = 12.0                                                   # H = 12: out of range -> IndexError like the reference
    with pytest.raises(IndexError):
        M.warp_landmarks(bad, dev(g["lm_df"]))


def test_recon_loss_options_through_the_model_api(ops):
    """--recon_loss mse dice / --regularizer jdet / --nondiagonal construct and step (reference train.py:29-32,157-166 options)"""
    from src.models import PULPo
    FB = list(O.FEEDBACK_DEFAULT)
    torch.manual_seed(0)
    m = PULPo(3, 2, 0.1, [16, 16, 16], feedback=FB, n0=4, recon_loss=["ncc", "mse", "dice"], regularizer="jdet", segs=True,
              nondiagonal=True).cuda().train()
    x, y = torch.rand(1, 1, 16, 16, 16).cuda(), torch.rand(1, 1, 16, 16, 16).cuda()
    seg = (torch.rand(1, 1, 16, 16, 16) > 0.5).float().cuda()
    e = torch.empty((0,), device="cuda")
    loss = m.training_step((x, y, seg, seg, e, e, e, e), 0)
    loss.backward()
    assert bool(torch.isfinite(loss))
    assert all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_kl_nondiagonal_golden(ops, golden):
    g = golden("metrics")
    mu, sg = dev(g["kln_mu"]).requires_grad_(True), dev(g["kln_sigma"]).requires_grad_(True)
    from src.losses import KL_nondiagonal
    kl = KL_nondiagonal(inshape=torch.tensor([5, 6, 7]), prior_lambda=20)
    l = kl.loss(torch.zeros_like(mu), torch.ones_like(sg), mu, sg)
    close(l, g["kln_loss"], rtol=1e-5)
    gm, gs = torch.autograd.grad(l, [mu, sg])
    close(gm, g["kln_gmu"], atol=1e-5, rtol=1e-4)
    close(gs, g["kln_gsigma"], atol=1e-4, rtol=1e-4)


# ================================================================================================ MC uncertainty statistics
@pytest.mark.parametrize("N,C,size", [(8, 3, (6, 7, 9)), (2, 1, (5, 5, 5)), (5, 3, (16, 16, 16))])
def test_streaming_mc_moments_vs_oracle(ops, N, C, size):
    """the streaming (mean, M2) kernel against evaluate.py's stack + torch.std + torch.mean (oracle.mc_std_map)"""
    gen = torch.Generator().manual_seed(N * 10 + C)
    stack = torch.randn(N, C, *size, generator=gen) * 3 + 1.5
    mask = (torch.rand(1, 1, *size, generator=gen) > 0.3).float() * torch.rand(1, 1, *size, generator=gen)
    sm = ops.StreamingMoments()
    for i in range(N):
        sm.update(stack[i:i + 1].cuda())
    close(sm.mean()[0], stack.mean(dim=0), atol=1e-5)
    close(sm.std_map()[0], O.mc_std_map(stack), atol=1e-5, rtol=1e-5)
    close(sm.std_map(scale=mask.cuda())[0], O.mc_std_map(stack, mask), atol=1e-5, rtol=1e-5)


def test_streaming_mc_moments_single_sample_is_nan_like_torch_std(ops):
    sm = ops.StreamingMoments()
    sm.update(torch.ones(1, 3, 4, 4, 4, device="cuda"))
    assert bool(torch.isnan(sm.std_map()).all())          # torch.std over one sample is NaN (unbiased estimator), evaluate.py:243
    with pytest.raises(ValueError):
        sm.update(torch.ones(1, 3, 4, 4, 5, device="cuda"))


# ================================================================================================ round-4 additions
@pytest.mark.parametrize("scale", [None, 0.1])
def test_weighted_sum_is_the_reference_scalar_chain(ops, scale):
    """ops.weighted_sum (one launch) against the reference's per-level scalar chain `all_levels[l] = w * term; total += all_levels[l]`
    (losses.py:262-276, 305-325, 343-355) and `kl * beta` (models.py:161-162), evaluated by torch on the same device scalars: forward values
    EXACTLY equal (explicit round-to-nearest products and sums in the kernel - no fused multiply-add), gradients equal to one rounding."""
    gen = torch.Generator().manual_seed(3)
    for n, weights in ((4, [1.0 / 8 * 4, 8.0, 64.0, 512.0]), (4, [0.3, 1.7, 8.0 / 3, 511.9]), (1, [9.0]), (7, [0.37 * (k + 1) for k in range(7)])):
        vals = (torch.randn(n, generator=gen) * 100).tolist()
        ta = [torch.tensor(v, device="cuda", requires_grad=True) for v in vals]
        tb = [torch.tensor(v, device="cuda", requires_grad=True) for v in vals]
        total, levels = ops.weighted_sum(ta, weights, scale)
        ref_total, ref_levels = 0.0, []
        for w, t in zip(weights, tb):
            ref_levels.append(w * t)
            ref_total = ref_total + ref_levels[-1]
        if scale is not None:
            ref_total = ref_total * scale
            ref_levels = [scale * v for v in ref_levels]
        assert float(total) == float(ref_total), (weights, float(total), float(ref_total))
        for a, b in zip(levels, ref_levels):
            assert float(a) == float(b)
        # gradients: through the total only (the training step), and through the total and the per-level outputs together
        coef = torch.randn(n, generator=gen).tolist()
        for use_levels in (False, True):
            for t in ta + tb:
                t.grad = None
            obj_a = total * 1.5 + (sum(c * v for c, v in zip(coef, levels)) if use_levels else 0.0)
            obj_b = ref_total * 1.5 + (sum(c * v for c, v in zip(coef, ref_levels)) if use_levels else 0.0)
            obj_a.backward(retain_graph=True)
            obj_b.backward(retain_graph=True)
            for a, b in zip(ta, tb):
                np.testing.assert_allclose(float(a.grad), float(b.grad), rtol=2e-7, atol=0)


class _TwiceNet(torch.nn.Module):
    """one ConvUnit and one 1x1x1 head applied TWICE per forward pass - shared weights, the second application optionally on a pooled
    (smaller) volume - plus a unit applied once in between: what the deferred parameter-gradient jobs of the stepper must survive"""

    def __init__(self, nb, pooled_second: bool):
        super().__init__()
        import types
        self.first = nb.ConvUnit([16, 16, 16], 8, 8)
        self.shared = nb.ConvUnit([16, 16, 16], 8, 8)
        self.head = nb.VelocityField([16, 16, 16], 3, 8, 3)          # ConvUnit(3 -> 8), ConvUnit(8 -> 8), 1x1x1 (8 -> 3)
        self.hparams = types.SimpleNamespace(lr=1e-3)
        self.pooled_second = pooled_second
        self.mid_hook = None

    def training_step(self, batch, idx):
        from pulpo_amd import ops as _ops
        x, z = batch
        h = self.shared(self.first(x))
        if self.mid_hook is not None and h.requires_grad:
            h.register_hook(self.mid_hook)              # fires between the two backward passes of `shared`
        h2 = _ops.avg_pool2(h) if self.pooled_second else h
        h3 = self.shared(h2)
        f1 = self.head(z)
        f2 = self.head(f1)                               # the whole VelocityField (two ConvUnits + the head kernel) twice
        return (h3 * h3).mean() + (f2 * f2).mean() + 0.5 * (f1 * f1).mean() + h.mean()


@pytest.mark.parametrize("pooled_second", [False, True])
@pytest.mark.parametrize("mode", ["async", "inline", "flush-between"])
def test_unit_and_head_applied_twice_in_one_stepper_step(ops, pooled_second, mode, monkeypatch):
    """advisor (round 3): `_pending_src` - a ConvUnit / 1x1x1 head applied twice inside one DataParallelStepper step accumulates both weight
    gradients into ONE persistent scratch with ONE finishing job, the second bias / head backward takes the immediate path while the first
    is still deferred, and a mid-backward flush (the bucket hook of a multi-rank run) clears the pending list in between.  All three must
    give plain autograd's gradients; also with the weight gradients in line (async_wgrad off)."""
    from pulpo_amd import dp
    import src.network_blocks as nb
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(1, 8, 16, 16, 16, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    z = torch.randn(1, 3, 16, 16, 16, generator=gen).cuda()

    def make():
        torch.manual_seed(4)
        return _TwiceNet(nb, pooled_second).cuda().train()

    ref = make()
    ref.training_step((x, z), 0).backward()
    want = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
    net = make()
    monkeypatch.setenv("PULPO_WGRAD_SIDE_STREAM", "1")   # (fp32 steps keep the weight gradients in line by default: this test is about the side stream)
    stepper = dp.DataParallelStepper(net, lr=0.0, async_wgrad=(mode != "inline"))
    assert stepper.async_wgrad == (mode != "inline") and stepper.wgrad_on_side_stream() == (mode != "inline")
    if mode == "flush-between":
        net.mid_hook = lambda g: (ops.join_async_wgrad(), None)[1]
    stepper.opt.step = lambda scale: None                # keep the weights: the gradients are what is compared
    for rep in range(2):                                 # twice: the persistent scratch buffers must come back zeroed
        stepper.step((x, z))
        torch.cuda.synchronize()
        assert not ops._PENDING_GRAD_JOBS and not ops._BN_TILE_PARTS
        for k, p in net.named_parameters():
            if k.endswith("_op.0.bias") and not k.endswith("head._op.2.bias"):
                wmax = float(want[k[:-4] + "weight"].abs().max())       # bias in front of a BatchNorm: rounding noise on both sides
                assert float((p.grad - want[k]).abs().max()) <= 1e-4 * max(wmax, 1e-6), (rep, k)
                continue
            assert rel_l2(p.grad, want[k]) < 2e-5, (rep, k, rel_l2(p.grad, want[k]))


def _load_block(block, g, tag):
    sd = {k[len(tag) + 4:]: T(v.copy()) for k, v in g.items() if k.startswith(tag + ".sd.")}
    block.load_state_dict(sd, strict=True)
    return block.cuda()


def test_velocity_field_depth_0_and_1_golden(ops, golden):
    """VelocityField depth 0 (identity) and depth 1 (a bare, UNPADDED 3x3x3 convolution) - network_blocks.py:70-79 - against the
    reference's own class (blocks_r4.npz), values and gradients, on volumes and on slices"""
    import src.network_blocks as nb
    g = golden("blocks_r4")
    vf0 = nb.VelocityField([9, 8, 10], 3, 8, 0)
    x = dev(g["vf0.x"]).requires_grad_(True)
    y = vf0(x)
    assert torch.equal(y.detach().cpu(), T(g["vf0.y"]))
    gx, = torch.autograd.grad((y * dev(g["vf0.up"])).sum(), [x])
    close(gx, g["vf0.gx"], atol=0)
    for tag, size, zdim in (("vf1", [9, 8, 10], 3), ("vf1_2d", [9, 8], 2)):
        vf = _load_block(nb.VelocityField(size, zdim, 8, 1), g, tag).train()
        x = dev(g[tag + ".x"]).requires_grad_(True)
        y = vf(x)
        assert tuple(y.shape) == g[tag + ".y"].shape               # two voxels smaller along every axis
        close(y, g[tag + ".y"], atol=3e-6)
        grads = torch.autograd.grad((y * dev(g[tag + ".up"])).sum(), [x] + list(vf.parameters()))
        close(grads[0], g[tag + ".gx"], atol=1e-5)
        for (n, _), gr in zip(vf.named_parameters(), grads[1:]):
            close(gr, g[f"{tag}.g.{n}"], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("tag,vel,nd", [("rs_half_even", 2, 3), ("rs_half_odd", 2, 3), ("rs_x1p5", 1 / 1.5, 3), ("rs_x0p625", 1.6, 3), ("rs_half_2d", 2, 2)])
def test_resize_transform_scale_factor_semantics_golden(ops, golden, tag, vel, nd):
    """ResizeTransform (network_blocks.py:124-150) with factor < 1 (resize, then scale) and with sizes where floor(in * factor) is not
    in * factor: F.interpolate(scale_factor=...) maps coordinates with 1 / factor there, not with in / out - against the reference's class"""
    import src.network_blocks as nb
    g = golden("blocks_r4")
    rt = nb.ResizeTransform(vel, nd)
    x = dev(g[tag + ".x"]).requires_grad_(True)
    y = rt(x)
    assert tuple(y.shape) == g[tag + ".y"].shape
    close(y, g[tag + ".y"], atol=2e-6)
    gx, = torch.autograd.grad((y * dev(g[tag + ".up"])).sum(), [x])
    close(gx, g[tag + ".gx"], atol=1e-5)


@pytest.mark.parametrize("zdim", [5, 1])
def test_mu_sigma_block_with_zdim_other_than_ndims_golden(ops, golden, zdim):
    """MuSigmaBlock(zdim != ndims) (network_blocks.py:49-60; the model itself always builds zdim = ndims, models.py:88, but the class is part
    of the operator surface): values and every gradient against the reference's class; and a depth-3 VelocityField fed by a 5-channel latent"""
    import src.network_blocks as nb
    g = golden("blocks_r4")
    t = f"ms{zdim}"
    ms = _load_block(nb.MuSigmaBlock([4, 5, 6], 8, zdim), g, t)
    x = dev(g[t + ".x"]).requires_grad_(True)
    mu, sigma, z = ms.sample(x, dev(g[t + ".eps"]))
    close(mu, g[t + ".mu"], atol=3e-6); close(sigma, g[t + ".sigma"], atol=3e-6); close(z, g[t + ".z"], atol=5e-6)
    mu2, sigma2 = ms(x)
    close(mu2, g[t + ".mu"], atol=3e-6); close(sigma2, g[t + ".sigma"], atol=3e-6)
    grads = torch.autograd.grad((z * dev(g[t + ".up"])).sum() + (mu * mu).sum() + sigma.sum(), [x] + list(ms.parameters()))
    close(grads[0], g[t + ".gx"], atol=2e-5)
    for (n, _), gr in zip(ms.named_parameters(), grads[1:]):
        close(gr, g[f"{t}.g.{n}"], atol=2e-4, rtol=1e-5)
    if zdim == 5:
        vf = _load_block(nb.VelocityField([4, 5, 6], 5, 8, 3), g, "vf_z5").eval()
        with torch.no_grad():
            close(vf(dev(g["vf_z5.z"])), g["vf_z5.y"], atol=3e-6)


# ================================================================================================ bf16 activation storage (configs 4-5)
def _bf(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("Cin,Cout,size,pool", [(32, 32, (16, 16, 16), False), (16, 96, (8, 16, 16), True), (3, 32, (12, 8, 16), False), (40, 24, (9, 7, 10), True),
                                                # from 64^3 up with whole 32-channel chunks: the persistent kernel (one chunk; two / three chunks and three cout tiles)
                                                (32, 32, (64, 72, 64), False), (64, 96, (64, 64, 64), True)])
def test_bf16_storage_conv_unit_vs_definition(ops, Cin, Cout, size, pool):
    """One ConvUnit with bf16 ACTIVATION STORAGE (ops.ACT_BF16, BASELINE configs 4-5) against the oracle's definition (O.ACT_PRECISION:
    conv operands bf16, y / z and their gradients rounded to bf16 where they are stored, fp32 arithmetic, statistics of y as stored):
    forward z, the pooled tensor, running statistics, and the input / parameter gradients.  A stored value sits on a different side of a
    bf16 rounding boundary in the two evaluations with probability ~ fp32 error / bf16 spacing ~ 3e-4, so the tensors agree except for a
    small fraction of one-ulp differences: relative L2 <= 1e-3 forward, 5e-3 for gradients (which carry the flips of y, dy and dz)."""
    import src.network_blocks as nb
    gen = torch.Generator().manual_seed(17)
    torch.manual_seed(3)
    unit = nb.ConvUnit(list(size), Cin, Cout)
    with torch.no_grad():
        unit._op[1].weight.copy_(torch.rand(Cout, generator=gen) * 0.5 + 0.75)
        unit._op[1].bias.copy_(torch.randn(Cout, generator=gen) * 0.1)
    sd = {"u." + k: v.detach().clone() for k, v in unit.state_dict().items()}
    x = _bf(torch.randn(2, Cin, *size, generator=gen))
    up = torch.randn(2, Cout, *size, generator=gen)
    unit = unit.cuda().train()
    O.CONV_PRECISION, O.ACT_PRECISION = "bf16", "bf16"
    ops.set_conv_precision("bf16", activations="bf16")
    try:
        xr = x.clone().requires_grad_(True)
        sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
        zr = O.conv_unit(xr, sdr, "u", training=True)
        names = ["u._op.0.weight", "u._op.1.weight", "u._op.1.bias"]
        gr = torch.autograd.grad((zr * up).sum(), [xr] + [sdr[n] for n in names])
        xg = (x.cuda().contiguous(memory_format=torch.channels_last_3d) if Cin > 3 else x.cuda()).requires_grad_(True)
        z = unit(xg, pool_after=pool)
        assert z.dtype == torch.bfloat16
        pooled = getattr(z, "_pulpo_pooled", None)
        g = torch.autograd.grad((z.float() * up.cuda()).sum(), [xg, unit._op[0].weight, unit._op[1].weight, unit._op[1].bias])
    finally:
        O.CONV_PRECISION, O.ACT_PRECISION = "fp32", "fp32"
        ops.set_conv_precision("fp32")
    assert rel_l2(z.float(), zr) <= 1e-3, rel_l2(z.float(), zr)
    diff = (z.float().cpu() - zr).abs()
    # at most ~one bf16 ulp of the pre-norm value (through the BatchNorm scale), on few elements
    assert float(diff.max()) <= 2.0 ** -6 * float(zr.abs().max()) and float((diff > 0).float().mean()) < 0.02, (float(diff.max()), float((diff > 0).float().mean()))
    if pool and pooled is not None:
        pr = _bf(F.avg_pool3d(zr.detach(), 2, 2, ceil_mode=True))
        assert pooled[0].dtype == torch.bfloat16 and rel_l2(pooled[0].float(), pr) <= 1.5e-3
    close(unit._op[1].running_mean, sdr["u._op.1.running_mean"], atol=2e-5, rtol=1e-4)
    close(unit._op[1].running_var, sdr["u._op.1.running_var"], atol=2e-5, rtol=2e-4)
    for got, want, nme in zip(g, gr, ["x"] + names):
        assert rel_l2(got.float(), want) <= 5e-3, (nme, rel_l2(got.float(), want))


def test_bf16_storage_elementwise_kernels_round_the_fp32_result(ops):
    """the typed (bf16-storage) forms of the pooling, head, feedback and eval-mode kernels: on bf16-representable inputs each equals the fp32
    operator's result rounded to bf16, up to one ulp on the elements whose fp32 value sits at a rounding boundary"""
    import src.network_blocks as nb
    gen = torch.Generator().manual_seed(23)

    def ulp_close(a, b, frac=0.01):
        a, b = a.float().cpu(), b.float().cpu()
        d = (a - b).abs() / b.abs().clamp_min(1e-2)
        assert float(d.max()) <= 2.0 ** -6, float(d.max())
        assert float((d > 0).float().mean()) <= frac, float((d > 0).float().mean())

    # ---- average pooling (ceil mode) forward / backward, odd sizes
    x = _bf(torch.randn(2, 16, 9, 10, 7, generator=gen)).cuda().contiguous(memory_format=torch.channels_last_3d)
    xb = x.bfloat16().requires_grad_(True)
    p = ops.avg_pool2(xb)
    assert p.dtype == torch.bfloat16
    ulp_close(p, _bf(F.avg_pool3d(x, 2, 2, ceil_mode=True)))
    up = _bf(torch.randn(p.shape, generator=gen)).cuda()
    gx, = torch.autograd.grad(p, xb, up.bfloat16())
    xr = x.clone().requires_grad_(True)
    gr, = torch.autograd.grad(F.avg_pool3d(xr, 2, 2, ceil_mode=True), xr, up)
    ulp_close(gx, _bf(gr))
    # ---- 1x1x1 heads on a bf16 feature map: planar fp32 outputs, bf16 input gradient
    ms = nb.MuSigmaBlock([6, 8, 8], 16, 3).cuda()
    h = _bf(torch.randn(2, 16, 6, 8, 8, generator=gen)).cuda().contiguous(memory_format=torch.channels_last_3d)
    eps = torch.randn(2, 3, 6, 8, 8, generator=gen).cuda()
    hb = h.bfloat16().requires_grad_(True)
    hf = h.clone().requires_grad_(True)
    mu_b, sg_b, z_b = ms.sample(hb, eps)
    mu_f, sg_f, z_f = ms.sample(hf, eps)
    assert mu_b.dtype == torch.float32
    close(mu_b, mu_f, atol=1e-6); close(sg_b, sg_f, atol=1e-6); close(z_b, z_f, atol=2e-6)
    gb, = torch.autograd.grad((z_b * eps).sum() + sg_b.sum(), hb)
    gf, = torch.autograd.grad((z_f * eps).sum() + sg_f.sum(), hf)
    assert gb.dtype == torch.bfloat16
    ulp_close(gb, _bf(gf))
    # ---- feedback gather (x2 up-sampling + concatenation) written as bf16
    srcs = [torch.randn(1, c, 4, 5, 6, generator=gen).cuda().requires_grad_(True) for c in (3, 3, 3, 3, 3, 1)]
    ref = ops.feedback_up2(srcs)
    ops.set_conv_precision("bf16", activations="bf16")
    try:
        fb = ops.feedback_up2(srcs)
        assert fb.dtype == torch.bfloat16
        ulp_close(fb, _bf(ref.detach()), frac=1.0)               # (every element is a fresh rounding here: one ulp at most, anywhere)
        upf = _bf(torch.randn(fb.shape, generator=gen)).cuda().contiguous(memory_format=torch.channels_last_3d)
        g_b = torch.autograd.grad(fb, srcs, upf.bfloat16())
    finally:
        ops.set_conv_precision("fp32")
    g_f = torch.autograd.grad(ref, srcs, upf)
    for a, b in zip(g_b, g_f):
        close(a, b, atol=1e-5, rtol=1e-5)                        # (planar fp32 gradients of bf16-representable upstream values)


@pytest.mark.parametrize("Cin,Cout,size", [(32, 64, (8, 16, 16)), (3, 32, (9, 8, 8)), (32, 32, (64, 64, 72)), (64, 64, (64, 64, 64))])
def test_bf16_storage_eval_mode_unit(ops, Cin, Cout, size):
    """eval-mode ConvUnit with bf16 activation storage: one fused kernel when operand and result share the storage type, convolution +
    typed apply pass behind the exact-fp32 kernel of the narrow input layers - both equal the definition (oracle) to a bf16 rounding"""
    import src.network_blocks as nb
    gen = torch.Generator().manual_seed(29)
    torch.manual_seed(5)
    unit = nb.ConvUnit(list(size), Cin, Cout)
    with torch.no_grad():
        unit._op[1].running_mean.copy_(torch.randn(Cout, generator=gen) * 0.1)
        unit._op[1].running_var.copy_(torch.rand(Cout, generator=gen) * 0.5 + 0.75)
    sd = {"u." + k: v.detach().clone() for k, v in unit.state_dict().items()}
    x = _bf(torch.randn(1, Cin, *size, generator=gen))
    unit = unit.cuda().eval()
    O.CONV_PRECISION, O.ACT_PRECISION = "bf16", "bf16"
    ops.set_conv_precision("bf16", activations="bf16")
    try:
        with torch.no_grad():
            zr = O.conv_unit(x, sd, "u", training=False)
            z = unit(x.cuda().contiguous(memory_format=torch.channels_last_3d) if Cin > 3 else x.cuda())
    finally:
        O.CONV_PRECISION, O.ACT_PRECISION = "fp32", "fp32"
        ops.set_conv_precision("fp32")
    assert z.dtype == torch.bfloat16
    # (the fused kernel never writes y, but rounds it as the unfused path would have stored it)
    assert rel_l2(z.float(), zr) <= 1e-3, rel_l2(z.float(), zr)


# ================================================================================================ deterministic mode (PULPO_DETERMINISTIC)
@pytest.fixture
def deterministic(ops):
    import os
    ops.set_deterministic(True)
    yield
    ops.set_deterministic(os.environ.get("PULPO_DETERMINISTIC", "0") == "1")


@pytest.mark.parametrize("B,Cin,Cout,size,precision", [(1, 32, 32, (64, 64, 64), "fp32"), (1, 96, 64, (32, 32, 32), "fp32"), (2, 16, 96, (10, 20, 28), "fp32"),
                                                       (1, 64, 32, (7, 12, 17), "fp32"), (1, 2, 32, (64, 64, 64), "fp32"), (2, 3, 32, (20, 17, 33), "fp32"),
                                                       (1, 40, 24, (12, 10, 9), "fp32"), (1, 192, 192, (20, 20, 20), "fp32"), (1, 32, 64, (32, 32, 32), "bf16")])
def test_deterministic_weight_gradient_is_bit_reproducible(ops, deterministic, B, Cin, Cout, size, precision):
    """ordered per-split slabs instead of float atomics between the workgroups of a (ci tile, co tile): every weight-gradient kernel (F(2x2x2) and
    F(2x2) Winograd, direct, narrow-input, bf16 operands) gives the same bits on every run, and the same value as the atomic form up to the
    order of the sums (1e-6) and as the double-precision gradient (the kernels' own test bound)"""
    gen = torch.Generator().manual_seed(Cin + 3 * Cout)
    x = torch.randn(B, Cin, *size, generator=gen).cuda()
    dy = torch.randn(B, Cout, *size, generator=gen).cuda()
    if Cin > 4:
        x = x.contiguous(memory_format=torch.channels_last_3d)
    dy = dy.contiguous(memory_format=torch.channels_last_3d)
    ops.set_conv_precision(precision)
    try:
        a = ops._wgrad_raw(x, dy, Cin, Cout)
        for _ in range(3):
            assert torch.equal(ops._wgrad_raw(x, dy, Cin, Cout), a)
        acc = torch.full_like(a, 0.5)
        ops._wgrad_raw(x, dy, Cin, Cout, into=acc)            # (accumulating form: dw += ...)
        assert rel_l2(acc - 0.5, a) < 1e-5
        ops.set_deterministic(False)
        plain = ops._wgrad_raw(x, dy, Cin, Cout)
    finally:
        ops.set_conv_precision("fp32")
    assert rel_l2(a, plain) < 2e-6
    if precision == "fp32":
        ref = torch.nn.grad.conv3d_weight(x.double().cpu().contiguous(), (Cout, Cin, 3, 3, 3), dy.double().cpu().contiguous(), padding=1)
        assert rel_l2(a, ref) < 1e-5


@pytest.mark.parametrize("size,amp", [((24, 20, 28), 1.0), ((16, 32, 18), 12.0), ((10, 10, 10), 2.0), ((40, 40, 40), 3.0)])
def test_deterministic_vecint_backward_fixed_point_vs_oracle(ops, deterministic, size, amp):
    """VecInt backward with the scatter accumulated in 64-bit fixed point (2^46 units per the largest upstream gradient of a squaring step):
    bit-identical on every run, and against autograd through the oracle's VecInt in double at the plain kernel's bound"""
    gen = torch.Generator().manual_seed(int(amp * 10) + size[0])
    v = torch.randn(2, 3, *size, generator=gen) * amp
    up = torch.randn(2, 3, *size, generator=gen)
    vg = v.cuda().requires_grad_(True)

    def grad():
        out = ops.vecint(vg, 7)
        return torch.autograd.grad((out * up.cuda()).sum(), [vg])[0]

    g0 = grad()
    for _ in range(3):
        assert torch.equal(grad(), g0)
    vr = v.double().requires_grad_(True)
    gr, = torch.autograd.grad((O.vecint(vr, 7) * up.double()).sum(), [vr])
    v32 = v.clone().requires_grad_(True)
    g32, = torch.autograd.grad((O.vecint(v32, 7) * up).sum(), [v32])
    assert rel_l2(g0, gr) < max(1e-4, 3.0 * rel_l2(g32, gr))
    ops.set_deterministic(False)
    assert rel_l2(grad(), g0) < max(1e-5, 0.5 * rel_l2(g32, gr))       # (floor() of a sample coordinate at a cell boundary: both forms compute it alike)


def test_deterministic_warp_and_resize_backward(ops, deterministic):
    """SpatialTransformer backward with an image that requires a gradient (fixed-point scatter) and the trilinear resize backward at ratios other
    than the exact x2 (gather): reproducible bits, autograd's values"""
    gen = torch.Generator().manual_seed(4)
    df = (torch.randn(2, 3, 12, 14, 10, generator=gen) * 2.0)
    img = torch.rand(2, 2, 9, 11, 13, generator=gen)
    up = torch.randn(2, 2, 12, 14, 10, generator=gen)
    dfg, imgg = df.cuda().requires_grad_(True), img.cuda().requires_grad_(True)

    def grads():
        return torch.autograd.grad((ops.warp3d(dfg, imgg) * up.cuda()).sum(), [dfg, imgg])

    a = grads()
    for _ in range(3):
        b = grads()
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    dfr, imgr = df.double().requires_grad_(True), img.double().requires_grad_(True)
    r = torch.autograd.grad((O.warp(dfr, imgr) * up.double()).sum(), [dfr, imgr])
    assert rel_l2(a[0], r[0]) < 1e-5 and rel_l2(a[1], r[1]) < 1e-5
    for in_size, out_size in (((5, 6, 7), (9, 8, 16)), ((8, 8, 8), (32, 32, 32)), ((12, 9, 10), (6, 5, 4)), ((4, 5, 6), (4, 5, 6))):
        xg = torch.randn(2, 3, *in_size, generator=gen)
        u = torch.randn(2, 3, *out_size, generator=gen)
        xc = xg.cuda().requires_grad_(True)
        g0, = torch.autograd.grad((ops.resize_trilinear(xc, out_size, mult=1.5) * u.cuda()).sum(), [xc])
        g1, = torch.autograd.grad((ops.resize_trilinear(xc, out_size, mult=1.5) * u.cuda()).sum(), [xc])
        assert torch.equal(g0, g1)
        xr = xg.double().requires_grad_(True)
        gr, = torch.autograd.grad((1.5 * F.interpolate(xr, size=out_size, mode="trilinear", align_corners=False) * u.double()).sum(), [xr])
        assert rel_l2(g0, gr) < 1e-6, (in_size, out_size, rel_l2(g0, gr))


# ================================================================================================ channel-blocked gradient of the pre-norm tensor (ABI 5)
def _blocked_from_cl(ops, t):
    """channels-last (B, C, D, H, W) -> the blocked buffer [C / 8][B][D][H][W][8]"""
    B, C, D, H, W = t.shape
    g = ops._BlockedGrad(B, C, D, H, W, t.device)
    g.buf.copy_(t.permute(0, 2, 3, 4, 1).reshape(B, D, H, W, C // 8, 8).permute(4, 0, 1, 2, 3, 5).reshape(-1))
    return g


@pytest.mark.parametrize("B,K,N,size", [(2, 32, 32, (32, 32, 32)), (1, 64, 32, (32, 64, 32)), (1, 24, 48, (64, 32, 32))])
def test_data_gradient_on_the_channel_blocked_operand_is_bit_identical(ops, B, K, N, size):
    """pulpo_conv3d_k3_fwd_wino3_kb / _dgrad_wino3_bnred_kb read the SAME numbers through another address map ([K / 8][B][D][H][W][8] instead of
    channels-last): results, BatchNorm statistics rows and the fused BatchNorm-backward partial sums must be bit-identical to the channels-last entry
    points (several batch elements: the batch stride lies inside a channel block; 24 channels: three chunks)"""
    from pulpo_amd._lib import lib
    gen = torch.Generator().manual_seed(K + N)
    D, H, W = size
    dy = torch.randn(B, K, D, H, W, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    w = (torch.randn(K, N, 3, 3, 3, generator=gen) * 0.1).cuda()                     # (a data gradient: weight (Cout = K, Cin = N))
    wpt = ops._pack_weight(w, True, shape=(B, D, H, W))
    assert wpt._pulpo_algo == "wino3"
    blk = _blocked_from_cl(ops, dy)
    assert torch.equal(blk.to_cl(), dy)
    dx0, dx1 = ops.new_cl(B, N, D, H, W, dy.device), ops.new_cl(B, N, D, H, W, dy.device)
    ntile = lib.query("pulpo_conv3d_k3_stat_tiles", B, D, H, W)
    st0, st1 = torch.zeros(ntile * 2 * N, device="cuda"), torch.zeros(ntile * 2 * N, device="cuda")
    ops._conv_raw(dy, wpt, None, dx0, K, N, st0)
    ops._conv_raw(blk, wpt, None, dx1, K, N, st1)
    assert torch.equal(dx0, dx1) and torch.equal(st0, st1)
    # the form with the previous unit's BatchNorm-backward reduction in the epilogue
    y_prev = torch.randn(B, N, D, H, W, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    coef = torch.rand(8 * N, generator=gen).cuda()
    outs = []
    for op_ in (dy, blk):
        dx = ops.new_cl(B, N, D, H, W, dy.device)
        if not ops._dgrad_with_bn_reduction((y_prev, coef), dx, op_, wpt, dx, K, N):
            assert (B, K, N) != (2, 32, 32), "the fused reduction must take the 32 -> 32 case"
            return
        part, nt = ops._take_bn_tile_parts(y_prev, coef, dx)
        outs.append((dx, part.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], dx0)


@pytest.mark.parametrize("pooled", [False, True])
@pytest.mark.parametrize("B,C,size", [(2, 32, (8, 12, 10)), (1, 64, (6, 6, 6)), (1, 24, (5, 7, 9))])
def test_batchnorm_backward_apply_writes_the_blocked_layout(ops, B, C, size, pooled):
    """pulpo_bn_lrelu_bwd_apply_kb_t / _pooled_kb_t: the second BatchNorm-backward pass with its result in [C / 8][B][D][H][W][8] - the same values and the
    same column sums as the channels-last pass (odd sizes: ceil-mode pooling windows at the rims)"""
    from pulpo_amd._lib import lib
    gen = torch.Generator().manual_seed(C + sum(size))
    D, H, W = size
    npix = B * D * H * W
    y = torch.randn(B, C, D, H, W, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    coef = torch.zeros(8 * C)
    coef[:C] = torch.randn(C, generator=gen) * 0.1
    coef[2 * C:3 * C] = torch.rand(C, generator=gen) + 0.5
    coef[3 * C:4 * C] = torch.randn(C, generator=gen) * 0.1
    cd = coef[4 * C:].view(torch.float64)
    cd[:C] = coef[:C].double() + 1e-9
    cd[C:] = torch.rand(C, generator=gen, dtype=torch.float64) + 0.5
    coef = coef.cuda()
    totd = (torch.randn(2 * C, generator=gen, dtype=torch.float64) * 0.01).cuda()
    nblk = lib.query("pulpo_bn_bwd_blocks", npix, C)
    p0, p1 = torch.empty(nblk * C, device="cuda"), torch.empty(nblk * C, device="cuda")
    dy0 = ops.new_cl(B, C, D, H, W, y.device)
    dy1 = ops._BlockedGrad(B, C, D, H, W, y.device)
    st = ops._stream()
    if pooled:
        Do, Ho, Wo = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
        gp = torch.randn(B, C, Do, Ho, Wo, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
        gz = torch.randn(B, C, D, H, W, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
        lib.call("pulpo_bn_lrelu_bwd_apply_pooled_t", ops._ptr(gp), gp.stride(4), ops._ptr(gz), gz.stride(4), 0, ops._ptr(y), 0, y.stride(4), ops._ptr(coef),
                 ops._ptr(totd), ops._ptr(dy0), dy0.stride(4), 0.2, ops._ptr(p0), B, D, H, W, C, st)
        lib.call("pulpo_bn_lrelu_bwd_apply_pooled_kb_t", ops._ptr(gp), gp.stride(4), ops._ptr(gz), gz.stride(4), 0, ops._ptr(y), y.stride(4), ops._ptr(coef),
                 ops._ptr(totd), ops._ptr(dy1.buf), dy1.ps, dy1.kb, 0.2, ops._ptr(p1), B, D, H, W, C, st)
    else:
        dz = torch.randn(B, C, D, H, W, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
        lib.call("pulpo_bn_lrelu_bwd_apply_t", ops._ptr(dz), 0, dz.stride(4), ops._ptr(y), 0, y.stride(4), ops._ptr(coef), ops._ptr(totd), ops._ptr(dy0),
                 dy0.stride(4), npix, C, 0.2, ops._ptr(p0), st)
        lib.call("pulpo_bn_lrelu_bwd_apply_kb_t", ops._ptr(dz), 0, dz.stride(4), 8, ops._ptr(y), y.stride(4), ops._ptr(coef), ops._ptr(totd), ops._ptr(dy1.buf),
                 dy1.ps, dy1.kb, npix, C, 0.2, ops._ptr(p1), st)
        # dz blocked as well (the gradient of a blocked activation), dy channels-last and blocked
        dzb = ops.cl_to_blocked(dz)
        assert torch.equal(ops.blocked_to_cl(dzb), dz)
        dy2, p2 = ops.new_cl(B, C, D, H, W, y.device), torch.empty(nblk * C, device="cuda")
        lib.call("pulpo_bn_lrelu_bwd_apply_kb_t", ops._ptr(dzb), 0, 8, npix * 8, ops._ptr(y), y.stride(4), ops._ptr(coef), ops._ptr(totd), ops._ptr(dy2),
                 dy2.stride(4), 8, npix, C, 0.2, ops._ptr(p2), st)
        assert torch.equal(dy2, dy0) and torch.equal(p2, p0)
        dy3 = ops._BlockedGrad(B, C, D, H, W, y.device)
        lib.call("pulpo_bn_lrelu_bwd_apply_kb_t", ops._ptr(dzb), 0, 8, npix * 8, ops._ptr(y), y.stride(4), ops._ptr(coef), ops._ptr(totd), ops._ptr(dy3.buf),
                 dy3.ps, dy3.kb, npix, C, 0.2, ops._ptr(p2), st)
        assert torch.equal(dy3.to_cl(), dy0) and torch.equal(p2, p0)
        # the forward pass's counterpart: z blocked
        z0, z1 = ops.new_cl(B, C, D, H, W, y.device), torch.empty(C // 8, B, D, H, W, 8, device="cuda")
        lib.call("pulpo_bn_lrelu_apply", ops._ptr(y), y.stride(4), ops._ptr(z0), z0.stride(4), ops._ptr(coef), npix, C, 0.2, st)
        lib.call("pulpo_bn_lrelu_apply_kb", ops._ptr(y), y.stride(4), ops._ptr(z1), 8, npix * 8, ops._ptr(coef), npix, C, 0.2, st)
        assert torch.equal(ops.blocked_to_cl(z1), z0)
    assert torch.equal(dy1.to_cl(), dy0)
    assert torch.equal(p0, p1)


@pytest.mark.parametrize("det", [False, True])
@pytest.mark.parametrize("B,Cin,Cout,size", [(2, 32, 32, (32, 32, 32)), (1, 16, 96, (10, 20, 28)), (1, 64, 40, (6, 12, 17))])
def test_weight_gradient_on_the_channel_blocked_gradient(ops, B, Cin, Cout, size, det):
    """pulpo_conv3d_k3_wgrad_kb: the F(2x2x2,3x3x3) weight-gradient kernel reading dy as [Cout / 8][B][D][H][W][8] (ragged rows and columns, 40 output
    channels: a partly empty 32-channel tile whose last block is the fifth) - bit-identical to the channels-last call in deterministic mode, within the
    atomics' summation-order noise otherwise"""
    gen = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, *size, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    dy = torch.randn(B, Cout, *size, generator=gen).cuda().contiguous(memory_format=torch.channels_last_3d)
    blk = _blocked_from_cl(ops, dy)
    was = ops.DETERMINISTIC
    ops.set_deterministic(det)
    try:
        g0 = ops._wgrad_raw(x, dy, Cin, Cout)
        g1 = ops._wgrad_raw(x, blk, Cin, Cout)
    finally:
        ops.set_deterministic(was)
    if det:
        assert torch.equal(g0, g1)
    else:
        assert rel_l2(g1, g0) < 1e-5
    ref = torch.nn.grad.conv3d_weight(x.double(), (Cout, Cin, 3, 3, 3), dy.double(), padding=1)
    assert rel_l2(g1, ref) < 2e-5


@pytest.mark.parametrize("B,chans,size", [(1, (2, 32, 32, 32), (64, 64, 64)), (2, (16, 32, 64, 64), (32, 64, 64)), (1, (24, 48, 48), (64, 32, 64))])
def test_conv_sequence_with_blocked_activations_between_its_units(ops, B, chans, size):
    """the activations between the ConvUnits of a sequence as (C / 8, B, D, H, W, 8) tensors (ops.BLOCKED_Z; the input layer's fused BatchNorm backward
    reads a blocked dz; two batch elements; 24 -> 48 -> 48 channels): outputs bit-identical to the channels-last run, every parameter and input
    gradient within the summation-order noise of the atomics, hooks see the 6-D tensors"""
    import src.network_blocks as nb
    torch.manual_seed(5)
    units = [nb.ConvUnit(list(size), chans[k], chans[k + 1]) for k in range(len(chans) - 1)]
    seq = nb.ConvSequence(list(size), chans[0], chans[1], 1)
    seq._op = torch.nn.Sequential(*units)
    seq = seq.cuda().train()
    state = {k: v.clone() for k, v in seq.state_dict().items()}
    gen = torch.Generator().manual_seed(B + sum(chans))
    x = torch.randn(B, chans[0], *size, generator=gen).cuda().requires_grad_(chans[0] > 3)
    up = torch.randn(B, chans[-1], *size, generator=gen).cuda()
    seen = []
    for u in units[:-1]:
        u.register_forward_hook(lambda m, i, o: seen.append(o.dim()))

    def run(on):
        seq.load_state_dict(state)
        was, was_min = ops.BLOCKED_Z, ops.BLOCKED_Z_MIN_VOXELS
        ops.BLOCKED_Z, ops.BLOCKED_Z_MIN_VOXELS = on, 1
        try:
            z = seq(x)
            ps = list(seq.parameters()) + ([x] if x.requires_grad else [])
            return z, torch.autograd.grad((z * up).sum(), ps)
        finally:
            ops.BLOCKED_Z, ops.BLOCKED_Z_MIN_VOXELS = was, was_min

    h0 = ops.BLOCKED_Z_HITS
    z1, g1 = run(True)
    assert ops.BLOCKED_Z_HITS > h0 and 6 in seen, "no unit of the sequence produced a blocked activation"
    seen.clear()
    z0, g0 = run(False)
    assert seen and all(d == 5 for d in seen)
    assert z1.dim() == 5 and torch.equal(z1, z0)
    names = [n for n, _ in seq.named_parameters()] + (["x"] if x.requires_grad else [])
    for a, b, nme in zip(g1, g0, names):
        if nme.endswith("0.bias"):                   # (conv bias in front of a BatchNorm: true gradient zero)
            assert float((a - b).abs().max()) <= 1e-4 * max(float(g0[0].abs().max()), 1e-6), nme
        else:
            assert rel_l2(a, b) < 2e-5, (nme, rel_l2(a, b))
