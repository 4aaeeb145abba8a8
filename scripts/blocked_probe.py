"""The F(2x2x2,3x3x3) forward / data-gradient kernel on a channels-last operand against the same operand in the channel-blocked layout
[K / 8][D][H][W][8] (pulpo_conv3d_k3_fwd_wino3_kb): results must be bit-identical, the time is what the tap loads' cache-line footprint costs.
usage: python scripts/blocked_probe.py [reps]"""
import sys, torch
sys.path.insert(0, '.')
from pulpo_amd import ops
from pulpo_amd._lib import lib

lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def t(fn, n=reps):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


tot = [0.0, 0.0]
for ci, co, S, cnt in ((32, 32, 80, 0), (32, 32, 160, 2), (32, 64, 80, 1), (64, 64, 80, 3), (96, 96, 80, 1), (160, 64, 80, 1), (64, 128, 40, 1), (128, 128, 40, 3), (224, 128, 40, 1)):
    x = torch.randn(1, ci, S, S, S, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    V = S ** 3
    xb = x.permute(0, 2, 3, 4, 1).reshape(V, ci // 8, 8).permute(1, 0, 2).contiguous()          # [K / 8][V][8]
    w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
    wp = ops._pack_weight(w, False, shape=(1, S, S, S))
    assert wp._pulpo_algo == "wino3", wp._pulpo_algo
    y0, y1 = ops.new_cl(1, co, S, S, S, x.device), ops.new_cl(1, co, S, S, S, x.device)
    st = ops._stream()
    y3 = torch.empty(co // 8, V, 8, device="cuda")
    f0 = lambda: lib.call("pulpo_conv3d_k3_fwd_wino3_kb", ops._ptr(x), x.stride(0), x.stride(4), 8, ops._ptr(wp), None, None, 0.0, ops._ptr(y0), y0.stride(0), y0.stride(4), 8,
                          None, 1, S, S, S, ci, co, st)
    f1 = lambda: lib.call("pulpo_conv3d_k3_fwd_wino3_kb", ops._ptr(xb), xb.numel(), 8, V * 8, ops._ptr(wp), None, None, 0.0, ops._ptr(y1), y1.stride(0), y1.stride(4), 8,
                          None, 1, S, S, S, ci, co, st)
    f3 = lambda: lib.call("pulpo_conv3d_k3_fwd_wino3_kb", ops._ptr(xb), xb.numel(), 8, V * 8, ops._ptr(wp), None, None, 0.0, ops._ptr(y3), y3.numel(), 8, V * 8,
                          None, 1, S, S, S, ci, co, st)
    t0, t1, t3 = t(f0), t(f1), t(f3)
    same = torch.equal(y0, y1) and torch.equal(y3.permute(1, 0, 2).reshape(V, co), y0.permute(0, 2, 3, 4, 1).reshape(V, co))
    if not same:
        y2 = ops.new_cl(1, co, S, S, S, x.device)
        ops._conv_raw(x, wp, None, y2, ci, co, None)
        ref = torch.nn.functional.conv3d(x.double(), w.double(), padding=1) if S <= 80 else None
        d = (y0 - y1).abs()
        print(f"    max |cl - blocked| {d.max().item():.3e} at {tuple(int(v) for v in torch.unravel_index(d.argmax(), d.shape))}; legacy == cl: {torch.equal(y2, y0)}; legacy == blocked: {torch.equal(y2, y1)}"
              + (f"; vs fp64: cl {(y0 - ref).abs().max().item():.2e} blocked {(y1 - ref).abs().max().item():.2e}" if ref is not None else ""))
    tot[0] += t0 * cnt; tot[1] += t1 * cnt
    print(f"{ci:4d}->{co:3d} @{S:3d}^3 x{cnt}: channels-last {t0:.3f} ms   blocked in {t1:.3f} ms ({(t1 / t0 - 1) * 100:+.1f} %)   blocked in + out {t3:.3f} ms ({(t3 / t0 - 1) * 100:+.1f} %)   bit-identical: {same}")
print(f"count-weighted: channels-last {tot[0]:.2f} ms, blocked {tot[1]:.2f} ms")

# ---- the other two kernels that touch dy: the BatchNorm-backward apply pass that writes it, the weight gradient that reads it
print("\nbn_lrelu_bwd_apply (writes dy) and conv3d_k3_wgrad (reads dy): channels-last / blocked, ms")
for ci, co, S in ((32, 32, 160), (64, 64, 80), (96, 96, 80), (128, 128, 40)):
    V = S ** 3
    x = torch.randn(1, ci, S, S, S, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    y = torch.randn(1, co, S, S, S, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    dz = torch.randn(1, co, S, S, S, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    coef = torch.rand(8 * co, device="cuda")
    totd = torch.zeros(2 * co, device="cuda", dtype=torch.float64)
    nblk = lib.query("pulpo_bn_bwd_blocks", V, co)
    p2 = torch.empty(nblk * co, device="cuda")
    dy0 = ops.new_cl(1, co, S, S, S, x.device)
    dy1 = ops._BlockedGrad(1, co, S, S, S, x.device)
    st = ops._stream()
    a0 = t(lambda: lib.call("pulpo_bn_lrelu_bwd_apply_t", ops._ptr(dz), 0, dz.stride(4), ops._ptr(y), 0, y.stride(4), ops._ptr(coef), ops._ptr(totd), ops._ptr(dy0),
                            dy0.stride(4), V, co, 0.2, ops._ptr(p2), st))
    a1 = t(lambda: lib.call("pulpo_bn_lrelu_bwd_apply_kb_t", ops._ptr(dz), 0, dz.stride(4), ops._ptr(y), y.stride(4), ops._ptr(coef), ops._ptr(totd), ops._ptr(dy1.buf),
                            dy1.ps, dy1.kb, V, co, 0.2, ops._ptr(p2), st))
    w0 = t(lambda: ops._wgrad_raw(x, dy0, ci, co))
    w1 = t(lambda: ops._wgrad_raw(x, dy1, ci, co))
    xk = x.permute(0, 2, 3, 4, 1).reshape(V, ci // 8, 8).permute(1, 0, 2).contiguous()
    nscr = lib.query("pulpo_conv3d_k3_wgrad_scratch_floats", ci, co)
    scr, dw = torch.empty(nscr, device="cuda"), torch.empty(co, ci, 3, 3, 3, device="cuda")
    w2 = t(lambda: lib.call("pulpo_conv3d_k3_wgrad_kb", ops._ptr(xk), xk.numel(), 8, V * 8, ops._ptr(dy1.buf), dy1.bs, dy1.ps, dy1.kb, ops._ptr(dw), 0, ops._ptr(scr), None, 0,
                            1, S, S, S, ci, co, st))
    ref = ops._wgrad_raw(x, dy1, ci, co)
    err = float((dw - ref).norm() / ref.norm())
    print(f"{ci:4d}->{co:3d} @{S:3d}^3: bwd_apply {a0:.3f} / {a1:.3f} ({(a1 / a0 - 1) * 100:+.1f} %)   wgrad {w0:.3f} / dy blocked {w1:.3f} ({(w1 / w0 - 1) * 100:+.1f} %) / x and dy blocked {w2:.3f} "
          f"({(w2 / w0 - 1) * 100:+.1f} %, rel. diff {err:.1e})")
