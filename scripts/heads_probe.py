"""heads_fwd / heads_bwd (the 1x1x1 mu / sigma / velocity heads) at the step's shapes, alone on the machine: time and achieved bandwidth.
Round 5: 2.3 TB/s at 96 channels x 6 outputs (85 / 168 us at 80^3; 3.9 / 5.7 TB/s at 32 channels x 3 outputs).  Larger grids (4096 workgroups: forward 85 -> 70 us),
2 / 4 pixels per trip in the backward kernel (168 -> 165 / 233 us): not what bounds it - the backward kernel issued 16 load instructions per wave and
trip, 15 of them the same five planar gradient / noise / sigma values for every thread of a pixel's row.  Those now go through LDS (fetched once per row):
170 -> 130 us at 80^3 x 96 channels alone on the machine; inside the step the difference is below the run-to-run noise (0.3 ms of heads kernels in total).
usage: python scripts/heads_probe.py"""
import sys, os, torch
sys.path.insert(0, '.')
from pulpo_amd import ops
from pulpo_amd._lib import lib
lib.load()
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for C, S, nout in ((96, 80, 6), (96, 40, 6), (96, 20, 6), (32, 80, 3), (32, 40, 3)):
    V = S ** 3
    h = ops.new_cl(1, C, S, S, S, "cuda").normal_()
    Wt = torch.randn(nout, C, device="cuda"); bias = torch.randn(nout, device="cuda")
    eps = torch.randn(1, 3, S, S, S, device="cuda")
    o = [torch.empty(1, 3, S, S, S, device="cuda") for _ in range(3)]
    g = [torch.randn(1, 3, S, S, S, device="cuda") for _ in range(3)]
    dh = torch.empty_like(h)
    nblk = lib.query("pulpo_heads_bwd_blocks", 1, V, C)
    part = torch.empty(nblk * (nout * C + nout), device="cuda")
    st = ops._stream()
    tf = t(lambda: lib.call("pulpo_heads_fwd", ops._ptr(h), h.stride(4), ops._ptr(Wt), ops._ptr(bias), ops._ptr(eps) if nout == 6 else None, ops._ptr(o[0]),
                            ops._ptr(o[1]) if nout == 6 else None, ops._ptr(o[2]) if nout == 6 else None, nout, 1, V, C, st))
    sig = o[1].abs() + 0.1
    tb = t(lambda: lib.call("pulpo_heads_bwd", ops._ptr(h), h.stride(4), ops._ptr(Wt), ops._ptr(g[0]), ops._ptr(g[1]) if nout == 6 else None, ops._ptr(g[2]) if nout == 6 else None,
                            ops._ptr(eps) if nout == 6 else None, ops._ptr(sig) if nout == 6 else None, ops._ptr(dh), dh.stride(4), ops._ptr(part), nout, 1, V, C, st))
    print(f"C={C:3d} S={S:3d} nout={nout}: fwd {tf*1e6:6.1f} us ({4.0*C*V/tf/1e12:.2f} TB/s)   bwd {tb*1e6:6.1f} us ({8.0*C*V/tb/1e12:.2f} TB/s)   [{nblk} bwd blocks]")
