"""Achieved HBM bandwidth of the three BatchNorm passes (apply, backward reduce, backward apply) at the step's layer shapes, next to a plain device
copy of the same tensors (round 5: the passes run AT the copy's rate - 5.1 - 5.9 TB/s on 524 MB tensors, 6.2 - 6.9 on 131 MB ones that the Infinity
Cache helps; a channel-bound thread with register constants and 1 / 2 / 4 pixels in flight, 1024 - 8192 workgroups and offset buffers all measured the
same, profiles/r5_bn_bandwidth.txt - what is left in them is bytes, not rate).
--offset N: the second / third tensor of a pass starts N bytes behind a 2 MiB boundary (do the streams of a pass collide on HBM channels?)"""
import sys, argparse, torch
sys.path.insert(0, '.')
from pulpo_amd import ops
from pulpo_amd._lib import lib

ap = argparse.ArgumentParser()
ap.add_argument("--offset", type=int, default=0)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
lib.load()


def t(fn, n=a.reps):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


def cl(C, S, k):
    """channels-last [1, C, S, S, S] view whose storage starts k * offset bytes behind an allocation boundary"""
    n = C * S ** 3
    pad = (k * a.offset) // 4
    buf = torch.empty(n + pad, device="cuda")
    return buf[pad:].view(1, S, S, S, C).permute(0, 4, 1, 2, 3).normal_()


print(f"offset={a.offset}")
tot = [0.0, 0.0, 0.0]
for C, S, cnt in ((32, 160, 3), (64, 80, 5), (96, 80, 3), (128, 40, 5), (192, 20, 5)):
    y, z, dz, dy = cl(C, S, 0), cl(C, S, 1), cl(C, S, 2), cl(C, S, 3)
    coef = torch.rand(8 * C, device="cuda")
    totd = torch.zeros(2 * C, device="cuda", dtype=torch.float64)
    npix = S ** 3
    nblk = lib.query("pulpo_bn_bwd_blocks", npix, C)
    part = torch.empty(nblk * 2 * C, device="cuda")
    part2 = torch.empty(nblk * C, device="cuda")
    st = ops._stream()
    ta = t(lambda: lib.call("pulpo_bn_lrelu_apply_t", ops._ptr(y), 0, y.stride(4), ops._ptr(z), 0, z.stride(4), ops._ptr(coef), npix, C, 0.2, st))
    tr = t(lambda: lib.call("pulpo_bn_lrelu_bwd_reduce_t", ops._ptr(dz), 0, dz.stride(4), ops._ptr(y), 0, y.stride(4), ops._ptr(coef), npix, C, 0.2, ops._ptr(part), st))
    tb = t(lambda: lib.call("pulpo_bn_lrelu_bwd_apply_t", ops._ptr(dz), 0, dz.stride(4), ops._ptr(y), 0, y.stride(4), ops._ptr(coef), ops._ptr(totd), ops._ptr(dy),
                            dy.stride(4), npix, C, 0.2, ops._ptr(part2), st))
    tc = t(lambda: z.copy_(y))
    n4 = 4.0 * C * npix
    tot[0] += ta * cnt; tot[1] += tr * cnt; tot[2] += tb * cnt
    print(f"C={C:3d} S={S:3d}: apply {2*n4/ta/1e12:.2f} TB/s ({ta*1e6:6.1f} us)  bwd_reduce {2*n4/tr/1e12:.2f} TB/s ({tr*1e6:6.1f} us)  "
          f"bwd_apply {3*n4/tb/1e12:.2f} TB/s ({tb*1e6:6.1f} us)  torch copy {2*n4/tc/1e12:.2f} TB/s ({tc*1e6:6.1f} us)")
print(f"count-weighted: apply {tot[0]*1e3:.3f} ms  bwd_reduce {tot[1]*1e3:.3f} ms  bwd_apply {tot[2]*1e3:.3f} ms")
