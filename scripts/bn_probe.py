"""Stand-alone bandwidth of the three BatchNorm/LeakyReLU streaming kernels at the shapes of the 160^3 training step.

    python scripts/bn_probe.py

Measured (round 2): 4.7-6.6 TB/s at every level of 40^3 and above, i.e. 75-100 % of the 6.3 TB/s a float4 copy reaches on this part;
unrolling the grid-stride loops 2x / 4x (more loads in flight per thread) changed nothing, so these passes are only removed by
fusing them into their neighbours, not sped up.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulpo_amd._lib import lib
from pulpo_amd.ops import _ptr, _stream


SHAPES = [(160, 32), (160, 16), (80, 96), (80, 64), (40, 128), (40, 96), (20, 160)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    dev = torch.device("cuda:0")
    tot = {}
    for u in ("-",):
        for S, C in SHAPES:
            npix = S ** 3
            y = torch.randn(npix, C, device=dev)
            dz = torch.randn(npix, C, device=dev)
            z = torch.empty_like(y)
            coef = torch.zeros(8 * C, device=dev)
            coef[C:2 * C] = 1; coef[2 * C:3 * C] = 1
            cd = coef[4 * C:].view(torch.float64); cd[C:] = 1.0
            totd = torch.zeros(2 * C, dtype=torch.float64, device=dev)
            nblk = lib.query("pulpo_bn_bwd_blocks", npix, C)
            part = torch.empty(nblk * 2 * C, device=dev)
            res = []
            for name, nb, fn in (
                ("apply", 8, lambda: lib.call("pulpo_bn_lrelu_apply", _ptr(y), C, _ptr(z), C, _ptr(coef), npix, C, 0.2, _stream())),
                ("bwd_reduce", 8, lambda: lib.call("pulpo_bn_lrelu_bwd_reduce", _ptr(dz), C, _ptr(y), C, _ptr(coef), npix, C, 0.2, _ptr(part), _stream())),
                ("bwd_apply", 12, lambda: lib.call("pulpo_bn_lrelu_bwd_apply", _ptr(dz), C, _ptr(y), C, _ptr(coef), _ptr(totd), _ptr(z), C, npix, C, 0.2,
                                                   _ptr(part), _stream())),
            ):
                t = timeit(fn)
                res.append(f"{name} {t*1e6:7.1f} us {nb*npix*C/t/1e12:5.2f} TB/s")
                tot[(u, name)] = tot.get((u, name), 0.0) + t
            print(f"U={u} {S}^3 x {C:3d}: " + " | ".join(res), flush=True)
    for k, v in tot.items():
        print(k, f"{v*1e3:.3f} ms")


if __name__ == "__main__":
    main()
