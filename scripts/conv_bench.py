"""Micro-benchmark of the MFMA convolution kernels on the layer shapes of the 160^3 / T5 / L4 training step.
usage: python scripts/conv_bench.py [--reps 10] [--only fwd|dgrad|wgrad] [--precision fp32|bf16] [--activations fp32|bf16]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulpo_amd import ops
from pulpo_amd._lib import lib

SHAPES = [  # (Cin, Cout, S, count in the step as fwd)
    (32, 32, 160, 2), (32, 64, 80, 1), (64, 64, 80, 3), (96, 96, 80, 1), (160, 64, 80, 1), (16, 96, 80, 1), (32, 32, 80, 1),
    (64, 128, 40, 1), (128, 128, 40, 3), (96, 96, 40, 1), (224, 128, 40, 1),
    (128, 192, 20, 1), (192, 192, 20, 3), (288, 192, 20, 1), (192, 192, 10, 3),
]

def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=10); ap.add_argument("--only", default="")
    ap.add_argument("--shapes", type=int, nargs="*", default=None)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"], help="conv operand precision (bf16 = BASELINE configs 4-5)")
    ap.add_argument("--algo", default=None, choices=["direct", "wino2", "wino3"], help="force the forward / data-gradient kernel among those valid for a shape (default: the library's choice; PULPO_CONV_WINO3=0 / PULPO_CONV_WINO3_MINK=<k> move the F(2x2x2,3x3x3) policy)")
    ap.add_argument("--activations", default="fp32", choices=["fp32", "bf16"], help="activation storage of the operands and results (bf16 with --precision bf16: configs 4-5)")
    a = ap.parse_args()
    lib.load()
    ops.set_conv_precision(a.precision, activations=a.activations)
    adt = torch.bfloat16 if a.activations == "bf16" else torch.float32
    ops.CONV_ALGO = a.algo
    tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
    print(f"{'shape':>22s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF/s':>6s} | {'dgrad ms':>8s} {'TF/s':>6s} | {'wgrad ms':>8s} {'TF/s':>6s}")
    for idx, (ci, co, S, cnt) in enumerate(SHAPES):
        if a.shapes is not None and idx not in a.shapes: continue
        x = torch.randn(1, ci, S, S, S, device="cuda").to(adt).contiguous(memory_format=torch.channels_last_3d)
        dy = torch.randn(1, co, S, S, S, device="cuda").to(adt).contiguous(memory_format=torch.channels_last_3d)
        w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
        y = ops.new_cl(1, co, S, S, S, x.device, adt); dx = ops.new_cl(1, ci, S, S, S, x.device, adt)
        stats = torch.empty(lib.query("pulpo_conv3d_k3_stat_tiles", 1, S, S, S) * 2 * co * 2, device="cuda")     # (x2: room for either precision's tiling)
        wp, wpt = ops._pack_weight(w, False, shape=(1, S, S, S)), ops._pack_weight(w, True, shape=(1, S, S, S))
        fl = 54.0 * ci * co * S ** 3
        res = []
        for name, fn in (("fwd", lambda: ops._conv_raw(x, wp, None, y, ci, co, stats)), ("dgrad", lambda: ops._conv_raw(dy, wpt, None, dx, co, ci, None)),
                         ("wgrad", lambda: ops._wgrad_raw(x, dy, ci, co))):
            if a.only and a.only != name: res += [float("nan"), float("nan")]; continue
            t = timeit(fn, a.reps); res += [t * 1e3, fl / t / 1e12]
            tot[name][0] += t * cnt; tot[name][1] += fl * cnt
        print(f"{ci:4d}->{co:3d} @{S:3d}^3 x{cnt:d}   {fl/1e9:8.1f} | {res[0]:8.3f} {res[1]:6.1f} | {res[2]:8.3f} {res[3]:6.1f} | {res[4]:8.3f} {res[5]:6.1f}")
    for k, (t, f) in tot.items():
        if t: print(f"step-weighted {k}: {t*1e3:.2f} ms, {f/t/1e12:.1f} TFLOP/s")

if __name__ == "__main__":
    main()
