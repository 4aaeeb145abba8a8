"""Per-shape table of the convolution / weight-gradient launches of one 160^3 training step: in the overlapped step (two streams) and
serialized (weight gradients on the main stream), from HIP events around every launch.

    python scripts/conv_shapes.py [size]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulpo_amd import dp, ops, synthetic
from src.models import PULPo

FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
torch.manual_seed(0)
dev = torch.device("cuda")
model = PULPo(5, 4, 0.1, [S, S, S], feedback=FB, n0=32).to(dev).train()
x, y = synthetic.uniform_pair([S, S, S], 1, 1234, dev)
e = torch.empty((0,), device=dev)
batch = (x, y, e, e, e, e, e, e)


def run(overlap: bool, nstep: int = 3):
    os.environ["PULPO_DP_OVERLAP"] = "1" if overlap else "0"
    stepper = dp.DataParallelStepper(model)
    for _ in range(2):
        stepper.step(batch)
    torch.cuda.synchronize()
    ops.CONV_TRACE, ops.CONV_TRACE_STRIDE = [], 1
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(nstep):
        stepper.step(batch)
    t1.record()
    torch.cuda.synchronize()
    trace, ops.CONV_TRACE = ops.CONV_TRACE, None
    agg = {}
    for name, flops, a, b, nbytes in trace:
        k = (name, flops, nbytes)
        c = agg.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += a.elapsed_time(b) * 1e-3
    return agg, t0.elapsed_time(t1) / nstep, nstep


over, ms_o, n = run(True)
ser, ms_s, _ = run(False)
print(f"step: overlapped {ms_o:.2f} ms (events around every launch), serialized {ms_s:.2f} ms")
rows = sorted(over, key=lambda k: -over[k][1])
tot_o = tot_s = 0.0
print(f"{'kernel':34s} {'GFLOP':>8s} {'MB':>7s} {'n/step':>6s} | {'us over':>8s} {'TF':>6s} | {'us serial':>9s} {'TF':>6s} | ms/step over, serial")
for k in rows:
    name, flops, nbytes = k
    c, t = over[k]
    cs, ts = ser.get(k, (0, 0.0))
    us_o = t / c * 1e6
    us_s = ts / cs * 1e6 if cs else float("nan")
    tot_o += t / n; tot_s += ts / n
    print(f"{name:34s} {flops/1e9:8.1f} {nbytes/1e6:7.0f} {c/n:6.1f} | {us_o:8.1f} {flops/(t/c)/1e12:6.1f} | {us_s:9.1f} {flops/(ts/cs)/1e12 if cs else 0:6.1f} | {t/n*1e3:6.2f} {ts/n*1e3:6.2f}")
print(f"total conv-class kernel time per step: overlapped {tot_o*1e3:.2f} ms, serialized {tot_s*1e3:.2f} ms")
