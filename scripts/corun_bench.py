"""do the Winograd data-gradient and weight-gradient kernels of a layer overlap when queued on two streams?  (time of both queued
concurrently vs the sum of their stand-alone times)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulpo_amd import ops
from pulpo_amd._lib import lib
lib.load()
side = torch.cuda.Stream()
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for ci, co, S in ((32, 32, 160), (64, 64, 80), (96, 96, 80), (128, 128, 40)):
    x = torch.randn(1, ci, S, S, S, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    dy = torch.randn(1, co, S, S, S, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
    dx = ops.new_cl(1, ci, S, S, S, x.device)
    wpt = ops._pack_weight(w, True, shape=(1, S, S, S))
    dw = torch.zeros_like(w)
    def dgrad(): ops._conv_raw(dy, wpt, None, dx, co, ci, None)
    def wgrad(): ops._wgrad_raw(x, dy, ci, co, into=dw)
    def both():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            wgrad()
        dgrad()
        main.wait_stream(side)
    td, tw, tb = timeit(dgrad), timeit(wgrad), timeit(both)
    print(f"{ci}->{co}@{S}^3: dgrad {td:.3f} ms  wgrad {tw:.3f} ms  sum {td+tw:.3f}  concurrent {tb:.3f} ms  ({(td+tw)/tb:.2f}x)")
