"""Stand-alone timing of the 2-/3-channel input layers (forward convolution and weight gradient), with a correctness spot check."""
import sys, os, torch
sys.path.insert(0, "/root/repo")
from pulpo_amd import ops
from pulpo_amd._lib import lib
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
for ci, co, S in ((2, 32, 160), (3, 32, 80), (3, 32, 40)):
    x = torch.randn(1, ci, S, S, S, device="cuda")
    dy = torch.randn(1, co, S, S, S, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.1
    y = ops.new_cl(1, co, S, S, S, x.device)
    stats = torch.empty(lib.query("pulpo_conv3d_k3_stat_tiles", 1, S, S, S) * 2 * co, device="cuda")
    wp = ops._pack_weight(w, False, shape=(1, S, S, S))
    t1 = timeit(lambda: ops._conv_raw(x, wp, None, y, ci, co, stats))
    t2 = timeit(lambda: ops._wgrad_raw(x, dy, ci, co))
    ref = torch.nn.functional.conv3d(x.double().cpu()[:, :, :20, :20, :20], w.double().cpu(), padding=1)[..., 1:19, 1:19, 1:19]
    err = float((y.cpu().double()[:, :, 1:19, 1:19, 1:19] - ref).abs().max())
    print(f"{ci}->{co} @{S}^3: fwd {t1:7.1f} us  wgrad {t2:7.1f} us   fwd err {err:.2e}")
