"""Forward convolution reading the producing unit's pre-norm tensor (BatchNorm + LeakyReLU applied while the operand is staged, z written on
the way) against the plain kernel + the separate apply pass it replaces, per layer shape of the 160^3 step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulpo_amd import ops
from pulpo_amd._lib import lib
from pulpo_amd.ops import _ptr, _stream


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for ci, co, S in ((32, 32, 160), (64, 64, 80), (96, 96, 80), (128, 128, 40), (32, 64, 80)):
    yprev = torch.randn(1, ci, S, S, S, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    z = torch.empty_like(yprev)
    coef = torch.rand(8 * ci, device="cuda") + 0.5
    w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
    out = ops.new_cl(1, co, S, S, S, yprev.device)
    stats = torch.empty(lib.query("pulpo_conv3d_k3_stat_tiles", 1, S, S, S) * 2 * co, device="cuda")
    wp = ops._pack_weight(w, False, shape=(1, S, S, S))
    npix = S ** 3
    t_apply = timeit(lambda: lib.call("pulpo_bn_lrelu_apply", _ptr(yprev), ci, _ptr(z), ci, _ptr(coef), npix, ci, 0.2, _stream()))
    t_plain = timeit(lambda: ops._conv_raw(z, wp, None, out, ci, co, stats))
    t_pre = timeit(lambda: ops._conv_raw_prenorm(yprev, coef, z, wp, None, out, ci, co, stats))
    print(f"{ci:3d}->{co:3d} @{S:3d}^3: apply {t_apply:7.1f} us + conv {t_plain:7.1f} us = {t_apply + t_plain:7.1f} | prenorm conv {t_pre:7.1f} us")
