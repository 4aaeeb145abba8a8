import sys, torch, ctypes
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
for C, S in ((32, 160), (64, 80), (128, 40)):
    y = ops.new_cl(1, C, S, S, S, "cuda").normal_()
    z = torch.empty_like(y)
    coef = torch.rand(8 * C, device="cuda")
    npix = S ** 3
    nb = 8.0 * C * npix
    ta = t(lambda: lib.call("pulpo_bn_lrelu_apply", ops._ptr(y), y.stride(4), ops._ptr(z), z.stride(4), ops._ptr(coef), npix, C, 0.2, ops._stream()))
    tc = t(lambda: z.copy_(y))
    print(f"C={C} S={S}: bn_apply {nb/ta/1e12:.2f} TB/s ({ta*1e6:.0f} us)   torch copy {nb/tc/1e12:.2f} TB/s ({tc*1e6:.0f} us)")
