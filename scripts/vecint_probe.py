"""VecInt forward / backward stand-alone at the pyramid's field sizes (HIP events, one squaring chain of 7 steps)
usage: python scripts/vecint_probe.py [sizes ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulpo_amd import ops
from pulpo_amd._lib import lib
lib.load()
sizes = [int(s) for s in sys.argv[1:]] or [80, 40, 20, 10]
for S in sizes:
    torch.manual_seed(S)
    v = (torch.randn(1, 3, S, S, S, device="cuda") * float(os.environ.get("AMP", "2.0"))).requires_grad_(True)
    up = torch.randn(1, 3, S, S, S, device="cuda")
    def fwd(): return ops.vecint(v, 7)
    out = fwd()
    def bwd(): return torch.autograd.grad((out * up).sum(), [v], retain_graph=True)
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        print(f"{S:4d}^3 vecint {name}: {s.elapsed_time(e) / 20 * 1e3:8.1f} us per call (7 squaring steps)")
