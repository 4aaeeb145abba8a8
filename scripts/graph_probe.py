"""Does the training step capture into a HIP graph, and what does replay buy?  (probe; the product integration is dp.DataParallelStepper(graph=True))"""
import os, sys, time, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulpo_amd import dp, ops, synthetic
from pulpo_amd._lib import lib
from src.models import PULPo

FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
T, L = (5, 4) if S >= 64 else (3, 2)
lib.load()
torch.manual_seed(0)
dev = torch.device("cuda")
model = PULPo(T, L, 0.1, [S, S, S], feedback=FB, n0=32).to(dev).train()
stepper = dp.DataParallelStepper(model)
x, y = synthetic.uniform_pair([S, S, S], 1, 1234, dev)
e = torch.empty((0,), device=dev)
batch = (x, y, e, e, e, e, e, e)

def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

if os.environ.get("EAGER_FIRST", "0") == "1":
    for _ in range(3): stepper.step(batch)
    print("eager ms/step", timeit(lambda: stepper.step(batch)))

# capture forward + backward (gradients into the arena); Adam stays eager
model._arm_nan_probe = lambda *_a, **_k: None
model._check_previous_step_for_nan = lambda: None
side = stepper._side
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        stepper.arena.zero_grad()
        ops.DIRECT_PARAM_GRADS = True; ops.ASYNC_WGRAD_STREAM = side
        loss = model.training_step(batch, 0); loss.backward(); ops.join_async_wgrad()
        ops.DIRECT_PARAM_GRADS = False; ops.ASYNC_WGRAD_STREAM = None
        del loss
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
MODE = os.environ.get("PROBE", "full")
print("capturing", MODE, flush=True)
with torch.cuda.graph(g):
    stepper.arena.grad.zero_()
    ops.DIRECT_PARAM_GRADS = True; ops.ASYNC_WGRAD_STREAM = side if MODE == "full" else None
    print(" fwd", flush=True)
    static_loss = model.training_step(batch, 0)
    if MODE != "fwd":
        print(" bwd", flush=True)
        static_loss.backward()
        ops.join_async_wgrad()
    ops.DIRECT_PARAM_GRADS = False; ops.ASYNC_WGRAD_STREAM = None
print("captured")
def gstep():
    g.replay()
    stepper.opt.step(1.0)
print("graph ms/step", timeit(gstep), "loss", float(static_loss))
