"""How a 160^3 training step splits into forward and backward + optimizer (HIP events on the main stream, a synchronisation in between)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulpo_amd import dp, synthetic, ops
from pulpo_amd._lib import lib
from src.models import PULPo
FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
lib.load(); torch.manual_seed(0)
dev = torch.device("cuda")
model = PULPo(5, 4, 0.1, [S, S, S], feedback=FB, n0=32).to(dev).train()
stepper = dp.DataParallelStepper(model)
x, y = synthetic.uniform_pair([S, S, S], 1, 1234, dev)
e = torch.empty((0,), device=dev)
batch = (x, y, e, e, e, e, e, e)
for _ in range(3): stepper.step(batch)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
N = 10
orig = model.training_step
def timed(b, i):
    ev[0].record()
    out = orig(b, i)
    ev[1].record()
    return out
model.training_step = timed
for _ in range(N):
    stepper.step(batch)
    ev[2].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print(f"forward (incl. losses) {tf / N:.2f} ms, backward + optimizer {tb / N:.2f} ms (a synchronisation after every step)")
