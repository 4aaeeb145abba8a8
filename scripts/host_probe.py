"""Where does the host spend a training step once the launch queue has filled (steady state)?  Wall-clock stamps around the stepper's phases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulpo_amd import dp, ops, synthetic
from pulpo_amd._lib import lib
from src.models import PULPo
FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
lib.load(); torch.manual_seed(0)
dev = torch.device("cuda")
model = PULPo(5, 4, 0.1, [160] * 3, feedback=FB, n0=32).to(dev).train()
st = dp.DataParallelStepper(model)
x, y = synthetic.uniform_pair([160] * 3, 1, 1234, dev)
e = torch.empty((0,), device=dev)
batch = (x, y, e, e, e, e, e, e)
for _ in range(3): st.step(batch)
torch.cuda.synchronize()
rows = []
for i in range(14):
    t0 = time.perf_counter()
    st.arena.zero_grad(); st._works, st._launched = [], 0
    loss = model.training_step(batch, 0)
    t1 = time.perf_counter()
    ops.DIRECT_PARAM_GRADS = True; ops.ASYNC_WGRAD_STREAM = st._side
    loss.backward()
    t2 = time.perf_counter()
    ops.DIRECT_PARAM_GRADS = False; ops.join_async_wgrad(); ops.ASYNC_WGRAD_STREAM = None
    st.opt.step(1.0)
    t3 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2))
torch.cuda.synchronize()
for i, r in enumerate(rows):
    print(f"step {i:2d}: host forward {r[0]*1e3:6.1f} ms  backward {r[1]*1e3:6.1f} ms  join+adam {r[2]*1e3:5.1f} ms")
