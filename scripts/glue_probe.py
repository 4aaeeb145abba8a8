"""Which torch (ATen) operators still launch kernels inside a training step, and from where (the glue around the HIP operators)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pulpo_amd import dp, synthetic
from pulpo_amd._lib import lib
from src.models import PULPo
FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib.load(); torch.manual_seed(0)
dev = torch.device("cuda")
model = PULPo(5, 4, 0.1, [S, S, S], feedback=FB, n0=32).to(dev).train()
stepper = dp.DataParallelStepper(model)
x, y = synthetic.uniform_pair([S, S, S], 1, 1234, dev)
e = torch.empty((0,), device=dev)
batch = (x, y, e, e, e, e, e, e)
for _ in range(3): stepper.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    stepper.step(batch)
    torch.cuda.synchronize()
rows = [ev for ev in prof.key_averages(group_by_stack_n=6) if ev.key.startswith("aten::") and ev.device_time_total > 0]
rows.sort(key=lambda ev: -ev.count)
for ev in rows[:45]:
    st = [s for s in ev.stack if "/repo/" in s][:3]
    print(f"{ev.count:4d} x {ev.key:28s} dev {ev.device_time_total:8.0f}us  shapes {str(ev.input_shapes)[:60]:60s} {' <- '.join(s.split('/repo/')[-1][:60] for s in st)}")
