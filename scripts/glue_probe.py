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
def ours(s):
    return any(t in s for t in ("pulpo_amd/", "/src/", "bench.py", "scripts/"))
rows = [ev for ev in prof.key_averages(group_by_stack_n=12) if ev.key.startswith("aten::") and ev.device_time_total > 0]
rows.sort(key=lambda ev: -ev.count)
print(f"{sum(ev.count for ev in rows)} ATen operator calls with device time in one step")
for ev in rows[:70]:
    st = [s for s in ev.stack if ours(s)][:3]
    where = " <- ".join(s.split("pulpo_amd/")[-1].split("/src/")[-1][:70] for s in st)
    print(f"{ev.count:4d} x {ev.key:24s} dev {ev.device_time_total:7.0f}us  {str(ev.input_shapes)[:50]:50s} {where}")
# device activities by name (kernels, copies, fills - whoever launched them)
from collections import Counter
cnt, tim = Counter(), Counter()
for ev in prof.events():
    if str(ev.device_type).endswith("CUDA"):
        cnt[ev.name[:60]] += 1; tim[ev.name[:60]] += ev.device_time
print("device activities in one step:")
for name, c in cnt.most_common(60):
    print(f"{c:5d} x {name:60s} {tim[name]:9.0f} us")
# runtime calls of copies / fills with their Python frames are not attributed by the profiler: count them
rt = Counter(ev.name for ev in prof.events() if ev.name.startswith("hipMem"))
print("runtime:", dict(rt))
