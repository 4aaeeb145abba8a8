import time, torch, sys
sys.path.insert(0, '/root/repo' if len(sys.argv) < 2 else sys.argv[1])
from pulpo_amd import dp, ops, synthetic
from pulpo_amd._lib import lib
from src.models import PULPo
lib.load()
FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
torch.manual_seed(0)
m = PULPo(5, 4, 0.1, [160]*3, feedback=FB, n0=32).cuda().train()
st = dp.DataParallelStepper(m)
x, y = synthetic.uniform_pair([160]*3, 1, 1, "cuda")
e = torch.empty((0,), device="cuda")
b = (x, y, e, e, e, e, e, e)
for _ in range(3): st.step(b)
torch.cuda.synchronize()
enq = []
t00 = time.perf_counter()
for _ in range(8):
    t0 = time.perf_counter(); st.step(b); enq.append(time.perf_counter() - t0)
torch.cuda.synchronize()
tot = time.perf_counter() - t00
print("enqueue ms/step", [round(v*1e3,1) for v in enq], "wall ms/step", tot/8*1e3)
