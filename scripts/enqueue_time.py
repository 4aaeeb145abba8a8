"""host enqueue time per step against wall time per step (is the step GPU-bound?).
usage: python scripts/enqueue_time.py [repo root] [fp32 | bf16 | bf16act] [D H W T L]"""
import time, torch, sys
sys.path.insert(0, '/root/repo' if len(sys.argv) < 2 else sys.argv[1])
from pulpo_amd import dp, ops, synthetic
from pulpo_amd._lib import lib
from src.models import PULPo
lib.load()
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
if mode != "fp32":
    ops.set_conv_precision("bf16", activations="bf16" if mode == "bf16act" else "fp32")
FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
torch.manual_seed(0)
size, (T, L) = ([int(v) for v in sys.argv[3:6]], [int(v) for v in sys.argv[6:8]]) if len(sys.argv) > 7 else ([160] * 3, (5, 4))
m = PULPo(T, L, 0.1, size, feedback=FB, n0=32).cuda().train()
st = dp.DataParallelStepper(m)
x, y = synthetic.uniform_pair(size, 1, 1, "cuda")
e = torch.empty((0,), device="cuda")
b = (x, y, e, e, e, e, e, e)
for _ in range(3): st.step(b)
torch.cuda.synchronize()
import gc, os
if os.environ.get("PULPO_GC", "1") == "0":
    gc.collect(); gc.freeze(); gc.disable()
gc.callbacks.append(lambda phase, info: print(f"   [gc {phase} gen {info['generation']}]") if phase == "start" and info["generation"] == 2 else None)
enq = []
t00 = time.perf_counter()
for _ in range(8):
    t0 = time.perf_counter(); st.step(b); enq.append(time.perf_counter() - t0)
torch.cuda.synchronize()
tot = time.perf_counter() - t00
print(mode, "enqueue ms/step", [round(v*1e3,1) for v in enq], "wall ms/step", tot/8*1e3)
# forward / backward split with one synchronisation each
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.zero_grad(); loss = m.training_step(b, 0); th = time.perf_counter() - t0
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st.backward(loss); tb = time.perf_counter() - t1
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st.reduce_and_update(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"  forward: host {th*1e3:.1f} ms, done after {(t1-t0)*1e3:.1f} ms; backward: host {tb*1e3:.1f} ms, done after {(t2-t1)*1e3:.1f} ms; update {(t3-t2)*1e3:.1f} ms")
