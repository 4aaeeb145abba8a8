"""Every ATen operator call on large GPU tensors inside one training step, with shapes, strides and the Python frames that issued it
(TorchDispatchMode; calls from the autograd engine have no Python frames of ours - their shapes identify them).
usage: python scripts/aten_audit.py [size=160] [min elements=1e5]"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from pulpo_amd import dp, synthetic
from pulpo_amd._lib import lib
from src.models import PULPo
FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 160
MIN = float(sys.argv[2]) if len(sys.argv) > 2 else 1e5
lib.load(); torch.manual_seed(0)
dev = torch.device("cuda")
model = PULPo(5, 4, 0.1, [S, S, S], feedback=FB, n0=32).to(dev).train()
stepper = dp.DataParallelStepper(model)
x, y = synthetic.uniform_pair([S, S, S], 1, 1234, dev)
e = torch.empty((0,), device=dev)
batch = (x, y, e, e, e, e, e, e)
for _ in range(2): stepper.step(batch)
torch.cuda.synchronize()
log = collections.Counter()
SKIP = ("aten.empty", "aten.view", "aten._unsafe_view", "aten.as_strided", "aten.detach", "aten.alias", "aten.t.", "aten.permute", "aten.select",
        "aten.slice", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.reshape", "aten.transpose", "aten.new_empty", "aten.empty_like",
        "aten._local_scalar", "aten.lift_fresh", "aten.is_", "aten.stride", "aten.size", "aten.narrow", "aten.unbind", "aten.split", "aten.chunk")
class Audit(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if name.startswith(("aten._to_copy", "aten.copy_", "aten._local_scalar_dense", "aten.item")):
            # host <-> device traffic: every one of these is a blit kernel (or a synchronisation) inside the step
            allt = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
            devs = {str(t.device.type) for t in allt} | ({str(kwargs["device"]).split(":")[0]} if kwargs and kwargs.get("device") is not None else set())
            if len(devs) > 1 or name.startswith(("aten._local_scalar_dense", "aten.item")):
                fr = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack() if ("pulpo_amd" in f.filename or "/src/" in f.filename)][-3:]
                log[("H<->D " + name, ", ".join(f"{tuple(t.shape)}@{t.device.type}" for t in allt[:2]), " <- ".join(reversed(fr)))] += 1
        if not name.startswith(SKIP):
            ts = [a for a in args if isinstance(a, torch.Tensor) and a.is_cuda]
            for a in args:
                if isinstance(a, (list, tuple)): ts += [b for b in a if isinstance(b, torch.Tensor) and b.is_cuda]
            big = [t for t in ts if t.numel() >= MIN]
            if big:
                desc = ", ".join(f"{tuple(t.shape)}/{tuple(t.stride())}" for t in ts[:3])
                fr = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack() if ("pulpo_amd" in f.filename or "/src/" in f.filename)][-3:]
                log[(name, desc, " <- ".join(reversed(fr)))] += 1
        return func(*args, **(kwargs or {}))
with Audit():
    stepper.step(batch)
torch.cuda.synchronize()
for (name, desc, fr), c in sorted(log.items(), key=lambda kv: -kv[1]):
    print(f"{c:3d} x {name:28s} {desc:110s} {fr}")
