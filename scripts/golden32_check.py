"""gradient distance of the GPU step from the REAL reference's 32^3 golden step, per convolution algorithm"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import src.models as models, src.network_blocks as nb
from pulpo_amd import ops
from pulpo_amd._lib import lib
lib.load()
g = dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "step_T3L2_n8_32.npz")))
T = torch.from_numpy
FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))
for algo in ("direct", "wino2"):
    for wg in ("1", "0"):
        Tl, L, n0, B, *size = [int(v) for v in g["cfg"]]
        model = models.PULPo(Tl, L, 0.1, size, feedback=FB, n0=n0)
        sd = model.state_dict()
        for k, v in g.items():
            if k.startswith("sd0."): sd[k[4:]] = T(v.copy())
        model.load_state_dict(sd); model = model.cuda().train()
        for l in range(L):
            model.autoencoder.encoders[l].sampler = nb.FixedNoiseSampler(T(g[f"eps.{l}"]).cuda())
        ops.CONV_ALGO = algo
        outs, _, (total, kl, rec, reg), _ = model._forward_and_losses(T(g["x"]).cuda(), T(g["y"]).cuda())
        total.backward()
        errs = {k: rel(p.grad, g["grad." + k]) for k, p in model.named_parameters() if "grad." + k in g and not (k.endswith("_op.0.bias") and "velocity_field._op.2" not in k)}
        worst = max(errs, key=errs.get)
        print(algo, "total", float(total), float(g["train.total"]), "worst grad", worst, f"{errs[worst]:.2e}", "median", f"{np.median(list(errs.values())):.2e}")
        break

# who is closer to the fp64 evaluation of the same arithmetic: this implementation or the reference's own fp32 run?
from oracle import pulpo_oracle as O
Tl, L, n0, B, *size = [int(v) for v in g["cfg"]]
cfg = O.Cfg(Tl, L, size, n0=n0)
sd64 = {k[4:]: (T(v.copy()).double() if v.dtype.kind == "f" else T(v.copy())) for k, v in g.items() if k.startswith("sd0.")}
eps64 = {l: T(g[f"eps.{l}"]).double() for l in range(L)}
_, g64, _ = O.train_step(O.clone_sd(sd64, requires_grad=True), cfg, T(g["x"]).double(), T(g["y"]).double(), eps64)
ops.CONV_ALGO = None
rows = []
for k, p in model.named_parameters():
    if g64.get(k) is None or "grad." + k not in g or (k.endswith("_op.0.bias") and "velocity_field._op.2" not in k):
        continue
    rows.append((rel(p.grad, g64[k]), rel(T(g["grad." + k]), g64[k]), rel(p.grad, g["grad." + k])))
a = np.array(rows)
print("rel-L2 vs fp64:  this (wino2) median %.2e max %.2e | reference fp32 median %.2e max %.2e | this vs reference median %.2e" % (np.median(a[:, 0]), a[:, 0].max(), np.median(a[:, 1]), a[:, 1].max(), np.median(a[:, 2])))
