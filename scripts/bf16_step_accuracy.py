"""How far is the bf16-operand training step from (a) its own definition evaluated in fp64, (b) the fp32 step?
Prints, per parameter, relative-L2 distances between four gradients: GPU bf16 mode, oracle bf16 mode in fp64 and in fp32,
and the oracle's fp32-mode fp64 run (ground truth of the reference's arithmetic).  n0 = 32, 32^3, T3/L2 (BASELINE config 1 shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import pulpo_oracle as O
import src.models as models, src.network_blocks as nb
from pulpo_amd import ops
from pulpo_amd._lib import lib

def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))

lib.load()
cfg = O.Cfg(3, 2, [32, 32, 32], n0=32)
sd = O.init_state_dict(cfg, seed=1)
gen = torch.Generator().manual_seed(9)
x, y = torch.rand(1, 1, 32, 32, 32, generator=gen), torch.rand(1, 1, 32, 32, 32, generator=gen)
eps = {0: torch.randn(1, 3, 16, 16, 16, generator=gen), 1: torch.randn(1, 3, 8, 8, 8, generator=gen)}
sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
e64 = {l: e.double() for l, e in eps.items()}
_, g_true, _ = O.train_step(O.clone_sd(sd64, requires_grad=True), cfg, x.double(), y.double(), e64)
O.CONV_PRECISION = "bf16"
_, g_def64, _ = O.train_step(O.clone_sd(sd64, requires_grad=True), cfg, x.double(), y.double(), e64)
_, g_def32, _ = O.train_step(O.clone_sd(sd, requires_grad=True), cfg, x, y, eps)
O.CONV_PRECISION = "fp32"
model = models.PULPo(3, 2, 0.1, [32, 32, 32], feedback=list(O.FEEDBACK_DEFAULT), n0=32)
model.load_state_dict({k: v.clone() for k, v in sd.items()})
model = model.cuda().train()
for l in range(2):
    model.autoencoder.encoders[l].sampler = nb.FixedNoiseSampler(eps[l].cuda())
ops.set_conv_precision("bf16")
outs, _, (total, kl, rec, reg), _ = model._forward_and_losses(x.cuda(), y.cuda())
total.backward()
print(f"{'parameter':70s} gpu-def64  def32-def64  gpu-true  def64-true")
rows = []
for k, p in model.named_parameters():
    if g_true.get(k) is None or p.grad is None or (k.endswith("_op.0.bias") and "velocity_field._op.2" not in k):
        continue
    r = (rel(p.grad, g_def64[k]), rel(g_def32[k], g_def64[k]), rel(p.grad, g_true[k]), rel(g_def64[k], g_true[k]))
    rows.append(r)
    if k.endswith("weight") and "_op.0.weight" in k:
        print(f"{k:70s} {r[0]:.2e}   {r[1]:.2e}   {r[2]:.2e}   {r[3]:.2e}")
import numpy as np
a = np.array(rows)
print("max     ", a.max(0)); print("median  ", np.median(a, 0))
