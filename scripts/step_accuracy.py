import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import pulpo_oracle as O
import src.models as models, src.network_blocks as nb
FB = list(O.FEEDBACK_DEFAULT)
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
cfg = O.Cfg(3, 2, [32, 32, 32], n0=32)
sd = O.init_state_dict(cfg, seed=1)
gen = torch.Generator().manual_seed(9)
x, y = torch.rand(1, 1, 32, 32, 32, generator=gen), torch.rand(1, 1, 32, 32, 32, generator=gen)
eps = {0: torch.randn(1, 3, 16, 16, 16, generator=gen), 1: torch.randn(1, 3, 8, 8, 8, generator=gen)}
orig_unit = O.conv_unit
rec = {}
def patched(h, sd_, prefix, training):
    if h.requires_grad: h.retain_grad()
    rec[cur][prefix + ':in'] = h
    out = orig_unit(h, sd_, prefix, training)
    out.retain_grad(); rec[cur][prefix] = out
    return out
O.conv_unit = patched
def run_oracle(dt, tag):
    global cur
    cur = tag; rec[tag] = {}
    s = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    s = O.clone_sd(s, requires_grad=True)
    outs = O.forward(s, cfg, x.to(dt), y.to(dt), {l: e.to(dt) for l, e in eps.items()}, True)
    O.losses(outs, y.to(dt), cfg)[0].backward()
run_oracle(torch.float64, 'f64'); run_oracle(torch.float32, 'f32')
model = models.PULPo(3, 2, 0.1, [32, 32, 32], feedback=FB, n0=32)
model.load_state_dict({k: v.clone() for k, v in sd.items()})
model = model.cuda().train()
for l in range(2):
    model.autoencoder.encoders[l].sampler = nb.FixedNoiseSampler(eps[l].cuda())
rec['gpu'] = {}
def mk(name):
    def pre(mod, inp):
        if inp[0].requires_grad: inp[0].retain_grad()
        rec['gpu'][name + ':in'] = inp[0]
    def post(mod, inp, out):
        out.retain_grad(); rec['gpu'][name] = out
    return pre, post
for name, mod in model.named_modules():
    if isinstance(mod, nb.ConvUnit):
        pre, post = mk(name)
        mod.register_forward_pre_hook(pre); mod.register_forward_hook(post)
empty = torch.empty((0,))
model.training_step((x.cuda(), y.cuda(), empty, empty, empty, empty, empty, empty), 0).backward()
for k in rec['f64']:
    a, b, c = rec['gpu'].get(k), rec['f32'][k], rec['f64'][k]
    if a is None: print('missing', k); continue
    gv = f"val gpu {rel(a, c):.1e} cpu32 {rel(b, c):.1e}"
    if a.grad is not None and c.grad is not None:
        gv += f" | grad gpu {rel(a.grad, c.grad):.1e} cpu32 {rel(b.grad, c.grad):.1e}"
        d = (a.grad.cpu().double() - c.grad)
        dims = (0, 2, 3, 4)
        gv += f" | DCerr/rms {float(d.mean(dim=dims).abs().mean() / (c.grad.pow(2).mean().sqrt()+1e-30)):.1e}"
    print(f"{k:62s} {gv}")


print("sign mismatches of ConvUnit outputs (LeakyReLU slope flips) vs the fp64 oracle:")
tot_g = tot_c = 0
for k in rec['f64']:
    if k.endswith(':in'): continue
    a, b, c = rec['gpu'][k].detach().cpu(), rec['f32'][k].detach(), rec['f64'][k].detach()
    fg = int(((a > 0) != (c > 0)).sum()); fc = int(((b > 0) != (c > 0)).sum())
    tot_g += fg; tot_c += fc
    if fg or fc: print(f"   {k:60s} gpu {fg}  cpu32 {fc}  of {c.numel()}")
print('total flips: gpu', tot_g, ' cpu32', tot_c)
