import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from pulpo_amd import ops
from oracle import pulpo_oracle as O
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
gen = torch.Generator().manual_seed(0)
def report(name, fn_gpu, fn_ref, inputs, up):
    """inputs: list of cpu fp32 tensors requiring grad"""
    res = {}
    for tag, dt in (('f64', torch.float64), ('f32', torch.float32)):
        ins = [t.detach().to(dt).clone().requires_grad_(True) for t in inputs]
        out = fn_ref(*ins)
        gr = torch.autograd.grad((out * up.to(dt)).sum(), ins)
        res[tag] = (out, gr)
    ins = [t.detach().cuda().requires_grad_(True) for t in inputs]
    out = fn_gpu(*ins)
    gr = torch.autograd.grad((out * up.cuda()).sum(), ins)
    line = f"{name:28s} out gpu {rel(out, res['f64'][0]):.1e} cpu32 {rel(res['f32'][0], res['f64'][0]):.1e} |"
    for i in range(len(inputs)):
        line += f" g{i} gpu {rel(gr[i], res['f64'][1][i]):.1e} cpu32 {rel(res['f32'][1][i], res['f64'][1][i]):.1e}"
    print(line)
S = (8, 8, 8)
fine = (16, 16, 16)
srcs = [torch.randn(1, c, *S, generator=gen) for c in (3, 3, 3, 3, 3, 1)]
report('feedback_up2', lambda *s: ops.feedback_up2(list(s)), lambda *s: torch.cat([O.resize_to(t, fine) for t in s], 1), srcs, torch.randn(1, 16, *fine, generator=gen))
a, b = torch.randn(1, 3, *S, generator=gen), torch.randn(1, 3, *fine, generator=gen)
report('resize x2 mult2 + add', lambda x, y: ops.resize_trilinear(x, fine, 2.0, y), lambda x, y: O.resize_field(x, 0.5) + y, [a, b], torch.randn(1, 3, *fine, generator=gen))
v = torch.randn(1, 3, *S, generator=gen) * 2
report('vecint 8^3', lambda x: ops.vecint(x, 7), lambda x: O.vecint(x, 7), [v], torch.randn(1, 3, *S, generator=gen))
v = torch.randn(1, 3, *fine, generator=gen) * 2
report('vecint 16^3', lambda x: ops.vecint(x, 7), lambda x: O.vecint(x, 7), [v], torch.randn(1, 3, *fine, generator=gen))
df, img = torch.randn(1, 3, *S, generator=gen) * 1.5, torch.rand(1, 1, *S, generator=gen)
report('warp C=1', ops.warp3d, O.warp, [df, img], torch.randn(1, 1, *S, generator=gen))
p, t = torch.rand(1, 1, *S, generator=gen), torch.rand(1, 1, *S, generator=gen)
one = torch.ones(())
report('ncc w3 8^3', lambda x: ops.ncc_loss(x, t.cuda(), 3, 0.05), lambda x: O.ncc(x, t.to(x.dtype), 3, 0.05), [p], one)
p, t = torch.rand(1, 1, *fine, generator=gen), torch.rand(1, 1, *fine, generator=gen)
report('ncc w5 16^3', lambda x: ops.ncc_loss(x, t.cuda(), 5, 0.05), lambda x: O.ncc(x, t.to(x.dtype), 5, 0.05), [p], one)
p, t = torch.rand(1, 1, 32, 32, 32, generator=gen), torch.rand(1, 1, 32, 32, 32, generator=gen)
report('ncc w9 32^3', lambda x: ops.ncc_loss(x, t.cuda(), 9, 0.05), lambda x: O.ncc(x, t.to(x.dtype), 9, 0.05), [p], one)
d = torch.randn(1, 3, *S, generator=gen)
report('l2reg', lambda x: ops.l2_reg(x, 0.025), lambda x: O.l2_reg(x, 0.025), [d], one)
mu, sg = torch.randn(1, 3, *S, generator=gen), F.softplus(torch.randn(1, 3, *S, generator=gen))
report('kl', ops.kl_std_normal, lambda m, s: O.kl_diag(m, s), [mu, sg], one)
im = torch.rand(1, 1, *fine, generator=gen)
report('avgpool C=1', ops.avg_pool2, O.pool2, [im], torch.randn(1, 1, *S, generator=gen))
x = torch.randn(1, 64, *S, generator=gen)
report('avgpool C=64', lambda t: ops.avg_pool2(t.contiguous(memory_format=torch.channels_last_3d)), O.pool2, [x], torch.randn(1, 64, 4, 4, 4, generator=gen))
h = torch.randn(1, 64, *S, generator=gen)
wm, bm, wsg, bsg = torch.randn(3, 64, 1, 1, 1, generator=gen) * 0.1, torch.randn(3, generator=gen), torch.randn(3, 64, 1, 1, 1, generator=gen) * 0.1, torch.randn(3, generator=gen)
ep = torch.randn(1, 3, *S, generator=gen)
def ms_gpu(h, wm, bm, wsg, bsg):
    mu, sg, z = ops.mu_sigma_sample(h.contiguous(memory_format=torch.channels_last_3d), wm, bm, wsg, bsg, ep.cuda())
    return torch.cat([mu, sg, z], 1)
def ms_ref(h, wm, bm, wsg, bsg):
    mu = F.conv3d(h, wm, bm); sg = F.softplus(F.conv3d(h, wsg, bsg)); return torch.cat([mu, sg, mu + sg * ep.to(h.dtype)], 1)
report('mu_sigma', ms_gpu, ms_ref, [h, wm, bm, wsg, bsg], torch.randn(1, 9, *S, generator=gen))
h = torch.randn(1, 32, *S, generator=gen)
w, b = torch.randn(3, 32, 1, 1, 1, generator=gen) * 0.1, torch.randn(3, generator=gen)
report('conv1x1', lambda h, w, b: ops.conv1x1_to3(h.contiguous(memory_format=torch.channels_last_3d), w, b), lambda h, w, b: F.conv3d(h, w, b), [h, w, b], torch.randn(1, 3, *S, generator=gen))
x = torch.randn(1, 3, *S, generator=gen)
w = torch.randn(32, 3, 3, 3, 3, generator=gen) * 0.1
report('conv 3->32 planar', lambda x, w: ops.conv3d_k3(x, w), lambda x, w: F.conv3d(x, w, padding=1), [x, w], torch.randn(1, 32, *S, generator=gen))
x = torch.randn(1, 160, *S, generator=gen)
w = torch.randn(64, 160, 3, 3, 3, generator=gen) * 0.02
report('conv 160->64 cat', lambda x, w: ops.conv3d_k3(torch.cat([x[:, :96].contiguous(memory_format=torch.channels_last_3d), x[:, 96:].contiguous(memory_format=torch.channels_last_3d)], 1), w), lambda x, w: F.conv3d(x, w, padding=1), [x, w], torch.randn(1, 64, *S, generator=gen))
