"""debug: gradients of a replayed HIP-graph step against the eager step on the same weights and batch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pulpo_amd import dp, ops
from pulpo_amd._lib import lib
import src.models as models, src.network_blocks as nb
lib.load()
FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
size, Tl, L, n0 = [32, 32, 32], 3, 2, 8
gen = torch.Generator().manual_seed(3)
batches = [tuple([torch.rand(1, 1, *size, generator=gen).cuda() for _ in range(2)] + [torch.empty((0,), device="cuda")] * 6) for _ in range(2)]
eps = [torch.randn(1, 3, *[s_ // 2 ** (l + 1) for s_ in size], generator=gen).cuda() for l in range(L)]
torch.manual_seed(0)
model = models.PULPo(Tl, L, 0.1, size, feedback=FB, n0=n0).cuda().train()
for l in range(L):
    model.autoencoder.encoders[l].sampler = nb.FixedNoiseSampler(eps[l])
st = dp.DataParallelStepper(model, graph=True)
for i in range(3):
    print("step", i, float(st.step(batches[i % 2])))
g, static, loss, levels, sig = st._graph
for trial in range(2):
    b = batches[(trial + 1) % 2]
    for dst, src in zip(static, b):
        if dst.numel(): dst.copy_(src)
    g.replay(); torch.cuda.synchronize()
    G = st.arena.grad.clone(); lg = float(loss)
    st.zero_grad(); l2 = model.training_step(b, 0); st.backward(l2); torch.cuda.synchronize()
    E = st.arena.grad.clone()
    print("trial", trial, "loss graph", lg, "eager", float(l2), "grad rel diff", float((G - E).norm() / E.norm()))
    worst = []
    for (n, p), o in zip([(n, p) for n, p in model.named_parameters() if p.requires_grad], []):
        pass
    names = {id(p): n for n, p in model.named_parameters()}
    for p, o in zip(st.arena.params, st.arena.offsets):
        a, e = G[o:o + p.numel()], E[o:o + p.numel()]
        d = float((a - e).norm() / (e.norm() + 1e-30))
        if d > 1e-4: worst.append((d, names[id(p)]))
    print("  params off:", len(worst), sorted(worst, reverse=True)[:12])
