"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header declares, the host mirror of
the reference API has the reference's state-dict inventory / hyper-parameter tables / error behaviour, the product
path refuses to run without a GPU (no CPU fallback), and the data-parallel plumbing works over gloo (world size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

FB = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]



def _free_port() -> str:
    """a TCP port that is free right now (the rendezvous of the multi-process tests)"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])

def test_library_exports_every_declared_symbol():
    from pulpo_amd._lib import HEADER, LIB_PATH, lib, parse_header
    protos = parse_header()
    declared = set(re.findall(r"\b(pulpo_\w+)\s*\(", re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)))
    assert declared == set(protos), declared ^ set(protos)
    assert len(protos) >= 36
    assert os.path.exists(LIB_PATH), "build the library first: python -m pulpo_amd.build"
    dll = ctypes.CDLL(LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in include/pulpo_hip.h but not exported"
    lib.load()
    from pulpo_amd._lib import header_abi_version
    assert lib.query("pulpo_abi_version") == header_abi_version() >= 3
    # pure host-side size queries (no device work)
    assert lib.query("pulpo_conv3d_k3_packed_floats", 32, 32) == 27 * 32 * 64
    assert lib.query("pulpo_conv3d_k3_packed_floats", 2, 32) == 27 * 2 * 64               # 2-channel chunks for the image-pair layer
    assert lib.query("pulpo_conv3d_k3_packed_floats", 3, 32) == 27 * 4 * 64
    assert lib.query("pulpo_conv3d_k3_stat_tiles", 1, 160, 160, 160) == 40 * 20 * 20      # 4x8x8 voxel tiles from 20^3 up (depth % 4 == 0)
    assert lib.query("pulpo_conv3d_k3_stat_tiles", 2, 32, 32, 32) == 2 * 8 * 4 * 4
    assert lib.query("pulpo_conv3d_k3_stat_tiles", 1, 10, 10, 10) == 5 * 2 * 2             # 2x8x8 otherwise
    assert lib.query("pulpo_conv3d_k3_packed_bf16_elems", 48, 32) == 2 * 27 * 64 * 32
    assert lib.query("pulpo_conv3d_k3_algo", 1, 40, 40, 40, 12, 128) == 2 and lib.query("pulpo_conv3d_k3_algo", 1, 10, 10, 10, 192, 192) == 2      # (10^3: split-K pipelined kernel)
    assert lib.query("pulpo_conv3d_k3_algo", 1, 40, 40, 40, 24, 128) == 3
    # F(2x2x2,3x3x3): whole 4x8x8 tiles, >= 16 reduction channels (multiple of 8), couts a multiple of 32, >= 256 work items
    assert lib.query("pulpo_conv3d_k3_algo", 1, 160, 160, 160, 32, 32) == 3 and lib.query("pulpo_conv3d_k3_algo", 1, 80, 80, 80, 16, 96) == 3
    assert lib.query("pulpo_conv3d_k3_algo", 1, 80, 80, 80, 8, 96) == 2
    assert lib.query("pulpo_conv3d_k3_algo", 1, 40, 40, 40, 64, 128) == 3 and lib.query("pulpo_conv3d_k3_algo", 1, 80, 80, 80, 96, 96) == 3
    assert lib.query("pulpo_conv3d_k3_algo", 1, 80, 80, 80, 96, 16) == 3 and lib.query("pulpo_conv3d_k3_algo", 1, 20, 20, 20, 192, 192) == 2     # (a half-empty cout tile)
    assert lib.query("pulpo_conv3d_k3_algo", 1, 80, 80, 80, 96, 12) == 2 and lib.query("pulpo_conv3d_k3_algo", 1, 80, 80, 80, 96, 18) == 2
    assert lib.query("pulpo_conv3d_k3_algo", 1, 44, 40, 40, 64, 64) == 3 and lib.query("pulpo_conv3d_k3_algo", 1, 40, 44, 40, 64, 64) == 2
    assert lib.query("pulpo_conv3d_k3_packed_wino3_floats", 64, 32) == 8 * 64 * 8 * 64
    assert lib.query("pulpo_conv3d_k3_algo", 1, 10, 10, 10, 20, 12) == 0 and lib.query("pulpo_conv3d_k3_algo", 1, 6, 6, 6, 192, 192) == 0
    assert lib.query("pulpo_conv3d_k3_algo", 1, 160, 160, 160, 2, 32) == 0                  # image input layers: direct kernel
    assert lib.query("pulpo_conv3d_k3_packed_wino2_floats", 20, 12) == 3 * 3 * 16 * 8 * 64             # 3 chunks of 8 channels x 3 dz x 16 points
    # the pipelined (y, x) kernel takes channels-last operands with a multiple of 8 channels and 32-bit halo offsets
    assert lib.query("pulpo_conv3d_k3_wino2_pipelined", 160, 160, 160, 32, 32) == 1 and lib.query("pulpo_conv3d_k3_wino2_pipelined", 64, 64, 64, 20, 20) == 0
    assert lib.query("pulpo_conv3d_k3_wino2_pipelined", 1024, 1024, 1024, 32, 32) == 0
    assert lib.query("pulpo_conv3d_k3_wgrad_scratch_floats", 160, 64) == 27 * 160 * 64


def test_bad_arguments_return_error_codes_not_crashes():
    from pulpo_amd._lib import lib
    lib.load()
    rc = lib.raw("pulpo_conv3d_k3_fwd")(None, 0, 0, 0, None, None, None, 0, 0, 0, None, None, 1, 8, 8, 8, 4, 4, None)
    assert rc != 0 and b"null pointer" in lib.raw("pulpo_last_error")()
    rc = lib.raw("pulpo_ncc_fwd")(None, None, None, None, None, 1, 8, 8, 8, 4, None)
    assert rc != 0


def test_state_dict_inventory_matches_reference():
    from src.models import PULPo
    want = {}
    for line in open(os.path.join(GOLDEN, "state_keys.txt")):
        if line.startswith("#"):
            continue
        tl, key, rest = line.split(" ", 2)
        shape, dt = rest.rsplit(" ", 1)
        want.setdefault(tl, []).append((key, eval(shape), dt.strip()))
    for tl, (Tl, L) in {"3/2": (3, 2), "5/4": (5, 4)}.items():
        model = PULPo(Tl, L, 0.1, [32, 32, 32], feedback=FB, n0=32)
        sd = model.state_dict()
        assert [k for k, _, _ in want[tl]] == list(sd.keys())          # same keys in the same order
        for k, shape, dt in want[tl]:
            assert tuple(sd[k].shape) == shape and str(sd[k].dtype).replace("torch.", "") == dt, k


def test_hparams_tables_and_errors(golden):
    from src.models import PULPo
    from src.components.pulpo import Autoencoder
    from src.network_blocks import gauss_sampler
    g = golden("init_tables")
    for key, tab in g.items():
        Tl, L = int(key[1]), int(key[3])
        m = PULPo(Tl, L, 0.1, [32, 32, 32], feedback=FB, n0=2)
        got = np.array([[m.window_size[l], m.hierarchical_kl_loss.weight_dict[l], m.hierarchical_recon_loss.weight_dict[l],
                         m.hierarchical_regularization.weight_dict[l]] for l in range(L)])
        np.testing.assert_array_equal(got, tab)
        assert m.lk_offset == Tl - L and m.ndims == 3 and m.hparams.lr == 1e-4 and m.hparams.beta == 0.1
    with pytest.raises(ValueError, match="velocity_field"):       # the shipped default list does not construct (Appendix A.1)
        PULPo(3, 2, 0.1, [16, 16, 16], n0=2)
    with pytest.raises(ValueError, match="regularizer"):
        PULPo(3, 2, 0.1, [16, 16, 16], feedback=FB, n0=2, regularizer="tv")
    with pytest.raises(ValueError, match="Decoder"):
        Autoencoder(gauss_sampler, "bspline", 3, 2, 3, [16, 16, 16], FB, "level_res", 2, 3)
    m = PULPo(3, 2, 0.1, [16, 16, 16], feedback=["samples", "control_points"], n0=2)    # old alias is accepted
    assert m.autoencoder.up_blocks[1]._op[0]._op[0].in_channels == 6


def test_module_int_dict():
    import torch.nn as nn
    from src.utils import ModuleIntDict
    d = ModuleIntDict({0: nn.Linear(1, 1)})
    d[3] = nn.Linear(2, 2)
    assert list(d.keys()) == [0, 3] and 3 in d and d[3].in_features == 2
    assert [k for k, _ in d.items()] == [0, 3]
    holder = nn.Module()
    holder.blocks = d
    assert list(holder.state_dict().keys()) == ["blocks.0.weight", "blocks.0.bias", "blocks.3.weight", "blocks.3.bias"]


def test_product_path_has_no_cpu_fallback():
    from pulpo_amd._lib import PulpoHipError
    from src.models import PULPo
    m = PULPo(3, 2, 0.1, [16, 16, 16], feedback=FB, n0=2)
    x = torch.rand(1, 1, 16, 16, 16)
    with pytest.raises(PulpoHipError, match="GPU only"):
        m(x, x)
    # nothing under pulpo_amd/ or src/ imports the oracle
    for base in ("pulpo_amd", "src"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(".py"):
                    text = open(os.path.join(dirpath, f)).read()
                    assert "oracle" not in text.replace("# oracle", ""), os.path.join(dirpath, f)


def test_flat_arena_views_and_zero_grad():
    from pulpo_amd.dp import FlatArena
    lin = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.Linear(5, 2))
    ref = [p.detach().clone() for p in lin.parameters()]
    arena = FlatArena(lin)
    for p, r in zip(lin.parameters(), ref):
        assert torch.equal(p.detach(), r)
        assert p.data_ptr() >= arena.data.data_ptr() and p.grad.data_ptr() >= arena.grad.data_ptr()
    lin(torch.ones(4, 3)).sum().backward()
    assert float(arena.grad.abs().sum()) > 0          # autograd accumulated straight into the arena
    g0 = arena.grad.clone()
    lin(torch.ones(4, 3)).sum().backward()
    assert torch.allclose(arena.grad, 2 * g0)
    arena.zero_grad()
    assert float(arena.grad.abs().sum()) == 0 and all(float(p.grad.abs().sum()) == 0 for p in lin.parameters())


def test_fused_adam_state_dict_is_torch_adam_layout_whatever_the_arena_order():
    """FusedAdam.state_dict() is torch.optim.Adam's layout (per parameter, module.parameters() order), independent of the arena's internal
    (bucket) order: round trip through torch.optim.Adam and into an arena with a different permutation"""
    import pytest
    from pulpo_amd.dp import FlatArena, FusedAdam
    torch.manual_seed(0)

    def net():
        return torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.Linear(5, 2), torch.nn.Linear(2, 7))

    a = net()
    pa = list(a.parameters())
    arena = FlatArena(a, params=pa[::-1])                     # permuted arena, like the completion-ordered buckets
    opt = FusedAdam(arena, lr=3e-4)
    opt.m.copy_(torch.randn(arena.numel)); opt.v.copy_(torch.rand(arena.numel)); opt.t = 5
    sd = opt.state_dict()
    assert sd["param_names"] == [n for n, _ in a.named_parameters()]
    for i, p in enumerate(pa):
        o = arena.offsets[arena.params.index(p)] if False else arena.offsets[[id(q) for q in arena.params].index(id(p))]
        assert torch.equal(sd["state"][i]["exp_avg"], opt.m[o:o + p.numel()].view_as(p))
        assert float(sd["state"][i]["step"]) == 5.0
    topt = torch.optim.Adam(a.parameters(), lr=1.0)
    topt.load_state_dict({k: v for k, v in sd.items() if k != "param_names"})
    assert topt.param_groups[0]["lr"] == 3e-4
    back = topt.state_dict()
    b = net()
    pb = list(b.parameters())
    arena_b = FlatArena(b, params=[pb[2], pb[0], pb[1], pb[5], pb[3], pb[4]])
    opt_b = FusedAdam(arena_b)
    opt_b.load_state_dict(back)
    assert opt_b.t == 5 and opt_b.lr == 3e-4
    sd_b = opt_b.state_dict()
    for i in range(len(pa)):
        assert torch.equal(sd_b["state"][i]["exp_avg"], sd["state"][i]["exp_avg"]) and torch.equal(sd_b["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
    # by name: a checkpoint whose parameter order differs is scattered correctly; wrong shapes / counts are refused
    perm = [3, 0, 5, 1, 4, 2]
    shuffled = {"state": {j: sd["state"][i] for j, i in enumerate(perm)}, "param_groups": [dict(sd["param_groups"][0])],
                "param_names": [sd["param_names"][i] for i in perm]}
    opt_c = FusedAdam(FlatArena(net()))
    opt_c.load_state_dict(shuffled)
    assert all(torch.equal(opt_c.state_dict()["state"][i]["exp_avg"], sd["state"][i]["exp_avg"]) for i in range(len(pa)))
    bad = {"state": {i: dict(st) for i, st in sd["state"].items()}, "param_groups": sd["param_groups"]}
    bad["state"][0]["exp_avg"] = torch.zeros(4, 4)
    with pytest.raises(ValueError):
        FusedAdam(FlatArena(net())).load_state_dict(bad)
    with pytest.raises(ValueError):
        FusedAdam(FlatArena(net())).load_state_dict({"step": 1, "numel": 10})
    # a checkpoint that fails validation part-way (a wrong shape in the LAST entry, differing step counts) leaves the optimizer as it was
    opt_d = FusedAdam(FlatArena(net()), lr=7e-4)
    opt_d.m.fill_(0.25); opt_d.v.fill_(0.5); opt_d.t = 9
    bad_last = {"state": {i: dict(st) for i, st in sd["state"].items()}, "param_groups": sd["param_groups"]}
    bad_last["state"][len(pa) - 1]["exp_avg_sq"] = torch.zeros(3)
    bad_steps = {"state": {i: dict(st) for i, st in sd["state"].items()}, "param_groups": sd["param_groups"]}
    bad_steps["state"][2]["step"] = torch.tensor(6.0)
    for broken in (bad_last, bad_steps):
        with pytest.raises(ValueError):
            opt_d.load_state_dict(broken)
        assert bool((opt_d.m == 0.25).all()) and bool((opt_d.v == 0.5).all()) and opt_d.t == 9 and opt_d.lr == 7e-4


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pulpo_amd.dp import FlatArena, allreduce_sum_, init_from_env, world
init_from_env("gloo")
rank = dist.get_rank()
torch.manual_seed(100 + rank)                       # different initial weights per rank on purpose
net = torch.nn.Sequential(torch.nn.Linear(4, 6), torch.nn.Linear(6, 3))
arena = FlatArena(net)
dist.broadcast(arena.data, src=0)                   # what DataParallelStepper does at start-up
w0 = arena.data.clone()
x = torch.full((2, 4), float(rank + 1))             # rank-dependent shard of the batch
arena.zero_grad()
net(x).sum().backward()
local = arena.grad.clone()
allreduce_sum_(arena.grad)
gathered = [torch.zeros_like(local) for _ in range(world())]
dist.all_gather(gathered, local)
assert torch.allclose(arena.grad, sum(gathered)), "all-reduce != sum of per-rank gradients"
wl = [torch.zeros_like(w0) for _ in range(world())]
dist.all_gather(wl, w0)
assert all(torch.equal(w, wl[0]) for w in wl), "weights differ after broadcast"
print(f"rank {rank} ok world {world()}")
dist.destroy_process_group()
'''


def test_data_parallel_plumbing_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok world 2" in o


BUCKET_WORKER = r'''
import os, sys, types, torch, torch.nn as nn, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pulpo_amd import dp
from pulpo_amd.utils import ModuleIntDict
dp.init_from_env("gloo")
rank = dist.get_rank()

class Down(nn.Module):                       # PULPo's module structure (downpath.down_blocks[k], autoencoder) on plain CPU ops
    def __init__(self, T):
        super().__init__()
        self.total_levels = T
        self.down_blocks = ModuleIntDict()
        for k in range(T):
            self.down_blocks[k] = nn.Sequential(nn.Linear(8, 8), nn.Tanh())
    def forward(self, h):
        acts = {}
        for k in range(self.total_levels):
            h = self.down_blocks[k](h)
            acts[k] = h
        return acts

class Net(nn.Module):
    def __init__(self, T=4):
        super().__init__()
        self.downpath = Down(T)
        self.autoencoder = nn.ModuleList([nn.Linear(8, 8) for _ in range(T)])
        self.extra = nn.Linear(8, 1)
        self.hparams = types.SimpleNamespace(lr=1e-3)
    def training_step(self, batch, idx):
        acts = self.downpath(batch)
        h = 0
        for k in reversed(range(len(self.autoencoder))):       # coarse to fine, every level feeds the finer ones
            h = torch.tanh(self.autoencoder[k](acts[k] + h))
        return self.extra(h).pow(2).mean()

torch.manual_seed(0)
net = Net()
launches = []
stepper = dp.DataParallelStepper(net, overlap=True)
assert len(stepper.buckets) == 3 and stepper.overlap
names = {id(p): n for n, p in net.named_parameters()}
order = [names[id(p)] for p in stepper.arena.params]
assert order[0].startswith("autoencoder.") and order[-1].startswith(("downpath.down_blocks.0", "downpath.down_blocks.1", "extra"))
stepper.opt.step = lambda scale: None        # the fused Adam is a GPU kernel; this test is about the gradient exchange
orig = stepper._launch_upto
def spy(i):
    launches.append((i, stepper._launched))
    return orig(i)
stepper._launch_upto = spy
x = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + rank))
# expected: sum over ranks of the local gradients, from a plain backward without the stepper
ref = Net(); ref.load_state_dict(net.state_dict())
ref.training_step(x, 0).backward()
exp = {n: p.grad.clone() for n, p in ref.named_parameters()}
for t in exp.values():
    dist.all_reduce(t)
stepper.step(x)
assert [i for i, _ in launches] == [0, 1, 2], launches      # two hooks during backward, the rest after it
for n, p in net.named_parameters():
    assert torch.allclose(p.grad, exp[n], rtol=1e-6, atol=1e-7), n
# the same pieces in Lightning's order (training_step -> optimizer_zero_grad -> backward -> optimizer.step): forward BEFORE the fill
launches.clear()
stepper.arm()
loss = net.training_step(x, 0)
stepper.zero_grad()
stepper.backward(loss)
stepper.reduce_and_update()
assert [i for i, _ in launches] == [0, 1, 2], launches
for n, p in net.named_parameters():
    assert torch.allclose(p.grad, exp[n], rtol=1e-6, atol=1e-7), n
# a backward pass that does NOT go through stepper.backward() (manual optimisation, a user's own loop) with the triggers armed: no bucket
# leaves during it, the whole arena is exchanged in one piece by reduce_and_update()
launches.clear()
stepper.arm()
loss = net.training_step(x, 0)
stepper.zero_grad()
loss.backward()
assert launches == [] and stepper._launched == 0
stepper.reduce_and_update()
for n, p in net.named_parameters():
    assert torch.allclose(p.grad, exp[n], rtol=1e-6, atol=1e-7), n
# ... and the step after it is a normal overlapped one again (no stale bucket state)
launches.clear()
stepper.step(x)
assert [i for i, _ in launches] == [0, 1, 2], launches
for n, p in net.named_parameters():
    assert torch.allclose(p.grad, exp[n], rtol=1e-6, atol=1e-7), n
# gradients a DistributedDataParallel wrapper has already averaged: no exchange, no 1/world scale
seen = []
stepper.opt.step = lambda scale: seen.append(scale)
before = stepper.arena.grad.clone()
stepper.reduce_and_update(reduced_elsewhere=True)
assert seen == [1.0] and torch.equal(stepper.arena.grad, before)
stepper.reduce_and_update()
assert seen[-1] == 0.5
print(f"rank {rank} buckets ok")
dist.destroy_process_group()
'''


def test_bucketed_overlapped_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "bucket_worker.py"
    script.write_text(BUCKET_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} buckets ok" in o


def test_2d_model_state_dict_matches_reference_fixture(golden):
    """train.py --ndims 2: the 2-D model builds on the CPU with the reference's parameter / buffer names and shapes (Conv2d / BatchNorm2d
    containers), taken from the state dict of the 2-D golden step generated from the real reference"""
    import src.models as models
    g = golden("step2d_T3L2_n4_32x24")
    Tl, L, n0, B, *size = [int(v) for v in g["cfg"]]
    model = models.PULPo(Tl, L, 0.1, size, feedback=["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"], n0=n0)
    sd = model.state_dict()
    ref = {k[4:]: v.shape for k, v in g.items() if k.startswith("sd0.")}
    assert set(ref) == {k for k in sd if not k.endswith(".grid")}
    for k, shp in ref.items():
        assert tuple(sd[k].shape) == shp, k
    assert model.autoencoder.encoders[0].mu_sigma._conv_mu.weight.shape[0] == 2          # zdim = ndims


def test_public_header_is_plain_c(tmp_path):
    """include/pulpo_hip.h is the drop-in boundary: it must compile as C99 and as C++ without any HIP / torch header"""
    import shutil
    src = tmp_path / "h.c"
    src.write_text('#include "pulpo_hip.h"\nint main(void) { return 0; }\n')
    inc = os.path.join(ROOT, "include")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, "-x", "c++", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_src_shim_extends_its_path_so_reference_subpackages_resolve(tmp_path):
    """INTEGRATION.md option 1: this repository's root AHEAD of the reference on PYTHONPATH.  `src.models` etc. must come from the shim
    while `src.data.*` (train.py:7-8, evaluate.py:19-20; not provided here) resolves from the reference's src/ further down the path.
    A stand-in tree plays the reference (its real dataset modules need h5py, absent from the image)."""
    ref = tmp_path / "ref"
    (ref / "src" / "data" / "OASIS").mkdir(parents=True)
    (ref / "src" / "__init__.py").write_text("")
    (ref / "src" / "models.py").write_text("raise ImportError('the reference models module must be shadowed by the shim')\n")
    (ref / "src" / "data" / "OASIS" / "oasis.py").write_text("class Oasis:\n    where = 'reference tree'\n")
    code = ("import src.models, src.losses, src.components.pulpo, src.components.utils, src.network_blocks, src.utils\n"
            "from src.data.OASIS import oasis\n"
            "import os\n"
            "assert oasis.Oasis.where == 'reference tree'\n"
            "assert 'pulpo_amd' in src.models.PULPo.__module__, src.models.PULPo.__module__\n"
            "assert callable(src.components.utils.warp_landmarks)\n"
            "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, str(ref)]))
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


def _bench(args, env_extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_bench_gpus_2_starts_its_own_ranks_without_torchrun():
    """`python bench.py --gpus 2` with NO torchrun environment must start two fresh rank processes itself (before any GPU call), rendezvous
    them on 127.0.0.1, and relay exactly one JSON line with n_gpus 2 from rank 0.  No GPU here, so the ranks run the plumbing-only body
    (launcher, rendezvous, barrier, max-over-ranks timing, JSON) over gloo; the same launcher code starts the real ranks on a GPU box
    (tests/test_gpu_step.py::test_bench_two_rank_rehearsal)."""
    import json
    r = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--plumbing-only"], {"PULPO_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["parallelism"] == "dp2" and d["valid"] is False
    # the audit fields of a multi-rank record: which loop, which stepper switches, did a fallback happen, what did the process group see
    assert d["loop"] == "none" and d["stepper"]["fallback"] == "none"
    assert d["dist"] == {"backend": "gloo", "world": 2, "devices": ["cpu (rank 0)", "cpu (rank 1)"]}


def test_bench_launcher_fails_loudly_when_a_rank_dies():
    """a rank that exits non-zero ends the whole job with a non-zero code (no silent single-GPU result, no hang)"""
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"PULPO_DIST_BACKEND": "gloo", "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0                          # no GPU: every rank refuses to run the product path
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], r.stdout
    assert "needs a ROCm GPU" in r.stderr or "exited with code" in r.stderr, r.stderr[-2000:]


def test_bench_refuses_a_world_size_mismatch():
    """--gpus N under a torchrun environment of a different size is an error, not a silently different measurement"""
    r = _bench(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
