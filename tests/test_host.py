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
    assert lib.query("pulpo_abi_version") == 1
    # pure host-side size queries (no device work)
    assert lib.query("pulpo_conv3d_k3_packed_floats", 32, 32) == 27 * 32 * 64
    assert lib.query("pulpo_conv3d_k3_packed_floats", 2, 32) == 27 * 4 * 64
    assert lib.query("pulpo_conv3d_k3_stat_tiles", 1, 160, 160, 160) == 40 * 20 * 20      # 4x8x8 voxel tiles from 20^3 up (depth % 4 == 0)
    assert lib.query("pulpo_conv3d_k3_stat_tiles", 2, 32, 32, 32) == 2 * 8 * 4 * 4
    assert lib.query("pulpo_conv3d_k3_stat_tiles", 1, 10, 10, 10) == 5 * 2 * 2             # 2x8x8 otherwise
    assert lib.query("pulpo_conv3d_k3_packed_bf16_elems", 48, 32) == 2 * 27 * 64 * 32
    assert lib.query("pulpo_conv3d_k3_algo", 1, 40, 40, 40, 64, 128) in (1, 2) and lib.query("pulpo_conv3d_k3_algo", 1, 10, 10, 10, 192, 192) == 0
    assert lib.query("pulpo_conv3d_k3_algo", 1, 160, 160, 160, 2, 32) == 0                  # image input layers: direct kernel
    assert lib.query("pulpo_conv3d_k3_packed_wino_floats", 20, 12) == 3 * 9 * 4 * 8 * 64
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
