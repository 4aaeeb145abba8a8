#!/usr/bin/env python3
"""Throughput of PULPo's registration training step (forward + backward + gradient all-reduce + Adam) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` with N > 1 and no torchrun environment (WORLD_SIZE unset): this process becomes a launcher.  Before anything touches the
GPU it starts N fresh rank processes of this same script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, rank r
pinned to GPU r), relays rank 0's single JSON line, and exits non-zero if any rank fails.  Under torchrun the ranks are the driver's.

Workload (BASELINE.json `metric`): synthetic 160^3 fp32 volume pairs, 4-level latent pyramid (total_levels 5,
latent_levels 4, n0 = 32), batch 1 per GPU, weak scaling over GPUs.  One "step" = one training step on one pair per
GPU.  Inputs live in HBM before the timed region.  Rank 0 prints ONE JSON line (see the task contract) carrying
  roofline      : the dominant kernel (the MFMA 3x3x3 convolution), algorithmic FLOP / launch over its HIP-event
                  measured mean launch time inside the timed region, against the dense fp32 MFMA peak (157.3 TFLOP/s);
  cpu_baseline  : the CPU oracle (oracle/pulpo_oracle.py, the same ATen op sequence as the reference) timed on this
                  host's cores: 1 warm-up + 2 timed forward+backward+Adam steps of the same 160^3 workload (N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FEEDBACK = ["samples", "velocity_fields", "individual_dfs", "combined_dfs", "final_dfs", "transformed"]
PEAK_BF16_MFMA_TFLOPS = 2500.0         # dense bf16 matrix peak
PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 matrix peak (= vector peak)
FLOP_ALG_PER_PAIR_160 = 5.708e12       # conv FLOPs fwd+bwd per pair at 160^3 / T5 / L4
BYTES_ALG_PER_PAIR_160 = 41.98e9       # fused-kernel compulsory bytes per pair


def ISSUED_FRACTION(kernel_name: str) -> float:
    """matrix-pipe FLOP issued per direct-convolution FLOP for a traced kernel name"""
    if "wino3" in kernel_name or "wgrad_w3" in kernel_name:
        return 8.0 / 27.0
    if "wgrad_w2" in kernel_name or "wino2" in kernel_name:
        return 4.0 / 9.0
    if "wino" in kernel_name:
        return 2.0 / 3.0
    return 1.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, nargs=3, default=[160, 160, 160])
    ap.add_argument("--levels", type=int, nargs=2, default=[5, 4], help="total_levels latent_levels")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="conv operand precision: fp32 (the metric's configuration, default) or bf16 operands / fp32 accumulate "
                         "(BASELINE configs 4-5; reported under its own metric name, never as the fp32 headline)")
    ap.add_argument("--activations", default=None, choices=["fp32", "bf16"],
                    help="storage type of the multi-channel activation tensors and their gradients; default: fp32 with fp32 operands, bf16 with "
                         "--precision bf16 (BASELINE configs 4-5: bf16 operands AND bf16 activation storage, fp32 arithmetic / statistics / losses / Adam)")
    ap.add_argument("--data", default="uniform", choices=["uniform", "oasis"],
                    help="uniform: U[0,1) volumes (configs 1-3); oasis: masked smooth anatomy + smooth random deformation (configs 4-5)")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "mc8"],
                    help="train (the metric: fwd+bwd+all-reduce+Adam), infer (eval-mode predict_deterministic) or mc8 (8-sample Monte-Carlo "
                         "uncertainty maps of one pair, BASELINE config 5) - the latter two are reported under their own metric names")
    ap.add_argument("--loop", default="stepper", choices=["stepper", "lightning", "plain-autograd"],
                    help="who drives the training step: stepper = dp.DataParallelStepper.step (default); lightning = the LightningModule hooks of "
                         "src.models.PULPo called in pytorch_lightning 1.8's order (pulpo_amd/_lightning.py::HookOrderTrainer - what an unchanged "
                         "train.py runs); plain-autograd = loss.backward() + torch.optim.Adam with every fast-path switch off (one GPU only)")
    ap.add_argument("--deterministic", action="store_true",
                    help="run the measured loop in deterministic mode (pulpo_amd.ops.set_deterministic: ordered sums instead of float atomics in the "
                         "backward kernels; bit-identical gradients run to run).  Default: off; the line reports the mode's step time either way")
    ap.add_argument("--graph", action="store_true",
                    help="stepper loop only: zero_grad + forward + backward + gradient finishing replayed from ONE HIP graph (dp.DataParallelStepper(graph=True)); "
                         "gradient exchange, fused Adam and weight re-pack stay eager.  For the short bf16 steps, which the host barely keeps ahead of")
    ap.add_argument("--no-loops", action="store_true", help="do not time the other two loops after the measurement (one GPU, train mode)")
    ap.add_argument("--host-input", action="store_true",
                    help="feed every step from host memory through pulpo_amd.prefetch.DevicePrefetcher (PCIe-inclusive rate; the default "
                         "keeps the pair resident in HBM as the metric prescribes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trace", action="store_true", help="do not bracket conv launches with HIP events")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="TEST HOOK, not a measurement: run only the launch / rendezvous / barrier / max-over-ranks / JSON plumbing with a trivial "
                         "CPU all-reduce as the 'step' (no kernels, no GPU); the line it prints carries \"valid\": false")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ self-launch (--gpus N without torchrun)
EXIT_OVERLAP_FAILED = 17        # a rank's overlapped stepper failed on its first step: the launcher starts all ranks again with the plain one


def _free_port() -> int:
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def _spawn_ranks(n: int, extra_env: dict) -> int:
    """start n rank processes of this script, wait for all; returns the job's exit code.  Rank 0's stdout is relayed to ours (the one JSON
    line); the other ranks' stdout goes to our stderr.  The first rank to fail ends the job: the others are terminated by PID."""
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr, text=(r == 0) or None))
    rc = 0
    pending = set(range(n))
    deadline = time.time() + float(os.environ.get("PULPO_BENCH_TIMEOUT_S", "3000"))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench launcher] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
        if (rc != 0 or time.time() > deadline) and pending:
            if rc == 0:
                rc = 124
                print("[bench launcher] time limit reached; stopping the ranks", file=sys.stderr)
            for r in pending:
                procs[r].terminate()
            t_end = time.time() + 15
            for r in pending:
                try:
                    procs[r].wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            pending.clear()
        if pending:
            time.sleep(0.2)
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    if rc == 0:
        sys.stdout.write(out0)
        sys.stdout.flush()
    else:
        sys.stderr.write(out0)
    return rc


def launch_ranks(args) -> int:
    """parent of a self-launched multi-GPU run.  NOTHING here initialises the GPU (torch.cuda.device_count() does not, on this image)."""
    n = args.gpus
    backend = os.environ.get("PULPO_DIST_BACKEND", "nccl")
    if backend == "nccl" and not args.plumbing_only:
        have = torch.cuda.device_count()
        if have < n:
            print(f"[bench launcher] --gpus {n} but only {have} GPU(s) are visible (one RCCL rank per GPU)", file=sys.stderr)
            return 2
    rc = _spawn_ranks(n, {})
    if rc == EXIT_OVERLAP_FAILED:
        print("[bench launcher] starting all ranks again with the plain stepper (PULPO_DP_OVERLAP=0 PULPO_ASYNC_WGRAD=0)", file=sys.stderr)
        rc = _spawn_ranks(n, {"PULPO_DP_OVERLAP": "0", "PULPO_ASYNC_WGRAD": "0", "PULPO_BENCH_RELAUNCHED": "1"})     # (the line says so: stepper.fallback)
    return rc


def plumbing_only(args) -> None:
    """the multi-rank plumbing of this script with a trivial CPU 'step' (test hook for boxes without a GPU; prints "valid": false)"""
    from pulpo_amd import dp
    dp.init_from_env("gloo")
    world = dp.world()
    rank = dist.get_rank() if world > 1 else 0
    buf = torch.ones(1024)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        dp.allreduce_sum_(buf.clone())
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = dp.allreduce_sum_(buf.clone())
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert float(got[0]) == float(world)
    names = [None] * world
    if world > 1:
        dist.all_gather_object(names, f"cpu (rank {rank})")
    else:
        names = ["cpu (rank 0)"]
    if rank == 0:
        print(json.dumps({"metric": "plumbing only (no kernels run; not a measurement)", "valid": False, "value": None, "unit": "volume-pairs/s",
                          "loop": "none", "stepper": {"overlap": False, "async_wgrad": False, "wgrad_side_stream": False,
                                                      "fallback": "relaunched" if os.environ.get("PULPO_BENCH_RELAUNCHED") == "1" else "none"},
                          "dist": {"backend": dist.get_backend() if world > 1 else None, "world": world, "devices": names},
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / max(1, args.steps) * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none",
                          "config": {"workload": "launcher / rendezvous / barrier / max-over-ranks rehearsal", "global_batch": world,
                                     "parallelism": f"dp{world}"}, "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.destroy_process_group()


def usable_cores() -> int:
    """cores this process may actually use: min(affinity mask, cgroup CPU quota); os.cpu_count() reports the whole host"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError, IndexError):
            pass
    return int(os.environ.get("PULPO_CPU_CORES", min(n, 32)))


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(size, T, L, B, warm: int = 1, timed: int = 2):
    """the CPU oracle's training step (forward + backward + torch.optim.Adam, the reference's optimizer - models.py:398-400) on the same
    workload: `warm` untimed + `timed` timed steps (1 + 2), autograd anomaly mode off"""
    from oracle import pulpo_oracle as O
    torch.set_num_threads(usable_cores())
    cfg = O.Cfg(T, L, list(size), n0=32)
    sd = O.clone_sd(O.init_state_dict(cfg, seed=0), requires_grad=True)
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4)
    g = torch.Generator().manual_seed(1234)
    x, y = torch.rand(B, 1, *size, generator=g), torch.rand(B, 1, *size, generator=g)

    def step():
        _, grads, _ = O.train_step(sd, cfg, x, y, None)
        for k, v in sd.items():
            if v.requires_grad:
                v.grad = grads.get(k)
        opt.step()

    for _ in range(warm):
        step()
    t0 = time.perf_counter()
    for _ in range(timed):
        step()
    dt = (time.perf_counter() - t0) / timed
    return {"value": B / dt, "unit": "volume-pairs/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{warm} warm-up + {timed} timed fwd+bwd+Adam steps of the same {size[0]}x{size[1]}x{size[2]} T{T}/L{L} B={B} fp32 workload, "
                      f"oracle/pulpo_oracle.py on torch-CPU ({torch.get_num_threads()} threads, {cpu_model()}), {dt:.1f} s per step"}


def pmc_traffic(kernel: str) -> dict:
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes of this same command (the newest
    profiles/r*_bench160_pmc_traffic.json, made by scripts/pmc_traffic.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs; counters
    cannot be read from inside the process).  Launch-weighted over the kernel's tile variants."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench160_pmc_traffic.json")))
    if not files:
        return {"traffic": None}
    path = files[-1]
    rows = json.load(open(path))["per_launch"]
    stem = kernel.replace(" ", "").rstrip(">")
    hit = [r for k, r in rows.items() if k.replace(" ", "").startswith(stem)]
    n = sum(r["launches"] for r in hit)
    if n == 0:
        return {"traffic": None, "traffic_source": f"{os.path.relpath(path, ROOT)} holds no launch of {kernel} (regenerate it for this build)"}
    factors = sorted({r["fetch_factor"] for r in hit})
    return {"traffic": sum(r["traffic_bytes"] * r["launches"] for r in hit) / n, "traffic_unit": "bytes/launch",
            "traffic_fetch_bytes": sum(r["fetch_raw_bytes"] * r["launches"] for r in hit) / n,
            "traffic_write_bytes": sum(r["write_bytes"] * r["launches"] for r in hit) / n,
            "traffic_source": f"{os.path.relpath(path, ROOT)}: rocprofv3 --pmc passes of this command, traffic = FETCH_SIZE x {'/'.join(f'{f:g}' for f in factors)} "
                              "+ WRITE_SIZE (factor 2 = the guide's gfx950 correction for 16 B/lane reads, confirmed for this kernel's 32-byte "
                              "gathers by scripts/probes/fetch_calib.hip, profiles/r3_fetch_calibration.md)"}


def make_loop(kind: str, model, graph: bool = False):
    """-> (run(batch) -> loss, the dp.DataParallelStepper behind it or None, the name reported in the line's "loop" field)"""
    from pulpo_amd import dp
    if kind == "stepper":
        stepper = dp.DataParallelStepper(model, graph=graph)
        return stepper.step, stepper, "stepper (HIP graph)" if graph else "stepper"
    if kind == "lightning":
        from pulpo_amd._lightning import HookOrderTrainer
        trainer = HookOrderTrainer()
        trainer.attach(model)                       # configure_optimizers() -> dp.ArenaAdam; every step goes through the module's hooks
        eng = model._engine()
        if eng is None:
            raise SystemExit("bench: --loop lightning with PULPO_LIGHTNING_FAST=0 is the plain-autograd loop; ask for that one")
        return (lambda batch: trainer.run_batch(batch, 0)), eng, "lightning-hooks"
    opt = torch.optim.Adam(model.parameters(), lr=float(model.hparams.lr))          # the reference's optimizer as it stands (models.py:398-400)

    def run(batch):
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        loss.backward()
        opt.step()
        return loss.detach()
    return run, None, "plain-autograd"


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))           # (nothing has touched the GPU in this process)
    if args.plumbing_only:
        return plumbing_only(args)
    # (a full Python garbage collection costs 60 - 100 ms of host time every few steps - a sixth of the short bf16 steps: the stepper moves
    #  what is alive after its second step to the collector's permanent generation.  Opt-in; this process is nothing but the benchmark.)
    os.environ.setdefault("PULPO_GC_FREEZE", "1")
    from pulpo_amd import dp, ops
    from pulpo_amd._lib import lib
    local = dp.init_from_env(os.environ.get("PULPO_DIST_BACKEND", "nccl"))     # (gloo: rehearsal of the multi-rank path on a one-GPU box)
    world = dp.world()
    rank = dist.get_rank() if world > 1 else 0
    if world != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but WORLD_SIZE is {world}: launch one rank per GPU (or drop the torchrun environment "
                         "and let bench.py start the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    if local >= torch.cuda.device_count():           # only in the gloo rehearsal with more ranks than GPUs
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib.load()
    acts = args.activations or ("bf16" if args.precision == "bf16" else "fp32")
    ops.set_conv_precision(args.precision, activations=acts)
    if args.deterministic:
        ops.set_deterministic(True)
    bf16 = args.precision == "bf16"
    prec_name = "fp32" if not bf16 else f"bf16 conv operands (fp32 accumulate), {acts} activation storage"

    from src.models import PULPo
    T, L = args.levels
    size, B = list(args.size), args.batch
    torch.manual_seed(0)
    if args.loop == "plain-autograd" and world > 1:
        raise SystemExit("bench: --loop plain-autograd has no gradient exchange; one GPU only")
    model = PULPo(T, L, 0.1, size, feedback=FEEDBACK, n0=32).to(dev).train()
    if args.graph and args.loop != "stepper":
        raise SystemExit("bench: --graph belongs to --loop stepper")
    run_step, stepper, loop_name = make_loop(args.loop, model, graph=args.graph)       # stepper: the dp.DataParallelStepper that owns arena / streams (None: plain autograd)
    fallback = "relaunched" if os.environ.get("PULPO_BENCH_RELAUNCHED") == "1" else "none"
    from pulpo_amd import synthetic
    x, y = (synthetic.oasis_like_pair if args.data == "oasis" else synthetic.uniform_pair)(size, B, 1234 + rank, dev)
    empty = torch.empty((0,), device=dev)
    batch = (x, y, empty, empty, empty, empty, empty, empty)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    infer = args.mode in ("infer", "mc8")
    if args.mode == "mc8":
        from pulpo_amd.uncertainty import mc_uncertainty
        model.eval()

        def one_step():
            res = mc_uncertainty(model, x[:1], y[:1], 8)
            return res["output_std"][0].sum()
    elif infer:
        model.eval()

        def one_step():
            with torch.no_grad():
                out, _ = model.predict_deterministic(x, y)
            return out[0].sum()
    elif args.host_input:
        from pulpo_amd.prefetch import DevicePrefetcher
        host_batch = tuple(t.cpu() for t in batch)
        feed = iter(DevicePrefetcher((host_batch for _ in range(args.warmup + args.steps + 8)), dev))

        def one_step():
            return run_step(next(feed))
    else:
        def one_step():
            return run_step(batch)

    # Insurance for the multi-rank RCCL runs, which a one-GPU box cannot rehearse: the first step with the overlapped gradient exchange /
    # second-stream weight gradients is tried, and the outcome is agreed on by ALL ranks before anybody goes on - a flag all-reduce over a
    # separate gloo (CPU) group, issued before the GPU is synchronised, so a rank whose peer failed is not left waiting inside a collective
    # the peer never joins.  All ranks failed (a deterministic failure is the same everywhere and leaves no half-issued collective): every
    # rank switches to the plain single-all-reduce stepper.  Some failed: nothing in this process can be trusted any more - every rank
    # exits with EXIT_OVERLAP_FAILED and the launcher starts the job again with PULPO_DP_OVERLAP=0 PULPO_ASYNC_WGRAD=0.
    if world > 1 and not infer and args.warmup > 0 and stepper is not None and (stepper.overlap or stepper.async_wgrad):
        import datetime
        side = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=float(os.environ.get("PULPO_BENCH_AGREE_TIMEOUT_S", "300"))))
        ok = 1.0
        try:
            if os.environ.get("PULPO_BENCH_INJECT_FAILURE", "") in ("all", str(rank)):      # test hook
                raise RuntimeError("injected first-step failure")
            one_step()
        except Exception as exc:  # noqa: BLE001
            ok = 0.0
            print(f"[bench] rank {rank}: overlapped stepper failed on its first step ({type(exc).__name__}: {exc})", file=sys.stderr)
        votes = torch.tensor([ok, 1.0])
        try:
            dist.all_reduce(votes, op=dist.ReduceOp.SUM, group=side)
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] rank {rank}: no agreement on the first step's outcome ({type(exc).__name__}: {exc})", file=sys.stderr)
            os._exit(EXIT_OVERLAP_FAILED)
        n_ok = int(votes[0].item())
        if n_ok == world:
            torch.cuda.synchronize()
        elif n_ok == 0:
            if rank == 0:
                print("[bench] every rank failed alike: falling back to overlap=False, async_wgrad=False on all ranks", file=sys.stderr)
            ops.ASYNC_WGRAD_STREAM = None
            ops.DIRECT_PARAM_GRADS = False
            ops.reset_param_grad_buffers(model)
            stepper.overlap, stepper.async_wgrad = False, False          # the same engine, plain: one all-reduce after backward, weight gradients in line
            stepper._works, stepper._launched, stepper._armed = [], 0, False
            fallback = "in-place"
        else:
            print(f"[bench] rank {rank}: {world - n_ok} of {world} ranks failed their first step; leaving with code {EXIT_OVERLAP_FAILED}", file=sys.stderr)
            sys.stderr.flush()
            os._exit(EXIT_OVERLAP_FAILED)
    for _ in range(args.warmup + (3 if args.graph else 0)):      # (graph mode: two eager steps and the capture step come first)
        one_step()
    barrier()
    graph_replay = bool(args.graph and stepper is not None and stepper._graph is not None)
    if not args.no_trace and not graph_replay:
        ops.CONV_TRACE = []
        ops.CONV_TRACE_STRIDE = 7        # one in seven conv launches of the timed region is bracketed (drawn at random: no fixed stride can lock onto a layer)
    ops.CONV_TRACE_STRIDE_USED = ops.CONV_TRACE_STRIDE
    if stepper is not None and world > 1:
        stepper.exchange_events = []        # HIP events around the waits for the gradient exchange: what of it is NOT hidden under the backward pass
    host_enq = []                           # host time spent inside each step's enqueue (no synchronisation inside a step: the host runs ahead)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        th = time.perf_counter()
        loss = one_step()
        host_enq.append(time.perf_counter() - th)
    barrier()
    dt = time.perf_counter() - t0
    peak_gb = torch.cuda.max_memory_allocated(dev) / 1e9     # (of the measured loop: the extra steps below - other loops, deterministic mode - allocate their own)
    trace, ops.CONV_TRACE = ops.CONV_TRACE, None
    trace_steps = args.steps
    if graph_replay and not args.no_trace:
        # the timed region replayed a HIP graph (no per-launch events): the dominant kernel's brackets come from two extra EAGER steps of the same
        # loop, every launch bracketed (the step falls back to its eager form while a trace is armed)
        ops.CONV_TRACE, ops.CONV_TRACE_STRIDE = [], 1
        for _ in range(2):
            one_step()
        torch.cuda.synchronize()
        trace, ops.CONV_TRACE = ops.CONV_TRACE, None
        ops.CONV_TRACE_STRIDE_USED, trace_steps = 1, 2
    # The memory-bound kernel classes (BatchNorm / LeakyReLU passes, warp, VecInt, NCC, pooling / resizing, heads, KL, regulariser, Adam) are
    # bracketed in two EXTRA untimed steps of the same overlapped loop (every launch): their brackets in the timed region would cost host
    # time per launch that the short bf16 steps cannot hide (measured: 17.0 -> 23.6 ms per step), and they are a supplementary report - the
    # dominant kernel's roofline above comes from the timed region as the contract asks.
    hbm_trace = None
    if trace is not None and not infer:
        ops.HBM_TRACE, ops.CONV_TRACE_STRIDE = [], 1
        for _ in range(2):
            one_step()
        torch.cuda.synchronize()
        hbm_trace, ops.HBM_TRACE = ops.HBM_TRACE, None
    # Inside the timed region the weight gradients run on a second stream next to the main stream's kernels, so a kernel's HIP-event
    # bracket there includes the time it shares the CUs.  Two extra, untimed steps with that overlap switched off give the same
    # kernels' stand-alone durations (reported as roofline["serialized"]; the throughput value is NOT taken from these steps).
    trace_serial = None
    exchange_ms = None
    if stepper is not None and stepper.exchange_events:
        exchange_ms = sum(a_.elapsed_time(b_) for a_, b_ in stepper.exchange_events) / len(stepper.exchange_events)
    if stepper is not None:
        stepper.exchange_events = None
    if trace is not None and not infer and world == 1 and stepper is not None and stepper.async_wgrad:
        stepper.async_wgrad = False
        one_step()
        torch.cuda.synchronize()
        ops.CONV_TRACE = []
        ops.CONV_TRACE_STRIDE = 1
        for _ in range(2):
            one_step()
        torch.cuda.synchronize()
        trace_serial, ops.CONV_TRACE = ops.CONV_TRACE, None
        stepper.async_wgrad = True
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if not bool(torch.isfinite(loss)):
        raise SystemExit("bench: non-finite loss")
    devices = [None] * world
    mine = f"{torch.cuda.get_device_name(dev)} (cuda:{local})"
    if world > 1:
        dist.all_gather_object(devices, mine)
    else:
        devices = [mine]
    # the other two loops on fresh models of the same seed, a few steps each (one GPU, train mode): what an unchanged train.py gets
    # ("lightning-hooks") and what it got before the hooks existed ("plain-autograd"), beside the measured loop
    loops_ms = None
    if world == 1 and not infer and not args.no_loops and not args.host_input:
        loops_ms = {loop_name: dt / args.steps * 1e3}
        n_ab = max(3, min(args.steps, 10))
        for other, name2 in (("stepper", "stepper"), ("lightning", "lightning-hooks"), ("plain-autograd", "plain-autograd")):
            if name2 in loops_ms:
                continue
            torch.manual_seed(0)
            m2 = PULPo(T, L, 0.1, size, feedback=FEEDBACK, n0=32).to(dev).train()
            run2, _, _ = make_loop(other, m2)
            for _ in range(2):
                run2(batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n_ab):
                run2(batch)
            torch.cuda.synchronize()
            loops_ms[name2] = (time.perf_counter() - t1) / n_ab * 1e3
            del m2, run2
            ops.invalidate_weight_packs()
            torch.cuda.empty_cache()

    # ... and the same loop in the other determinism mode (a few untimed-region steps): what the ordered sums cost
    det_ms = None
    if world == 1 and not infer and not args.no_loops and not args.host_input:
        ops.set_deterministic(not args.deterministic)
        for _ in range(2):
            one_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_det = max(3, min(args.steps, 10))
        for _ in range(n_det):
            one_step()
        torch.cuda.synchronize()
        det_ms = (time.perf_counter() - t1) / n_det * 1e3
        ops.set_deterministic(args.deterministic)

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / dt
        is160_cfg = size == [160, 160, 160] and (T, L) == (5, 4) and B == 1
        # ---- dominant kernel from the live HIP-event trace
        roof = None
        per_kernel = {}
        if trace:
            for name, flops, s_, e_, nb in trace:
                k = per_kernel.setdefault(name, [0, 0.0, 0.0, 0.0])
                k[0] += 1
                k[1] += flops
                k[2] += s_.elapsed_time(e_) * 1e-3
                k[3] += nb
            dom = max(per_kernel.items(), key=lambda kv: kv[1][2])
            side_stream = bool(stepper is not None and stepper.wgrad_on_side_stream())
            n, fl, sec, nbytes = dom[1]
            peak = PEAK_BF16_MFMA_TFLOPS if "bf16" in dom[0] else PEAK_FP32_MFMA_TFLOPS
            # FLOP the kernel ISSUES on the matrix pipe per algorithmic (direct-convolution) FLOP: the Winograd forms evaluate the
            # same convolution with 2/3 (F(2,3) along x) or 4/9 (F(2x2,3x3) in y and x) of the direct form's multiply-adds
            issued = ISSUED_FRACTION(dom[0])
            eff = fl / sec / 1e12
            roof = {"bound": "mfma", "kernel": dom[0], "achieved": eff * issued, "peak": peak, "unit": "TFLOP/s",
                    "frac": eff * issued / peak, "traffic": None, "launches": n,
                    "launch_sampling": (f"every {ops.CONV_TRACE_STRIDE_USED}th conv launch of the timed region" if not graph_replay else
                                        "every conv launch of two extra eager steps (the timed region replays a HIP graph: no per-launch events)"),
                    "avg_launch_ms": sec / n * 1e3,
                    "issued_flop_per_launch": fl * issued / n,
                    "effective_TFLOPs": eff, "effective_flop_per_launch": fl / n,
                    "alg_bytes_per_launch": nbytes / n,
                    "flop_definition": "achieved / frac = FLOP the kernel issues on the matrix pipe (utilisation, <= 1); effective_TFLOPs = the direct "
                                       "convolution's 54*K*N*V FLOP (SURVEY 8(d)) over the same time" + ("" if issued == 1.0 else
                                       f"; this Winograd kernel issues {issued:.4f} of them")}
            if trace_serial:
                ts = [(fl_, s_.elapsed_time(e_) * 1e-3) for name_, fl_, s_, e_, _ in trace_serial if name_ == dom[0]]
                if ts:
                    fls, secs = sum(a_ for a_, _ in ts), sum(b_ for _, b_ in ts)
                    roof["serialized"] = {"achieved": fls * issued / secs / 1e12, "frac": fls * issued / secs / 1e12 / peak,
                                          "effective_TFLOPs": fls / secs / 1e12, "avg_launch_ms": secs / len(ts) * 1e3, "launches": len(ts),
                                          "note": ("same kernel with the weight gradients queued in line instead of on the second stream (2 untimed steps)"
                                                   if side_stream else
                                                   "the same brackets repeated in 2 untimed steps with every launch bracketed; the weight gradients already run in "
                                                   "line in the timed region (stepper.wgrad_side_stream = false), so this is a repeat measurement, not a different schedule")}
            roof["overlap_note"] = ("timed-region brackets include CU sharing with the weight-gradient kernels on the second stream" if side_stream else
                                    "nothing overlaps in this step: weight gradients run in line on the main stream (stepper.wgrad_side_stream = false), "
                                    "every bracket is the kernel alone on the machine")
            if is160_cfg and not bf16:
                roof.update(pmc_traffic(dom[0]))
                if roof.get("traffic"):
                    roof["traffic_over_alg_bytes"] = roof["traffic"] / roof["alg_bytes_per_launch"]
        # ---- the HBM-bound family (BatchNorm / LeakyReLU passes): algorithmic bytes over the same live brackets, against 8 TB/s
        hbm_roof = None
        if hbm_trace:
            fam = {}
            for name, nbytes, s_, e_ in hbm_trace:
                k = fam.setdefault(name, [0, 0.0, 0.0])
                k[0] += 1
                k[1] += nbytes
                k[2] += s_.elapsed_time(e_) * 1e-3
            hbm_roof = {name: {"bound": "hbm", "achieved": v[1] / v[2] / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": v[1] / v[2] / 8.0e12,
                               "launches_sampled": v[0], "avg_launch_ms": v[2] / v[0] * 1e3} for name, v in fam.items()}
        is160 = is160_cfg
        # ---- where the step's time goes, from this run's own HIP events: matrix kernels (sampled brackets of the timed region, scaled by the
        # sampling stride), the bracketed memory-bound classes (every launch of the two extra untimed steps), the host's enqueue time
        time_split = None
        if trace and not infer:
            ms_step = dt / args.steps * 1e3
            matrix_ms = sum(v[2] for v in per_kernel.values()) * ops.CONV_TRACE_STRIDE_USED / trace_steps * 1e3
            hbm_ms = (sum(s_.elapsed_time(e_) for _, _, s_, e_ in hbm_trace) / 2.0) if hbm_trace else None
            enq = sorted(host_enq)
            overlapped = bool(stepper is not None and stepper.wgrad_on_side_stream())
            time_split = {"ms_per_step": ms_step, "matrix_ms_per_step": matrix_ms,
                          "non_matrix_ms_per_step": None if overlapped else ms_step - matrix_ms,
                          "hbm_classes_ms_per_step": hbm_ms,
                          "host_enqueue_ms_per_step": enq[len(enq) // 2] * 1e3 if enq else None,
                          "note": "matrix = every 3x3x3 convolution / weight-gradient launch (1-in-%d sampled HIP-event brackets of the timed region, scaled); "
                                  "non_matrix = step - matrix (%s); hbm_classes = sum of the brackets of the memory-bound kernel classes listed under "
                                  "hbm_rooflines in two extra untimed steps (finalize / column-sum / torch glue launches and kernel boundaries are in "
                                  "non_matrix but not in hbm_classes); host_enqueue = median host time inside one step's enqueue (timed region).  CAVEAT: an "
                                  "event bracket spans start-event -> end-event, i.e. the kernel plus its launch turnaround (2 - 3 us each), so bracket sums run ~8 %% "
                                  "above the kernels' own durations (matrix + hbm_classes exceed the step) and non_matrix = step - matrix is a LOWER bound; the "
                                  "kernel-only split of this build is in profiles/r5_bench160_stream_cost.txt (rocprofv3 --kernel-trace: matrix 21.4, other 6.6 ms)"
                                  % (ops.CONV_TRACE_STRIDE_USED, "weight gradients on a second stream: not defined" if overlapped else "nothing overlaps: weight gradients in line")}
        out = {
            "metric": ("volume-pairs/sec, 8-sample MC uncertainty maps per pair, " if args.mode == "mc8" else "volume-pairs/sec inference (predict_deterministic), " if infer
                       else "volume-pairs/sec fwd+bwd, ") + ("160^3 " if size == [160, 160, 160] else f"{size[0]}x{size[1]}x{size[2]} ") + prec_name,
            "value": value, "unit": "volume-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": (f"bf16xbf16->f32 convs, {acts} activations in HBM, f32 arithmetic / statistics / losses / optimizer" if bf16 else "f32"), "input": "host memory via DevicePrefetcher (PCIe-inclusive)" if args.host_input else "resident in HBM",
            "data": ("synthetic OASIS-style pair (masked smooth anatomy, smooth random deformation)" if args.data == "oasis" else "synthetic U[0,1) volumes")
            + ", default-initialised weights (manual_seed 0)",
            "config": {"workload": f"{size[0]}x{size[1]}x{size[2]} synthetic pair, {L}-level pyramid (total_levels {T}), {prec_name}, batch {B} per GPU, "
                                   + ("eval-mode forward (mu path, no sampling)" if infer else "fwd+bwd+grad all-reduce+Adam"), "global_batch": world * B, "parallelism": f"dp{world}"},
            "loop": loop_name,
            "deterministic": {"measured_in_deterministic_mode": bool(args.deterministic),
                              ("ms_per_step_atomic_mode" if args.deterministic else "ms_per_step_deterministic_mode"): det_ms,
                              "note": "deterministic mode (PULPO_DETERMINISTIC=1): weight-gradient partial sums through ordered per-split slabs, VecInt / warp "
                                      "backward scatter in 64-bit fixed point - bit-identical gradients run to run, as the reference's CPU backward"},
            "loops_ms_per_step": loops_ms,
            "stepper": None if stepper is None else {"overlap": bool(stepper.overlap and world > 1), "async_wgrad": bool(stepper.async_wgrad),
                                                     "wgrad_side_stream": bool(stepper.wgrad_on_side_stream()),
                                                     "buckets": len(stepper.buckets) if (stepper.overlap and world > 1) else 1, "fallback": fallback,
                                                     "exposed_exchange_ms_per_step": exchange_ms},
            "dist": {"backend": dist.get_backend() if world > 1 else None, "world": dist.get_world_size() if world > 1 else 1, "devices": devices},
            "parity_note": "outputs / losses within 1e-4 of the CPU oracle at this size; whole-step parameter gradients within SURVEY 8(c)'s 5e-3 per "
                           "parameter at 160^3 (measured maximum 3.9e-3, in the atomic and the deterministic mode alike: profiles/r5_parity_160.md); on the "
                           "32^3 reference golden 5e-3 in deterministic mode / 1e-2 with the direct kernels forced, against a suggested 1e-3 that the "
                           "reference's own fp32-vs-fp64 distance (2e-3) does not leave room for (DESIGN.md section 4)",
            "roofline": roof,
            "time_split": time_split,
            "hbm_rooflines": hbm_roof,
            "hbm_peak_allocated_GB": peak_gb,
            "conv_kernels": {k: {"launches_sampled": v[0], "effective_TFLOPs": v[1] / v[2] / 1e12, "matrix_pipe_TFLOPs": v[1] * ISSUED_FRACTION(k) / v[2] / 1e12,
                                 "ms_total_per_step_est": v[2] * ops.CONV_TRACE_STRIDE_USED / trace_steps * 1e3}
                             for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][2])},
        }
        if is160 and not infer:
            per_gpu = value / world
            mpeak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
            out["step_rooflines"] = {f"effective_conv_TFLOPs_over_{mpeak:g}TF": FLOP_ALG_PER_PAIR_160 * per_gpu / 1e12 / mpeak,
                                     "alg_bytes_frac_of_8TBps": BYTES_ALG_PER_PAIR_160 * per_gpu / 8.0e12,
                                     "note": "whole step: SURVEY 8(d)'s 5.708 TFLOP (direct-convolution count) and 41.98 GB per pair over the step time; "
                                             "the first is an effective rate (Winograd kernels issue fewer FLOP), not a utilisation"}
        if world == 1 and not args.no_cpu_baseline and not bf16 and not infer:
            out["cpu_baseline"] = cpu_baseline(size, T, L, B)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
