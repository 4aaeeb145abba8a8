#!/usr/bin/env python3
"""Throughput of PULPo's registration training step (forward + backward + gradient all-reduce + Adam) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json `metric`): synthetic 160^3 fp32 volume pairs, 4-level latent pyramid (total_levels 5,
latent_levels 4, n0 = 32), batch 1 per GPU, weak scaling over GPUs.  One "step" = one training step on one pair per
GPU.  Inputs live in HBM before the timed region.  Rank 0 prints ONE JSON line (see the task contract) carrying
  roofline      : the dominant kernel (the MFMA 3x3x3 convolution), algorithmic FLOP / launch over its HIP-event
                  measured mean launch time inside the timed region, against the dense fp32 MFMA peak (157.3 TFLOP/s);
  cpu_baseline  : the CPU oracle (oracle/pulpo_oracle.py, the same ATen op sequence as the reference) timed on this
                  host's cores for ONE forward+backward step of the same 160^3 workload (N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
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
    ap.add_argument("--data", default="uniform", choices=["uniform", "oasis"],
                    help="uniform: U[0,1) volumes (configs 1-3); oasis: masked smooth anatomy + smooth random deformation (configs 4-5)")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "mc8"],
                    help="train (the metric: fwd+bwd+all-reduce+Adam), infer (eval-mode predict_deterministic) or mc8 (8-sample Monte-Carlo "
                         "uncertainty maps of one pair, BASELINE config 5) - the latter two are reported under their own metric names")
    ap.add_argument("--host-input", action="store_true",
                    help="feed every step from host memory through pulpo_amd.prefetch.DevicePrefetcher (PCIe-inclusive rate; the default "
                         "keeps the pair resident in HBM as the metric prescribes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trace", action="store_true", help="do not bracket conv launches with HIP events")
    return ap.parse_args()


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


def cpu_baseline(size, T, L, B):
    """one un-warmed forward+backward of the CPU oracle on the same workload (bounded sample: 1 step)"""
    from oracle import pulpo_oracle as O
    torch.set_num_threads(usable_cores())
    cfg = O.Cfg(T, L, list(size), n0=32)
    sd = O.clone_sd(O.init_state_dict(cfg, seed=0), requires_grad=True)
    g = torch.Generator().manual_seed(1234)
    x, y = torch.rand(B, 1, *size, generator=g), torch.rand(B, 1, *size, generator=g)
    t0 = time.perf_counter()
    O.train_step(sd, cfg, x, y, None)
    dt = time.perf_counter() - t0
    return {"value": B / dt, "unit": "volume-pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 un-warmed fwd+bwd step (no optimizer) of the same {size[0]}x{size[1]}x{size[2]} T{T}/L{L} B={B} fp32 workload, "
                      f"oracle/pulpo_oracle.py on torch-CPU, {dt:.1f} s"}


def pmc_traffic(kernel: str) -> dict:
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes of this same command
    (profiles/r1_bench160_pmc_traffic.json, made by scripts/pmc_traffic.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs;
    counters cannot be read from inside the process).  Launch-weighted over the kernel's tile variants."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_bench160_pmc_traffic.json")
    if not os.path.exists(path):
        return {"traffic": None}
    rows = json.load(open(path))["per_launch"]
    stem = kernel.replace(" ", "").rstrip(">")
    hit = [r for k, r in rows.items() if k.replace(" ", "").startswith(stem)]
    n = sum(r["launches"] for r in hit)
    if n == 0:
        return {"traffic": None}
    return {"traffic": sum(r["traffic_bytes"] * r["launches"] for r in hit) / n, "traffic_unit": "bytes/launch",
            "traffic_source": "profiles/r1_bench160_pmc_traffic.json (rocprofv3 --pmc passes of this command: FETCH_SIZE raw for the conv kernels' short gathers + WRITE_SIZE)"}


def main():
    args = parse()
    from pulpo_amd import dp, ops
    from pulpo_amd._lib import lib
    local = dp.init_from_env(os.environ.get("PULPO_DIST_BACKEND", "nccl"))     # (gloo: rehearsal of the multi-rank path on a one-GPU box)
    world = dp.world()
    rank = dist.get_rank() if world > 1 else 0
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE is {world}; using {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    if local >= torch.cuda.device_count():           # only in the gloo rehearsal with more ranks than GPUs
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    lib.load()
    ops.set_conv_precision(args.precision)
    bf16 = args.precision == "bf16"

    from src.models import PULPo
    T, L = args.levels
    size, B = list(args.size), args.batch
    torch.manual_seed(0)
    model = PULPo(T, L, 0.1, size, feedback=FEEDBACK, n0=32).to(dev).train()
    stepper = dp.DataParallelStepper(model)
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
            return stepper.step(next(feed))
    else:
        def one_step():
            return stepper.step(batch)

    # insurance for the multi-rank runs, which cannot be rehearsed with RCCL on a one-GPU box: if the first step fails with the overlapped
    # gradient exchange / second-stream weight gradients, every rank (the failure would be deterministic) falls back to the plain
    # single-all-reduce stepper and says so on stderr
    if world > 1 and not infer and args.warmup > 0:
        try:
            one_step()
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] rank {rank}: overlapped stepper failed ({type(exc).__name__}: {exc}); falling back to overlap=False, async_wgrad=False",
                  file=sys.stderr)
            ops.ASYNC_WGRAD_STREAM = None
            ops.DIRECT_PARAM_GRADS = False
            stepper = dp.DataParallelStepper(model, overlap=False, async_wgrad=False)
    for _ in range(args.warmup):
        one_step()
    barrier()
    if not args.no_trace:
        ops.CONV_TRACE = []
        ops.HBM_TRACE = []
        ops.CONV_TRACE_STRIDE = 7        # every 7th conv launch of the timed region is bracketed (the launch count per step is not a multiple of 7)
    ops.CONV_TRACE_STRIDE_USED = ops.CONV_TRACE_STRIDE
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    barrier()
    dt = time.perf_counter() - t0
    trace, ops.CONV_TRACE = ops.CONV_TRACE, None
    hbm_trace, ops.HBM_TRACE = ops.HBM_TRACE, None
    # Inside the timed region the weight gradients run on a second stream next to the main stream's kernels, so a kernel's HIP-event
    # bracket there includes the time it shares the CUs.  Two extra, untimed steps with that overlap switched off give the same
    # kernels' stand-alone durations (reported as roofline["serialized"]; the throughput value is NOT taken from these steps).
    trace_serial = None
    if trace is not None and not infer and world == 1 and stepper.async_wgrad:
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

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / dt
        is160_cfg = size == [160, 160, 160] and (T, L) == (5, 4) and B == 1
        # ---- dominant kernel from the live HIP-event trace
        roof = None
        per_kernel = {}
        if trace:
            for name, flops, s, e in trace:
                k = per_kernel.setdefault(name, [0, 0.0, 0.0])
                k[0] += 1
                k[1] += flops
                k[2] += s.elapsed_time(e) * 1e-3
            dom = max(per_kernel.items(), key=lambda kv: kv[1][2])
            n, fl, sec = dom[1]
            peak = PEAK_BF16_MFMA_TFLOPS if "bf16" in dom[0] else PEAK_FP32_MFMA_TFLOPS
            roof = {"bound": "mfma", "kernel": dom[0], "achieved": fl / sec / 1e12, "peak": peak, "unit": "TFLOP/s",
                    "frac": fl / sec / 1e12 / peak, "traffic": None, "launches": n, "launch_sampling": f"every {ops.CONV_TRACE_STRIDE_USED}th conv launch of the timed region",
                    "avg_launch_ms": sec / n * 1e3,
                    "flop_per_launch": fl / n}
            if "wino" in dom[0]:
                # algorithmic FLOP are those of the direct convolution; the Winograd kernels issue 2/3 (F(2,3) along x) or
                # 4/9 (F(2x2,3x3) in y and x) of them
                issued = 4.0 / 9.0 if "wino2" in dom[0] else 2.0 / 3.0
                roof["matrix_pipe_TFLOPs"] = roof["achieved"] * issued
                roof["matrix_pipe_frac"] = roof["matrix_pipe_TFLOPs"] / peak
                roof["note"] = ("achieved/frac count the direct convolution's 54*K*N*V FLOP; the kernel evaluates the "
                                + ("y and x taps with F(2x2,3x3) and issues 24*K*N*V" if "wino2" in dom[0] else "x taps with F(2,3) and issues 36*K*N*V"))
            if trace_serial:
                ts = [(fl_, s_.elapsed_time(e_) * 1e-3) for name_, fl_, s_, e_ in trace_serial if name_ == dom[0]]
                if ts:
                    fls, secs = sum(a_ for a_, _ in ts), sum(b_ for _, b_ in ts)
                    roof["serialized"] = {"achieved": fls / secs / 1e12, "frac": fls / secs / 1e12 / peak, "avg_launch_ms": secs / len(ts) * 1e3,
                                          "launches": len(ts), "note": "same kernel, weight-gradient stream overlap switched off (2 untimed steps)"}
                roof["overlap_note"] = "timed-region brackets include CU sharing with the weight-gradient kernels on the second stream"
            if is160_cfg and not bf16:
                roof.update(pmc_traffic(dom[0]))
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
        out = {
            "metric": ("volume-pairs/sec, 8-sample MC uncertainty maps per pair, " if args.mode == "mc8" else "volume-pairs/sec inference (predict_deterministic), " if infer
                       else "volume-pairs/sec fwd+bwd, ") + ("160^3 " if size == [160, 160, 160] else f"{size[0]}x{size[1]}x{size[2]} ") + ("bf16 conv operands (fp32 accumulate, fp32 activations)" if bf16 else "fp32"),
            "value": value, "unit": "volume-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16xbf16->f32 convs, f32 elsewhere" if bf16 else "f32", "input": "host memory via DevicePrefetcher (PCIe-inclusive)" if args.host_input else "resident in HBM",
            "data": ("synthetic OASIS-style pair (masked smooth anatomy, smooth random deformation)" if args.data == "oasis" else "synthetic U[0,1) volumes")
            + ", default-initialised weights (manual_seed 0)",
            "config": {"workload": f"{size[0]}x{size[1]}x{size[2]} synthetic pair, {L}-level pyramid (total_levels {T}), {'bf16 conv operands' if bf16 else 'fp32'}, batch {B} per GPU, "
                                   + ("eval-mode forward (mu path, no sampling)" if infer else "fwd+bwd+grad all-reduce+Adam"), "global_batch": world * B, "parallelism": f"dp{world}"},
            "roofline": roof,
            "hbm_rooflines": hbm_roof,
            "hbm_peak_allocated_GB": torch.cuda.max_memory_allocated(dev) / 1e9,
            "conv_kernels": {k: {"launches_sampled": v[0], "TFLOP/s": v[1] / v[2] / 1e12,
                                 "ms_total_per_step_est": v[2] * ops.CONV_TRACE_STRIDE_USED / args.steps * 1e3}
                             for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][2])},
        }
        if is160 and not infer:
            per_gpu = value / world
            mpeak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
            out["step_rooflines"] = {f"conv_flop_frac_of_{mpeak:g}TF": FLOP_ALG_PER_PAIR_160 * per_gpu / 1e12 / mpeak,
                                     "alg_bytes_frac_of_8TBps": BYTES_ALG_PER_PAIR_160 * per_gpu / 8.0e12}
        if world == 1 and not args.no_cpu_baseline and not bf16 and not infer:
            out["cpu_baseline"] = cpu_baseline(size, T, L, B)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
