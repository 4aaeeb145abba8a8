"""Data-parallel training over volume pairs: one process per GPU, flat fp32 parameter / gradient arenas, the gradient arena
summed over ranks by RCCL over xGMI in up to three contiguous buckets that are launched while the backward pass is still
running, fused Adam over the arenas.

The reference has no distributed code: pairs are independent, the only exchange is the gradient
sum.  BatchNorm statistics stay per replica, exactly as un-synchronised DDP over the reference would behave.
torch.distributed's "nccl" backend is RCCL on ROCm; "gloo" is used by the CPU tests of this module's plumbing.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


class FlatArena:
    """All parameters of `module` as views into ONE contiguous buffer, their gradients as views into another.

    autograd accumulates into the existing .grad views in place, so after backward() `self.grad` IS the flat
    gradient: it is all-reduced in a single call and consumed by one fused Adam launch; no flatten / unflatten copies.
    Call after module.to(device); do not move the module afterwards."""

    ALIGN = 4   # floats (16 bytes)

    def __init__(self, module: nn.Module, params: Optional[List[nn.Parameter]] = None):
        """params: optional explicit order of the module's trainable parameters inside the arena (bucketed all-reduce keeps
        every bucket contiguous); default = module.parameters() order"""
        self.module_order: List[nn.Parameter] = [p for p in module.parameters() if p.requires_grad]     # torch.optim.Adam(module.parameters()) order
        self.module_names: List[str] = [n for n, p in module.named_parameters() if p.requires_grad]
        if params is None:
            params = list(self.module_order)
        elif {id(p) for p in params} != {id(p) for p in module.parameters() if p.requires_grad}:
            raise ValueError("FlatArena: `params` must be a permutation of the module's trainable parameters")
        if not params:
            raise ValueError("FlatArena: module has no trainable parameters")
        dev, dt = params[0].device, params[0].dtype
        self.params: List[nn.Parameter] = params
        self.offsets: List[int] = []
        off = 0
        for p in params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("FlatArena: parameters must share one device and dtype")
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.data = torch.zeros(off, device=dev, dtype=dt)
        self.grad = torch.zeros(off, device=dev, dtype=dt)
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                view = self.data[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def zero_grad(self) -> None:
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):     # re-attach if something replaced a .grad (e.g. optimizer.zero_grad(set_to_none))
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def grad_of(self, p: nn.Parameter) -> torch.Tensor:
        return p.grad


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_sum_(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """in-place sum over ranks of the flat gradient arena; a single collective (58.7 MB at T5/L4: ~0.1-0.7 ms over xGMI)"""
    if world() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


class FusedAdam:
    """torch.optim.Adam defaults (reference models.py:398-400) as one HIP launch over the arenas.
    The gradient is pre-scaled by 1/world_size inside the kernel (mean over replicas, as DDP does)."""

    def __init__(self, arena: FlatArena, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        self.arena = arena
        self.lr, self.betas, self.eps = lr, betas, eps
        self.m = torch.zeros_like(arena.data)
        self.v = torch.zeros_like(arena.data)
        self.t = 0

    def state_dict(self) -> dict:
        """Optimizer state in torch.optim.Adam's own layout - `state[i] = {step, exp_avg, exp_avg_sq}` and `param_groups[0]['params'] =
        [0..n-1]` with i the parameter's index in `module.parameters()` order - so a checkpoint written here resumes under
        `configure_optimizers()`'s torch.optim.Adam (Lightning `optimizer_states`, reference models.py:398-400) and the reverse.  The arena's
        internal (bucket) order never reaches the file; `param_names` additionally keys every entry by parameter name."""
        a = self.arena
        slot = {id(p): (o, p) for p, o in zip(a.params, a.offsets)}
        state = {}
        for i, p in enumerate(a.module_order):
            o, _ = slot[id(p)]
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.t)), "exp_avg": self.m[o:o + n].view_as(p).clone(),
                        "exp_avg_sq": self.v[o:o + n].view_as(p).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(a.module_order)))}
        return {"state": state, "param_groups": [group], "param_names": list(a.module_names)}

    def load_state_dict(self, sd: dict) -> None:
        """accepts state_dict() of this class or of torch.optim.Adam over the same module's parameters; entries are scattered into the
        arena by parameter (matched by name when the checkpoint carries `param_names`, else by index) and shape-checked"""
        if "state" not in sd or "param_groups" not in sd:
            raise ValueError("FusedAdam.load_state_dict: expected torch.optim.Adam's layout ('state', 'param_groups')")
        a = self.arena
        slot = {id(p): o for p, o in zip(a.params, a.offsets)}
        group = sd["param_groups"][0]
        if len(sd["param_groups"]) != 1 or len(group["params"]) != len(a.module_order):
            raise ValueError(f"FusedAdam: checkpoint holds {sum(len(g['params']) for g in sd['param_groups'])} parameters in "
                             f"{len(sd['param_groups'])} group(s), this model {len(a.module_order)} in one")
        if group.get("amsgrad") or group.get("weight_decay", 0) or group.get("maximize"):
            raise ValueError("FusedAdam: amsgrad / weight_decay / maximize are not part of the reference's optimizer (models.py:399)")
        names = sd.get("param_names")
        if names is not None:
            if sorted(names) != sorted(a.module_names):
                raise ValueError("FusedAdam: checkpoint parameter names differ from this model's")
            index_of = {n: i for i, n in enumerate(names)}
            order = [index_of[n] for n in a.module_names]           # checkpoint index of our i-th parameter
        else:
            order = list(range(len(a.module_order)))
        # first pass: validate every entry (shapes, one shared step count) - a checkpoint that fails leaves the optimizer untouched
        steps = set()
        todo = []
        for i, p in enumerate(a.module_order):
            st = sd["state"].get(group["params"][order[i]])
            if st is None:                    # torch creates state lazily: a parameter that never received a gradient has none
                continue
            for key in ("exp_avg", "exp_avg_sq"):
                if tuple(st[key].shape) != tuple(p.shape):
                    raise ValueError(f"FusedAdam: {key} of '{a.module_names[i]}' has shape {tuple(st[key].shape)}, parameter {tuple(p.shape)}")
            steps.add(int(float(st["step"])))
            todo.append((p, st))
        if len(steps) > 1:
            raise ValueError(f"FusedAdam: per-parameter step counts differ ({sorted(steps)}); one shared step count is assumed")
        # second pass: commit
        self.m.zero_()
        self.v.zero_()
        for p, st in todo:
            for key, dst in (("exp_avg", self.m), ("exp_avg_sq", self.v)):
                dst[slot[id(p)]:slot[id(p)] + p.numel()].view_as(p).copy_(st[key])
        self.t = steps.pop() if steps else 0
        self.lr, self.betas, self.eps = float(group["lr"]), tuple(group["betas"]), float(group["eps"])

    def step(self, grad_scale: float = 1.0) -> None:
        from . import ops
        self.t += 1
        ops.adam_step(self.arena.data, self.arena.grad, self.m, self.v, self.lr, self.t, self.betas[0], self.betas[1], self.eps, grad_scale)


_GC_FROZEN = False


def _freeze_garbage_collector() -> None:
    """Once per process, after the second training step: everything alive now - torch, the model, the cached tables of pulpo_amd.ops - is
    moved to the collector's permanent generation (gc.freeze()), so that a full collection only walks what later steps allocate.  A step
    creates enough container objects for a full collection every few steps, and one over the whole heap takes 60 - 100 ms of HOST time:
    invisible while the GPU is 20 ms per step behind the host (160^3 fp32), a third of the step time in the bf16 modes, whose steps the host
    barely stays ahead of (measured at 192x224x160 / T6 / L5: 31.8 -> 28.0 ms per step).
    OPT-IN (a process-wide side effect a library call should not have by default: reference cycles through the frozen objects are never
    collected afterwards): PULPO_GC_FREEZE=1 in the environment (bench.py sets it) or DataParallelStepper(freeze_gc=True)."""
    global _GC_FROZEN
    if _GC_FROZEN:
        return
    import gc
    gc.collect()
    gc.freeze()
    _GC_FROZEN = True


def _gradient_buckets(model: nn.Module):
    """Order the parameters by the time their gradient completes in the backward pass of PULPo and cut the order into buckets:

        0: autoencoder.*                 complete when the gradient of the coarsest DownPath activation has been accumulated
        1: downpath.down_blocks[k >= 2]  complete when the gradient of down_blocks[1]'s output has been accumulated
        2: everything else (down_blocks[0..1]: 0.3 M of the 14.7 M parameters at T5/L4) - complete when backward() returns

    The full-resolution blocks 0 and 1 carry ~45 % of the backward time and almost no parameters, so buckets 0 and 1 (98 % of
    the bytes) travel over xGMI underneath them.  Returns (ordered params, [(start, stop) index ranges], trigger modules)."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    down = getattr(model, "downpath", None)
    if down is None or not hasattr(down, "down_blocks") or not hasattr(model, "autoencoder"):
        return [p for _, p in named], [(0, len(named))], []
    T = down.total_levels

    def bucket(name: str) -> int:
        if name.startswith("autoencoder."):
            return 0
        if name.startswith("downpath.down_blocks."):
            return 1 if int(name.split(".")[2]) >= 2 else 2
        return 2

    order = sorted(range(len(named)), key=lambda i: (bucket(named[i][0]), i))
    params = [named[i][1] for i in order]
    ids = [bucket(named[i][0]) for i in order]
    ranges, triggers = [], []
    for b, trig in ((0, down.down_blocks[T - 1]), (1, down.down_blocks[1] if T >= 3 else None), (2, None)):
        idx = [j for j, v in enumerate(ids) if v == b]
        if idx:
            ranges.append((idx[0], idx[-1] + 1))
            triggers.append(trig)
    return params, ranges, triggers


class DataParallelStepper:
    """forward + backward + gradient all-reduce + Adam for one batch of volume pairs per rank (weak scaling).

    overlap=True (default, world_size > 1): the gradient arena is reduced in the buckets of _gradient_buckets(); a bucket's
    all-reduce is issued from a tensor hook the moment autograd has finished accumulating the gradient that precedes the
    remaining, parameter-poor part of the backward pass.  RCCL orders the collective after the kernels already enqueued on
    the compute stream, and the Adam step waits for all buckets."""

    def __init__(self, model: nn.Module, lr: Optional[float] = None, overlap: bool = True, async_wgrad: bool = True,
                 freeze_gc: Optional[bool] = None, graph: Optional[bool] = None):
        self.model = model
        # graph: zero_grad + forward + backward + gradient finishing are captured into ONE HIP graph on the third step and replayed from then on
        # (step() only; the gradient exchange, the fused Adam update - its step count is a kernel argument - and the weight re-pack stay eager).
        # For steps the host barely keeps ahead of: the bf16 configurations issue ~570 launches in ~12 ms of host time per ~14 ms step.
        # None = the environment's PULPO_STEP_GRAPH == "1".  Needs a CUDA model; batches of another shape fall back to the eager step.
        self.graph = (os.environ.get("PULPO_STEP_GRAPH", "0") == "1") if graph is None else bool(graph)
        self._graph = None                       # (torch.cuda.CUDAGraph, static batch, static loss, static reg levels, signature)
        self._graph_warm = 0
        # freeze_gc: after the second step move everything alive to the garbage collector's permanent generation (see
        # _freeze_garbage_collector); None = the environment's PULPO_GC_FREEZE == "1" (off unless asked for)
        self.freeze_gc = (os.environ.get("PULPO_GC_FREEZE", "0") == "1") if freeze_gc is None else bool(freeze_gc)
        # weight gradients on a second stream (ops.ASYNC_WGRAD_STREAM): overlaps them with the BatchNorm backward passes
        dev0 = next(model.parameters()).device
        self.async_wgrad = bool(async_wgrad) and dev0.type == "cuda" and os.environ.get("PULPO_ASYNC_WGRAD", "1") != "0"
        self._side = torch.cuda.Stream(device=dev0) if self.async_wgrad else None
        self._one = None
        params, ranges, triggers = _gradient_buckets(model)
        self.arena = FlatArena(model, params)
        off = self.arena.offsets + [self.arena.numel]
        self.buckets = [(off[a], off[b]) for a, b in ranges]            # float ranges of the arena, in completion order
        self.overlap = bool(overlap) and len(self.buckets) > 1 and os.environ.get("PULPO_DP_OVERLAP", "1") != "0"
        self._works: List = []
        self._launched = 0
        self._armed = False
        self._in_backward = False
        self._steps_done = 0
        self.exchange_events: Optional[List] = None      # bench.py: [] -> (start, end) HIP events around the waits for the gradient exchange
        if self.overlap:
            for i, trig in enumerate(triggers):
                if trig is not None:
                    trig.register_forward_hook(self._make_forward_hook(i))
        self.opt = FusedAdam(self.arena, lr=lr if lr is not None else float(model.hparams.lr))
        if world() > 1:     # start from identical weights: broadcast rank 0's arena (and BN buffers)
            dist.broadcast(self.arena.data, src=0)
            from . import ops
            ops.invalidate_weight_packs()        # (parameters are views of the arena: their version counters did not move)
            for b in model.buffers():
                if b.is_floating_point():
                    dist.broadcast(b, src=0)

    # bucket i is complete once the gradient of trigger module i's OUTPUT has been accumulated (see _gradient_buckets)
    def _make_forward_hook(self, i: int):
        def fwd_hook(_module, _inputs, output):
            if self._armed and isinstance(output, torch.Tensor) and output.requires_grad:
                # (a backward pass that does not go through self.backward() exchanges its gradients in one piece, in reduce_and_update)
                output.register_hook(lambda g, i=i: self._launch_upto(i) if self._in_backward else None)
        return fwd_hook

    def _launch_upto(self, i: int) -> None:
        """issue the all-reduce of every not yet launched bucket up to and including i (buckets complete in index order)"""
        if self._launched <= i and world() > 1:
            from . import ops
            ops.join_async_wgrad()           # weight gradients of the bucket may still be running on the side stream
        while self._launched <= i:
            a, b = self.buckets[self._launched]
            if world() > 1:
                self._works.append(dist.all_reduce(self.arena.grad[a:b], op=dist.ReduceOp.SUM, async_op=True))
            self._launched += 1
        return None

    # ---- the pieces of a step.  step() below strings them together; the LightningModule hooks of pulpo_amd.models.PULPo call the same
    # ---- pieces from Lightning's own loop (training_step -> optimizer_zero_grad -> backward -> optimizer.step), so that an unchanged
    # ---- train.py runs the path bench.py times.
    def zero_grad(self) -> None:
        self.arena.zero_grad()

    def arm(self, on: bool = True) -> None:
        """the next forward pass registers the bucket triggers of the overlapped gradient exchange (multi-rank only)"""
        self._armed = bool(on) and self.overlap and world() > 1

    def backward(self, loss: torch.Tensor, direct: bool = True) -> None:
        """loss.backward() with the parameter gradients written straight into the arena, the weight gradients on the side stream and ONE
        finishing launch for all deferred parameter gradients.  direct=False: parameter gradients through autograd's AccumulateGrad (what a
        DistributedDataParallel wrapper needs to see), everything else alike."""
        from . import ops
        self._works, self._launched = [], 0
        ops._BN_TILE_PARTS.clear()                 # (BatchNorm-backward sums a data-gradient kernel left for a unit whose backward never ran)
        ops.DIRECT_PARAM_GRADS = bool(direct)      # conv / BN backward kernels add straight into the arena's .grad views
        ops.ASYNC_WGRAD_STREAM = self._side if (self.wgrad_on_side_stream() and direct) else None
        self._in_backward = True
        try:
            if self._one is None or self._one.device != loss.device or self._one.dtype != loss.dtype:
                self._one = torch.ones((), device=loss.device, dtype=loss.dtype)       # (backward() would fill a fresh one per step)
            loss.backward(self._one)
        except BaseException:
            ops.reset_param_grad_buffers(self.model)        # deferred gradient sums of an interrupted backward pass are void
            raise
        finally:
            self._in_backward = False
            self._armed = False
            ops.DIRECT_PARAM_GRADS = False
            ops.join_async_wgrad()             # (also finishes the deferred weight / bias gradients in one launch)
            ops.ASYNC_WGRAD_STREAM = None

    def reduce_and_update(self, reduced_elsewhere: bool = False) -> None:
        """gradient exchange (what is left of it) + fused Adam.  reduced_elsewhere: a DistributedDataParallel wrapper has already averaged
        the gradients (Lightning's ddp strategy): no exchange here, no 1/world scale."""
        if reduced_elsewhere:
            self.opt.step(1.0)
            self._steps_done += 1
            if self._steps_done == 2 and self.freeze_gc:
                _freeze_garbage_collector()
            return
        ev = None
        if self.exchange_events is not None and world() > 1 and self.arena.grad.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.overlap and world() > 1 and self._launched > 0:       # buckets already travelling: issue the rest, wait for all
            self._launch_upto(len(self.buckets) - 1)
            for w in self._works:
                w.wait()
        else:
            allreduce_sum_(self.arena.grad)
        self._works, self._launched = [], 0
        if ev is not None:
            ev[1].record()
            self.exchange_events.append(ev)
        self.opt.step(1.0 / world())
        self._steps_done += 1
        if self._steps_done == 2 and self.freeze_gc:
            _freeze_garbage_collector()

    def step(self, batch) -> torch.Tensor:
        if self.graph and self._graph_usable(batch):
            return self._step_graphed(batch)
        self.zero_grad()
        self.arm()
        try:
            loss = self.model.training_step(batch, 0)
            self.backward(loss)
        finally:
            self._armed = False
        self.reduce_and_update()
        return loss.detach()

    # ---- the step as a HIP graph (graph=True) -------------------------------------------------------------------------------------
    @staticmethod
    def _signature(batch):
        return tuple((tuple(t.shape), t.dtype, t.device) for t in batch)

    def _graph_usable(self, batch) -> bool:
        from . import ops
        if not all(torch.is_tensor(t) and t.is_cuda for t in batch):
            return False
        if ops.CONV_TRACE is not None or ops.HBM_TRACE is not None:
            return False                      # (per-launch event brackets: bench.py's extra steps run eagerly)
        return self._graph is None or self._graph[4] == self._signature(batch)

    def _graph_body(self, batch):
        # (no bucketed exchange from inside the backward pass: collectives stay outside the graph - reduce_and_update exchanges the arena in one piece)
        self.zero_grad()
        self._armed = False
        loss = self.model.training_step(batch, 0)
        self.backward(loss)
        return loss

    def _step_graphed(self, batch) -> torch.Tensor:
        model = self.model
        if self._graph is None:
            if self._graph_warm < 2:              # two eager steps first: weight packs, job tables and persistent scratch exist before the capture
                self._graph_warm += 1
                self.zero_grad()
                try:
                    loss = model.training_step(batch, 0)
                    self.backward(loss)
                finally:
                    self._armed = False
                self.reduce_and_update()
                return loss.detach()
            static = tuple(t.clone() for t in batch)
            cap = torch.cuda.Stream(device=static[0].device)
            cap.wait_stream(torch.cuda.current_stream())
            model._pulpo_in_graph = True          # training_step leaves the host-side NaN probe to us (it synchronises / allocates pinned memory)
            try:
                with torch.cuda.stream(cap):
                    loss_w = self._graph_body(static).detach()      # (autograd's engine streams are bound on first use: warm up ON the capture stream)
                torch.cuda.current_stream().wait_stream(cap)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap):
                    loss = self._graph_body(static)
                    levels = getattr(model, "_pulpo_last_reg_levels", None)
            finally:
                model._pulpo_in_graph = False
            self._graph = (g, static, loss.detach(), levels, self._signature(batch))
            self.reduce_and_update()              # (the capture-stream warm-up pass has left this step's gradients in the arena: it IS the step)
            return loss_w
        g, static, loss, levels, _ = self._graph
        for dst, src in zip(static, batch):
            if dst.numel() and dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        if hasattr(model, "_check_previous_step_for_nan"):
            model._check_previous_step_for_nan()
        g.replay()
        if levels is not None and hasattr(model, "_arm_nan_probe"):
            model._arm_nan_probe(levels)
        self.reduce_and_update()
        return loss

    def wgrad_on_side_stream(self) -> bool:
        """whether this step's weight gradients go to the side stream.  With async_wgrad enabled (the default) the choice follows what was
        measured: the bf16 weight gradient leaves room on its CUs and hides the BatchNorm backward passes (config 4: 14.3 against 14.9 ms per
        step), the fp32 F(2x2x2,3x3x3) weight gradient holds its CUs whole - all of the LDS, 2 x 234 registers - so that on a second stream it
        only delays the main stream's small kernels (160^3 step 29.0 against 28.7 ms in line).  PULPO_WGRAD_SIDE_STREAM=1 / 0 forces it."""
        if not self.async_wgrad:
            return False
        force = os.environ.get("PULPO_WGRAD_SIDE_STREAM", "auto")
        if force in ("0", "1"):
            return force == "1"
        from . import ops
        return ops.CONV_PRECISION == "bf16"

    def describe(self) -> dict:
        return {"overlap": bool(self.overlap and world() > 1), "async_wgrad": bool(self.async_wgrad), "wgrad_side_stream": self.wgrad_on_side_stream(),
                "buckets": len(self.buckets)}


class ArenaAdam(torch.optim.Adam):
    """What `PULPo.configure_optimizers()` returns on a GPU: a torch.optim.Adam (same param_groups, same state_dict layout, same defaults
    as the reference's optimizer, models.py:398-400) whose step is the fused HIP Adam over the flat arenas and whose zero_grad is one fill.
    It owns the DataParallelStepper the LightningModule hooks drive (`engine`); used without those hooks - `opt.zero_grad();
    loss.backward(); opt.step()` - it is plain autograd accumulation into the arena's gradient views + the fused update."""

    def __init__(self, model: nn.Module, lr: float = 1e-4, **stepper_kw):
        super().__init__([p for p in model.parameters() if p.requires_grad], lr=lr)
        self.engine = DataParallelStepper(model, lr=lr, **stepper_kw)
        self.reduced_elsewhere = False          # set by the module's hooks when a DistributedDataParallel wrapper averages the gradients

    def zero_grad(self, set_to_none: bool = True) -> None:      # noqa: ARG002  (the gradient views stay attached: one fill of the arena)
        self.engine.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]                # a learning-rate scheduler writes here
        fa = self.engine.opt
        fa.lr, fa.betas, fa.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])
        if g.get("weight_decay", 0) or g.get("amsgrad") or g.get("maximize"):
            raise ValueError("ArenaAdam: weight_decay / amsgrad / maximize are not part of the reference's optimizer (models.py:399)")
        self.engine.reduce_and_update(reduced_elsewhere=self.reduced_elsewhere)
        return loss

    def state_dict(self) -> dict:
        sd = self.engine.opt.state_dict()
        sd["param_groups"][0].update({k: v for k, v in self.param_groups[0].items() if k not in ("params",) and k in sd["param_groups"][0]})
        return sd

    def load_state_dict(self, sd: dict) -> None:
        self.engine.opt.load_state_dict(sd)
        fa = self.engine.opt
        self.param_groups[0].update(lr=fa.lr, betas=tuple(fa.betas), eps=fa.eps)


def init_from_env(backend: Optional[str] = None) -> int:
    """torchrun-style rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); returns the local rank"""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=ws)
    return local
