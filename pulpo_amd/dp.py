"""Data-parallel training over volume pairs: one process per GPU, flat fp32 parameter / gradient arenas, one RCCL
all-reduce (sum) of the gradient arena per step over xGMI, fused Adam over the arenas.

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

    def __init__(self, module: nn.Module):
        params = [p for p in module.parameters() if p.requires_grad]
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

    def step(self, grad_scale: float = 1.0) -> None:
        from . import ops
        self.t += 1
        ops.adam_step(self.arena.data, self.arena.grad, self.m, self.v, self.lr, self.t, self.betas[0], self.betas[1], self.eps, grad_scale)


class DataParallelStepper:
    """forward + backward + gradient all-reduce + Adam for one batch of volume pairs per rank (weak scaling)."""

    def __init__(self, model: nn.Module, lr: Optional[float] = None):
        self.model = model
        self.arena = FlatArena(model)
        self.opt = FusedAdam(self.arena, lr=lr if lr is not None else float(model.hparams.lr))
        if world() > 1:     # start from identical weights: broadcast rank 0's arena (and BN buffers)
            dist.broadcast(self.arena.data, src=0)
            for b in model.buffers():
                if b.is_floating_point():
                    dist.broadcast(b, src=0)

    def step(self, batch) -> torch.Tensor:
        from . import ops
        self.arena.zero_grad()
        loss = self.model.training_step(batch, 0)
        ops.DIRECT_PARAM_GRADS = True          # conv / BN backward kernels add straight into the arena's .grad views
        try:
            loss.backward()
        finally:
            ops.DIRECT_PARAM_GRADS = False
        allreduce_sum_(self.arena.grad)
        self.opt.step(1.0 / world())
        return loss.detach()


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
