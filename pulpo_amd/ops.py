"""torch.autograd.Function wrappers around the libpulpo_hip.so C ABI (include/pulpo_hip.h).

PyTorch supplies device memory, the current HIP stream and the autograd tape; every arithmetic step of the hot path
is a hand-written HIP kernel.  No operator here has a CPU path: tensors must live on a ROCm device.

Layouts: multi-channel activations are torch.channels_last_3d (N,D,H,W,C in memory) — possibly channel slices of a
wider buffer; 1- and 3-channel images / fields are plain contiguous (N,C,D,H,W) like the reference's tensors.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import List, Optional, Sequence, Tuple

import torch

from ._lib import PulpoHipError, lib

CL = torch.channels_last_3d
LRELU_SLOPE = 0.2


# ------------------------------------------------------------------------------------------------ helpers
def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_gpu(*ts: Optional[torch.Tensor], act: bool = False):
    """act: the tensors are multi-channel activations, which may live in HBM as bf16 (ACT_BF16, BASELINE configs 4-5); everything else is fp32"""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise PulpoHipError("pulpo_amd operators run on the GPU only (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.float32 and not (act and t.dtype == torch.bfloat16):
            raise PulpoHipError(f"pulpo_amd operators are fp32 (got {t.dtype})" + (" or bf16 activations" if act else ""))


def _dt(t: torch.Tensor) -> int:
    """dtype code of the typed C entry points (`*_t`): 0 fp32, 1 bf16"""
    return 1 if t.dtype == torch.bfloat16 else 0


def _esize(t: torch.Tensor) -> float:
    return 2.0 if t.dtype == torch.bfloat16 else 4.0


def _dense_grid(t: torch.Tensor) -> bool:
    """spatial dims form a dense voxel grid with a single pixel stride"""
    _, _, D, H, W = t.shape
    ps = t.stride(4)
    return (W == 1 or ps > 0) and (H == 1 or t.stride(3) == W * ps) and (D == 1 or t.stride(2) == H * W * ps)


def grid_strides(t: torch.Tensor) -> Tuple[int, int, int]:
    """(batch, pixel, channel) strides in elements of a (B,C,D,H,W) tensor whose voxels form a dense grid"""
    return t.stride(0), t.stride(4), t.stride(1)


def as_grid(t: torch.Tensor) -> torch.Tensor:
    """any 5-D tensor -> one the conv kernels can address (planar or channels-last, incl. channel slices)"""
    if _dense_grid(t) and (t.shape[1] == 1 or t.stride(1) in (1, t.shape[2] * t.shape[3] * t.shape[4] * t.stride(4))):
        return t
    return t.contiguous(memory_format=CL) if t.shape[1] > 3 else t.contiguous()


def is_cl(t: torch.Tensor) -> bool:
    B, C, D, H, W = t.shape
    return _dense_grid(t) and (t.stride(1) == 1 or C == 1) and t.stride(0) == D * H * W * t.stride(4)


def to_cl(t: torch.Tensor) -> torch.Tensor:
    """channels-last view/copy with cs == 1 and bs == V*ps (what the streaming kernels assume)"""
    if is_cl(t) and (t.shape[1] > 1 or t.stride(4) == 1):
        return t
    if t.shape[1] == 1:
        return t.contiguous()
    return t.contiguous(memory_format=CL)


def new_cl(B: int, C: int, D: int, H: int, W: int, device, dtype=torch.float32) -> torch.Tensor:
    return torch.empty((B, C, D, H, W), device=device, dtype=dtype, memory_format=CL)


def planar(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------- 2-D mode (train.py --ndims 2)
# Slices are run as depth-1 volumes through the same kernels: (B,C,H,W) -> (B,C,1,H,W); 3x3 weights sit in the middle depth slice of a
# 3x3x3 kernel (the other two slices multiply zero padding); 2-channel fields / latents get a leading zero depth channel.  The kernels
# with ndims-dependent arithmetic (warp normalisation, NCC window count, L2_reg, Jacobian, KL_nondiagonal) switch to the reference's
# 2-D form when the depth is 1.  All lifting is done with differentiable torch views / pads on small tensors.
def _is2d(t) -> bool:
    return t is not None and t.dim() == 4


def _lift(t):
    return None if t is None else t.unsqueeze(2)


def _lift_field(f):
    """(B,2,H,W) -> (B,3,1,H,W) with a zero depth component in front"""
    return None if f is None else torch.cat([torch.zeros_like(f[:, :1]), f], dim=1).unsqueeze(2)


def _unlift_field(f5):
    return f5[:, 1:, 0]


def _lift_w3(w):
    """(Cout,Cin,3,3) -> (Cout,Cin,3,3,3): taps in the middle depth slice"""
    return torch.nn.functional.pad(w.unsqueeze(2), (0, 0, 0, 0, 1, 1))


def _colsum(partials: torch.Tensor, nrow: int, ncol: int, scale: float = 1.0, into: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """column sums of [nrow][ncol] fp32 partials (double accumulation); `into` -> added to that tensor in place, returns None"""
    if into is not None:
        lib.call("pulpo_colsum", _ptr(partials), nrow, ncol, _ptr(into), scale, 1, _stream())
        return None
    out = torch.empty(ncol, device=partials.device, dtype=torch.float32)
    lib.call("pulpo_colsum", _ptr(partials), nrow, ncol, _ptr(out), scale, 0, _stream())
    return out


# When True (set by dp.DataParallelStepper, which always drives autograd through .backward() on arena-backed parameters),
# the conv / BatchNorm backward kernels add parameter gradients straight into the parameters' existing .grad storage and
# hand `None` to autograd: no per-parameter temporary, no AccumulateGrad add kernel (~170 tiny launches per step).
DIRECT_PARAM_GRADS = False


def _grad_slot(p: torch.Tensor) -> Optional[torch.Tensor]:
    if not DIRECT_PARAM_GRADS or not p.is_leaf:         # (lifted 2-D weights are derived tensors: their gradient goes through autograd)
        return None
    g = getattr(p, "grad", None)
    if g is None or not g.is_cuda or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
        return None
    return g


# ------------------------------------------------------------------------------------------------ conv 3x3x3
# Optional live kernel timing (bench.py): when CONV_TRACE is a list, every conv / wgrad launch is bracketed by HIP events
# recorded on the launch stream and (kernel name, algorithmic FLOPs, start, end) is appended.  No synchronisation here.
CONV_TRACE = None
CONV_TRACE_STRIDE = 1          # > 1: bracket only every n-th launch (a stride co-prime with the launches per step samples every layer)
CONV_TRACE_STRIDE_USED = 1     # the stride of the last timed trace (bench.py scales sampled totals with it)
_trace_rng = __import__("random").Random(0)     # which launches are bracketed: an unbiased 1-in-stride draw (a fixed stride can lock onto the
                                                # same launches of every step when the launch count per step is a multiple of it)


def _sampled() -> bool:
    return CONV_TRACE_STRIDE <= 1 or _trace_rng.random() * CONV_TRACE_STRIDE < 1.0


def _trace_begin():
    if CONV_TRACE is None or not _sampled():
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


# HBM-bound kernels (BatchNorm / LeakyReLU passes, warp, VecInt, NCC, pooling / resizing, heads, KL, regulariser, Adam): same bracket,
# algorithmic BYTES instead of FLOP
HBM_TRACE = None


_HBM_SEEN: dict = {}


def _hbm_begin(name: str):
    """brackets the first launches of every kernel class (a class with one launch per step - Adam - must not depend on the draw) and a
    1-in-stride draw of the rest"""
    trace = HBM_TRACE
    if trace is None:
        return None
    if not trace:
        _HBM_SEEN.clear()
    n = _HBM_SEEN[name] = _HBM_SEEN.get(name, 0) + 1
    if n > 2 and not _sampled():
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _hbm_end(start, name: str, nbytes: float):
    if start is None or HBM_TRACE is None:
        return
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    HBM_TRACE.append((name, nbytes, start, end))


def _trace_end(start, name: str, flops: float, nbytes: float = 0.0):
    """nbytes: algorithmic HBM bytes of the launch (operands read once + result written once)"""
    if start is None:
        return
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    CONV_TRACE.append((name, flops, start, end, nbytes))


# Operand precision of the 3x3x3 convolutions: "fp32" (the reference's arithmetic; exact-fp32 MFMA) or "bf16" (BASELINE configs
# 4-5: operands rounded to bf16 while staged, fp32 accumulation; activations, statistics, losses and gradients stay fp32).
# Layers with <= 4 reduction channels (the 2-channel image input) always run the fp32 kernel.
CONV_PRECISION = "fp32"
# bf16 ACTIVATION STORAGE (BASELINE configs 4-5, on top of bf16 operands): the multi-channel activation tensors - a ConvUnit's pre-norm
# output y and its output z, pooled / concatenated / up-sampled feature maps - and their gradients live in HBM as bf16; every kernel
# computes in fp32 (BatchNorm statistics in double) and rounds on the store.  Images, latent samples, displacement fields, losses,
# parameters, their gradients and the optimizer state stay fp32; so do the outputs of the layers with <= 4 reduction channels, which run
# the exact-fp32 kernel.  A definition of this repository (the test suite's CPU checker emulates it), "parity unpinned" against the reference.
ACT_BF16 = False
# None: the library's choice per shape (pulpo_conv3d_k3_algo); "direct" | "wino2": force that forward / data-gradient kernel
# wherever a Winograd kernel would be eligible (A/B runs and the full-size consistency test)
CONV_ALGO = None


def set_conv_precision(precision: str, activations: str = "fp32") -> None:
    """precision: operand type of the 3x3x3 convolutions; activations: storage type of the multi-channel activation tensors ("bf16" only
    together with bf16 operands)"""
    global CONV_PRECISION, ACT_BF16
    if precision not in ("fp32", "bf16"):
        raise ValueError(f"conv precision is {precision}. Not a known option.")
    if activations not in ("fp32", "bf16") or (activations == "bf16" and precision != "bf16"):
        raise ValueError(f"activation storage is {activations} with {precision} operands. Not a known option.")
    CONV_PRECISION = precision
    ACT_BF16 = activations == "bf16"


def act_dtype():
    return torch.bfloat16 if ACT_BF16 else torch.float32


# Deterministic mode (PULPO_DETERMINISTIC=1 / set_deterministic(True)): the backward kernels that add with float atomics - the weight-gradient
# flush of concurrent workgroups, the image-gradient scatter of the warp / VecInt backward, the generic trilinear-resize backward - are replaced
# by their ordered forms (`*_det` entry points: per-split slabs + an ordered sum; 64-bit fixed-point accumulation; a gather): two runs of the same
# build on the same inputs give bit-identical gradients, as the reference's CPU backward does.  Everything else in a step is
# deterministic already (two-stage reductions in fixed order).  Not covered: the `jdet` regulariser's backward (raises in this mode).
DETERMINISTIC = os.environ.get("PULPO_DETERMINISTIC", "0") == "1"


def set_deterministic(on: bool = True) -> None:
    global DETERMINISTIC
    DETERMINISTIC = bool(on)


def _use_bf16(K: int) -> bool:
    return CONV_PRECISION == "bf16" and K > 4


# Weights written behind torch's back (the fused Adam kernel, a broadcast into the parameter arena) do not move a tensor's version counter:
# whoever does that calls invalidate_weight_packs()
_WEIGHT_EPOCH = 0


def invalidate_weight_packs() -> None:
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1
    _PACK_REGISTRY.clear()


# id(packed buffer) -> (weakref to the weight, packed buffer, Cin, Cout, dgrad, kind) of every live pack of a LEAF weight that
# pulpo_conv3d_k3_pack_weights_multi can rewrite in place (kinds 0, 2, 3); a pack of another family (Winograd-x) makes the set
# unrefreshable.  Keyed by the pack itself, so every pack of a weight (both orientations, several volume shapes, both precisions) has its
# own entry; the weight is held weakly, so a discarded model's packs go with it (dead entries are pruned as new ones arrive).
_PACK_REGISTRY: dict = {}
_PACK_TABLES: dict = {}
_UNREFRESHABLE_PACKS = False
_registrations = 0


def _register_pack(w: torch.Tensor, wp: torch.Tensor, Cin: int, Cout: int, dgrad: bool, kind: int) -> None:
    global _registrations
    import weakref
    _PACK_REGISTRY[id(wp)] = (weakref.ref(w), wp, Cin, Cout, dgrad, kind)
    _registrations += 1
    if _registrations % 64 == 0:
        for k in [k for k, e in _PACK_REGISTRY.items() if e[0]() is None]:
            del _PACK_REGISTRY[k]


def refresh_weight_packs() -> None:
    """after an update of the parameters behind torch's back (the fused Adam kernel): rewrite every cached weight pack in place with ONE
    launch on the current stream, instead of forgetting them and packing layer by layer (55 small launches on the critical path of the
    next forward and backward pass).  Falls back to invalidate_weight_packs() when a cached pack is of a kind the kernel does not write."""
    global _UNREFRESHABLE_PACKS
    if _UNREFRESHABLE_PACKS or not _PACK_REGISTRY:
        _UNREFRESHABLE_PACKS = False
        invalidate_weight_packs()
        return
    jobs = []
    device = None
    for key, (wref, wp, Cin, Cout, dgrad, kind) in list(_PACK_REGISTRY.items()):
        w = wref()
        cache = getattr(w, "_pulpo_packs", None) if w is not None else None
        if cache is None or cache[0] != (w._version, w.data_ptr(), _WEIGHT_EPOCH) or not any(v is wp for v in cache[1].values()):
            del _PACK_REGISTRY[key]                  # superseded (weight gone, replaced or modified through torch): the next use packs afresh
            continue
        jobs.append((w.data_ptr(), wp.data_ptr(), Cin, Cout, int(dgrad), kind))
        device = w.device
    if not jobs:
        invalidate_weight_packs()
        return
    tkey = tuple(jobs)
    table = _PACK_TABLES.get(tkey)
    if table is None:
        import struct
        raw = b"".join(struct.pack("<QQiiii", *job) for job in tkey)
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        if len(_PACK_TABLES) > 4:
            _PACK_TABLES.clear()
        _PACK_TABLES[tkey] = table
    lib.call("pulpo_conv3d_k3_pack_weights_multi", _ptr(table), len(tkey), _stream())


def _pack_weight(w: torch.Tensor, dgrad: bool, shape=None, both: bool = False) -> torch.Tensor:
    """cached front end of _pack_weight_now: a weight is packed once per (version, orientation, kernel family, precision); `both` = also
    produce the other orientation now (training forward: the data-gradient kernel of the backward pass finds its weights ready, and a
    repeated forward - Monte-Carlo sampling, evaluation loops - packs nothing at all)"""
    if not w.is_leaf:
        # a temporary (the 2-D mode's lifted 3x3 weight, a view, a re-parametrisation): packed per call - a cache on it would die with it,
        # and one that outlived it (the refresh registry) would pin it and every pack made from it
        return _pack_weight_now(w, dgrad, shape, register=False)
    cache = getattr(w, "_pulpo_packs", None)
    ver = (w._version, w.data_ptr(), _WEIGHT_EPOCH)
    if cache is None or cache[0] != ver:
        cache = (ver, {})
        try:
            w._pulpo_packs = cache
        except AttributeError:                       # (non-leaf views cannot carry attributes on some builds: pack uncached)
            return _pack_weight_now(w, dgrad, shape, register=False)
    def key(d):
        return (d, CONV_PRECISION, CONV_ALGO, None if shape is None else tuple(shape))
    if both and key(not dgrad) not in cache[1]:
        cache[1][key(not dgrad)] = _pack_weight_now(w, not dgrad, shape)
    if key(dgrad) not in cache[1]:
        cache[1][key(dgrad)] = _pack_weight_now(w, dgrad, shape)
    return cache[1][key(dgrad)]


def _pack_weight_now(w: torch.Tensor, dgrad: bool, shape=None, register: bool = True) -> torch.Tensor:
    """GEMM-ordered copy of a (Cout, Cin, 3, 3, 3) weight for the forward (dgrad=False) or data-gradient (True) convolution.
    shape = (B, D, H, W) of the volume it will be applied to: large volumes use the Winograd-x kernel, which has its own packing
    (the returned tensor carries the choice in `_pulpo_algo`)."""
    Cout, Cin = w.shape[0], w.shape[1]
    K, N = (Cout, Cin) if dgrad else (Cin, Cout)
    global _UNREFRESHABLE_PACKS
    if _use_bf16(K):
        wp = torch.empty(lib.query("pulpo_conv3d_k3_packed_bf16_elems", K, N), device=w.device, dtype=torch.int16)
        lib.call("pulpo_conv3d_k3_pack_weight_bf16", _ptr(w.contiguous()), _ptr(wp), Cin, Cout, int(dgrad), _stream())
        wp._pulpo_algo = "bf16"
        if register:
            if w.is_contiguous():
                _register_pack(w, wp, Cin, Cout, dgrad, 3)
            else:
                _UNREFRESHABLE_PACKS = True
        return wp
    algo = lib.query("pulpo_conv3d_k3_algo", *shape, K, N) if shape is not None else 0
    if CONV_ALGO is not None and algo != 0:            # diagnostic override; only among the kernels valid for this shape
        algo = {"direct": 0, "wino2": 2, "wino3": algo}[CONV_ALGO]          # ("wino3": wherever the library's own policy picks it)
    if algo == 3:
        wp = torch.empty(lib.query("pulpo_conv3d_k3_packed_wino3_floats", K, N), device=w.device, dtype=torch.float32)
        lib.call("pulpo_conv3d_k3_pack_weight_wino3", _ptr(w.contiguous()), _ptr(wp), Cin, Cout, int(dgrad), _stream())
        wp._pulpo_algo = "wino3"
        if register:
            if w.is_contiguous():
                _register_pack(w, wp, Cin, Cout, dgrad, 4)
            else:
                _UNREFRESHABLE_PACKS = True
        return wp
    if algo == 2:
        wp = torch.empty(lib.query("pulpo_conv3d_k3_packed_wino2_floats", K, N), device=w.device, dtype=torch.float32)
        lib.call("pulpo_conv3d_k3_pack_weight_wino2", _ptr(w.contiguous()), _ptr(wp), Cin, Cout, int(dgrad), _stream())
        wp._pulpo_algo = "wino2"
        if register:
            if w.is_contiguous():
                _register_pack(w, wp, Cin, Cout, dgrad, 2)
            else:
                _UNREFRESHABLE_PACKS = True
        return wp
    wp = torch.empty(lib.query("pulpo_conv3d_k3_packed_floats", K, N), device=w.device, dtype=torch.float32)
    lib.call("pulpo_conv3d_k3_pack_weight", _ptr(w.contiguous()), _ptr(wp), Cin, Cout, int(dgrad), _stream())
    wp._pulpo_algo = "direct"
    if register:
        if w.is_contiguous():
            _register_pack(w, wp, Cin, Cout, dgrad, 0)
        else:
            _UNREFRESHABLE_PACKS = True
    return wp


def _conv_raw(x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, K: int, N: int,
              stats: Optional[torch.Tensor], coef: Optional[torch.Tensor] = None):
    """coef: eval-mode BatchNorm coefficients -> BatchNorm + LeakyReLU are applied by the convolution's store (one kernel per ConvUnit)"""
    B, _, D, H, W = _dims5(x)
    ob, op, oc = (0, 0, 1) if is_blocked(out) else grid_strides(out)
    algo = getattr(wp, "_pulpo_algo", "bf16" if wp.dtype == torch.int16 else "direct")
    if isinstance(x, _BlockedGrad) or is_blocked(x) or is_blocked(out):
        # operand and / or result in the channel-blocked layout: the F(2x2x2,3x3x3) kernel's *_kb entry (the callers have checked the kernel family)
        xt, xb_, xp_, xkb, xblk = _opnd(x)
        ot, ob_, op_, okb, oblk = _opnd(out)
        if algo != "wino3" or (not oblk and (oc != 1 or op % 4 or ob % 4 or out.data_ptr() % 16)) or (not xblk and grid_strides(x)[2] != 1):
            raise PulpoHipError("conv3d on channel-blocked tensors: F(2x2x2,3x3x3) kernel, channels-last or blocked fp32 operands only")
        t0 = _trace_begin()
        lib.call("pulpo_conv3d_k3_fwd_wino3_kb", _ptr(xt), xb_, xp_, xkb, _ptr(wp), _ptr(bias), _ptr(coef), LRELU_SLOPE, _ptr(ot), ob_, op_, okb, _ptr(stats),
                 B, D, H, W, K, N, _stream())
        _trace_end(t0, "conv3d_k3_wino3_mfma<false>", 54.0 * K * N * B * D * H * W, 4.0 * (K + N) * B * D * H * W)
        return
    xb, xp, xc = grid_strides(x)
    bf16 = algo == "bf16"
    vec_ok = xc == 1 and xp % 4 == 0 and xb % 4 == 0 and K % 4 == 0 and x.data_ptr() % 16 == 0
    if algo == "wino3":
        # F(2x2x2,3x3x3): channels-last, 16-byte aligned operand and result (the deep layers' tensors are; anything else is copied into that form)
        if not vec_ok:
            x = to_cl(x)
            xb, xp, xc = grid_strides(x)
        if oc != 1 or op % 4 or ob % 4 or out.data_ptr() % 16:
            raise PulpoHipError("conv3d (F(2x2x2,3x3x3) kernel): the result must be channels-last and 16-byte aligned")
        t0 = _trace_begin()
        lib.call("pulpo_conv3d_k3_fwd_wino3", _ptr(x), xb, xp, xc, _ptr(wp), _ptr(bias), _ptr(coef), LRELU_SLOPE, _ptr(out), ob, op, oc, _ptr(stats),
                 B, D, H, W, K, N, _stream())
        _trace_end(t0, "conv3d_k3_wino3_mfma<false>", 54.0 * K * N * B * D * H * W, 4.0 * (K + N) * B * D * H * W)
        return
    if algo == "wino2":
        if (D % 4 or D * H * W < 8000) and not vec_ok:     # (volumes below 20^3 - the 10^3 level - run on the pipelined kernel only: channels-last operand)
            x = to_cl(x)
            xb, xp, xc = grid_strides(x)
            vec_ok = True
        nscr = lib.query("pulpo_conv3d_k3_fwd_wino2_scratch_floats", B, D, H, W, K, N)
        scratch = torch.empty(nscr, device=x.device, dtype=torch.float32) if nscr else None
        t0 = _trace_begin()
        lib.call("pulpo_conv3d_k3_fwd_wino2", _ptr(x), xb, xp, xc, _ptr(wp), _ptr(bias), _ptr(coef), LRELU_SLOPE, _ptr(out), ob, op, oc, _ptr(stats),
                 _ptr(scratch), B, D, H, W, K, N, _stream())
        kname = f"conv3d_k3_wino2_mfma<{'true' if vec_ok else 'false'}>"
        if vec_ok and t0 is not None and lib.query("pulpo_conv3d_k3_wino2_pipelined", D, H, W, K, xp):
            kname = "conv3d_k3_wino2p_mfma<false>"
        _trace_end(t0, kname, 54.0 * K * N * B * D * H * W, 4.0 * (K + N) * B * D * H * W)
        return
    sfx = "_bf16" if bf16 else ""
    nscr = lib.query(f"pulpo_conv3d_k3_fwd{sfx}_scratch_floats", B, D, H, W, K, N)
    scratch = torch.empty(nscr, device=x.device, dtype=torch.float32) if nscr else None
    t0 = _trace_begin()
    if bf16:
        # the typed entry points: operand and result share one storage type (fp32: rounded while staged; bf16: stored that way)
        if x.dtype != out.dtype:
            raise PulpoHipError(f"conv3d (bf16 operands): input {x.dtype} and output {out.dtype} must share one storage type")
        if coef is None:
            lib.call("pulpo_conv3d_k3_fwd_bf16_t", _ptr(x), xb, xp, xc, _ptr(wp), _ptr(bias), _ptr(out), ob, op, oc, _dt(x), _ptr(stats), _ptr(scratch),
                     B, D, H, W, K, N, _stream())
        else:
            lib.call("pulpo_conv3d_k3_fwd_bn_lrelu_bf16_t", _ptr(x), xb, xp, xc, _ptr(wp), _ptr(bias), _ptr(coef), LRELU_SLOPE, _ptr(out), ob, op, oc,
                     _dt(x), _ptr(scratch), B, D, H, W, K, N, _stream())
    elif coef is None:
        lib.call("pulpo_conv3d_k3_fwd", _ptr(x), xb, xp, xc, _ptr(wp), _ptr(bias), _ptr(out), ob, op, oc, _ptr(stats), _ptr(scratch), B, D, H,
                 W, K, N, _stream())
    else:
        lib.call("pulpo_conv3d_k3_fwd_bn_lrelu", _ptr(x), xb, xp, xc, _ptr(wp), _ptr(bias), _ptr(coef), LRELU_SLOPE, _ptr(out), ob, op, oc,
                 _ptr(scratch), B, D, H, W, K, N, _stream())
    if t0 is not None:
        if bf16:
            name = f"conv3d_k3_mfma_bf16<{64 if N % 64 == 0 else 32},{'true' if vec_ok else 'false'}>"
        else:
            cfg = lib.query("pulpo_conv3d_k3_tile_config", K, N)
            name = f"conv3d_k3_mfma<{cfg // 1000},{cfg % 1000},{'true' if vec_ok and cfg // 1000 >= 16 else 'false'}>"
        _trace_end(t0, name, 54.0 * K * N * B * D * H * W, (_esize(x) * K + _esize(out) * N) * B * D * H * W)


# ---- deferred parameter-gradient epilogues (data-parallel stepper): inside a step the weight gradients stay in their packed scratch and
# the conv-bias gradients in their BatchNorm-backward partials until flush_param_grads() finishes ALL of them with one launch
# (pulpo_grad_finish_multi) - instead of a memset + an unpack + a column-sum launch per layer, each of which has to find room on CUs the
# persistent convolution kernels hold.  The scratch buffers persist on the parameter (they are returned all zero by the finishing kernel).
_PENDING_GRAD_JOBS: List[Tuple[int, int, int, int, int, int]] = []      # (src ptr, dst ptr, kind, a, b, c)
_PENDING_KEEPALIVE: List[torch.Tensor] = []
_JOB_TABLES: dict = {}


def _persistent_buffer(owner: torch.Tensor, name: str, numel: int, zero: bool) -> torch.Tensor:
    buf = getattr(owner, name, None)
    if buf is None or buf.numel() != numel or buf.device != owner.device:
        buf = (torch.zeros if zero else torch.empty)(numel, device=owner.device, dtype=torch.float32)
        setattr(owner, name, buf)
    return buf


def _pending_src(buf: torch.Tensor) -> bool:
    """is this persistent buffer already the source of a deferred job of the current step?"""
    p_ = buf.data_ptr()
    return any(job[0] == p_ for job in _PENDING_GRAD_JOBS)


def flush_param_grads() -> None:
    """finish every deferred weight / bias gradient on the current stream (callers have joined the weight-gradient stream first)"""
    if not _PENDING_GRAD_JOBS:
        return
    key = tuple(_PENDING_GRAD_JOBS)
    table = _JOB_TABLES.get(key)
    if table is None:
        import struct
        raw = b"".join(struct.pack("<QQiiii", *job) for job in key)
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(_PENDING_KEEPALIVE[0].device)
        if len(_JOB_TABLES) > 8:
            _JOB_TABLES.clear()
        _JOB_TABLES[key] = table
    lib.call("pulpo_grad_finish_multi", _ptr(table), len(key), _stream())
    _PENDING_GRAD_JOBS.clear()
    _PENDING_KEEPALIVE.clear()


def reset_param_grad_buffers(module: Optional[torch.nn.Module] = None) -> None:
    """after an interrupted step: forget the pending jobs and drop the persistent scratch buffers (they may hold partial sums)"""
    _PENDING_GRAD_JOBS.clear()
    _PENDING_KEEPALIVE.clear()
    _BN_TILE_PARTS.clear()
    if module is not None:
        for p_ in module.parameters():
            for name in ("_pulpo_wgrad_scratch", "_pulpo_dbias_part", "_pulpo_dbias_part_in", "_pulpo_heads_part"):
                if hasattr(p_, name):
                    delattr(p_, name)


# ---- the gradient of a ConvUnit's pre-norm tensor in a channel-BLOCKED layout (round 5).  dy leaves the BatchNorm backward for exactly two readers,
# the unit's data- and weight-gradient convolution (aten::convolution_backward, src/network_blocks.py:23): its layout is nobody else's business.
# Channels-last, a staging item of the F(2x2x2,3x3x3) data-gradient kernel gathers 32 useful bytes from each of four 128-byte voxel lines per
# 8-channel chunk and the kernel waits on those lines (32 -> 32 at 160^3: 0.94 ms, 0.80 with every tap a cache hit); as [C / 8][B][D][H][W][8] the four
# taps are 128 consecutive bytes: 0.80 ms (scripts/blocked_probe.py).  Used where BOTH readers run their F(2x2x2) kernel on a volume of at least
# BLOCKED_DY_MIN_VOXELS voxels (below ~64^3 the tensors live in the caches and the layouts tie).  PULPO_BLOCKED_DY=0: channels-last everywhere.
BLOCKED_DY = os.environ.get("PULPO_BLOCKED_DY", "1") != "0"
BLOCKED_DY_MIN_VOXELS = int(os.environ.get("PULPO_BLOCKED_DY_MIN_VOXELS", str(64 ** 3)))
BLOCKED_DY_HITS = 0              # gradients written in the blocked layout so far (tests look at it)


class _BlockedGrad:
    """fp32 operand in the layout [C / 8][B][D][H][W][8]: element (b, v, c) at  b * bs + (c // 8) * kb + v * ps + c % 8  of `buf`"""
    __slots__ = ("buf", "shape", "bs", "ps", "kb", "dtype", "device")

    def __init__(self, B, C, D, H, W, dev, buf=None):
        self.buf = torch.empty(B * C * D * H * W, device=dev, dtype=torch.float32) if buf is None else buf
        self.shape = (B, C, D, H, W)
        self.ps, self.bs, self.kb = 8, D * H * W * 8, B * D * H * W * 8
        self.dtype, self.device = torch.float32, self.buf.device

    @classmethod
    def of(cls, t6: torch.Tensor):
        """the operand view of a blocked 6-D tensor (is_blocked)"""
        B, C, D, H, W = blocked_shape(t6)
        return cls(B, C, D, H, W, t6.device, buf=t6)

    def to_cl(self) -> torch.Tensor:
        """the same values as a channels-last (B, C, D, H, W) tensor (tests, fallbacks)"""
        B, C, D, H, W = self.shape
        return self.buf.view(C // 8, B, D, H, W, 8).permute(1, 0, 5, 2, 3, 4).reshape(B, C, D, H, W).contiguous(memory_format=CL)


# ---- blocked ACTIVATIONS between the ConvUnits of a ConvSequence (round 5).  The output z of every unit but the last has two readers, the next
# unit's convolution (forward pass) and weight gradient (src/network_blocks.py:40-46) - where both run their F(2x2x2,3x3x3) kernel it is produced as a
# contiguous fp32 tensor of shape (C / 8, B, D, H, W, 8) (`is_blocked`: six dimensions), and autograd carries its gradient in the same shape: the next
# unit's data-gradient kernel writes dz blocked, this unit's BatchNorm backward reads it that way.  A forward hook on such a ConvUnit sees the 6-D
# tensor (blocked_to_cl() gives the usual view).
# OFF by default (PULPO_BLOCKED_Z=1 / ops.BLOCKED_Z = True turns it on): alone on the machine the 32 -> 32 forward convolution at 160^3 gains 15 - 18 %
# (0.95 -> 0.78 ms, profiles/r5_blocked_probe.txt), but inside the training step it already runs at 0.80 ms on channels-last input - the activation has
# just been written and its tail still sits in the 256 MB Infinity Cache - and gains 3 - 5 %, while the data-gradient kernel that now stores a blocked
# dz loses 2.5 % and the weight gradient 1.5 %: 26.53 against 26.55 ms per step, same box (profiles/r5_blocked_z_ab.txt).  The blocked dy above, whose
# producer and consumers are all streaming / staging kernels of the backward pass, keeps 0.14 ms.
BLOCKED_Z = os.environ.get("PULPO_BLOCKED_Z", "0") == "1"
BLOCKED_Z_MIN_VOXELS = int(os.environ.get("PULPO_BLOCKED_Z_MIN_VOXELS", str(64 ** 3)))
BLOCKED_Z_HITS = 0               # activations written in the blocked layout so far (tests look at it)


def is_blocked(t) -> bool:
    return isinstance(t, torch.Tensor) and t.dim() == 6


def blocked_shape(t: torch.Tensor):
    Cb, B, D, H, W, e = t.shape
    if e != 8:
        raise PulpoHipError(f"a six-dimensional activation must be channel-blocked (C / 8, B, D, H, W, 8), got {tuple(t.shape)}")
    return B, Cb * 8, D, H, W


def blocked_to_cl(t: torch.Tensor) -> torch.Tensor:
    """(C / 8, B, D, H, W, 8) -> channels-last (B, C, D, H, W); differentiable"""
    B, C, D, H, W = blocked_shape(t)
    return t.permute(1, 0, 5, 2, 3, 4).reshape(B, C, D, H, W).contiguous(memory_format=CL)


def cl_to_blocked(t: torch.Tensor) -> torch.Tensor:
    B, C, D, H, W = t.shape
    return t.permute(0, 2, 3, 4, 1).reshape(B, D, H, W, C // 8, 8).permute(4, 0, 1, 2, 3, 5).contiguous()


def _opnd(t):
    """(tensor to take the pointer of, batch stride, pixel stride, block stride, blocked?) of a convolution operand / result"""
    if isinstance(t, _BlockedGrad):
        return t.buf, t.bs, t.ps, t.kb, True
    if t.dim() == 6:
        B, C, D, H, W = blocked_shape(t)
        return t, D * H * W * 8, 8, B * D * H * W * 8, True
    b, p, c = grid_strides(t)
    return t, b, p, 8, False


def _dims5(t):
    return t.shape if isinstance(t, _BlockedGrad) else (blocked_shape(t) if t.dim() == 6 else tuple(t.shape))


def blocked_z_wanted(x, unit_weight, next_weight, training: bool) -> bool:
    """should the ConvUnit with `unit_weight`, applied to x, hand its output to the unit with `next_weight` in the blocked layout?"""
    if not (BLOCKED_Z and training and torch.is_grad_enabled() and CONV_PRECISION == "fp32" and not ACT_BF16 and next_weight.requires_grad
            and isinstance(x, torch.Tensor) and x.dim() in (5, 6) and x.is_cuda):
        return False
    B, _, D, H, W = _dims5(x)
    C, Cn = unit_weight.shape[0], next_weight.shape[0]
    if C % 8 or next_weight.shape[1] != C or D * H * W < BLOCKED_Z_MIN_VOXELS or 4 * B * max(C, Cn) * D * H * W >= 2 ** 31:
        return False
    if lib.query("pulpo_conv3d_k3_wgrad_algo", B, D, H, W, C, Cn, 1) != 3:
        return False
    shape = (B, D, H, W)
    return (getattr(_pack_weight(next_weight, False, shape=shape, both=True), "_pulpo_algo", "") == "wino3"
            and getattr(_pack_weight(next_weight, True, shape=shape), "_pulpo_algo", "") == "wino3")


def _blocked_dy_ok(x, y, weight, wpt, need_dx: bool, need_dw: bool) -> bool:
    B, Cin, D, H, W = _dims5(x)
    Cout = weight.shape[0]
    if not (BLOCKED_DY and need_dx and y.dtype == torch.float32 and Cout % 8 == 0 and D * H * W >= BLOCKED_DY_MIN_VOXELS
            and getattr(wpt, "_pulpo_algo", "") == "wino3" and 4 * B * Cout * D * H * W < 2 ** 31):
        return False
    if need_dw:
        if _use_bf16(Cin) or x.dtype != torch.float32 or lib.query("pulpo_conv3d_k3_wgrad_algo", B, D, H, W, Cin, Cout, 1) != 3:
            return False
        if not is_blocked(x):
            xb, xp, xc = grid_strides(x)
            if xc != 1 or xp % 4 or xb % 4 or Cin % 4 or x.data_ptr() % 16:
                return False
    return True


def _wgrad_raw(x: torch.Tensor, dy: torch.Tensor, Cin: int, Cout: int, into: Optional[torch.Tensor] = None,
               owner: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """weight gradient; `into` -> accumulated into that (Cout,Cin,3,3,3) tensor in place, returns None.  With `owner` (the weight parameter,
    data-parallel stepper) the accumulation is DEFERRED to flush_param_grads(): the packed sums stay in the parameter's persistent scratch."""
    B, _, D, H, W = _dims5(x)
    if is_blocked(x) and (_use_bf16(Cin) or lib.query("pulpo_conv3d_k3_wgrad_algo", B, D, H, W, Cin, Cout, 1) != 3):
        x = blocked_to_cl(x)                         # (a blocked operand outside the F(2x2x2,3x3x3) kernel's shapes: a copy - BLOCKED_Z switched between passes)
    deferred = into is not None and owner is not None
    dw = into if into is not None else torch.empty((Cout, Cin, 3, 3, 3), device=x.device, dtype=torch.float32)
    nscr = lib.query("pulpo_conv3d_k3_wgrad_scratch_floats", Cin, Cout)
    scratch = _persistent_buffer(owner, "_pulpo_wgrad_scratch", nscr, zero=True) if deferred else torch.empty(nscr, device=x.device, dtype=torch.float32)
    xblk = is_blocked(x)
    xt, xb, xp, xkb, _ = _opnd(x)
    xc = 1 if xblk else grid_strides(x)[2]
    blocked = isinstance(dy, _BlockedGrad) or xblk
    if blocked and not isinstance(dy, _BlockedGrad):
        dy = to_cl(dy.float())
    db, dp, dc = (dy.bs, dy.ps, 1) if isinstance(dy, _BlockedGrad) else grid_strides(dy)
    t0 = _trace_begin()
    sfx = "_bf16" if _use_bf16(Cin) else ""
    det = ()
    if DETERMINISTIC:
        # one zero-initialised copy of the packed sums per spatial split of the grid (<= ~110 MB, transient), added up in fixed order
        nslab = lib.query("pulpo_conv3d_k3_wgrad_det_slabs", Cin, Cout)
        slabs = torch.empty(nslab * nscr, device=x.device, dtype=torch.float32)
        det = (_ptr(slabs), nslab)
    if blocked:
        # (_blocked_dy_ok has checked: fp32 operands, channels-last x, the F(2x2x2,3x3x3) weight-gradient kernel takes the shape)
        dyt, _, _, dkb, _ = _opnd(dy)
        lib.call("pulpo_conv3d_k3_wgrad_kb", _ptr(xt), xb, xp, xkb, _ptr(dyt), db, dp, dkb, _ptr(dw), 2 if deferred else int(into is not None), _ptr(scratch),
                 *(det if det else (None, 0)), B, D, H, W, Cin, Cout, _stream())
    elif sfx:
        if x.dtype != dy.dtype:                      # (one storage type per launch; a mixed pair - a user's fp32 input to a bf16-storage unit - is rare)
            x, dy = x.float(), dy.float()
            xb, xp, xc = grid_strides(x)
            db, dp, dc = grid_strides(dy)
        lib.call("pulpo_conv3d_k3_wgrad_bf16_det_t" if det else "pulpo_conv3d_k3_wgrad_bf16_t", _ptr(x), xb, xp, xc, _ptr(dy), db, dp, dc, _dt(x), _ptr(dw),
                 2 if deferred else int(into is not None), _ptr(scratch), *det, B, D, H, W, Cin, Cout, _stream())
    else:
        if x.dtype != torch.float32 or dy.dtype != torch.float32:
            x, dy = x.float(), dy.float()
            xb, xp, xc = grid_strides(x)
            db, dp, dc = grid_strides(dy)
        lib.call("pulpo_conv3d_k3_wgrad_det" if det else "pulpo_conv3d_k3_wgrad", _ptr(x), xb, xp, xc, _ptr(dy), db, dp, dc, _ptr(dw),
                 2 if deferred else int(into is not None), _ptr(scratch), *det, B, D, H, W, Cin, Cout, _stream())
    if deferred and not _pending_src(scratch):
        # (ONE finishing job per scratch: a unit applied twice in a step - shared weights, two forward passes - has accumulated both weight
        #  gradients into the same packed sums by the time the job runs)
        _PENDING_GRAD_JOBS.append((scratch.data_ptr(), dw.data_ptr(), 0, Cin, Cout, (Cout + 63) // 64 * 64))
        _PENDING_KEEPALIVE.append(scratch)
    if t0 is not None:
        name = "conv3d_k3_wgrad_bf16"
        if blocked:
            name = "conv3d_k3_wgrad_w3x"
        elif not sfx:
            vec = (xc == 1 and xp % 4 == 0 and xb % 4 == 0 and Cin % 4 == 0 and x.data_ptr() % 16 == 0 and dc == 1 and dp % 4 == 0 and db % 4 == 0
                   and Cout % 4 == 0 and dy.data_ptr() % 16 == 0)
            name = ("conv3d_k3_wgrad_mfma", "conv3d_k3_wgrad_wino", "conv3d_k3_wgrad_w2", "conv3d_k3_wgrad_w3x")[lib.query("pulpo_conv3d_k3_wgrad_algo", B, D, H, W, Cin, Cout, int(vec))]
        _trace_end(t0, name + ("" if deferred else "(+memset,unpack)"), 54.0 * Cin * Cout * B * D * H * W, ((4 if blocked else _esize(x)) * Cin + (4 if blocked else _esize(dy)) * Cout) * B * D * H * W)
    return None if into is not None else dw


# Weight gradients off the critical path.  Inside a data-parallel step the weight gradient goes straight into the gradient arena and
# nothing reads it before the all-reduce, so it need not finish before the backward pass moves on: it is queued on a second HIP
# stream BEHIND this unit's data-gradient kernel, where the matrix-bound persistent kernel (one workgroup per CU) runs next to the
# HBM-bound BatchNorm / LeakyReLU backward passes of the preceding unit on the main stream.  Joined before the gradient exchange
# (dp.DataParallelStepper).  None = disabled (plain autograd use: gradients are returned on the caller's stream).
ASYNC_WGRAD_STREAM = None


def _wgrad_on_side_stream(x, dy, Cin, Cout, slot_w, owner):
    main = torch.cuda.current_stream()
    side = ASYNC_WGRAD_STREAM
    side.wait_stream(main)                       # after everything queued so far: dy, this unit's data gradient, zero_grad
    with torch.cuda.stream(side):
        _wgrad_raw(x, dy, Cin, Cout, into=slot_w, owner=owner)
    x.record_stream(side)                        # keep both operands' memory out of the allocator's hands until the side stream is done
    (dy.buf if isinstance(dy, _BlockedGrad) else dy).record_stream(side)


def join_async_wgrad():
    """make the current stream wait for every weight gradient queued on the side stream, then finish the deferred parameter gradients"""
    if ASYNC_WGRAD_STREAM is not None:
        torch.cuda.current_stream().wait_stream(ASYNC_WGRAD_STREAM)
    flush_param_grads()


# ---- BatchNorm-backward reduction inside the data-gradient convolution.  In a ConvSequence unit u consumes z = lrelu(bn(y)) of unit u-1,
# and the gradient dz that unit u-1's backward receives is exactly what unit u's data-gradient kernel stores: that kernel's epilogue has
# every dz element in registers, so with y of unit u-1 read alongside it also delivers the per-tile sums (sum dbn, sum dbn * xhat) that
# unit u-1 would otherwise compute in a pass of its own over dz and y (pulpo_bn_lrelu_bwd_reduce).  The forward pass hands (y, coef) of
# the producer to the consumer on the tensor z itself (`_pulpo_bn_src`); the backward pass of the consumer leaves the sums here, keyed by
# the producer's y, and the producer takes them only if the gradient it is given IS that kernel's output, untouched (same storage, same
# version: a gradient that autograd accumulated from several consumers is a different tensor or carries a bumped version).
BN_REDUCE_IN_DGRAD = os.environ.get("PULPO_BN_REDUCE_IN_DGRAD", "1") != "0"      # (A/B switch)
_BN_TILE_PARTS: dict = {}


def _dgrad_with_bn_reduction(bn_src, x, dy, wpt, dx, K: int, N: int) -> bool:
    algo = getattr(wpt, "_pulpo_algo", "")
    if bn_src is None or not BN_REDUCE_IN_DGRAD or algo not in ("wino2", "wino3"):
        return False
    y_prev, coef_prev = bn_src
    B, _, D, H, W = _dims5(dy)
    dyt, db, dp, dkb, blocked = _opnd(dy)
    dc = 1 if blocked else grid_strides(dy)[2]
    dxt, ob, op, okb, oblk = _opnd(dx)
    oc = 1 if oblk else grid_strides(dx)[2]
    if (blocked or oblk) and algo != "wino3":
        return False
    yb, yp, yc = grid_strides(y_prev)
    # (the C entry point also needs the gradient operand vectorisable: channels-last, 16-byte aligned, K % 4 == 0 - checked here so that a
    #  consumer unit with an odd channel count falls back to the separate reduction pass instead of raising in the middle of backward)
    vec_ok = blocked or (dc == 1 and dp % 4 == 0 and db % 4 == 0 and K % 4 == 0 and dy.data_ptr() % 16 == 0)
    if (not vec_ok or tuple(y_prev.shape) != tuple(_dims5(dx)) or oc != 1 or yc != 1 or op % 4 or ob % 4 or yp % 4 or yb % 4 or dx.data_ptr() % 16
            or y_prev.data_ptr() % 16 or not lib.query("pulpo_conv3d_k3_dgrad_wino2_bnred_ok", B, D, H, W, K, N)):
        return False
    ntile = lib.query("pulpo_conv3d_k3_stat_tiles", B, D, H, W)
    part = torch.empty(ntile * 2 * N, device=dy.device, dtype=torch.float32)
    t0 = _trace_begin()
    if blocked or oblk:
        lib.call("pulpo_conv3d_k3_dgrad_wino3_bnred_kb", _ptr(dyt), db, dp, dkb, _ptr(wpt), _ptr(dxt), ob, op, okb, _ptr(y_prev), yb, yp, _ptr(coef_prev),
                 LRELU_SLOPE, _ptr(part), B, D, H, W, K, N, _stream())
    else:
        lib.call(f"pulpo_conv3d_k3_dgrad_{algo}_bnred", _ptr(dy), db, dp, dc, _ptr(wpt), _ptr(dx), ob, op, _ptr(y_prev), yb, yp, _ptr(coef_prev), LRELU_SLOPE,
                 _ptr(part), B, D, H, W, K, N, _stream())
    kname = "conv3d_k3_wino3_mfma<true>" if algo == "wino3" else "conv3d_k3_wino2_mfma<true>"
    if algo == "wino2" and t0 is not None and lib.query("pulpo_conv3d_k3_wino2_pipelined", D, H, W, K, dp):
        kname = "conv3d_k3_wino2p_mfma<true>"
    _trace_end(t0, kname, 54.0 * K * N * B * D * H * W, 4.0 * (K + 2 * N) * B * D * H * W)
    _BN_TILE_PARTS[y_prev.data_ptr()] = (part, ntile, coef_prev.data_ptr(), dx.data_ptr(), dx._version, tuple(dx.shape), tuple(dx.stride()))
    return True


def _take_bn_tile_parts(y: torch.Tensor, coef: torch.Tensor, dz: torch.Tensor):
    entry = _BN_TILE_PARTS.pop(y.data_ptr(), None)
    if entry is None:
        return None
    part, ntile, coef_ptr, ptr, version, shape, stride = entry
    if coef.data_ptr() != coef_ptr or dz.data_ptr() != ptr or dz._version != version or tuple(dz.shape) != shape or tuple(dz.stride()) != stride:
        return None
    return part, ntile


_TLS = threading.local()


# ---- producers that write straight into a slice of a wider channels-last buffer (round 5).  PULPoEncoder concatenates the up-sampled
# feedback path's output with the DownPath activation of its level (torch.cat([feedback, down_activation], 1), components/pulpo.py:252): two
# strided copy kernels per level and step (111 us at 80^3).  DownPath allocates the concatenation's buffer up front and its last ConvUnit, like
# the feedback path's later, writes its output into its channel range - every kernel addresses operands through explicit pixel strides, so the
# slices are ordinary operands, and the concatenation is the buffer itself (cat_prewritten).
def _take_out_slot(B, C, D, H, W, dev, dtype):
    slot = getattr(_TLS, "out_slot", None)
    _TLS.out_slot = None
    if slot is None:
        return None
    buf, off = slot
    if (buf.dim() != 5 or tuple(buf.shape[2:]) != (D, H, W) or buf.shape[0] != B or buf.dtype != dtype or buf.device != dev or off < 0 or off + C > buf.shape[1]
            or not is_cl(buf) or off % 8 or buf.shape[1] % 8):
        return None
    return buf[:, off:off + C]


CAT_PREWRITTEN_HITS = 0          # concatenations that cost nothing so far (tests look at it)


class _CatPrewritten(torch.autograd.Function):
    """cat([a, b], 1) where a and b ARE the two channel ranges of `buf`: the result is the buffer, the gradient is split by position"""

    @staticmethod
    def forward(ctx, a, b, buf):
        ctx.ca = a.shape[1]
        return buf.as_strided(buf.shape, buf.stride(), buf.storage_offset())

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.ca], g[:, ctx.ca:], None


def cat_channels(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], dim=1); free when both were produced into the two channel ranges of one buffer (`_pulpo_cat` tags)"""
    ta, tb = getattr(a, "_pulpo_cat", None), getattr(b, "_pulpo_cat", None)
    if ta is not None and tb is not None and ta[0] is tb[0]:
        buf = ta[0]
        es = buf.element_size()
        if (ta[1] == 0 and tb[1] == a.shape[1] and a.shape[1] + b.shape[1] == buf.shape[1] and a.data_ptr() == buf.data_ptr()
                and b.data_ptr() == buf.data_ptr() + es * tb[1] and a.stride() == buf.stride() and b.stride() == buf.stride()
                and a.shape[2:] == buf.shape[2:] and b.shape[2:] == buf.shape[2:] and a.shape[0] == buf.shape[0] == b.shape[0]):
            global CAT_PREWRITTEN_HITS
            CAT_PREWRITTEN_HITS += 1
            return _CatPrewritten.apply(a, b, buf)
    return torch.cat([a, b], dim=1)


class _ConvBNLReLU(torch.autograd.Function):
    """ConvUnit: Conv3d(k3,p1,bias) -> BatchNorm3d -> LeakyReLU(0.2)   (reference src/network_blocks.py:22-26)"""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, num_batches_tracked, training: bool, momentum: float, eps: float,
                bn_src=None, pool_after: bool = False, pool_only: bool = False, blocked_out: bool = False):
        _require_gpu(x, act=True)
        _require_gpu(weight, bias, gamma, beta)
        ctx.bn_src = bn_src
        # x: (B, C, D, H, W), or the blocked output (C / 8, B, D, H, W, 8) of the previous ConvUnit of the sequence (is_blocked)
        ctx.dx_blocked = is_blocked(x)
        if not ctx.dx_blocked:
            x = as_grid(x)
        B, Cin, D, H, W = _dims5(x)
        Cout = weight.shape[0]
        dev = x.device
        wp = _pack_weight(weight, dgrad=False, shape=(B, D, H, W), both=bool(training and ctx.needs_input_grad[0]))
        # storage types (ACT_BF16): z - what the next operator reads - is bf16; the pre-norm tensor y is bf16 when the bf16-operand kernel
        # produces it and fp32 behind the exact-fp32 kernel of the <= 4-channel input layers; the kernels take operand and result in ONE type
        half_conv = ACT_BF16 and wp._pulpo_algo == "bf16"
        if ctx.dx_blocked and (wp._pulpo_algo != "wino3" or half_conv or not training):
            x = blocked_to_cl(x)                     # (a blocked activation in front of another kernel family: the producer's check and this call disagree - a copy)
        if x.dtype != (torch.bfloat16 if half_conv else torch.float32):
            x = x.to(torch.bfloat16 if half_conv else torch.float32)
        ydt = torch.bfloat16 if half_conv else torch.float32
        zdt = act_dtype()
        y = new_cl(B, Cout, D, H, W, dev, ydt)
        coef = torch.empty(8 * Cout, device=dev, dtype=torch.float32)      # [4][C] floats + [2][C] doubles
        if training:
            ntile = lib.query("pulpo_conv3d_k3_fwd_bf16_stat_tiles" if wp._pulpo_algo == "bf16" else "pulpo_conv3d_k3_stat_tiles", B, D, H, W)
            stats = torch.empty(ntile * 2 * Cout, device=dev, dtype=torch.float32)
            _conv_raw(x, wp, bias, y, Cin, Cout, stats)
            nsd = lib.query("pulpo_bn_fwd_finalize_scratch_doubles", ntile, Cout)
            scratch = torch.empty(nsd, device=dev, dtype=torch.float64) if nsd else None
            lib.call("pulpo_bn_fwd_finalize", _ptr(stats), ntile, Cout, float(B * D * H * W), _ptr(gamma), _ptr(beta), _ptr(running_mean),
                     _ptr(running_var), _ptr(num_batches_tracked), momentum, eps, _ptr(coef), _ptr(scratch), _stream())
        else:
            lib.call("pulpo_bn_eval_coef", _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), eps, Cout, _ptr(coef), _stream())
            if not any(ctx.needs_input_grad) and ydt == zdt:
                # inference: conv + folded BatchNorm + LeakyReLU in one kernel, the pre-norm tensor is never written
                zo = _take_out_slot(B, Cout, D, H, W, dev, zdt)
                if zo is not None:
                    y = zo
                _conv_raw(x, wp, bias, y, Cin, Cout, None, coef=coef)
                return y
            _conv_raw(x, wp, bias, y, Cin, Cout, None)
        # pool_only: nobody reads the un-pooled activation (DownPath levels above the first latent level: only AvgPool(z) goes on) - it is not written
        pool_only = bool(pool_only and pool_after and lib.query("pulpo_bn_lrelu_apply_pool2_ok", Cout, y.stride(4), Cout, Cout))
        blocked_out = bool(blocked_out and training and not pool_after and not pool_only and zdt == torch.float32 and ydt == torch.float32 and Cout % 8 == 0)
        z = None if (pool_only or blocked_out) else _take_out_slot(B, Cout, D, H, W, dev, zdt)
        if z is None and not pool_only and not blocked_out:
            z = new_cl(B, Cout, D, H, W, dev, zdt)
        pooled = None
        nbytes = (_esize(y) + (0 if pool_only else (2.0 if zdt == torch.bfloat16 else 4.0))) * Cout * B * D * H * W             # read y, write z
        if blocked_out:
            # the next ConvUnit of the sequence reads z through its F(2x2x2,3x3x3) kernels only (blocked_z_wanted): [Cout / 8][B][D][H][W][8]
            global BLOCKED_Z_HITS
            BLOCKED_Z_HITS += 1
            z = torch.empty((Cout // 8, B, D, H, W, 8), device=dev, dtype=torch.float32)
            t0 = _hbm_begin("bn_lrelu_apply")
            lib.call("pulpo_bn_lrelu_apply_kb", _ptr(y), y.stride(4), _ptr(z), 8, B * D * H * W * 8, _ptr(coef), B * D * H * W, Cout, LRELU_SLOPE, _stream())
            _hbm_end(t0, "bn_lrelu_apply", nbytes)
        elif pool_only or (pool_after and lib.query("pulpo_bn_lrelu_apply_pool2_ok", Cout, y.stride(4), z.stride(4), Cout)):
            # the caller pools this output next (DownPath): z and AvgPool(z) from one read of y; avg_pool2_skip() picks the pooled tensor up
            pooled = new_cl(B, Cout, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2, dev, zdt)
            t0 = _hbm_begin("bn_lrelu_apply")
            lib.call("pulpo_bn_lrelu_apply_pool2_t", _ptr(y), _dt(y), y.stride(4), _ptr(z), _dt(pooled), z.stride(4) if z is not None else Cout, _ptr(pooled),
                     pooled.stride(4), _ptr(coef), B, D, H, W, Cout, LRELU_SLOPE, _stream())
            _hbm_end(t0, "bn_lrelu_apply", nbytes)
        else:
            t0 = _hbm_begin("bn_lrelu_apply")
            lib.call("pulpo_bn_lrelu_apply_t", _ptr(y), _dt(y), y.stride(4), _ptr(z), _dt(z), z.stride(4), _ptr(coef), B * D * H * W, Cout, LRELU_SLOPE,
                     _stream())
            _hbm_end(t0, "bn_lrelu_apply", nbytes)
        ctx.save_for_backward(x, weight, y, coef)
        ctx.training = training
        ctx.params = (weight, bias, gamma, beta)      # for DIRECT_PARAM_GRADS (their .grad slots)
        _TLS.produced = (y, coef, pooled)            # read back by conv_bn_lrelu (the Function returns tensors only)
        ctx.pooled_out = pooled is not None
        ctx.pool_only = pool_only
        if pool_only:
            ctx.set_materialize_grads(False)
            _TLS.pool_only_done = True
            return pooled
        if pooled is not None:
            # (round 5) z AND AvgPool(z) are outputs of this node: their gradients arrive together, and the backward pass forms
            # dz = gz + avg_pool_backward(gpooled) per element inside the BatchNorm-backward passes instead of writing it
            ctx.set_materialize_grads(False)
            return z, pooled
        return z

    @staticmethod
    def backward(ctx, dz, dpool=None):
        if ctx.pool_only:                            # (the node's only output is the pooled tensor)
            dz, dpool = None, dz
        x, weight, y, coef = ctx.saved_tensors
        B, Cin, D, H, W = _dims5(x)
        Cout = weight.shape[0]
        dev = x.device
        npix = B * D * H * W
        nblk = lib.query("pulpo_bn_bwd_blocks", npix, Cout)
        NG = 15                                      # inputs of forward()
        dz_blk = is_blocked(dz)                      # the gradient of a blocked activation arrives blocked (the next unit's data-gradient kernel wrote it so)
        if dz_blk and (dz.dtype != torch.float32 or not dz.is_contiguous() or tuple(blocked_shape(dz)) != (B, Cout, D, H, W)):
            dz, dz_blk = blocked_to_cl(dz.float()), False
        pooled_src = None                            # (gpool, gskip or None): dz = gskip + avg_pool_backward(gpool), never written
        if dpool is not None:
            gp = to_cl(dpool)
            gz = dz
            if gz is not None and gz.dtype != gp.dtype:
                gz = gz.to(gp.dtype)
            grp = 4 * int(_esize(gp))
            skip_ok = gz is None
            if gz is not None:
                sb, sp, sc = grid_strides(gz)
                skip_ok = _dense_grid(gz) and sc == 1 and sb == D * H * W * sp and sp % 4 == 0 and gz.data_ptr() % grp == 0
            if (Cout % 4 == 0 and Cout // 4 <= 256 and skip_ok and gp.stride(4) % 4 == 0 and gp.data_ptr() % grp == 0 and y.stride(1) == 1
                    and y.stride(4) % 4 == 0 and y.data_ptr() % (4 * int(_esize(y))) == 0 and _dense_grid(y) and POOLED_BN_BACKWARD):
                pooled_src = (gp, gz)
            else:                                    # shapes the fused passes do not take: the gradient as a tensor, then the plain path
                gin = new_cl(B, Cout, D, H, W, dev, gp.dtype)
                if skip_ok and gz is not None:
                    lib.call("pulpo_avgpool2_bwd_t", _ptr(gp), gp.stride(4), _ptr(gz), grid_strides(gz)[1], _ptr(gin), gin.stride(4), _dt(gp), B, D, H, W, Cout, _stream())
                else:
                    lib.call("pulpo_avgpool2_bwd_t", _ptr(gp), gp.stride(4), None, 0, _ptr(gin), gin.stride(4), _dt(gp), B, D, H, W, Cout, _stream())
                    if gz is not None:
                        gin = gin + gz
                dz = gin
        elif dz is None:
            return (None,) * NG
        tiles = None
        if pooled_src is not None:
            gp, gz = pooled_src
            part = torch.empty(nblk * 2 * Cout, device=dev, dtype=torch.float32)
            t0 = _hbm_begin("avgpool2_bwd_bnred")
            lib.call("pulpo_avgpool2_bwd_bnred_t", _ptr(gp), gp.stride(4), _ptr(gz), grid_strides(gz)[1] if gz is not None else 0, None, 0, _dt(gp), _ptr(y), _dt(y),
                     y.stride(4), _ptr(coef), LRELU_SLOPE, _ptr(part), B, D, H, W, Cout, _stream())
            # read the pooled gradient, the skip gradient and y (the summed gradient is not written)
            _hbm_end(t0, "avgpool2_bwd_bnred", Cout * (_esize(gp) * (gp.numel() // Cout + (npix if gz is not None else 0)) + _esize(y) * npix))
        else:
            if not dz_blk:
                dz = to_cl(dz)
            # first pass (sum dbn, sum dbn * xhat): already done by the epilogue of the data-gradient convolution that PRODUCED dz, if that was
            # the ConvUnit behind this one (see _BN_TILE_PARTS); else a pass of its own over dz and y
            tiles = _take_bn_tile_parts(y, coef, dz)
            if tiles is None and dz_blk:             # (the separate reduction pass reads channels-last: autograd summed several gradients of z - a copy)
                dz, dz_blk = blocked_to_cl(dz), False
        if tiles is None and pooled_src is None:
            part = torch.empty(nblk * 2 * Cout, device=dev, dtype=torch.float32)
            t0 = _hbm_begin("bn_lrelu_bwd_reduce")
            lib.call("pulpo_bn_lrelu_bwd_reduce_t", _ptr(dz), _dt(dz), dz.stride(4), _ptr(y), _dt(y), y.stride(4), _ptr(coef), npix, Cout, LRELU_SLOPE,
                     _ptr(part), _stream())
            _hbm_end(t0, "bn_lrelu_bwd_reduce", (_esize(dz) + _esize(y)) * Cout * npix)                # read dz, y
        w_p, b_p, g_p, be_p = ctx.params
        slot_w, slot_b, slot_g, slot_be = (_grad_slot(t) if need else None
                                           for t, need in zip((w_p, b_p, g_p, be_p), ctx.needs_input_grad[1:5]))
        direct_bn = slot_g is not None and slot_be is not None
        tot = None if direct_bn else torch.empty(2 * Cout, device=dev, dtype=torch.float32)           # dbeta | dgamma
        totd = torch.empty(2 * Cout, device=dev, dtype=torch.float64)          # mean(dbn) | mean(dbn * xhat), kept in double
        # eval-mode BatchNorm is a fixed affine map (dy = scale * dbn): the batch means do not enter
        fin_out = (_ptr(slot_be if direct_bn else tot), _ptr(slot_g) if direct_bn else ctypes.c_void_p(tot.data_ptr() + 4 * Cout), int(direct_bn), _ptr(totd))
        rows, nrow = (part, nblk) if tiles is None else tiles       # (pooled_src: `part` from the pooled first pass above)
        nsd = lib.query("pulpo_bn_bwd_finalize_scratch_doubles", nrow, Cout)
        scratch = torch.empty(nsd, device=dev, dtype=torch.float64) if nsd else None
        lib.call("pulpo_bn_bwd_finalize", _ptr(rows), nrow, Cout, _ptr(coef), float(npix), int(ctx.training), *fin_out, _ptr(scratch), _stream())
        # The input layer (image pair -> 32 channels at full resolution): nobody asks for its data gradient, so dy has ONE reader - the weight
        # gradient, which then forms it per element while staging (pulpo_conv3d_k3_wgrad_bn) instead of a pass that reads dz and y and writes dy
        # (0.29 ms at 160^3 x 32 channels).  PULPO_FUSE_INPUT_WGRAD=0: the separate pass (A/B switch).
        if (FUSE_INPUT_WGRAD and pooled_src is None and Cin <= 2 and not ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not DETERMINISTIC and y.dtype == torch.float32
                and Cout % 4 == 0 and is_cl(y) and (dz_blk or (is_cl(dz) and dz.stride(4) % 4 == 0)) and y.stride(4) % 4 == 0 and x.dtype == torch.float32):
            return _ConvBNLReLU._backward_input_layer(ctx, dz, x, weight, y, coef, totd, tot, direct_bn, (slot_w, slot_b), (w_p, b_p))
        # the data-gradient weights now (cached pack): their kernel family decides dy's layout
        wpt = _pack_weight(weight, dgrad=True, shape=(B, D, H, W)) if ctx.needs_input_grad[0] else None
        blocked = (_blocked_dy_ok(x, y, weight, wpt, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
                   and (pooled_src is not None or dz_blk or (is_cl(dz) and dz.stride(4) % 4 == 0)) and is_cl(y) and y.stride(4) % 4 == 0)
        if blocked:
            global BLOCKED_DY_HITS
            BLOCKED_DY_HITS += 1
            dy = _BlockedGrad(B, Cout, D, H, W, dev)
        else:
            dy = new_cl(B, Cout, D, H, W, dev, y.dtype)        # (the gradient of the pre-norm tensor is stored like the tensor)
        defer_b = DIRECT_PARAM_GRADS and ctx.needs_input_grad[2] and slot_b is not None
        part2 = _persistent_buffer(b_p, "_pulpo_dbias_part", nblk * Cout, zero=False) if defer_b else None
        if defer_b and _pending_src(part2):          # this unit has already run a backward pass in this step: its partials are still waiting
            defer_b = False                          # for flush_param_grads() - this pass takes the immediate path into the same slot
        if not defer_b:
            part2 = torch.empty(nblk * Cout, device=dev, dtype=torch.float32)
        t0 = _hbm_begin("bn_lrelu_bwd_apply")
        if pooled_src is not None:
            gp, gz = pooled_src
            if blocked:
                lib.call("pulpo_bn_lrelu_bwd_apply_pooled_kb_t", _ptr(gp), gp.stride(4), _ptr(gz), grid_strides(gz)[1] if gz is not None else 0, _dt(gp), _ptr(y),
                         y.stride(4), _ptr(coef), _ptr(totd), _ptr(dy.buf), dy.ps, dy.kb, LRELU_SLOPE, _ptr(part2), B, D, H, W, Cout, _stream())
            else:
                lib.call("pulpo_bn_lrelu_bwd_apply_pooled_t", _ptr(gp), gp.stride(4), _ptr(gz), grid_strides(gz)[1] if gz is not None else 0, _dt(gp), _ptr(y), _dt(y),
                         y.stride(4), _ptr(coef), _ptr(totd), _ptr(dy), dy.stride(4), LRELU_SLOPE, _ptr(part2), B, D, H, W, Cout, _stream())
            _hbm_end(t0, "bn_lrelu_bwd_apply", Cout * (_esize(gp) * (gp.numel() // Cout + (npix if gz is not None else 0)) + 2 * _esize(y) * npix))
        else:
            if blocked or dz_blk:
                dzs = (8, npix * 8) if dz_blk else (dz.stride(4), 8)
                dys = (_ptr(dy.buf), dy.ps, dy.kb) if blocked else (_ptr(dy), dy.stride(4), 8)
                lib.call("pulpo_bn_lrelu_bwd_apply_kb_t", _ptr(dz), _dt(dz), *dzs, _ptr(y), y.stride(4), _ptr(coef), _ptr(totd), *dys, npix, Cout, LRELU_SLOPE,
                         _ptr(part2), _stream())
            else:
                lib.call("pulpo_bn_lrelu_bwd_apply_t", _ptr(dz), _dt(dz), dz.stride(4), _ptr(y), _dt(y), y.stride(4), _ptr(coef), _ptr(totd), _ptr(dy), dy.stride(4),
                         npix, Cout, LRELU_SLOPE, _ptr(part2), _stream())
            _hbm_end(t0, "bn_lrelu_bwd_apply", (_esize(dz) + 2 * _esize(y)) * Cout * npix)                # read dz, y; write dy
        defer_w = ctx.needs_input_grad[1] and slot_w is not None and ASYNC_WGRAD_STREAM is not None
        if defer_b:
            _PENDING_GRAD_JOBS.append((part2.data_ptr(), slot_b.data_ptr(), 1, nblk, Cout, 0))
            _PENDING_KEEPALIVE.append(part2)
        dbias = _colsum(part2, nblk, Cout, into=slot_b) if (ctx.needs_input_grad[2] and not defer_b) else None
        dbeta, dgamma = (None, None) if direct_bn else (tot[:Cout], tot[Cout:])
        dw = _wgrad_raw(x, dy, Cin, Cout, into=slot_w, owner=w_p if slot_w is not None else None) if (ctx.needs_input_grad[1] and not defer_w) else None
        dx = None
        if ctx.needs_input_grad[0]:
            # (the bf16-operand kernel takes operand and result in one storage type; every other kernel is fp32)
            dyc = dy if (wpt._pulpo_algo == "bf16" or dy.dtype == torch.float32) else dy.float()
            dx_blk = ctx.dx_blocked and wpt._pulpo_algo == "wino3" and dyc.dtype == torch.float32
            if dx_blk:                                   # the input was a blocked activation: its gradient in the same layout, straight from the kernel
                dx = torch.empty((Cin // 8, B, D, H, W, 8), device=dev, dtype=torch.float32)
            elif is_blocked(x) or not (x.is_contiguous() and Cin <= 3):
                dx = new_cl(B, Cin, D, H, W, dev, dyc.dtype)
            else:
                dx = torch.empty_like(x, dtype=dyc.dtype)
            if not _dgrad_with_bn_reduction(ctx.bn_src, x, dyc, wpt, dx, Cout, Cin):
                _conv_raw(dyc, wpt, None, dx, Cout, Cin, None)
            if ctx.dx_blocked and not dx_blk:
                dx = cl_to_blocked(dx.float())
        if defer_w:
            _wgrad_on_side_stream(x, dy, Cin, Cout, slot_w, w_p)
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


def _backward_input_layer(ctx, dz, x, weight, y, coef, totd, tot, direct_bn, slots, params):
    slot_w, slot_b = slots
    w_p, b_p = params
    B, Cin, D, H, W = x.shape
    Cout = weight.shape[0]
    dev = x.device
    nrow = lib.query("pulpo_conv3d_k3_wgrad_bn_rows", B, D, H, W, Cout)
    defer_b = DIRECT_PARAM_GRADS and ctx.needs_input_grad[2] and slot_b is not None
    part2 = _persistent_buffer(b_p, "_pulpo_dbias_part_in", nrow * Cout, zero=False) if defer_b else None
    if defer_b and _pending_src(part2):
        defer_b = False
    if not defer_b:
        part2 = torch.empty(nrow * Cout, device=dev, dtype=torch.float32)
    deferred = slot_w is not None and w_p is not None
    dw = slot_w if slot_w is not None else torch.empty((Cout, Cin, 3, 3, 3), device=dev, dtype=torch.float32)
    nscr = lib.query("pulpo_conv3d_k3_wgrad_scratch_floats", Cin, Cout)
    scratch = _persistent_buffer(w_p, "_pulpo_wgrad_scratch", nscr, zero=True) if deferred else torch.empty(nscr, device=dev, dtype=torch.float32)
    xb, xp, xc = grid_strides(x)

    def launch():
        t0 = _trace_begin()
        if is_blocked(dz):                           # (the gradient of a blocked activation: the next unit's data-gradient kernel wrote it that way)
            lib.call("pulpo_conv3d_k3_wgrad_bn_kb", _ptr(x), xb, xp, xc, _ptr(dz), D * H * W * 8, 8, B * D * H * W * 8, _ptr(y), y.stride(0), y.stride(4), _ptr(coef),
                     _ptr(totd), LRELU_SLOPE, _ptr(dw), 2 if deferred else int(slot_w is not None), _ptr(scratch), _ptr(part2), B, D, H, W, Cin, Cout, _stream())
        else:
            lib.call("pulpo_conv3d_k3_wgrad_bn", _ptr(x), xb, xp, xc, _ptr(dz), _dt(dz), dz.stride(0), dz.stride(4), _ptr(y), y.stride(0), y.stride(4), _ptr(coef),
                     _ptr(totd), LRELU_SLOPE, _ptr(dw), 2 if deferred else int(slot_w is not None), _ptr(scratch), _ptr(part2), B, D, H, W, Cin, Cout, _stream())
        _trace_end(t0, "conv3d_k3_wgrad_smallc(+bn backward)" + ("" if deferred else "(+memset,unpack)"), 54.0 * Cin * Cout * B * D * H * W,
                   (4.0 * Cin + (_esize(dz) + 4.0) * Cout) * B * D * H * W)

    side = ASYNC_WGRAD_STREAM
    if side is not None and deferred:
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            launch()
        for t in (x, dz, y, coef, totd):
            t.record_stream(side)
    else:
        launch()
    if deferred and not _pending_src(scratch):
        _PENDING_GRAD_JOBS.append((scratch.data_ptr(), dw.data_ptr(), 0, Cin, Cout, (Cout + 63) // 64 * 64))
        _PENDING_KEEPALIVE.append(scratch)
    if defer_b:
        _PENDING_GRAD_JOBS.append((part2.data_ptr(), slot_b.data_ptr(), 1, nrow, Cout, 0))
        _PENDING_KEEPALIVE.append(part2)
    dbias = _colsum(part2, nrow, Cout, into=slot_b) if (ctx.needs_input_grad[2] and not defer_b) else None
    dbeta, dgamma = (None, None) if direct_bn else (tot[:Cout], tot[Cout:])
    return None, (None if slot_w is not None else dw), dbias, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None


_ConvBNLReLU._backward_input_layer = staticmethod(_backward_input_layer)
FUSE_INPUT_WGRAD = os.environ.get("PULPO_FUSE_INPUT_WGRAD", "1") != "0"
# the gradient of a pooled ConvUnit output formed inside both BatchNorm-backward passes instead of written (A/B switch: "0" materialises it)
POOLED_BN_BACKWARD = os.environ.get("PULPO_POOLED_BN_BACKWARD", "1") != "0"


def conv_bn_lrelu(x, weight, bias, gamma, beta, running_mean, running_var, training=True, momentum=0.1, eps=1e-5, num_batches_tracked=None,
                  pool_after: bool = False, out=None, pool_only: bool = False, blocked_out: bool = False):
    """ConvUnit forward.  In training mode running_mean / running_var / num_batches_tracked are updated in place by the kernel.
    pool_after: the caller applies avg_pool2_skip() to the result next - where the shapes allow, the pooled tensor is produced by the same
    pass that writes the result and waits on it (`_pulpo_pooled`).
    out: (buffer, first channel) - the result is written into that channel range of a wider channels-last buffer and returned as its slice,
    tagged `_pulpo_cat` (see cat_channels); ignored where the shapes do not fit.
    x may be the blocked output (C / 8, B, D, H, W, 8) of the previous ConvUnit of a sequence; blocked_out: produce this unit's output in that form
    (network_blocks.ConvSequence asks ops.blocked_z_wanted first)."""
    if _is2d(x):
        return conv_bn_lrelu(_lift(x), _lift_w3(weight), bias, gamma, beta, running_mean, running_var, training, momentum, eps,
                             num_batches_tracked).squeeze(2)
    _TLS.out_slot = out
    src = getattr(x, "_pulpo_bn_src", None)          # x is the untouched output of another ConvUnit: (y, coef, version at production)
    bn_src = src[:2] if (src is not None and src[2] == x._version and training and torch.is_grad_enabled()) else None
    # pool_only (with pool_after): the caller reads ONLY AvgPool(result) - where the fused pass is available the un-pooled tensor is not written and
    # the call returns (None, pooled); otherwise (result, None) as without the flag
    want_tuple = bool(pool_only)
    pool_only = bool(pool_only and pool_after and training and torch.is_grad_enabled())
    _TLS.pool_only_done = False
    z = _ConvBNLReLU.apply(x, weight, bias, gamma, beta, running_mean, running_var, num_batches_tracked, bool(training), float(momentum),
                           float(eps), bn_src, bool(pool_after), pool_only, bool(blocked_out))
    pooled_out = None
    if isinstance(z, tuple):
        z, pooled_out = z
    elif getattr(_TLS, "pool_only_done", False):         # the pooled tensor alone came back
        _TLS.produced = None
        _TLS.out_slot = None
        _TLS.pool_only_done = False
        return None, z
    produced = getattr(_TLS, "produced", None)
    _TLS.produced = None
    _TLS.out_slot = None
    if produced is not None:
        z._pulpo_bn_src = (produced[0], produced[1], z._version)
        if pooled_out is not None:
            z._pulpo_pooled = (pooled_out, z._version, True)      # (an output of the same autograd node: avg_pool2_skip hands it out as it is)
    if out is not None and z.dim() == 5 and z.data_ptr() == out[0].data_ptr() + out[0].element_size() * out[1] and z.stride() == out[0].stride():
        z._pulpo_cat = (out[0], out[1])
    return (z, None) if want_tuple else z


class _Conv3dK3(torch.autograd.Function):
    """bare padded 3x3x3 convolution (no norm / activation); used by tests and by VelocityField-like heads"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x, act=True)
        _require_gpu(weight, bias)
        x = as_grid(x.float())                        # (a bare convolution - VelocityField depth 1, tests - keeps fp32 storage)
        B, Cin, D, H, W = x.shape
        Cout = weight.shape[0]
        y = new_cl(B, Cout, D, H, W, x.device)
        _conv_raw(x, _pack_weight(weight, False, shape=(B, D, H, W)), bias, y, Cin, Cout, None)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, Cin, D, H, W = x.shape
        Cout = weight.shape[0]
        dy = as_grid(dy.float())
        dw = _wgrad_raw(x, dy, Cin, Cout) if ctx.needs_input_grad[1] else None
        db = dy.sum(dim=(0, 2, 3, 4)) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x) if (x.is_contiguous() and Cin <= 3) else new_cl(B, Cin, D, H, W, x.device)
            _conv_raw(dy, _pack_weight(weight, True, shape=(B, D, H, W)), None, dx, Cout, Cin, None)
        return dx, dw, db


def conv3d_k3(x, weight, bias=None):
    if _is2d(x):
        return conv3d_k3(_lift(x), _lift_w3(weight), bias).squeeze(2)
    return _Conv3dK3.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------------ 1x1x1 heads
class _Heads(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, Wt, bias, eps, nout: int, params=None):
        """params: the leaf parameters Wt / bias were assembled from, as ((tensor, first row of Wt, rows), ...) for the weights and
        ((tensor, first element of bias, elements), ...) for the biases: inside the data-parallel stepper their gradients are finished by
        flush_param_grads() straight from the backward kernel's partial rows (no column sum, no split, no AccumulateGrad add per parameter)"""
        _require_gpu(h, act=True)
        _require_gpu(Wt, bias, eps)
        ctx.params = params
        h = to_cl(h)
        B, C, D, H, W = h.shape
        V = D * H * W
        dev = h.device
        outs = [torch.empty((B, 3, D, H, W), device=dev, dtype=torch.float32) for _ in range(1 if nout == 3 else 3)]
        epsc = planar(eps) if eps is not None else None
        t0 = _hbm_begin("heads_fwd")
        lib.call("pulpo_heads_fwd_t", _ptr(h), _dt(h), h.stride(4), _ptr(Wt), _ptr(bias), _ptr(epsc), _ptr(outs[0]), _ptr(outs[1]) if nout == 6 else None,
                 _ptr(outs[2]) if nout == 6 else None, nout, B, V, C, _stream())
        _hbm_end(t0, "heads_fwd", B * V * (_esize(h) * C + 4.0 * (12 if nout == 6 else 3)))       # read h (+ eps), write mu / sigma / z (or the field)
        ctx.nout = nout
        ctx.save_for_backward(h, Wt, epsc, outs[1] if nout == 6 else None)
        return outs[0] if nout == 3 else tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        h, Wt, eps, sigma = ctx.saved_tensors
        nout = ctx.nout
        B, C, D, H, W = h.shape
        V = D * H * W
        dev = h.device
        g = [planar(t) if t is not None else None for t in gs] + [None, None]
        if nout == 3 and g[0] is None:
            g[0] = torch.zeros((B, 3, D, H, W), device=dev)
        dh = new_cl(B, C, D, H, W, dev, h.dtype)
        nblk = lib.query("pulpo_heads_bwd_blocks", B, V, C)
        rowlen = nout * C + nout
        # inside the stepper: the partial rows go to a persistent buffer (stable address: the finishing launch's job table is cached) and
        # flush_param_grads() adds their column sums to the parameters' .grad; a head applied twice in a step takes the immediate path
        slots = None
        if ctx.params is not None and DIRECT_PARAM_GRADS and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            wparts, bparts = ctx.params
            slots = [(_grad_slot(t), off, n) for t, off, n in wparts] + [(_grad_slot(t), off, n) for t, off, n in bparts]
            if not all(sl is not None for sl, _, _ in slots):
                slots = None
        part = _persistent_buffer(ctx.params[0][0][0], "_pulpo_heads_part", nblk * rowlen, zero=False) if slots is not None else None
        if part is None or _pending_src(part):
            slots = None
            part = torch.empty(nblk * rowlen, device=dev, dtype=torch.float32)
        t0 = _hbm_begin("heads_bwd")
        lib.call("pulpo_heads_bwd_t", _ptr(h), _dt(h), h.stride(4), _ptr(Wt), _ptr(g[0]), _ptr(g[1]), _ptr(g[2]), _ptr(eps), _ptr(sigma), _ptr(dh),
                 dh.stride(4), _ptr(part), nout, B, V, C, _stream())
        _hbm_end(t0, "heads_bwd", B * V * (2 * _esize(h) * C + 4.0 * (15 if nout == 6 else 3)))   # read h, the output gradients (+ eps, sigma), write dh
        if slots is not None:
            nw = len(ctx.params[0])
            for k, (sl, off, n) in enumerate(slots):
                col0, ncol = (off * C, n * C) if k < nw else (nout * C + off, n)
                _PENDING_GRAD_JOBS.append((part.data_ptr() + 4 * col0, sl.data_ptr(), 1, nblk, ncol, rowlen))
            _PENDING_KEEPALIVE.append(part)
            return dh, None, None, None, None, None
        tot = _colsum(part, nblk, rowlen)
        return dh, tot[: nout * C].view(nout, C), tot[nout * C:], None, None, None


def mu_sigma_sample(h, w_mu, b_mu, w_sigma, b_sigma, eps):
    """MuSigmaBlock + sampler: returns (mu, sigma, z) planar (B,zdim,D,H,W); eps=None -> z = mu.
    w_*: (zdim, C, 1, 1, 1) conv weights (reference src/network_blocks.py:54-57).  zdim == 3 on volumes / 2 on slices (zdim = ndims,
    models.py:88) is ONE launch of the head kernel; any other zdim runs it over groups of three latent channels, the last group padded with
    zero rows (mu 0, sigma softplus(0), noise 0), which are dropped again."""
    C, zdim = w_mu.shape[1], w_mu.shape[0]
    if _is2d(h) and zdim == 2:
        # 2-D: two latent channels -> rows (0, mu_y, mu_x) / (0, sigma_y, sigma_x) of the three-channel head kernel; the padded channel
        # (mu 0, sigma softplus(0), noise 0 -> sample 0) is dropped again
        z1, zb = w_mu.new_zeros(1, C), b_mu.new_zeros(1)
        Wt = torch.cat([z1, w_mu.reshape(2, C), z1, w_sigma.reshape(2, C)], dim=0)
        bias = torch.cat([zb, b_mu, zb, b_sigma], dim=0)
        mu, sigma, z = _Heads.apply(_lift(h), Wt, bias, _lift_field(eps), 6)
        return _unlift_field(mu), _unlift_field(sigma), _unlift_field(z)
    if zdim != 3 or _is2d(h):
        two_d = _is2d(h)
        h5 = _lift(h) if two_d else h
        e5 = (_lift(eps) if two_d else eps) if eps is not None else None
        mus, sigmas, zs = [], [], []
        for c0 in range(0, zdim, 3):
            n = min(3, zdim - c0)
            pad_w, pad_b = w_mu.new_zeros(3 - n, C), b_mu.new_zeros(3 - n)
            Wt = torch.cat([w_mu[c0:c0 + n].reshape(n, C), pad_w, w_sigma[c0:c0 + n].reshape(n, C), pad_w], dim=0)
            bias = torch.cat([b_mu[c0:c0 + n], pad_b, b_sigma[c0:c0 + n], pad_b], dim=0)
            eg = None
            if e5 is not None:
                eg = e5[:, c0:c0 + n]
                if n < 3:
                    eg = torch.cat([eg, eg.new_zeros((eg.shape[0], 3 - n) + tuple(eg.shape[2:]))], dim=1)
            mu, sigma, z = _Heads.apply(h5, Wt, bias, eg, 6)
            mus.append(mu[:, :n]); sigmas.append(sigma[:, :n]); zs.append(z[:, :n])
        out = [torch.cat(t, dim=1) if len(t) > 1 else t[0].contiguous() for t in (mus, sigmas, zs)]
        return tuple(o.squeeze(2) for o in out) if two_d else tuple(out)
    Wt = torch.cat([w_mu.reshape(3, C), w_sigma.reshape(3, C)], dim=0)
    bias = torch.cat([b_mu, b_sigma], dim=0)
    return _Heads.apply(h, Wt, bias, eps, 6, (((w_mu, 0, 3), (w_sigma, 3, 3)), ((b_mu, 0, 3), (b_sigma, 3, 3))))


def conv1x1_to3(h, w, b):
    """Conv3d(C, 3, kernel_size=1) with planar output (reference src/network_blocks.py:81)"""
    if _is2d(h):
        C = w.shape[1]
        Wt = torch.cat([w.new_zeros(1, C), w.reshape(2, C)], dim=0)
        return _unlift_field(_Heads.apply(_lift(h), Wt, torch.cat([b.new_zeros(1), b]), None, 3))
    return _Heads.apply(h, w.reshape(3, w.shape[1]), b, None, 3, (((w, 0, 3),), ((b, 0, 3),)))


# ------------------------------------------------------------------------------------------------ resampling
class _AvgPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, act=True)
        x = to_cl(x)
        B, C, D, H, W = x.shape
        out = new_cl(B, C, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2, x.device, x.dtype) if C > 1 else \
            torch.empty((B, 1, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2), device=x.device, dtype=x.dtype)
        t0 = _hbm_begin("avgpool2_fwd")
        lib.call("pulpo_avgpool2_fwd_t", _ptr(x), x.stride(4), _ptr(out), out.stride(4), _dt(x), B, D, H, W, C, _stream())
        _hbm_end(t0, "avgpool2_fwd", _esize(x) * C * (x.numel() // C + out.numel() // C))
        ctx.shape = (B, C, D, H, W)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, D, H, W = ctx.shape
        g = to_cl(g)
        gin = new_cl(B, C, D, H, W, g.device, g.dtype) if C > 1 else torch.empty((B, 1, D, H, W), device=g.device, dtype=g.dtype)
        lib.call("pulpo_avgpool2_bwd_t", _ptr(g), g.stride(4), None, 0, _ptr(gin), gin.stride(4), _dt(g), B, D, H, W, C, _stream())
        return gin


def avg_pool2(x):
    """AvgPool3d(kernel 2, stride 2, ceil_mode=True)"""
    if _is2d(x):
        return avg_pool2(_lift(x)).squeeze(2)
    return _AvgPool2.apply(x)


class _AvgPool2Skip(torch.autograd.Function):
    """(x, pool(x)) for an activation that is pooled AND used as a skip connection (DownPath: components/pulpo.py:52-59): the backward pass
    forms both gradients' sum in ONE pass (pulpo_avgpool2_bwd_add) instead of a pooling backward plus autograd's accumulation add on a
    strided slice of the concatenation's gradient (101 us at 80^3 x 64 channels at 1 TB/s in torch's generic strided kernel)"""

    @staticmethod
    def forward(ctx, x, ready=None, bn_y=None, bn_coef=None):
        """bn_y / bn_coef: x is the untouched output of a ConvUnit, these are its pre-norm tensor and coefficient block - the backward pass then
        also delivers that unit's BatchNorm-backward partial sums (see _BN_TILE_PARTS)"""
        _require_gpu(x, act=True)
        ctx.set_materialize_grads(False)
        ctx.bn = (bn_y, bn_coef) if (bn_y is not None and BN_REDUCE_IN_DGRAD) else None
        xc = to_cl(x)
        B, C, D, H, W = xc.shape
        if ready is not None:                          # AvgPool(x) already written by the pass that wrote x (conv_bn_lrelu(pool_after=True))
            out = ready
        else:
            out = new_cl(B, C, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2, x.device, x.dtype) if C > 1 else \
                torch.empty((B, 1, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2), device=x.device, dtype=x.dtype)
            lib.call("pulpo_avgpool2_fwd_t", _ptr(xc), xc.stride(4), _ptr(out), out.stride(4), _dt(xc), B, D, H, W, C, _stream())
        ctx.shape = (B, C, D, H, W)
        # (the alias keeps x's exact strides - view_as() would renumber the batch stride of a B = 1 tensor, and torch.cat then no longer
        #  recognises the channels-last layout of its inputs)
        return x.as_strided(x.shape, x.stride(), x.storage_offset()), out

    @staticmethod
    def backward(ctx, gskip, gpool):
        B, C, D, H, W = ctx.shape
        if gpool is None:
            return gskip, None, None, None
        g = to_cl(gpool)
        if gskip is not None and gskip.dtype != g.dtype:
            gskip = gskip.to(g.dtype)
        gin = new_cl(B, C, D, H, W, g.device, g.dtype) if C > 1 else torch.empty((B, 1, D, H, W), device=g.device, dtype=g.dtype)
        grp = 4 * int(_esize(g))                      # bytes of a four-channel group
        skip_ok = False
        if gskip is not None:
            sb, sp, sc = grid_strides(gskip)
            skip_ok = _dense_grid(gskip) and sc == 1 and sb == D * H * W * sp and C > 1 and sp % 4 == 0 and gskip.data_ptr() % grp == 0
        if ctx.bn is not None and C % 4 == 0 and C // 4 <= 256 and (gskip is None or skip_ok) and g.stride(4) % 4 == 0 and g.data_ptr() % grp == 0:
            # the producing ConvUnit's first BatchNorm-backward pass rides along: this kernel has every element of its gradient in registers
            y, coef = ctx.bn
            if tuple(y.shape) == (B, C, D, H, W) and y.stride(1) == 1 and y.stride(4) % 4 == 0 and y.data_ptr() % (4 * int(_esize(y))) == 0 and _dense_grid(y):
                nblk = lib.query("pulpo_bn_bwd_blocks", B * D * H * W, C)
                part = torch.empty(nblk * 2 * C, device=g.device, dtype=torch.float32)
                t0 = _hbm_begin("avgpool2_bwd_bnred")
                lib.call("pulpo_avgpool2_bwd_bnred_t", _ptr(g), g.stride(4), _ptr(gskip), grid_strides(gskip)[1] if gskip is not None else 0, _ptr(gin),
                         gin.stride(4), _dt(g), _ptr(y), _dt(y), y.stride(4), _ptr(coef), LRELU_SLOPE, _ptr(part), B, D, H, W, C, _stream())
                # read the pooled gradient, the skip gradient and y, write the summed gradient
                _hbm_end(t0, "avgpool2_bwd_bnred", C * (_esize(g) * (g.numel() // C + (2 if gskip is not None else 1) * B * D * H * W) + _esize(y) * B * D * H * W))
                _BN_TILE_PARTS[y.data_ptr()] = (part, nblk, coef.data_ptr(), gin.data_ptr(), gin._version, tuple(gin.shape), tuple(gin.stride()))
                return gin, None, None, None
        if skip_ok:
            t0 = _hbm_begin("avgpool2_bwd_add")
            lib.call("pulpo_avgpool2_bwd_t", _ptr(g), g.stride(4), _ptr(gskip), grid_strides(gskip)[1], _ptr(gin), gin.stride(4), _dt(g), B, D, H, W, C, _stream())
            _hbm_end(t0, "avgpool2_bwd_add", _esize(g) * C * (g.numel() // C + 2 * B * D * H * W))
            return gin, None, None, None
        lib.call("pulpo_avgpool2_bwd_t", _ptr(g), g.stride(4), None, 0, _ptr(gin), gin.stride(4), _dt(g), B, D, H, W, C, _stream())
        return (gin if gskip is None else gskip + gin), None, None, None


def avg_pool2_skip(x):
    """(x, AvgPool(x)) where x goes on to other consumers as well; see _AvgPool2Skip"""
    if _is2d(x):
        return x, avg_pool2(x)
    ready = getattr(x, "_pulpo_pooled", None)
    if ready is not None and ready[1] == x._version and len(ready) > 2:
        return x, ready[0]                            # produced (and differentiated) together with x by the ConvUnit's own autograd node
    src = getattr(x, "_pulpo_bn_src", None)          # x is the untouched output of a ConvUnit: (y, coef, version at production)
    src = src if (src is not None and src[2] == x._version and torch.is_grad_enabled()) else None
    alias, pooled = _AvgPool2Skip.apply(x, ready[0] if (ready is not None and ready[1] == x._version) else None, src[0] if src else None, src[1] if src else None)
    tag = getattr(x, "_pulpo_cat", None)
    if tag is not None:
        alias._pulpo_cat = tag                      # (the skip connection's alias is the same slice of the concatenation buffer)
    return alias, pooled


class _Resize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, mult: float, add, scale):
        _require_gpu(x, add)
        x = planar(x)
        B, C, Di, Hi, Wi = x.shape
        Do, Ho, Wo = size
        out = torch.empty((B, C, Do, Ho, Wo), device=x.device, dtype=torch.float32)
        addc = planar(add) if add is not None else None
        t0 = _hbm_begin("resize_trilinear_fwd")
        lib.call("pulpo_resize_trilinear_scaled_fwd", _ptr(x), _ptr(addc), _ptr(out), B * C, Di, Hi, Wi, Do, Ho, Wo, *scale, mult, _stream())
        _hbm_end(t0, "resize_trilinear_fwd", 4.0 * (x.numel() + out.numel() * (2 if addc is not None else 1)))
        ctx.dims = (B, C, Di, Hi, Wi, Do, Ho, Wo)
        ctx.mult = mult
        ctx.scale = scale
        ctx.has_add = add is not None
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, Di, Hi, Wi, Do, Ho, Wo = ctx.dims
        g = planar(g)
        gin = None
        if ctx.needs_input_grad[0]:
            gin = torch.empty((B, C, Di, Hi, Wi), device=g.device, dtype=torch.float32)
            t0 = _hbm_begin("resize_trilinear_bwd")
            lib.call("pulpo_resize_trilinear_scaled_bwd_det" if DETERMINISTIC else "pulpo_resize_trilinear_scaled_bwd", _ptr(g), _ptr(gin), B * C, Di, Hi, Wi,
                     Do, Ho, Wo, *ctx.scale, ctx.mult, _stream())
            _hbm_end(t0, "resize_trilinear_bwd", 4.0 * (g.numel() + gin.numel()))
        return gin, None, None, (g if ctx.has_add and ctx.needs_input_grad[3] else None), None


def resize_trilinear(x, size, mult: float = 1.0, add=None, scale_factor: Optional[float] = None):
    """mult * F.interpolate(x, size, 'trilinear', align_corners=False) (+ add).  scale_factor: the call being replaced is
    F.interpolate(x, scale_factor=...) - coordinates are then mapped with 1 / scale_factor on every axis instead of in / out (they differ
    wherever in * scale_factor is not an integer); `size` is still the output size, floor(in * scale_factor)."""
    if _is2d(x):                                       # bilinear = trilinear over a depth-1 volume
        return resize_trilinear(_lift(x), [1] + [int(v) for v in size], mult, _lift(add), scale_factor).squeeze(2)
    step = 0.0 if scale_factor is None else float(torch.tensor(1.0 / float(scale_factor), dtype=torch.float32))     # ATen: static_cast<float>(1.0 / scale)
    scale = (0.0 if x.shape[2] == 1 and int(size[0]) == 1 else step, step, step)       # (the lifted depth axis of a slice keeps its identity mapping)
    return _Resize.apply(x, tuple(int(s) for s in size), float(mult), add, scale)


class _FeedbackUp2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *srcs):
        _require_gpu(*srcs)
        srcs = [planar(s) for s in srcs]
        B, _, Di, Hi, Wi = srcs[0].shape
        chans = [int(s.shape[1]) for s in srcs]
        ctot = sum(chans)
        out = new_cl(B, ctot, 2 * Di, 2 * Hi, 2 * Wi, srcs[0].device, act_dtype())
        n = len(srcs)
        ptrs = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
        ch = (ctypes.c_int * n)(*chans)
        t0 = _hbm_begin("feedback_up2_fwd")
        lib.call("pulpo_feedback_up2_fwd_t", ptrs, ch, n, _ptr(out), _dt(out), out.stride(4), B, Di, Hi, Wi, _stream())
        _hbm_end(t0, "feedback_up2_fwd", ctot * B * Di * Hi * Wi * (4.0 + 8 * _esize(out)))             # read the sources, write 8x as many voxels
        ctx.meta = (B, Di, Hi, Wi, chans)
        ctx.keep = srcs      # keep the sources alive until the kernel has been enqueued (same stream: safe afterwards)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Di, Hi, Wi, chans = ctx.meta
        g = to_cl(g)
        n = len(chans)
        gs = [torch.empty((B, c, Di, Hi, Wi), device=g.device, dtype=torch.float32) if ctx.needs_input_grad[i] else None
              for i, c in enumerate(chans)]
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() if t is not None else None for t in gs])
        ch = (ctypes.c_int * n)(*chans)
        t0 = _hbm_begin("feedback_up2_bwd")
        lib.call("pulpo_feedback_up2_bwd_t", _ptr(g), _dt(g), g.stride(4), ptrs, ch, n, B, Di, Hi, Wi, _stream())
        _hbm_end(t0, "feedback_up2_bwd", sum(chans) * B * Di * Hi * Wi * (4.0 + 8 * _esize(g)))
        return tuple(gs)


def feedback_up2(srcs: Sequence[torch.Tensor]) -> torch.Tensor:
    """cat([interpolate(s, x2) for s in srcs], dim=1) as one channels-last tensor"""
    return _FeedbackUp2.apply(*srcs)


# ------------------------------------------------------------------------------------------------ warp / vecint
class _Warp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, df, img):
        _require_gpu(df, img)
        df, img = planar(df), planar(img)
        B, _, Dg, Hg, Wg = df.shape
        _, C, Di, Hi, Wi = img.shape
        out = torch.empty((B, C, Dg, Hg, Wg), device=df.device, dtype=torch.float32)
        t0 = _hbm_begin("warp3d_fwd")
        lib.call("pulpo_warp3d_fwd", _ptr(df), _ptr(img), _ptr(out), B, C, Dg, Hg, Wg, Di, Hi, Wi, _stream())
        _hbm_end(t0, "warp3d_fwd", 4.0 * (df.numel() + img.numel() + out.numel()))          # (3 + 2C) V 4
        ctx.save_for_backward(df, img)
        return out

    @staticmethod
    def backward(ctx, g):
        df, img = ctx.saved_tensors
        g = planar(g)
        B, _, Dg, Hg, Wg = df.shape
        _, C, Di, Hi, Wi = img.shape
        gdf = torch.empty_like(df) if ctx.needs_input_grad[0] else None
        gimg = torch.empty_like(img) if ctx.needs_input_grad[1] else None
        t0 = _hbm_begin("warp3d_bwd")
        if DETERMINISTIC and gimg is not None:          # (the displacement gradient alone is a gather: nothing to order)
            ws = torch.empty(lib.query("pulpo_warp3d_bwd_det_ws_bytes", B, C, Di, Hi, Wi), device=df.device, dtype=torch.uint8)
            lib.call("pulpo_warp3d_bwd_det", _ptr(df), _ptr(img), _ptr(g), _ptr(gdf), _ptr(gimg), _ptr(ws), B, C, Dg, Hg, Wg, Di, Hi, Wi, _stream())
        else:
            lib.call("pulpo_warp3d_bwd", _ptr(df), _ptr(img), _ptr(g), _ptr(gdf), _ptr(gimg), B, C, Dg, Hg, Wg, Di, Hi, Wi, _stream())
        _hbm_end(t0, "warp3d_bwd", 4.0 * (df.numel() + img.numel() + g.numel() + (gdf.numel() if gdf is not None else 0)
                                          + (2 * gimg.numel() if gimg is not None else 0)))      # (image gradient: zero fill + scatter)
        return gdf, gimg


def warp3d(df, img):
    """SpatialTransformer.forward(df, img)"""
    if _is2d(df):                                      # 2-D SpatialTransformer: channels (y, x) -> (0, y, x), depth-1 grid and image
        return warp3d(_lift_field(df), _lift(img)).squeeze(2)
    return _Warp.apply(df, img)


class _VecInt(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, nsteps: int):
        _require_gpu(v)
        v = planar(v)
        B, _, D, H, W = v.shape
        work = torch.empty((nsteps + 1, B, 3, D, H, W), device=v.device, dtype=torch.float32)
        t0 = _hbm_begin("vecint_fwd")
        lib.call("pulpo_vecint_fwd", _ptr(v), _ptr(work), B, D, H, W, nsteps, _stream())
        _hbm_end(t0, "vecint_fwd", 4.0 * v.numel() * 2 * (nsteps + 1))                       # every step: one read, one write of the field
        ctx.save_for_backward(work)
        ctx.nsteps = nsteps
        return work[nsteps]

    @staticmethod
    def backward(ctx, g):
        (work,) = ctx.saved_tensors
        g = planar(g)
        _, B, _, D, H, W = work.shape
        gin = torch.empty((B, 3, D, H, W), device=g.device, dtype=torch.float32)
        t0 = _hbm_begin("vecint_bwd")
        if DETERMINISTIC:
            nws = lib.query("pulpo_vecint_bwd_det_ws_bytes", B, D, H, W, ctx.nsteps)
            ws = torch.empty(nws, device=g.device, dtype=torch.uint8) if nws else None
            lib.call("pulpo_vecint_bwd_det", _ptr(work), _ptr(g), _ptr(gin), _ptr(ws), B, D, H, W, ctx.nsteps, _stream())
        else:
            ntmp = lib.query("pulpo_vecint_bwd_tmp_floats", B, D, H, W, ctx.nsteps)
            tmp = torch.empty(ntmp, device=g.device, dtype=torch.float32) if ntmp else None
            lib.call("pulpo_vecint_bwd", _ptr(work), _ptr(g), _ptr(gin), _ptr(tmp), B, D, H, W, ctx.nsteps, _stream())
        _hbm_end(t0, "vecint_bwd", 4.0 * gin.numel() * (3 * ctx.nsteps + 2))                 # every step: read the field and the gradient, write a gradient
        return gin, None


def vecint(v, nsteps: int = 7):
    if _is2d(v):
        return _unlift_field(vecint(_lift_field(v), nsteps))
    return _VecInt.apply(v, int(nsteps))


# ------------------------------------------------------------------------------------------------ losses
class _NCC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, true, win: int, gamma: float):
        _require_gpu(pred, true)
        pred, true = planar(pred), planar(true)
        B, C, D, H, W = pred.shape
        if C != 1:
            raise PulpoHipError("ncc: single-channel volumes expected")
        N = B * D * H * W
        dev = pred.device
        S = torch.empty(5 * N, device=dev, dtype=torch.float32)
        T = torch.empty(10 * N, device=dev, dtype=torch.float32)
        nblk = lib.query("pulpo_loss_blocks", N)
        part = torch.empty(nblk, device=dev, dtype=torch.float32)
        t0 = _hbm_begin("ncc_fwd")
        lib.call("pulpo_ncc_fwd", _ptr(true), _ptr(pred), _ptr(S), _ptr(T), _ptr(part), B, D, H, W, win, _stream())
        _hbm_end(t0, "ncc_fwd", 4.0 * 22 * N)                  # 2 images in; three separable passes over 5 box-sum channels (write 5, read 5, write 5, read 5)
        loss = _colsum(part, nblk, 1, -gamma / B)
        ctx.save_for_backward(pred, true, S)
        ctx.win, ctx.gamma = win, gamma
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, true, S = ctx.saved_tensors
        B, _, D, H, W = pred.shape
        N = B * D * H * W
        T = torch.empty(6 * N, device=pred.device, dtype=torch.float32)
        gJ = torch.empty_like(pred)
        g = g.contiguous()
        t0 = _hbm_begin("ncc_bwd")
        lib.call("pulpo_ncc_bwd", _ptr(true), _ptr(pred), _ptr(S), _ptr(T), _ptr(g), -ctx.gamma / B, _ptr(gJ), B, D, H, W, ctx.win, _stream())
        _hbm_end(t0, "ncc_bwd", 4.0 * 20 * N)                  # 2 images + 5 sums in; three passes over 3 channels (write 3, read 3, write 3, read 3); gradient out
        return gJ, None, None, None


def ncc_loss(pred, true, win: int = 9, gamma: float = 0.05):
    if _is2d(pred):                                    # depth 1 selects the win x win window count in the kernel
        return ncc_loss(_lift(pred), _lift(true), win, gamma)
    return _NCC.apply(pred, true, int(win), float(gamma))


class _KL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, sigma, mu1, sigma1):
        _require_gpu(mu, sigma, mu1, sigma1)
        mu, sigma = planar(mu), planar(sigma)
        mu1 = planar(mu1) if mu1 is not None else None
        sigma1 = planar(sigma1) if sigma1 is not None else None
        n = mu.numel()
        nblk = lib.query("pulpo_loss_blocks", n)
        part = torch.empty(nblk, device=mu.device, dtype=torch.float32)
        t0 = _hbm_begin("kl_fwd")
        lib.call("pulpo_kl_fwd", _ptr(mu), _ptr(sigma), _ptr(mu1), _ptr(sigma1), n, _ptr(part), _stream())
        _hbm_end(t0, "kl_fwd", 4.0 * n * (2 + (mu1 is not None) + (sigma1 is not None)))
        ctx.save_for_backward(mu, sigma, mu1, sigma1)
        return _colsum(part, nblk, 1, 0.5 / mu.shape[0]).reshape(())

    @staticmethod
    def backward(ctx, g):
        mu, sigma, mu1, sigma1 = ctx.saved_tensors
        gmu, gsg = torch.empty_like(mu), torch.empty_like(sigma)
        g = g.contiguous()
        t0 = _hbm_begin("kl_bwd")
        lib.call("pulpo_kl_bwd", _ptr(mu), _ptr(sigma), _ptr(mu1), _ptr(sigma1), _ptr(g), 1.0 / mu.shape[0], _ptr(gmu), _ptr(gsg), mu.numel(),
                 _stream())
        _hbm_end(t0, "kl_bwd", 4.0 * mu.numel() * (4 + (mu1 is not None) + (sigma1 is not None)))
        return gmu, gsg, None, None


def kl_diag(mu, sigma, mu1=None, sigma1=None):
    """KL[N(mu, sigma^2) || N(mu1, sigma1^2)] (sum over features, mean over batch); mu1/sigma1 None = N(0,1).
    Gradients flow to (mu, sigma) only: the prior is a constant in the reference (pulpo.py:330-341)."""
    if _is2d(mu):
        return kl_diag(_lift(mu), _lift(sigma), _lift(mu1), _lift(sigma1))
    return _KL.apply(mu, sigma, mu1, sigma1)


def kl_std_normal(mu, sigma):
    return _KL.apply(mu, sigma, None, None)


class _L2Reg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, df, lamb: float):
        _require_gpu(df)
        df = planar(df)
        B, C, D, H, W = df.shape
        n = df.numel()
        nblk = lib.query("pulpo_loss_blocks", n)
        part = torch.empty(nblk, device=df.device, dtype=torch.float32)
        t0 = _hbm_begin("l2reg_fwd")
        lib.call("pulpo_l2reg_fwd", _ptr(df), B * C, D, H, W, _ptr(part), _stream())
        _hbm_end(t0, "l2reg_fwd", 4.0 * n)
        coef = lamb * D * H * W / float(B * C * max(D - 1, 1) * (H - 1) * (W - 1))      # D == 1: the 2-D form (no depth difference)
        ctx.save_for_backward(df)
        ctx.coef = coef
        return _colsum(part, nblk, 1, coef).reshape(())

    @staticmethod
    def backward(ctx, g):
        (df,) = ctx.saved_tensors
        B, C, D, H, W = df.shape
        gdf = torch.empty_like(df)
        g = g.contiguous()
        t0 = _hbm_begin("l2reg_bwd")
        lib.call("pulpo_l2reg_bwd", _ptr(df), _ptr(g), ctx.coef, _ptr(gdf), B * C, D, H, W, _stream())
        _hbm_end(t0, "l2reg_bwd", 8.0 * df.numel())
        return gdf, None


def l2_reg(df, lamb: float = 0.0):
    if _is2d(df):
        return l2_reg(_lift(df), lamb)
    return _L2Reg.apply(df, float(lamb))


class _WeightedSum(torch.autograd.Function):
    """(sum_i w_i t_i, [w_0 t_0, ..., w_{n-1} t_{n-1}]) (optionally * scale): the per-level weighting and summation of the Hierarchical*
    losses as ONE launch (and one in the backward pass) instead of a mul + an add kernel per level and their autograd counterparts"""

    @staticmethod
    def forward(ctx, terms, weights_dev, scale):
        _require_gpu(terms, weights_dev)
        n = terms.numel()
        levels = torch.empty(n, device=terms.device, dtype=torch.float32)
        total = torch.empty((), device=terms.device, dtype=torch.float32)
        lib.call("pulpo_weighted_sum_fwd", _ptr(terms), _ptr(weights_dev), n, 1.0 if scale is None else float(scale), int(scale is not None),
                 _ptr(levels), _ptr(total), _stream())
        ctx.save_for_backward(weights_dev)
        ctx.scale = 1.0 if scale is None else float(scale)
        ctx.set_materialize_grads(False)
        return total, levels

    @staticmethod
    def backward(ctx, gtotal, glevels):
        (weights_dev,) = ctx.saved_tensors
        n = weights_dev.numel()
        if gtotal is None and glevels is None:
            return None, None, None
        gt = torch.empty(n, device=weights_dev.device, dtype=torch.float32)
        lib.call("pulpo_weighted_sum_bwd", _ptr(gtotal.contiguous() if gtotal is not None else None),
                 _ptr(glevels.contiguous() if glevels is not None else None), _ptr(weights_dev), n, ctx.scale, _ptr(gt), _stream())
        return gt, None, None


_WEIGHT_VECTORS: dict = {}


def weighted_sum(terms, weights, scale=None):
    """terms: list of 0-d device tensors, weights: list of python floats -> (sum, [w_i * term_i]) as 0-d views of one device vector.
    scale (optional) multiplies the sum and every level term afterwards (models.py:161-162: kl_loss * beta)."""
    dev = terms[0].device
    key = (tuple(float(w) for w in weights), dev)
    wd = _WEIGHT_VECTORS.get(key)
    if wd is None:
        if len(_WEIGHT_VECTORS) > 64:
            _WEIGHT_VECTORS.clear()
        wd = _WEIGHT_VECTORS[key] = torch.tensor(key[0], device=dev, dtype=torch.float32)
    total, levels = _WeightedSum.apply(torch.stack([t.reshape(()).float() for t in terms]), wd, scale)
    return total, [levels[i] for i in range(len(terms))]


# ------------------------------------------------------------------------------------------------ alternative losses / metrics
class _SqDiff(torch.autograd.Function):
    """L2_loss: spatial sum of squared differences, mean over batch and channels"""

    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a, b)
        a, b = planar(a), planar(b)
        n = a.numel()
        nblk = lib.query("pulpo_metric_blocks", n)
        part = torch.empty(nblk, device=a.device, dtype=torch.float32)
        lib.call("pulpo_sqdiff_fwd", _ptr(a), _ptr(b), n, _ptr(part), _stream())
        ctx.save_for_backward(a, b)
        ctx.coef = 1.0 / (a.shape[0] * a.shape[1])
        return _colsum(part, nblk, 1, ctx.coef).reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        lib.call("pulpo_sqdiff_bwd", _ptr(a), _ptr(b), _ptr(g.contiguous()), ctx.coef, _ptr(ga), a.numel(), _stream())
        return ga, None


def l2_loss(inp, target):
    if _is2d(inp):
        return l2_loss(_lift(inp), _lift(target))
    return _SqDiff.apply(inp, target)


class _Dice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, tgt, dice_factor: float):
        _require_gpu(inp, tgt)
        inp, tgt = planar(inp), planar(tgt)
        nplanes = inp.shape[0] * inp.shape[1]
        V = inp.numel() // nplanes
        nb = lib.query("pulpo_dice_blocks", V)
        part = torch.empty(nplanes * nb * 3, device=inp.device, dtype=torch.float32)
        numden = torch.empty(2 * nplanes, device=inp.device, dtype=torch.float64)
        loss = torch.empty((), device=inp.device, dtype=torch.float32)
        lib.call("pulpo_dice_fwd", _ptr(inp), _ptr(tgt), nplanes, V, dice_factor, _ptr(part), _ptr(numden), _ptr(loss), _stream())
        ctx.save_for_backward(inp, tgt, numden)
        ctx.meta = (nplanes, V, dice_factor)
        return loss

    @staticmethod
    def backward(ctx, g):
        inp, tgt, numden = ctx.saved_tensors
        nplanes, V, df = ctx.meta
        ginp = torch.empty_like(inp)
        lib.call("pulpo_dice_bwd", _ptr(inp), _ptr(tgt), _ptr(numden), _ptr(g.contiguous()), nplanes, V, df, _ptr(ginp), _stream())
        return ginp, None, None


def soft_dice_loss(inp, target, dice_factor=1):
    if _is2d(inp):
        return soft_dice_loss(_lift(inp), _lift(target), dice_factor)
    return _Dice.apply(inp, target, float(dice_factor))


def jacobian_det(df, normalize: bool = True):
    """determinant of the Jacobian of x + u(x), (B,3,D,H,W) -> (B,D,H,W); evaluation metric, not differentiable here"""
    if _is2d(df):                                      # (B,2,H,W) -> (B,H,W)
        return jacobian_det(_lift(df), normalize)[:, 0]
    _require_gpu(df)
    d = planar(df.detach())
    B, C, D, H, W = d.shape
    if C != (2 if D == 1 else 3):
        raise PulpoHipError("jacobian_det: displacement field (B,3,D,H,W) or, for slices, (B,2,1,H,W) expected")
    out = torch.empty((B, D, H, W), device=d.device, dtype=torch.float32)
    lib.call("pulpo_jacdet_fwd", _ptr(d), _ptr(out), None, B, D, H, W, int(bool(normalize)), _stream())
    return out


class _JDetStd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, df, lamb: float, normalize: bool):
        _require_gpu(df)
        df = planar(df)
        B, C, D, H, W = df.shape
        n = B * D * H * W
        jd = torch.empty((B, D, H, W), device=df.device, dtype=torch.float32)
        part = torch.empty(2 * lib.query("pulpo_metric_blocks", n), device=df.device, dtype=torch.float32)
        stat = torch.empty(2, device=df.device, dtype=torch.float64)
        loss = torch.empty((), device=df.device, dtype=torch.float32)
        lib.call("pulpo_jacdet_fwd", _ptr(df), _ptr(jd), _ptr(part), B, D, H, W, int(normalize), _stream())
        lib.call("pulpo_jdetstd_finalize", _ptr(part), n, lamb, _ptr(stat), _ptr(loss), _stream())
        ctx.save_for_backward(df, jd, stat)
        ctx.meta = (lamb, normalize)
        return loss

    @staticmethod
    def backward(ctx, g):
        df, jd, stat = ctx.saved_tensors
        lamb, normalize = ctx.meta
        B, _, D, H, W = df.shape
        if DETERMINISTIC:
            raise PulpoHipError("the `jdet` regulariser's backward scatters with float atomics and has no deterministic form (PULPO_DETERMINISTIC covers "
                                "the default training path: ncc / mse / dice + L2 regulariser)")
        gdf = torch.empty_like(df)
        lib.call("pulpo_jdetstd_bwd", _ptr(df), _ptr(jd), _ptr(stat), _ptr(g.contiguous()), lamb, _ptr(gdf), B, D, H, W, int(normalize), _stream())
        return gdf, None, None


def jdet_std(df, lamb: float = 0.0, normalize: bool = True):
    if _is2d(df):
        return jdet_std(_lift(df), lamb, normalize)
    return _JDetStd.apply(df, float(lamb), bool(normalize))


class _KLNonDiag(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, sigma, prior_lambda: float):
        _require_gpu(mu, sigma)
        mu, sigma = planar(mu), planar(sigma)
        B, C, D, H, W = mu.shape
        part = torch.empty(4 * lib.query("pulpo_metric_blocks", mu.numel()), device=mu.device, dtype=torch.float32)
        loss = torch.empty((), device=mu.device, dtype=torch.float32)
        lib.call("pulpo_kl_nondiag_fwd", _ptr(mu), _ptr(sigma), B * C, D, H, W, prior_lambda, _ptr(part), _ptr(loss), _stream())
        ctx.save_for_backward(mu, sigma)
        ctx.lam = prior_lambda
        return loss

    @staticmethod
    def backward(ctx, g):
        mu, sigma = ctx.saved_tensors
        B, C, D, H, W = mu.shape
        gmu, gsg = torch.empty_like(mu), torch.empty_like(sigma)
        lib.call("pulpo_kl_nondiag_bwd", _ptr(mu), _ptr(sigma), _ptr(g.contiguous()), B * C, D, H, W, ctx.lam, _ptr(gmu), _ptr(gsg), _stream())
        return gmu, gsg, None


def kl_nondiagonal(mu, sigma, prior_lambda: float = 20.0):
    if _is2d(mu):
        return kl_nondiagonal(_lift(mu), _lift(sigma), prior_lambda)
    return _KLNonDiag.apply(mu, sigma, float(prior_lambda))


# ------------------------------------------------------------------------------------------------ optimizer
# ------------------------------------------------------------------------------------------------ MC uncertainty
class StreamingMoments:
    """running per-voxel mean / unbiased std over Monte-Carlo samples (evaluate.py:222-251 keeps all N samples instead).
    update() folds one (B, C, D, H, W) sample in; std_map() = torch.mean(torch.std(stack, axis=0), axis=<channel>) -> (B, D, H, W)."""

    def __init__(self) -> None:
        self.count = 0
        self._mean = None
        self._m2 = None

    def update(self, sample: torch.Tensor) -> None:
        _require_gpu(sample)
        s = sample.detach().contiguous()
        if self._mean is None:
            self._mean, self._m2 = torch.empty_like(s), torch.empty_like(s)
        elif s.shape != self._mean.shape:
            raise ValueError(f"StreamingMoments: sample shape {tuple(s.shape)} differs from {tuple(self._mean.shape)}")
        self.count += 1
        lib.call("pulpo_mc_moments_update", _ptr(s), _ptr(self._mean), _ptr(self._m2), s.numel(), self.count, _stream())

    def mean(self) -> torch.Tensor:
        if self._mean is None:
            raise ValueError("StreamingMoments: no samples")
        return self._mean

    def std_map(self, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self._m2 is None:
            raise ValueError("StreamingMoments: no samples")
        B, C = self._m2.shape[0], self._m2.shape[1]
        V = self._m2[0, 0].numel()
        out = torch.empty((B,) + tuple(self._m2.shape[2:]), device=self._m2.device, dtype=torch.float32)
        sc = None
        if scale is not None:
            _require_gpu(scale)
            sc = scale.detach().expand((B, 1) + tuple(self._m2.shape[2:])).contiguous()
        lib.call("pulpo_mc_moments_std", _ptr(self._m2), _ptr(sc), _ptr(out), B, C, V, self.count, _stream())
        return out


def adam_step(p, g, m, v, lr: float, step: int, beta1=0.9, beta2=0.999, eps=1e-8, gscale: float = 1.0):
    _require_gpu(p, g, m, v)
    t0 = _hbm_begin("adam_step")
    lib.call("pulpo_adam_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps, int(step), gscale, _stream())
    _hbm_end(t0, "adam_step", 28.0 * p.numel())               # read p, g, m, v; write p, m, v
    refresh_weight_packs()                       # the kernel rewrote parameters through raw pointers: the cached packs follow, in one launch


# ------------------------------------------------------------------------------------------------ evaluation scalars (evaluate.py)
def rmse(inp, target):
    """sqrt(MSELoss(inp, target)) as a 0-d device tensor (evaluate.py:315-319)"""
    _require_gpu(inp, target)
    a, b = inp.detach().contiguous(), target.detach().expand_as(inp).contiguous()
    n = a.numel()
    part = torch.empty(lib.query("pulpo_metric_blocks", n), device=a.device, dtype=torch.float32)
    out = torch.empty((), device=a.device, dtype=torch.float32)
    lib.call("pulpo_rmse", _ptr(a), _ptr(b), n, _ptr(part), _ptr(out), _stream())
    return out


def dsc(inp, target):
    """dice similarity coefficient of two (soft) segmentation maps (evaluate.py:321-327), 0-d device tensor"""
    _require_gpu(inp, target)
    a, b = inp.detach().contiguous(), target.detach().expand_as(inp).contiguous()
    nplanes = a.shape[0] * a.shape[1]
    V = a.numel() // nplanes
    part = torch.empty(nplanes * lib.query("pulpo_dice_blocks", V) * 3, device=a.device, dtype=torch.float32)
    out = torch.empty((), device=a.device, dtype=torch.float32)
    lib.call("pulpo_dsc", _ptr(a), _ptr(b), nplanes, V, _ptr(part), _ptr(out), _stream())
    return out


def percent_leq0(x):
    """100 * (x <= 0).sum() / x.numel() as a 0-d device tensor (the 'JDetLeq0' metric, evaluate.py:1441-1446)"""
    _require_gpu(x)
    a = x.detach().contiguous()
    n = a.numel()
    part = torch.empty(lib.query("pulpo_metric_blocks", n), device=a.device, dtype=torch.float32)
    out = torch.empty((), device=a.device, dtype=torch.float32)
    lib.call("pulpo_percent_leq0", _ptr(a), n, _ptr(part), _ptr(out), _stream())
    return out


def warp_landmarks(lm, df):
    """lm.long() - df[:, :, lm[0,:,0], lm[0,:,1], lm[0,:,2]].transpose(-2, -1)   (evaluate.py:410-423, src/components/utils.py:15-25)
    lm: (1, n_landmarks, ndims); df: (n_samples, ndims, ...) -> (n_samples, n_landmarks, ndims) float.  Out-of-range landmarks raise
    IndexError like the reference's tensor indexing (one host read of a device flag: an evaluation-time helper)."""
    _require_gpu(df)
    nd = df.dim() - 2
    if lm.dim() != 3 or lm.shape[0] != 1 or lm.shape[2] != nd or df.shape[1] != nd or nd not in (2, 3):
        raise PulpoHipError(f"warp_landmarks: lm (1, n, ndims) and df (samples, ndims, ...) expected, got {tuple(lm.shape)} and {tuple(df.shape)}")
    d = df.detach().contiguous()
    l = lm.detach().to(device=d.device, dtype=torch.float32).contiguous()
    nlm, ns = int(lm.shape[1]), int(d.shape[0])
    D, H, W = (1, *d.shape[2:]) if nd == 2 else d.shape[2:]
    out = torch.empty((ns, nlm, nd), device=d.device, dtype=torch.float32)
    flag = torch.empty(1, device=d.device, dtype=torch.int32)
    lib.call("pulpo_warp_landmarks", _ptr(l), _ptr(d), _ptr(out), nlm, ns, nd, int(D), int(H), int(W),
             ctypes.cast(flag.data_ptr(), ctypes.POINTER(ctypes.c_int)), _stream())
    if int(flag.item()):
        raise IndexError("warp_landmarks: landmark index out of bounds of the displacement field")
    return out
