"""Phase stamps of the persistent bf16 forward kernel (diagnostic build: bash scripts/build_variant.sh pwst conv3d_bf16 -DPULPO_PW_ABL=64, run with
PULPO_HIP_LIB=.../libpulpo_hip_pwst.so): clocks per tile between the phase boundaries, median over workgroups, tiles 2..8 of each workgroup.
usage: python scripts/stamps_pw.py Cin Cout S"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pulpo_amd import ops
from pulpo_amd._lib import lib

def main():
    ci, co, S = (int(v) for v in sys.argv[1:4])
    lib.load()
    ops.set_conv_precision("bf16", activations="bf16")
    x = torch.randn(1, ci, S, S, S, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
    y = ops.new_cl(1, co, S, S, S, x.device, torch.bfloat16)
    stats = torch.empty(lib.query("pulpo_conv3d_k3_fwd_bf16_stat_tiles", 1, S, S, S) * 2 * co * 2, device="cuda")
    wp = ops._pack_weight(w, False, shape=(1, S, S, S))
    for _ in range(3):
        ops._conv_raw(x, wp, None, y, ci, co, stats)
    torch.cuda.synchronize()
    buf = np.zeros(512 * 128, dtype=np.uint64)
    f = lib._dll.pulpo_debug_read_stamps_pw
    f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert f(buf.ctypes.data, buf.nbytes) == 0
    st = buf.reshape(512, 128).astype(np.int64)
    ng = 3 if co % 64 else 9
    order = [0, 1] + [v for g in range(ng) for v in (8 + g, 2 + g)] if ng == 3 else None
    if order is None:
        print("stamps are laid out for the 3-group (32-cout) kernel"); return
    order += [7, 11, 5, 6]
    names = ["tile head -> halo stored, barrier"] + [n for g in range(ng) for n in (f"group {g}: loads issued + 36 MFMAs", f"group {g}: weight store + barrier")] + \
            ["next halo -> LDS (waits for its last slice)", "epilogue, first slab", "epilogue, second slab", "statistics partials"]
    tot = np.zeros(len(order) - 1); gap = 0.0
    tiles = range(2, 8)
    for t in tiles:
        for k in range(len(order) - 1):
            tot[k] += np.median(st[:, 12 * t + order[k + 1]] - st[:, 12 * t + order[k]])
        gap += np.median(st[:, 12 * (t + 1)] - st[:, 12 * t + 6])
    tot /= len(tiles); gap /= len(tiles)
    print(f"{ci}->{co} @{S}^3: clocks per tile {tot.sum() + gap:.0f}")
    for n, v in zip(names + ["loop back"], list(tot) + [gap]):
        print(f"    {n:45s} {v:8.0f}  {100 * v / (tot.sum() + gap):5.1f} %")

if __name__ == "__main__":
    main()
