"""Stamps of the persistent (y, x) Winograd kernel (diagnostic build -DPULPO_ABL=9): which workgroups share a CU, how their tile
boundaries are phased against each other, clocks per tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pulpo_amd import ops
from pulpo_amd._lib import lib

def main():
    ci, co, S = (int(v) for v in sys.argv[1:4])
    lib.load()
    x = torch.randn(1, ci, S, S, S, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(co, ci, 3, 3, 3, device="cuda") * 0.05
    y = ops.new_cl(1, co, S, S, S, x.device)
    stats = torch.empty(lib.query("pulpo_conv3d_k3_stat_tiles", 1, S, S, S) * 2 * co * 2, device="cuda")
    wp = ops._pack_weight(w, False, shape=(1, S, S, S))
    for _ in range(3):
        ops._conv_raw(x, wp, None, y, ci, co, stats)
    torch.cuda.synchronize()
    buf = np.zeros(512 * 80, dtype=np.uint64)
    f = lib._dll.pulpo_debug_read_stamps
    f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert f(buf.ctypes.data, buf.nbytes) == 0
    st = buf.reshape(512, 80).astype(np.int64)
    hw, xcc, t0 = st[:, 0], st[:, 1] & 15, st[:, 2]
    wave_id, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ntile = int(((st[0, 3:] > 0).sum()) // 2)
    print(f"{ci}->{co} @{S}^3: tiles per workgroup {ntile}; wave slots seen {sorted(set(wave_id.tolist()))}; distinct CU keys {len(set(key.tolist()))}")
    ends = st[:, 4:4 + 2 * ntile:2]
    loop_ends = st[:, 3:3 + 2 * ntile:2]
    dur = np.diff(np.concatenate([t0[:, None], ends], axis=1), axis=1)
    print(f"  clocks per tile: median {np.median(dur[:, 1:-1]):.0f}  (first tile {np.median(dur[:, 0]):.0f}); epilogue median {np.median(ends - loop_ends):.0f}")
    # phase of the partner: for every CU with two workgroups, offset of the second's tile ends against the first's, as a fraction of the tile time
    offs = []
    for k in set(key.tolist()):
        idx = np.where(key == k)[0]
        if len(idx) == 2:
            a_, b_ = idx
            T = np.median(dur[a_, 1:-1])
            mid = ntile // 2
            d = (ends[b_, mid] - ends[a_, mid]) / T
            offs.append(d - np.floor(d))
    offs = np.array(offs)
    print(f"  CUs with two workgroups: {len(offs)}; partner phase (fraction of a tile, 0 = lockstep): "
          f"hist {np.histogram(offs, bins=10, range=(0, 1))[0].tolist()}  block ids of a pair e.g. {np.where(key == key[0])[0].tolist()}")
    print(f"  start clock spread {int(t0.max() - t0.min())}")

if __name__ == "__main__":
    main()
