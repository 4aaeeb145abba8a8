"""Phase stamps of the pipelined (y, x) Winograd kernel (diagnostic build -DPULPO_ABL=64 of conv3d_wino2p.hip): per tile, clocks of the
matrix loop and of the epilogue's phases (head loads, first exchange, first row tile's stores, second row tile, statistics)."""
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
    nt = min(12, int((st[0, 3::6] > 0).sum()))
    names = ["matrix loop", "epilogue head (waits for weights / bias)", "acc -> LDS, barrier", "row tile 0: read, transform, store", "row tile 1", "statistics + final barrier"]
    prev = st[:, 2]
    tot = np.zeros(6)
    for t in range(1, nt - 1):                      # (skip the first and the last recorded tile)
        base = st[:, 8 + 6 * (t - 1)]
        seq = [st[:, 3 + 6 * t + k] for k in range(6)]
        last = base
        for k in range(6):
            tot[k] += np.median(seq[k] - last); last = seq[k]
    tot /= max(nt - 2, 1)
    print(f"{ci}->{co} @{S}^3: clocks per tile {tot.sum():.0f}")
    for n, v in zip(names, tot):
        print(f"    {n:45s} {v:8.0f}  {100 * v / tot.sum():5.1f} %")

if __name__ == "__main__":
    main()
