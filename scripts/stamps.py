"""Phase stamps of the (y, x) Winograd kernel (diagnostic build -DPULPO_ABL=9): median clock counts between phase boundaries."""
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
    buf = np.zeros(20000 * 32, dtype=np.uint64)
    f = lib._dll.pulpo_debug_read_stamps
    f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert f(buf.ctypes.data, buf.nbytes) == 0
    nblk = min(20000, (S // 4) * (S // 8) * (S // 8) * ((co + 31) // 32))
    st = buf.reshape(20000, 32)[:nblk].astype(np.int64)
    names = {0: "start", 1: "prologue (index math, first loads issued)", 24: "main loop end", 25: "barrier before exchange", 26: "exchange written + barrier",
             27: "exchange read + stores issued", 28: "end"}
    for c in range(4):
        names[2 + c * 5] = f"chunk {c}: barrier"
        names[3 + c * 5] = f"chunk {c}: staged (stores issued)"
        for dz in range(3):
            names[4 + c * 5 + dz] = f"chunk {c} dz {dz}: weights stored + barrier"
    nch = (ci + 7) // 8
    slots = [s for s in sorted(names) if not (2 <= s < 22 and (s - 2) // 5 >= nch)]
    # steady-state blocks only (skip the first round, which starts cold)
    sel = st[1024:] if nblk > 2048 else st
    prev = slots[0]
    print(f"{ci}->{co} @{S}^3: {nblk} workgroups; median clocks per phase (wave 0), total median {np.median(sel[:, 28] - sel[:, 0]):.0f}")
    for s_ in slots[1:]:
        d = sel[:, s_] - sel[:, prev]
        print(f"  {names[s_]:48s} {np.median(d):9.0f}   (p10 {np.percentile(d, 10):7.0f}  p90 {np.percentile(d, 90):7.0f})")
        prev = s_
    # lockstep? start-time differences between blocks b and b+256 in the first round
    first = st[:512, 0]
    print("first-round start spread (clocks): ", int(first.max() - first.min()), " second-residents minus first: median", int(np.median(first[256:512] - first[:256])))

if __name__ == "__main__":
    main()
