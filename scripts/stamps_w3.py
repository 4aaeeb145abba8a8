"""Shader-clock stamps of the F(2x2x2,3x3x3) forward kernel's phases (variant build: bash scripts/build_variant.sh w3st conv3d_wino3 -DPULPO_W3_STAMPS=1,
run with PULPO_HIP_LIB=.../libpulpo_hip_w3st.so): per wave of every workgroup, the second tile's chunks (start, first MFMA ready, last MFMA issued /
barrier reached, barrier passed) and epilogue (start, first parity exchanged, first parity stored, end); medians over workgroups.
usage: python scripts/stamps_w3.py Cin Cout S"""
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
    assert wp._pulpo_algo == "wino3", wp._pulpo_algo
    for _ in range(3):
        ops._conv_raw(x, wp, None, y, ci, co, stats)
    torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 32, dtype=np.uint32)
    f = lib._dll.pulpo_debug_read_stamps_w3
    f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert f(buf.ctypes.data, buf.nbytes) == 0
    st = buf.reshape(256, 8, 32).astype(np.int64)
    nchunk = ci // 8
    m = lambda k: np.median(st[:, :, k] & 0xFFFFFFFF)
    print(f"{ci}->{co} @{S}^3, {nchunk} chunks per tile; clocks of the second tile of every workgroup, median over 256 workgroups x 8 waves")
    print(f"  chunk loop: rows of the first pair + combination {m(0):7.0f} | MFMAs + side work {m(1):7.0f} | barrier wait {m(2):7.0f}   (sum {m(3):7.0f}; "
          f"matrix instructions alone: {nchunk * 32 * 64 * 2} clocks per SIMD for its two waves)")
    print(f"  epilogue: x transform, first exchange + barrier {m(4):6.0f} | first parity: read, transform, store {m(5):6.0f} | second parity + statistics {m(6):6.0f}")
    print(f"  tile total {m(7):7.0f}")
    print("  per wave (py = wave & 3, z half = wave >> 2; waves w and w + 4 share a SIMD): rows | MFMA phase | barrier wait | epilogue parts")
    for w_ in range(8):
        v = [np.median(st[:, w_, k] & 0xFFFFFFFF) for k in range(8)]
        print(f"    wave {w_}: {v[0]:6.0f} | {v[1]:7.0f} | {v[2]:7.0f} | {v[4]:6.0f} {v[5]:6.0f} {v[6]:6.0f} | total {v[7]:7.0f}")

if __name__ == "__main__":
    main()
