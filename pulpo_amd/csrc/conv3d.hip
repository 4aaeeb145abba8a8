// 3x3x3 / pad 1 / stride 1 convolution for PULPo's ConvUnit (reference: src/network_blocks.py:23), fp32,
// as an implicit GEMM on the CDNA4 matrix cores:  v_mfma_f32_32x32x2_f32 (exact fp32 = k-ordered fmaf chain).
//
//   forward / dgrad :  D[voxel][cout] += A[voxel][(tap,cin)] * B[(tap,cin)][cout]
//   wgrad           :  D[(tap,cin)][cout] += A[(tap,cin)][voxel] * B[voxel][cout]
//
// Data layout: activations are channels-last (N,D,H,W,C) with explicit batch/pixel/channel strides, so channel
// slices of a concatenation buffer and planar 1-3 channel volumes go through the same kernels.
// Direct kernel: one workgroup = 256 threads = 4 waves (one per SIMD) owns a 2x8x8 or 4x8x8 voxel tile; the halo of a 16-channel
// input chunk is staged once in LDS ([voxel][CH+1], odd stride -> conflict-free ds_read_b32 for the A fragment; all global loads of the
// tile issued back to back) and re-used by all 27 taps; the 27 weight slabs stream through a double-buffered LDS tile, prefetched
// global->registers one tap ahead.
// Volumes of >= 20^3 voxels (depth % 4 == 0) run the Winograd forms further down instead: F(2x2,3x3) in (y, x) for forward / data
// gradient (conv3d_k3_wino2p_mfma / conv3d_k3_wino2_mfma) and F(2x2,3x3) / F(2,3) along x for the weight gradient
// (conv3d_k3_wgrad_wino) - fewer matrix instructions, all arithmetic still fp32.
#include "conv_shared.h"
#include "act_io.h"
#include <stdlib.h>

namespace {

using namespace pulpo_conv;

// stage the halo tile of channels [c0, c0+CH) into xs[HV][CH+1]; zero outside the volume / beyond Cin
template <int CH, bool VEC, int TZv = TZ>
__device__ __forceinline__ void stage_halo(float* xs, const float* __restrict__ in, long in_ps, long in_cs, int c0, int Cin,
                                           int z0, int y0, int x0, int D, int H, int W, int tid) {
    constexpr int CP = CH + 1;
    constexpr int HV = (TZv + 2) * HY * HX;          // halo voxels of a TZv x 8 x 8 tile (shadows the 2x8x8 constant)
    if constexpr (VEC) {
        // all loads of the tile are issued back to back (NIT float4 per thread in flight), then written to LDS:
        // one exposed memory latency per chunk instead of one per loop iteration
        constexpr int Q = CH / 4;
        constexpr int NIT = (HV * Q + 255) / 256;
        float4 v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            const int hv = j / Q, q = j - hv * Q;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < HV * Q && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c0 + 4 * q < Cin)
                v[u] = *reinterpret_cast<const float4*>(in + ((long)(gz * H + gy) * W + gx) * in_ps + c0 + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            if (j < HV * Q) {
                const int hv = j / Q, q = j - hv * Q;
                float* d = xs + hv * CP + 4 * q;
                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
            }
        }
    } else if constexpr (CH <= 4) {
        // the 2-/3-channel input layers (planar operands): as above, every load of the tile in flight before the first LDS write;
        // thread -> (channel, halo voxel) with the voxel fastest: a wave reads runs of 10 consecutive x of ONE channel plane
        constexpr int NIT = (HV * CH + 255) / 256;
        float v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            const int c = j / HV, hv = j - c * HV;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            v[u] = 0.f;
            if (j < HV * CH && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c0 + c < Cin)
                v[u] = in[((long)(gz * H + gy) * W + gx) * in_ps + (long)(c0 + c) * in_cs];
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            if (j < HV * CH) xs[(j % HV) * CP + (j / HV)] = v[u];
        }
    } else {
        for (int j = tid; j < HV * CH; j += 256) {
            const int hv = j / CH, c = j - hv * CH;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            float v = 0.f;
            if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c0 + c < Cin)
                v = in[((long)(gz * H + gy) * W + gx) * in_ps + (long)(c0 + c) * in_cs];
            xs[hv * CP + c] = v;
        }
    }
}

template <int CH, int NT, bool VEC, int TZv>
__global__ __launch_bounds__(256, TZv == 4 ? 3 : 2) void conv3d_k3_mfma(ConvArgs a) {
    constexpr int CP = CH + 1;
    constexpr int NN = NT / 32;
    constexpr int MT = TZv / 2;                      // 32-voxel row tiles per wave: the workgroup tile is TZv x 8 x 8 voxels
    constexpr int HV = (TZv + 2) * HY * HX;
    constexpr int XS = (HV * CP + 3) & ~3;
    // taps per weight stage: the 2-/3-channel input layers (one chunk, 2-4 MFMAs per tap and wave) keep ALL 27 tap slabs in LDS - one
    // barrier per tile instead of one per tap, which was what bounded them (400 us for 2->32 at 160^3, 205 us of matrix time)
    constexpr int TPS = CH <= 4 ? 27 : 1;
    constexpr int SPC = 27 / TPS;                  // stages per chunk
    constexpr int WF4 = TPS * CH * NT / 4;         // float4 per weight stage
    constexpr int NW = (WF4 + 255) / 256;          // float4 per thread per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ws = smem + XS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid0 = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int split = lid0 % a.ksplit;          // splits of one tile are neighbours: they share the halo in L2
    const int lid = lid0 / a.ksplit;
    const int cot = lid % a.ncot;
    const int tile_lin = lid / a.ncot;
    int t = tile_lin;
    const int tx_ = t % a.ntx; t /= a.ntx;
    const int ty_ = t % a.nty; t /= a.nty;
    const int tz_ = t % a.ntz;
    const int b = t / a.ntz;
    const int z0 = tz_ * TZv, y0 = ty_ * TY, x0 = tx_ * TX;
    const int co0 = cot * NT;
    const int nchunk_all = (a.Cin + CH - 1) / CH;
    const int cper = (nchunk_all + a.ksplit - 1) / a.ksplit;
    const int chunk0 = split * cper, chunk1 = min(nchunk_all, chunk0 + cper);
    const int it0 = chunk0 * SPC, niter = chunk1 * SPC;
    const float* in_b = a.in + (long)b * a.in_bs;

    // weight slab prefetch registers
    float4 wreg[NW];
    auto load_w = [&](int it) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int j = tid + u * 256;
            if (WF4 % 256 == 0 || j < WF4) {
                const int row = j / (NT / 4), c4 = j - row * (NT / 4);
                wreg[u] = *reinterpret_cast<const float4*>(a.wp + ((long)it * TPS * CH + row) * a.NPad + co0 + c4 * 4);
            }
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int j = tid + u * 256;
            if (WF4 % 256 == 0 || j < WF4) *reinterpret_cast<float4*>(ws + buf * TPS * CH * NT + j * 4) = wreg[u];
        }
    };

    const int i = lane & 31, kk = lane >> 5;
    int hb[MT];                                       // halo index of this lane's voxel in each of the wave's row tiles
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int v = (wave * MT + m) * 32 + i;
        hb[m] = ((v >> 6) * HY + ((v >> 3) & 7)) * HX + (v & 7);
    }

    f32x16 acc[MT][NN];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    if (chunk0 < chunk1) load_w(it0);
    int buf = 0, it = it0;
    for (int chunk = chunk0; chunk < chunk1; ++chunk) {
        __syncthreads();   // every wave is done reading xs (previous chunk)
        stage_halo<CH, VEC, TZv>(xs, in_b, a.in_ps, a.in_cs, chunk * CH, a.Cin, z0, y0, x0, a.D, a.H, a.W, tid);
        for (int stg = 0; stg < SPC; ++stg, ++it) {
            store_w(buf);
            __syncthreads();
            if (it + 1 < niter) load_w(it + 1);
#pragma unroll
            for (int tt = 0; tt < TPS; ++tt) {
                const int tap = stg * TPS + tt;
                const float* xa[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) xa[m] = xs + (hb[m] + tap_halo_offset(tap)) * CP + kk;
                const float* wb = ws + (buf * TPS + tt) * CH * NT + kk * NT + i;
                // (fragment reads are left to hipcc's placement here: with several waves per SIMD the other waves cover each read's
                //  latency, and forcing all reads of the tap ahead of its MFMAs measured ~8 % slower)
#pragma unroll
                for (int s = 0; s < CH / 2; ++s) {
                    float av[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) av[m] = xa[m][2 * s];
#pragma unroll
                    for (int n = 0; n < NN; ++n) {
                        const float bv = wb[2 * s * NT + n * 32];
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv, acc[m][n], 0, 0, 0);
                    }
                }
            }
            buf ^= 1;
        }
    }

    // ---- epilogue: bias, store, per-tile BatchNorm partial statistics
    float* out_b = a.out + (long)b * a.out_bs;
    float ssum[NN], ssq[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        const int co = co0 + n * 32 + i;
        const bool cok = co < a.Cout;
        const float bv = (a.bias != nullptr && cok && split == 0) ? a.bias[co] : 0.f;
        const bool fuse = a.coef != nullptr && cok;
        const float fsc = fuse ? a.coef[2 * a.Cout + co] : 1.f, fsh = fuse ? a.coef[3 * a.Cout + co] : 0.f;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
                const int vv = (wave * MT + m) * 32 + row;
                const int gz = z0 + (vv >> 6), gy = y0 + ((vv >> 3) & 7), gx = x0 + (vv & 7);
                if (cok && gz < a.D && gy < a.H && gx < a.W) {
                    float val = acc[m][n][r] + bv;
                    const long vox = (long)(gz * a.H + gy) * a.W + gx;
                    if (a.ksplit > 1) a.part[(((long)split * a.B + b) * a.D * a.H * a.W + vox) * a.Cout + co] = val;   // this split's partial sum
                    else {
                        if (fuse) {                       // the arithmetic of bn_lrelu_apply_kernel
                            const float t = val * fsc + fsh;
                            val = t > 0.f ? t : t * a.slope;
                        }
                        out_b[vox * a.out_ps + (long)co * a.out_cs] = val;
                    }
                    s += val;
                    q += val * val;
                }
            }
        }
        ssum[n] = s + __shfl_xor(s, 32, 64);
        ssq[n] = q + __shfl_xor(q, 32, 64);
    }
    if (a.stats != nullptr && a.ksplit == 1) {
        __syncthreads();               // ws no longer read by any wave
        float* red = ws;               // [4 waves][2][NT]
        if (lane < 32) {
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                red[(wave * 2 + 0) * NT + n * 32 + i] = ssum[n];
                red[(wave * 2 + 1) * NT + n * 32 + i] = ssq[n];
            }
        }
        __syncthreads();
        if (tid < 2 * NT) {
            const int which = tid / NT, c = tid - which * NT;
            if (co0 + c < a.Cout) {
                const float tot = red[(0 * 2 + which) * NT + c] + red[(1 * 2 + which) * NT + c] + red[(2 * 2 + which) * NT + c] +
                                  red[(3 * 2 + which) * NT + c];
                a.stats[((long)tile_lin * 2 + which) * a.Cout + co0 + c] = tot;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ the 2- / 3-channel input layers, persistent
// conv3d_k3_mfma<2 | 4, 32> above spends a tile's time around its matrix instructions (2 -> 32 at 160^3: 54 MFMAs per wave and tile = 90 us of
// matrix time, 524 MB of output = 70 us of HBM, 258 us measured): one tile per workgroup, the halo's gather, the 27 weight slabs and 32 dword
// stores per lane all exposed.  This variant (whole 4 x 8 x 8 tiles, a multiple of 32 output channels, channels-last fp32 output) keeps
//   * the weights in REGISTERS for the workgroup's lifetime (a lane's B operand of tap t, k-step s is one float: 27 * CH / 2 registers),
//   * persistent workgroups whose next halo (600 voxels x CH floats: 5 - 10 dword buffer loads per thread, zeros from beyond num_records
//     outside the volume) travels to registers during the matrix loop and into the OTHER of two LDS images right behind the tile's barrier:
//     one barrier per tile,
//   * an epilogue that passes each wave's 64 x 32 slab through a wave-private LDS image and stores 16 bytes per lane (eight
//     buffer_store_dwordx4 of 8 voxels x 128 bytes per wave instead of 32 dword stores); statistics' partial sums are added behind the next
//     tile's barrier.
template <int CH>
constexpr size_t smallk_pw_lds_bytes() { return (size_t)(2 * (((TZ + 4) * HY * HX * (CH + 1) + 3) & ~3) + 4 * 64 * 32 + 2 * 4 * 2 * 32) * sizeof(float); }

template <int CH>
__global__ __launch_bounds__(256, CH == 2 ? 3 : 2) void conv3d_k3_smallk_pw(ConvArgs a) {
    constexpr int CP = CH + 1, KS = CH / 2, MT = 2, NT = 32;
    constexpr int PHV = 6 * HY * HX;                    // halo voxels of a 4 x 8 x 8 tile
    constexpr int XS = (PHV * CP + 3) & ~3;
    constexpr int NHP = (PHV * CH + 255) / 256;         // halo floats per thread
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                   // [2][PHV][CP]
    float* xch_all = smem + 2 * XS;                     // [4 waves][64 voxels][32 channels]: the epilogue's exchange images
    float* red_all = xch_all + 4 * 64 * 32;             // [tile parity][4 waves][2][32]: wave 0 reads a tile's sums behind the NEXT tile's barrier while the
                                                        // other waves may already be writing that tile's - into the other half

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int nwork = a.B * a.ntz * a.nty * a.ntx * a.ncot, nwg = gridDim.x;

    struct Tile { int tile_lin, b, z0, y0, x0, co0; };
    auto describe = [&](int work) {
        Tile t;
        const int cot = work % a.ncot;
        t.tile_lin = work / a.ncot;
        int q = t.tile_lin;
        const int tx_ = q % a.ntx; q /= a.ntx;
        const int ty_ = q % a.nty; q /= a.nty;
        const int tz_ = q % a.ntz;
        t.b = q / a.ntz;
        t.z0 = tz_ * 4; t.y0 = ty_ * TY; t.x0 = tx_ * TX;
        t.co0 = cot * NT;
        return t;
    };

    // ---- this thread's halo floats: j = channel * PHV + halo voxel (voxel fastest: planar operands are read in runs along x)
    const int in_bytes = (int)((((long)a.Cin - 1) * a.in_cs + ((long)a.D * a.H * a.W - 1) * a.in_ps + 1) * 4);
    unsigned hrel[NHP], hbit[NHP];
    int hlds[NHP];
#pragma unroll
    for (int u = 0; u < NHP; ++u) {
        const int j = tid + u * 256;
        const int c = j / PHV, hv = j - c * PHV;
        const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
        hrel[u] = (unsigned)(((long)((hz * a.H + hy) * a.W + hx) * a.in_ps + (long)c * a.in_cs) * 4);
        hbit[u] = (j < PHV * CH && c < a.Cin) ? (1u << hz) | (1u << (6 + hy)) | (1u << (16 + hx)) : 0x80000000u;
        hlds[u] = hv * CP + c;
    }
    float hreg[NHP];
    auto load_halo = [&](const Tile& t) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + (long)t.b * a.in_bs), 0, in_bytes, 0x00020000);
        auto run = [](int first, int extent, int n) {   // bits h in [0, n) with 0 <= first + h < extent
            const int lo = first < 0 ? -first : 0, hi = extent - first < n ? extent - first : n;
            return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
        };
        const unsigned mask = run(t.z0 - 1, a.D, 6) | (run(t.y0 - 1, a.H, HY) << 6) | (run(t.x0 - 1, a.W, HX) << 16);
        const unsigned origin = (unsigned)((long)(((t.z0 - 1) * a.H + (t.y0 - 1)) * a.W + (t.x0 - 1)) * a.in_ps * 4);      // modulo 2^32
#pragma unroll
        for (int u = 0; u < NHP; ++u)
            hreg[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)((mask & hbit[u]) == hbit[u] ? origin + hrel[u] : OOB), 0, 0));
    };
    auto store_halo = [&](float* img) {
#pragma unroll
        for (int u = 0; u < NHP; ++u)
            if (u + 1 < NHP || tid + u * 256 < PHV * CH) img[hlds[u]] = hreg[u];
    };

    int xa[MT];                                         // this lane's A operand of slab m at tap (0, 0, 0), k-step 0: floats into an image
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int v = (wave * MT + m) * 32 + i;
        xa[m] = (((v >> 6) * HY + ((v >> 3) & 7)) * HX + (v & 7)) * CP + kk;
    }

    int work = pulpo::xcd_remap(blockIdx.x, nwg);
    Tile cur = describe(work);
    float wreg[27][KS];
    int wco0 = -1;
    load_halo(cur);
    store_halo(xs);
    int buf = 0;
    int pend_tile = -1, pend_co0 = 0;
    auto flush_stats = [&]() {
        if (a.stats != nullptr && pend_tile >= 0 && tid < 2 * NT) {
            const int which = tid / NT, c = tid - which * NT;
            const float* red = red_all + (buf ^ 1) * (4 * 2 * NT);      // (buf has been flipped since the pending tile wrote its sums)
            const float tot = red[(0 * 2 + which) * NT + c] + red[(1 * 2 + which) * NT + c] + red[(2 * 2 + which) * NT + c] + red[(3 * 2 + which) * NT + c];
            a.stats[((long)pend_tile * 2 + which) * a.Cout + pend_co0 + c] = tot;
        }
    };
    float* xch = xch_all + wave * (64 * 32);

    for (;;) {
        const int next_work = work + nwg;
        const bool has_next = next_work < nwork;
        const Tile nxt = has_next ? describe(next_work) : cur;
        if (cur.co0 != wco0) {                          // (uniform; once per workgroup unless the layer has several cout tiles)
#pragma unroll
            for (int tap = 0; tap < 27; ++tap)
#pragma unroll
                for (int s = 0; s < KS; ++s) wreg[tap][s] = a.wp[(long)(tap * CH + 2 * s + kk) * a.NPad + cur.co0 + i];
            wco0 = cur.co0;
        }
        const int co = cur.co0 + i;
        const float bias = a.bias != nullptr ? a.bias[co] : 0.f;
        const bool fused = a.coef != nullptr;
        const float fsc = fused ? a.coef[2 * a.Cout + co] : 1.f, fsh = fused ? a.coef[3 * a.Cout + co] : 0.f;

        __syncthreads();                                // this tile's image is in place; every wave has left the other one
        flush_stats(); pend_tile = -1;
        load_halo(nxt);                                 // (after the last tile: its own halo again, into an image nobody reads)
        const float* img = xs + buf * XS;

        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int off = tap_halo_offset(tap) * CP;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(img[xa[m] + off + 2 * s], wreg[tap][s], acc[m], 0, 0, 0);
        }
        store_halo(xs + (buf ^ 1) * XS);                // (the other image was last read before this tile's barrier)

        // ---- epilogue.  Accumulator row r of slab m = voxel (x = (r & 3) + 4 kk, y = 4 m + (r >> 2)) of the wave's z plane, at channel i.
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(a.out + (long)cur.b * a.out_bs, 0, (int)((long)a.D * a.H * a.W * a.out_ps * 4), 0x00020000);
        const unsigned ops_b = (unsigned)a.out_ps * 4u, row_b = (unsigned)a.W * ops_b;
        const unsigned obase = (unsigned)(((cur.z0 + wave) * a.H + cur.y0) * a.W + cur.x0) * ops_b + (unsigned)cur.co0 * 4u;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float val = acc[m][r] + bias;
                if (fused) {
                    const float t = val * fsc + fsh;
                    val = t > 0.f ? t : t * a.slope;
                }
                s += val;
                q += val * val;
                xch[(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk) * 32 + i] = val;
            }
        }
        // 16-byte pieces: piece p = lane + 64 t covers voxel row p / 8 of the slab pair (y = row / 8, x = row % 8), channels 4 (p % 8) ..
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(xch + (lane + 64 * t) * 4);
            const unsigned voff = obase + (unsigned)(lane >> 3) * ops_b + (unsigned)(lane & 7) * 16u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), ors, (int)voff, (int)((unsigned)t * row_b), 0);
        }
        if (a.stats != nullptr) {
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            float* red = red_all + buf * (4 * 2 * NT);
            if (lane < 32) { red[(wave * 2 + 0) * NT + i] = s; red[(wave * 2 + 1) * NT + i] = q; }
            pend_tile = cur.tile_lin; pend_co0 = cur.co0;
        }
        buf ^= 1;
        if (!has_next) break;
        work = next_work;
        cur = nxt;
    }
    __syncthreads();
    flush_stats();
}

// split-K finish: out = sum_s part[s] in fixed order (deterministic), plus the per-row (sum, sum of squares) BatchNorm partials.
// Row r of stats covers voxels [r*V/nrow, (r+1)*V/nrow): any partition is fine for the double-precision finalize.
template <typename TO = float>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ksplit, TO* __restrict__ out, long obs, long ops,
                                                              long ocs, int B, long V, int C, int nrow, float* __restrict__ stats,
                                                              const float* __restrict__ coef, float slope) {
    // grid = (nrow, ceil(C/32)): one workgroup per (voxel slice, 32-channel group); 32 channels x 8 voxel lanes
    __shared__ float red[2][256];
    const int r = blockIdx.x, c0 = blockIdx.y * 32;
    const long npix = (long)B * V;
    const long p0 = npix * r / nrow, p1 = npix * (r + 1) / nrow;
    const int c = c0 + (threadIdx.x & 31), prow = threadIdx.x >> 5;
    float s = 0.f, q = 0.f;
    const bool fuse = coef != nullptr && c < C;
    const float fsc = fuse ? coef[2 * C + c] : 1.f, fsh = fuse ? coef[3 * C + c] : 0.f;
    if (c < C)
        for (long p = p0 + prow; p < p1; p += 8) {
            float v = 0.f;
            for (int k = 0; k < ksplit; ++k) v += part[((long)k * npix + p) * C + c];
            const long b = p / V, vox = p - b * V;
            if (fuse) {
                const float t = pulpo::as_stored<TO>(v) * fsc + fsh;
                v = t > 0.f ? t : t * slope;
            }
            v = pulpo::as_stored<TO>(v);
            float v1[1] = {v};
            pulpo::stv<1>(out + b * obs + vox * ops + (long)c * ocs, v1);
            s += v;
            q += v * v;
        }
    if (stats != nullptr) {
        red[0][threadIdx.x] = s;
        red[1][threadIdx.x] = q;
        __syncthreads();
        if (threadIdx.x < 64) {
            const int which = threadIdx.x >> 5, cc = threadIdx.x & 31;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[which][k * 32 + cc];
            if (c0 + cc < C) stats[((long)r * 2 + which) * C + c0 + cc] = t;
        }
    }
}

// channels-last, 16-byte aligned output with C % 4 == 0: four channels per lane (8 lanes x 16 bytes = one voxel's 128-byte line), 32 voxels
// per pass, the slabs' loads of a pass independent of each other (the scalar kernel above walks 8 voxels per pass: at the 20^3 / 10^3
// levels its 20 passes of dependent loads took 15 - 19 us per launch)
template <typename TO = float>
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const float* __restrict__ part, int ksplit, TO* __restrict__ out, long obs, long ops,
                                                                  int B, long V, int C, int nrow, float* __restrict__ stats,
                                                                  const float* __restrict__ coef, float slope) {
    __shared__ float4 red[2][256];
    const int r = blockIdx.x, c0 = blockIdx.y * 32;
    const long npix = (long)B * V;
    const long p0 = npix * r / nrow, p1 = npix * (r + 1) / nrow;
    const int q = threadIdx.x & 7, prow = threadIdx.x >> 3;
    const int c = c0 + 4 * q;
    const bool cok = c < C;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sq = s;
    const bool fuse = coef != nullptr && cok;
    const float4 fsc = fuse ? *reinterpret_cast<const float4*>(coef + 2 * C + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 fsh = fuse ? *reinterpret_cast<const float4*>(coef + 3 * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok)
        for (long p = p0 + prow; p < p1; p += 32) {
            float4 v = *reinterpret_cast<const float4*>(part + p * C + c);
            for (int k = 1; k < ksplit; ++k) {
                const float4 t = *reinterpret_cast<const float4*>(part + ((long)k * npix + p) * C + c);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            const long b = p / V, vox = p - b * V;
            if (fuse) {
                auto act = [&](float x, float sc, float sh) { const float t = pulpo::as_stored<TO>(x) * sc + sh; return t > 0.f ? t : t * slope; };
                v = make_float4(act(v.x, fsc.x, fsh.x), act(v.y, fsc.y, fsh.y), act(v.z, fsc.z, fsh.z), act(v.w, fsc.w, fsh.w));
            }
            v = make_float4(pulpo::as_stored<TO>(v.x), pulpo::as_stored<TO>(v.y), pulpo::as_stored<TO>(v.z), pulpo::as_stored<TO>(v.w));
            const float v4[4] = {v.x, v.y, v.z, v.w};
            pulpo::stv<4>(out + b * obs + vox * ops + c, v4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            sq.x += v.x * v.x; sq.y += v.y * v.y; sq.z += v.z * v.z; sq.w += v.w * v.w;
        }
    if (stats != nullptr) {
        red[0][threadIdx.x] = s;
        red[1][threadIdx.x] = sq;
        __syncthreads();
        if (threadIdx.x < 16) {
            const int which = threadIdx.x >> 3, qq = threadIdx.x & 7;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (int k = 0; k < 32; ++k) { const float4 u = red[which][k * 8 + qq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            if (c0 + 4 * qq < C) *reinterpret_cast<float4*>(stats + ((long)r * 2 + which) * C + c0 + 4 * qq) = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight packing
// w: PyTorch layout [Cout][Cin][27].  forward : K = Cin,  N = Cout, wp[k/CH][tap][k%CH][n] = w[n][k][tap]
//                                     dgrad   : K = Cout, N = Cin,  wp[k/CH][tap][k%CH][n] = w[k][n][26 - tap]
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int CH, int NPad, int dgrad,
                                   long total) {
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e % NPad);
        long r = e / NPad;
        const int kc = (int)(r % CH); r /= CH;
        const int tap = (int)(r % 27);
        const int chunk = (int)(r / 27);
        const int k = chunk * CH + kc;
        float val = 0.f;
        if (k < K && n < N) val = dgrad ? w[((long)k * Cin + n) * 27 + (26 - tap)] : w[((long)n * Cin + k) * 27 + tap];
        wp[e] = val;
    }
}

// Cin chunk staged per pass: 16 channels (27 KB halo tile + 8 KB weight double buffer => 4 workgroups per CU; measured
// faster than 32- and 8-channel chunks on MI355X), 2 / 4 for the 2- / 3-channel input layers
int pick_ch(int K) { return direct_ch(K); }

template <int CH, int NT, bool VEC, int TZv>
int launch_conv_tz(const ConvArgs& a, int nblk, hipStream_t st) {
    constexpr size_t lds = (size_t)((((TZv + 2) * HY * HX * (CH + 1) + 3) & ~3) + 2 * (CH <= 4 ? 27 : 1) * CH * NT) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_mfma<CH, NT, VEC, TZv>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(conv3d): %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3d_k3_mfma<CH, NT, VEC, TZv>), dim3(nblk), dim3(256), lds, st, a);
    return pulpo::check_launch("conv3d_k3_mfma");
}

template <int CH, int NT, bool VEC>
int launch_conv(const ConvArgs& a, int nblk, hipStream_t st, int tz) {
    if (tz == 4) return launch_conv_tz<CH, NT, VEC, 4>(a, nblk, st);
    return launch_conv_tz<CH, NT, VEC, 2>(a, nblk, st);
}

}  // namespace


// z extent of the forward voxel tile: 4 (the Winograd kernel's tile; in the direct kernel each wave = two 32-voxel MFMA row tiles)
// when the volume's depth divides evenly and there are enough tiles (measured: the (y, x) Winograd kernel pays from 20^3 up)
template <int CH>
int launch_smallk_pw(const ConvArgs& a, long nwork, hipStream_t st) {
    constexpr size_t lds = smallk_pw_lds_bytes<CH>();
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_smallk_pw<CH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(conv3d smallk pw): %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int nwg = (int)std::min<long>(nwork, CH == 2 ? 768 : 512);       // persistent workgroups: three (two) per CU
    hipLaunchKernelGGL((conv3d_k3_smallk_pw<CH>), dim3(nwg), dim3(256), lds, st, a);
    return pulpo::check_launch("conv3d_k3_smallk_pw");
}

int pulpo_conv::conv_tz(int D, int H, int W) {
    return (D % 4 == 0 && (long)D * H * W >= 20L * 20 * 20) ? 4 : 2;
}

int pulpo_conv::launch_splitk_reduce(const float* part, int ksplit, void* out, long obs, long ops, long ocs, int B, long V, int C, int nrow,
                                     float* stats, const float* coef, float slope, hipStream_t st, int out_dt) {
    const int eb = out_dt ? 2 : 4;
    const bool vec = ocs == 1 && C % 4 == 0 && ops % 4 == 0 && obs % 4 == 0 && (((uintptr_t)part | (uintptr_t)stats | (uintptr_t)coef) & 15) == 0 &&
                     ((uintptr_t)out % (4 * eb)) == 0;
    const dim3 grid(nrow, pulpo::cdiv(C, 32));
    if (out_dt) {
        pulpo::bf16_t* o = (pulpo::bf16_t*)out;
        if (vec) hipLaunchKernelGGL((splitk_reduce_vec_kernel<pulpo::bf16_t>), grid, dim3(256), 0, st, part, ksplit, o, obs, ops, B, V, C, nrow, stats, coef, slope);
        else hipLaunchKernelGGL((splitk_reduce_kernel<pulpo::bf16_t>), grid, dim3(256), 0, st, part, ksplit, o, obs, ops, ocs, B, V, C, nrow, stats, coef, slope);
    } else {
        float* o = (float*)out;
        if (vec) hipLaunchKernelGGL((splitk_reduce_vec_kernel<float>), grid, dim3(256), 0, st, part, ksplit, o, obs, ops, B, V, C, nrow, stats, coef, slope);
        else hipLaunchKernelGGL((splitk_reduce_kernel<float>), grid, dim3(256), 0, st, part, ksplit, o, obs, ops, ocs, B, V, C, nrow, stats, coef, slope);
    }
    return pulpo::check_launch("splitk_reduce");
}

// ================================================================================================ C ABI
PULPO_API size_t pulpo_conv3d_k3_packed_floats(int K, int N) {
    const int CH = pick_ch(K);
    return (size_t)((K + CH - 1) / CH) * 27 * CH * npad(N);
}

PULPO_API int pulpo_conv3d_k3_pack_weight(const float* w, float* wp, int Cin, int Cout, int dgrad, void* stream) {
    PULPO_REQUIRE(w && wp && Cin > 0 && Cout > 0, "pack_weight: bad arguments");
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const int CH = pick_ch(K), NP = npad(N);
    const long total = (long)pulpo_conv3d_k3_packed_floats(K, N);
    const int nblk = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, w, wp, Cin, Cout, CH, NP, dgrad, total);
    return pulpo::check_launch("pack_weight");
}

// which kernel instantiation pulpo_conv3d_k3_fwd dispatches to: CH * 1000 + NT (vector/scalar staging is decided by the strides)
PULPO_API int pulpo_conv3d_k3_tile_config(int K, int N) {
    const int NT = (N % 64 == 0) ? 64 : 32;        // 64-wide cout tiles only when none would be half empty
    return pick_ch(K) * 1000 + NT;
}

PULPO_API int pulpo_conv3d_k3_stat_tiles(int B, int D, int H, int W);

// small volumes (the 20^3 / 10^3 pyramid levels) have too few tiles to fill 256 CUs x 4 workgroups: split the Cin chunks
// over several workgroups per tile (deterministic: partial slabs + ordered reduce)
int conv_ksplit(int B, int D, int H, int W, int K, int N) {
    const int cfg = pulpo_conv3d_k3_tile_config(K, N);
    const int CH = cfg / 1000, NT = cfg % 1000;
    const long nblk = (long)B * pulpo::cdiv(D, conv_tz(D, H, W)) * pulpo::cdiv(H, TY) * pulpo::cdiv(W, TX) * pulpo::cdiv(N, NT);
    const int nchunk = (K + CH - 1) / CH;
    if (nblk >= 512 || nchunk <= 1) return 1;
    return (int)std::max<long>(1, std::min<long>(std::min(nchunk, 8), 1024 / nblk));   // one resident round of <= 1024 workgroups
}

PULPO_API size_t pulpo_conv3d_k3_fwd_scratch_floats(int B, int D, int H, int W, int K, int N) {
    const int ks = conv_ksplit(B, D, H, W, K, N);
    return ks > 1 ? (size_t)ks * B * D * H * W * N : 0;
}

// Generic entry: computes out[b][vox][n] = sum_{tap,k} in[b][vox+tap-1][k] * wp[...] (+ bias[n]).
// K / N are the GEMM's reduction / output channel counts (forward: Cin/Cout; dgrad: Cout/Cin with dgrad-packed wp).
static int conv_fwd_impl(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias, float* out,
                         int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats, float* scratch, const float* coef, float slope, int B,
                         int D, int H, int W, int K, int N, void* stream) {
    PULPO_REQUIRE(in && wp && out, "conv3d_k3_fwd: null pointer");
    PULPO_REQUIRE(!(coef && stats), "conv3d_k3_fwd: batch statistics are not available from the fused eval-mode epilogue");
    PULPO_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && K > 0 && N > 0, "conv3d_k3_fwd: bad dims");
    ConvArgs a{};
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps; a.in_cs = in_cs;
    a.wp = wp; a.bias = bias;
    a.out = out; a.out_bs = out_bs; a.out_ps = out_ps; a.out_cs = out_cs;
    a.stats = stats;
    a.coef = coef; a.slope = slope;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = K; a.Cout = N; a.NPad = npad(N);
    const int cfg = pulpo_conv3d_k3_tile_config(K, N);
    const int CH = cfg / 1000, NT = cfg % 1000;
    const int tz = conv_tz(D, H, W);
    a.ntz = pulpo::cdiv(D, tz); a.nty = pulpo::cdiv(H, TY); a.ntx = pulpo::cdiv(W, TX);
    a.ncot = pulpo::cdiv(N, NT);
    const long nblk_l = (long)B * a.ntz * a.nty * a.ntx * a.ncot;
    PULPO_REQUIRE(nblk_l < (1L << 31), "conv3d_k3_fwd: grid too large");
    a.ksplit = conv_ksplit(B, D, H, W, K, N);
    a.part = scratch;
    PULPO_REQUIRE(a.ksplit == 1 || scratch != nullptr, "conv3d_k3_fwd: scratch of pulpo_conv3d_k3_fwd_scratch_floats() floats required");
    const int nblk = (int)nblk_l * a.ksplit;
    const bool vec = (in_cs == 1) && (in_ps % 4 == 0) && (in_bs % 4 == 0) && (K % 4 == 0) && (((uintptr_t)in & 15) == 0) && CH >= 16;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // the persistent kernel of the 2- / 3-channel input layers (PULPO_CONV_SMALLK_PW=0: the one-tile-per-workgroup kernel, A/B switch)
    static int spw = -1;
    if (spw < 0) { const char* e = getenv("PULPO_CONV_SMALLK_PW"); spw = e ? atoi(e) : 1; }
    const long in_span = (((long)K - 1) * in_cs + ((long)D * H * W - 1) * in_ps + 1) * 4;
    if (spw && CH <= 4 && tz == 4 && N % 32 == 0 && D % 4 == 0 && H % TY == 0 && W % TX == 0 && out_cs == 1 && out_ps % 4 == 0 && out_bs % 4 == 0 &&
        (((uintptr_t)out & 15) == 0) && in_span > 0 && in_span < (1L << 31) && in_cs >= 0 && in_ps >= 0 && (long)D * H * W * out_ps * 4 < (1L << 31)) {
        a.ncot = N / 32;
        const long nwork = (long)B * a.ntz * a.nty * a.ntx * a.ncot;
        return CH == 2 ? launch_smallk_pw<2>(a, nwork, st) : launch_smallk_pw<4>(a, nwork, st);
    }
    if (CH == 2) rc = NT == 64 ? launch_conv<2, 64, false>(a, nblk, st, tz) : launch_conv<2, 32, false>(a, nblk, st, tz);
    else if (CH == 4) rc = NT == 64 ? launch_conv<4, 64, false>(a, nblk, st, tz) : launch_conv<4, 32, false>(a, nblk, st, tz);
    else if (vec) rc = NT == 64 ? launch_conv<16, 64, true>(a, nblk, st, tz) : launch_conv<16, 32, true>(a, nblk, st, tz);
    else rc = NT == 64 ? launch_conv<16, 64, false>(a, nblk, st, tz) : launch_conv<16, 32, false>(a, nblk, st, tz);
    if (rc == 0 && a.ksplit > 1) {
        rc = pulpo_conv::launch_splitk_reduce(scratch, a.ksplit, out, (long)out_bs, (long)out_ps, (long)out_cs, B, (long)D * H * W, N,
                                              pulpo_conv3d_k3_stat_tiles(B, D, H, W), stats, coef, slope, st);
    }
    return rc;
}

PULPO_API int pulpo_conv3d_k3_fwd(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias,
                                  float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats, float* scratch, int B, int D, int H,
                                  int W, int K, int N, void* stream) {
    return conv_fwd_impl(in, in_bs, in_ps, in_cs, wp, bias, out, out_bs, out_ps, out_cs, stats, scratch, nullptr, 0.f, B, D, H, W, K, N, stream);
}

// eval-mode ConvUnit in one kernel: out = LeakyReLU(BatchNorm_eval(conv + bias)) with coef from pulpo_bn_eval_coef
PULPO_API int pulpo_conv3d_k3_fwd_bn_lrelu(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias,
                                           const float* coef, float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs,
                                           float* scratch, int B, int D, int H, int W, int K, int N, void* stream) {
    PULPO_REQUIRE(coef, "conv3d_k3_fwd_bn_lrelu: null coef");
    return conv_fwd_impl(in, in_bs, in_ps, in_cs, wp, bias, out, out_bs, out_ps, out_cs, nullptr, scratch, coef, slope, B, D, H, W, K, N, stream);
}

PULPO_API int pulpo_conv3d_k3_stat_tiles(int B, int D, int H, int W) {
    return B * pulpo::cdiv(D, conv_tz(D, H, W)) * pulpo::cdiv(H, TY) * pulpo::cdiv(W, TX);
}

// forward / data-gradient kernel for a shape: 3 = Winograd F(2x2x2,3x3x3) in all three axes (conv3d_wino3.hip: whole 4x8x8 tiles, from 64
// reduction channels up, channels-last operands), 2 = Winograd F(2x2,3x3) in (y, x) (where it applies: 4x8x8-tiled volumes, > 4 reduction
// channels; the 10^3 level - depth not a multiple of 4 - where the pipelined kernel runs it with split-K work items), 0 = direct implicit GEMM.  (1, F(2,3) along x only, was retired in round 3.)
PULPO_API int pulpo_conv3d_k3_algo(int B, int D, int H, int W, int K, int N) {
    static int force = -1;
    if (force < 0) { const char* e = getenv("PULPO_CONV_WINOGRAD"); force = e ? atoi(e) + 1 : 0; }     // unset: policy; 0 / 1: force off / on
    if (force == 1) return 0;
    if (pulpo_conv::wino3_shape_ok(B, D, H, W, K, N)) return 3;
    return (K > 4 && (conv_tz(D, H, W) == 4 || pulpo_conv::wino2_ragged_depth_ok(B, D, H, W, K, N))) ? 2 : 0;
}
