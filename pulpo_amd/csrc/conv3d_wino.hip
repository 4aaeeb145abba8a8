// Winograd F(2x2,3x3) in (y, x) of the 3x3x3 convolution (forward and data gradient) for volumes tiled 4x8x8, direct over the z taps, all
// arithmetic fp32 on v_mfma_f32_32x32x2_f32.  This file: the C entry points, the weight packing, and conv3d_k3_wino2_mfma - the round-2
// kernel (LDS-DMA weight slabs, one halo image, four barriers per chunk), which serves the operands the pipelined kernel of
// conv3d_wino2p.hip does not take: planar inputs, channel counts that are not multiples of 8, volumes of 2 GiB and more.
// Same argument block, tile order, BatchNorm partial statistics and fused eval-mode epilogue as the direct kernel in conv3d.hip.
#include "conv_shared.h"
#include "../../include/pulpo_hip.h"
#include "wino3_pack.h"
#include <stdlib.h>


#ifndef PULPO_ABL
#define PULPO_ABL 0          // diagnostic builds (scripts/ablate.py): 9 = in-kernel stamps of the (y, x) Winograd kernel
#endif
#if PULPO_ABL == 9
// g_stamps[block][0] = HW_REG_HW_ID, [1] = HW_REG_XCC_ID, [2] = start clock, [3 + 2k] / [4 + 2k] = main-loop end / tile end of the block's k-th tile
__device__ unsigned long long g_stamps[512 * 80];
#define STAMP(slot, val) do { if (threadIdx.x == 0 && (slot) < 80) g_stamps[blockIdx.x * 80 + (slot)] = (val); } while (0)
PULPO_API int pulpo_debug_read_stamps(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#else
#define STAMP(slot, val) do {} while (0)
#endif

namespace {

using namespace pulpo_conv;

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS starting at the WAVE-UNIFORM byte address lds_addr.
// Issued through inline asm on purpose: behind the builtin, hipcc makes every following ds_read wait for vmcnt(0) (the DMA may alias it),
// which would expose the whole global latency at every use; the kernel waits for its DMA pieces itself (s_waitcnt vmcnt(0) in front of
// the barrier that hands a slab over).  Compiler-issued loads in flight at the same time are only ever over-waited for (in-order return).
__device__ __forceinline__ void dma16(const float* src, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_address(const float* p) {
    return (unsigned)(uintptr_t)((const __attribute__((address_space(3))) float*)p);
}

// F(2,3) along one axis: two neighbouring outputs from four transformed inputs and four transformed weights
//   input  (staging)  : v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3
//   weights (packing) : u0 = g0, u1 = (g0 + g1 + g2)/2, u2 = (g0 - g1 + g2)/2, u3 = g2
//   output            : y_even = m0 + m1 + m2, y_odd = m1 - m2 - m3
// applied along x while the halo is staged (LDS rows ordered (hz, px, hy, x-pair)) and along y by the waves (wave py owns the points (py, 0..3))
constexpr int WN_CH = 8, WN_HZ = 6, WN_PL = HY * 4;

// staging items of a halo chunk (vector path): (hz, hy, x-pair, channel quad), WN_NIT per thread
constexpr int WN_Q = WN_CH / 4, WN_NITEM = WN_HZ * HY * 4 * WN_Q, WN_NIT = (WN_NITEM + 255) / 256;

// ---- LDS image of the (y, x) kernel: the same (hz, px, hy, x-pair) row order as above, but rows of 12 floats (8 channels + 4 pad = 48 bytes,
// 16-byte aligned) and hz planes of 164 rows, so that the MFMA operands of FOUR consecutive k-steps arrive with ONE ds_read_b128: lane (row i,
// half kk) reads channels 4 kk .. 4 kk + 3 of its row and k-step s uses channel 4 kk + s (the weights are packed to match).  A row is three
// 16-byte bank slots and 3 is a unit mod 16, the planes are 4 rows apart mod 16: the 16 lanes of every ds_read_b128 lane group hit 16 distinct slots.
constexpr int W2_RS = 12, W2_PLROWS = 4 * WN_PL + 4, W2_PS = W2_PLROWS * W2_RS, W2_XS = WN_HZ * W2_PS;

__device__ __forceinline__ void w2_store_transformed(float* xs, const float4 (&d)[WN_NIT][4], int tid) {
#pragma unroll
    for (int u = 0; u < WN_NIT; ++u) {
        const int j = tid + u * 256;
        if (j < WN_NITEM) {
            const int q = j % WN_Q, rb = j / WN_Q;                // rb = (hz*HY + hy)*4 + xb
            const int hz = rb / (HY * 4), yx = rb - hz * (HY * 4);
            float* o = xs + hz * W2_PS + yx * W2_RS + 4 * q;      // row (hz, point 0, hy, xb); points are WN_PL rows apart
            const float4 d0 = d[u][0], d1 = d[u][1], d2 = d[u][2], d3 = d[u][3];
            *reinterpret_cast<float4*>(o) = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
            *reinterpret_cast<float4*>(o + WN_PL * W2_RS) = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
            *reinterpret_cast<float4*>(o + 2 * WN_PL * W2_RS) = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
            *reinterpret_cast<float4*>(o + 3 * WN_PL * W2_RS) = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
        }
    }
}

// scalar staging into the same image (planar inputs, channel counts that are not multiples of 4)
__device__ __forceinline__ void w2_stage_scalar(float* xs, const float* __restrict__ in, long in_ps, long in_cs, int c0, int Cin, int z0, int y0, int x0,
                                                int D, int H, int W, int tid) {
    for (int j = tid; j < WN_HZ * HY * 4 * WN_CH; j += 256) {
        const int c = j % WN_CH, rb = j / WN_CH;
        const int xb = rb & 3, hrow = rb >> 2;
        const int hz = hrow / HY, hy = hrow - hz * HY;
        const int gz = z0 - 1 + hz, gy = y0 - 1 + hy;
        float d[4] = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && c0 + c < Cin) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int gx = x0 - 1 + 2 * xb + t;
                if ((unsigned)gx < (unsigned)W) d[t] = in[((long)(gz * H + gy) * W + gx) * in_ps + (long)(c0 + c) * in_cs];
            }
        }
        float* o = xs + hz * W2_PS + (hy * 4 + xb) * W2_RS + c;
        o[0] = d[0] - d[2];
        o[WN_PL * W2_RS] = d[1] + d[2];
        o[2 * WN_PL * W2_RS] = d[2] - d[1];
        o[3 * WN_PL * W2_RS] = d[1] - d[3];
    }
}

// ------------------------------------------------------------------------------------------------ Winograd F(2x2,3x3) in (y, x)
// The y taps get the same treatment as the x taps: 16 transformed points per 2x2 output block, 3 (dz) x 16 matrix products per four
// outputs = 2.25x fewer than the direct kernel (1.5x fewer than F(2,3) along x alone).  The halo is staged x-transformed; WAVE py OWNS THE
// FOUR POINTS (py, px = 0..3) and forms the y combination of its A fragments as they are read (wave-uniform tap pair), for all 64 blocks
// of the 4x8x8 tile (two MFMA row tiles of 2 z-planes x 4 x 4 blocks).  The x inverse transform is in-lane; the y inverse transform sums
// over the four waves through LDS, one row tile at a time.
//
// What the phase stamps of a diagnostic build showed (scripts/ablate.py, scripts/stamps.py; 32->32 @160^3: 84k clocks per tile of which
// 49k are the two co-resident waves' matrix time), and what this version does about it:
//   * 2.5 ds_read_b32 per MFMA kept the LDS issue path, not the matrix pipe, busy  -> operand rows of four k-steps by ONE ds_read_b128
//     (row / plane strides chosen for a conflict-free bank map), 0.63 reads per MFMA;
//   * a scalar-store epilogue of 16k clocks (ds_read -> wait -> 64-bit address -> 4-byte store, 32 times per lane)  -> rows re-mapped so
//     that a lane holds four channels: ds_read_b128 + 16-byte stores, one 128-byte line per 8 lanes;
//   * weight slabs through registers + ds_write, and hipcc's vmcnt(0) in front of every LDS read that follows a DMA  -> LDS-DMA issued
//     from inline asm, one piece per point step behind the step's first MFMAs, waited for by hand at the hand-over barrier;
//   * 10k clocks of prologue per tile (argument loads, index arithmetic, first-touch latency of halo and weights)  -> PERSISTENT
//     workgroups (two per CU) that fetch the next tile's first slab and halo chunk during the last dz iteration of the current tile.
// A start-up offset of the second workgroup of each CU (to break the lockstep of the pair) was measured without effect and is not kept.
template <bool VEC, bool BNR = false>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wino2_mfma(ConvArgs a) {
    constexpr int CH = WN_CH, NT = 32;
    constexpr int XS = W2_XS;
    constexpr int WSL = 16 * CH * NT;                // floats of one dz weight slab set: [py][px][n][k]
    constexpr int RH = 4 * 2 * 16 * 64;              // floats of the cross-wave exchange buffer of ONE row tile: [py][ox][r][lane] (inside xs)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ws = smem + XS;
    static_assert(RH + 4 * 2 * NT <= XS, "exchange buffer + statistics rows must stay inside the halo image (the next tile's slab lands in ws)");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kk = lane >> 5;
    const int nchunk = (a.Cin + CH - 1) / CH;
    const int niter = nchunk * 3;
    const int nwork = a.B * a.ntz * a.nty * a.ntx * a.ncot;
    const int nwg = gridDim.x;
    const unsigned ws_lds = lds_address(ws);
    const unsigned ps_bytes = (unsigned)a.in_ps * 4u;
    const int rq = tid % WN_Q;

    // ---- tile-invariant per-lane data
    // halo gather (VEC): byte offsets of a thread's items relative to the tile's halo origin
    unsigned roff[VEC ? WN_NIT : 1];
    if constexpr (VEC) {
#pragma unroll
        for (int u = 0; u < WN_NIT; ++u) {
            const int j = tid + u * 256;
            const int xb = (j / WN_Q) & 3, hrow = j / (4 * WN_Q);
            const int hz = hrow / HY, hy = hrow - hz * HY;
            roff[u] = ((unsigned)((hz * a.H + hy) * a.W + 2 * xb) * (unsigned)a.in_ps + 4u * rq) * 4u;
        }
    }
    // y combination of this wave's point row: v = X[2 yb + ta] + sa * X[2 yb + tb]   (same table as the x transform)
    const int py = wave;
    const int ta = py == 0 ? 0 : py == 2 ? 2 : 1;
    const int tb = py == 2 ? 1 : py == 3 ? 3 : 2;
    const float sa = py == 1 ? 1.f : -1.f;
    // MFMA row i of row tile m = block (z = 2 m + (i >> 4), yb = (i >> 2) & 3, xb = i & 3); LDS rows are (hz, px, hy, xb)
    const int lrow = ((i >> 2) & 3) * 8 + (i & 3);
    const float* pa = xs + (i >> 4) * W2_PS + (lrow + ta * 4) * W2_RS + 4 * kk;
    const float* pb = xs + (i >> 4) * W2_PS + (lrow + tb * 4) * W2_RS + 4 * kk;
    const float* wbase = ws + ((py * 4) * NT + i) * CH + 4 * kk;

    // ---- per-tile descriptor
    struct Tile {
        int tile_lin, b, z0, y0, x0, co0;
        const char* origin;          // halo origin voxel (z0 - 1, y0 - 1, x0 - 1) of the input, channel 0 (may lie in front of the tensor)
        const float* wsrc;           // this wave's source of DMA piece 0 of slab 0 (uniform)
        unsigned rmask;              // in-volume bits of the thread's (item, tap) loads
    };
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
        t.origin = reinterpret_cast<const char*>(a.in + (long)t.b * a.in_bs) + ((long)((t.z0 - 1) * a.H + (t.y0 - 1)) * a.W + (t.x0 - 1)) * a.in_ps * 4;
        t.wsrc = a.wp + ((long)wave * a.NPad + t.co0) * CH;                      // (wave-uniform; the lane's 16 bytes are added at the issue)
        t.rmask = 0;
        if constexpr (VEC) {
#pragma unroll
            for (int u = 0; u < WN_NIT; ++u) {
                const int j = tid + u * 256;
                const int xb = (j / WN_Q) & 3, hrow = j / (4 * WN_Q);
                const int hz = hrow / HY, hy = hrow - hz * HY;
                const bool rowok = j < WN_NITEM && (unsigned)(t.z0 - 1 + hz) < (unsigned)a.D && (unsigned)(t.y0 - 1 + hy) < (unsigned)a.H;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
                    if (rowok && (unsigned)(t.x0 - 1 + 2 * xb + tt) < (unsigned)a.W) t.rmask |= 1u << (u * 4 + tt);
            }
        }
        return t;
    };

    // Weight slabs travel global -> LDS by LDS-DMA (no staging registers, no ds_write): slab `it` of a cout tile = 16 pieces (points) of
    // 1 KB ([32 n][8 k], contiguous in the packed weights); wave w copies pieces w, w + 4, w + 8, w + 12.
    auto dma_w_piece = [&](const float* wsrc, int it, int buf, int u) {
        dma16(wsrc + (long)it * 16 * CH * a.NPad + (long)u * 4 * CH * a.NPad + lane * 4, ws_lds + (unsigned)(buf * WSL + wave * 256 + u * 4 * 256) * 4u);
    };
    float4 raw[VEC ? WN_NIT : 1][4];
    // loads 2 * part, 2 * part + 1 of the eight (item, tap) loads of a halo chunk: the in-loop prefetch issues two per point step
    auto load_raw_part = [&](const Tile& t, int c0, int part) {
        const char* base = t.origin + (long)c0 * 4;
        const bool cok = c0 + 4 * rq < a.Cin;
#pragma unroll
        for (int k = 2 * part; k < 2 * part + 2; ++k) {
            const int u = k >> 2, tt = k & 3;
            raw[u][tt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cok && ((t.rmask >> (u * 4 + tt)) & 1u)) raw[u][tt] = *reinterpret_cast<const float4*>(base + (roff[u] + tt * ps_bytes));
        }
    };

    STAMP(0, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));        // HW_REG_HW_ID
    STAMP(1, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));       // HW_REG_XCC_ID
    STAMP(2, __builtin_amdgcn_s_memtime());
    [[maybe_unused]] int tile_no = 0;
    int work = pulpo::xcd_remap(blockIdx.x, nwg);           // static deal of the tiles: tile = block id + k * grid (remapped: an XCD's workgroups hold neighbouring tiles)
    Tile cur = describe(work);
#pragma unroll
    for (int u = 0; u < 4; ++u) dma_w_piece(cur.wsrc, 0, 0, u);
    if constexpr (VEC) {
#pragma unroll
        for (int part = 0; part < 4; ++part) load_raw_part(cur, 0, part);
    }
    int buf = 0;
    float4 ra[2][2], rb[2][2], rw[2];                   // two register sets of operand rows (activations: (ta, tb) x two row tiles; weights)

    for (;;) {
        int next_work = nwork;
        bool has_next = false;
        Tile nxt = cur;

        f32x16 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][p][r] = 0.f;

        const float* in_b = a.in + (long)cur.b * a.in_bs;
        int it = 0;
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            __syncthreads();                            // every wave has finished reading xs (previous chunk / previous tile's exchange)
            if constexpr (VEC) {
                // every thread "uses" its raw registers here, unconditionally: the compiler's wait for those loads then sits in straight-line
                // code, and it does not have to assume them still in flight (and drain the queue, DMA included) when they are reloaded
#pragma unroll
                for (int u = 0; u < WN_NIT; ++u)
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        asm volatile("" : : "v"(raw[u][tt].x), "v"(raw[u][tt].y), "v"(raw[u][tt].z), "v"(raw[u][tt].w));
                w2_store_transformed(xs, raw, tid);
            } else {
                w2_stage_scalar(xs, in_b, a.in_ps, a.in_cs, chunk * CH, a.Cin, cur.z0, cur.y0, cur.x0, a.D, a.H, a.W, tid);
            }
            const bool last_chunk = chunk + 1 == nchunk;
#pragma unroll
            for (int dz = 0; dz < 3; ++dz, ++it) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces of slab `it` have landed
                __syncthreads();                        // all pieces landed, staged rows visible, everybody has left ws[buf ^ 1]
                const float* xa = pa + dz * W2_PS;
                const float* xb_ = pb + dz * W2_PS;
                const float* wb = wbase + buf * WSL;
                // what is fetched underneath this iteration's MFMAs: the next slab of this tile, or - in the tile's last iteration - slab 0 of
                // the next tile; and (dz == 2) the next halo chunk of this tile, or chunk 0 of the next tile
                const bool tile_end = dz == 2 && last_chunk;
                if (tile_end) {
                    next_work = work + nwg;
                    has_next = next_work < nwork;
                }
                if (tile_end && has_next) nxt = describe(next_work);
                const bool more_w = !tile_end || has_next;
                const float* w_src = tile_end ? nxt.wsrc : cur.wsrc;
                const int w_it = tile_end ? 0 : it + 1;
                const bool more_raw = VEC && dz == 2 && (!last_chunk || has_next);
                const int raw_c0 = tile_end ? 0 : (chunk + 1) * CH;
                // 4 point steps (px) of 8 MFMAs: one ds_read_b128 per operand row delivers four k-steps (two row tiles x (ta, tb) rows + the
                // weight row = 5 reads per 8 MFMAs).  The reads of step px + 1 are requested before the MFMAs of step px are issued (two register
                // sets); the activation rows of the next dz iteration's first step before its barrier (the weight row has to wait for it).
                auto fetch_a = [&](const float* xa_, const float* xbb_, int px, int slot) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int off = 2 * m * W2_PS + px * WN_PL * W2_RS;
                        ra[slot][m] = *reinterpret_cast<const float4*>(xa_ + off);
                        rb[slot][m] = *reinterpret_cast<const float4*>(xbb_ + off);
                    }
                };
                if (dz == 0) fetch_a(xa, xb_, 0, 0);
                rw[0] = *reinterpret_cast<const float4*>(wb);
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    if (px + 1 < 4) {
                        rw[(px + 1) & 1] = *reinterpret_cast<const float4*>(wb + (px + 1) * NT * CH);
                        fetch_a(xa, xb_, px + 1, (px + 1) & 1);
                    } else if (dz < 2) {
                        fetch_a(xa + W2_PS, xb_ + W2_PS, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int sl = px & 1;
                    const float wv[4] = {rw[sl].x, rw[sl].y, rw[sl].z, rw[sl].w};
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const float a_ = s2 == 0 ? ra[sl][m].x : s2 == 1 ? ra[sl][m].y : s2 == 2 ? ra[sl][m].z : ra[sl][m].w;
                            const float b_ = s2 == 0 ? rb[sl][m].x : s2 == 1 ? rb[sl][m].y : s2 == 2 ? rb[sl][m].z : rb[sl][m].w;
                            acc[m][px] = __builtin_amdgcn_mfma_f32_32x32x2f32(fmaf(sa, b_, a_), wv[s2], acc[m][px], 0, 0, 0);
                        }
                        if (s2 == 0) {                  // behind the step's first MFMAs: their 128 pipe clocks cover the issue of these
                            __builtin_amdgcn_sched_barrier(0);
                            if (more_w) dma_w_piece(w_src, w_it, buf ^ 1, px);
                            if constexpr (VEC) {
                                if (more_raw) load_raw_part(tile_end ? nxt : cur, raw_c0, px);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                buf ^= 1;
            }
        }

        STAMP(3 + 2 * tile_no, __builtin_amdgcn_s_memtime());
        // ---- epilogue: x inverse transform in registers, y inverse transform across the four waves through LDS, one row tile (two z-planes) at
        // a time so that the exchange buffer stays inside the halo image (the next tile's first weight slab is landing in ws meanwhile)
        float* R = smem;                                    // [py][ox][r][lane]
        float* red = smem + RH;                             // [4 waves][2][NT]
        float* out_b = a.out + (long)cur.b * a.out_bs;
        const int z0 = cur.z0, y0 = cur.y0, x0 = cur.x0, co0 = cur.co0;
        // Fast path (whole tile inside the volume, all 32 couts real, channels-last 16-byte aligned output - every tile of the BASELINE layers):
        // lane = (channel quad q, row half, row group): the four waves' partial rows are fetched with ds_read_b128 (all issued before the first
        // use), the y inverse transform is done on float4s, and each voxel leaves as one 128-byte line written by 8 lanes x 16 bytes.
        // (BNR: the host launches this instantiation only when every tile qualifies; without bias and without the eval-mode store)
        const bool fast = BNR || (a.out_cs == 1 && (a.out_ps & 3) == 0 && (a.out_bs & 3) == 0 && (((uintptr_t)a.out) & 15) == 0 && co0 + NT <= a.Cout &&
                          z0 + 4 <= a.D && y0 + TY <= a.H && x0 + TX <= a.W &&
                          (a.bias == nullptr || (((uintptr_t)a.bias) & 15) == 0) &&
                          (a.coef == nullptr || ((((uintptr_t)a.coef) & 15) == 0 && (a.Cout & 3) == 0)));
        // (the lane id passes through an opaque asm here: everything the epilogue derives from it - output addresses, bias / coefficient
        //  loads - is then computed here and not hoisted above the main loop, where it would sit in registers the loop needs)
        int elane = lane;
        asm volatile("" : "+v"(elane));
        const int q = elane & 7, kh = (elane >> 3) & 1, g = elane >> 4;
        const int ei = elane & 31, ekk = elane >> 5;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 b4 = zero4, sc4 = zero4, sh4 = zero4, s4 = zero4, q4 = zero4;        // fast path: four channels per lane
        float bias1 = 0.f, fsc1 = 1.f, fsh1 = 0.f, ssum = 0.f, ssq = 0.f;             // general path: channel co0 + i
        const bool fuse = !BNR && a.coef != nullptr;
        const bool cok = co0 + ei < a.Cout;
        // data-gradient launch with the BatchNorm-backward reduction of the unit in front fused in (host: every tile takes the fast path)
        const bool bnr = BNR && fast;        // (own instantiation: its extra epilogue registers stay out of the plain kernel)
        float4 bm4 = zero4;                  // the channel means rounded to fp32 (pulpo_bn_bwd_finalize corrects for the rounding)
        const float* bn_b = bnr ? a.bn_y + (long)cur.b * a.bn_y_bs + co0 + 4 * q : nullptr;
        if (fast) {
            if (!BNR && a.bias != nullptr) b4 = *reinterpret_cast<const float4*>(a.bias + co0 + 4 * q);
            if (fuse) {
                sc4 = *reinterpret_cast<const float4*>(a.coef + 2 * a.Cout + co0 + 4 * q);
                sh4 = *reinterpret_cast<const float4*>(a.coef + 3 * a.Cout + co0 + 4 * q);
            }
            if (bnr) {
                sc4 = *reinterpret_cast<const float4*>(a.bn_coef + 2 * a.Cout + co0 + 4 * q);
                sh4 = *reinterpret_cast<const float4*>(a.bn_coef + 3 * a.Cout + co0 + 4 * q);
                bm4 = *reinterpret_cast<const float4*>(a.bn_coef + co0 + 4 * q);
            }
        } else if (cok) {
            if (a.bias != nullptr) bias1 = a.bias[co0 + ei];
            if (fuse) { fsc1 = a.coef[2 * a.Cout + co0 + ei]; fsh1 = a.coef[3 * a.Cout + co0 + ei]; }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            __syncthreads();                                // every wave has left xs (main loop) / the exchange buffer (previous row tile)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m0 = acc[m][0][r], m1 = acc[m][1][r], m2 = acc[m][2][r], m3 = acc[m][3][r];
                R[((py * 2 + 0) * 16 + r) * 64 + elane] = m0 + m1 + m2;
                R[((py * 2 + 1) * 16 + r) * 64 + elane] = m1 - m2 - m3;
            }
            float4 yv[2][2];                                // (bnr) the pre-norm activations of this lane's four voxels, requested once this row tile's accumulators are dead
            if (bnr) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int combo = h * 16 + wave * 4 + g;
                    const int ox = combo >> 4, r = combo & 15;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int gz = z0 + 2 * m + (row >> 4), gy = y0 + 2 * ((row >> 2) & 3), gx = x0 + 2 * (row & 3) + ox;
                    const long vox = (long)(gz * a.H + gy) * a.W + gx;
                    yv[h][0] = *reinterpret_cast<const float4*>(bn_b + vox * a.bn_y_ps);
                    yv[h][1] = *reinterpret_cast<const float4*>(bn_b + (vox + a.W) * a.bn_y_ps);
                }
            }
            __syncthreads();
            if (fast) {
                float4 tq[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int combo = h * 16 + wave * 4 + g;               // (ox, r) = (combo >> 4, combo & 15)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        tq[h][p] = *reinterpret_cast<const float4*>(R + ((p * 2 + (combo >> 4)) * 16 + (combo & 15)) * 64 + kh * 32 + 4 * q);
                }
                float* obase = out_b + co0 + 4 * q;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int combo = h * 16 + wave * 4 + g;
                    const int ox = combo >> 4, r = combo & 15;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int gz = z0 + 2 * m + (row >> 4), gy = y0 + 2 * ((row >> 2) & 3), gx = x0 + 2 * (row & 3) + ox;
                    const long vox = (long)(gz * a.H + gy) * a.W + gx;
                    const float4 t0 = tq[h][0], t1 = tq[h][1], t2 = tq[h][2], t3 = tq[h][3];
                    float4 v0 = make_float4(t0.x + t1.x + t2.x + b4.x, t0.y + t1.y + t2.y + b4.y, t0.z + t1.z + t2.z + b4.z, t0.w + t1.w + t2.w + b4.w);
                    float4 v1 = make_float4(t1.x - t2.x - t3.x + b4.x, t1.y - t2.y - t3.y + b4.y, t1.z - t2.z - t3.z + b4.z, t1.w - t2.w - t3.w + b4.w);
                    if (bnr) {
                        // dbn = dz * lrelu'(bn(y));  sums of dbn and of dbn * (y - fp32 mean): all fp32 (the difference of two floats carries
                        // a relative error of 2^-24 however close they are; what the ROUNDED mean leaves out is added back, in double, by
                        // the finalize kernel: sum dbn * xhat = rstd * (sum dbn * (y - m32) - (mean - m32) * sum dbn))
                        auto red1 = [&](float dzv, float yy, float sc, float sh, float m32, float& s_, float& q_) {
                            const float bn = yy * sc + sh;
                            const float d = bn > 0.f ? dzv : dzv * a.slope;
                            s_ += d;
                            q_ = fmaf(d, yy - m32, q_);
                        };
                        red1(v0.x, yv[h][0].x, sc4.x, sh4.x, bm4.x, s4.x, q4.x);
                        red1(v0.y, yv[h][0].y, sc4.y, sh4.y, bm4.y, s4.y, q4.y);
                        red1(v0.z, yv[h][0].z, sc4.z, sh4.z, bm4.z, s4.z, q4.z);
                        red1(v0.w, yv[h][0].w, sc4.w, sh4.w, bm4.w, s4.w, q4.w);
                        red1(v1.x, yv[h][1].x, sc4.x, sh4.x, bm4.x, s4.x, q4.x);
                        red1(v1.y, yv[h][1].y, sc4.y, sh4.y, bm4.y, s4.y, q4.y);
                        red1(v1.z, yv[h][1].z, sc4.z, sh4.z, bm4.z, s4.z, q4.z);
                        red1(v1.w, yv[h][1].w, sc4.w, sh4.w, bm4.w, s4.w, q4.w);
                    } else {
                        s4.x += v0.x + v1.x; s4.y += v0.y + v1.y; s4.z += v0.z + v1.z; s4.w += v0.w + v1.w;
                        q4.x += v0.x * v0.x + v1.x * v1.x; q4.y += v0.y * v0.y + v1.y * v1.y; q4.z += v0.z * v0.z + v1.z * v1.z; q4.w += v0.w * v0.w + v1.w * v1.w;
                    }
                    if (fuse) {
                        auto act = [&](float v, float sc, float sh) { const float tt = v * sc + sh; return tt > 0.f ? tt : tt * a.slope; };
                        v0 = make_float4(act(v0.x, sc4.x, sh4.x), act(v0.y, sc4.y, sh4.y), act(v0.z, sc4.z, sh4.z), act(v0.w, sc4.w, sh4.w));
                        v1 = make_float4(act(v1.x, sc4.x, sh4.x), act(v1.y, sc4.y, sh4.y), act(v1.z, sc4.z, sh4.z), act(v1.w, sc4.w, sh4.w));
                    }
                    *reinterpret_cast<float4*>(obase + vox * a.out_ps) = v0;
                    *reinterpret_cast<float4*>(obase + (vox + a.W) * a.out_ps) = v1;
                }
            } else {
                // general path: ragged tiles (volume edge), partial cout tiles, planar / strided outputs.  Wave w finishes x parity w & 1, rows
                // 8 (w >> 1) .. 8 (w >> 1) + 7 of the row tile; lane = (row half kk, channel i)
                const int fox = wave & 1;
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = 8 * (wave >> 1) + rr;
                    float tq[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) tq[p] = R[((p * 2 + fox) * 16 + r) * 64 + elane];
                    float v0 = tq[0] + tq[1] + tq[2] + bias1, v1 = tq[1] - tq[2] - tq[3] + bias1;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * ekk;
                    const int gz = z0 + 2 * m + (row >> 4), gy = y0 + 2 * ((row >> 2) & 3), gx = x0 + 2 * (row & 3) + fox;
                    if (cok && gz < a.D && gx < a.W) {
                        const long vox = (long)(gz * a.H + gy) * a.W + gx;
                        if (gy < a.H) {
                            ssum += v0; ssq += v0 * v0;
                            if (fuse) { const float tt = v0 * fsc1 + fsh1; v0 = tt > 0.f ? tt : tt * a.slope; }
                            out_b[vox * a.out_ps + (long)(co0 + ei) * a.out_cs] = v0;
                        }
                        if (gy + 1 < a.H) {
                            ssum += v1; ssq += v1 * v1;
                            if (fuse) { const float tt = v1 * fsc1 + fsh1; v1 = tt > 0.f ? tt : tt * a.slope; }
                            out_b[(vox + a.W) * a.out_ps + (long)(co0 + ei) * a.out_cs] = v1;
                        }
                    }
                }
            }
        }
        if (a.stats != nullptr) {
            // per-tile BatchNorm partial sums: reduce over the lanes that hold the same channel(s), then over the four waves
            if (fast) {
#pragma unroll
                for (int o = 8; o <= 32; o <<= 1) {
                    s4.x += __shfl_xor(s4.x, o, 64); s4.y += __shfl_xor(s4.y, o, 64); s4.z += __shfl_xor(s4.z, o, 64); s4.w += __shfl_xor(s4.w, o, 64);
                    q4.x += __shfl_xor(q4.x, o, 64); q4.y += __shfl_xor(q4.y, o, 64); q4.z += __shfl_xor(q4.z, o, 64); q4.w += __shfl_xor(q4.w, o, 64);
                }
                if (elane < 8) {
                    *reinterpret_cast<float4*>(red + (wave * 2 + 0) * NT + 4 * q) = s4;
                    *reinterpret_cast<float4*>(red + (wave * 2 + 1) * NT + 4 * q) = q4;
                }
            } else {
                ssum += __shfl_xor(ssum, 32, 64);
                ssq += __shfl_xor(ssq, 32, 64);
                if (elane < 32) {
                    red[(wave * 2 + 0) * NT + ei] = ssum;
                    red[(wave * 2 + 1) * NT + ei] = ssq;
                }
            }
            __syncthreads();
            if (tid < 2 * NT) {
                const int which = tid / NT, c = tid - which * NT;
                if (co0 + c < a.Cout) {
                    const float tot = red[(0 * 2 + which) * NT + c] + red[(1 * 2 + which) * NT + c] + red[(2 * 2 + which) * NT + c] +
                                      red[(3 * 2 + which) * NT + c];
                    a.stats[((long)cur.tile_lin * 2 + which) * a.Cout + co0 + c] = tot;
                }
            }
        }
        STAMP(4 + 2 * tile_no, __builtin_amdgcn_s_memtime());
        ++tile_no;
        if (!has_next) break;
        cur = nxt;
        work = next_work;
    }
}

// packing for the (y, x) Winograd kernel: wp[k/8][dz][py][px][n][k%8] = sum_dy sum_dx G[py][dy] G[px][dx] g[dz][dy][dx]
// element e = one (chunk, dz, n, k % 8): nine taps in, sixteen transformed points out
__device__ __forceinline__ void pack_wino2_one(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int NPad, int dgrad, long e) {
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const int kc = (int)(e % WN_CH);                 // (k fastest: consecutive threads write consecutive floats of the [n][k % 8] rows)
    long r = e / WN_CH;
    const int n = (int)(r % NPad); r /= NPad;
    const int dz = (int)(r % 3);
    const int chunk = (int)(r / 3);
    const int k = chunk * WN_CH + kc;
    float ux[3][4];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        float g[3] = {0.f, 0.f, 0.f};
        if (k < K && n < N) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int tap = (dz * 3 + dy) * 3 + dx;
                g[dx] = dgrad ? w[((long)k * Cin + n) * 27 + (26 - tap)] : w[((long)n * Cin + k) * 27 + tap];
            }
        }
        ux[dy][0] = g[0]; ux[dy][1] = 0.5f * (g[0] + g[1] + g[2]); ux[dy][2] = 0.5f * (g[0] - g[1] + g[2]); ux[dy][3] = g[2];
    }
    float* o = wp + (((long)(chunk * 3 + dz) * 16) * NPad + n) * WN_CH + kc;          // + (py * 4 + px) * NPad * WN_CH
#pragma unroll
    for (int px = 0; px < 4; ++px) {
        const float u0 = ux[0][px], u1 = ux[1][px], u2 = ux[2][px];
        o[(long)(0 * 4 + px) * WN_CH * NPad] = u0;
        o[(long)(1 * 4 + px) * WN_CH * NPad] = 0.5f * (u0 + u1 + u2);
        o[(long)(2 * 4 + px) * WN_CH * NPad] = 0.5f * (u0 - u1 + u2);
        o[(long)(3 * 4 + px) * WN_CH * NPad] = u2;
    }
}

__global__ void pack_weight_wino2_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int NPad, int dgrad, long total) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
        pack_wino2_one(w, wp, Cin, Cout, NPad, dgrad, e);
}

// every packed weight of a model in ONE launch (after an optimizer step that wrote the parameters): grid = (blocks per job, jobs)
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PulpoPackJob* __restrict__ jobs) {
    const PulpoPackJob j = jobs[blockIdx.y];
    const int K = j.dgrad ? j.Cout : j.Cin, N = j.dgrad ? j.Cin : j.Cout;
    const int NPad = npad(N);
    const long step = (long)gridDim.x * blockDim.x;
    if (j.kind == 2) {
        const long total = (long)((K + WN_CH - 1) / WN_CH) * 3 * WN_CH * NPad;
        for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += step) pack_wino2_one(j.w, j.wp, j.Cin, j.Cout, NPad, j.dgrad, e);
    } else if (j.kind == 4) {
        // the F(2x2x2,3x3x3) kernel's layout (wino3_pack.h)
        const long total = (long)((K + 7) / 8) * 8 * NPad;
        for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += step) pulpo_conv::pack_wino3_one(j.w, j.wp, j.Cin, j.Cout, NPad, j.dgrad, e);
    } else if (j.kind == 3) {
        // the bf16-operand kernels' layout (conv3d_bf16.hip pack_weight_bf16_kernel): bf16 wp[k / 32][tap][n][k % 32]
        uint16_t* wp16 = reinterpret_cast<uint16_t*>(j.wp);
        const long total = (long)((K + 31) / 32) * 27 * NPad * 32;
        for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += step) {
            const int kc = (int)(e % 32);
            long r = e / 32;
            const int n = (int)(r % NPad); r /= NPad;
            const int tap = (int)(r % 27);
            const int k = (int)(r / 27) * 32 + kc;
            float val = 0.f;
            if (k < K && n < N) val = j.dgrad ? j.w[((long)k * j.Cin + n) * 27 + (26 - tap)] : j.w[((long)n * j.Cin + k) * 27 + tap];
            wp16[e] = __builtin_bit_cast(uint16_t, (__bf16)val);
        }
    } else {
        // the direct kernel's layout (conv3d.hip pack_weight_kernel): wp[k / CH][tap][k % CH][n]
        const int CH = direct_ch(K);
        const long total = (long)((K + CH - 1) / CH) * 27 * CH * NPad;
        for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += step) {
            const int n = (int)(e % NPad);
            long r = e / NPad;
            const int kc = (int)(r % CH); r /= CH;
            const int tap = (int)(r % 27);
            const int k = (int)(r / 27) * CH + kc;
            float val = 0.f;
            if (k < K && n < N) val = j.dgrad ? j.w[((long)k * j.Cin + n) * 27 + (26 - tap)] : j.w[((long)n * j.Cin + k) * 27 + tap];
            j.wp[e] = val;
        }
    }
}

}  // namespace

// ================================================================================================ C ABI
// ---- (y, x) Winograd: same contract as pulpo_conv3d_k3_fwd / _fwd_bn_lrelu (coef nullable), own weight packing
PULPO_API size_t pulpo_conv3d_k3_packed_wino2_floats(int K, int N) { return (size_t)((K + WN_CH - 1) / WN_CH) * 3 * 16 * WN_CH * npad(N); }

PULPO_API int pulpo_conv3d_k3_pack_weight_wino2(const float* w, float* wp, int Cin, int Cout, int dgrad, void* stream) {
    PULPO_REQUIRE(w && wp && Cin > 0 && Cout > 0, "conv3d_k3_pack_weight_wino2: bad arguments");
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const long total = (long)pulpo_conv3d_k3_packed_wino2_floats(K, N) / 16;          // threads: one per (chunk, dz, k, n)
    const int nb = (int)std::min<long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(pack_weight_wino2_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wp, Cin, Cout, npad(N), dgrad, total);
    return pulpo::check_launch("pack_weight_wino2");
}

// jobs: DEVICE array; kind 0 = the layout of pulpo_conv3d_k3_pack_weight, 2 = of pulpo_conv3d_k3_pack_weight_wino2, 3 = of pulpo_conv3d_k3_pack_weight_bf16,
// 4 = of pulpo_conv3d_k3_pack_weight_wino3
PULPO_API int pulpo_conv3d_k3_pack_weights_multi(const PulpoPackJob* jobs, int njobs, void* stream) {
    PULPO_REQUIRE(jobs && njobs > 0, "conv3d_k3_pack_weights_multi: bad arguments");
    hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(160, njobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return pulpo::check_launch("pack_weights_multi");
}

template <bool VEC, bool BNR = false>
static int launch_wino2(const ConvArgs& a, int nblk, hipStream_t st) {
    constexpr size_t lds = (size_t)(W2_XS + 2 * 16 * WN_CH * 32) * sizeof(float);
    static_assert(lds >= (size_t)(4 * 2 * 2 * 16 * 64 + 4 * 2 * 32) * sizeof(float), "exchange buffer must fit");
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wino2_mfma<VEC, BNR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(conv3d wino2): %s", hipGetErrorString(e));
        attr_set = true;
    }
    // persistent workgroups: two per CU (LDS and registers admit exactly two)
    hipLaunchKernelGGL((conv3d_k3_wino2_mfma<VEC, BNR>), dim3(std::min(nblk, 512)), dim3(256), lds, st, a);
    return pulpo::check_launch("conv3d_k3_wino2_mfma");
}

static int wino2_pipe_enabled() {                      // PULPO_W2_PIPE=0 keeps the round-2 kernel for every operand (A/B switch)
    static int pipe = -1;
    if (pipe < 0) { const char* e = getenv("PULPO_W2_PIPE"); pipe = e ? atoi(e) : 1; }
    return pipe;
}

// 1 when pulpo_conv3d_k3_fwd_wino2 / _dgrad_wino2_bnred run the pipelined kernel (conv3d_k3_wino2p_mfma) for a channels-last, 16-byte
// aligned operand of K channels with voxel stride in_ps: K a multiple of 8 and the volume below 2^31 bytes.  (Names the kernel in traces.)
PULPO_API int pulpo_conv3d_k3_wino2_pipelined(int D, int H, int W, int K, int64_t in_ps) {
    ConvArgs a{};
    a.D = D; a.H = H; a.W = W; a.Cin = K; a.Cout = 32; a.in_ps = in_ps;      // (output channel counts up to 1024 qualify)
    return wino2_pipe_enabled() && wino2p_ok(a);
}

// Small volumes (the 20^3 pyramid level: 45 voxel tiles x 4 - 9 channel tiles for 512 resident workgroups) leave most compute units with one
// workgroup: the pipelined kernel then splits the reduction channels of every (voxel tile, channel tile) over `ks` work items that store
// partial slabs, and splitk_reduce_kernel adds the slabs in fixed order (deterministic) while it produces the BatchNorm partial sums.
// Measured on MI355X (scripts/conv_bench.py, 20^3, PULPO_W2P_KSPLIT = 1 / 2 / 3 / 4 / 8): 192 -> 192 150 / 139 / 136 / 157 / 170 us,
// 288 -> 192 218 / 187 / 180 / 205 / 228 us, 128 -> 192 110 / 111 / 116 / 127 / 149 us, 192 -> 288 (405 items) 149 / 174 / 184 / 193 / 215 us:
// three splits pay from 192 reduction channels up while the items fill at most about half of the slots, nothing else does.
// PULPO_W2P_KSPLIT=<n> forces the split (1 = off) for measurements.
static int wino2p_ksplit(int B, int D, int H, int W, int K, int N) {
    static int force = -1;
    if (force < 0) { const char* e = getenv("PULPO_W2P_KSPLIT"); force = e ? atoi(e) : 0; }
    if (!wino2_pipe_enabled() || K % 8 != 0) return 1;
    const long items = (long)B * pulpo::cdiv(D, 4) * pulpo::cdiv(H, TY) * pulpo::cdiv(W, TX) * pulpo::cdiv(N, 32);
    const int nchunk = K / 8;
    if (force > 0) return std::min(force, nchunk);
    if (items <= 96 && nchunk >= 8) return std::min(nchunk / 4, 6);          // the 10^3 level: 12 voxel tiles (see wino2_ragged_depth_ok)
    return (items <= 288 && nchunk >= 24) ? 3 : 1;
}

// The 10^3 level (192 -> 192 channels, 12 tiles of 4 x 8 x 8 with the last depth tile half empty): the direct kernel took 70 us per launch
// there (27 barrier-separated weight taps per 16-channel chunk on 1 - 2 chunks per workgroup).  The pipelined kernel masks every access
// by the volume's extent, so a ragged depth tile costs only its empty rows; statistics then always come from the split-K reduction,
// whose row count (pulpo_conv3d_k3_stat_tiles) does not depend on the tiling.
int pulpo_conv::wino2_ragged_depth_ok(int B, int D, int H, int W, int K, int N) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("PULPO_W2P_RAGGED_DEPTH"); on = e ? atoi(e) : 1; }      // (A/B switch)
    return on && D >= 8 && (long)D * H * W >= 1000 && K % 8 == 0 && wino2_pipe_enabled() && wino2p_ksplit(B, D, H, W, K, N) > 1;
}

// floats of scratch pulpo_conv3d_k3_fwd_wino2 needs for the shape (0: none, scratch may be NULL)
PULPO_API size_t pulpo_conv3d_k3_fwd_wino2_scratch_floats(int B, int D, int H, int W, int K, int N) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || K <= 0 || N <= 0) return 0;
    const int ks = wino2p_ksplit(B, D, H, W, K, N);
    return ks > 1 ? (size_t)ks * B * D * H * W * N : 0;
}

static int fwd_wino2_impl(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias, const float* coef,
                          float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats, float* scratch, const float* bn_y,
                          int64_t bn_y_bs, int64_t bn_y_ps, const float* bn_coef, int B, int D, int H, int W, int K, int N, void* stream);

PULPO_API int pulpo_conv3d_k3_fwd_wino2(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias,
                                        const float* coef, float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats,
                                        float* scratch, int B, int D, int H, int W, int K, int N, void* stream) {
    return fwd_wino2_impl(in, in_bs, in_ps, in_cs, wp, bias, coef, slope, out, out_bs, out_ps, out_cs, stats, scratch, nullptr, 0, 0, nullptr, B, D, H,
                          W, K, N, stream);
}

// 1 when pulpo_conv3d_k3_dgrad_wino2_bnred accepts the shape: every voxel tile whole (4 x 8 x 8) and every channel tile full (32).  (It
// never splits the reduction channels: the BatchNorm-backward sums come out of the convolution's own epilogue.)
PULPO_API int pulpo_conv3d_k3_dgrad_wino2_bnred_ok(int B, int D, int H, int W, int K, int N) {
    return B > 0 && K > 0 && N > 0 && conv_tz(D, H, W) == 4 && D % 4 == 0 && H % TY == 0 && W % TX == 0 && N % 32 == 0;
}

// The data-gradient convolution of ConvUnit u (dz = conv^T(dy_u), in = dy_u, N = that unit's input channels) with the first pass of the
// BatchNorm/LeakyReLU backward of ConvUnit u-1 - whose output z = lrelu(bn(y)) fed unit u - fused into its store: part[tile][2][N] receives
// per voxel tile sum(dbn) and sum(dbn * (y - fp32 mean)), dbn = dz * lrelu'(y * scale + shift), for pulpo_bn_bwd_finalize.  bn_y: the pre-norm
// tensor y of unit u-1 (channels-last, N channels, 16-byte aligned rows); bn_coef: its coefficient block (pulpo_bn_fwd_finalize).
// Replaces one pulpo_bn_lrelu_bwd_reduce pass (a read of dz and y from HBM) by a read of y underneath the convolution's epilogue.
PULPO_API int pulpo_conv3d_k3_dgrad_wino2_bnred(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, float* out,
                                                int64_t out_bs, int64_t out_ps, const float* bn_y, int64_t bn_y_bs, int64_t bn_y_ps,
                                                const float* bn_coef, float slope, float* part, int B, int D, int H, int W, int K, int N,
                                                void* stream) {
    PULPO_REQUIRE(bn_y && bn_coef && part, "conv3d_k3_dgrad_wino2_bnred: null pointer");
    PULPO_REQUIRE(pulpo_conv3d_k3_dgrad_wino2_bnred_ok(B, D, H, W, K, N), "conv3d_k3_dgrad_wino2_bnred: shape %dx%dx%d, %d -> %d channels has ragged tiles",
                  D, H, W, K, N);
    PULPO_REQUIRE(out_ps % 4 == 0 && out_bs % 4 == 0 && (((uintptr_t)out) & 15) == 0 && bn_y_ps % 4 == 0 && bn_y_bs % 4 == 0 &&
                      (((uintptr_t)bn_y) & 15) == 0 && (((uintptr_t)bn_coef) & 15) == 0,
                  "conv3d_k3_dgrad_wino2_bnred: output, pre-norm tensor and coefficients must be channels-last and 16-byte aligned");
    return fwd_wino2_impl(in, in_bs, in_ps, in_cs, wp, nullptr, nullptr, slope, out, out_bs, out_ps, 1, part, nullptr, bn_y, bn_y_bs, bn_y_ps, bn_coef,
                          B, D, H, W, K, N, stream);
}

static int fwd_wino2_impl(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias, const float* coef,
                          float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats, float* scratch, const float* bn_y,
                          int64_t bn_y_bs, int64_t bn_y_ps, const float* bn_coef, int B, int D, int H, int W, int K, int N, void* stream) {
    PULPO_REQUIRE(in && wp && out, "conv3d_k3_fwd_wino2: null pointer");
    PULPO_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && K > 0 && N > 0, "conv3d_k3_fwd_wino2: bad dims");
    PULPO_REQUIRE(conv_tz(D, H, W) == 4 || (bn_y == nullptr && wino2_ragged_depth_ok(B, D, H, W, K, N)),
                  "conv3d_k3_fwd_wino2: volume %dx%dx%d is not tiled 4x8x8 (see pulpo_conv3d_k3_algo)", D, H, W);
    PULPO_REQUIRE(!(coef && stats), "conv3d_k3_fwd_wino2: batch statistics are not available from the fused eval-mode epilogue");
    ConvArgs a{};
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps; a.in_cs = in_cs;
    a.wp = wp; a.bias = bias;
    a.out = out; a.out_bs = out_bs; a.out_ps = out_ps; a.out_cs = out_cs;
    a.stats = stats;
    a.coef = coef; a.slope = slope;
    a.bn_y = bn_y; a.bn_y_bs = bn_y_bs; a.bn_y_ps = bn_y_ps; a.bn_coef = bn_coef;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = K; a.Cout = N; a.NPad = npad(N);
    a.ntz = pulpo::cdiv(D, 4); a.nty = pulpo::cdiv(H, TY); a.ntx = pulpo::cdiv(W, TX);
    a.ncot = pulpo::cdiv(N, 32);
    a.ksplit = 1; a.part = nullptr;
    const long nblk_l = (long)B * a.ntz * a.nty * a.ntx * a.ncot;
    PULPO_REQUIRE(nblk_l < (1L << 31), "conv3d_k3_fwd_wino2: grid too large");
    const bool vec = (in_cs == 1) && (in_ps % 4 == 0) && (in_bs % 4 == 0) && (K % 4 == 0) && (((uintptr_t)in & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    // channels-last operands: the pipelined kernel (conv3d_wino2p.hip); PULPO_W2_PIPE=0 keeps the round-2 kernel
    const int pipe = wino2_pipe_enabled();
    PULPO_REQUIRE(conv_tz(D, H, W) == 4 || (vec && wino2p_ok(a)),
                  "conv3d_k3_fwd_wino2: a volume of depth %d needs a channels-last, 16-byte aligned operand (pipelined kernel, split-K)", D);
    if (bn_y != nullptr) {
        PULPO_REQUIRE(vec, "conv3d_k3_dgrad_wino2_bnred: the gradient operand must be channels-last, 16-byte aligned, with a multiple of 4 channels");
        if (pipe && wino2p_ok(a)) return launch_wino2p(a, (int)nblk_l, true, st);
        return launch_wino2<true, true>(a, (int)nblk_l, st);
    }
    if (vec && pipe && wino2p_ok(a)) {
        const int ks = wino2p_ksplit(B, D, H, W, K, N);
        if (ks == 1) return launch_wino2p(a, (int)nblk_l, false, st);
        PULPO_REQUIRE(scratch != nullptr, "conv3d_k3_fwd_wino2: scratch of pulpo_conv3d_k3_fwd_wino2_scratch_floats() floats required");
        PULPO_REQUIRE(nblk_l * ks < (1L << 31), "conv3d_k3_fwd_wino2: grid too large");
        a.ksplit = ks; a.part = scratch;
        a.stats = nullptr; a.coef = nullptr;            // both are the reduction's
        const int rc = launch_wino2p(a, (int)(nblk_l * ks), false, st);
        if (rc != 0) return rc;
        return pulpo_conv::launch_splitk_reduce(scratch, ks, out, (long)out_bs, (long)out_ps, (long)out_cs, B, (long)D * H * W, N,
                                                pulpo_conv3d_k3_stat_tiles(B, D, H, W), stats, coef, slope, st);
    }
    return vec ? launch_wino2<true>(a, (int)nblk_l, st) : launch_wino2<false>(a, (int)nblk_l, st);
}
