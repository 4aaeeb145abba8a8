// Winograd forms of the 3x3x3 convolution (forward and data gradient) for volumes tiled 4x8x8: F(2,3) along x (conv3d_k3_wino_mfma) and
// F(2x2,3x3) in (y, x) (conv3d_k3_wino2_mfma, the default), both direct over the z taps, all arithmetic fp32 on v_mfma_f32_32x32x2_f32.
// Same argument block, tile order, BatchNorm partial statistics and fused eval-mode epilogue as the direct kernel in conv3d.hip.
#include "conv_shared.h"
#include <stdlib.h>

namespace {

using namespace pulpo_conv;

// ------------------------------------------------------------------------------------------------ Winograd F(2,3) along x
// Large volumes: the 3 x-taps of the 3x3x3 stencil are evaluated with the minimal-filtering identity F(2,3) (two neighbouring
// outputs from four transformed inputs and four transformed weights instead of 2 x 3 products): 36 instead of 54 MFMA row
// products per output pair = 1.5x fewer matrix instructions, all arithmetic still fp32.
//   input  (staging)  : v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3        per (z, y, x-pair, channel)
//   weights (packing) : u0 = g0, u1 = (g0 + g1 + g2)/2, u2 = (g0 - g1 + g2)/2, u3 = g2  per (dz, dy, cin, cout)
//   output (registers): y_even = m0 + m1 + m2, y_odd = m1 - m2 - m3                     the four m live in the same lane
// Workgroup = 4 x 8 x 8 output voxels = 4 z-planes (one per wave) x 32 (y, x-pair) blocks; the MFMA rows are the blocks, one
// accumulator set per transformed point.  8-channel chunks: 960 transformed halo rows x 9 floats (34.5 KB) + the double-buffered
// (dz, dy) weight slabs [4 points][8][NT] (16 KB at NT = 64) => 3 workgroups per CU.
// LDS rows are ordered (hz, point, hy, x-pair): the 32 (y, x-pair) blocks an A fragment reads are 32 consecutive rows of 9 floats
// (odd stride => one bank per lane), a (dz, dy) tap moves the window by dz * 4 * WN_PL + dy * 4 rows
constexpr int WN_CH = 8, WN_CP = WN_CH + 1, WN_HZ = 6, WN_PL = HY * 4;
// floats per hz plane: 160 rows + 4 floats, so that blocks of neighbouring z-planes (the (y, x) kernel's row tiles span two) fall on
// disjoint LDS banks
constexpr int WN_PS = 4 * WN_PL * WN_CP + 4;

template <bool VEC>
__device__ __forceinline__ void stage_halo_wino(float* xs, const float* __restrict__ in, long in_ps, long in_cs, int c0, int Cin, int z0, int y0,
                                                int x0, int D, int H, int W, int tid) {
    if constexpr (VEC) {
        constexpr int Q = WN_CH / 4;
        constexpr int NITEM = WN_HZ * HY * 4 * Q;              // (hz, hy, x-pair, channel quad)
        constexpr int NIT = (NITEM + 255) / 256;
        float4 d[NIT][4];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            const int q = j % Q, xb = (j / Q) & 3, hrow = j / (4 * Q);
            const int hz = hrow / HY, hy = hrow - hz * HY;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy;
            const bool rowok = j < NITEM && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && c0 + 4 * q < Cin;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int gx = x0 - 1 + 2 * xb + t;
                d[u][t] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rowok && (unsigned)gx < (unsigned)W) d[u][t] = *reinterpret_cast<const float4*>(in + ((long)(gz * H + gy) * W + gx) * in_ps + c0 + 4 * q);
            }
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            if (j < NITEM) {
                const int q = j % Q, rb = j / Q;                // rb = (hz*HY + hy)*4 + xb
                const int hz = rb / (HY * 4), yx = rb - hz * (HY * 4);
                float* o = xs + hz * WN_PS + yx * WN_CP + 4 * q;      // row (hz, point 0, hy, xb); points are WN_PL rows apart
                const float4 d0 = d[u][0], d1 = d[u][1], d2 = d[u][2], d3 = d[u][3];
                o[0] = d0.x - d2.x; o[1] = d0.y - d2.y; o[2] = d0.z - d2.z; o[3] = d0.w - d2.w;
                o += WN_PL * WN_CP;
                o[0] = d1.x + d2.x; o[1] = d1.y + d2.y; o[2] = d1.z + d2.z; o[3] = d1.w + d2.w;
                o += WN_PL * WN_CP;
                o[0] = d2.x - d1.x; o[1] = d2.y - d1.y; o[2] = d2.z - d1.z; o[3] = d2.w - d1.w;
                o += WN_PL * WN_CP;
                o[0] = d1.x - d3.x; o[1] = d1.y - d3.y; o[2] = d1.z - d3.z; o[3] = d1.w - d3.w;
            }
        }
    } else {
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
            float* o = xs + hz * WN_PS + (hrow * 4 - hz * (HY * 4) + xb) * WN_CP + c;
            o[0] = d[0] - d[2];
            o[WN_PL * WN_CP] = d[1] + d[2];
            o[2 * WN_PL * WN_CP] = d[2] - d[1];
            o[3 * WN_PL * WN_CP] = d[1] - d[3];
        }
    }
}

// the VEC staging of stage_halo_wino split into its two halves, so that a kernel can issue the raw loads of the next chunk early
constexpr int WN_Q = WN_CH / 4, WN_NITEM = WN_HZ * HY * 4 * WN_Q, WN_NIT = (WN_NITEM + 255) / 256;

__device__ __forceinline__ void wino_load_raw(float4 (&d)[WN_NIT][4], const float* __restrict__ in, long in_ps, int c0, int Cin, int z0, int y0, int x0,
                                              int D, int H, int W, int tid) {
#pragma unroll
    for (int u = 0; u < WN_NIT; ++u) {
        const int j = tid + u * 256;
        const int q = j % WN_Q, xb = (j / WN_Q) & 3, hrow = j / (4 * WN_Q);
        const int hz = hrow / HY, hy = hrow - hz * HY;
        const int gz = z0 - 1 + hz, gy = y0 - 1 + hy;
        const bool rowok = j < WN_NITEM && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && c0 + 4 * q < Cin;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int gx = x0 - 1 + 2 * xb + t;
            d[u][t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rowok && (unsigned)gx < (unsigned)W) d[u][t] = *reinterpret_cast<const float4*>(in + ((long)(gz * H + gy) * W + gx) * in_ps + c0 + 4 * q);
        }
    }
}

__device__ __forceinline__ void wino_store_transformed(float* xs, const float4 (&d)[WN_NIT][4], int tid) {
#pragma unroll
    for (int u = 0; u < WN_NIT; ++u) {
        const int j = tid + u * 256;
        if (j < WN_NITEM) {
            const int q = j % WN_Q, rb = j / WN_Q;                // rb = (hz*HY + hy)*4 + xb
            const int hz = rb / (HY * 4), yx = rb - hz * (HY * 4);
            float* o = xs + hz * WN_PS + yx * WN_CP + 4 * q;
            const float4 d0 = d[u][0], d1 = d[u][1], d2 = d[u][2], d3 = d[u][3];
            o[0] = d0.x - d2.x; o[1] = d0.y - d2.y; o[2] = d0.z - d2.z; o[3] = d0.w - d2.w;
            o += WN_PL * WN_CP;
            o[0] = d1.x + d2.x; o[1] = d1.y + d2.y; o[2] = d1.z + d2.z; o[3] = d1.w + d2.w;
            o += WN_PL * WN_CP;
            o[0] = d2.x - d1.x; o[1] = d2.y - d1.y; o[2] = d2.z - d1.z; o[3] = d2.w - d1.w;
            o += WN_PL * WN_CP;
            o[0] = d1.x - d3.x; o[1] = d1.y - d3.y; o[2] = d1.z - d3.z; o[3] = d1.w - d3.w;
        }
    }
}

template <int NT, bool VEC>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wino_mfma(ConvArgs a) {
    constexpr int CH = WN_CH, CP = WN_CP;
    constexpr int NN = NT / 32;
    constexpr int XS = WN_HZ * WN_PS;
    constexpr int WSL = 4 * CH * NT;                 // floats of one (dz, dy) weight slab set: [point][k][NT]
    constexpr int WF4 = WSL / 4;
    constexpr int NW = WF4 / 256;                    // float4 per thread per slab set (1 at NT = 32, 2 at NT = 64)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ws = smem + XS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int cot = lid % a.ncot;
    const int tile_lin = lid / a.ncot;
    int t = tile_lin;
    const int tx_ = t % a.ntx; t /= a.ntx;
    const int ty_ = t % a.nty; t /= a.nty;
    const int tz_ = t % a.ntz;
    const int b = t / a.ntz;
    const int z0 = tz_ * 4, y0 = ty_ * TY, x0 = tx_ * TX;
    const int co0 = cot * NT;
    const int nchunk = (a.Cin + CH - 1) / CH;
    const int niter = nchunk * 9;
    const float* in_b = a.in + (long)b * a.in_bs;

    float4 wreg[NW];
    auto load_w = [&](int it) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int j = tid + u * 256;
            const int row = j / (NT / 4), c4 = j - row * (NT / 4);              // row = point * CH + k
            wreg[u] = *reinterpret_cast<const float4*>(a.wp + ((long)it * 4 * CH + row) * a.NPad + co0 + c4 * 4);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NW; ++u) *reinterpret_cast<float4*>(ws + buf * WSL + (tid + u * 256) * 4) = wreg[u];
    };

    const int i = lane & 31, kk = lane >> 5;
    // MFMA row i of wave w = block (z = w, y = i >> 2, x-pair = i & 3); its transformed rows start at rowbase (+ point)
    const int rowbase = wave * WN_PS + i * WN_CP;       // float offset of (hz = wave, point 0, block i); + point * WN_PL * CP

    f32x16 acc[4][NN];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][n][r] = 0.f;

    load_w(0);
    int buf = 0, it = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        __syncthreads();
        stage_halo_wino<VEC>(xs, in_b, a.in_ps, a.in_cs, chunk * CH, a.Cin, z0, y0, x0, a.D, a.H, a.W, tid);
        for (int zy = 0; zy < 9; ++zy, ++it) {
            store_w(buf);
            __syncthreads();
            if (it + 1 < niter) load_w(it + 1);
            const float* xa = xs + rowbase + (zy / 3) * WN_PS + (zy % 3) * 4 * CP + kk;
            const float* wb = ws + buf * WSL + kk * NT + i;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
#pragma unroll
                for (int s = 0; s < CH / 2; ++s) {
                    const float av = xa[p * WN_PL * CP + 2 * s];
#pragma unroll
                    for (int n = 0; n < NN; ++n) {
                        const float bv = wb[(p * CH + 2 * s) * NT + n * 32];
                        acc[p][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[p][n], 0, 0, 0);
                    }
                }
            }
            buf ^= 1;
        }
    }

    // ---- epilogue: inverse transform in registers, bias, [BatchNorm + LeakyReLU], store, per-tile BatchNorm partial statistics
    float* out_b = a.out + (long)b * a.out_bs;
    float ssum[NN], ssq[NN];
    const int gz = z0 + wave;
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        const int co = co0 + n * 32 + i;
        const bool cok = co < a.Cout;
        const float bv = (a.bias != nullptr && cok) ? a.bias[co] : 0.f;
        const bool fuse = a.coef != nullptr && cok;
        const float fsc = fuse ? a.coef[2 * a.Cout + co] : 1.f, fsh = fuse ? a.coef[3 * a.Cout + co] : 0.f;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int gy = y0 + (row >> 2), gx = x0 + 2 * (row & 3);
            const float m0 = acc[0][n][r], m1 = acc[1][n][r], m2 = acc[2][n][r], m3 = acc[3][n][r];
            float ve = m0 + m1 + m2 + bv, vo = m1 - m2 - m3 + bv;
            if (cok && gz < a.D && gy < a.H) {
                const long vox = (long)(gz * a.H + gy) * a.W + gx;
                if (gx < a.W) {
                    s += ve; q += ve * ve;
                    if (fuse) { const float tt = ve * fsc + fsh; ve = tt > 0.f ? tt : tt * a.slope; }
                    out_b[vox * a.out_ps + (long)co * a.out_cs] = ve;
                }
                if (gx + 1 < a.W) {
                    s += vo; q += vo * vo;
                    if (fuse) { const float tt = vo * fsc + fsh; vo = tt > 0.f ? tt : tt * a.slope; }
                    out_b[(vox + 1) * a.out_ps + (long)co * a.out_cs] = vo;
                }
            }
        }
        ssum[n] = s + __shfl_xor(s, 32, 64);
        ssq[n] = q + __shfl_xor(q, 32, 64);
    }
    if (a.stats != nullptr) {
        __syncthreads();
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

// Winograd weight packing: wp[k/8][dz*3+dy][point][k%8][n]  (forward: K = Cin, N = Cout, g_t = w[n][k][dz][dy][t];
// dgrad: K = Cout, N = Cin, g_t = w[k][n][2-dz][2-dy][2-t])
__global__ void pack_weight_wino_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int NPad, int dgrad, long total) {
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e % NPad);
        long r = e / NPad;
        const int kc = (int)(r % WN_CH); r /= WN_CH;
        const int pt = (int)(r % 4); r /= 4;
        const int zy = (int)(r % 9);
        const int chunk = (int)(r / 9);
        const int k = chunk * WN_CH + kc;
        float val = 0.f;
        if (k < K && n < N) {
            float g[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int tap = zy * 3 + t;
                g[t] = dgrad ? w[((long)k * Cin + n) * 27 + (26 - tap)] : w[((long)n * Cin + k) * 27 + tap];
            }
            val = pt == 0 ? g[0] : pt == 1 ? 0.5f * (g[0] + g[1] + g[2]) : pt == 2 ? 0.5f * (g[0] - g[1] + g[2]) : g[2];
        }
        wp[e] = val;
    }
}

// ------------------------------------------------------------------------------------------------ Winograd F(2x2,3x3) in (y, x)
// The y taps get the same treatment as the x taps: 16 transformed points per 2x2 output block, 3 (dz) x 16 matrix products per four
// outputs = 2.25x fewer than the direct kernel (1.5x fewer than F(2,3) along x alone).  The halo is staged x-transformed exactly as
// for the x-only kernel; WAVE py OWNS THE FOUR POINTS (py, px = 0..3) and forms the y combination of its A fragments as they are
// read (two ds_read + one fma per MFMA, wave-uniform tap pair), for all 64 blocks of the 4x8x8 tile (two MFMA row tiles of
// 2 z-planes x 4 x 4 blocks).  The x inverse transform is in-lane; the y inverse transform sums over the four waves through LDS
// once per tile, after which wave w finishes row tile w >> 1, x parity w & 1 (bias, BatchNorm partials, store).
template <bool VEC>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wino2_mfma(ConvArgs a) {
    constexpr int CH = WN_CH, CP = WN_CP, NT = 32;
    constexpr int XS = WN_HZ * WN_PS;
    constexpr int WSL = 16 * CH * NT;                // floats of one dz weight slab set: [py][px][k][NT]
    constexpr int RED = 4 * 2 * 2 * 16 * 64;         // floats of the cross-wave exchange buffer (reuses xs / ws)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* ws = smem + XS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int cot = lid % a.ncot;
    const int tile_lin = lid / a.ncot;
    int t = tile_lin;
    const int tx_ = t % a.ntx; t /= a.ntx;
    const int ty_ = t % a.nty; t /= a.nty;
    const int tz_ = t % a.ntz;
    const int b = t / a.ntz;
    const int z0 = tz_ * 4, y0 = ty_ * TY, x0 = tx_ * TX;
    const int co0 = cot * NT;
    const int nchunk = (a.Cin + CH - 1) / CH;
    const int niter = nchunk * 3;
    const float* in_b = a.in + (long)b * a.in_bs;

    // weight slab prefetch: 16 KB per dz = four float4 per thread (scalars, not an array: the array form ended up in scratch)
    float4 w0, w1, w2, w3;
    const float* wsrc = a.wp + (long)(tid >> 3) * a.NPad + co0 + (tid & 7) * 4;      // row = (py * 4 + px) * CH + k; 32 rows per 256 threads
    auto load_w = [&](int it) {
        const float* p = wsrc + (long)it * 16 * CH * a.NPad;
        w0 = *reinterpret_cast<const float4*>(p);
        w1 = *reinterpret_cast<const float4*>(p + 32L * a.NPad);
        w2 = *reinterpret_cast<const float4*>(p + 64L * a.NPad);
        w3 = *reinterpret_cast<const float4*>(p + 96L * a.NPad);
    };
    auto store_w = [&](int buf) {
        float* d = ws + buf * WSL + tid * 4;
        *reinterpret_cast<float4*>(d) = w0;
        *reinterpret_cast<float4*>(d + 1024) = w1;
        *reinterpret_cast<float4*>(d + 2048) = w2;
        *reinterpret_cast<float4*>(d + 3072) = w3;
    };

    const int i = lane & 31, kk = lane >> 5;
    const int py = __builtin_amdgcn_readfirstlane(wave);
    // y combination of this wave's point row: v = X[2 yb + ta] + sa * X[2 yb + tb]   (same table as the x transform)
    const int ta = py == 0 ? 0 : py == 2 ? 2 : 1;
    const int tb = py == 2 ? 1 : py == 3 ? 3 : 2;
    const float sa = py == 1 ? 1.f : -1.f;
    // MFMA row i of row tile m = block (z = 2 m + (i >> 4), yb = (i >> 2) & 3, xb = i & 3); LDS rows are (hz, px, hy, xb)
    const int lrow = ((i >> 2) & 3) * 8 + (i & 3);
    const float* pa = xs + (i >> 4) * WN_PS + (lrow + ta * 4) * CP + kk;
    const float* pb = xs + (i >> 4) * WN_PS + (lrow + tb * 4) * CP + kk;
    const float* wbase = ws + (py * 4 * CH + kk) * NT + i;

    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][p][r] = 0.f;

    load_w(0);
    int buf = 0, it = 0;
    // (VEC) the raw halo loads of chunk c+1 are issued in front of the last dz iteration of chunk c: their latency hides behind
    // its 32 MFMAs, and the registers are only live for that third of the loop
    float4 raw[VEC ? WN_NIT : 1][4];
    if constexpr (VEC) wino_load_raw(raw, in_b, a.in_ps, 0, a.Cin, z0, y0, x0, a.D, a.H, a.W, tid);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        __syncthreads();
        if constexpr (VEC) wino_store_transformed(xs, raw, tid);
        else stage_halo_wino<false>(xs, in_b, a.in_ps, a.in_cs, chunk * CH, a.Cin, z0, y0, x0, a.D, a.H, a.W, tid);
#pragma unroll
        for (int dz = 0; dz < 3; ++dz, ++it) {
            store_w(buf);
            __syncthreads();
            if (it + 1 < niter) load_w(it + 1);
            if constexpr (VEC) {
                if (dz == 2 && chunk + 1 < nchunk) wino_load_raw(raw, in_b, a.in_ps, (chunk + 1) * CH, a.Cin, z0, y0, x0, a.D, a.H, a.W, tid);
            }
            const float* xa = pa + dz * WN_PS;
            const float* xb_ = pb + dz * WN_PS;
            const float* wb = wbase + buf * WSL;
            // 16 steps (px, s) of two MFMAs (row tiles m = 0, 1).  The five LDS words of step n+2 are requested before the MFMAs of step
            // n are issued (three-slot register ring, pinned by sched_barrier), so no ds_read -> s_waitcnt -> v_mfma chain is exposed.
            float ra[3][2], rb[3][2], rw[3];
            auto fetch = [&](int st, int slot) {
                const int px = st >> 2, s2 = st & 3;
                rw[slot] = wb[(px * CH + 2 * s2) * NT];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int off = 2 * m * WN_PS + px * WN_PL * CP + 2 * s2;
                    ra[slot][m] = xa[off];
                    rb[slot][m] = xb_[off];
                }
            };
            fetch(0, 0);
            fetch(1, 1);
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                if (st + 2 < 16) fetch(st + 2, (st + 2) % 3);         // two steps (four MFMAs) of slack for the LDS round trip (three measured slower)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const float av = fmaf(sa, rb[st % 3][m], ra[st % 3][m]);
                    acc[m][st >> 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, rw[st % 3], acc[m][st >> 2], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }
    }

    // ---- x inverse transform in registers, y inverse transform across the four waves through LDS
    __syncthreads();                                   // every wave has left xs / ws
    float* R = smem;                                   // [py][m][ox][r][lane]
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = acc[m][0][r], m1 = acc[m][1][r], m2 = acc[m][2][r], m3 = acc[m][3][r];
            R[(((py * 2 + m) * 2 + 0) * 16 + r) * 64 + lane] = m0 + m1 + m2;
            R[(((py * 2 + m) * 2 + 1) * 16 + r) * 64 + lane] = m1 - m2 - m3;
        }
    }
    __syncthreads();
    const int fm = wave >> 1, fox = wave & 1;           // this wave finishes row tile fm, x parity fox
    float* out_b = a.out + (long)b * a.out_bs;
    const int co = co0 + i;
    const bool cok = co < a.Cout;
    const float bias = (a.bias != nullptr && cok) ? a.bias[co] : 0.f;
    const bool fuse = a.coef != nullptr && cok;
    const float fsc = fuse ? a.coef[2 * a.Cout + co] : 1.f, fsh = fuse ? a.coef[3 * a.Cout + co] : 0.f;
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float tq[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) tq[p] = R[(((p * 2 + fm) * 2 + fox) * 16 + r) * 64 + lane];
        float v0 = tq[0] + tq[1] + tq[2] + bias, v1 = tq[1] - tq[2] - tq[3] + bias;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
        const int gz = z0 + 2 * fm + (row >> 4), gy = y0 + 2 * ((row >> 2) & 3), gx = x0 + 2 * (row & 3) + fox;
        if (cok && gz < a.D && gx < a.W) {
            const long vox = (long)(gz * a.H + gy) * a.W + gx;
            if (gy < a.H) {
                ssum += v0; ssq += v0 * v0;
                if (fuse) { const float tt = v0 * fsc + fsh; v0 = tt > 0.f ? tt : tt * a.slope; }
                out_b[vox * a.out_ps + (long)co * a.out_cs] = v0;
            }
            if (gy + 1 < a.H) {
                ssum += v1; ssq += v1 * v1;
                if (fuse) { const float tt = v1 * fsc + fsh; v1 = tt > 0.f ? tt : tt * a.slope; }
                out_b[(vox + a.W) * a.out_ps + (long)co * a.out_cs] = v1;
            }
        }
    }
    ssum += __shfl_xor(ssum, 32, 64);
    ssq += __shfl_xor(ssq, 32, 64);
    if (a.stats != nullptr) {
        float* red = smem + RED;                        // [4 waves][2][NT], behind the exchange buffer
        if (lane < 32) {
            red[(wave * 2 + 0) * NT + i] = ssum;
            red[(wave * 2 + 1) * NT + i] = ssq;
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

// packing for the (y, x) Winograd kernel: wp[k/8][dz][py][px][k%8][n] = sum_dy sum_dx G[py][dy] G[px][dx] g[dz][dy][dx]
__global__ void pack_weight_wino2_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int NPad, int dgrad, long total) {
    // one thread per (chunk, dz, k, n): nine taps in, sixteen transformed points out
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e % NPad);
        long r = e / NPad;
        const int kc = (int)(r % WN_CH); r /= WN_CH;
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
        float* o = wp + (((long)(chunk * 3 + dz) * 16) * WN_CH + kc) * NPad + n;          // + (py * 4 + px) * WN_CH * NPad
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            const float u0 = ux[0][px], u1 = ux[1][px], u2 = ux[2][px];
            o[(long)(0 * 4 + px) * WN_CH * NPad] = u0;
            o[(long)(1 * 4 + px) * WN_CH * NPad] = 0.5f * (u0 + u1 + u2);
            o[(long)(2 * 4 + px) * WN_CH * NPad] = 0.5f * (u0 - u1 + u2);
            o[(long)(3 * 4 + px) * WN_CH * NPad] = u2;
        }
    }
}

}  // namespace

// ================================================================================================ C ABI
PULPO_API size_t pulpo_conv3d_k3_packed_wino_floats(int K, int N) { return (size_t)((K + WN_CH - 1) / WN_CH) * 9 * 4 * WN_CH * npad(N); }

PULPO_API int pulpo_conv3d_k3_pack_weight_wino(const float* w, float* wp, int Cin, int Cout, int dgrad, void* stream) {
    PULPO_REQUIRE(w && wp && Cin > 0 && Cout > 0, "conv3d_k3_pack_weight_wino: bad arguments");
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const long total = (long)pulpo_conv3d_k3_packed_wino_floats(K, N);
    const int nb = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weight_wino_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wp, Cin, Cout, npad(N), dgrad, total);
    return pulpo::check_launch("pack_weight_wino");
}

template <int NT, bool VEC>
static int launch_wino(const ConvArgs& a, int nblk, hipStream_t st) {
    constexpr size_t lds = (size_t)(WN_HZ * WN_PS + 2 * 4 * WN_CH * NT) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wino_mfma<NT, VEC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(conv3d wino): %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3d_k3_wino_mfma<NT, VEC>), dim3(nblk), dim3(256), lds, st, a);
    return pulpo::check_launch("conv3d_k3_wino_mfma");
}

// same contract as pulpo_conv3d_k3_fwd / _fwd_bn_lrelu (coef nullable) with weights from pulpo_conv3d_k3_pack_weight_wino;
// only for shapes where pulpo_conv3d_k3_algo() returns 1
PULPO_API int pulpo_conv3d_k3_fwd_wino(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias,
                                       const float* coef, float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats,
                                       int B, int D, int H, int W, int K, int N, void* stream) {
    PULPO_REQUIRE(in && wp && out, "conv3d_k3_fwd_wino: null pointer");
    PULPO_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && K > 0 && N > 0, "conv3d_k3_fwd_wino: bad dims");
    PULPO_REQUIRE(conv_tz(D, H, W) == 4, "conv3d_k3_fwd_wino: volume %dx%dx%d is not tiled 4x8x8 (see pulpo_conv3d_k3_algo)", D, H, W);
    PULPO_REQUIRE(!(coef && stats), "conv3d_k3_fwd_wino: batch statistics are not available from the fused eval-mode epilogue");
    ConvArgs a;
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps; a.in_cs = in_cs;
    a.wp = wp; a.bias = bias;
    a.out = out; a.out_bs = out_bs; a.out_ps = out_ps; a.out_cs = out_cs;
    a.stats = stats;
    a.coef = coef; a.slope = slope;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = K; a.Cout = N; a.NPad = npad(N);
    a.ntz = pulpo::cdiv(D, 4); a.nty = pulpo::cdiv(H, TY); a.ntx = pulpo::cdiv(W, TX);
    const int NT = 32;                 // 64-wide tiles need 128 accumulator registers and measured slower (2 instead of 3 waves per SIMD)
    a.ncot = pulpo::cdiv(N, NT);
    a.ksplit = 1; a.part = nullptr;
    const long nblk_l = (long)B * a.ntz * a.nty * a.ntx * a.ncot;
    PULPO_REQUIRE(nblk_l < (1L << 31), "conv3d_k3_fwd_wino: grid too large");
    const bool vec = (in_cs == 1) && (in_ps % 4 == 0) && (in_bs % 4 == 0) && (K % 4 == 0) && (((uintptr_t)in & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    return vec ? launch_wino<32, true>(a, (int)nblk_l, st) : launch_wino<32, false>(a, (int)nblk_l, st);
}

// ---- (y, x) Winograd variant: same contract as the x-only entry points
PULPO_API size_t pulpo_conv3d_k3_packed_wino2_floats(int K, int N) { return (size_t)((K + WN_CH - 1) / WN_CH) * 3 * 16 * WN_CH * npad(N); }

PULPO_API int pulpo_conv3d_k3_pack_weight_wino2(const float* w, float* wp, int Cin, int Cout, int dgrad, void* stream) {
    PULPO_REQUIRE(w && wp && Cin > 0 && Cout > 0, "conv3d_k3_pack_weight_wino2: bad arguments");
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const long total = (long)pulpo_conv3d_k3_packed_wino2_floats(K, N) / 16;          // threads: one per (chunk, dz, k, n)
    const int nb = (int)std::min<long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(pack_weight_wino2_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wp, Cin, Cout, npad(N), dgrad, total);
    return pulpo::check_launch("pack_weight_wino2");
}

template <bool VEC>
static int launch_wino2(const ConvArgs& a, int nblk, hipStream_t st) {
    constexpr size_t lds = (size_t)(WN_HZ * WN_PS + 2 * 16 * WN_CH * 32) * sizeof(float);
    static_assert(lds >= (size_t)(4 * 2 * 2 * 16 * 64 + 4 * 2 * 32) * sizeof(float), "exchange buffer must fit");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wino2_mfma<VEC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(conv3d wino2): %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3d_k3_wino2_mfma<VEC>), dim3(nblk), dim3(256), lds, st, a);
    return pulpo::check_launch("conv3d_k3_wino2_mfma");
}

PULPO_API int pulpo_conv3d_k3_fwd_wino2(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias,
                                        const float* coef, float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats,
                                        int B, int D, int H, int W, int K, int N, void* stream) {
    PULPO_REQUIRE(in && wp && out, "conv3d_k3_fwd_wino2: null pointer");
    PULPO_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && K > 0 && N > 0, "conv3d_k3_fwd_wino2: bad dims");
    PULPO_REQUIRE(conv_tz(D, H, W) == 4, "conv3d_k3_fwd_wino2: volume %dx%dx%d is not tiled 4x8x8 (see pulpo_conv3d_k3_algo)", D, H, W);
    PULPO_REQUIRE(!(coef && stats), "conv3d_k3_fwd_wino2: batch statistics are not available from the fused eval-mode epilogue");
    ConvArgs a;
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps; a.in_cs = in_cs;
    a.wp = wp; a.bias = bias;
    a.out = out; a.out_bs = out_bs; a.out_ps = out_ps; a.out_cs = out_cs;
    a.stats = stats;
    a.coef = coef; a.slope = slope;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = K; a.Cout = N; a.NPad = npad(N);
    a.ntz = pulpo::cdiv(D, 4); a.nty = pulpo::cdiv(H, TY); a.ntx = pulpo::cdiv(W, TX);
    a.ncot = pulpo::cdiv(N, 32);
    a.ksplit = 1; a.part = nullptr;
    const long nblk_l = (long)B * a.ntz * a.nty * a.ntx * a.ncot;
    PULPO_REQUIRE(nblk_l < (1L << 31), "conv3d_k3_fwd_wino2: grid too large");
    const bool vec = (in_cs == 1) && (in_ps % 4 == 0) && (in_bs % 4 == 0) && (K % 4 == 0) && (((uintptr_t)in & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    return vec ? launch_wino2<true>(a, (int)nblk_l, st) : launch_wino2<false>(a, (int)nblk_l, st);
}
