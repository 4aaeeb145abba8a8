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
// gradient (conv3d_k3_wino2_mfma; conv3d_k3_wino_mfma is the x-only predecessor) and F(2,3) along x for the weight gradient
// (conv3d_k3_wgrad_wino) - fewer matrix instructions, all arithmetic still fp32.
#include "conv_shared.h"
#include <stdlib.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

using pulpo_conv::TY; using pulpo_conv::TX; using pulpo_conv::HY; using pulpo_conv::HX;
using pulpo_conv::conv_tz; using pulpo_conv::npad;
constexpr int TZ = 2, MV = TZ * TY * TX;          // base output voxel tile (wgrad; forward uses conv_tz())
constexpr int HZ = TZ + 2, HV = HZ * HY * HX;     // halo tile

struct ConvArgs {
    const float* in;
    long in_bs, in_ps, in_cs;     // batch / pixel / channel strides in floats
    const float* wp;              // packed [nchunk][27][CH][NPad]
    const float* bias;            // nullable
    float* out;
    long out_bs, out_ps, out_cs;
    float* stats;                 // nullable: [voxel tile][2][Cout]  (sum, sum of squares of conv+bias)
    int B, D, H, W, Cin, Cout, NPad;
    int ntz, nty, ntx, ncot;
    int ksplit;                   // > 1: the Cin chunks are split over ksplit workgroups, each storing a partial slab into `part`
    float* part;                  // [ksplit][B*V][Cout] dense partial outputs (reduced in fixed order by splitk_reduce_kernel)
    const float* coef;            // nullable: eval-mode BatchNorm coefficients (scale at [2C], shift at [3C]) + LeakyReLU fused into the store
    float slope;
};

// 64 bytes of zeros: source address of out-of-volume / out-of-channel lanes of an LDS-DMA piece
__device__ float4 g_zero_page[4];

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS starting at the WAVE-UNIFORM address lds_piece
__device__ __forceinline__ void dma16(const float* src, float* lds_piece) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}

__device__ __forceinline__ int tap_halo_offset(int tap) {
    return ((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3;
}

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
    constexpr int WF4 = CH * NT / 4;               // float4 per weight slab
    constexpr int NW = (WF4 + 255) / 256;          // float4 per thread per slab
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
    const int it0 = chunk0 * 27, niter = chunk1 * 27;
    const float* in_b = a.in + (long)b * a.in_bs;

    // weight slab prefetch registers
    float4 wreg[NW];
    auto load_w = [&](int it) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int j = tid + u * 256;
            if (WF4 % 256 == 0 || j < WF4) {
                const int row = j / (NT / 4), c4 = j - row * (NT / 4);
                wreg[u] = *reinterpret_cast<const float4*>(a.wp + ((long)it * CH + row) * a.NPad + co0 + c4 * 4);
            }
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int j = tid + u * 256;
            if (WF4 % 256 == 0 || j < WF4) *reinterpret_cast<float4*>(ws + buf * CH * NT + j * 4) = wreg[u];
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
        for (int tap = 0; tap < 27; ++tap, ++it) {
            store_w(buf);
            __syncthreads();
            if (it + 1 < niter) load_w(it + 1);
            const float* xa[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) xa[m] = xs + (hb[m] + tap_halo_offset(tap)) * CP + kk;
            const float* wb = ws + buf * CH * NT + kk * NT + i;
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

// split-K finish: out = sum_s part[s] in fixed order (deterministic), plus the per-row (sum, sum of squares) BatchNorm partials.
// Row r of stats covers voxels [r*V/nrow, (r+1)*V/nrow): any partition is fine for the double-precision finalize.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ksplit, float* __restrict__ out, long obs, long ops,
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
                const float t = v * fsc + fsh;
                v = t > 0.f ? t : t * slope;
            }
            out[b * obs + vox * ops + (long)c * ocs] = v;
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

// ------------------------------------------------------------------------------------------------ weight gradient
struct WgradArgs {
    const float* in;
    long in_bs, in_ps, in_cs;
    const float* dy;
    long dy_bs, dy_ps, dy_cs;
    float* dwp;                   // zero-initialised scratch [27][Cin][NPad], accumulated with float atomics
    int B, D, H, W, Cin, Cout, NPad;
    int ntz, nty, ntx, ncit, ncot, nsplit;
};

constexpr int WG_CH = 32, WG_NT = 32, WG_CP = WG_CH + 1;

// NTW = row tiles (32 (tap,ci) pairs each) per wave: 7 for a full 32-channel ci tile (27 tiles over 4 waves), fewer for
// narrow inputs.  The MFMAs of the hot loop are unconditional (rows beyond the matrix compute garbage that is never flushed), so the loop
// body is one basic block and the compiler can run the LDS reads ahead of the matrix pipe.
template <bool VEC, int NTW>
__global__ __launch_bounds__(256, VEC ? 1 : 2) void conv3d_k3_wgrad_mfma(WgradArgs a) {
    constexpr int XS = (HV * WG_CP + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int npair = a.ncit * a.ncot;
    const int pair = lid % npair, split = lid / npair;
    const int cit = pair / a.ncot, cot = pair - cit * a.ncot;
    const int ci0 = cit * WG_CH, co0 = cot * WG_NT;
    const int Cc = min(WG_CH, a.Cin - ci0);
    const int rows = 27 * Cc;
    const int nrt = (rows + 31) >> 5;                 // row tiles of 32 (tap,ci) pairs; host guarantees nrt <= 4 * NTW
    const int i = lane & 31, kk = lane >> 5;
    constexpr int STR = VEC ? 32 : WG_CP;             // voxel stride of the halo image (DMA image is unpadded)

    int rowoff[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int r = 32 * (wave + 4 * u) + i;
        const int tap = r < rows ? r / Cc : 0, ci = r < rows ? r - tap * Cc : 0;
        rowoff[u] = tap_halo_offset(tap) * STR + ci;
    }
    f32x16 acc[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    const int ntile = a.B * a.ntz * a.nty * a.ntx;
    const int per = (ntile + a.nsplit - 1) / a.nsplit;
    const int t_begin = split * per, t_end = min(ntile, t_begin + per);

    auto decode = [&](int tl, int& b, int& z0, int& y0, int& x0) {
        int t = tl;
        const int tx_ = t % a.ntx; t /= a.ntx;
        const int ty_ = t % a.nty; t /= a.nty;
        const int tz_ = t % a.ntz;
        b = t / a.ntz;
        z0 = tz_ * TZ; y0 = ty_ * TY; x0 = tx_ * TX;
    };
    // One "body" = 4 voxel-pair steps.  Rows beyond the matrix (last partial row tile, or a whole spare tile) read valid
    // LDS words of tap 0 and accumulate garbage into accumulator rows that the flush never writes: MFMA rows are independent,
    // so no masking is needed.
    auto load_body = [&](float (&A)[4][NTW], float (&Bv)[4], const float* xs_c, const float* dys_c, int sb) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int vox = 2 * (sb * 4 + q4) + kk;
            const int hbk = (((vox >> 6) * HY + ((vox >> 3) & 7)) * HX + (vox & 7)) * STR;
            Bv[q4] = dys_c[vox * WG_NT + i];
#pragma unroll
            for (int u = 0; u < NTW; ++u) A[q4][u] = xs_c[rowoff[u] + hbk];
        }
    };
    auto mma_body = [&](const float (&A)[4][NTW], const float (&Bv)[4]) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int u = 0; u < NTW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[q4][u], Bv[q4], acc[u], 0, 0, 0);
    };
    float A0[4][NTW], A1[4][NTW], B0[4], B1[4];       // register double buffer of the fragments

    if constexpr (VEC) {
        // LDS-DMA pipeline.  The A-operand lanes index consecutive (tap, ci) rows, so the halo image needs no padding
        // ([halo voxel][32 ci], 128 B per voxel) and is filled by global_load_lds: no staging registers.  Two image sets
        // (X 50 KiB + dY 16 KiB each) ping-pong: the 17 DMA pieces of tile t+1 are issued one per 4 voxel-pair steps inside
        // the MFMA loop of tile t (their address arithmetic hides behind the matrix pipe) and are drained by the
        // s_waitcnt vmcnt(0) + barrier at the tile boundary.
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const float* zero = reinterpret_cast<const float*>(g_zero_page);
        constexpr int XIMG = 13 * 4 * 256;            // halo image padded to 52 DMA pieces (50 used): every piece is unconditional
        constexpr int SET = XIMG + MV * WG_NT;        // floats per image set
        int nb = 0, nz0 = 0, ny0 = 0, nx0 = 0;        // next tile
        bool more = false;
        // piece pc in [0, 17): 0..12 = halo (piece 12 of waves 2,3 is padding), 13..16 = dY.  Branch-free: lanes with nothing
        // to fetch (out of volume / channel range / no next tile / padding) source the zero page.
        auto issue_piece = [&](int pc, float* xd) {
            if (pc < 13) {
                const int j = tid + pc * 256;
                const int hv = j >> 3, q = j & 7;
                const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
                const int gz = nz0 - 1 + hz, gy = ny0 - 1 + hy, gx = nx0 - 1 + hx;
                const bool ok = more && hv < HV && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W &&
                                ci0 + 4 * q < a.Cin;
                const float* src = ok ? a.in + (long)nb * a.in_bs + ((long)(gz * a.H + gy) * a.W + gx) * a.in_ps + ci0 + 4 * q : zero;
                dma16(src, xd + (wave_u + 4 * pc) * 256);
            } else {
                const int u = pc - 13;
                const int j = tid + u * 256;
                const int vv = j >> 3, q = j & 7;
                const int gz = nz0 + (vv >> 6), gy = ny0 + ((vv >> 3) & 7), gx = nx0 + (vv & 7);
                const bool ok = more && gz < a.D && gy < a.H && gx < a.W && co0 + 4 * q < a.Cout;
                const float* src = ok ? a.dy + (long)nb * a.dy_bs + ((long)(gz * a.H + gy) * a.W + gx) * a.dy_ps + co0 + 4 * q : zero;
                dma16(src, xd + XIMG + (wave_u + 4 * u) * 256);
            }
        };
        if (t_begin < t_end) {
            more = true;
            decode(t_begin, nb, nz0, ny0, nx0);
#pragma unroll
            for (int pc = 0; pc < 17; ++pc) issue_piece(pc, smem);
        }
        int cur = 0;
        for (int tl = t_begin; tl < t_end; ++tl, cur ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of tile tl have landed
            __syncthreads();      // (a) everybody's pieces have landed  (b) everybody left the other image set
            more = tl + 1 < t_end;
            if (more) decode(tl + 1, nb, nz0, ny0, nx0);
            const float* xs_c = smem + cur * SET;
            const float* dys_c = xs_c + XIMG;
            float* xnext = smem + (cur ^ 1) * SET;
            load_body(A0, B0, xs_c, dys_c, 0);
#pragma unroll
            for (int sb = 0; sb < 16; sb += 2) {            // fully unrolled: 16 bodies of 4 steps, fragments one body ahead
                load_body(A1, B1, xs_c, dys_c, sb + 1);
                __builtin_amdgcn_sched_barrier(0);
                // the next tile's 17 DMA pieces go out during the first nine bodies, so they have half a tile of MFMAs to land
                if (sb < 8) { issue_piece(2 * sb, xnext); issue_piece(2 * sb + 1, xnext); }
                if (sb == 8) issue_piece(16, xnext);
                mma_body(A0, B0);
                load_body(A0, B0, xs_c, dys_c, (sb + 2) & 15);   // (wraps to body 0 on the last trip: harmless re-read)
                __builtin_amdgcn_sched_barrier(0);
                if (sb < 8) { issue_piece(2 * sb + 2, xnext); issue_piece(2 * sb + 3, xnext); }
                mma_body(A1, B1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the (zero-sourced) pieces issued during the last tile
    } else {
        // scalar-staging path (planar / narrow inputs: the 2- and 3-channel first layers).  Only the Cc real channels of the
        // halo are staged (rows of other channels are never flushed); dY goes through float4 when it is channels-last.
        float* xs = smem;
        float* dys = smem + XS;
        const bool dyvec = (a.dy_cs == 1) && (a.dy_ps % 4 == 0) && (a.dy_bs % 4 == 0) && (a.Cout % 4 == 0) && (((uintptr_t)a.dy & 15) == 0);
        for (int tl = t_begin; tl < t_end; ++tl) {
            int b, z0, y0, x0;
            decode(tl, b, z0, y0, x0);
            __syncthreads();
            const float* in_b = a.in + (long)b * a.in_bs;
            for (int j = tid; j < HV * Cc; j += 256) {
                const int c = j / HV, hv = j - c * HV;              // voxel fastest: coalesced for planar inputs
                const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
                const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                float v = 0.f;
                if ((unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                    v = in_b[((long)(gz * a.H + gy) * a.W + gx) * a.in_ps + (long)(ci0 + c) * a.in_cs];
                xs[hv * WG_CP + c] = v;
            }
            const float* dyb = a.dy + (long)b * a.dy_bs;
            if (dyvec) {
                float4 val[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = tid + u * 256;
                    const int vv = j >> 3, q = j & 7;
                    const int gz = z0 + (vv >> 6), gy = y0 + ((vv >> 3) & 7), gx = x0 + (vv & 7);
                    val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gz < a.D && gy < a.H && gx < a.W && co0 + 4 * q < a.Cout)
                        val[u] = *reinterpret_cast<const float4*>(dyb + ((long)(gz * a.H + gy) * a.W + gx) * a.dy_ps + co0 + 4 * q);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = tid + u * 256;
                    *reinterpret_cast<float4*>(dys + (j >> 3) * WG_NT + 4 * (j & 7)) = val[u];
                }
            } else {
                for (int j = tid; j < MV * WG_NT; j += 256) {
                    const int vv = j >> 5, c = j & 31;
                    const int gz = z0 + (vv >> 6), gy = y0 + ((vv >> 3) & 7), gx = x0 + (vv & 7);
                    float val = 0.f;
                    if (gz < a.D && gy < a.H && gx < a.W && co0 + c < a.Cout)
                        val = dyb[((long)(gz * a.H + gy) * a.W + gx) * a.dy_ps + (long)(co0 + c) * a.dy_cs];
                    dys[vv * WG_NT + c] = val;
                }
            }
            __syncthreads();
            load_body(A0, B0, xs, dys, 0);
#pragma unroll 1
            for (int sb = 0; sb < 16; sb += 2) {
                load_body(A1, B1, xs, dys, sb + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma_body(A0, B0);
                load_body(A0, B0, xs, dys, (sb + 2) & 15);
                __builtin_amdgcn_sched_barrier(0);
                mma_body(A1, B1);
            }
        }
    }

    // flush: one 128-byte run of couts per (tap, ci) row -> float atomics at full rate
    const int co = co0 + i;
    if (co < a.Cout) {
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            if (wave + 4 * u < nrt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rg = 32 * (wave + 4 * u) + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (rg < rows) {
                        const int tap = rg / Cc, ci = rg - tap * Cc;
                        atomicAdd(a.dwp + ((long)tap * a.Cin + ci0 + ci) * a.NPad + co, acc[u][r]);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient, Winograd along x
// The transpose of the forward identity: with V = B^T d (four transformed inputs per x-pair) and E = A dy (four combinations of the
// pair's two output gradients), M_p[(dz,dy,ci)][co] = sum over x-pairs V_p * E_p and dw[.., t] = G^T M - 36 instead of 54 matrix
// products per x-pair.  Wave p of the workgroup owns transformed point p (all nine (dz, dy) row tiles of its 32-channel slice), so
// both operand transforms are wave-uniform two-term combinations formed from the raw LDS-DMA images as the fragments are read
// (A: two ds_read + one fma, B: two ds_read + two fma per nine MFMAs), and every wave adds its share of G^T M at the flush.
// Same persistent one-workgroup-per-CU LDS-DMA pipeline as conv3d_k3_wgrad_mfma<true, NTW>.
template <int NTW>
__global__ __launch_bounds__(256, 1) void conv3d_k3_wgrad_wino(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int npair = a.ncit * a.ncot;
    const int pair = lid % npair, split = lid / npair;
    const int cit = pair / a.ncot, cot = pair - cit * a.ncot;
    const int ci0 = cit * WG_CH, co0 = cot * WG_NT;
    const int Cc = min(WG_CH, a.Cin - ci0);
    const int rows = 9 * Cc;
    const int nrt = (rows + 31) >> 5;                 // host guarantees nrt <= NTW
    const int i = lane & 31, kk = lane >> 5;
    const int pt = __builtin_amdgcn_readfirstlane(wave);
    // A = X[x + ta] + sa * X[x + tb];  B = c0 * dY[x] + c1 * dY[x + 1]
    const int ta = pt == 0 ? 0 : pt == 2 ? 2 : 1;
    const int tb = pt == 2 ? 1 : pt == 3 ? 3 : 2;
    const float sa = pt == 1 ? 1.f : -1.f;
    const float c0 = pt == 3 ? 0.f : 1.f;
    const float c1 = pt == 0 ? 0.f : pt == 1 ? 1.f : -1.f;

    int rowoff[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int r = 32 * u + i;
        const int zy = r < rows ? r / Cc : 0, ci = r < rows ? r - zy * Cc : 0;
        rowoff[u] = ((zy / 3) * HY + zy % 3) * HX * 32 + ci;
    }
    f32x16 acc[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    const int ntile = a.B * a.ntz * a.nty * a.ntx;
    const int per = (ntile + a.nsplit - 1) / a.nsplit;
    const int t_begin = split * per, t_end = min(ntile, t_begin + per);
    auto decode = [&](int tl, int& b, int& z0, int& y0, int& x0) {
        int t = tl;
        const int tx_ = t % a.ntx; t /= a.ntx;
        const int ty_ = t % a.nty; t /= a.nty;
        const int tz_ = t % a.ntz;
        b = t / a.ntz;
        z0 = tz_ * TZ; y0 = ty_ * TY; x0 = tx_ * TX;
    };
    // one body = 2 K-steps of two x-pairs each; x-pair b = (z, y, xb) with xb fastest: 64 per 2x8x8 tile = 16 bodies.
    // The RAW operand pairs are fetched one body ahead; the two-term combinations are formed right in front of the MFMA that
    // consumes them (a VALU op in the shadow of the previous MFMA), so no LDS latency is exposed between bodies.
    constexpr int NQ = 2;
    // per-lane LDS addresses are loop invariant (row offset, the lane's half of the x-pair couple, the point's two taps); the position
    // of the K-step inside the tile is a compile-time constant of the unrolled body => every ds_read is base register + immediate
    int offa[NTW], offb[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        offa[u] = rowoff[u] + kk * 64 + ta * 32;
        offb[u] = rowoff[u] + kk * 64 + tb * 32;
    }
    const int offy = kk * 2 * WG_NT + i;
    auto load_body = [&](float (&Ra)[NQ][NTW], float (&Rb)[NQ][NTW], float (&Y0)[NQ], float (&Y1)[NQ], const float* xs_c, const float* dys_c, int sb) {
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) {
            const int blk = 2 * (sb * NQ + q4);                 // even x-pair of the couple; the odd one is 2 voxels further along x
            const int z = blk >> 5, y = (blk >> 2) & 7, xb = blk & 3;
            const int hbk = ((z * HY + y) * HX + 2 * xb) * 32;
            const int v0 = ((z * 8 + y) * 8 + 2 * xb) * WG_NT;
            Y0[q4] = dys_c[offy + v0];
            Y1[q4] = dys_c[offy + v0 + WG_NT];
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                Ra[q4][u] = xs_c[offa[u] + hbk];
                Rb[q4][u] = xs_c[offb[u] + hbk];
            }
        }
    };
    auto mma_body = [&](const float (&Ra)[NQ][NTW], const float (&Rb)[NQ][NTW], const float (&Y0)[NQ], const float (&Y1)[NQ]) {
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) {
            const float bv = fmaf(c1, Y1[q4], c0 * Y0[q4]);
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const float av = fmaf(sa, Rb[q4][u], Ra[q4][u]);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u], 0, 0, 0);
            }
        }
    };
    float Ra0[NQ][NTW], Rb0[NQ][NTW], Ra1[NQ][NTW], Rb1[NQ][NTW], Ya0[NQ], Yb0[NQ], Ya1[NQ], Yb1[NQ];

    const int wave_u = pt;
    const float* zero = reinterpret_cast<const float*>(g_zero_page);
    constexpr int XIMG = 13 * 4 * 256;
    constexpr int SET = XIMG + MV * WG_NT;
    int nb = 0, nz0 = 0, ny0 = 0, nx0 = 0;
    bool more = false;
    auto issue_piece = [&](int pc, float* xd) {
        if (pc < 13) {
            const int j = tid + pc * 256;
            const int hv = j >> 3, q = j & 7;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = nz0 - 1 + hz, gy = ny0 - 1 + hy, gx = nx0 - 1 + hx;
            const bool ok = more && hv < HV && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W &&
                            ci0 + 4 * q < a.Cin;
            const float* src = ok ? a.in + (long)nb * a.in_bs + ((long)(gz * a.H + gy) * a.W + gx) * a.in_ps + ci0 + 4 * q : zero;
            dma16(src, xd + (wave_u + 4 * pc) * 256);
        } else {
            const int u = pc - 13;
            const int j = tid + u * 256;
            const int vv = j >> 3, q = j & 7;
            const int gz = nz0 + (vv >> 6), gy = ny0 + ((vv >> 3) & 7), gx = nx0 + (vv & 7);
            const bool ok = more && gz < a.D && gy < a.H && gx < a.W && co0 + 4 * q < a.Cout;
            const float* src = ok ? a.dy + (long)nb * a.dy_bs + ((long)(gz * a.H + gy) * a.W + gx) * a.dy_ps + co0 + 4 * q : zero;
            dma16(src, xd + XIMG + (wave_u + 4 * u) * 256);
        }
    };
    if (t_begin < t_end) {
        more = true;
        decode(t_begin, nb, nz0, ny0, nx0);
#pragma unroll
        for (int pc = 0; pc < 17; ++pc) issue_piece(pc, smem);
    }
    int cur = 0;
    for (int tl = t_begin; tl < t_end; ++tl, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        more = tl + 1 < t_end;
        if (more) decode(tl + 1, nb, nz0, ny0, nx0);
        const float* xs_c = smem + cur * SET;
        const float* dys_c = xs_c + XIMG;
        float* xnext = smem + (cur ^ 1) * SET;
        load_body(Ra0, Rb0, Ya0, Yb0, xs_c, dys_c, 0);
#pragma unroll
        for (int sb = 0; sb < 16; sb += 2) {            // 16 bodies; one DMA piece of the next tile per body (+ the 17th on the last trip)
            // the raw reads of the next body are threaded through the MFMAs of the current one (1 MFMA : 3 LDS reads : 2 VALU), so
            // neither their issue slots nor their latency stall the matrix pipe of this single-wave-per-SIMD kernel
            load_body(Ra1, Rb1, Ya1, Yb1, xs_c, dys_c, sb + 1);
            if (sb < 8) { issue_piece(2 * sb, xnext); issue_piece(2 * sb + 1, xnext); }     // all 17 pieces in the first nine bodies
            if (sb == 8) issue_piece(16, xnext);
            mma_body(Ra0, Rb0, Ya0, Yb0);
#pragma unroll
            for (int g = 0; g < NQ * NTW; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_body(Ra0, Rb0, Ya0, Yb0, xs_c, dys_c, (sb + 2) & 15);
            if (sb < 8) { issue_piece(2 * sb + 2, xnext); issue_piece(2 * sb + 3, xnext); }
            mma_body(Ra1, Rb1, Ya1, Yb1);
#pragma unroll
            for (int g = 0; g < NQ * NTW; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // flush: dw[.., t] = G^T M needs the four points of a (dz, dy, ci, co) entry, which live in the four waves.  They meet in LDS (the image
    // sets are free now), eight row tiles at a time, and each entry leaves as three float atomics (tap 0: M0 + (M1 + M2)/2, tap 1:
    // (M1 - M2)/2, tap 2: (M1 + M2)/2 + M3) in 128-byte runs of couts - 2.7x fewer atomics than flushing every wave's share separately.
    __syncthreads();
    float* X = smem;                                       // [point][tile slot 0..7][r][lane]
    const int co = co0 + i;
    constexpr int GRP = 8;
    for (int g0 = 0; g0 < nrt; g0 += GRP) {
        const int ng = min(GRP, nrt - g0);
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            if (u >= g0 && u < g0 + ng) {
#pragma unroll
                for (int r = 0; r < 16; ++r) X[((pt * GRP + (u - g0)) * 16 + r) * 64 + lane] = acc[u][r];
            }
        }
        __syncthreads();
        for (int j = wave; j < ng * 16; j += 4) {
            const int tl = j >> 4, r = j & 15;
            const float m0 = X[((0 * GRP + tl) * 16 + r) * 64 + lane], m1 = X[((1 * GRP + tl) * 16 + r) * 64 + lane];
            const float m2 = X[((2 * GRP + tl) * 16 + r) * 64 + lane], m3 = X[((3 * GRP + tl) * 16 + r) * 64 + lane];
            const int rg = 32 * (g0 + tl) + (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (rg < rows && co < a.Cout) {
                const int zy = rg / Cc, ci = rg - zy * Cc;
                float* d = a.dwp + ((long)(zy * 3) * a.Cin + ci0 + ci) * a.NPad + co;
                const float hs = 0.5f * (m1 + m2);
                atomicAdd(d, m0 + hs);
                atomicAdd(d + (long)a.Cin * a.NPad, 0.5f * (m1 - m2));
                atomicAdd(d + 2L * a.Cin * a.NPad, hs + m3);
            }
        }
        __syncthreads();
    }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cin, int Cout, int NPad, long total, int accumulate) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(e % 27);
        const long r = e / 27;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        const float val = dwp[((long)tap * Cin + ci) * NPad + co];
        dw[e] = accumulate ? dw[e] + val : val;
    }
}

// Cin chunk staged per pass: 16 channels (27 KB halo tile + 8 KB weight double buffer => 4 workgroups per CU; measured
// faster than 32- and 8-channel chunks on MI355X), 4 for the 2-/3-channel input layers
int pick_ch(int K) { return K <= 4 ? 4 : 16; }

template <int CH, int NT, bool VEC, int TZv>
int launch_conv_tz(const ConvArgs& a, int nblk, hipStream_t st) {
    constexpr size_t lds = (size_t)((((TZv + 2) * HY * HX * (CH + 1) + 3) & ~3) + 2 * CH * NT) * sizeof(float);
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
int pulpo_conv::conv_tz(int D, int H, int W) {
    return (D % 4 == 0 && (long)D * H * W >= 20L * 20 * 20) ? 4 : 2;
}

int pulpo_conv::launch_splitk_reduce(const float* part, int ksplit, float* out, long obs, long ops, long ocs, int B, long V, int C, int nrow,
                                     float* stats, const float* coef, float slope, hipStream_t st) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nrow, pulpo::cdiv(C, 32)), dim3(256), 0, st, part, ksplit, out, obs, ops, ocs, B, V, C, nrow, stats,
                       coef, slope);
    return pulpo::check_launch("splitk_reduce");
}

int pulpo_conv::launch_unpack_wgrad(const float* packed, float* dw, int Cin, int Cout, int accumulate, hipStream_t st) {
    const long total = (long)Cout * Cin * 27;
    const int ub = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(ub), dim3(256), 0, st, packed, dw, Cin, Cout, npad(Cout), total, accumulate);
    return pulpo::check_launch("unpack_wgrad");
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
    ConvArgs a;
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
    if (CH == 4) rc = NT == 64 ? launch_conv<4, 64, false>(a, nblk, st, tz) : launch_conv<4, 32, false>(a, nblk, st, tz);
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

PULPO_API size_t pulpo_conv3d_k3_wgrad_scratch_floats(int Cin, int Cout) { return (size_t)27 * Cin * npad(Cout); }

// dw[Cout][Cin][27] (+)= sum_vox in[vox+tap-1][ci] * dy[vox][co]  (accumulate != 0 adds to dw, e.g. a parameter's .grad storage).
// scratch: pulpo_conv3d_k3_wgrad_scratch_floats floats.
PULPO_API int pulpo_conv3d_k3_wgrad(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dy, int64_t dy_bs,
                                    int64_t dy_ps, int64_t dy_cs, float* dw, int accumulate, float* scratch, int B, int D, int H, int W,
                                    int Cin, int Cout, void* stream) {
    PULPO_REQUIRE(in && dy && dw && scratch, "conv3d_k3_wgrad: null pointer");
    PULPO_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3d_k3_wgrad: bad dims");
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a;
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps; a.in_cs = in_cs;
    a.dy = dy; a.dy_bs = dy_bs; a.dy_ps = dy_ps; a.dy_cs = dy_cs;
    a.dwp = scratch;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.NPad = npad(Cout);
    a.ntz = pulpo::cdiv(D, TZ); a.nty = pulpo::cdiv(H, TY); a.ntx = pulpo::cdiv(W, TX);
    a.ncit = pulpo::cdiv(Cin, WG_CH); a.ncot = pulpo::cdiv(Cout, WG_NT);
    const int ntile = B * a.ntz * a.nty * a.ntx;
    const int npair = a.ncit * a.ncot;
    const bool vec = (in_cs == 1) && (in_ps % 4 == 0) && (in_bs % 4 == 0) && (Cin % 4 == 0) && (((uintptr_t)in & 15) == 0) &&
                     (dy_cs == 1) && (dy_ps % 4 == 0) && (dy_bs % 4 == 0) && (Cout % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    // DMA variant: one resident workgroup per CU -> one round of <= 256 persistent workgroups (fewest atomic flushes);
    // scalar variant: two per CU
    int nsplit = std::max(1, (vec ? 256 : 512) / npair);
    nsplit = std::min(nsplit, ntile);
    a.nsplit = nsplit;
    hipError_t e = hipMemsetAsync(scratch, 0, pulpo_conv3d_k3_wgrad_scratch_floats(Cin, Cout) * sizeof(float), st);
    if (e != hipSuccess) return pulpo::fail((int)e, "wgrad memset: %s", hipGetErrorString(e));
    const int nrt_max = (27 * std::min(Cin, WG_CH) + 31) / 32;
    const int ntw = (nrt_max + 3) / 4;                 // 1..7
    constexpr size_t lds = (size_t)(((HV * WG_CP + 3) & ~3) + MV * WG_NT) * sizeof(float);
    constexpr size_t lds_dma = (size_t)2 * (13 * 4 * 256 + MV * WG_NT) * sizeof(float);
    const int nblk = npair * nsplit;
    int rc = 0;
#define PULPO_WGRAD(VECV, NTWV)                                                                                                   \
    {                                                                                                                             \
        static bool attr = false;                                                                                                 \
        const size_t bytes = VECV ? lds_dma : lds;                                                                                \
        if (!attr) {                                                                                                              \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_mfma<VECV, NTWV>),                 \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                        \
            if (ea != hipSuccess) return pulpo::fail((int)ea, "hipFuncSetAttribute(wgrad): %s", hipGetErrorString(ea));           \
            attr = true;                                                                                                          \
        }                                                                                                                         \
        hipLaunchKernelGGL((conv3d_k3_wgrad_mfma<VECV, NTWV>), dim3(nblk), dim3(256), bytes, st, a);                              \
    }
    static int wino = -1;
    if (wino < 0) { const char* e = getenv("PULPO_WGRAD_WINOGRAD"); wino = e ? atoi(e) : 1; }
    if (vec && wino && Cin >= 8 && (long)D * H * W >= 32L * 32 * 32) {      // (measured: no gain on the 20^3 / 10^3 pyramid levels)
        // Winograd-x variant: wave = transformed point, nine (dz, dy) row tiles of <= 32 channels
        const int nrt9 = (9 * std::min(Cin, WG_CH) + 31) / 32;
#define PULPO_WGRAD_W(NTWV)                                                                                                       \
    {                                                                                                                             \
        static bool attr = false;                                                                                                 \
        if (!attr) {                                                                                                              \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_wino<NTWV>),                       \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);                      \
            if (ea != hipSuccess) return pulpo::fail((int)ea, "hipFuncSetAttribute(wgrad wino): %s", hipGetErrorString(ea));      \
            attr = true;                                                                                                          \
        }                                                                                                                         \
        hipLaunchKernelGGL((conv3d_k3_wgrad_wino<NTWV>), dim3(nblk), dim3(256), lds_dma, st, a);                                  \
    }
        if (nrt9 <= 3) PULPO_WGRAD_W(3) else if (nrt9 <= 5) PULPO_WGRAD_W(5) else PULPO_WGRAD_W(9)
#undef PULPO_WGRAD_W
    } else if (vec) {
        if (ntw <= 1) PULPO_WGRAD(true, 1) else if (ntw <= 2) PULPO_WGRAD(true, 2) else if (ntw <= 4) PULPO_WGRAD(true, 4) else PULPO_WGRAD(true, 7)
    } else {
        if (ntw <= 1) PULPO_WGRAD(false, 1) else if (ntw <= 2) PULPO_WGRAD(false, 2) else if (ntw <= 4) PULPO_WGRAD(false, 4) else PULPO_WGRAD(false, 7)
    }
#undef PULPO_WGRAD
    rc = pulpo::check_launch("conv3d_k3_wgrad_mfma");
    if (rc) return rc;
    return pulpo_conv::launch_unpack_wgrad(scratch, dw, Cin, Cout, accumulate, st);
}

// forward / data-gradient kernel for a shape: 2 = Winograd F(2x2,3x3) in (y, x) (default where a Winograd kernel applies: 4x8x8-tiled
// volumes, > 4 reduction channels), 1 = Winograd F(2,3) along x only, 0 = direct implicit GEMM
PULPO_API int pulpo_conv3d_k3_algo(int B, int D, int H, int W, int K, int N) {
    static int force = -1;
    if (force < 0) { const char* e = getenv("PULPO_CONV_WINOGRAD"); force = e ? atoi(e) + 1 : 0; }     // unset: policy; 0 / 1: force off / on
    if (force == 1) return 0;
    const bool shape_ok = K > 4 && conv_tz(D, H, W) == 4;
    static int two = -1;
    if (two < 0) { const char* e = getenv("PULPO_CONV_WINOGRAD_2D"); two = e ? atoi(e) : 1; }      // default: the (y, x) kernel
    return shape_ok ? (two ? 2 : 1) : 0;
}

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
