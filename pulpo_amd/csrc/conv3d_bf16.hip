// 3x3x3 / pad 1 convolution with bf16 OPERANDS and fp32 accumulation (BASELINE configs 4-5: "bf16"): activations stay fp32
// in HBM, are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while the halo tile is staged into LDS, and the products run on
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).  The reference has no such mode; its definition here
// is "conv operands rounded to bf16, everything else fp32", and the oracle emulates exactly that.
//
//   forward / dgrad :  D[voxel][cout] += A[voxel][(tap,cin)] * B[(tap,cin)][cout]        A, B bf16; D fp32
//   wgrad           :  D[(tap,cin)][cout] += A[(tap,cin)][voxel] * B[voxel][cout]
//
// Same tiling, grid order, split-K and BatchNorm partial statistics as conv3d.hip; a 32-channel chunk per pass,
// LDS tiles [voxel][32 + 8] bf16 (80-byte rows: 16-byte fragments, conflict-free ds_read_b128).
#include "conv_shared.h"
#include "act_io.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

namespace {

using pulpo_conv::TY; using pulpo_conv::TX; using pulpo_conv::HY; using pulpo_conv::HX;
using pulpo_conv::npad;

// z extent of this kernel's voxel tile (its own policy: the 4x8x8 tile pays from 64^3 up; pulpo_conv3d_k3_fwd_bf16_stat_tiles follows it)
inline int conv_tz(int D, int H, int W) { return (D % 4 == 0 && (long)D * H * W >= 64L * 64 * 64) ? 4 : 2; }

constexpr int CH = 32;            // channels per staged chunk = two K=16 MFMA steps
constexpr int CP = CH + 8;        // LDS row length in bf16 elements

struct ConvArgsH {
    const void* in;               // fp32 or bf16 (the kernel's T); strides in elements
    long in_bs, in_ps, in_cs;
    const uint16_t* wp;           // packed bf16 [nchunk][27][NPad][32]
    const float* bias;
    void* out;                    // same element type as `in`
    long out_bs, out_ps, out_cs;
    float* stats;
    int B, D, H, W, Cin, Cout, NPad;
    int ntz, nty, ntx, ncot;
    int ksplit;
    float* part;
    const float* coef;            // nullable: fused eval-mode BatchNorm + LeakyReLU (see conv3d.hip)
    float slope;
};

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    bf16x2 p = {(__bf16)lo, (__bf16)hi};            // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(uint32_t, p);
}

__device__ __forceinline__ int tap_halo_offset(int tap) { return ((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3; }

// stage the halo tile of channels [c0, c0+32) as bf16 into xs[HV][CP]; zero outside the volume / beyond Cin
// T = pulpo::bf16_t: the tensor already holds bf16 - the vector path is one 16-byte load and one 16-byte LDS store per 8-channel piece
template <bool VEC, int TZv, typename T>
__device__ __forceinline__ void stage_halo_bf16(uint16_t* xs, const T* __restrict__ in, long in_ps, long in_cs, int c0, int Cin, int z0, int y0,
                                                int x0, int D, int H, int W, int tid) {
    constexpr int HV = (TZv + 2) * HY * HX;
    if constexpr (VEC && sizeof(T) == 2) {
        constexpr int Q = CH / 8;
        constexpr int NIT = (HV * Q + 255) / 256;
        uint4 v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            const int hv = j / Q, q = j - hv * Q;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            v[u] = make_uint4(0, 0, 0, 0);
            if (j < HV * Q && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c0 + 8 * q < Cin)
                v[u] = *reinterpret_cast<const uint4*>(in + ((long)(gz * H + gy) * W + gx) * in_ps + c0 + 8 * q);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            if (j < HV * Q) {
                const int hv = j / Q, q = j - hv * Q;
                *reinterpret_cast<uint4*>(xs + hv * CP + 8 * q) = v[u];
            }
        }
    } else if constexpr (VEC) {
        constexpr int Q = CH / 8;                       // 8-channel pieces per voxel: two float4 in, one 16-byte LDS store out
        constexpr int NIT = (HV * Q + 255) / 256;
        float4 v[NIT][2];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            const int hv = j / Q, q = j - hv * Q;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            v[u][0] = v[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < HV * Q && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                const float* p = in + ((long)(gz * H + gy) * W + gx) * in_ps + c0 + 8 * q;
                if (c0 + 8 * q < Cin) v[u][0] = *reinterpret_cast<const float4*>(p);
                if (c0 + 8 * q + 4 < Cin) v[u][1] = *reinterpret_cast<const float4*>(p + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            if (j < HV * Q) {
                const int hv = j / Q, q = j - hv * Q;
                uint4 o;
                o.x = pack2(v[u][0].x, v[u][0].y); o.y = pack2(v[u][0].z, v[u][0].w);
                o.z = pack2(v[u][1].x, v[u][1].y); o.w = pack2(v[u][1].z, v[u][1].w);
                *reinterpret_cast<uint4*>(xs + hv * CP + 8 * q) = o;
            }
        }
    } else {
        for (int j = tid; j < HV * (CH / 2); j += 256) {
            const int hv = j / (CH / 2), c = 2 * (j - hv * (CH / 2));
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            float v0[1] = {0.f}, v1[1] = {0.f};
            if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                const T* p = in + ((long)(gz * H + gy) * W + gx) * in_ps;
                if (c0 + c < Cin) pulpo::ldv<1>(p + (long)(c0 + c) * in_cs, v0);
                if (c0 + c + 1 < Cin) pulpo::ldv<1>(p + (long)(c0 + c + 1) * in_cs, v1);
            }
            *reinterpret_cast<uint32_t*>(xs + hv * CP + c) = pack2(v0[0], v1[0]);
        }
    }
}

template <int NT, bool VEC, int TZv, typename T = float>
__global__ __launch_bounds__(256, 2) void conv3d_k3_mfma_bf16(ConvArgsH a) {
    constexpr int NN = NT / 32;
    constexpr int MT = TZv / 2;
    constexpr int HV = (TZv + 2) * HY * HX;
#ifndef PULPO_BF16_TPB32
#define PULPO_BF16_TPB32 1
#endif
    constexpr int TPB = (TZv == 4 && (NT == 64 || PULPO_BF16_TPB32 == 3)) ? 3 : 1;  // taps (one dx row) per barrier: 3 on the big 64-cout tiles = 24 MFMAs per wave between
                                                          // barriers (on the 32-cout tiles the extra registers cost the third wave per SIMD: measured slower)
    constexpr int WSLAB = TPB * NT * CP;                // bf16 elements of one LDS weight slab set [TPB][NT][CP]
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_h[];
    uint16_t* xs = smem_h;                              // [HV][CP]
    uint16_t* ws = smem_h + HV * CP;                    // [2][TPB][NT][CP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid0 = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int split = lid0 % a.ksplit;
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
    constexpr int NIT = 27 / TPB;                       // barrier iterations per chunk
    const int it0 = chunk0 * NIT, niter = chunk1 * NIT;
    const T* in_b = reinterpret_cast<const T*>(a.in) + (long)b * a.in_bs;

    // weight slab of one tap: NT rows (cout) x 32 k bf16 = NT*4 pieces of 16 bytes; one piece per thread (NT = 64) or per low thread (NT = 32)
    const int wrow = tid >> 2, wpiece = tid & 3;
    const bool wact = wrow < NT;
    uint4 wr0 = make_uint4(0, 0, 0, 0), wr1 = wr0, wr2 = wr0;      // (scalars: an array indexed in the lambdas ends up in scratch)
    auto load_w = [&](int it) {
        if (wact) {
            const uint16_t* p = a.wp + (((long)it * TPB) * a.NPad + co0 + wrow) * CH + wpiece * 8;
            wr0 = *reinterpret_cast<const uint4*>(p);
            if constexpr (TPB == 3) {
                wr1 = *reinterpret_cast<const uint4*>(p + (long)a.NPad * CH);
                wr2 = *reinterpret_cast<const uint4*>(p + 2L * a.NPad * CH);
            }
        }
    };
    auto store_w = [&](int buf) {
        if (wact) {
            uint16_t* d = ws + buf * WSLAB + wrow * CP + wpiece * 8;
            *reinterpret_cast<uint4*>(d) = wr0;
            if constexpr (TPB == 3) {
                *reinterpret_cast<uint4*>(d + NT * CP) = wr1;
                *reinterpret_cast<uint4*>(d + 2 * NT * CP) = wr2;
            }
        }
    };

    const int i = lane & 31, kk = lane >> 5;
    int hb[MT];
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
        __syncthreads();
        stage_halo_bf16<VEC, TZv, T>(xs, in_b, a.in_ps, a.in_cs, chunk * CH, a.Cin, z0, y0, x0, a.D, a.H, a.W, tid);
        for (int itc = 0; itc < NIT; ++itc, ++it) {
            store_w(buf);
            __syncthreads();
            if (it + 1 < niter) load_w(it + 1);
            const uint16_t* wb = ws + buf * WSLAB + i * CP + kk * 8;
            // steps = (tap of this iteration, K half); the fragments of step n + (SLOTS - 1) are requested before the MFMAs of step n
            constexpr int STEPS = TPB * (CH / 16);
            constexpr int SLOTS = STEPS < 3 ? STEPS : 3;
            bf16x8 av[SLOTS][MT], bv[SLOTS][NN];
            auto fetch = [&](int st, int slot) {
                const int d = st / (CH / 16), ks = st % (CH / 16);
                const int off = tap_halo_offset(itc * TPB + d);
#pragma unroll
                for (int m = 0; m < MT; ++m) av[slot][m] = *reinterpret_cast<const bf16x8*>(xs + (hb[m] + off) * CP + ks * 16 + kk * 8);
#pragma unroll
                for (int n = 0; n < NN; ++n) bv[slot][n] = *reinterpret_cast<const bf16x8*>(wb + (d * NT + n * 32) * CP + ks * 16);
            };
#pragma unroll
            for (int st = 0; st < SLOTS - 1; ++st) fetch(st, st);
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                if (st + SLOTS - 1 < STEPS) fetch(st + SLOTS - 1, (st + SLOTS - 1) % SLOTS);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NN; ++n)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[st % SLOTS][m], bv[st % SLOTS][n], acc[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }
    }

    // ---- epilogue: bias, store, per-tile BatchNorm partial statistics (identical to the fp32 kernel).  bf16 output: neighbouring lanes hold
    // neighbouring channels of one voxel - the even lane takes its neighbour's value (DPP) and stores both as one packed dword; the
    // statistics describe the tensor as stored (rounded)
    T* out_b = reinterpret_cast<T*>(a.out) + (long)b * a.out_bs;
    constexpr bool HALF = sizeof(T) == 2;
    const bool pair_ok = HALF && a.out_cs == 1 && (a.Cout & 1) == 0 && a.ksplit == 1;
    float ssum[NN], ssq[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
        const int co = co0 + n * 32 + i;
        const bool cok = co < a.Cout;
        const float bias = (a.bias != nullptr && cok && split == 0) ? a.bias[co] : 0.f;
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
                const bool inb = gz < a.D && gy < a.H && gx < a.W;
                float val = acc[m][n][r] + bias;
                if (a.ksplit == 1) {
                    if (fuse) {
                        const float t = pulpo::as_stored<T>(val) * fsc + fsh;        // (the pre-norm value as the unfused path would have stored it)
                        val = t > 0.f ? t : t * a.slope;
                    }
                    val = pulpo::as_stored<T>(val);
                }
                const long vox = (long)(gz * a.H + gy) * a.W + gx;
                if constexpr (HALF) {
                    const float nb = __shfl_xor(val, 1, 64);               // (all lanes take part)
                    if (pair_ok) {
                        if (cok && inb && (i & 1) == 0)
                            *reinterpret_cast<uint32_t*>(out_b + vox * a.out_ps + co) = pulpo::pack_bf2(val, nb);
                    } else if (cok && inb && a.ksplit == 1) {
                        out_b[vox * a.out_ps + (long)co * a.out_cs] = pulpo::f2bf(val);
                    }
                } else {
                    if (cok && inb && a.ksplit == 1) out_b[vox * a.out_ps + (long)co * a.out_cs] = val;
                }
                if (cok && inb) {
                    if (a.ksplit > 1) a.part[(((long)split * a.B + b) * a.D * a.H * a.W + vox) * a.Cout + co] = val;
                    s += val;
                    q += val * val;
                }
            }
        }
        ssum[n] = s + __shfl_xor(s, 32, 64);
        ssq[n] = q + __shfl_xor(q, 32, 64);
    }
    if (a.stats != nullptr && a.ksplit == 1) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(ws);      // [4 waves][2][NT] floats = 2 KB <= the weight double buffer
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

// w: PyTorch layout [Cout][Cin][27] fp32 -> bf16 wp[k/32][tap][n (NPad)][k%32]
//   forward: K = Cin, N = Cout, value w[n][k][tap];  dgrad: K = Cout, N = Cin, value w[k][n][26 - tap]
__global__ void pack_weight_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp, int Cin, int Cout, int NPad, int dgrad, long total) {
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int kc = (int)(e % CH);
        long r = e / CH;
        const int n = (int)(r % NPad); r /= NPad;
        const int tap = (int)(r % 27);
        const int chunk = (int)(r / 27);
        const int k = chunk * CH + kc;
        float val = 0.f;
        if (k < K && n < N) val = dgrad ? w[((long)k * Cin + n) * 27 + (26 - tap)] : w[((long)n * Cin + k) * 27 + tap];
        wp[e] = __builtin_bit_cast(uint16_t, (__bf16)val);
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct WgradArgsH {
    const void* in;               // fp32 or bf16 (the kernel's T, shared by both operands); strides in elements
    long in_bs, in_ps, in_cs;
    const void* dy;
    long dy_bs, dy_ps, dy_cs;
    float* dwp;                   // zero-initialised fp32 scratch [27][Cin][NPad], accumulated with float atomics
    int B, D, H, W, Cin, Cout, NPad;
    int ntz, nty, ntx, ncit, ncot, nsplit;
};

constexpr int WTZ = 2, WMV = WTZ * TY * TX, WHV = (WTZ + 2) * HY * HX;     // 2x8x8 voxel tiles: 8 K=16 steps (one x-row of voxels per half-wave)

// stage `nvox` voxels x 32 channels [c0, c0+32) as bf16 into dst[nvox][CP]; voxel -> global coordinates through `coord`
template <bool VEC, int NVOX, bool HALO, typename T>
__device__ __forceinline__ void stage_tile_bf16(uint16_t* dst, const T* __restrict__ src, long ps, long cs, int c0, int C, int z0, int y0, int x0,
                                                int D, int H, int W, int tid) {
    auto coord = [&](int v, int& gz, int& gy, int& gx) {
        if constexpr (HALO) {
            const int hz = v / (HY * HX), rem = v - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            gz = z0 - 1 + hz; gy = y0 - 1 + hy; gx = x0 - 1 + hx;
        } else {
            gz = z0 + (v >> 6); gy = y0 + ((v >> 3) & 7); gx = x0 + (v & 7);
        }
    };
    if constexpr (VEC && sizeof(T) == 2) {
        constexpr int Q = CH / 8;
        constexpr int NIT = (NVOX * Q + 255) / 256;
        uint4 v[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            const int vv = j / Q, q = j - vv * Q;
            int gz, gy, gx;
            coord(vv, gz, gy, gx);
            v[u] = make_uint4(0, 0, 0, 0);
            if (j < NVOX * Q && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c0 + 8 * q < C)
                v[u] = *reinterpret_cast<const uint4*>(src + ((long)(gz * H + gy) * W + gx) * ps + c0 + 8 * q);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            if (j < NVOX * Q) {
                const int vv = j / Q, q = j - vv * Q;
                *reinterpret_cast<uint4*>(dst + vv * CP + 8 * q) = v[u];
            }
        }
    } else if constexpr (VEC) {
        constexpr int Q = CH / 8;
        constexpr int NIT = (NVOX * Q + 255) / 256;
        float4 v[NIT][2];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            const int vv = j / Q, q = j - vv * Q;
            int gz, gy, gx;
            coord(vv, gz, gy, gx);
            v[u][0] = v[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < NVOX * Q && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                const float* p = src + ((long)(gz * H + gy) * W + gx) * ps + c0 + 8 * q;
                if (c0 + 8 * q < C) v[u][0] = *reinterpret_cast<const float4*>(p);
                if (c0 + 8 * q + 4 < C) v[u][1] = *reinterpret_cast<const float4*>(p + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int j = tid + u * 256;
            if (j < NVOX * Q) {
                const int vv = j / Q, q = j - vv * Q;
                uint4 o;
                o.x = pack2(v[u][0].x, v[u][0].y); o.y = pack2(v[u][0].z, v[u][0].w);
                o.z = pack2(v[u][1].x, v[u][1].y); o.w = pack2(v[u][1].z, v[u][1].w);
                *reinterpret_cast<uint4*>(dst + vv * CP + 8 * q) = o;
            }
        }
    } else {
        for (int j = tid; j < NVOX * (CH / 2); j += 256) {
            const int vv = j / (CH / 2), c = 2 * (j - vv * (CH / 2));
            int gz, gy, gx;
            coord(vv, gz, gy, gx);
            float v0[1] = {0.f}, v1[1] = {0.f};
            if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                const T* p = src + ((long)(gz * H + gy) * W + gx) * ps;
                if (c0 + c < C) pulpo::ldv<1>(p + (long)(c0 + c) * cs, v0);
                if (c0 + c + 1 < C) pulpo::ldv<1>(p + (long)(c0 + c + 1) * cs, v1);
            }
            *reinterpret_cast<uint32_t*>(dst + vv * CP + c) = pack2(v0[0], v1[0]);
        }
    }
}

// K = voxels: lane (row i, half kk) supplies the 8 voxels of one x-row of the tile at its fixed channel -> eight 2-byte LDS reads
// per fragment (voxel-major images, 80-byte rows).  The matrix pipe is 16x faster than in fp32, so this kernel is bound by
// those reads; it still beats the fp32 wgrad several times over.
__device__ __forceinline__ bf16x8 gather8(const uint16_t* p) {          // p[t * CP], t = 0..7
    uint4 o;
    o.x = (uint32_t)p[0 * CP] | ((uint32_t)p[1 * CP] << 16);
    o.y = (uint32_t)p[2 * CP] | ((uint32_t)p[3 * CP] << 16);
    o.z = (uint32_t)p[4 * CP] | ((uint32_t)p[5 * CP] << 16);
    o.w = (uint32_t)p[6 * CP] | ((uint32_t)p[7 * CP] << 16);
    return __builtin_bit_cast(bf16x8, o);
}

// TR: the fragments come out of the voxel-major images through gfx950's transposing LDS read (ds_read_b64_tr_b16: a group of 16 lanes reads
// 4 voxel rows x 16 channel columns and each lane receives ONE column's four voxels): two reads per 8-voxel fragment instead of eight 2-byte
// reads and four packing operations - the LDS instruction issue that bounded this kernel.  Needs every 16-lane group's 16 GEMM rows to be
// 16 consecutive channels of one tap: Cin % 16 == 0 (every layer of the BASELINE configurations).
typedef short tr_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* p, int hi_off) {
    const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(p));
    const tr_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(p + hi_off));
    typedef short tr_v8s __attribute__((ext_vector_type(8)));
    const tr_v8s v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return __builtin_bit_cast(bf16x8, v);
}

template <int NTW, bool VEC, typename T = float, bool TR = false>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wgrad_bf16(WgradArgsH a) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_h[];
    uint16_t* xs = smem_h;                     // [WHV][CP]
    uint16_t* dys = smem_h + WHV * CP;         // [WMV][CP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int npair = a.ncit * a.ncot;
    const int pair = lid % npair, split = lid / npair;
    const int cit = pair / a.ncot, cot = pair - cit * a.ncot;
    const int ci0 = cit * CH, co0 = cot * 32;
    const int Cc = min(CH, a.Cin - ci0);
    const int rows = 27 * Cc;
    const int nrt = (rows + 31) >> 5;                 // host guarantees nrt <= 4 * NTW
    const int i = lane & 31, kk = lane >> 5;

    int rowoff[NTW];
    // (TR) lane 4 q + p of a 16-lane group supplies the address of voxel row q, channels 4 p .. 4 p + 3 of the group's 16 channel columns
    const int tg = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        if constexpr (TR) {
            const int r = 32 * (wave + 4 * u) + 16 * tg;                                  // first GEMM row of this lane's group
            const int tap = r < rows ? r / Cc : 0, ci = r < rows ? r - tap * Cc : 0;      // spare groups read tap 0 / channel 0; never flushed
            rowoff[u] = (tap_halo_offset(tap) + tq) * CP + ci + 4 * tp;
        } else {
            const int r = 32 * (wave + 4 * u) + i;
            const int tap = r < rows ? r / Cc : 0, ci = r < rows ? r - tap * Cc : 0;      // spare rows read tap 0 / channel 0; never flushed
            rowoff[u] = tap_halo_offset(tap) * CP + ci;
        }
    }
    const int bvoff = TR ? tq * CP + 16 * tg + 4 * tp : i;
    f32x16 acc[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    const int ntile = a.B * a.ntz * a.nty * a.ntx;
    const int per = (ntile + a.nsplit - 1) / a.nsplit;
    const int t_begin = split * per, t_end = min(ntile, t_begin + per);
    for (int tl = t_begin; tl < t_end; ++tl) {
        int t = tl;
        const int tx_ = t % a.ntx; t /= a.ntx;
        const int ty_ = t % a.nty; t /= a.nty;
        const int tz_ = t % a.ntz;
        const int b = t / a.ntz;
        const int z0 = tz_ * WTZ, y0 = ty_ * TY, x0 = tx_ * TX;
        __syncthreads();
        stage_tile_bf16<VEC, WHV, true, T>(xs, reinterpret_cast<const T*>(a.in) + (long)b * a.in_bs, a.in_ps, a.in_cs, ci0, a.Cin, z0, y0, x0, a.D, a.H, a.W, tid);
        stage_tile_bf16<VEC, WMV, false, T>(dys, reinterpret_cast<const T*>(a.dy) + (long)b * a.dy_bs, a.dy_ps, a.dy_cs, co0, a.Cout, z0, y0, x0, a.D, a.H, a.W, tid);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < WMV / 16; ++ks) {
            const int vrow = 2 * ks + kk;                                   // x-row of the tile: z = vrow >> 3, y = vrow & 7
            const int hbase = ((vrow >> 3) * HY + (vrow & 7)) * HX * CP;
            const bf16x8 bv = TR ? tr_frag(dys + vrow * 8 * CP + bvoff, 4 * CP) : gather8(dys + vrow * 8 * CP + bvoff);
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const bf16x8 av = TR ? tr_frag(xs + hbase + rowoff[u], 4 * CP) : gather8(xs + hbase + rowoff[u]);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[u], 0, 0, 0);
            }
        }
    }

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

int nt_for(int N) { return (N % 64 == 0) ? 64 : 32; }

int conv_ksplit_bf16(int B, int D, int H, int W, int K, int N) {
    const long nblk = (long)B * pulpo::cdiv(D, conv_tz(D, H, W)) * pulpo::cdiv(H, TY) * pulpo::cdiv(W, TX) * pulpo::cdiv(N, nt_for(N));
    const int nchunk = (K + CH - 1) / CH;
    if (nblk >= 512 || nchunk <= 1) return 1;
    return (int)std::max<long>(1, std::min<long>(std::min(nchunk, 8), 1024 / nblk));
}

template <int NT, bool VEC, int TZv, typename T>
int launch_bf16(const ConvArgsH& a, int nblk, hipStream_t st) {
    constexpr size_t lds = (size_t)((TZv + 2) * HY * HX * CP + 2 * ((TZv == 4 && (NT == 64 || PULPO_BF16_TPB32 == 3)) ? 3 : 1) * NT * CP) * sizeof(uint16_t);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_mfma_bf16<NT, VEC, TZv, T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(conv3d bf16): %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3d_k3_mfma_bf16<NT, VEC, TZv, T>), dim3(nblk), dim3(256), lds, st, a);
    return pulpo::check_launch("conv3d_k3_mfma_bf16");
}

}  // namespace

// ================================================================================================ C ABI
PULPO_API size_t pulpo_conv3d_k3_packed_bf16_elems(int K, int N) { return (size_t)((K + CH - 1) / CH) * 27 * npad(N) * CH; }

PULPO_API int pulpo_conv3d_k3_pack_weight_bf16(const float* w, uint16_t* wp, int Cin, int Cout, int dgrad, void* stream) {
    PULPO_REQUIRE(w && wp && Cin > 0 && Cout > 0, "conv3d_k3_pack_weight_bf16: bad arguments");
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const long total = (long)pulpo_conv3d_k3_packed_bf16_elems(K, N);
    const int nb = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wp, Cin, Cout, npad(N), dgrad, total);
    return pulpo::check_launch("pack_weight_bf16");
}

PULPO_API size_t pulpo_conv3d_k3_fwd_bf16_scratch_floats(int B, int D, int H, int W, int K, int N) {
    const int ks = conv_ksplit_bf16(B, D, H, W, K, N);
    return ks > 1 ? (size_t)ks * B * D * H * W * N : 0;
}

PULPO_API int pulpo_conv3d_k3_fwd_bf16_stat_tiles(int B, int D, int H, int W) {
    return B * pulpo::cdiv(D, conv_tz(D, H, W)) * pulpo::cdiv(H, TY) * pulpo::cdiv(W, TX);
}

// dt: dtype code of `in` AND `out` (0 fp32: the operands are rounded while they are staged; 1 bf16: activations are stored as bf16, the
// result is rounded on the store and the BatchNorm partials describe it as stored); strides in elements
static int conv_fwd_bf16_impl(const void* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const uint16_t* wp, const float* bias, void* out,
                              int64_t out_bs, int64_t out_ps, int64_t out_cs, int dt, float* stats, float* scratch, const float* coef, float slope,
                              int B, int D, int H, int W, int K, int N, void* stream) {
    PULPO_REQUIRE(in && wp && out, "conv3d_k3_fwd_bf16: null pointer");
    PULPO_REQUIRE(!(coef && stats), "conv3d_k3_fwd_bf16: batch statistics are not available from the fused eval-mode epilogue");
    PULPO_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && K > 0 && N > 0, "conv3d_k3_fwd_bf16: bad dims");
    PULPO_REQUIRE_DT(dt, "conv3d_k3_fwd_bf16");
    ConvArgsH a;
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps; a.in_cs = in_cs;
    a.wp = wp; a.bias = bias;
    a.out = out; a.out_bs = out_bs; a.out_ps = out_ps; a.out_cs = out_cs;
    a.stats = stats;
    a.coef = coef; a.slope = slope;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = K; a.Cout = N; a.NPad = npad(N);
    const int tz = conv_tz(D, H, W), NT = nt_for(N);
    a.ntz = pulpo::cdiv(D, tz); a.nty = pulpo::cdiv(H, TY); a.ntx = pulpo::cdiv(W, TX);
    a.ncot = pulpo::cdiv(N, NT);
    const long nblk_l = (long)B * a.ntz * a.nty * a.ntx * a.ncot;
    PULPO_REQUIRE(nblk_l < (1L << 31), "conv3d_k3_fwd_bf16: grid too large");
    a.ksplit = conv_ksplit_bf16(B, D, H, W, K, N);
    a.part = scratch;
    PULPO_REQUIRE(a.ksplit == 1 || scratch != nullptr, "conv3d_k3_fwd_bf16: scratch of pulpo_conv3d_k3_fwd_bf16_scratch_floats() floats required");
    const int nblk = (int)nblk_l * a.ksplit;
    // vector staging: 16-byte pieces = 4 fp32 / 8 bf16 channels
    const int g = dt ? 8 : 4;
    const bool vec = (in_cs == 1) && (in_ps % g == 0) && (in_bs % g == 0) && (K % g == 0) && (((uintptr_t)in & 15) == 0);
    if (dt) PULPO_REQUIRE(out_cs != 1 || (N & 1) || ((out_ps % 2 == 0) && (out_bs % 2 == 0) && (((uintptr_t)out & 3) == 0)),
                          "conv3d_k3_fwd_bf16: bf16 channels-last output must be 4-byte aligned with even strides");
    hipStream_t st = (hipStream_t)stream;
    int rc;
#define PULPO_BF16(NTV, VECV, TT) (tz == 4 ? launch_bf16<NTV, VECV, 4, TT>(a, nblk, st) : launch_bf16<NTV, VECV, 2, TT>(a, nblk, st))
#define PULPO_BF16_T(TT)                                                                          \
    do {                                                                                          \
        if (vec) rc = NT == 64 ? PULPO_BF16(64, true, TT) : PULPO_BF16(32, true, TT);             \
        else rc = NT == 64 ? PULPO_BF16(64, false, TT) : PULPO_BF16(32, false, TT);               \
    } while (0)
    if (dt) PULPO_BF16_T(pulpo::bf16_t); else PULPO_BF16_T(float);
#undef PULPO_BF16_T
#undef PULPO_BF16
    if (rc == 0 && a.ksplit > 1)
        rc = pulpo_conv::launch_splitk_reduce(scratch, a.ksplit, out, (long)out_bs, (long)out_ps, (long)out_cs, B, (long)D * H * W, N,
                                              pulpo_conv3d_k3_fwd_bf16_stat_tiles(B, D, H, W), stats, coef, slope, st, dt);
    return rc;
}

PULPO_API int pulpo_conv3d_k3_fwd_bf16_t(const void* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const uint16_t* wp, const float* bias, void* out,
                                         int64_t out_bs, int64_t out_ps, int64_t out_cs, int dt, float* stats, float* scratch, int B, int D, int H,
                                         int W, int K, int N, void* stream) {
    return conv_fwd_bf16_impl(in, in_bs, in_ps, in_cs, wp, bias, out, out_bs, out_ps, out_cs, dt, stats, scratch, nullptr, 0.f, B, D, H, W, K, N,
                              stream);
}

PULPO_API int pulpo_conv3d_k3_fwd_bf16(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const uint16_t* wp, const float* bias,
                                       float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats, float* scratch, int B, int D,
                                       int H, int W, int K, int N, void* stream) {
    return conv_fwd_bf16_impl(in, in_bs, in_ps, in_cs, wp, bias, out, out_bs, out_ps, out_cs, 0, stats, scratch, nullptr, 0.f, B, D, H, W, K, N,
                              stream);
}

PULPO_API int pulpo_conv3d_k3_fwd_bn_lrelu_bf16_t(const void* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const uint16_t* wp, const float* bias,
                                                  const float* coef, float slope, void* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, int dt,
                                                  float* scratch, int B, int D, int H, int W, int K, int N, void* stream) {
    PULPO_REQUIRE(coef, "conv3d_k3_fwd_bn_lrelu_bf16: null coef");
    return conv_fwd_bf16_impl(in, in_bs, in_ps, in_cs, wp, bias, out, out_bs, out_ps, out_cs, dt, nullptr, scratch, coef, slope, B, D, H, W, K, N,
                              stream);
}

PULPO_API int pulpo_conv3d_k3_fwd_bn_lrelu_bf16(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const uint16_t* wp,
                                                const float* bias, const float* coef, float slope, float* out, int64_t out_bs, int64_t out_ps,
                                                int64_t out_cs, float* scratch, int B, int D, int H, int W, int K, int N, void* stream) {
    return pulpo_conv3d_k3_fwd_bn_lrelu_bf16_t(in, in_bs, in_ps, in_cs, wp, bias, coef, slope, out, out_bs, out_ps, out_cs, 0, scratch, B, D, H, W, K,
                                               N, stream);
}

/* weight gradient with bf16 operands: dw[Cout][Cin][27] (+)= sum_voxels bf16(in[v + tap - 1][ci]) * bf16(dy[v][co]), fp32 accumulation */
PULPO_API size_t pulpo_conv3d_k3_wgrad_scratch_floats(int Cin, int Cout);

PULPO_API int pulpo_conv3d_k3_wgrad_bf16_t(const void* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const void* dy, int64_t dy_bs, int64_t dy_ps,
                                           int64_t dy_cs, int dt, float* dw, int accumulate, float* scratch, int B, int D, int H, int W, int Cin,
                                           int Cout, void* stream) {
    PULPO_REQUIRE(in && dy && scratch && (dw || accumulate == 2), "conv3d_k3_wgrad_bf16: null pointer");
    PULPO_REQUIRE_DT(dt, "conv3d_k3_wgrad_bf16");
    PULPO_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3d_k3_wgrad_bf16: bad dims");
    hipStream_t st = (hipStream_t)stream;
    WgradArgsH a;
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps; a.in_cs = in_cs;
    a.dy = dy; a.dy_bs = dy_bs; a.dy_ps = dy_ps; a.dy_cs = dy_cs;
    a.dwp = scratch;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.NPad = npad(Cout);
    a.ntz = pulpo::cdiv(D, WTZ); a.nty = pulpo::cdiv(H, TY); a.ntx = pulpo::cdiv(W, TX);
    a.ncit = pulpo::cdiv(Cin, CH); a.ncot = pulpo::cdiv(Cout, 32);
    const int ntile = B * a.ntz * a.nty * a.ntx;
    const int npair = a.ncit * a.ncot;
    // ONE resident workgroup per CU.  The kernel holds 242 registers per wave: two workgroups per CU (the round-1 choice: one stages while the
    // other multiplies) take 484 of a SIMD's 512, and no kernel of the main stream starts on a CU until a workgroup retires - every
    // BatchNorm-backward launch of the step then waited ~100 us for one (colsum_slices 109 instead of 8 us: 3 ms per bf16 step).  Measured
    // per 160^3 bf16 step for 512 / 448 / 384 / 320 / 256 / 192 / 128 workgroups: 23.1 / 20.3 / 20.3 / 20.2 / 20.3 / 21.0 / 24.4 ms.
    // PULPO_WGRAD_BF16_WGS overrides (A/B switch).
    static int wgs = -1;
    if (wgs < 0) { const char* e = getenv("PULPO_WGRAD_BF16_WGS"); wgs = e ? atoi(e) : 256; }
    a.nsplit = std::min(std::max(1, wgs / npair), ntile);
    const bool deferred = accumulate == 2;                 // see pulpo_conv3d_k3_wgrad
    if (!deferred) {
        hipError_t e = hipMemsetAsync(scratch, 0, pulpo_conv3d_k3_wgrad_scratch_floats(Cin, Cout) * sizeof(float), st);
        if (e != hipSuccess) return pulpo::fail((int)e, "wgrad_bf16 memset: %s", hipGetErrorString(e));
    }
    const int g = dt ? 8 : 4;             // channels per 16-byte piece
    const bool vec = (in_cs == 1) && (in_ps % g == 0) && (in_bs % g == 0) && (Cin % g == 0) && (((uintptr_t)in & 15) == 0) &&
                     (dy_cs == 1) && (dy_ps % g == 0) && (dy_bs % g == 0) && (Cout % g == 0) && (((uintptr_t)dy & 15) == 0);
    const int nrt_max = (27 * std::min(Cin, CH) + 31) / 32;
    const int ntw = (nrt_max + 3) / 4;
    constexpr size_t lds = (size_t)(WHV + WMV) * CP * sizeof(uint16_t);
    const int nblk = npair * a.nsplit;
    // transposing LDS reads where every 16-lane group's rows are 16 channels of one tap (PULPO_WGRAD_BF16_TR=0: the 2-byte gathers, A/B switch)
    static int tr_on = -1;
    if (tr_on < 0) { const char* e = getenv("PULPO_WGRAD_BF16_TR"); tr_on = e ? atoi(e) : 1; }
    const bool tr = tr_on && Cin % 16 == 0;
#define PULPO_WGRAD_H(NTWV, VECV, TT)                                                                                             \
    do {                                                                                                                            \
        if (tr) hipLaunchKernelGGL((conv3d_k3_wgrad_bf16<NTWV, VECV, TT, true>), dim3(nblk), dim3(256), lds, st, a);                \
        else hipLaunchKernelGGL((conv3d_k3_wgrad_bf16<NTWV, VECV, TT, false>), dim3(nblk), dim3(256), lds, st, a);                  \
    } while (0)
#define PULPO_WGRAD_T(TT)                                                                                                                           \
    do {                                                                                                                                            \
        if (vec) {                                                                                                                                  \
            if (ntw <= 1) PULPO_WGRAD_H(1, true, TT); else if (ntw <= 2) PULPO_WGRAD_H(2, true, TT); else if (ntw <= 4) PULPO_WGRAD_H(4, true, TT); \
            else PULPO_WGRAD_H(7, true, TT);                                                                                                        \
        } else {                                                                                                                                    \
            if (ntw <= 1) PULPO_WGRAD_H(1, false, TT); else if (ntw <= 2) PULPO_WGRAD_H(2, false, TT);                                              \
            else if (ntw <= 4) PULPO_WGRAD_H(4, false, TT); else PULPO_WGRAD_H(7, false, TT);                                                       \
        }                                                                                                                                           \
    } while (0)
    if (dt) PULPO_WGRAD_T(pulpo::bf16_t); else PULPO_WGRAD_T(float);
#undef PULPO_WGRAD_T
#undef PULPO_WGRAD_H
    int rc = pulpo::check_launch("conv3d_k3_wgrad_bf16");
    if (rc || deferred) return rc;
    return pulpo_conv::launch_unpack_wgrad(scratch, dw, Cin, Cout, accumulate, st);
}

PULPO_API int pulpo_conv3d_k3_wgrad_bf16(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dy, int64_t dy_bs,
                                         int64_t dy_ps, int64_t dy_cs, float* dw, int accumulate, float* scratch, int B, int D, int H, int W,
                                         int Cin, int Cout, void* stream) {
    return pulpo_conv3d_k3_wgrad_bf16_t(in, in_bs, in_ps, in_cs, dy, dy_bs, dy_ps, dy_cs, 0, dw, accumulate, scratch, B, D, H, W, Cin, Cout, stream);
}
