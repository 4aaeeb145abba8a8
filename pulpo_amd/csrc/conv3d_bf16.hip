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

#ifndef PULPO_PW_ABL
#define PULPO_PW_ABL 0                                  // diagnostic builds of the persistent kernel (scripts/build_variant.sh): 1 no output stores, 2 no halo
#endif                                                  // loads, 4 no MFMAs, 8 no halo LDS stores, 16 no weight loads / stores, 64 phase stamps
#if PULPO_PW_ABL & 64
// g_pw_stamps[block][8 k + p]: clock at phase p of the block's k-th tile (0 start, 1 halo in LDS, 2.. end of each weight group, then stores issued, statistics done)
__device__ unsigned long long g_pw_stamps[512 * 128];
#define PW_STAMP(k, p) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && (k) < 10) g_pw_stamps[blockIdx.x * 128 + 12 * (k) + (p)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
PULPO_API int pulpo_debug_read_stamps_pw(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_pw_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#else
#define PW_STAMP(k, p) do {} while (0)
#endif

#include <stdlib.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;

namespace {

using pulpo_conv::TY; using pulpo_conv::TX; using pulpo_conv::HY; using pulpo_conv::HX;
using pulpo_conv::npad;

// z extent of this kernel's voxel tile (its own policy: the 4x8x8 tile pays from 64^3 up; pulpo_conv3d_k3_fwd_bf16_stat_tiles follows it)
inline long conv_tz4_min_voxels() {
    static long v = -1;
    if (v < 0) { const char* e = getenv("PULPO_CONV_BF16_TZ4_MIN"); v = e ? atol(e) : 40L * 40 * 40; }     // (round 5: 40^3 - config 4 14.51 -> 14.38 ms per step; 64^3 before)
    return v;
}
inline int conv_tz(int D, int H, int W) { return (D % 4 == 0 && (long)D * H * W >= conv_tz4_min_voxels()) ? 4 : 2; }

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

// ------------------------------------------------------------------------------------------------ persistent forward / dgrad kernel
// The kernel above takes ONE tile per workgroup and a barrier per tap (4 MFMAs per wave between barriers on the 32-cout tiles, the tap's
// weights one L2 trip ahead): on the 32-channel 160^3 layers its matrix pipe was 23 % busy.  This variant, for bf16-stored activations with
// whole 32-channel chunks on 4 x 8 x 8 tiles, keeps the tiling and the MFMA loop but
//   * walks tiles with persistent workgroups (two per CU): the halo of the next chunk / tile travels to REGISTERS while this one is
//     multiplied, requested in slices after each weight group's loads so that vmcnt (in order) never makes a weight store wait for a
//     younger halo load, and only its LDS store stands between two tiles;
//   * stages the weights in GROUPS of TPB taps (9 = one dz plane on 32-cout tiles: 36 MFMAs per wave between barriers; 3 on 64-cout
//     tiles: 24), double-buffered, loaded a group ahead: 4 (10) barriers per chunk instead of 28;
//   * drops the LDS row padding for XOR swizzles (64-byte rows; halo rows keyed by their y, weight rows by n / 4: every ds_read_b128
//     lane group - 4 x 4 voxels of 4 consecutive y rows, or 16 rows n - lands on 16 distinct 16-byte slots), which is what lets two
//     workgroups share the CU's 160 KB.
constexpr int PW_HV = 6 * HY * HX;                      // halo voxels of a 4 x 8 x 8 tile
constexpr int PW_HP = (PW_HV * 4 + 255) / 256;          // 16-byte halo pieces per thread (the last one on the low threads only)
constexpr unsigned PW_OOB = 0x80000000u;

template <int NN, int TPB>
constexpr size_t pw_lds_bytes() { return (size_t)(PW_HV * CH + 2 * TPB * NN * 32 * CH) * sizeof(uint16_t) + 4 * 2 * NN * 32 * sizeof(float); }

template <int NN, int TPB>
__global__ __launch_bounds__(256, 2) void conv3d_k3_mfma_bf16_pw(ConvArgsH a) {
    using T = pulpo::bf16_t;
    constexpr int NT = NN * 32, MT = 2, NG = 27 / TPB;
    constexpr int WGE = TPB * NT * CH;                  // bf16 elements of one weight group
    constexpr int WPC = TPB * NT * 4;                   // its 16-byte pieces
    constexpr int WP = (WPC + 255) / 256;               // ... per thread
    static_assert(NG * TPB == 27, "taps per group must divide 27");
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_h[];
    uint16_t* xs = smem_h;                              // [PW_HV][32], slot s of row hv at s ^ (hy & 3)
    uint16_t* ws = smem_h + PW_HV * CH;                 // [2][TPB][NT][32], slot s of row n at s ^ ((n >> 2) & 3)
    float* red = reinterpret_cast<float*>(ws + 2 * WGE);// [4 waves][2][NT]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int nwork = a.B * a.ntz * a.nty * a.ntx * a.ncot, nwg = gridDim.x;
    const int nchunk = a.Cin / CH;

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

    // ---- halo pieces of this thread: piece j = (halo voxel j / 4, 8-channel slot j % 4)
    const unsigned ps_bytes = (unsigned)a.in_ps * 2u;
    const int in_bytes = (int)((long)a.D * a.H * a.W * a.in_ps * 2);
    unsigned hrel[PW_HP];                               // byte offset relative to the tile's halo origin
    unsigned hbit[PW_HP];                               // 1 << hz | 1 << (6 + hy) | 1 << (16 + hx); bit 31 where the thread has no piece
    int hlds[PW_HP];                                    // element offset in xs
#pragma unroll
    for (int u = 0; u < PW_HP; ++u) {
        const int j = tid + u * 256;
        const int hv = j >> 2, q = j & 3;
        const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
        hrel[u] = (unsigned)((hz * a.H + hy) * a.W + hx) * ps_bytes + q * 16u;
        hbit[u] = j < PW_HV * 4 ? (1u << hz) | (1u << (6 + hy)) | (1u << (16 + hx)) : 0x80000000u;
        hlds[u] = hv * CH + ((q ^ (hy & 3)) << 3);
    }
    uint4 hreg[PW_HP];
    if (PULPO_PW_ABL & 2) {
#pragma unroll
        for (int u = 0; u < PW_HP; ++u) hreg[u] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    }
    // which halo planes / rows / columns of a tile lie inside the volume, as one mask in hbit's layout (a piece is inside when all three of
    // its bits are set); outside pieces load from beyond num_records: zeros
    auto inside_mask = [&](const Tile& t) {
        auto run = [](int first, int extent, int n) {   // bits h in [0, n) with 0 <= first + h < extent
            const int lo = first < 0 ? -first : 0, hi = extent - first < n ? extent - first : n;
            return ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        };
        return run(t.z0 - 1, a.D, 6) | (run(t.y0 - 1, a.H, HY) << 6) | (run(t.x0 - 1, a.W, HX) << 16);
    };
    auto load_halo = [&](const Tile& t, int chunk, int u0, int u1) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(reinterpret_cast<const T*>(a.in) + (long)t.b * a.in_bs), 0, in_bytes, 0x00020000);
        const unsigned origin = (unsigned)(((t.z0 - 1) * a.H + (t.y0 - 1)) * a.W + (t.x0 - 1)) * ps_bytes;        // modulo 2^32
        const unsigned mask = inside_mask(t);
        const int c0b = chunk * CH * 2;
        if (PULPO_PW_ABL & 2) return;
#pragma unroll
        for (int u = u0; u < u1; ++u)
            hreg[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((mask & hbit[u]) == hbit[u] ? origin + hrel[u] : PW_OOB), c0b, 0));
    };
    auto store_halo = [&]() {
        if (PULPO_PW_ABL & 8) return;
#pragma unroll
        for (int u = 0; u < PW_HP; ++u)
            if (u + 1 < PW_HP || tid < PW_HV * 4 - (PW_HP - 1) * 256) *reinterpret_cast<uint4*>(xs + hlds[u]) = hreg[u];
    };

    // ---- weight pieces of this thread: piece j of a group = (tap j / (4 NT), row n, slot j % 4).  (Named registers: an array indexed inside
    //      the lambdas ends up in scratch.)
    uint4 wr0, wr1, wr2, wr3, wr4;
    wr0 = wr1 = wr2 = wr3 = wr4 = make_uint4(0, 0, 0, 0);
    static_assert(WP <= 5, "weight pieces per thread");
    auto w_has = [&](int u) { return u < WP && (u + 1 < WP || WPC % 256 == 0 || tid + u * 256 < WPC); };
    // a group's pieces sit at fixed per-thread byte offsets from its first row ((chunk, first tap, co0): the scalar offset of a buffer load)
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.wp), 0, -1, 0x00020000);
    auto w_voff = [&](int u) {
        const int j = tid + u * 256;
        const int tapl = j / (4 * NT), n = (j >> 2) % NT, q = j & 3;
        return ((tapl * a.NPad + n) * CH + q * 8) * 2;
    };
    const int wv0 = w_voff(0), wv1 = w_voff(1), wv2 = w_voff(2), wv3 = w_voff(3), wv4 = w_voff(4);
    auto w_dst = [&](int buf, int u) {
        const int j = tid + u * 256;
        const int tapl = j / (4 * NT), n = (j >> 2) % NT, q = j & 3;
        return reinterpret_cast<uint4*>(ws + buf * WGE + (tapl * NT + n) * CH + ((q ^ ((n >> 2) & 3)) << 3));
    };
    auto load_wg = [&](int co0, int chunk, int g) {
        if (PULPO_PW_ABL & 16) return;
        const int soff = ((chunk * 27 + g * TPB) * a.NPad + co0) * CH * 2;
        if (w_has(0)) wr0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wv0, soff, 0));
        if (w_has(1)) wr1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wv1, soff, 0));
        if (w_has(2)) wr2 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wv2, soff, 0));
        if (w_has(3)) wr3 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wv3, soff, 0));
        if (w_has(4)) wr4 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wv4, soff, 0));
    };
    auto store_wg = [&](int buf) {
        if (PULPO_PW_ABL & 16) return;
        if (w_has(0)) *w_dst(buf, 0) = wr0;
        if (w_has(1)) *w_dst(buf, 1) = wr1;
        if (w_has(2)) *w_dst(buf, 2) = wr2;
        if (w_has(3)) *w_dst(buf, 3) = wr3;
        if (w_has(4)) *w_dst(buf, 4) = wr4;
    };

    // ---- operand addresses (elements): A rows of the wave's two 32-voxel slabs per dy (the swizzle key is the halo row's y) and K half;
    //      the tap's halo offset is an immediate
    int xa[MT][3][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int v = (wave * MT + m) * 32 + i;
        const int vz = v >> 6, vy = (v >> 3) & 7, vx = v & 7;
        const int hb = (vz * HY + vy) * HX + vx;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xa[m][dy][ks] = hb * CH + (((ks * 2 + kk) ^ ((vy + dy) & 3)) << 3);
    }
    int wa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wa[ks] = i * CH + (((ks * 2 + kk) ^ ((i >> 2) & 3)) << 3);

    int work = pulpo::xcd_remap(blockIdx.x, nwg);
    Tile cur = describe(work);
    load_wg(cur.co0, 0, 0);
    load_halo(cur, 0, 0, PW_HP);
    store_wg(0);
    store_halo();
    int buf = 0;
    [[maybe_unused]] int tile_no = 0;
    int pend_tile = -1, pend_co0 = 0;                   // tile whose partial statistics wait in `red`
    auto flush_stats = [&]() {                          // (behind a barrier that followed the writes of `red`; the next writes are four barriers away)
        if (a.stats != nullptr && pend_tile >= 0 && tid < 2 * NT) {
            const int which = tid / NT, c = tid - which * NT;
            const float tot = red[(0 * 2 + which) * NT + c] + red[(1 * 2 + which) * NT + c] + red[(2 * 2 + which) * NT + c] +
                              red[(3 * 2 + which) * NT + c];
            a.stats[((long)pend_tile * 2 + which) * a.Cout + pend_co0 + c] = tot;
        }
    };

    for (;;) {
        PW_STAMP(tile_no, 0);
        const int next_work = work + nwg;
        const bool has_next = next_work < nwork;
        const Tile nxt = has_next ? describe(next_work) : cur;      // (after the last tile: its own halo again, into registers nobody stores)

        // the epilogue's per-channel constants, requested before any of the tile's loads: consuming them later waits for nothing younger
        // (a load at the epilogue would drain the halo slices still in flight)
        const bool fused = a.coef != nullptr;
        float ebias[NN], efsc[NN], efsh[NN];
#pragma unroll
        for (int n = 0; n < NN; ++n) {
            const int co = cur.co0 + n * 32 + i;
            ebias[n] = a.bias != nullptr ? a.bias[co] : 0.f;
            efsc[n] = fused ? a.coef[2 * a.Cout + co] : 1.f;
            efsh[n] = fused ? a.coef[3 * a.Cout + co] : 0.f;
        }
        f32x16 acc[MT][NN];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

        int chunk = 0;
        do {                                            // (at least one chunk: a zero-trip path would make the compiler's vmcnt bookkeeping drain
                                                        //  every load in flight at the epilogue's first use of a constant loaded above)
            __syncthreads();                            // the chunk's halo and its first weight group are in place
            PW_STAMP(tile_no, 1);
            if (chunk == 0) { flush_stats(); pend_tile = -1; }
            const bool lastc = chunk + 1 == nchunk;
            const Tile& ht = lastc ? nxt : cur;         // whose halo the registers take next
            const int hchunk = lastc ? 0 : chunk + 1;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                // the next group's weights, then this group's slice of the next halo
                {
                    const bool same = g + 1 < NG;
                    load_wg(same || !lastc ? cur.co0 : nxt.co0, same ? chunk : hchunk, same ? g + 1 : 0);
                    const int u0 = (PW_HP * g) / NG, u1 = (PW_HP * (g + 1)) / NG;
                    if (u1 > u0) load_halo(ht, hchunk, u0, u1);
                }
                const uint16_t* wb = ws + buf * WGE;
                constexpr int STEPS = TPB * 2;
                constexpr int SLOTS = 3;
                bf16x8 av[SLOTS][MT], bv[SLOTS][NN];
                auto fetch = [&](int st, int slot) {
                    const int d = st >> 1, ks = st & 1;
                    const int tap = g * TPB + d;
                    const int off = tap_halo_offset(tap) * CH;
#pragma unroll
                    for (int m = 0; m < MT; ++m) av[slot][m] = *reinterpret_cast<const bf16x8*>(xs + xa[m][(tap / 3) % 3][ks] + off);
#pragma unroll
                    for (int n = 0; n < NN; ++n) bv[slot][n] = *reinterpret_cast<const bf16x8*>(wb + wa[ks] + (d * NT + n * 32) * CH);
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
                        for (int m = 0; m < MT; ++m) {
                            if (PULPO_PW_ABL & 4) { acc[m][n][0] += (float)av[st % SLOTS][m][0] * (float)bv[st % SLOTS][n][0]; continue; }
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[st % SLOTS][m], bv[st % SLOTS][n], acc[m][n], 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                PW_STAMP(tile_no, 8 + g);
                store_wg(buf ^ 1);                      // (the other buffer was last read a group ago: every wave has passed that barrier)
                __syncthreads();
                PW_STAMP(tile_no, 2 + g);
                buf ^= 1;
            }
            // the registers' halo (the next chunk's, or the next tile's first) goes to LDS as soon as every wave has left this one: in front
            // of the epilogue, whose arithmetic and stores then cover the writes and the wait for the last slice
            store_halo();
            PW_STAMP(tile_no, 7);
        } while (++chunk < nchunk);

        // ---- epilogue: bias, optional eval-mode BatchNorm + LeakyReLU, rounding, stores, partial statistics of the tensor as stored.
        // A lane holds channel co of 16 voxels per 32-voxel slab (accumulator row r -> voxel x = (r & 3) + 4 kk, y = r >> 2); neighbouring lanes
        // hold neighbouring channels.  Rows are taken in pairs (x, x + 1): a lane exchanges one packed pair with its neighbour (DPP) so that
        // the even lane owns (co, co + 1) of voxel x and the odd lane (co - 1, co) of voxel x + 1 - one dword each.  The slab's dwords pass
        // through a wave-private 2 KB image [32 voxels][32 channels] in the weight buffer that is idle until the next tile's second group
        // (DS operations of one wave execute in order: no wait between its writes and reads), and leave as 16-byte pieces: two
        // buffer_store_dwordx4 per slab (16 voxels x 64 bytes each) instead of sixteen dword stores.
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T*>(a.out) + (long)cur.b * a.out_bs, 0,
                                                                             (int)((long)a.D * a.H * a.W * a.out_ps * 2), 0x00020000);
        const unsigned ops_b = (unsigned)a.out_ps * 2u;
        const unsigned row_b = (unsigned)a.W * ops_b;
        const unsigned obase = (unsigned)(((cur.z0 + wave) * a.H + cur.y0) * a.W + cur.x0) * ops_b + (unsigned)cur.co0 * 2u;
        const bool odd = i & 1;
        // v_perm_b32 selector of the owned dword from (neighbour's pair, own pair): even lane (own x, neighbour's x), odd lane (neighbour's
        // x + 1, own x + 1) - bytes 0-3 of the operand pair are the own dword, 4-7 the neighbour's
        const uint32_t psel = odd ? 0x03020706u : 0x05040100u;
        uint16_t* xch = ws + (buf ^ 1) * WGE + wave * (32 * 32);                   // (after the toggle `buf` holds the next tile's first group)
        uint32_t* xw = reinterpret_cast<uint32_t*>(xch + (4 * kk + (odd ? 1 : 0)) * 32 + (i - (odd ? 1 : 0)));
        const uint4* xr = reinterpret_cast<const uint4*>(xch) + lane;              // piece `lane` (+ 64): voxel row lane / 4 (+ 16), channels 8 (lane % 4) ..
        const unsigned st_off = obase + (unsigned)(lane >> 5) * row_b + (unsigned)((lane >> 2) & 7) * ops_b + (unsigned)(lane & 3) * 16u;
        float ssum[NN], ssq[NN];
#pragma unroll
        for (int n = 0; n < NN; ++n) {
            const float bias = ebias[n], fsc = efsc[n], fsh = efsh[n];
            using f32x2 = __attribute__((ext_vector_type(2))) float;
            f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    float v0 = acc[m][n][r] + bias, v1 = acc[m][n][r + 1] + bias;
                    if (fused) {
                        const float t0 = pulpo::as_stored<T>(v0) * fsc + fsh, t1 = pulpo::as_stored<T>(v1) * fsc + fsh;
                        v0 = t0 > 0.f ? t0 : t0 * a.slope;
                        v1 = t1 > 0.f ? t1 : t1 * a.slope;
                    }
                    const uint32_t h = pulpo::pack_bf2(v0, v1);                    // (x, x + 1) of this lane's channel, rounded
                    const f32x2 rr = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
                    s2 += rr;
                    q2 = __builtin_elementwise_fma(rr, rr, q2);
                    const uint32_t nbh = (uint32_t)__builtin_amdgcn_mov_dpp((int)h, 0xB1, 0xF, 0xF, true);         // quad_perm [1, 0, 3, 2]
                    xw[((r & 3) + 8 * (r >> 2)) * 16] = __builtin_amdgcn_perm(nbh, h, psel);
                }
                const uint4 p0 = xr[0], p1 = xr[64];
                if (!(PULPO_PW_ABL & 1) || p0.x == 0x12345678u) {
                    const unsigned soff = (unsigned)(m * 4) * row_b + (unsigned)n * 64u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, p0), ors, (int)st_off, (int)soff, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, p1), ors, (int)st_off, (int)(soff + 2 * row_b), 0);
                }
                if (n == 0 && m == 0) PW_STAMP(tile_no, 11);
            }
            const float s = s2.x + s2.y, q = q2.x + q2.y;
            ssum[n] = s + __shfl_xor(s, 32, 64);
            ssq[n] = q + __shfl_xor(q, 32, 64);
        }
        PW_STAMP(tile_no, 5);
        // the waves' partial sums go to LDS; they are added and written behind the NEXT barrier every wave passes anyway (the next tile's
        // "halo in place", or the one after the loop) - no barrier of their own
        if (a.stats != nullptr && lane < 32) {
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                red[(wave * 2 + 0) * NT + n * 32 + i] = ssum[n];
                red[(wave * 2 + 1) * NT + n * 32 + i] = ssq[n];
            }
        }
        pend_tile = cur.tile_lin; pend_co0 = cur.co0;
        PW_STAMP(tile_no, 6);
        ++tile_no;
        if (!has_next) break;
        work = next_work;
        cur = nxt;
    }
    __syncthreads();
    flush_stats();
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
    long split_stride;            // 0, or (deterministic mode) floats between the per-split copies of dwp (see WgradArgs in conv3d_wgrad.hip)
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

// PF (bf16-stored operands in 16-byte pieces, tensors below 2 GB): the NEXT tile's halo and dy pieces travel to registers (buffer loads; pieces
// outside the volume or beyond the channel count read zeros from beyond num_records) while this tile is multiplied, so that only their LDS
// stores and two barriers stand between the matrix loops of consecutive tiles - with one resident workgroup per CU nothing else hides a
// staging phase.
template <int NTW, bool VEC, typename T = float, bool TR = false, bool PF = false>
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

    // ---- (PF) this thread's pieces: halo piece j = (halo voxel j / 4, 8-channel slot j % 4), dy piece likewise over the tile's voxels
    constexpr int NHP = (WHV * 4 + 255) / 256, NDP = (WMV * 4) / 256;
    constexpr unsigned OOB = 0x80000000u;
    [[maybe_unused]] uint4 hr[NHP], dr[NDP];
    [[maybe_unused]] unsigned hrel[NHP], hbit[NHP], drel[NDP];
    [[maybe_unused]] const unsigned ips_b = (unsigned)a.in_ps * 2u, dps_b = (unsigned)a.dy_ps * 2u;
    if constexpr (PF) {
#pragma unroll
        for (int u = 0; u < NHP; ++u) {
            const int j = tid + u * 256;
            const int hv = j >> 2, q = j & 3;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            hrel[u] = (unsigned)((hz * a.H + hy) * a.W + hx) * ips_b + (unsigned)(ci0 + 8 * q) * 2u;
            hbit[u] = (j < WHV * 4 && ci0 + 8 * q < a.Cin) ? (1u << hz) | (1u << (4 + hy)) | (1u << (14 + hx)) : 0x80000000u;
        }
#pragma unroll
        for (int u = 0; u < NDP; ++u) {
            const int j = tid + u * 256;
            const int v = j >> 2, q = j & 3;
            drel[u] = co0 + 8 * q < a.Cout ? (unsigned)(((v >> 6) * a.H + ((v >> 3) & 7)) * a.W + (v & 7)) * dps_b + (unsigned)(co0 + 8 * q) * 2u : OOB;
        }
    }
    auto load_tile = [&](int tl) {
        int t = tl;
        const int tx_ = t % a.ntx; t /= a.ntx;
        const int ty_ = t % a.nty; t /= a.nty;
        const int tz_ = t % a.ntz;
        const int b = t / a.ntz;
        const int z0 = tz_ * WTZ, y0 = ty_ * TY, x0 = tx_ * TX;
        const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(reinterpret_cast<const T*>(a.in) + (long)b * a.in_bs), 0, (int)((long)a.D * a.H * a.W * a.in_ps * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(reinterpret_cast<const T*>(a.dy) + (long)b * a.dy_bs), 0, (int)((long)a.D * a.H * a.W * a.dy_ps * 2), 0x00020000);
        auto run = [](int first, int extent, int n) {   // bits h in [0, n) with 0 <= first + h < extent
            const int lo = first < 0 ? -first : 0, hi = extent - first < n ? extent - first : n;
            return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
        };
        const unsigned mask = run(z0 - 1, a.D, WTZ + 2) | (run(y0 - 1, a.H, HY) << 4) | (run(x0 - 1, a.W, HX) << 14);
        const unsigned origin = (unsigned)(((z0 - 1) * a.H + (y0 - 1)) * a.W + (x0 - 1)) * ips_b;            // modulo 2^32
#pragma unroll
        for (int u = 0; u < NHP; ++u)
            hr[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(irs, (int)((mask & hbit[u]) == hbit[u] ? origin + hrel[u] : OOB), 0, 0));
        // dy: the tile itself (whole in x and y by the host's condition; ragged only in z, where the rows lie beyond num_records)
        const unsigned dorigin = (unsigned)((z0 * a.H + y0) * a.W + x0) * dps_b;
#pragma unroll
        for (int u = 0; u < NDP; ++u)
            dr[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(drs, (int)(drel[u] == OOB ? OOB : dorigin + drel[u]), 0, 0));
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < NHP; ++u) {
            const int j = tid + u * 256;
            if (u + 1 < NHP || j < WHV * 4) *reinterpret_cast<uint4*>(xs + (j >> 2) * CP + 8 * (j & 3)) = hr[u];
        }
#pragma unroll
        for (int u = 0; u < NDP; ++u) {
            const int j = tid + u * 256;
            *reinterpret_cast<uint4*>(dys + (j >> 2) * CP + 8 * (j & 3)) = dr[u];
        }
    };
    if constexpr (PF) { if (t_begin < t_end) load_tile(t_begin); }

    for (int tl = t_begin; tl < t_end; ++tl) {
        if constexpr (PF) {
            __syncthreads();
            store_tile();
            __syncthreads();
            load_tile(min(tl + 1, t_end - 1));          // (after the last tile: its own pieces again, into registers nobody stores)
        } else {
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
        }
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
                        atomicAdd(a.dwp + (long)split * a.split_stride + ((long)tap * a.Cin + ci0 + ci) * a.NPad + co, acc[u][r]);
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

template <int NN, int TPB>
int launch_bf16_pw(const ConvArgsH& a, long nwork, hipStream_t st) {
    constexpr size_t lds = pw_lds_bytes<NN, TPB>();
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_mfma_bf16_pw<NN, TPB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(conv3d bf16 pw): %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int nwg = (int)std::min<long>(nwork, 512);    // persistent workgroups: two per CU
    hipLaunchKernelGGL((conv3d_k3_mfma_bf16_pw<NN, TPB>), dim3(nwg), dim3(256), lds, st, a);
    return pulpo::check_launch("conv3d_k3_mfma_bf16_pw");
}

// PULPO_CONV_BF16_PW (A/B switch): 0 = never, 1 = the 32-cout tiles only, 2 (default) = the 64-cout tiles as well.
// Measured on one MI355X (scripts/pw_ab.sh), modes 0 / 1 / 2: config 4 step 61.7 / 65.4 / 67.3 pairs/s, config 5 step 38.1 / 40.4 / 41.6,
// config 4 inference 190 / 211 / 222.
int bf16_pw_mode() {
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("PULPO_CONV_BF16_PW"); mode = e ? atoi(e) : 2; }
    return mode;
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
    // the persistent kernel: bf16-stored activations in 16-byte pieces, whole 32-channel chunks, whole 4 x 8 x 8 tiles, whole cout tiles
    const bool pw_ok = dt == 1 && vec && tz == 4 && a.ksplit == 1 && K % CH == 0 && N % NT == 0 && D % 4 == 0 && H % TY == 0 && W % TX == 0 &&
                       (long)D * H * W * in_ps * 2 < (1L << 31) && out_cs == 1 && (long)D * H * W * out_ps * 2 < (1L << 31) &&
                       out_ps % 8 == 0 && out_bs % 8 == 0 && (((uintptr_t)out & 15) == 0) &&
                       bf16_pw_mode() >= (NT == 64 ? 2 : 1);
    if (pw_ok) return NT == 64 ? launch_bf16_pw<2, 3>(a, nblk_l, st) : launch_bf16_pw<1, 9>(a, nblk_l, st);
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

static int wgrad_bf16_impl(const void* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const void* dy, int64_t dy_bs, int64_t dy_ps,
                           int64_t dy_cs, int dt, float* dw, int accumulate, float* scratch, float* slabs, int nslab, int B, int D, int H, int W, int Cin,
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
    if (slabs) a.nsplit = std::min(a.nsplit, nslab);
    const bool deferred = accumulate == 2;                 // see pulpo_conv3d_k3_wgrad
    if (!deferred) {
        hipError_t e = hipMemsetAsync(scratch, 0, pulpo_conv3d_k3_wgrad_scratch_floats(Cin, Cout) * sizeof(float), st);
        if (e != hipSuccess) return pulpo::fail((int)e, "wgrad_bf16 memset: %s", hipGetErrorString(e));
    }
    const size_t base = pulpo_conv3d_k3_wgrad_scratch_floats(Cin, Cout);
    a.split_stride = 0;
    if (slabs) {                                           // deterministic mode, see pulpo_conv3d_k3_wgrad_det
        a.dwp = slabs; a.split_stride = (long)base;
        hipError_t e = hipMemsetAsync(slabs, 0, (size_t)a.nsplit * base * sizeof(float), st);
        if (e != hipSuccess) return pulpo::fail((int)e, "wgrad_bf16 slab memset: %s", hipGetErrorString(e));
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
    // register-prefetched staging (PULPO_WGRAD_BF16_PF=0: the staged form, A/B switch): bf16 tensors in 16-byte pieces below 2 GB, tiles whole in x / y
    static int pf_on = -1;
    if (pf_on < 0) { const char* e = getenv("PULPO_WGRAD_BF16_PF"); pf_on = e ? atoi(e) : 1; }
    const bool pf = pf_on && tr && dt == 1 && vec && H % TY == 0 && W % TX == 0 && (long)D * H * W * in_ps * 2 < (1L << 31) &&
                    (long)D * H * W * dy_ps * 2 < (1L << 31);
#define PULPO_WGRAD_H(NTWV, VECV, TT)                                                                                             \
    do {                                                                                                                            \
        if constexpr (VECV && sizeof(TT) == 2) {                                                                                    \
            if (pf) { hipLaunchKernelGGL((conv3d_k3_wgrad_bf16<NTWV, VECV, TT, true, true>), dim3(nblk), dim3(256), lds, st, a); break; } \
        }                                                                                                                           \
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
    if (rc == 0 && slabs) rc = pulpo_conv::launch_wgrad_slab_reduce(scratch, slabs, a.nsplit, (long)base, st, npad(Cout), Cout);
    if (rc || deferred) return rc;
    return pulpo_conv::launch_unpack_wgrad(scratch, dw, Cin, Cout, accumulate, st);
}

PULPO_API int pulpo_conv3d_k3_wgrad_bf16_t(const void* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const void* dy, int64_t dy_bs, int64_t dy_ps,
                                           int64_t dy_cs, int dt, float* dw, int accumulate, float* scratch, int B, int D, int H, int W, int Cin,
                                           int Cout, void* stream) {
    return wgrad_bf16_impl(in, in_bs, in_ps, in_cs, dy, dy_bs, dy_ps, dy_cs, dt, dw, accumulate, scratch, nullptr, 0, B, D, H, W, Cin, Cout, stream);
}

// deterministic form of the bf16-operand weight gradient (see pulpo_conv3d_k3_wgrad_det; same slab count query)
PULPO_API int pulpo_conv3d_k3_wgrad_bf16_det_t(const void* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const void* dy, int64_t dy_bs,
                                               int64_t dy_ps, int64_t dy_cs, int dt, float* dw, int accumulate, float* scratch, float* slabs,
                                               int nslab, int B, int D, int H, int W, int Cin, int Cout, void* stream) {
    PULPO_REQUIRE(slabs && nslab >= 1, "conv3d_k3_wgrad_bf16_det: slabs of nslab >= 1 copies of the packed scratch required");
    return wgrad_bf16_impl(in, in_bs, in_ps, in_cs, dy, dy_bs, dy_ps, dy_cs, dt, dw, accumulate, scratch, slabs, nslab, B, D, H, W, Cin, Cout, stream);
}

PULPO_API int pulpo_conv3d_k3_wgrad_bf16(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dy, int64_t dy_bs,
                                         int64_t dy_ps, int64_t dy_cs, float* dw, int accumulate, float* scratch, int B, int D, int H, int W,
                                         int Cin, int Cout, void* stream) {
    return pulpo_conv3d_k3_wgrad_bf16_t(in, in_bs, in_ps, in_cs, dy, dy_bs, dy_ps, dy_cs, 0, dw, accumulate, scratch, B, D, H, W, Cin, Cout, stream);
}
