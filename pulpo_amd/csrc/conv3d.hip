// 3x3x3 / pad 1 / stride 1 convolution for PULPo's ConvUnit (reference: src/network_blocks.py:23), fp32,
// as an implicit GEMM on the CDNA4 matrix cores:  v_mfma_f32_32x32x2_f32 (exact fp32 = k-ordered fmaf chain).
//
//   forward / dgrad :  D[voxel][cout] += A[voxel][(tap,cin)] * B[(tap,cin)][cout]
//   wgrad           :  D[(tap,cin)][cout] += A[(tap,cin)][voxel] * B[voxel][cout]
//
// Data layout: activations are channels-last (N,D,H,W,C) with explicit batch/pixel/channel strides, so channel
// slices of a concatenation buffer and planar 1-3 channel volumes go through the same kernels.
// One workgroup = 256 threads = 4 waves (one per SIMD), 2 workgroups per CU.  A workgroup owns a 2x8x8 voxel tile;
// the (4x10x10)-voxel halo of a 32-channel input chunk is staged once in LDS ([voxel][CH+1], odd stride ->
// conflict-free ds_read_b32 for the A fragment) and re-used by all 27 taps; the 27 weight slabs stream through a
// double-buffered LDS tile, prefetched global->registers one tap ahead.
#include "common.h"

using f32x16 = __attribute__((ext_vector_type(16))) float;

namespace {

constexpr int TZ = 2, TY = 8, TX = 8, MV = TZ * TY * TX;          // output voxel tile
constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HV = HZ * HY * HX;  // halo tile

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
};

__device__ __forceinline__ int tap_halo_offset(int tap) {
    return ((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3;
}

// stage the halo tile of channels [c0, c0+CH) into xs[HV][CH+1]; zero outside the volume / beyond Cin
template <int CH, bool VEC>
__device__ __forceinline__ void stage_halo(float* xs, const float* __restrict__ in, long in_ps, long in_cs, int c0, int Cin,
                                           int z0, int y0, int x0, int D, int H, int W, int tid) {
    constexpr int CP = CH + 1;
    if constexpr (VEC) {
        constexpr int Q = CH / 4;
        for (int j = tid; j < HV * Q; j += 256) {
            const int hv = j / Q, q = j - hv * Q;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c0 + 4 * q < Cin)
                v = *reinterpret_cast<const float4*>(in + ((long)(gz * H + gy) * W + gx) * in_ps + c0 + 4 * q);
            float* d = xs + hv * CP + 4 * q;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
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

template <int CH, int NT, bool VEC>
__global__ __launch_bounds__(256, 2) void conv3d_k3_mfma(ConvArgs a) {
    constexpr int CP = CH + 1;
    constexpr int NN = NT / 32;
    constexpr int XS = (HV * CP + 3) & ~3;
    constexpr int WF4 = CH * NT / 4;               // float4 per weight slab
    constexpr int NW = (WF4 + 255) / 256;          // float4 per thread per slab
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
    const int z0 = tz_ * TZ, y0 = ty_ * TY, x0 = tx_ * TX;
    const int co0 = cot * NT;
    const int nchunk = (a.Cin + CH - 1) / CH;
    const int niter = nchunk * 27;
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
    const int v = wave * 32 + i;
    const int hb = ((v >> 6) * HY + ((v >> 3) & 7)) * HX + (v & 7);

    f32x16 acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

    load_w(0);
    int buf = 0, it = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        __syncthreads();   // every wave is done reading xs (previous chunk)
        stage_halo<CH, VEC>(xs, in_b, a.in_ps, a.in_cs, chunk * CH, a.Cin, z0, y0, x0, a.D, a.H, a.W, tid);
        for (int tap = 0; tap < 27; ++tap, ++it) {
            store_w(buf);
            __syncthreads();
            if (it + 1 < niter) load_w(it + 1);
            const float* xa = xs + (hb + tap_halo_offset(tap)) * CP + kk;
            const float* wb = ws + buf * CH * NT + kk * NT + i;
#pragma unroll
            for (int s = 0; s < CH / 2; ++s) {
                const float av = xa[2 * s];
#pragma unroll
                for (int n = 0; n < NN; ++n) {
                    const float bv = wb[2 * s * NT + n * 32];
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[n], 0, 0, 0);
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
        const float bv = (a.bias != nullptr && cok) ? a.bias[co] : 0.f;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            const int vv = wave * 32 + row;
            const int gz = z0 + (vv >> 6), gy = y0 + ((vv >> 3) & 7), gx = x0 + (vv & 7);
            if (cok && gz < a.D && gy < a.H && gx < a.W) {
                const float val = acc[n][r] + bv;
                out_b[((long)(gz * a.H + gy) * a.W + gx) * a.out_ps + (long)co * a.out_cs] = val;
                s += val;
                q += val * val;
            }
        }
        ssum[n] = s + __shfl_xor(s, 32, 64);
        ssq[n] = q + __shfl_xor(q, 32, 64);
    }
    if (a.stats != nullptr) {
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

constexpr int WG_CH = 32, WG_NT = 32, WG_CP = WG_CH + 1, WG_MAXT = 7;

template <bool VEC>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wgrad_mfma(WgradArgs a) {
    constexpr int XS = (HV * WG_CP + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* dys = smem + XS;       // [MV][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int npair = a.ncit * a.ncot;
    const int pair = lid % npair, split = lid / npair;
    const int cit = pair / a.ncot, cot = pair - cit * a.ncot;
    const int ci0 = cit * WG_CH, co0 = cot * WG_NT;
    const int Cc = min(WG_CH, a.Cin - ci0);
    const int rows = 27 * Cc;
    const int nrt = (rows + 31) >> 5;                 // row tiles of 32 (tap,ci) pairs
    const int i = lane & 31, kk = lane >> 5;

    int rowoff[WG_MAXT];
    bool rvalid[WG_MAXT];
#pragma unroll
    for (int u = 0; u < WG_MAXT; ++u) {
        const int r = 32 * (wave + 4 * u) + i;
        rvalid[u] = r < rows;
        const int tap = rvalid[u] ? r / Cc : 0, ci = rvalid[u] ? r - tap * Cc : 0;
        rowoff[u] = tap_halo_offset(tap) * WG_CP + ci;
    }
    f32x16 acc[WG_MAXT];
#pragma unroll
    for (int u = 0; u < WG_MAXT; ++u)
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
        const int z0 = tz_ * TZ, y0 = ty_ * TY, x0 = tx_ * TX;
        __syncthreads();
        stage_halo<WG_CH, VEC>(xs, a.in + (long)b * a.in_bs, a.in_ps, a.in_cs, ci0, a.Cin, z0, y0, x0, a.D, a.H, a.W, tid);
        {   // dY tile [MV][32]
            const float* dyb = a.dy + (long)b * a.dy_bs;
            if constexpr (VEC) {
                for (int j = tid; j < MV * 8; j += 256) {
                    const int vv = j >> 3, q = j & 7;
                    const int gz = z0 + (vv >> 6), gy = y0 + ((vv >> 3) & 7), gx = x0 + (vv & 7);
                    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gz < a.D && gy < a.H && gx < a.W && co0 + 4 * q < a.Cout)
                        val = *reinterpret_cast<const float4*>(dyb + ((long)(gz * a.H + gy) * a.W + gx) * a.dy_ps + co0 + 4 * q);
                    *reinterpret_cast<float4*>(dys + vv * WG_NT + 4 * q) = val;
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
        }
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < MV / 2; ++s) {
            const int vox = 2 * s + kk;
            const int hbk = (((vox >> 6) * HY + ((vox >> 3) & 7)) * HX + (vox & 7)) * WG_CP;
            const float bv = dys[vox * WG_NT + i];
#pragma unroll
            for (int u = 0; u < WG_MAXT; ++u) {
                if (wave + 4 * u < nrt) {
                    float av = xs[rowoff[u] + hbk];
                    av = rvalid[u] ? av : 0.f;
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u], 0, 0, 0);
                }
            }
        }
    }

    // flush: one 128-byte run of couts per (tap, ci) row -> float atomics at full rate
    const int co = co0 + i;
    if (co < a.Cout) {
#pragma unroll
        for (int u = 0; u < WG_MAXT; ++u) {
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

__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cin, int Cout, int NPad, long total) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(e % 27);
        const long r = e / 27;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        dw[e] = dwp[((long)tap * Cin + ci) * NPad + co];
    }
}

int pick_ch(int K) { return K <= 4 ? 4 : (K <= 16 ? 16 : 32); }
int npad(int N) { return (N + 63) & ~63; }

template <int CH, int NT, bool VEC>
int launch_conv(const ConvArgs& a, int nblk, hipStream_t st) {
    constexpr size_t lds = (size_t)(((HV * (CH + 1) + 3) & ~3) + 2 * CH * NT) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_mfma<CH, NT, VEC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(conv3d): %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3d_k3_mfma<CH, NT, VEC>), dim3(nblk), dim3(256), lds, st, a);
    return pulpo::check_launch("conv3d_k3_mfma");
}

}  // namespace

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
    const int NT = (N % 64 == 0 || N > 96) ? 64 : 32;
    return pick_ch(K) * 1000 + NT;
}

// Generic entry: computes out[b][vox][n] = sum_{tap,k} in[b][vox+tap-1][k] * wp[...] (+ bias[n]).
// K / N are the GEMM's reduction / output channel counts (forward: Cin/Cout; dgrad: Cout/Cin with dgrad-packed wp).
PULPO_API int pulpo_conv3d_k3_fwd(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias,
                                  float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats, int B, int D, int H, int W,
                                  int K, int N, void* stream) {
    PULPO_REQUIRE(in && wp && out, "conv3d_k3_fwd: null pointer");
    PULPO_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && K > 0 && N > 0, "conv3d_k3_fwd: bad dims");
    ConvArgs a;
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps; a.in_cs = in_cs;
    a.wp = wp; a.bias = bias;
    a.out = out; a.out_bs = out_bs; a.out_ps = out_ps; a.out_cs = out_cs;
    a.stats = stats;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = K; a.Cout = N; a.NPad = npad(N);
    a.ntz = pulpo::cdiv(D, TZ); a.nty = pulpo::cdiv(H, TY); a.ntx = pulpo::cdiv(W, TX);
    const int cfg = pulpo_conv3d_k3_tile_config(K, N);
    const int CH = cfg / 1000, NT = cfg % 1000;
    a.ncot = pulpo::cdiv(N, NT);
    const long nblk_l = (long)B * a.ntz * a.nty * a.ntx * a.ncot;
    PULPO_REQUIRE(nblk_l < (1L << 31), "conv3d_k3_fwd: grid too large");
    const int nblk = (int)nblk_l;
    const bool vec = (in_cs == 1) && (in_ps % 4 == 0) && (in_bs % 4 == 0) && (K % 4 == 0) && (((uintptr_t)in & 15) == 0) && CH >= 16;
    hipStream_t st = (hipStream_t)stream;
    if (CH == 4) return NT == 64 ? launch_conv<4, 64, false>(a, nblk, st) : launch_conv<4, 32, false>(a, nblk, st);
    if (CH == 16) {
        if (vec) return NT == 64 ? launch_conv<16, 64, true>(a, nblk, st) : launch_conv<16, 32, true>(a, nblk, st);
        return NT == 64 ? launch_conv<16, 64, false>(a, nblk, st) : launch_conv<16, 32, false>(a, nblk, st);
    }
    if (vec) return NT == 64 ? launch_conv<32, 64, true>(a, nblk, st) : launch_conv<32, 32, true>(a, nblk, st);
    return NT == 64 ? launch_conv<32, 64, false>(a, nblk, st) : launch_conv<32, 32, false>(a, nblk, st);
}

PULPO_API int pulpo_conv3d_k3_stat_tiles(int B, int D, int H, int W) {
    return B * pulpo::cdiv(D, TZ) * pulpo::cdiv(H, TY) * pulpo::cdiv(W, TX);
}

PULPO_API size_t pulpo_conv3d_k3_wgrad_scratch_floats(int Cin, int Cout) { return (size_t)27 * Cin * npad(Cout); }

// dw[Cout][Cin][27] = sum_vox in[vox+tap-1][ci] * dy[vox][co].   scratch: pulpo_conv3d_k3_wgrad_scratch_floats floats.
PULPO_API int pulpo_conv3d_k3_wgrad(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dy, int64_t dy_bs,
                                    int64_t dy_ps, int64_t dy_cs, float* dw, float* scratch, int B, int D, int H, int W, int Cin,
                                    int Cout, void* stream) {
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
    int nsplit = std::max(1, 1024 / npair);
    nsplit = std::min(nsplit, ntile);
    a.nsplit = nsplit;
    hipError_t e = hipMemsetAsync(scratch, 0, pulpo_conv3d_k3_wgrad_scratch_floats(Cin, Cout) * sizeof(float), st);
    if (e != hipSuccess) return pulpo::fail((int)e, "wgrad memset: %s", hipGetErrorString(e));
    const bool vec = (in_cs == 1) && (in_ps % 4 == 0) && (in_bs % 4 == 0) && (Cin % 4 == 0) && (((uintptr_t)in & 15) == 0) &&
                     (dy_cs == 1) && (dy_ps % 4 == 0) && (dy_bs % 4 == 0) && (Cout % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    constexpr size_t lds = (size_t)(((HV * WG_CP + 3) & ~3) + MV * WG_NT) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_mfma<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_mfma<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess || e2 != hipSuccess) return pulpo::fail((int)(e1 != hipSuccess ? e1 : e2), "hipFuncSetAttribute(wgrad)");
        attr_set = true;
    }
    const int nblk = npair * nsplit;
    if (vec) hipLaunchKernelGGL(conv3d_k3_wgrad_mfma<true>, dim3(nblk), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(conv3d_k3_wgrad_mfma<false>, dim3(nblk), dim3(256), lds, st, a);
    int rc = pulpo::check_launch("conv3d_k3_wgrad_mfma");
    if (rc) return rc;
    const long total = (long)Cout * Cin * 27;
    const int ub = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(ub), dim3(256), 0, st, scratch, dw, Cin, Cout, a.NPad, total);
    return pulpo::check_launch("unpack_wgrad");
}
