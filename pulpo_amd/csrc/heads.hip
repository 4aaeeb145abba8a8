// 1x1x1 channel-mixing heads C -> 3 (or C -> 3+3):
//   MuSigmaBlock + gauss_sampler  (reference: src/network_blocks.py:49-60, :7-8):  mu = Wm h + bm ;
//        sigma = softplus(Ws h + bs) ; z = mu + sigma * eps
//   VelocityField's last layer    (reference: src/network_blocks.py:81):           v = W h + b
// Input h is channels-last [pixel][C]; outputs are planar (B, 3, D*H*W) like the reference's tensors.
// HBM-bound (reads C floats, writes 3-9 per voxel): 8 lanes share one voxel, each lane streams float4 channel
// slices (one 128-byte line per voxel per step), partial dot products are combined with wave shuffles.
#include "act_io.h"

namespace {

constexpr int G = 8;   // lanes per voxel

__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// NOUT = 3: plain head.  NOUT = 6: rows 0-2 mu, rows 3-5 sigma pre-activation.
template <int NOUT, bool VEC, typename TH = float>
__global__ __launch_bounds__(256) void heads_fwd_kernel(const TH* __restrict__ h, long ps, const float* __restrict__ Wt,
                                                          const float* __restrict__ bias, const float* __restrict__ eps, float* __restrict__ o0,
                                                          float* __restrict__ o1, float* __restrict__ o2, int B, long V, int C) {
    extern __shared__ float wl[];              // [NOUT][C]
    for (int j = threadIdx.x; j < NOUT * C; j += blockDim.x) wl[j] = Wt[j];
    __syncthreads();
    const int g = threadIdx.x & (G - 1);
    const long npix = (long)B * V;
    const long pstep = (long)gridDim.x * (blockDim.x / G);
    for (long p0 = (long)blockIdx.x * (blockDim.x / G); p0 < npix; p0 += pstep) {   // uniform trip count per block
        const long p = p0 + threadIdx.x / G;
        const bool live = p < npix;
        float acc[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) acc[j] = 0.f;
        if (live) {
            const TH* hp = h + p * ps;
            if constexpr (VEC) {
                for (int c = 4 * g; c < C; c += 4 * G) {
                    float x[4];
                    pulpo::ldv<4>(hp + c, x);
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) {
                        const float* w = wl + j * C + c;
                        acc[j] += x[0] * w[0] + x[1] * w[1] + x[2] * w[2] + x[3] * w[3];
                    }
                }
            } else {
                for (int c = g; c < C; c += G) {
                    float x1[1];
                    pulpo::ldv<1>(hp + c, x1);
                    const float x = x1[0];
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) acc[j] += x * wl[j * C + c];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NOUT; ++j) acc[j] = group_sum(acc[j]);
        // every lane of the group holds the sums (xor shuffles): lane g < 3 finishes output component g - one load / store instruction per output
        // tensor and trip instead of three from the group's first lane (the kernel ran at what the vector-memory unit issues, not at what HBM delivers)
        if (live && g < 3) {
            const long b = p / V, v = p - b * V;
            const long at = b * 3 * V + v + g * V;
            const float a_lo = g == 0 ? acc[0] : g == 1 ? acc[1] : acc[2];
            if constexpr (NOUT == 3) {
                o0[at] = a_lo + bias[g];
            } else {
                const float a_hi = g == 0 ? acc[3] : g == 1 ? acc[4] : acc[5];
                const float mu = a_lo + bias[g];
                const float sg = softplus_f(a_hi + bias[3 + g]);
                o0[at] = mu;
                o1[at] = sg;
                o2[at] = eps != nullptr ? mu + sg * eps[at] : mu;
            }
        }
    }
}

// backward.  dpre[j] (j < NOUT) per voxel is formed from the upstream gradients:
//   NOUT == 3 : dpre = g0
//   NOUT == 6 : dmu = g0 + g2 ; dsigma = g1 + g2 * eps ; dpre[3+j] = dsigma * (1 - exp(-sigma))   (softplus' = sigmoid)
// outputs: dh[pixel][C] ;  partial[blk][NOUT*C + NOUT] = per-block sums for dW and db.
// Thread (col,row) owns VEC channels and walks the block's pixels, so dW accumulates in NOUT*VEC registers.
template <int NOUT, int VEC, typename TH = float>
__global__ __launch_bounds__(256) void heads_bwd_kernel(const TH* __restrict__ h, long ps, const float* __restrict__ Wt,
                                                          const float* __restrict__ g0, const float* __restrict__ g1, const float* __restrict__ g2,
                                                          const float* __restrict__ eps, const float* __restrict__ sigma, TH* __restrict__ dh,
                                                          long dps, float* __restrict__ partial, int B, long V, int C) {
    extern __shared__ float red[];             // [RB][NOUT*C + NOUT]
    const int CV = C / VEC, RB = blockDim.x / CV;
    const int col = threadIdx.x % CV, row = threadIdx.x / CV;
    const int c = col * VEC;
    const int ROWLEN = NOUT * C + NOUT;
    const long npix = (long)B * V;
    float dw[NOUT][VEC], db[NOUT], w[NOUT][VEC];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        db[j] = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { dw[j][k] = 0.f; w[j][k] = (row < RB) ? Wt[j * C + c + k] : 0.f; }
    }
    if constexpr (NOUT == 6) {
        // The five planar 3-channel operands of a pixel (upstream gradients of mu / sigma / z, the noise, sigma) are the same 15 values for every
        // thread of the pixel's row.  Loaded by each thread they were 15 of the 16 load instructions a wave issued per trip, and the kernel ran at
        // what the vector-memory unit issues (2.3 TB/s at 96 channels) instead of what HBM delivers: the row's first threads now fetch them once,
        // the row reads them from LDS (two buffers by trip parity: one barrier per trip).  Uniform trip count per block for that barrier.
        float (*sc)[16] = reinterpret_cast<float (*)[16]>(red + (size_t)RB * ROWLEN);       // [2 * RB][16], behind the reduction rows (sized by the launcher)
        int it = 0;
        for (long pb = (long)blockIdx.x * RB; pb < npix; pb += (long)gridDim.x * RB, it ^= 1) {
            const long p = pb + row;
            const bool live = row < RB && p < npix;
            const long pc = live ? p : 0;
            const long b = pc / V, v = pc - b * V;
            const long base = b * 3 * V + v;
            if (live)
                for (int q = col; q < 15; q += CV) {
                    const int arr = q / 3, j = q - 3 * arr;
                    const float* src = arr == 0 ? g0 : arr == 1 ? g1 : arr == 2 ? g2 : arr == 3 ? eps : sigma;
                    sc[it * RB + row][q] = src != nullptr ? src[base + j * V] : 0.f;
                }
            float x[VEC];
            if (live) pulpo::ldv<VEC>(h + p * ps + c, x);
            __syncthreads();
            if (live) {
                const float* sv = sc[it * RB + row];
                float dpre[6];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float gz = sv[6 + j];
                    dpre[j] = sv[j] + gz;                                      // dmu = g0 + g2
                    const float gs = sv[3 + j] + gz * sv[9 + j];               // dsigma = g1 + g2 * eps (eps absent: stored as 0)
                    dpre[3 + j] = gs * (1.f - expf(-sv[12 + j]));
                }
                float o[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = 0.f;
#pragma unroll
                for (int j = 0; j < NOUT; ++j) {
                    db[j] += dpre[j];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) { o[k] += dpre[j] * w[j][k]; dw[j][k] += dpre[j] * x[k]; }
                }
                pulpo::stv<VEC>(dh + p * dps + c, o);
            }
        }
    } else if (row < RB) {
        for (long p = (long)blockIdx.x * RB + row; p < npix; p += (long)gridDim.x * RB) {
            const long b = p / V, v = p - b * V;
            const long base = b * 3 * V + v;
            float dpre[NOUT];
#pragma unroll
            for (int j = 0; j < 3; ++j) dpre[j] = g0[base + j * V];
            float x[VEC], o[VEC];
            pulpo::ldv<VEC>(h + p * ps + c, x);
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = 0.f;
#pragma unroll
            for (int j = 0; j < NOUT; ++j) {
                db[j] += dpre[j];
#pragma unroll
                for (int k = 0; k < VEC; ++k) { o[k] += dpre[j] * w[j][k]; dw[j][k] += dpre[j] * x[k]; }
            }
            pulpo::stv<VEC>(dh + p * dps + c, o);
        }
    }
    if (row < RB) {
#pragma unroll
        for (int j = 0; j < NOUT; ++j) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) red[row * ROWLEN + j * C + c + k] = dw[j][k];
            if (col == 0) red[row * ROWLEN + NOUT * C + j] = db[j];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < ROWLEN; j += blockDim.x) {
        float t = 0.f;
        for (int r = 0; r < RB; ++r) t += red[r * ROWLEN + j];
        partial[(long)blockIdx.x * ROWLEN + j] = t;
    }
}

inline int heads_blocks(long npix) { return (int)std::max<long>(1, std::min<long>((npix + 31) / 32, 1024)); }

}  // namespace

// Wt: [NOUT][C] (rows 0-2 = first conv, rows 3-5 = second conv for NOUT == 6); bias: [NOUT].  h_dt: dtype code of h (0 fp32, 1 bf16;
// stride in elements); the outputs are planar fp32
PULPO_API int pulpo_heads_fwd_t(const void* h, int h_dt, int64_t ps, const float* Wt, const float* bias, const float* eps, float* o0, float* o1,
                                float* o2, int nout, int B, int64_t V, int C, void* stream) {
    PULPO_REQUIRE(h && Wt && bias && o0 && B > 0 && V > 0 && C > 0, "heads_fwd: bad arguments");
    PULPO_REQUIRE(nout == 3 || (nout == 6 && o1 && o2), "heads_fwd: nout must be 3 or 6");
    PULPO_REQUIRE_DT(h_dt, "heads_fwd");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = C % 4 == 0 && ps % 4 == 0 && (((uintptr_t)h) % (h_dt ? 8 : 16)) == 0;
    const int nblk = heads_blocks((long)B * V);
    const size_t lds = (size_t)nout * C * sizeof(float);
    PULPO_DISPATCH_DT(h_dt, TH, {
        const TH* hp = (const TH*)h;
        if (nout == 3) {
            if (vec) hipLaunchKernelGGL((heads_fwd_kernel<3, true, TH>), dim3(nblk), dim3(256), lds, st, hp, ps, Wt, bias, eps, o0, o1, o2, B, V, C);
            else hipLaunchKernelGGL((heads_fwd_kernel<3, false, TH>), dim3(nblk), dim3(256), lds, st, hp, ps, Wt, bias, eps, o0, o1, o2, B, V, C);
        } else {
            if (vec) hipLaunchKernelGGL((heads_fwd_kernel<6, true, TH>), dim3(nblk), dim3(256), lds, st, hp, ps, Wt, bias, eps, o0, o1, o2, B, V, C);
            else hipLaunchKernelGGL((heads_fwd_kernel<6, false, TH>), dim3(nblk), dim3(256), lds, st, hp, ps, Wt, bias, eps, o0, o1, o2, B, V, C);
        }
    });
    return pulpo::check_launch("heads_fwd");
}

PULPO_API int pulpo_heads_fwd(const float* h, int64_t ps, const float* Wt, const float* bias, const float* eps, float* o0, float* o1,
                              float* o2, int nout, int B, int64_t V, int C, void* stream) {
    return pulpo_heads_fwd_t(h, 0, ps, Wt, bias, eps, o0, o1, o2, nout, B, V, C, stream);
}

PULPO_API int pulpo_heads_bwd_blocks(int B, int64_t V, int C) {
    const int vec = (C % 4 == 0) ? 4 : 1;
    const int RB = std::max(1, 256 / (C / vec));
    const long npix = (long)B * V;
    return (int)std::max<long>(1, std::min<long>((npix + RB * 8 - 1) / (RB * 8), 1024));
}

// partial: [pulpo_heads_bwd_blocks][nout*C + nout]; reduce with pulpo_colsum -> (dW[nout][C] | db[nout]).  h and dh share the dtype h_dt.
PULPO_API int pulpo_heads_bwd_t(const void* h, int h_dt, int64_t ps, const float* Wt, const float* g0, const float* g1, const float* g2,
                                const float* eps, const float* sigma, void* dh, int64_t dps, float* partial, int nout, int B, int64_t V, int C,
                                void* stream) {
    PULPO_REQUIRE(h && Wt && dh && partial && B > 0 && V > 0 && C > 0, "heads_bwd: bad arguments");
    PULPO_REQUIRE((nout == 3 && g0) || (nout == 6 && sigma), "heads_bwd: nout must be 3 (with g0) or 6 (with sigma)");
    PULPO_REQUIRE_DT(h_dt, "heads_bwd");
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = C % 4 == 0;
    if (v4) PULPO_REQUIRE(ps % 4 == 0 && dps % 4 == 0 && ((((uintptr_t)h) | ((uintptr_t)dh)) % (h_dt ? 8 : 16)) == 0, "heads_bwd: unaligned operands");
    PULPO_REQUIRE(C / (v4 ? 4 : 1) <= 256, "heads_bwd: too many channels");
    const int nblk = pulpo_heads_bwd_blocks(B, V, C);
    const int RB = std::max(1, 256 / (C / (v4 ? 4 : 1)));
    const size_t lds = ((size_t)RB * (nout * C + nout) + (nout == 6 ? (size_t)2 * RB * 16 : 0)) * sizeof(float);       // reduction rows + (nout 6) the per-pixel operand rows
    PULPO_REQUIRE(lds <= 64 * 1024, "heads_bwd: LDS budget exceeded");
    PULPO_DISPATCH_DT(h_dt, TH, {
        const TH* hp = (const TH*)h;
        TH* dhp = (TH*)dh;
        if (nout == 3) {
            if (v4) hipLaunchKernelGGL((heads_bwd_kernel<3, 4, TH>), dim3(nblk), dim3(256), lds, st, hp, ps, Wt, g0, g1, g2, eps, sigma, dhp, dps, partial, B, V, C);
            else hipLaunchKernelGGL((heads_bwd_kernel<3, 1, TH>), dim3(nblk), dim3(256), lds, st, hp, ps, Wt, g0, g1, g2, eps, sigma, dhp, dps, partial, B, V, C);
        } else {
            if (v4) hipLaunchKernelGGL((heads_bwd_kernel<6, 4, TH>), dim3(nblk), dim3(256), lds, st, hp, ps, Wt, g0, g1, g2, eps, sigma, dhp, dps, partial, B, V, C);
            else hipLaunchKernelGGL((heads_bwd_kernel<6, 1, TH>), dim3(nblk), dim3(256), lds, st, hp, ps, Wt, g0, g1, g2, eps, sigma, dhp, dps, partial, B, V, C);
        }
    });
    return pulpo::check_launch("heads_bwd");
}

PULPO_API int pulpo_heads_bwd(const float* h, int64_t ps, const float* Wt, const float* g0, const float* g1, const float* g2, const float* eps,
                              const float* sigma, float* dh, int64_t dps, float* partial, int nout, int B, int64_t V, int C, void* stream) {
    return pulpo_heads_bwd_t(h, 0, ps, Wt, g0, g1, g2, eps, sigma, dh, dps, partial, nout, B, V, C, stream);
}
