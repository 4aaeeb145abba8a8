// BatchNorm3d (training / eval) + LeakyReLU(0.2) around the conv epilogue statistics.
// Reference: src/network_blocks.py:24-25 (nn.BatchNorm3d(eps 1e-5, momentum 0.1) -> nn.LeakyReLU(0.2, inplace)).
// All kernels are HBM-bound streaming passes over channels-last [pixel][C] activations (pixel stride explicit,
// so a channel slice of a concatenation buffer is a valid operand).  Reductions are two-stage and deterministic:
// per-block fp32 partials, then a double-precision column sum.
#include "act_io.h"

namespace {

// ---------------------------------------------------------------------------------------------- column sums
// out[c] = sum_r partials[r][c]   (r < nrow), accumulated in double.  grid = ceil(ncol/32), block = (32, 32)
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ partials, int nrow, int ncol, float* __restrict__ out,
                                                        float scale, int accumulate) {
    __shared__ double red[32][33];
    const int cx = threadIdx.x, ry = threadIdx.y;
    const int c = blockIdx.x * 32 + cx;
    double s = 0.0;
    if (c < ncol)
        for (int r = ry; r < nrow; r += 32) s += (double)partials[(long)r * ncol + c];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < ncol) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += red[k][cx];
        const float r = (float)(t * (double)scale);
        out[c] = accumulate ? out[c] + r : r;
    }
}

// stage 1 (large tile counts only): slice-wise double-precision column sums  part[S][ncol],  grid = (ceil(ncol/32), S), block = (32, 8).
// 256 threads at a handful of registers: the kernel runs in the backward pass BESIDE the persistent weight-gradient kernels of the other
// stream and only starts on a CU where its whole workgroup fits (a 1024-thread build waited ~100 us per launch for a CU the bf16 weight
// gradient had left: 3.4 ms per bf16 step).
constexpr int CS_RY = 8;
__global__ __launch_bounds__(32 * CS_RY) void colsum_slices_kernel(const float* __restrict__ rows, int nrow, int ncol, double* __restrict__ part) {
    __shared__ double red[CS_RY][33];
    const int cx = threadIdx.x, ry = threadIdx.y;
    const int c = blockIdx.x * 32 + cx;
    const int S = gridDim.y, sl = blockIdx.y;
    double s = 0.0;
    if (c < ncol)
        for (int r = sl * CS_RY + ry; r < nrow; r += CS_RY * S) s += (double)rows[(long)r * ncol + c];
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && c < ncol) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < CS_RY; ++k) t += red[k][cx];
        part[(long)sl * ncol + c] = t;
    }
}

// stats (fp32 tile partials [ntile][2][C], or - when part != nullptr - their double slice sums [nslice][2][C])
//   -> coef = [4][C] floats (mean, rstd, scale, shift) + [2][C] doubles (mean, rstd); running statistics update
// block = (8 channels, 128 row lanes), grid = ceil(C / 8): up to 2048 tile rows are walked in 16 passes of independent loads (the 32 x 32
// layout took 64 passes on 2 - 9 workgroups: 11 us per launch on the forward pass's critical path, 35 launches per step)
constexpr int FIN_CG = 8, FIN_RL = 128;
__global__ __launch_bounds__(FIN_CG * FIN_RL) void bn_fwd_finalize_kernel(const float* __restrict__ stats, const double* __restrict__ part, int nrow, int C,
                                                                 double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                 long long* __restrict__ num_batches_tracked, float momentum, float eps,
                                                                 float* __restrict__ coef) {
    __shared__ double red[2][FIN_RL][FIN_CG + 1];
    const int cx = threadIdx.x, ry = threadIdx.y;
    const int c = blockIdx.x * FIN_CG + cx;
    double s = 0.0, q = 0.0;
    if (c < C) {
        if (part != nullptr) {
            for (int r = ry; r < nrow; r += FIN_RL) {
                s += part[((long)r * 2 + 0) * C + c];
                q += part[((long)r * 2 + 1) * C + c];
            }
        } else {
#pragma unroll 4
            for (int r = ry; r < nrow; r += FIN_RL) {
                s += (double)stats[((long)r * 2 + 0) * C + c];
                q += (double)stats[((long)r * 2 + 1) * C + c];
            }
        }
    }
    red[0][ry][cx] = s;
    red[1][ry][cx] = q;
    __syncthreads();
    // 128 -> 8 row lanes, then one thread per channel
    if (ry < 8) {
        double ts = 0.0, tq = 0.0;
#pragma unroll
        for (int k = 0; k < FIN_RL / 8; ++k) { ts += red[0][ry + 8 * k][cx]; tq += red[1][ry + 8 * k][cx]; }
        red[0][ry][cx] = ts;
        red[1][ry][cx] = tq;
    }
    __syncthreads();
    if (blockIdx.x == 0 && cx == 0 && ry == 0 && num_batches_tracked != nullptr) num_batches_tracked[0] += 1;
    if (ry == 0 && c < C) {
        double ts = 0.0, tq = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { ts += red[0][k][cx]; tq += red[1][k][cx]; }
        const double mean = ts / count;
        double var = tq / count - mean * mean;      // biased batch variance
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * rstd;
        coef[0 * C + c] = (float)mean;
        coef[1 * C + c] = rstd;
        coef[2 * C + c] = sc;
        coef[3 * C + c] = beta[c] - (float)mean * sc;
        double* cd = reinterpret_cast<double*>(coef + 4 * C);     // [2][C] doubles: mean, rstd (used by the backward)
        cd[c] = mean;
        cd[C + c] = 1.0 / sqrt(var + (double)eps);
        if (running_mean != nullptr) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

__global__ void bn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                    const float* __restrict__ rv, float eps, int C, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = 1.f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * rstd;
    coef[0 * C + c] = rm[c];
    coef[1 * C + c] = rstd;
    coef[2 * C + c] = sc;
    coef[3 * C + c] = beta[c] - rm[c] * sc;
    double* cd = reinterpret_cast<double*>(coef + 4 * C);
    cd[c] = (double)rm[c];
    cd[C + c] = 1.0 / sqrt((double)rv[c] + (double)eps);
}

// Vec<VEC>::ld / st on FLOAT arrays (coefficient blocks, LDS tables); activation tensors go through pulpo::ldv / stv (act_io.h), which
// also read and write bf16 storage
template <int VEC>
struct Vec {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[VEC]) { pulpo::ldv<VEC>(p, v); }
    static __device__ __forceinline__ void st(float* p, const float (&v)[VEC]) { pulpo::stv<VEC>(p, v); }
};

// z = leaky_relu(y * scale[c] + shift[c])
template <int VEC, typename TY = float, typename TZ = float>
__global__ __launch_bounds__(256) void bn_lrelu_apply_kernel(const TY* __restrict__ y, long yps, TZ* __restrict__ z, long zps,
                                                               const float* __restrict__ coef, long npix, int C, float slope, long zkb = 8) {
    const int CV = C / VEC;
    const long total = npix * CV;
    const float* scale = coef + 2 * C;
    const float* shift = coef + 3 * C;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long p = e / CV;
        const int c = (int)(e - p * CV) * VEC;
        float v[VEC], sc[VEC], sh[VEC];
        pulpo::ldv<VEC>(y + p * yps + c, v);
        Vec<VEC>::ld(scale + c, sc);
        Vec<VEC>::ld(shift + c, sh);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float t = v[k] * sc[k] + sh[k];
            v[k] = t > 0.f ? t : t * slope;
        }
        pulpo::stv<VEC>(z + (long)(c >> 3) * zkb + p * zps + (c & 7), v);      // (zkb = 8: channels-last, c; else the channel-blocked layout [C / 8][pixels][8])
    }
}

// z = leaky_relu(y * scale + shift) AND pooled = AvgPool3d(2, 2, ceil_mode)(z) in one pass (the last ConvUnit of an encoder level: its
// output is pooled for the next level, components/pulpo.py:58): a thread owns one pooled voxel x four channels, i.e. up to eight voxels of
// y / z; same expressions and the same summation order (z, y, x) as bn_lrelu_apply_kernel followed by avgpool2_fwd_kernel.
template <typename TY = float, typename TZ = float>
__global__ __launch_bounds__(256) void bn_lrelu_apply_pool2_kernel(const TY* __restrict__ y, long yps, TZ* __restrict__ z, long zps,
                                                                     TZ* __restrict__ pooled, long pps, const float* __restrict__ coef, int B,
                                                                     int D, int H, int W, int Do, int Ho, int Wo, int C, float slope) {
    const int CV = C / 4;
    const long total = (long)B * Do * Ho * Wo * CV;
    const float* scale = coef + 2 * C;
    const float* shift = coef + 3 * C;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % CV) * 4;
        long p = e / CV;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho); p /= Ho;
        const int oz = (int)(p % Do);
        const int b = (int)(p / Do);
        const int z1 = min(2 * oz + 2, D), y1 = min(2 * oy + 2, H), x1 = min(2 * ox + 2, W);
        float sc[4], sh[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
        Vec<4>::ld(scale + c, sc);
        Vec<4>::ld(shift + c, sh);
        for (int zz = 2 * oz; zz < z1; ++zz)
            for (int yy = 2 * oy; yy < y1; ++yy)
                for (int xx = 2 * ox; xx < x1; ++xx) {
                    const long vox = (((long)b * D + zz) * H + yy) * W + xx;
                    float v[4];
                    pulpo::ldv<4>(y + vox * yps + c, v);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t = v[k] * sc[k] + sh[k];
                        v[k] = pulpo::as_stored<TZ>(t > 0.f ? t : t * slope);        // (the pooled tensor averages z as stored)
                        acc[k] += v[k];
                    }
                    if (z != nullptr) pulpo::stv<4>(z + vox * zps + c, v);       // (null: only the pooled tensor has a reader - DownPath levels above the first latent level)
                }
        const float inv = 1.f / (float)((z1 - 2 * oz) * (y1 - 2 * oy) * (x1 - 2 * ox));   // ceil_mode: divisor = in-bounds taps
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] *= inv;
        pulpo::stv<4>(pooled + ((((long)b * Do + oz) * Ho + oy) * Wo + ox) * pps + c, acc);
    }
}

// pass 1 of the backward: partial[blk][0][c] = sum dbn, partial[blk][1][c] = sum dbn * (y - m32)
//   dbn = dz * (bn_out > 0 ? 1 : slope),  bn_out = y*scale + shift,  m32 = the batch mean rounded to fp32 (coef[c])
// All fp32: the difference of two floats carries a relative error of 2^-24 however close they are, and what the ROUNDED mean leaves out is
// added back in double by bn_bwd_finalize_kernel (sum dbn * xhat = rstd * (sum dbn * (y - m32) - (mean - m32) * sum dbn)).  No doubles per
// element: a wave needs 40 instead of 64 registers, so that two instead of one fit on a SIMD beside the weight-gradient kernel.
// POOL: dz is not read but PRODUCED here - the gradient of an activation that was pooled (gpool: gradient of the pooled tensor, spread over
// each 2 x 2 x 2 window with the ceil-mode divisor, the arithmetic of avgpool2_bwd_kernel) and possibly also used as a skip connection
// (add, nullable) - and written to dzout on the way: the pooling backward, autograd's accumulation and this reduction in one pass.
template <typename TG>
struct PoolGrad {
    const TG* gpool; long gpps;
    const TG* add; long aps;
    TG* dzout; long dzops;
    int D, H, W, Do, Ho, Wo;
};

template <int VEC, bool POOL = false, typename TG = float, typename TY = float>
__global__ __launch_bounds__(256) void bn_lrelu_bwd_reduce_kernel(const TG* __restrict__ dz, long dzps, const TY* __restrict__ y,
                                                                    long yps, const float* __restrict__ coef, long npix, int C,
                                                                    float slope, float* __restrict__ partial, PoolGrad<TG> pg = PoolGrad<TG>{}) {
    extern __shared__ float red[];                 // [RB][2][C]
    const int CV = C / VEC, RB = blockDim.x / CV;
    const int col = threadIdx.x % CV, row = threadIdx.x / CV;
    const int c = col * VEC;
    float s0[VEC], s1[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s0[k] = s1[k] = 0.f;
    if (row < RB) {
        float sc[VEC], sh[VEC], m32[VEC];
        Vec<VEC>::ld(coef + 2 * C + c, sc);
        Vec<VEC>::ld(coef + 3 * C + c, sh);
        Vec<VEC>::ld(coef + c, m32);
        for (long p = (long)blockIdx.x * RB + row; p < npix; p += (long)gridDim.x * RB) {
            float g[VEC], v[VEC];
            if constexpr (POOL) {
                long q = p;
                const int x_ = (int)(q % pg.W); q /= pg.W;
                const int y_ = (int)(q % pg.H); q /= pg.H;
                const int z_ = (int)(q % pg.D);
                const long b_ = q / pg.D;
                const int oz = z_ >> 1, oy = y_ >> 1, ox = x_ >> 1;
                const int cnt = (min(2 * oz + 2, pg.D) - 2 * oz) * (min(2 * oy + 2, pg.H) - 2 * oy) * (min(2 * ox + 2, pg.W) - 2 * ox);
                const float inv = 1.f / (float)cnt;
                pulpo::ldv<VEC>(pg.gpool + (((b_ * pg.Do + oz) * pg.Ho + oy) * pg.Wo + ox) * pg.gpps + c, g);
#pragma unroll
                for (int k = 0; k < VEC; ++k) g[k] *= inv;
                if (pg.add != nullptr) {
                    float u[VEC];
                    pulpo::ldv<VEC>(pg.add + p * pg.aps + c, u);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) g[k] = u[k] + g[k];
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) g[k] = pulpo::as_stored<TG>(g[k]);     // (the sums describe the gradient as stored)
                if (pg.dzout != nullptr) pulpo::stv<VEC>(pg.dzout + p * pg.dzops + c, g);       // (null: the second pass forms it again, see POOL below)
            } else {
                pulpo::ldv<VEC>(dz + p * dzps + c, g);
            }
            pulpo::ldv<VEC>(y + p * yps + c, v);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float bn = v[k] * sc[k] + sh[k];
                const float d = bn > 0.f ? g[k] : g[k] * slope;
                s0[k] += d;
                s1[k] = fmaf(d, v[k] - m32[k], s1[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            red[(row * 2 + 0) * C + c + k] = s0[k];
            red[(row * 2 + 1) * C + c + k] = s1[k];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * C; j += blockDim.x) {
        float t = 0.f;
        for (int r = 0; r < RB; ++r) t += red[r * 2 * C + j];
        partial[(long)blockIdx.x * 2 * C + j] = t;
    }
}

// pass 2: dy = scale * (dbn - mean(dbn) - xhat * mean(dbn*xhat));  partial2[blk][c] = sum dy  (conv-bias gradient)
// Per element in fp32 as  dy = A*dbn + B*(y - m32) + C  with per-channel constants formed in double from the exact means:
//   A = scale,  B = -scale * c2 * rstd,  C = -scale * (c1 + c2 * rstd * (m32 - mean)),  C carried as a (hi, lo) pair of floats.
// A mean rounded to fp32 would shift every dy of the channel by the same amount, an error that the following weight-gradient sum
// amplifies by the voxel count (ATen's CPU BatchNorm accumulates in double for the same reason): here the only systematic term, C, keeps 48
// bits, the (y - m32) factor is centred (a rounding of B does not shift the channel) and everything else rounds without bias.
// The six per-channel constants live in LDS and are read per element group (6 ds_read_b128 against 48 bytes of HBM traffic): the kernel
// then needs ~40 instead of 71 registers, i.e. two waves instead of one fit on a SIMD beside the weight-gradient kernel.
// POOL (round 5): dz is not read but formed per element from the gradient of the POOLED tensor (and the skip connection's gradient, nullable) -
// the arithmetic of the POOL form of bn_lrelu_bwd_reduce_kernel above - so that the gradient of a pooled ConvUnit output is never written:
// both passes read the (eight times smaller) pooled gradient instead.
template <int VEC, typename TG = float, typename TY = float, bool POOL = false>
__global__ __launch_bounds__(256) void bn_lrelu_bwd_apply_kernel(const TG* __restrict__ dz, long dzps, const TY* __restrict__ y,
                                                                   long yps, const float* __restrict__ coef, const double* __restrict__ totd,
                                                                   TY* __restrict__ dy, long dyps, long npix, int C,
                                                                   float slope, float* __restrict__ partial2, PoolGrad<TG> pg = PoolGrad<TG>{}, long dykb = 8, long dzkb = 8) {
    extern __shared__ float red[];                 // [RB][C] partial sums, then [6][C] constants: scale, shift, m32, B, C hi, C lo
    const int CV = C / VEC, RB = blockDim.x / CV;
    const int col = threadIdx.x % CV, row = threadIdx.x / CV;
    const int c = col * VEC;
    float* kst = red + RB * C;
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
        const double* cd = reinterpret_cast<const double*>(coef + 4 * C);
        const float sc_ = coef[2 * C + ch], m32_ = coef[ch];
        const double mean = cd[ch], rstd = cd[C + ch], c1 = totd[ch], c2 = totd[C + ch];
        const double b = -(double)sc_ * c2 * rstd;
        const double cc = -(double)sc_ * (c1 + c2 * rstd * ((double)m32_ - mean));
        const float chi_ = (float)cc;
        kst[0 * C + ch] = sc_;
        kst[1 * C + ch] = coef[3 * C + ch];
        kst[2 * C + ch] = m32_;
        kst[3 * C + ch] = (float)b;
        kst[4 * C + ch] = chi_;
        kst[5 * C + ch] = (float)(cc - (double)chi_);
    }
    __syncthreads();
    float s0[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s0[k] = 0.f;
    // dy: channel c of pixel p at (c / 8) * dykb + p * dyps + c % 8 - dykb = 8 is the channels-last tensor, dykb = npix * 8 with dyps = 8 the
    // channel-blocked layout [C / 8][pixels][8] that the F(2x2x2,3x3x3) data- / weight-gradient kernels read (pulpo_bn_lrelu_bwd_apply_kb_t)
    const long cdst = (long)(c >> 3) * dykb + (c & 7);
    const long csrc = (long)(c >> 3) * dzkb + (c & 7);      // (dz likewise: the gradient of a blocked activation arrives blocked)
    if (row < RB) {
        for (long p = (long)blockIdx.x * RB + row; p < npix; p += (long)gridDim.x * RB) {
            float g[VEC], v[VEC], o[VEC];
            if constexpr (POOL) {
                long q = p;
                const int x_ = (int)(q % pg.W); q /= pg.W;
                const int y_ = (int)(q % pg.H); q /= pg.H;
                const int z_ = (int)(q % pg.D);
                const long b_ = q / pg.D;
                const int oz = z_ >> 1, oy = y_ >> 1, ox = x_ >> 1;
                const int cnt = (min(2 * oz + 2, pg.D) - 2 * oz) * (min(2 * oy + 2, pg.H) - 2 * oy) * (min(2 * ox + 2, pg.W) - 2 * ox);
                const float inv = 1.f / (float)cnt;
                pulpo::ldv<VEC>(pg.gpool + (((b_ * pg.Do + oz) * pg.Ho + oy) * pg.Wo + ox) * pg.gpps + c, g);
#pragma unroll
                for (int k = 0; k < VEC; ++k) g[k] *= inv;
                if (pg.add != nullptr) {
                    float u[VEC];
                    pulpo::ldv<VEC>(pg.add + p * pg.aps + c, u);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) g[k] = u[k] + g[k];
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) g[k] = pulpo::as_stored<TG>(g[k]);
            } else {
                pulpo::ldv<VEC>(dz + p * dzps + csrc, g);
            }
            pulpo::ldv<VEC>(y + p * yps + c, v);
            int cl = c;                                 // (opaque: the constants are to be READ here every time, not kept in registers)
            asm volatile("" : "+v"(cl));
            // three stages, each with its own constants, so that no more than three constant vectors are live at a time
            {
                float sh[VEC], sc[VEC];
                Vec<VEC>::ld(kst + 0 * C + cl, sc);
                Vec<VEC>::ld(kst + 1 * C + cl, sh);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float bn = v[k] * sc[k] + sh[k];
                    g[k] = sc[k] * (bn > 0.f ? g[k] : g[k] * slope);          // scale * dbn
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                float m32[VEC], cb[VEC], chi[VEC];
                Vec<VEC>::ld(kst + 2 * C + cl, m32);
                Vec<VEC>::ld(kst + 3 * C + cl, cb);
                Vec<VEC>::ld(kst + 4 * C + cl, chi);
#pragma unroll
                for (int k = 0; k < VEC; ++k) v[k] = fmaf(cb[k], v[k] - m32[k], chi[k]);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                float clo[VEC];
                Vec<VEC>::ld(kst + 5 * C + cl, clo);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    o[k] = (g[k] + v[k]) + clo[k];
                    s0[k] += o[k];
                }
            }
            pulpo::stv<VEC>(dy + p * dyps + cdst, o);
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) red[row * C + c + k] = s0[k];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < C; j += blockDim.x) {
        float t = 0.f;
        for (int r = 0; r < RB; ++r) t += red[r * C + j];
        partial2[(long)blockIdx.x * C + j] = t;
    }
}

// rows [nrow][2][C] = (sum dbn, sum dbn * (y - m32)) - the block partials of bn_lrelu_bwd_reduce_kernel or the per-voxel-tile rows a
// convolution epilogue wrote (or, part != nullptr, their double slice sums); m32 = coef[c], the fp32-rounded batch mean
//   -> (dbeta | dgamma) as floats, totd[2C] = (mean dbn | mean dbn * xhat) as doubles;  sum dbn * xhat = rstd * (q - (mean - m32) * s).
// use_means == 0 (eval-mode BatchNorm is a fixed affine map): totd = 0.
__global__ __launch_bounds__(512) void bn_bwd_finalize_kernel(const float* __restrict__ rows, const double* __restrict__ part, int nrow, int C,
                                                                       const float* __restrict__ coef, double count, int use_means,
                                                                       float* __restrict__ dbeta, float* __restrict__ dgamma, int accumulate,
                                                                       double* __restrict__ totd) {
    // block = (8 channels, 64 row lanes), grid = ceil(C / 8) (round 5; was 32 channels x 16 row lanes on ceil(C / 32) workgroups: 2 workgroups
    // walked the 2000 rows of an 80^3 layer in 125 passes).  512 threads at <= 32 registers: the kernel still fits beside a weight-gradient
    // kernel of the other stream (bf16 mode).  Rows are summed in row order per lane, the lanes in lane order: deterministic.
    constexpr int CG = 8, RL = 64;
    __shared__ double red[2][RL][CG + 1];
    const int cx = threadIdx.x, ry = threadIdx.y;
    const int c = blockIdx.x * CG + cx;
    double s = 0.0, q = 0.0;
    if (c < C) {
        if (part != nullptr) {
#pragma unroll 1
            for (int r = ry; r < nrow; r += RL) {
                s += part[((long)r * 2 + 0) * C + c];
                q += part[((long)r * 2 + 1) * C + c];
            }
        } else {
            int r = ry;
#pragma unroll 1
            for (; r + 3 * RL < nrow; r += 4 * RL) {          // four rows of independent loads per pass
                const float a0 = rows[((long)r * 2 + 0) * C + c], b0 = rows[((long)r * 2 + 1) * C + c];
                const float a1 = rows[((long)(r + RL) * 2 + 0) * C + c], b1 = rows[((long)(r + RL) * 2 + 1) * C + c];
                const float a2 = rows[((long)(r + 2 * RL) * 2 + 0) * C + c], b2 = rows[((long)(r + 2 * RL) * 2 + 1) * C + c];
                const float a3 = rows[((long)(r + 3 * RL) * 2 + 0) * C + c], b3 = rows[((long)(r + 3 * RL) * 2 + 1) * C + c];
                s += (double)a0; q += (double)b0; s += (double)a1; q += (double)b1; s += (double)a2; q += (double)b2; s += (double)a3; q += (double)b3;
            }
#pragma unroll 1
            for (; r < nrow; r += RL) {
                s += (double)rows[((long)r * 2 + 0) * C + c];
                q += (double)rows[((long)r * 2 + 1) * C + c];
            }
        }
    }
    red[0][ry][cx] = s;
    red[1][ry][cx] = q;
    __syncthreads();
    if (ry == 0 && c < C) {
        double ts = 0.0, tq = 0.0;
#pragma unroll 1
        for (int k = 0; k < RL; ++k) { ts += red[0][k][cx]; tq += red[1][k][cx]; }
        const double* cd = reinterpret_cast<const double*>(coef + 4 * C);
        const double tx = cd[C + c] * (tq - (cd[c] - (double)coef[c]) * ts);
        dbeta[c] = accumulate ? dbeta[c] + (float)ts : (float)ts;
        dgamma[c] = accumulate ? dgamma[c] + (float)tx : (float)tx;
        totd[c] = use_means ? ts / count : 0.0;
        totd[C + c] = use_means ? tx / count : 0.0;
    }
}

inline bool vec_ok(const void* a, long aps, const void* b, long bps, int C) {
    return C % 4 == 0 && aps % 4 == 0 && bps % 4 == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
}
inline int stream_blocks(long items) { return (int)std::max<long>(1, std::min<long>((items + 255) / 256, 4096)); }

}  // namespace

// out[c] (+)= scale * sum_r partials[r][c];  accumulate != 0 adds to the existing value (gradient accumulation into a .grad)
PULPO_API int pulpo_colsum(const float* partials, int nrow, int ncol, float* out, float scale, int accumulate, void* stream) {
    PULPO_REQUIRE(partials && out && nrow > 0 && ncol > 0, "colsum: bad arguments");
    hipLaunchKernelGGL(colsum_kernel, dim3(pulpo::cdiv(ncol, 32)), dim3(32, 32), 0, (hipStream_t)stream, partials, nrow, ncol, out, scale, accumulate);
    return pulpo::check_launch("colsum");
}

// (slices of the first stage: 128 - the 16000 tile rows of a 160^3 layer then take 16 passes per thread of its 256 workgroups)
constexpr int FWD_SLICES = 128;
PULPO_API size_t pulpo_bn_fwd_finalize_scratch_doubles(int ntile, int C) { return ntile > 2048 ? (size_t)FWD_SLICES * 2 * C : 0; }

// scratch: pulpo_bn_fwd_finalize_scratch_doubles(ntile, C) doubles (may be NULL when that is 0).
// num_batches_tracked (nullable): int64 counter of nn.BatchNorm3d, incremented here.
PULPO_API int pulpo_bn_fwd_finalize(const float* stats, int ntile, int C, double count, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                                    float* coef, double* scratch, void* stream) {
    PULPO_REQUIRE(stats && gamma && beta && coef && ntile > 0 && C > 0 && count > 0, "bn_fwd_finalize: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const double* part = nullptr;
    int nrow = ntile;
    if (ntile > 2048) {
        PULPO_REQUIRE(scratch != nullptr, "bn_fwd_finalize: scratch required for %d tiles", ntile);
        hipLaunchKernelGGL(colsum_slices_kernel, dim3(pulpo::cdiv(2 * C, 32), FWD_SLICES), dim3(32, CS_RY), 0, st, stats, ntile, 2 * C, scratch);
        int rc = pulpo::check_launch("bn stats slices");
        if (rc) return rc;
        part = scratch;
        nrow = FWD_SLICES;
    }
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(pulpo::cdiv(C, FIN_CG)), dim3(FIN_CG, FIN_RL), 0, st, stats, part, nrow, C, count, gamma, beta,
                       running_mean, running_var, (long long*)num_batches_tracked, momentum, eps, coef);
    return pulpo::check_launch("bn_fwd_finalize");
}

PULPO_API int pulpo_bn_eval_coef(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                                 int C, float* coef, void* stream) {
    PULPO_REQUIRE(gamma && beta && running_mean && running_var && coef && C > 0, "bn_eval_coef: bad arguments");
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(pulpo::cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                       running_var, eps, C, coef);
    return pulpo::check_launch("bn_eval_coef");
}

// ---- typed entry points: activation tensors are fp32 (dtype code 0) or bf16 (1) in HBM, strides in ELEMENTS; arithmetic is fp32 either way.
// The untyped names below them are the fp32 forms.
namespace {
template <typename TY, typename TZ>
int apply_t(const TY* y, long yps, TZ* z, long zps, const float* coef, long npix, int C, float slope, hipStream_t st, long zkb = 8) {
    pulpo::GroupProbe g(C);
    g.add(y, yps, sizeof(TY)); g.add(z, zps, sizeof(TZ)); g.add(coef, 8, 4);
    constexpr bool half = sizeof(TY) == 2 || sizeof(TZ) == 2;
    if (half && g.ok8)
        hipLaunchKernelGGL((bn_lrelu_apply_kernel<8, TY, TZ>), dim3(stream_blocks(npix * (C / 8))), dim3(256), 0, st, y, yps, z, zps, coef, npix, C, slope, zkb);
    else if (g.ok4)
        hipLaunchKernelGGL((bn_lrelu_apply_kernel<4, TY, TZ>), dim3(stream_blocks(npix * (C / 4))), dim3(256), 0, st, y, yps, z, zps, coef, npix, C, slope, zkb);
    else
        hipLaunchKernelGGL((bn_lrelu_apply_kernel<1, TY, TZ>), dim3(stream_blocks(npix * C)), dim3(256), 0, st, y, yps, z, zps, coef, npix, C, slope, zkb);
    return pulpo::check_launch("bn_lrelu_apply");
}
}  // namespace

PULPO_API int pulpo_bn_lrelu_apply_t(const void* y, int y_dt, int64_t yps, void* z, int z_dt, int64_t zps, const float* coef, int64_t npix, int C,
                                     float slope, void* stream) {
    PULPO_REQUIRE(y && z && coef && npix > 0 && C > 0, "bn_lrelu_apply: bad arguments");
    PULPO_REQUIRE_DT(y_dt, "bn_lrelu_apply"); PULPO_REQUIRE_DT(z_dt, "bn_lrelu_apply");
    PULPO_DISPATCH_DT(y_dt, TY, PULPO_DISPATCH_DT(z_dt, TZ, return apply_t((const TY*)y, (long)yps, (TZ*)z, (long)zps, coef, (long)npix, C, slope, (hipStream_t)stream)));
    return -1;
}

PULPO_API int pulpo_bn_lrelu_apply(const float* y, int64_t yps, float* z, int64_t zps, const float* coef, int64_t npix, int C, float slope,
                                   void* stream) {
    return pulpo_bn_lrelu_apply_t(y, 0, yps, z, 0, zps, coef, npix, C, slope, stream);
}

// z in the channel-BLOCKED layout [C / 8][pixels][8] (zps = 8, zkb = npix * 8; see pulpo_conv3d_k3_fwd_wino3_kb): the activation between two ConvUnits
// of a ConvSequence is read by the next unit's convolution and weight gradient only.  fp32, C % 8 == 0.
PULPO_API int pulpo_bn_lrelu_apply_kb(const float* y, int64_t yps, float* z, int64_t zps, int64_t zkb, const float* coef, int64_t npix, int C, float slope,
                                      void* stream) {
    PULPO_REQUIRE(y && z && coef && npix > 0 && C > 0 && C % 8 == 0 && zps % 4 == 0 && zps >= 8 && zkb % 4 == 0 && zkb >= 8, "bn_lrelu_apply_kb: bad arguments");
    return apply_t(y, (long)yps, z, (long)zps, coef, (long)npix, C, slope, (hipStream_t)stream, (long)zkb);
}

// 1 when pulpo_bn_lrelu_apply_pool2 accepts the operands (groups of four channels: C % 4 == 0, strides % 4 == 0)
PULPO_API int pulpo_bn_lrelu_apply_pool2_ok(int C, int64_t yps, int64_t zps, int64_t pps) { return C % 4 == 0 && yps % 4 == 0 && zps % 4 == 0 && pps % 4 == 0; }

PULPO_API int pulpo_bn_lrelu_apply_pool2_t(const void* y, int y_dt, int64_t yps, void* z, int z_dt, int64_t zps, void* pooled, int64_t pps,
                                           const float* coef, int B, int D, int H, int W, int C, float slope, void* stream) {
    // (z nullable since ABI 4: the un-pooled activation is not written - its only reader would have been the pooling)
    PULPO_REQUIRE(y && pooled && coef && B > 0 && D > 0 && H > 0 && W > 0 && C > 0, "bn_lrelu_apply_pool2: bad arguments");
    PULPO_REQUIRE_DT(y_dt, "bn_lrelu_apply_pool2"); PULPO_REQUIRE_DT(z_dt, "bn_lrelu_apply_pool2");
    const int ey = y_dt ? 2 : 4, ez = z_dt ? 2 : 4;
    PULPO_REQUIRE(pulpo_bn_lrelu_apply_pool2_ok(C, yps, zps, pps) && (((uintptr_t)y) % (4 * ey)) == 0 && ((((uintptr_t)z) | ((uintptr_t)pooled)) % (4 * ez)) == 0 &&
                      (((uintptr_t)coef) & 15) == 0,
                  "bn_lrelu_apply_pool2: operands must be channels-last, aligned to four elements, C %% 4 == 0");
    const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int nb = stream_blocks((long)B * Do * Ho * Wo * (C / 4));
    PULPO_DISPATCH_DT(y_dt, TY, PULPO_DISPATCH_DT(z_dt, TZ,
        hipLaunchKernelGGL((bn_lrelu_apply_pool2_kernel<TY, TZ>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const TY*)y, (long)yps, (TZ*)z, (long)zps,
                           (TZ*)pooled, (long)pps, coef, B, D, H, W, Do, Ho, Wo, C, slope)));
    return pulpo::check_launch("bn_lrelu_apply_pool2");
}

PULPO_API int pulpo_bn_lrelu_apply_pool2(const float* y, int64_t yps, float* z, int64_t zps, float* pooled, int64_t pps, const float* coef, int B, int D,
                                         int H, int W, int C, float slope, void* stream) {
    return pulpo_bn_lrelu_apply_pool2_t(y, 0, yps, z, 0, zps, pooled, pps, coef, B, D, H, W, C, slope, stream);
}

// number of partial rows the two backward passes write (caller allocates partial[nblk][2C] and partial2[nblk][C])
PULPO_API int pulpo_bn_bwd_blocks(int64_t npix, int C) {
    const int vec = (C % 4 == 0) ? 4 : 1;
    const int RB = std::max(1, 256 / (C / vec));
    return (int)std::max<long>(1, std::min<long>((npix + RB * 8 - 1) / (RB * 8), 2048));
}

namespace {
// channel-group width of a backward pass: 8 when a bf16 tensor takes part and everything allows it, else 4, else 1 (-1: misaligned)
inline int bwd_group(const pulpo::GroupProbe& g, bool half, int C) {
    if (half && g.ok8 && C / 8 <= 256) return 8;
    if (g.ok4) return C / 4 <= 256 ? 4 : -2;
    if (C % 4 == 0) return -1;
    return C <= 256 ? 1 : -2;
}

template <typename TG, typename TY>
int reduce_t(const TG* dz, long dzps, const TY* y, long yps, const float* coef, long npix, int C, float slope, float* partial, hipStream_t st) {
    pulpo::GroupProbe g(C);
    g.add(dz, dzps, sizeof(TG)); g.add(y, yps, sizeof(TY)); g.add(coef, 8, 4);
    const int vec = bwd_group(g, sizeof(TG) == 2 || sizeof(TY) == 2, C);
    if (vec == -1) return pulpo::fail(-1, "bn_lrelu_bwd_reduce: operands must be aligned to four elements when C %% 4 == 0");
    if (vec == -2) return pulpo::fail(-1, "bn_lrelu_bwd_reduce: too many channels (%d)", C);
    const int nblk = pulpo_bn_bwd_blocks(npix, C);
    const int RB = std::max(1, 256 / (C / vec));
    const size_t lds = (size_t)RB * 2 * C * sizeof(float);
    if (vec == 8) hipLaunchKernelGGL((bn_lrelu_bwd_reduce_kernel<8, false, TG, TY>), dim3(nblk), dim3(256), lds, st, dz, dzps, y, yps, coef, npix, C, slope, partial, PoolGrad<TG>{});
    else if (vec == 4) hipLaunchKernelGGL((bn_lrelu_bwd_reduce_kernel<4, false, TG, TY>), dim3(nblk), dim3(256), lds, st, dz, dzps, y, yps, coef, npix, C, slope, partial, PoolGrad<TG>{});
    else hipLaunchKernelGGL((bn_lrelu_bwd_reduce_kernel<1, false, TG, TY>), dim3(nblk), dim3(256), lds, st, dz, dzps, y, yps, coef, npix, C, slope, partial, PoolGrad<TG>{});
    return pulpo::check_launch("bn_lrelu_bwd_reduce");
}

template <typename TG, typename TY>
int pool_reduce_t(const TG* gout, long gops, const TG* add, long aps, TG* gin, long gips, const TY* y, long yps, const float* coef, float slope,
                  float* partial, int B, int D, int H, int W, int C, hipStream_t st) {
    const long npix = (long)B * D * H * W;
    const int nblk = pulpo_bn_bwd_blocks(npix, C);
    const int RB = std::max(1, 256 / (C / 4));
    const size_t lds = (size_t)RB * 2 * C * sizeof(float);
    PoolGrad<TG> pg{gout, gops, add, aps, gin, gips, D, H, W, (D + 1) / 2, (H + 1) / 2, (W + 1) / 2};
    hipLaunchKernelGGL((bn_lrelu_bwd_reduce_kernel<4, true, TG, TY>), dim3(nblk), dim3(256), lds, st, (const TG*)nullptr, 0L, y, yps, coef, npix, C, slope,
                       partial, pg);
    return pulpo::check_launch("avgpool2_bwd_bnred");
}

template <typename TG, typename TY>
int bwd_apply_t(const TG* dz, long dzps, const TY* y, long yps, const float* coef, const double* totd, TY* dy, long dyps, long npix, int C, float slope,
                float* partial2, hipStream_t st, long dykb = 8, long dzkb = 8) {
    pulpo::GroupProbe g(C);
    g.add(dz, dzps, sizeof(TG)); g.add(y, yps, sizeof(TY)); g.add(dy, dyps, sizeof(TY)); g.add(coef, 8, 4);
    const int vec = bwd_group(g, sizeof(TG) == 2 || sizeof(TY) == 2, C);
    if (vec == -1) return pulpo::fail(-1, "bn_lrelu_bwd_apply: operands must be aligned to four elements when C %% 4 == 0");
    if (vec == -2) return pulpo::fail(-1, "bn_lrelu_bwd_apply: too many channels (%d)", C);
    const int nblk = pulpo_bn_bwd_blocks(npix, C);
    const int RB = std::max(1, 256 / (C / vec));
    const size_t lds = (size_t)(RB + 6) * C * sizeof(float);
    if (vec == 8) hipLaunchKernelGGL((bn_lrelu_bwd_apply_kernel<8, TG, TY>), dim3(nblk), dim3(256), lds, st, dz, dzps, y, yps, coef, totd, dy, dyps, npix, C, slope, partial2, PoolGrad<TG>{}, dykb, dzkb);
    else if (vec == 4) hipLaunchKernelGGL((bn_lrelu_bwd_apply_kernel<4, TG, TY>), dim3(nblk), dim3(256), lds, st, dz, dzps, y, yps, coef, totd, dy, dyps, npix, C, slope, partial2, PoolGrad<TG>{}, dykb, dzkb);
    else hipLaunchKernelGGL((bn_lrelu_bwd_apply_kernel<1, TG, TY>), dim3(nblk), dim3(256), lds, st, dz, dzps, y, yps, coef, totd, dy, dyps, npix, C, slope, partial2, PoolGrad<TG>{}, dykb, dzkb);
    return pulpo::check_launch("bn_lrelu_bwd_apply");
}
template <typename TG, typename TY>
int pool_apply_t(const TG* gout, long gops, const TG* add, long aps, const TY* y, long yps, const float* coef, const double* totd, TY* dy, long dyps,
                 float slope, float* partial2, int B, int D, int H, int W, int C, hipStream_t st, long dykb = 8) {
    const long npix = (long)B * D * H * W;
    const int nblk = pulpo_bn_bwd_blocks(npix, C);
    const int RB = std::max(1, 256 / (C / 4));
    const size_t lds = (size_t)(RB + 6) * C * sizeof(float);
    PoolGrad<TG> pg{gout, gops, add, aps, nullptr, 0, D, H, W, (D + 1) / 2, (H + 1) / 2, (W + 1) / 2};
    hipLaunchKernelGGL((bn_lrelu_bwd_apply_kernel<4, TG, TY, true>), dim3(nblk), dim3(256), lds, st, (const TG*)nullptr, 0L, y, yps, coef, totd, dy, dyps, npix,
                       C, slope, partial2, pg, dykb);
    return pulpo::check_launch("bn_lrelu_bwd_apply_pooled");
}
}  // namespace

PULPO_API int pulpo_bn_lrelu_bwd_reduce_t(const void* dz, int dz_dt, int64_t dzps, const void* y, int y_dt, int64_t yps, const float* coef, int64_t npix,
                                          int C, float slope, float* partial, void* stream) {
    PULPO_REQUIRE(dz && y && coef && partial && npix > 0 && C > 0, "bn_lrelu_bwd_reduce: bad arguments");
    PULPO_REQUIRE_DT(dz_dt, "bn_lrelu_bwd_reduce"); PULPO_REQUIRE_DT(y_dt, "bn_lrelu_bwd_reduce");
    PULPO_DISPATCH_DT(dz_dt, TG, PULPO_DISPATCH_DT(y_dt, TY,
        return reduce_t((const TG*)dz, (long)dzps, (const TY*)y, (long)yps, coef, (long)npix, C, slope, partial, (hipStream_t)stream)));
    return -1;
}

PULPO_API int pulpo_bn_lrelu_bwd_reduce(const float* dz, int64_t dzps, const float* y, int64_t yps, const float* coef, int64_t npix, int C,
                                        float slope, float* partial, void* stream) {
    return pulpo_bn_lrelu_bwd_reduce_t(dz, 0, dzps, y, 0, yps, coef, npix, C, slope, partial, stream);
}

// gin = (add +) avgpool2_bwd(gout) AND the first pass of the BatchNorm / LeakyReLU backward of the ConvUnit whose output was pooled (y, coef:
// that unit's pre-norm tensor and coefficient block): partial rows as pulpo_bn_lrelu_bwd_reduce writes them (same voxel-to-row assignment,
// same sums).  The last unit of every encoder level (components/pulpo.py:58): its gradient is read once instead of written, read, read.
// Channels-last operands in groups of four channels only (C % 4 == 0, rows aligned to four elements); add nullable.  gout / add / gin share
// one dtype (g_dt), y has its own.
PULPO_API int pulpo_avgpool2_bwd_bnred_t(const void* gout, int64_t gops, const void* add, int64_t aps, void* gin, int64_t gips, int g_dt, const void* y,
                                         int y_dt, int64_t yps, const float* coef, float slope, float* partial, int B, int D, int H, int W, int C,
                                         void* stream) {
    // (gin nullable since ABI 4: the caller then runs the second pass in its pooled form, pulpo_bn_lrelu_bwd_apply_pooled_t, and the gradient of
    //  the un-pooled tensor is never written)
    PULPO_REQUIRE(gout && y && coef && partial && B > 0 && D > 0 && H > 0 && W > 0 && C > 0, "avgpool2_bwd_bnred: bad arguments");
    PULPO_REQUIRE_DT(g_dt, "avgpool2_bwd_bnred"); PULPO_REQUIRE_DT(y_dt, "avgpool2_bwd_bnred");
    const int eg = g_dt ? 8 : 16, ey = y_dt ? 8 : 16;
    PULPO_REQUIRE(C % 4 == 0 && C / 4 <= 256 && gops % 4 == 0 && gips % 4 == 0 && yps % 4 == 0 && (add == nullptr || aps % 4 == 0) &&
                      ((((uintptr_t)gout) | ((uintptr_t)gin) | ((uintptr_t)add)) % eg) == 0 && (((uintptr_t)y) % ey) == 0 && (((uintptr_t)coef) & 15) == 0,
                  "avgpool2_bwd_bnred: operands must be channels-last, aligned to four elements, C %% 4 == 0");
    PULPO_DISPATCH_DT(g_dt, TG, PULPO_DISPATCH_DT(y_dt, TY,
        return pool_reduce_t((const TG*)gout, (long)gops, (const TG*)add, (long)aps, (TG*)gin, (long)gips, (const TY*)y, (long)yps, coef, slope, partial, B, D,
                             H, W, C, (hipStream_t)stream)));
    return -1;
}

PULPO_API int pulpo_avgpool2_bwd_bnred(const float* gout, int64_t gops, const float* add, int64_t aps, float* gin, int64_t gips, const float* y,
                                       int64_t yps, const float* coef, float slope, float* partial, int B, int D, int H, int W, int C, void* stream) {
    return pulpo_avgpool2_bwd_bnred_t(gout, gops, add, aps, gin, gips, 0, y, 0, yps, coef, slope, partial, B, D, H, W, C, stream);
}

// Second pass of the BatchNorm / LeakyReLU backward for a ConvUnit whose output was pooled (and possibly used as a skip connection): dz =
// (add +) avgpool2_bwd(gout) is formed per element, as pulpo_avgpool2_bwd_bnred_t with gin == NULL formed it for the first pass.  Same operand
// contract as that entry point; dy has y's dtype; partial2 as pulpo_bn_lrelu_bwd_apply_t (pulpo_bn_bwd_blocks(B*D*H*W, C) rows).
PULPO_API int pulpo_bn_lrelu_bwd_apply_pooled_t(const void* gout, int64_t gops, const void* add, int64_t aps, int g_dt, const void* y, int y_dt, int64_t yps,
                                                const float* coef, const double* totd, void* dy, int64_t dyps, float slope, float* partial2, int B, int D,
                                                int H, int W, int C, void* stream) {
    PULPO_REQUIRE(gout && y && coef && totd && dy && partial2 && B > 0 && D > 0 && H > 0 && W > 0 && C > 0, "bn_lrelu_bwd_apply_pooled: bad arguments");
    PULPO_REQUIRE_DT(g_dt, "bn_lrelu_bwd_apply_pooled"); PULPO_REQUIRE_DT(y_dt, "bn_lrelu_bwd_apply_pooled");
    const int eg = g_dt ? 8 : 16, ey = y_dt ? 8 : 16;
    PULPO_REQUIRE(C % 4 == 0 && C / 4 <= 256 && gops % 4 == 0 && yps % 4 == 0 && dyps % 4 == 0 && (add == nullptr || aps % 4 == 0) &&
                      ((((uintptr_t)gout) | ((uintptr_t)add)) % eg) == 0 && ((((uintptr_t)y) | ((uintptr_t)dy)) % ey) == 0 && (((uintptr_t)coef) & 15) == 0,
                  "bn_lrelu_bwd_apply_pooled: operands must be channels-last, aligned to four elements, C %% 4 == 0");
    PULPO_DISPATCH_DT(g_dt, TG, PULPO_DISPATCH_DT(y_dt, TY,
        return pool_apply_t((const TG*)gout, (long)gops, (const TG*)add, (long)aps, (const TY*)y, (long)yps, coef, totd, (TY*)dy, (long)dyps, slope, partial2, B,
                            D, H, W, C, (hipStream_t)stream)));
    return -1;
}

// dbeta / dgamma: [C] each, written (accumulate = 0) or added to (accumulate = 1, e.g. the parameters' .grad storage).
// rows: the block partials of pulpo_bn_lrelu_bwd_reduce (nrow = pulpo_bn_bwd_blocks) or the per-voxel-tile rows of
// pulpo_conv3d_k3_dgrad_wino2_bnred (nrow = pulpo_conv3d_k3_stat_tiles); coef = the unit's coefficient block.  With more than 64 rows they
// are first summed slice-wise in double (scratch: pulpo_bn_bwd_finalize_scratch_doubles(nrow, C) doubles, else NULL).
// (two-stage above 2048 rows - the 160^3 level; until round 5 from 64 rows up: 28 slice-sum launches per step that cost what they saved)
PULPO_API size_t pulpo_bn_bwd_finalize_scratch_doubles(int ntile, int C) { return ntile > 2048 ? (size_t)32 * 2 * C : 0; }

PULPO_API int pulpo_bn_bwd_finalize(const float* tile_part, int ntile, int C, const float* coef, double count, int use_means, float* dbeta,
                                          float* dgamma, int accumulate, double* totd, double* scratch, void* stream) {
    PULPO_REQUIRE(tile_part && coef && dbeta && dgamma && totd && ntile > 0 && C > 0 && count > 0, "bn_bwd_finalize: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const double* partd = nullptr;
    int nrow = ntile;
    if (pulpo_bn_bwd_finalize_scratch_doubles(ntile, C) != 0) {
        PULPO_REQUIRE(scratch != nullptr, "bn_bwd_finalize: scratch required for %d tiles", ntile);
        hipLaunchKernelGGL(colsum_slices_kernel, dim3(pulpo::cdiv(2 * C, 32), 32), dim3(32, CS_RY), 0, st, tile_part, ntile, 2 * C, scratch);
        int rc = pulpo::check_launch("bn backward tile slices");
        if (rc) return rc;
        partd = scratch;
        nrow = 32;
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(pulpo::cdiv(C, 8)), dim3(8, 64), 0, st, tile_part, partd, nrow, C, coef, count, use_means,
                       dbeta, dgamma, accumulate, totd);
    return pulpo::check_launch("bn_bwd_finalize");
}

// dy has y's dtype (the unit's pre-norm tensor and its gradient are stored alike); dz has its own
PULPO_API int pulpo_bn_lrelu_bwd_apply_t(const void* dz, int dz_dt, int64_t dzps, const void* y, int y_dt, int64_t yps, const float* coef,
                                         const double* totd, void* dy, int64_t dyps, int64_t npix, int C, float slope, float* partial2, void* stream) {
    PULPO_REQUIRE(dz && y && coef && totd && dy && partial2 && npix > 0 && C > 0, "bn_lrelu_bwd_apply: bad arguments");
    PULPO_REQUIRE_DT(dz_dt, "bn_lrelu_bwd_apply"); PULPO_REQUIRE_DT(y_dt, "bn_lrelu_bwd_apply");
    PULPO_DISPATCH_DT(dz_dt, TG, PULPO_DISPATCH_DT(y_dt, TY,
        return bwd_apply_t((const TG*)dz, (long)dzps, (const TY*)y, (long)yps, coef, totd, (TY*)dy, (long)dyps, (long)npix, C, slope, partial2,
                           (hipStream_t)stream)));
    return -1;
}

PULPO_API int pulpo_bn_lrelu_bwd_apply(const float* dz, int64_t dzps, const float* y, int64_t yps, const float* coef, const double* totd, float* dy,
                                       int64_t dyps, int64_t npix, int C, float slope, float* partial2, void* stream) {
    return pulpo_bn_lrelu_bwd_apply_t(dz, 0, dzps, y, 0, yps, coef, totd, dy, dyps, npix, C, slope, partial2, stream);
}

// The second pass with its result in the channel-BLOCKED layout: dy channel c of pixel p at dy + (c / 8) * dykb + p * dyps + c % 8 (dyps = 8,
// dykb = npix * 8: [C / 8][pixels][8]).  dy is read by nothing but the data- and the weight-gradient convolution of the unit; in this layout the
// four taps of a staging item of the F(2x2x2,3x3x3) data-gradient kernel are 128 consecutive bytes (pulpo_conv3d_k3_fwd_wino3_kb: 15 % faster
// at 160^3).  fp32 dy, C % 8 == 0.  The pooled form: as pulpo_bn_lrelu_bwd_apply_pooled_t.
PULPO_API int pulpo_bn_lrelu_bwd_apply_kb_t(const void* dz, int dz_dt, int64_t dzps, int64_t dzkb, const float* y, int64_t yps, const float* coef,
                                            const double* totd, float* dy, int64_t dyps, int64_t dykb, int64_t npix, int C, float slope, float* partial2,
                                            void* stream) {
    PULPO_REQUIRE(dz && y && coef && totd && dy && partial2 && npix > 0 && C > 0, "bn_lrelu_bwd_apply_kb: bad arguments");
    PULPO_REQUIRE_DT(dz_dt, "bn_lrelu_bwd_apply_kb");
    PULPO_REQUIRE(C % 8 == 0 && dyps % 4 == 0 && dyps >= 8 && dykb % 4 == 0 && dykb >= 8 && dzkb % 4 == 0 && dzkb >= 8 && (dzkb == 8 || dz_dt == 0),
                  "bn_lrelu_bwd_apply_kb: C %% 8 == 0, strides in whole four-channel groups, a blocked dz is fp32");
    PULPO_DISPATCH_DT(dz_dt, TG,
        return bwd_apply_t((const TG*)dz, (long)dzps, y, (long)yps, coef, totd, dy, (long)dyps, (long)npix, C, slope, partial2, (hipStream_t)stream, (long)dykb,
                           (long)dzkb));
    return -1;
}

PULPO_API int pulpo_bn_lrelu_bwd_apply_pooled_kb_t(const void* gout, int64_t gops, const void* add, int64_t aps, int g_dt, const float* y, int64_t yps,
                                                   const float* coef, const double* totd, float* dy, int64_t dyps, int64_t dykb, float slope,
                                                   float* partial2, int B, int D, int H, int W, int C, void* stream) {
    PULPO_REQUIRE(gout && y && coef && totd && dy && partial2 && B > 0 && D > 0 && H > 0 && W > 0 && C > 0, "bn_lrelu_bwd_apply_pooled_kb: bad arguments");
    PULPO_REQUIRE_DT(g_dt, "bn_lrelu_bwd_apply_pooled_kb");
    const int eg = g_dt ? 8 : 16;
    PULPO_REQUIRE(C % 8 == 0 && C / 4 <= 256 && gops % 4 == 0 && yps % 4 == 0 && dyps % 4 == 0 && dyps >= 8 && dykb % 4 == 0 && dykb >= 8 && (add == nullptr || aps % 4 == 0) &&
                      ((((uintptr_t)gout) | ((uintptr_t)add)) % eg) == 0 && ((((uintptr_t)y) | ((uintptr_t)dy)) & 15) == 0 && (((uintptr_t)coef) & 15) == 0,
                  "bn_lrelu_bwd_apply_pooled_kb: operands must be aligned to four elements, C %% 8 == 0");
    PULPO_DISPATCH_DT(g_dt, TG,
        return pool_apply_t((const TG*)gout, (long)gops, (const TG*)add, (long)aps, y, (long)yps, coef, totd, dy, (long)dyps, slope, partial2, B, D, H, W, C,
                            (hipStream_t)stream, (long)dykb));
    return -1;
}
