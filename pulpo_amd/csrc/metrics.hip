// Alternative reconstruction losses / regulariser / evaluation metrics of PULPo:
//   L2_loss (src/losses.py:79-83), Soft_dice_loss (:137-145), jacobian_det (:172-199), JDetStd (:202-204).
// All are HBM-bound streaming kernels with two-stage deterministic reductions (fp32 block partials -> double).
#include "common.h"

namespace {

inline int eblocks(long items, int cap = 4096) { return (int)std::max<long>(1, std::min<long>((items + 255) / 256, cap)); }

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = pulpo::wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

// ------------------------------------------------------------------------------------------------ L2_loss
__global__ __launch_bounds__(256) void sqdiff_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float local = 0.f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float d = a[e] - b[e];
        local += d * d;
    }
    const float t = block_sum_256(local, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void sqdiff_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gscale,
                                                           float coef, float* __restrict__ ga, long n) {
    const float k0 = 2.f * coef * (gscale != nullptr ? gscale[0] : 1.f);
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) ga[e] = k0 * (a[e] - b[e]);
}

// ------------------------------------------------------------------------------------------------ Soft dice
// grid (nb, nplanes): partial[(plane*nb + blk)*3 + {0,1,2}] = sum t*i, sum t*t, sum i*i over the block's voxels of that plane
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ inp, const float* __restrict__ tgt, long V, float* __restrict__ partial) {
    __shared__ float sh[4];
    const long base = (long)blockIdx.y * V;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < V; e += (long)gridDim.x * blockDim.x) {
        const float i = inp[base + e], t = tgt[base + e];
        s0 += t * i; s1 += t * t; s2 += i * i;
    }
    float* dst = partial + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 3;
    float t0 = block_sum_256(s0, sh), t1 = block_sum_256(s1, sh), t2 = block_sum_256(s2, sh);
    if (threadIdx.x == 0) { dst[0] = t0; dst[1] = t1; dst[2] = t2; }
}

// one block: numden[plane][2] = (2 s0 + eps, s1 + s2 + eps) as doubles; loss = mean(1 - num/den) * V / dice_factor
__global__ void dice_finalize_kernel(const float* __restrict__ partial, int nplanes, int nb, double V, double dice_factor, double* __restrict__ numden,
                                     float* __restrict__ loss) {
    __shared__ double acc;
    if (threadIdx.x == 0) acc = 0.0;
    __syncthreads();
    for (int p = threadIdx.x; p < nplanes; p += blockDim.x) {
        double s0 = 0, s1 = 0, s2 = 0;
        for (int k = 0; k < nb; ++k) {
            const float* q = partial + ((long)p * nb + k) * 3;
            s0 += q[0]; s1 += q[1]; s2 += q[2];
        }
        const double num = 2.0 * s0 + 1e-6, den = s1 + s2 + 1e-6;
        numden[2 * p] = num; numden[2 * p + 1] = den;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                       // (the planes' terms in plane order: a shared double atomic summed them in arrival order)
        for (int p = 0; p < nplanes; ++p) acc += 1.0 - numden[2 * p] / numden[2 * p + 1];
        loss[0] = (float)(acc / nplanes * V / dice_factor);
    }
}

// d loss / d inp = -(V / (dice_factor * nplanes)) * (2 t den - 2 i num) / den^2
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ inp, const float* __restrict__ tgt, const double* __restrict__ numden,
                                                         const float* __restrict__ gscale, float coef, float* __restrict__ ginp, long V) {
    const double num = numden[2 * blockIdx.y], den = numden[2 * blockIdx.y + 1];
    const float k0 = coef * (gscale != nullptr ? gscale[0] : 1.f);
    const float a = (float)(2.0 / den), b = (float)(2.0 * num / (den * den));
    const long base = (long)blockIdx.y * V;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < V; e += (long)gridDim.x * blockDim.x)
        ginp[base + e] = -k0 * (tgt[base + e] * a - inp[base + e] * b);
}

// ------------------------------------------------------------------------------------------------ Jacobian determinant
struct JacGeom {
    int B, D, H, W;
    float pre[3];     // 2/S_i when normalising, else 1: applied to ORIGINAL channel i
    float post[3];    // ((D-1,H-1,W-1)[c] - 1)/2 applied to FLIPPED channel c (= original channel 2-c)
    // D == 1: the reference's 2-D form (losses.py:153-170) on a 2-channel field (B,2,H,W): same recipe over (H, W)
    float pre2[2], post2[2];
};

// 2-D: J[a][c], a in (y, x), flipped channel c reads original channel 1-c; returns the 2x2 determinant
__device__ __forceinline__ float jac2_at(const float* __restrict__ df, const JacGeom& g, long b, int y, int x, float (*J)[2]) {
    const long V = (long)g.H * g.W;
    const int S[2] = {g.H, g.W};
    const int p[2] = {y, x};
    const long st[2] = {g.W, 1};
    const long v = (long)y * g.W + x;
    float Jl[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float* u = df + (b * 2 + (1 - c)) * V;
        const float sc = g.pre2[1 - c] * g.post2[c];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const long ip = v + (p[a] + 1 < S[a] ? st[a] : 0), im = v - (p[a] > 0 ? st[a] : 0);
            Jl[a][c] = 0.5f * sc * (u[ip] - u[im]) + (a == c ? 1.f : 0.f);
        }
    }
    if (J != nullptr) { J[0][0] = Jl[0][0]; J[0][1] = Jl[0][1]; J[1][0] = Jl[1][0]; J[1][1] = Jl[1][1]; }
    return Jl[0][0] * Jl[1][1] - Jl[1][0] * Jl[0][1];
}

// J[a][c] at voxel (z,y,x); returns the determinant, optionally the 9 entries
__device__ __forceinline__ float jac_at(const float* __restrict__ df, const JacGeom& g, long b, int z, int y, int x, float (*J)[3]) {
    const long V = (long)g.D * g.H * g.W, sz = (long)g.H * g.W, sy = g.W;
    const int S[3] = {g.D, g.H, g.W};
    const int p[3] = {z, y, x};
    const long st[3] = {sz, sy, 1};
    const long v = (long)z * sz + (long)y * sy + x;
    float Jl[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* u = df + (b * 3 + (2 - c)) * V;        // flipped channel c reads original channel 2-c
        const float sc = g.pre[2 - c] * g.post[c];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const long ip = v + (p[a] + 1 < S[a] ? st[a] : 0), im = v - (p[a] > 0 ? st[a] : 0);   // replicate padding
            Jl[a][c] = 0.5f * sc * (u[ip] - u[im]) + (a == c ? 1.f : 0.f);
        }
    }
    if (J != nullptr)
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) J[a][c] = Jl[a][c];
    return Jl[0][0] * (Jl[1][1] * Jl[2][2] - Jl[2][1] * Jl[1][2]) - Jl[0][1] * (Jl[1][0] * Jl[2][2] - Jl[2][0] * Jl[1][2]) +
           Jl[0][2] * (Jl[1][0] * Jl[2][1] - Jl[2][0] * Jl[1][1]);
}

__global__ __launch_bounds__(256) void jacdet_fwd_kernel(const float* __restrict__ df, JacGeom g, float* __restrict__ out, float* __restrict__ partial) {
    __shared__ float sh[4];
    const long V = (long)g.D * g.H * g.W, total = (long)g.B * V;
    float s = 0.f, q = 0.f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long b = e / V, v = e - b * V;
        const int x = (int)(v % g.W), y = (int)((v / g.W) % g.H), z = (int)(v / ((long)g.W * g.H));
        const float d = g.D == 1 ? jac2_at(df, g, b, y, x, nullptr) : jac_at(df, g, b, z, y, x, nullptr);
        out[e] = d;
        s += d; q += d * d;
    }
    if (partial != nullptr) {
        const float ts = block_sum_256(s, sh), tq = block_sum_256(q, sh);
        if (threadIdx.x == 0) { partial[2 * blockIdx.x] = ts; partial[2 * blockIdx.x + 1] = tq; }
    }
}

// stat[0] = mean, stat[1] = unbiased std (doubles); loss = lamb * std
__global__ void jdetstd_finalize_kernel(const float* __restrict__ partial, int nblk, double n, float lamb, double* __restrict__ stat, float* __restrict__ loss) {
    double s = 0, q = 0;
    for (int k = 0; k < nblk; ++k) { s += partial[2 * k]; q += partial[2 * k + 1]; }
    const double mean = s / n;
    double var = (q - s * mean) / (n - 1.0);
    if (var < 0) var = 0;
    stat[0] = mean; stat[1] = sqrt(var);
    loss[0] = (float)(lamb * sqrt(var));
}

// d (lamb*std) / d df: per voxel w = lamb (j - mean) / ((n-1) std); cofactors of J scatter +-0.5*scale*w to the two neighbours (atomics)
__global__ __launch_bounds__(256) void jdetstd_bwd_kernel(const float* __restrict__ df, JacGeom g, const float* __restrict__ jdet,
                                                            const double* __restrict__ stat, const float* __restrict__ gscale, float lamb, double n,
                                                            float* __restrict__ gdf) {
    const long V = (long)g.D * g.H * g.W, total = (long)g.B * V, sz = (long)g.H * g.W, sy = g.W;
    const double mean = stat[0], sd = stat[1];
    const float k0 = sd > 0 ? (float)(lamb / ((n - 1.0) * sd)) * (gscale != nullptr ? gscale[0] : 1.f) : 0.f;
    const int S[3] = {g.D, g.H, g.W};
    const long st[3] = {sz, sy, 1};
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long b = e / V, v = e - b * V;
        const int x = (int)(v % g.W), y = (int)((v / g.W) % g.H), z = (int)(v / sz);
        const int p[3] = {z, y, x};
        const float w = k0 * (float)((double)jdet[e] - mean);
        if (g.D == 1) {                  // 2-D form: 2x2 cofactors, two channels
            float J2[2][2];
            jac2_at(df, g, b, y, x, J2);
            const float Cf2[2][2] = {{J2[1][1], -J2[1][0]}, {-J2[0][1], J2[0][0]}};
            const int S2[2] = {g.H, g.W};
            const int p2[2] = {y, x};
            const long st2[2] = {sy, 1};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float* gu = gdf + (b * 2 + (1 - c)) * V;
                const float sc = 0.5f * g.pre2[1 - c] * g.post2[c] * w;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const long ip = v + (p2[a] + 1 < S2[a] ? st2[a] : 0), im = v - (p2[a] > 0 ? st2[a] : 0);
                    atomicAdd(gu + ip, sc * Cf2[a][c]);
                    atomicAdd(gu + im, -sc * Cf2[a][c]);
                }
            }
            continue;
        }
        float J[3][3];
        jac_at(df, g, b, z, y, x, J);
        // cofactor matrix: d det / d J[a][c]
        float Cf[3][3];
        Cf[0][0] = J[1][1] * J[2][2] - J[2][1] * J[1][2];
        Cf[0][1] = -(J[1][0] * J[2][2] - J[2][0] * J[1][2]);
        Cf[0][2] = J[1][0] * J[2][1] - J[2][0] * J[1][1];
        Cf[1][0] = -(J[0][1] * J[2][2] - J[0][2] * J[2][1]);
        Cf[1][1] = J[0][0] * J[2][2] - J[0][2] * J[2][0];
        Cf[1][2] = -(J[0][0] * J[2][1] - J[0][1] * J[2][0]);
        Cf[2][0] = J[0][1] * J[1][2] - J[0][2] * J[1][1];
        Cf[2][1] = -(J[0][0] * J[1][2] - J[0][2] * J[1][0]);
        Cf[2][2] = J[0][0] * J[1][1] - J[0][1] * J[1][0];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float* gu = gdf + (b * 3 + (2 - c)) * V;
            const float sc = 0.5f * g.pre[2 - c] * g.post[c] * w;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const long ip = v + (p[a] + 1 < S[a] ? st[a] : 0), im = v - (p[a] > 0 ? st[a] : 0);
                atomicAdd(gu + ip, sc * Cf[a][c]);
                atomicAdd(gu + im, -sc * Cf[a][c]);
            }
        }
    }
}

JacGeom make_geom(int B, int D, int H, int W, int normalize) {
    JacGeom g;
    g.B = B; g.D = D; g.H = H; g.W = W;
    const int S[3] = {D, H, W};
    for (int i = 0; i < 3; ++i) {
        g.pre[i] = normalize ? 2.f / (float)S[i] : 1.f;
        g.post[i] = ((float)(S[i] - 1) - 1.f) / 2.f;
    }
    for (int i = 0; i < 2; ++i) {
        g.pre2[i] = normalize ? 2.f / (float)S[i + 1] : 1.f;
        g.post2[i] = ((float)(S[i + 1] - 1) - 1.f) / 2.f;
    }
    return g;
}

}  // namespace

PULPO_API int pulpo_metric_blocks(int64_t n) { return eblocks(n, 1024); }

// L2_loss: partial[pulpo_metric_blocks(n)]; finish with pulpo_colsum(scale = 1/(B*C))
PULPO_API int pulpo_sqdiff_fwd(const float* a, const float* b, int64_t n, float* partial, void* stream) {
    PULPO_REQUIRE(a && b && partial && n > 0, "sqdiff_fwd: bad arguments");
    hipLaunchKernelGGL(sqdiff_fwd_kernel, dim3(pulpo_metric_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, (long)n, partial);
    return pulpo::check_launch("sqdiff_fwd");
}
PULPO_API int pulpo_sqdiff_bwd(const float* a, const float* b, const float* gscale, float coef, float* ga, int64_t n, void* stream) {
    PULPO_REQUIRE(a && b && ga && n > 0, "sqdiff_bwd: bad arguments");
    hipLaunchKernelGGL(sqdiff_bwd_kernel, dim3(eblocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, gscale, coef, ga, (long)n);
    return pulpo::check_launch("sqdiff_bwd");
}

// Soft_dice_loss: nplanes = B*C planes of V voxels. partial: nplanes*pulpo_dice_blocks(V)*3 floats; numden: 2*nplanes doubles (kept for backward)
PULPO_API int pulpo_dice_blocks(int64_t V) { return eblocks(V, 256); }
PULPO_API int pulpo_dice_fwd(const float* inp, const float* tgt, int nplanes, int64_t V, float dice_factor, float* partial, double* numden, float* loss,
                             void* stream) {
    PULPO_REQUIRE(inp && tgt && partial && numden && loss && nplanes > 0 && V > 0 && dice_factor > 0, "dice_fwd: bad arguments");
    const int nb = pulpo_dice_blocks(V);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dice_sums_kernel, dim3(nb, nplanes), dim3(256), 0, st, inp, tgt, (long)V, partial);
    int rc = pulpo::check_launch("dice_sums");
    if (rc) return rc;
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, st, partial, nplanes, nb, (double)V, (double)dice_factor, numden, loss);
    return pulpo::check_launch("dice_finalize");
}
PULPO_API int pulpo_dice_bwd(const float* inp, const float* tgt, const double* numden, const float* gscale, int nplanes, int64_t V, float dice_factor,
                             float* ginp, void* stream) {
    PULPO_REQUIRE(inp && tgt && numden && ginp && nplanes > 0 && V > 0, "dice_bwd: bad arguments");
    const float coef = (float)((double)V / ((double)dice_factor * nplanes));
    hipLaunchKernelGGL(dice_bwd_kernel, dim3(pulpo_dice_blocks(V), nplanes), dim3(256), 0, (hipStream_t)stream, inp, tgt, numden, gscale, coef, ginp, (long)V);
    return pulpo::check_launch("dice_bwd");
}

// jacobian_det: df planar (B,3,D,H,W) -> out (B,D,H,W).  partial (nullable): 2*pulpo_metric_blocks(B*D*H*W) floats of (sum, sum sq) for JDetStd
PULPO_API int pulpo_jacdet_fwd(const float* df, float* out, float* partial, int B, int D, int H, int W, int normalize, void* stream) {
    PULPO_REQUIRE(df && out && B > 0 && D > 0 && H > 0 && W > 0, "jacdet_fwd: bad arguments");
    const long n = (long)B * D * H * W;
    hipLaunchKernelGGL(jacdet_fwd_kernel, dim3(pulpo_metric_blocks(n)), dim3(256), 0, (hipStream_t)stream, df, make_geom(B, D, H, W, normalize), out, partial);
    return pulpo::check_launch("jacdet_fwd");
}
// JDetStd = lamb * std(jacobian_det): stat = (mean, std) doubles kept for backward
PULPO_API int pulpo_jdetstd_finalize(const float* partial, int64_t n, float lamb, double* stat, float* loss, void* stream) {
    PULPO_REQUIRE(partial && stat && loss && n > 1, "jdetstd_finalize: bad arguments");
    hipLaunchKernelGGL(jdetstd_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, partial, pulpo_metric_blocks(n), (double)n, lamb, stat, loss);
    return pulpo::check_launch("jdetstd_finalize");
}
PULPO_API int pulpo_jdetstd_bwd(const float* df, const float* jdet, const double* stat, const float* gscale, float lamb, float* gdf, int B, int D, int H,
                                int W, int normalize, void* stream) {
    PULPO_REQUIRE(df && jdet && stat && gdf && B > 0 && D > 0 && H > 0 && W > 0, "jdetstd_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)B * D * H * W;
    hipError_t e = hipMemsetAsync(gdf, 0, sizeof(float) * (D == 1 ? 2 : 3) * n, st);        // D == 1: two-channel 2-D field
    if (e != hipSuccess) return pulpo::fail((int)e, "jdetstd_bwd memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(jdetstd_bwd_kernel, dim3(eblocks(n)), dim3(256), 0, st, df, make_geom(B, D, H, W, normalize), jdet, stat, gscale, lamb, (double)n, gdf);
    return pulpo::check_launch("jdetstd_bwd");
}

// ------------------------------------------------------------------------------------------------ evaluation scalars (evaluate.py)
// rmse (evaluate.py:315-319), dsc (:321-327), % of voxels with |J| <= 0 (:1441-1446), landmark warp (:410-423 = src/components/utils.py:15-25)
namespace {

__global__ void rmse_finalize_kernel(const float* __restrict__ partial, int nblk, double n, float* __restrict__ out) {
    double s = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) s += partial[k];
    s = pulpo::wave_sum_d(s);
    if (threadIdx.x == 0) out[0] = (float)sqrt(s / n);
}

// dsc = mean over planes of ((2 t i).mean + 1e-6) / ((t^2).mean + (i^2).mean + 1e-6), means over the plane's V voxels
__global__ void dsc_finalize_kernel(const float* __restrict__ partial, int nplanes, int nb, double V, float* __restrict__ out) {
    double acc = 0.0;
    for (int p = threadIdx.x; p < nplanes; p += 64) {
        double s0 = 0, s1 = 0, s2 = 0;
        for (int k = 0; k < nb; ++k) {
            const float* q = partial + ((long)p * nb + k) * 3;
            s0 += q[0]; s1 += q[1]; s2 += q[2];
        }
        acc += (2.0 * s0 / V + 1e-6) / (s1 / V + s2 / V + 1e-6);
    }
    acc = pulpo::wave_sum_d(acc);
    if (threadIdx.x == 0) out[0] = (float)(acc / nplanes);
}

__global__ __launch_bounds__(256) void count_leq0_kernel(const float* __restrict__ x, long n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float local = 0.f;                               // <= 2^24 elements per thread: exact in fp32
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) local += x[e] <= 0.f ? 1.f : 0.f;
    const float t = block_sum_256(local, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void percent_finalize_kernel(const float* __restrict__ partial, int nblk, double n, float* __restrict__ out) {
    double s = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) s += partial[k];
    s = pulpo::wave_sum_d(s);
    if (threadIdx.x == 0) out[0] = (float)(s / n) * 100.f;          // (count / numel) * 100 in fp32 like the reference's tensor expression
}

// out[s][k][c] = trunc(lm[k][c]) - df[s][c][i0][i1][i2], i = trunc(lm[k]) with Python-style negative wrap; flag[0] = 1 if any index is out of range
__global__ void warp_landmarks_kernel(const float* __restrict__ lm, const float* __restrict__ df, float* __restrict__ out, int nlm, int nsamp, int nd,
                                      int D, int H, int W, int* __restrict__ flag) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nlm * nsamp) return;
    const int k = t % nlm, s = t / nlm;
    const int S[3] = {D, H, W};
    long idx[3] = {0, 0, 0};
    float base[3];
    bool ok = true;
    for (int c = 0; c < nd; ++c) {
        const long i = (long)lm[(long)k * nd + c];            // .long(): truncation toward zero
        base[c] = (float)i;
        const int dim = nd == 3 ? c : c + 1;                    // 2-D fields are (B,2,H,W): D == 1
        long j = i < 0 ? i + S[dim] : i;
        if (j < 0 || j >= S[dim]) { ok = false; j = 0; }
        idx[dim] = j;
    }
    if (!ok) atomicOr(flag, 1);
    const long V = (long)D * H * W, v = (idx[0] * H + idx[1]) * W + idx[2];
    for (int c = 0; c < nd; ++c)
        out[((long)s * nlm + k) * nd + c] = ok ? base[c] - df[((long)s * nd + c) * V + v] : __builtin_nanf("");
}

}  // namespace

// rmse = sqrt(mean((a - b)^2)) over all n elements; partial: pulpo_metric_blocks(n) floats; out: 1 float
PULPO_API int pulpo_rmse(const float* a, const float* b, int64_t n, float* partial, float* out, void* stream) {
    PULPO_REQUIRE(a && b && partial && out && n > 0, "rmse: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = pulpo_metric_blocks(n);
    hipLaunchKernelGGL(sqdiff_fwd_kernel, dim3(nblk), dim3(256), 0, st, a, b, (long)n, partial);
    int rc = pulpo::check_launch("rmse sums");
    if (rc) return rc;
    hipLaunchKernelGGL(rmse_finalize_kernel, dim3(1), dim3(64), 0, st, partial, nblk, (double)n, out);
    return pulpo::check_launch("rmse finalize");
}
// dice similarity coefficient of evaluate.py:321-327; nplanes = B*C planes of V voxels; partial: nplanes*pulpo_dice_blocks(V)*3 floats
PULPO_API int pulpo_dsc(const float* inp, const float* tgt, int nplanes, int64_t V, float* partial, float* out, void* stream) {
    PULPO_REQUIRE(inp && tgt && partial && out && nplanes > 0 && V > 0, "dsc: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nb = pulpo_dice_blocks(V);
    hipLaunchKernelGGL(dice_sums_kernel, dim3(nb, nplanes), dim3(256), 0, st, inp, tgt, (long)V, partial);
    int rc = pulpo::check_launch("dsc sums");
    if (rc) return rc;
    hipLaunchKernelGGL(dsc_finalize_kernel, dim3(1), dim3(64), 0, st, partial, nplanes, nb, (double)V, out);
    return pulpo::check_launch("dsc finalize");
}
// 100 * count(x <= 0) / n (the JDetLeq0 metric on a Jacobian-determinant map); partial: pulpo_metric_blocks(n) floats
PULPO_API int pulpo_percent_leq0(const float* x, int64_t n, float* partial, float* out, void* stream) {
    PULPO_REQUIRE(x && partial && out && n > 0, "percent_leq0: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = pulpo_metric_blocks(n);
    hipLaunchKernelGGL(count_leq0_kernel, dim3(nblk), dim3(256), 0, st, x, (long)n, partial);
    int rc = pulpo::check_launch("percent_leq0 count");
    if (rc) return rc;
    hipLaunchKernelGGL(percent_finalize_kernel, dim3(1), dim3(64), 0, st, partial, nblk, (double)n, out);
    return pulpo::check_launch("percent_leq0 finalize");
}
// lm: (nlm, nd) landmark coordinates (floats, truncated like .long()); df: (nsamp, nd, D, H, W) planar (nd == 2: D must be 1);
// out: (nsamp, nlm, nd); flag: one int, zeroed here, set to 1 when a landmark indexes outside the field (the reference raises IndexError)
PULPO_API int pulpo_warp_landmarks(const float* lm, const float* df, float* out, int nlm, int nsamp, int nd, int D, int H, int W, int* flag,
                                   void* stream) {
    PULPO_REQUIRE(lm && df && out && flag && nlm > 0 && nsamp > 0 && (nd == 3 || (nd == 2 && D == 1)) && D > 0 && H > 0 && W > 0,
                  "warp_landmarks: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), st);
    if (e != hipSuccess) return pulpo::fail((int)e, "warp_landmarks memset: %s", hipGetErrorString(e));
    const int n = nlm * nsamp;
    hipLaunchKernelGGL(warp_landmarks_kernel, dim3((n + 127) / 128), dim3(128), 0, st, lm, df, out, nlm, nsamp, nd, D, H, W, flag);
    return pulpo::check_launch("warp_landmarks");
}

// ------------------------------------------------------------------------------------------------ KL_nondiagonal
// src/losses.py:8-44.  loss = (mean(lambda*D*sigma^2 - log sigma^2) + lambda/2 * (0.5/3) * sum_axes mean(fwd diff of mu)^2) * 3 * 0.5 * V
// D(v) = number of in-volume voxels of the 3x3x3 neighbourhood minus one.
namespace {

__device__ __forceinline__ float degree_of(int z, int y, int x, int D, int H, int W) {
    const int nz = min(z + 1, D - 1) - max(z - 1, 0) + 1, ny = min(y + 1, H - 1) - max(y - 1, 0) + 1, nx = min(x + 1, W - 1) - max(x - 1, 0) + 1;
    return (float)(nz * ny * nx - 1);
}

// partial[blk][4] = sum sigma-term, sum dz^2, sum dy^2, sum dx^2
__global__ __launch_bounds__(256) void kln_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ sigma, long nplanes, int D, int H, int W,
                                                        float lambda, float* __restrict__ partial) {
    __shared__ float sh[4];
    const long V = (long)D * H * W, total = nplanes * V, sz = (long)H * W;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long v = e % V;
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / sz);
        const float sg2 = sigma[e] * sigma[e], m = mu[e];
        s0 += lambda * degree_of(z, y, x, D, H, W) * sg2 - logf(sg2);
        if (z >= 1) { const float d = m - mu[e - sz]; s1 += d * d; }
        if (y >= 1) { const float d = m - mu[e - W]; s2 += d * d; }
        if (x >= 1) { const float d = m - mu[e - 1]; s3 += d * d; }
    }
    const float t0 = block_sum_256(s0, sh), t1 = block_sum_256(s1, sh), t2 = block_sum_256(s2, sh), t3 = block_sum_256(s3, sh);
    if (threadIdx.x == 0) { float* p = partial + 4 * blockIdx.x; p[0] = t0; p[1] = t1; p[2] = t2; p[3] = t3; }
}

__global__ void kln_finalize_kernel(const float* __restrict__ partial, int nblk, double nplanes, int D, int H, int W, double lambda, float* __restrict__ loss) {
    double s[4] = {0, 0, 0, 0};
    for (int k = 0; k < nblk; ++k)
        for (int j = 0; j < 4; ++j) s[j] += partial[4 * k + j];
    const double V = (double)D * H * W;
    const double cnt[3] = {nplanes * (D - 1) * H * W, nplanes * D * (H - 1) * W, nplanes * D * H * (W - 1)};
    const double nd = D == 1 ? 2.0 : 3.0;                 // D == 1: the reference's 2-D form (no depth axis)
    const double precision = 0.5 * ((D == 1 ? 0.0 : s[1] / cnt[0]) + s[2] / cnt[1] + s[3] / cnt[2]) / nd;
    loss[0] = (float)((s[0] / (nplanes * V) + lambda / 2 * precision) * nd * 0.5 * V);
}

__global__ __launch_bounds__(256) void kln_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ sigma, const float* __restrict__ gscale,
                                                        long nplanes, int D, int H, int W, float lambda, float* __restrict__ gmu, float* __restrict__ gsigma) {
    const long V = (long)D * H * W, total = nplanes * V, sz = (long)H * W;
    const float g = gscale != nullptr ? gscale[0] : 1.f;
    const float nd = D == 1 ? 2.f : 3.f;
    const float outer = nd * 0.5f * (float)V * g;                   // ndims * 0.5 * V
    const float ks = outer / (float)(nplanes * V);
    const float kp = outer * (lambda * 0.5f) * (0.5f / nd) * 2.f;       // d(diff^2) = 2 diff
    const float cz = D == 1 ? 0.f : kp / (float)(nplanes * (D - 1) * (long)H * W), cy = kp / (float)(nplanes * D * (long)(H - 1) * W),
                cx = kp / (float)(nplanes * D * (long)H * (W - 1));
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long v = e % V;
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / sz);
        const float sg = sigma[e], m = mu[e];
        gsigma[e] = ks * (2.f * lambda * degree_of(z, y, x, D, H, W) * sg - 2.f / sg);
        float gm = 0.f;
        if (z >= 1) gm += cz * (m - mu[e - sz]);
        if (z + 1 < D) gm -= cz * (mu[e + sz] - m);
        if (y >= 1) gm += cy * (m - mu[e - W]);
        if (y + 1 < H) gm -= cy * (mu[e + W] - m);
        if (x >= 1) gm += cx * (m - mu[e - 1]);
        if (x + 1 < W) gm -= cx * (mu[e + 1] - m);
        gmu[e] = gm;
    }
}

}  // namespace

// mu, sigma planar (B,3,D,H,W); nplanes = B*3; partial: 4*pulpo_metric_blocks(nplanes*D*H*W) floats
PULPO_API int pulpo_kl_nondiag_fwd(const float* mu, const float* sigma, int64_t nplanes, int D, int H, int W, float prior_lambda, float* partial,
                                   float* loss, void* stream) {
    PULPO_REQUIRE(mu && sigma && partial && loss && nplanes > 0 && D >= 1 && H > 1 && W > 1, "kl_nondiag_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = pulpo_metric_blocks(nplanes * D * H * W);
    hipLaunchKernelGGL(kln_fwd_kernel, dim3(nblk), dim3(256), 0, st, mu, sigma, (long)nplanes, D, H, W, prior_lambda, partial);
    int rc = pulpo::check_launch("kl_nondiag_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(kln_finalize_kernel, dim3(1), dim3(1), 0, st, partial, nblk, (double)nplanes, D, H, W, (double)prior_lambda, loss);
    return pulpo::check_launch("kl_nondiag_finalize");
}
PULPO_API int pulpo_kl_nondiag_bwd(const float* mu, const float* sigma, const float* gscale, int64_t nplanes, int D, int H, int W, float prior_lambda,
                                   float* gmu, float* gsigma, void* stream) {
    PULPO_REQUIRE(mu && sigma && gmu && gsigma && nplanes > 0 && D >= 1 && H > 1 && W > 1, "kl_nondiag_bwd: bad arguments");
    hipLaunchKernelGGL(kln_bwd_kernel, dim3(eblocks(nplanes * D * H * W)), dim3(256), 0, (hipStream_t)stream, mu, sigma, gscale, (long)nplanes, D, H, W,
                       prior_lambda, gmu, gsigma);
    return pulpo::check_launch("kl_nondiag_bwd");
}
