// Loss stack: local normalised cross-correlation, diagonal-Gaussian KL against N(0,1), gradient-L2 regulariser.
// Reference: src/losses.py:85-135 (NCC_loss), :47-76 (KL_two_gauss_with_diag_cov), :208-222 (L2_reg).
//
// NCC: the reference evaluates five dense w^3 ones-kernel convolutions (zero padded).  Here the box sums are
// separable: the W pass holds 64 consecutive voxels of a row in a wavefront and forms the window sum with
// wave shuffles (no LDS), the H and D passes are strided streaming sums.  All passes are HBM/L2 bound.
// Every reduction is two-stage (per-block partial -> pulpo_colsum in double), hence deterministic.
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
    return t;   // valid in thread 0
}

// ------------------------------------------------------------------------------------------------ box sums
// W-axis pass.  MODE 0: in = (I, J) -> out = box_x of (I, J, I*I, J*J, I*J) [5 channels, planar stride N]
//               MODE 1: in = nch planes (stride N) -> out = box_x of each
// One wave covers 64 consecutive x of one row starting at x0 - pad; lanes pad .. 63-pad produce outputs.
template <int MODE>
__global__ __launch_bounds__(256) void box_x_kernel(const float* __restrict__ in0, const float* __restrict__ in1, float* __restrict__ out, long N,
                                                      long nrows, int W, int pad, int nch, int segs_per_row) {
    const int lane = threadIdx.x & 63;
    const int span = 64 - 2 * pad;
    const long nwork = nrows * segs_per_row;
    const long wave0 = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long nwave = ((long)gridDim.x * blockDim.x) >> 6;
    for (long wk = wave0; wk < nwork; wk += nwave) {
        const long row = wk / segs_per_row;
        const int seg = (int)(wk - row * segs_per_row);
        const int x = seg * span - pad + lane;
        const bool inb = x >= 0 && x < W;
        const long idx = row * W + x;
        const bool owner = lane >= pad && lane < 64 - pad && x < W;
        if constexpr (MODE == 0) {
            const float I = inb ? in0[idx] : 0.f, J = inb ? in1[idx] : 0.f;
            float v[5] = {I, J, I * I, J * J, I * J};
            float s[5] = {v[0], v[1], v[2], v[3], v[4]};
            for (int k = 1; k <= pad; ++k) {
#pragma unroll
                for (int c = 0; c < 5; ++c) s[c] += __shfl(v[c], lane - k, 64) + __shfl(v[c], lane + k, 64);
            }
            if (owner) {
#pragma unroll
                for (int c = 0; c < 5; ++c) out[c * N + idx] = s[c];
            }
        } else {
            for (int c = 0; c < nch; ++c) {
                const float v = inb ? in0[c * N + idx] : 0.f;
                float s = v;
                for (int k = 1; k <= pad; ++k) s += __shfl(v, lane - k, 64) + __shfl(v, lane + k, 64);
                if (owner) out[c * N + idx] = s;
            }
        }
    }
}

// strided axis pass (H: stride W, extent H ; D: stride H*W, extent D) over nch planar channels
__global__ __launch_bounds__(256) void box_axis_kernel(const float* __restrict__ in, float* __restrict__ out, long N, int nch, int extent, long stride,
                                                         int pad) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < N; e += (long)gridDim.x * blockDim.x) {
        const int pos = (int)((e / stride) % extent);
        const int lo = max(-pad, -pos), hi = min(pad, extent - 1 - pos);
        for (int c = 0; c < nch; ++c) {
            const float* s = in + c * N + e;
            float acc = 0.f;
            for (int k = lo; k <= hi; ++k) acc += s[k * stride];
            out[c * N + e] = acc;
        }
    }
}

// last (D) pass fused with the correlation coefficient and its block reduction.  S (5 channels) is kept for backward.
__global__ __launch_bounds__(256) void ncc_final_kernel(const float* __restrict__ in, float* __restrict__ S, long N, int extent, long stride, int pad,
                                                          float nwin, float* __restrict__ partial) {
    __shared__ float sh[4];
    float local = 0.f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < N; e += (long)gridDim.x * blockDim.x) {
        const int pos = (int)((e / stride) % extent);
        const int lo = max(-pad, -pos), hi = min(pad, extent - 1 - pos);
        float s[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const float* q = in + c * N + e;
            float acc = 0.f;
            for (int k = lo; k <= hi; ++k) acc += q[k * stride];
            s[c] = acc;
            S[c * N + e] = acc;
        }
        // losses.py:125-132, same expression order
        const float uI = s[0] / nwin, uJ = s[1] / nwin;
        const float cross = s[4] - uJ * s[0] - uI * s[1] + uI * uJ * nwin;
        const float Iv = s[2] - 2.f * uI * s[0] + uI * uI * nwin;
        const float Jv = s[3] - 2.f * uJ * s[1] + uJ * uJ * nwin;
        local += cross * cross / (Iv * Jv + 1e-8f);
    }
    const float t = block_sum_256(local, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// ---- marching form of the strided passes (windows up to 11): a thread walks a segment of one line along the axis and keeps the window's
// 2 PAD + 1 values of every channel in registers, so each input value is loaded once (the kernels above re-read it for each of the 9 taps:
// at 160^3 the D pass's working set - 9 planes x 5 channels, 4.6 MB - exceeds an XCD's L2 and the pass ran at 0.9 TB/s).  The window
// is summed in ascending position order with zeros outside the volume: bit-identical to the tap loops above.
//   MODE 0: out = box(in)                                  (nch = NCH channels)
//   MODE 1: S = box(in), partial[block] = sum of cc        (ncc_final_kernel)
//   MODE 2: gJ = k0 (box(a) + 2 J box(b) + I box(c))       (ncc_bwd_final_kernel)
// Work items = (segment, group of 256 lines), distributed over the workgroups in a strided loop.
template <int NCH, int PAD, int MODE>
__global__ __launch_bounds__(256) void box_march_kernel(const float* __restrict__ in, float* __restrict__ out, long N, long nlines, int extent,
                                                          long stride, int seglen, int nseg, float nwin, float* __restrict__ partial,
                                                          const float* __restrict__ I, const float* __restrict__ J, const float* __restrict__ gscale,
                                                          float coef) {
    constexpr int WIN = 2 * PAD + 1;
    __shared__ float sh[4];
    const long nlg = (nlines + 255) / 256;
    const long nitem = nlg * nseg;
    float local = 0.f;
    [[maybe_unused]] const float k0 = MODE == 2 ? coef * (gscale != nullptr ? gscale[0] : 1.f) : 0.f;
    for (long item = blockIdx.x; item < nitem; item += gridDim.x) {
        const long lg = item % nlg;
        const int seg = (int)(item / nlg);
        const long line = lg * 256 + threadIdx.x;
        if (line >= nlines) continue;
        const long outer = line / stride, inner = line - outer * stride;
        const long base = outer * extent * stride + inner;
        const int p0 = seg * seglen, p1 = min(extent, p0 + seglen);
        float ring[NCH][WIN];
#pragma unroll
        for (int j = 0; j < WIN - 1; ++j) {
            const int pos = p0 - PAD + j;
            const bool ok = pos >= 0 && pos < extent;
#pragma unroll
            for (int c = 0; c < NCH; ++c) ring[c][j] = ok ? in[c * N + base + pos * stride] : 0.f;
        }
        for (int t0 = p0; t0 < p1; t0 += WIN) {
#pragma unroll
            for (int u = 0; u < WIN; ++u) {
                const int p = t0 + u;
                if (p < p1) {
                    const int slot = (u + WIN - 1) % WIN;
                    const int pos = p + PAD;
                    const bool ok = pos < extent;
                    float sum[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        ring[c][slot] = ok ? in[c * N + base + pos * stride] : 0.f;
                        float acc = 0.f;
#pragma unroll
                        for (int j = 0; j < WIN; ++j) acc += ring[c][(u + j) % WIN];
                        sum[c] = acc;
                    }
                    const long e = base + p * stride;
                    if constexpr (MODE == 0) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) out[c * N + e] = sum[c];
                    } else if constexpr (MODE == 1) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) out[c * N + e] = sum[c];
                        // losses.py:125-132, same expression order
                        const float uI = sum[0] / nwin, uJ = sum[1] / nwin;
                        const float cross = sum[4] - uJ * sum[0] - uI * sum[1] + uI * uJ * nwin;
                        const float Iv = sum[2] - 2.f * uI * sum[0] + uI * uI * nwin;
                        const float Jv = sum[3] - 2.f * uJ * sum[1] + uJ * uJ * nwin;
                        local += cross * cross / (Iv * Jv + 1e-8f);
                    } else {
                        out[e] = k0 * (sum[0] + 2.f * J[e] * sum[1] + I[e] * sum[2]);
                    }
                }
            }
        }
    }
    if constexpr (MODE == 1) {
        const float t = block_sum_256(local, sh);
        if (threadIdx.x == 0) partial[blockIdx.x] = t;
    }
}

// segment length for the marching kernels: about 1024 work items where the lines allow, segments of at least 2 PAD positions
static inline void march_segments(long nlines, int extent, int pad, int& seglen, int& nseg) {
    const long nlg = (nlines + 255) / 256;
    const long want = std::max<long>(1, (1024 + nlg - 1) / nlg);
    nseg = (int)std::max<long>(1, std::min<long>(want, extent / std::max(1, 2 * pad)));
    seglen = (extent + nseg - 1) / nseg;
    nseg = (extent + seglen - 1) / seglen;
}

template <int NCH, int MODE>
static int launch_march(int pad, int grid, hipStream_t st, const float* in, float* out, long N, int extent, long stride, float nwin, float* partial,
                        const float* I, const float* J, const float* gscale, float coef) {
    const long nlines = N / extent;
    int seglen, nseg;
    march_segments(nlines, extent, pad, seglen, nseg);
#define PULPO_MARCH(P) hipLaunchKernelGGL((box_march_kernel<NCH, P, MODE>), dim3(grid), dim3(256), 0, st, in, out, N, nlines, extent, stride, seglen, nseg, nwin, partial, I, J, gscale, coef)
    switch (pad) {
        case 1: PULPO_MARCH(1); break;
        case 2: PULPO_MARCH(2); break;
        case 3: PULPO_MARCH(3); break;
        case 4: PULPO_MARCH(4); break;
        case 5: PULPO_MARCH(5); break;
        default: return -1;
    }
#undef PULPO_MARCH
    return pulpo::check_launch("ncc box march");
}

// backward stage 1: from the saved box sums form the three fields that get box-filtered again
__global__ __launch_bounds__(256) void ncc_abc_kernel(const float* __restrict__ S, float* __restrict__ A, long N, float nwin) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < N; e += (long)gridDim.x * blockDim.x) {
        const float SI = S[e], SJ = S[N + e], SII = S[2 * N + e], SJJ = S[3 * N + e], SIJ = S[4 * N + e];
        const float cross = SIJ - SI * SJ / nwin;
        const float Iv = SII - SI * SI / nwin;
        const float Jv = SJJ - SJ * SJ / nwin;
        const float Dn = Iv * Jv + 1e-8f;
        const float r = cross / Dn;                       // cross / D
        A[e] = -2.f * r * SI / nwin + 2.f * r * r * Iv * SJ / nwin;
        A[N + e] = -r * r * Iv;
        A[2 * N + e] = 2.f * r;
    }
}

// backward last (D) pass fused with the combine:  gJ = coef * (box(a) + 2 J box(b) + I box(c))
__global__ __launch_bounds__(256) void ncc_bwd_final_kernel(const float* __restrict__ in, const float* __restrict__ I, const float* __restrict__ J,
                                                              const float* __restrict__ gscale, float coef, float* __restrict__ gJ, long N, int extent,
                                                              long stride, int pad) {
    const float k0 = coef * (gscale != nullptr ? gscale[0] : 1.f);
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < N; e += (long)gridDim.x * blockDim.x) {
        const int pos = (int)((e / stride) % extent);
        const int lo = max(-pad, -pos), hi = min(pad, extent - 1 - pos);
        float s[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* q = in + c * N + e;
            float acc = 0.f;
            for (int k = lo; k <= hi; ++k) acc += q[k * stride];
            s[c] = acc;
        }
        gJ[e] = k0 * (s[0] + 2.f * J[e] * s[1] + I[e] * s[2]);
    }
}

// ------------------------------------------------------------------------------------------------ KL (diagonal Gaussians)
// mu1 / sigma1 nullable = the N(0,1) prior of PULPoPrior (src/components/pulpo.py:330-341)
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ sigma, const float* __restrict__ mu1,
                                                       const float* __restrict__ sigma1, long n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float local = 0.f;
    const float eps = 1e-10f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float s0 = sigma[e] * sigma[e];
        const float s1 = sigma1 != nullptr ? sigma1[e] * sigma1[e] : 1.f;
        const float dm = (mu1 != nullptr ? mu1[e] : 0.f) - mu[e];
        // losses.py:55-73:  (s0 + (mu1-mu0)^2)/(s1+eps) + log(s1+eps) - log(s0+eps) - 1
        local += (s0 + dm * dm) / (s1 + eps) + logf(s1 + eps) - logf(s0 + eps) - 1.f;
    }
    const float t = block_sum_256(local, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ sigma, const float* __restrict__ mu1,
                                                       const float* __restrict__ sigma1, const float* __restrict__ gscale, float coef,
                                                       float* __restrict__ gmu, float* __restrict__ gsigma, long n) {
    const float k0 = coef * (gscale != nullptr ? gscale[0] : 1.f);
    const float eps = 1e-10f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float s = sigma[e], m = mu[e];
        const float s1 = sigma1 != nullptr ? sigma1[e] * sigma1[e] : 1.f;
        const float m1 = mu1 != nullptr ? mu1[e] : 0.f;
        gmu[e] = k0 * (m - m1) / (s1 + eps);
        gsigma[e] = k0 * (s / (s1 + eps) - s / (s * s + eps));
    }
}

// ------------------------------------------------------------------------------------------------ L2 regulariser
__global__ __launch_bounds__(256) void l2reg_fwd_kernel(const float* __restrict__ df, long nplanes, int D, int H, int W, float* __restrict__ partial) {
    __shared__ float sh[4];
    float local = 0.f;
    const long V = (long)D * H * W, total = nplanes * V;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long v = e % V;
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((long)W * H));
        if (x >= 1 && y >= 1 && (z >= 1 || D == 1)) {          // D == 1: the 2-D form (losses.py:211-215), no difference along depth
            const float c = df[e];
            const float a = D == 1 ? 0.f : c - df[e - (long)H * W], b = c - df[e - W], d = c - df[e - 1];
            local += a * a + b * b + d * d;
        }
    }
    const float t = block_sum_256(local, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// the same sums for rows that are whole groups of four voxels (W % 4 == 0, 16-byte aligned): one item = four consecutive x of one row - 16-byte
// loads of the row, the row above (y - 1) and the row behind (z - 1), 32-bit index arithmetic per item instead of 64-bit divisions per voxel
// (round 5: 61 -> ~15 us at 3 x 160^3).  Which block sums what differs from the scalar kernel (fp32 rounding of the partial sums only).
__global__ __launch_bounds__(256) void l2reg_fwd_vec_kernel(const float* __restrict__ df, long nplanes, int D, int H, int W, float* __restrict__ partial) {
    __shared__ float sh[4];
    float local = 0.f;
    const int W4 = W >> 2;
    const long items = nplanes * D * H * W4;
    for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const int x4 = (int)(it % W4);
        const long row = it / W4;                         // (plane, z, y) linearised
        const int y = (int)(row % H);
        const int z = (int)((row / H) % D);
        if (y >= 1 && (z >= 1 || D == 1)) {
            const long e = row * W + 4 * x4;
            const float4 c = *reinterpret_cast<const float4*>(df + e);
            const float4 u = *reinterpret_cast<const float4*>(df + e - W);
            float4 bk = c;                                // (D == 1: no depth term)
            if (D != 1) bk = *reinterpret_cast<const float4*>(df + e - (long)H * W);
            const float cc[4] = {c.x, c.y, c.z, c.w}, uu[4] = {u.x, u.y, u.z, u.w}, bb[4] = {bk.x, bk.y, bk.z, bk.w};
            float left = x4 > 0 ? df[e - 1] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (4 * x4 + k >= 1) {
                    const float a = cc[k] - bb[k], b = cc[k] - uu[k], d = cc[k] - left;
                    local += a * a + b * b + d * d;
                }
                left = cc[k];
            }
        }
    }
    const float t = block_sum_256(local, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void l2reg_bwd_vec_kernel(const float* __restrict__ df, const float* __restrict__ gscale, float coef,
                                                              float* __restrict__ gdf, long nplanes, int D, int H, int W) {
    const float k0 = 2.f * coef * (gscale != nullptr ? gscale[0] : 1.f);
    const int W4 = W >> 2;
    const long items = nplanes * D * H * W4;
    const long sz = (long)H * W;
    const bool flat = D == 1;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const int x4 = (int)(it % W4);
        const long row = it / W4;
        const int y = (int)(row % H);
        const int z = (int)((row / H) % D);
        const long e = row * W + 4 * x4;
        const float4 c = *reinterpret_cast<const float4*>(df + e);
        const float4 zm = (z >= 1) ? *reinterpret_cast<const float4*>(df + e - sz) : zero;
        const float4 zp = (z + 1 < D) ? *reinterpret_cast<const float4*>(df + e + sz) : zero;
        const float4 ym = (y >= 1) ? *reinterpret_cast<const float4*>(df + e - W) : zero;
        const float4 yp = (y + 1 < H) ? *reinterpret_cast<const float4*>(df + e + W) : zero;
        const float xm = x4 > 0 ? df[e - 1] : 0.f, xp = x4 + 1 < W4 ? df[e + 4] : 0.f;
        const float cc[4] = {c.x, c.y, c.z, c.w}, a0[4] = {zm.x, zm.y, zm.z, zm.w}, a1[4] = {zp.x, zp.y, zp.z, zp.w};
        const float b0[4] = {ym.x, ym.y, ym.z, ym.w}, b1[4] = {yp.x, yp.y, yp.z, yp.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = 4 * x4 + k;
            const float cv = cc[k];
            const float lft = k > 0 ? cc[k - 1] : xm, rgt = k < 3 ? cc[k + 1] : xp;
            float g = 0.f;                                  // the scalar kernel's four terms, in its order
            if (x >= 1 && y >= 1 && (z >= 1 || flat)) g += (flat ? 0.f : cv - a0[k]) + (cv - b0[k]) + (cv - lft);
            if (z + 1 < D && y >= 1 && x >= 1) g -= a1[k] - cv;
            if (y + 1 < H && (z >= 1 || flat) && x >= 1) g -= b1[k] - cv;
            if (x + 1 < W && (z >= 1 || flat) && y >= 1) g -= rgt - cv;
            o[k] = k0 * g;
        }
        *reinterpret_cast<float4*>(gdf + e) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

__global__ __launch_bounds__(256) void l2reg_bwd_kernel(const float* __restrict__ df, const float* __restrict__ gscale, float coef,
                                                          float* __restrict__ gdf, long nplanes, int D, int H, int W) {
    const float k0 = 2.f * coef * (gscale != nullptr ? gscale[0] : 1.f);
    const long V = (long)D * H * W, total = nplanes * V;
    const long sz = (long)H * W, sy = W;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long v = e % V;
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / sz);
        const float c = df[e];
        float g = 0.f;
        const bool flat = D == 1;                                                                           // 2-D form: no depth term
        if (x >= 1 && y >= 1 && (z >= 1 || flat)) g += (flat ? 0.f : c - df[e - sz]) + (c - df[e - sy]) + (c - df[e - 1]);   // as the centre voxel
        if (z + 1 < D && y >= 1 && x >= 1) g -= df[e + sz] - c;                                         // as the -z neighbour
        if (y + 1 < H && (z >= 1 || flat) && x >= 1) g -= df[e + sy] - c;
        if (x + 1 < W && (z >= 1 || flat) && y >= 1) g -= df[e + 1] - c;
        gdf[e] = k0 * g;
    }
}

}  // namespace

PULPO_API int pulpo_loss_blocks(int64_t n) { return eblocks(n, 1024); }

// win^ndims of losses.py:124: a depth-1 volume is the reference's 2-D case (conv2d with a win x win window)
static inline float ncc_window_count(int win, int D) { return D == 1 ? (float)(win * win) : (float)(win * win * win); }

// I = y_true, J = y_pred, planar (B,1,D,H,W).  S: 5*N floats (saved for backward), T: 10*N floats scratch (N = B*D*H*W).
// partial: pulpo_loss_blocks(N) floats.  loss = -gamma/B * sum(partial)  (finish with pulpo_colsum(scale = -gamma/B)).
PULPO_API int pulpo_ncc_fwd(const float* I, const float* J, float* S, float* T, float* partial, int B, int D, int H, int W, int win, void* stream) {
    PULPO_REQUIRE(I && J && S && T && partial && B > 0 && D > 0 && H > 0 && W > 0, "ncc_fwd: bad arguments");
    PULPO_REQUIRE(win >= 1 && (win & 1) && win <= 31, "ncc_fwd: window must be odd and <= 31");
    hipStream_t st = (hipStream_t)stream;
    const long N = (long)B * D * H * W;
    const int pad = win / 2, span = 64 - 2 * pad, segs = pulpo::cdiv(W, span);
    const long nrows = (long)B * D * H;
    float* T1 = T;
    float* T2 = T + 5 * N;
    hipLaunchKernelGGL(box_x_kernel<0>, dim3(eblocks(nrows * segs * 64)), dim3(256), 0, st, I, J, T1, N, nrows, W, pad, 5, segs);
    int rc = pulpo::check_launch("ncc box_x");
    if (rc) return rc;
    const bool march = pad >= 1 && pad <= 5;            // windows 3 .. 11 (the model's: 9, 7, 5, 3 from the finest level down)
    if (march && H > 1) rc = launch_march<5, 0>(pad, eblocks(N, 1024), st, T1, T2, N, H, (long)W, 0.f, nullptr, nullptr, nullptr, nullptr, 0.f);
    else {
        hipLaunchKernelGGL(box_axis_kernel, dim3(eblocks(N)), dim3(256), 0, st, T1, T2, N, 5, H, (long)W, pad);
        rc = pulpo::check_launch("ncc box_y");
    }
    if (rc) return rc;
    if (march) return launch_march<5, 1>(pad, pulpo_loss_blocks(N), st, T2, S, N, D, (long)H * W, ncc_window_count(win, D), partial, nullptr, nullptr, nullptr, 0.f);
    hipLaunchKernelGGL(ncc_final_kernel, dim3(pulpo_loss_blocks(N)), dim3(256), 0, st, T2, S, N, D, (long)H * W, pad, ncc_window_count(win, D), partial);
    return pulpo::check_launch("ncc final");
}

// gJ = gscale[0] * (-gamma/B) * d(sum cc)/dJ.   T: 6*N floats scratch.
PULPO_API int pulpo_ncc_bwd(const float* I, const float* J, const float* S, float* T, const float* gscale, float coef, float* gJ, int B, int D, int H,
                            int W, int win, void* stream) {
    PULPO_REQUIRE(I && J && S && T && gJ && B > 0 && D > 0 && H > 0 && W > 0, "ncc_bwd: bad arguments");
    PULPO_REQUIRE(win >= 1 && (win & 1) && win <= 31, "ncc_bwd: window must be odd and <= 31");
    hipStream_t st = (hipStream_t)stream;
    const long N = (long)B * D * H * W;
    const int pad = win / 2, span = 64 - 2 * pad, segs = pulpo::cdiv(W, span);
    const long nrows = (long)B * D * H;
    float* T1 = T;
    float* T2 = T + 3 * N;
    hipLaunchKernelGGL(ncc_abc_kernel, dim3(eblocks(N)), dim3(256), 0, st, S, T1, N, ncc_window_count(win, D));
    int rc = pulpo::check_launch("ncc abc");
    if (rc) return rc;
    hipLaunchKernelGGL(box_x_kernel<1>, dim3(eblocks(nrows * segs * 64)), dim3(256), 0, st, T1, nullptr, T2, N, nrows, W, pad, 3, segs);
    rc = pulpo::check_launch("ncc bwd box_x");
    if (rc) return rc;
    const bool march = pad >= 1 && pad <= 5;
    if (march && H > 1) rc = launch_march<3, 0>(pad, eblocks(N, 1024), st, T2, T1, N, H, (long)W, 0.f, nullptr, nullptr, nullptr, nullptr, 0.f);
    else {
        hipLaunchKernelGGL(box_axis_kernel, dim3(eblocks(N)), dim3(256), 0, st, T2, T1, N, 3, H, (long)W, pad);
        rc = pulpo::check_launch("ncc bwd box_y");
    }
    if (rc) return rc;
    if (march) return launch_march<3, 2>(pad, eblocks(N, 1024), st, T1, gJ, N, D, (long)H * W, 0.f, nullptr, I, J, gscale, coef);
    hipLaunchKernelGGL(ncc_bwd_final_kernel, dim3(eblocks(N)), dim3(256), 0, st, T1, I, J, gscale, coef, gJ, N, D, (long)H * W, pad);
    return pulpo::check_launch("ncc bwd final");
}

// sum over all elements of the KL integrand; finish with pulpo_colsum(scale = 0.5 / B).  mu1 / sigma1 nullable = N(0,1).
PULPO_API int pulpo_kl_fwd(const float* mu, const float* sigma, const float* mu1, const float* sigma1, int64_t n, float* partial, void* stream) {
    PULPO_REQUIRE(mu && sigma && partial && n > 0, "kl_fwd: bad arguments");
    hipLaunchKernelGGL(kl_fwd_kernel, dim3(pulpo_loss_blocks(n)), dim3(256), 0, (hipStream_t)stream, mu, sigma, mu1, sigma1, (long)n, partial);
    return pulpo::check_launch("kl_fwd");
}

// gradients w.r.t. the first distribution only (the prior is a constant in the reference); coef = 1/B
PULPO_API int pulpo_kl_bwd(const float* mu, const float* sigma, const float* mu1, const float* sigma1, const float* gscale, float coef, float* gmu,
                           float* gsigma, int64_t n, void* stream) {
    PULPO_REQUIRE(mu && sigma && gmu && gsigma && n > 0, "kl_bwd: bad arguments");
    hipLaunchKernelGGL(kl_bwd_kernel, dim3(eblocks(n)), dim3(256), 0, (hipStream_t)stream, mu, sigma, mu1, sigma1, gscale, coef, gmu, gsigma, (long)n);
    return pulpo::check_launch("kl_bwd");
}

// sum of squared forward differences on the [1:,1:,1:] block of nplanes = B*3 volumes;
// finish with pulpo_colsum(scale = lamb*D*H*W / (nplanes*(D-1)*(H-1)*(W-1)))
PULPO_API int pulpo_l2reg_fwd(const float* df, int64_t nplanes, int D, int H, int W, float* partial, void* stream) {
    PULPO_REQUIRE(df && partial && nplanes > 0 && D >= 1 && H > 1 && W > 1, "l2reg_fwd: bad arguments");
    if (W % 4 == 0 && (((uintptr_t)df) & 15) == 0)
        hipLaunchKernelGGL(l2reg_fwd_vec_kernel, dim3(pulpo_loss_blocks(nplanes * D * H * W)), dim3(256), 0, (hipStream_t)stream, df, (long)nplanes, D, H, W, partial);
    else
        hipLaunchKernelGGL(l2reg_fwd_kernel, dim3(pulpo_loss_blocks(nplanes * D * H * W)), dim3(256), 0, (hipStream_t)stream, df, (long)nplanes, D, H, W, partial);
    return pulpo::check_launch("l2reg_fwd");
}

PULPO_API int pulpo_l2reg_bwd(const float* df, const float* gscale, float coef, float* gdf, int64_t nplanes, int D, int H, int W, void* stream) {
    PULPO_REQUIRE(df && gdf && nplanes > 0 && D >= 1 && H > 1 && W > 1, "l2reg_bwd: bad arguments");
    if (W % 4 == 0 && ((((uintptr_t)df) | ((uintptr_t)gdf)) & 15) == 0)
        hipLaunchKernelGGL(l2reg_bwd_vec_kernel, dim3(eblocks(nplanes * D * H * (W / 4))), dim3(256), 0, (hipStream_t)stream, df, gscale, coef, gdf, (long)nplanes, D, H, W);
    else
        hipLaunchKernelGGL(l2reg_bwd_kernel, dim3(eblocks(nplanes * D * H * W)), dim3(256), 0, (hipStream_t)stream, df, gscale, coef, gdf, (long)nplanes, D, H, W);
    return pulpo::check_launch("l2reg_bwd");
}

// ---- weighted sum of per-level loss terms (HierarchicalKLLoss / ...ReconstructionLoss / ...Regularization, src/losses.py:262-276, 305-325,
// 343-355: all_levels[l] = w_l * term_l, loss = 0 + all_levels[0] + all_levels[1] + ..., and models.py:161-162's kl * beta): ONE launch
// instead of a mul and an add kernel per level.  Same operation order as the reference's scalar arithmetic (fp32 products, summed in level
// order, the post-scale applied to the sum and to every level term), so the values are bit-identical to the torch expressions.
namespace {
__global__ void weighted_sum_fwd_kernel(const float* __restrict__ t, const float* __restrict__ w, int n, float scale, int scaled, float* __restrict__ levels,
                                        float* __restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // explicit round-to-nearest products and sums: the library is built with the compiler's default -ffp-contract=fast, under which
        // `s += w * t` may become fma(w, t, s) - one rounding instead of the two of the reference's `w * term` and `+=` kernels
        float s = 0.f;
        for (int i = 0; i < n; ++i) {
            const float v = __fmul_rn(w[i], t[i]);
            s = __fadd_rn(s, v);
            levels[i] = scaled ? __fmul_rn(v, scale) : v;
        }
        total[0] = scaled ? __fmul_rn(s, scale) : s;
    }
}
// g_t[i] = (g_total + g_levels[i]) * scale * w[i]   (either upstream gradient may be absent: the per-level outputs are normally only logged)
__global__ void weighted_sum_bwd_kernel(const float* __restrict__ gtotal, const float* __restrict__ glevels, const float* __restrict__ w, int n, float scale,
                                        float* __restrict__ gt) {
    const int i = threadIdx.x;
    // autograd's chain for v_i = w_i t_i, total = (sum_i v_i) * scale, level_i = v_i * scale: the two paths meet at v_i (their gradients,
    // g_total * scale and g_level_i * scale, are ADDED there) and the sum goes through the product with w_i
    if (blockIdx.x == 0 && i < n) {
        const float a = gtotal ? __fmul_rn(gtotal[0], scale) : 0.f;
        const float b = glevels ? __fmul_rn(glevels[i], scale) : 0.f;
        gt[i] = __fmul_rn(__fadd_rn(a, b), w[i]);
    }
}
}  // namespace

PULPO_API int pulpo_weighted_sum_fwd(const float* terms, const float* weights, int n, float scale, int scaled, float* levels, float* total, void* stream) {
    PULPO_REQUIRE(terms && weights && levels && total && n > 0 && n <= 64, "weighted_sum_fwd: bad arguments");
    hipLaunchKernelGGL(weighted_sum_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, terms, weights, n, scale, scaled, levels, total);
    return pulpo::check_launch("weighted_sum_fwd");
}

PULPO_API int pulpo_weighted_sum_bwd(const float* gtotal, const float* glevels, const float* weights, int n, float scale, float* gterms, void* stream) {
    PULPO_REQUIRE((gtotal || glevels) && weights && gterms && n > 0 && n <= 64, "weighted_sum_bwd: bad arguments");
    hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gtotal, glevels, weights, n, scale, gterms);
    return pulpo::check_launch("weighted_sum_bwd");
}
