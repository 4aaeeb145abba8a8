// Pyramid resampling operators (all HBM-bound streaming kernels):
//   avg_pool3d(k2, s2, ceil_mode)                  reference: src/components/pulpo.py:33,59,174-177
//   F.interpolate(trilinear, align_corners=False)  reference: pulpo.py:202 (feedback), network_blocks.py:141-147
//                                                  (ResizeTransform), losses.py:313 (y_target)
//   feedback gather: x2 trilinear up-sampling of the six planar level-(l+1) tensors fused with their channel
//   concatenation into one channels-last 16-channel tensor (pulpo.py:195-206)
#include "act_io.h"

namespace {

// ------------------------------------------------------------------------------------------------ avg pool (channels-last)
template <int VEC, typename T = float>
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const T* __restrict__ in, long ips, T* __restrict__ out, long ops, int B,
                                                             int D, int H, int W, int Do, int Ho, int Wo, int C) {
    const int CV = C / VEC;
    const long total = (long)B * Do * Ho * Wo * CV;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % CV) * VEC;
        long p = e / CV;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho); p /= Ho;
        const int oz = (int)(p % Do);
        const int b = (int)(p / Do);
        const int z1 = min(2 * oz + 2, D), y1 = min(2 * oy + 2, H), x1 = min(2 * ox + 2, W);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        for (int z = 2 * oz; z < z1; ++z)
            for (int y = 2 * oy; y < y1; ++y)
                for (int x = 2 * ox; x < x1; ++x) {
                    float t[VEC];
                    pulpo::ldv<VEC>(in + ((((long)b * D + z) * H + y) * W + x) * ips + c, t);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += t[k];
                }
        const float inv = 1.f / (float)((z1 - 2 * oz) * (y1 - 2 * oy) * (x1 - 2 * ox));   // ceil_mode: divisor = in-bounds taps
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] *= inv;
        pulpo::stv<VEC>(out + ((((long)b * Do + oz) * Ho + oy) * Wo + ox) * ops + c, acc);
    }
}

// (add, nullable: a second gradient of the same tensor - the pooled activation is also a skip connection - summed in the same pass:
//  gin = add + up(gout) / count, the operand order of autograd's own accumulation)
template <int VEC, typename T = float>
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const T* __restrict__ gout, long gops, T* __restrict__ gin, long gips, int B,
                                                             int D, int H, int W, int Do, int Ho, int Wo, int C, const T* __restrict__ add, long aps) {
    const int CV = C / VEC;
    const long total = (long)B * D * H * W * CV;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % CV) * VEC;
        long p = e / CV;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H); p /= H;
        const int z = (int)(p % D);
        const int b = (int)(p / D);
        const int oz = z >> 1, oy = y >> 1, ox = x >> 1;
        const int cnt = (min(2 * oz + 2, D) - 2 * oz) * (min(2 * oy + 2, H) - 2 * oy) * (min(2 * ox + 2, W) - 2 * ox);
        const float inv = 1.f / (float)cnt;
        float r[VEC];
        pulpo::ldv<VEC>(gout + ((((long)b * Do + oz) * Ho + oy) * Wo + ox) * gops + c, r);
#pragma unroll
        for (int k = 0; k < VEC; ++k) r[k] *= inv;
        if (add != nullptr) {
            float u[VEC];
            pulpo::ldv<VEC>(add + ((((long)b * D + z) * H + y) * W + x) * aps + c, u);
#pragma unroll
            for (int k = 0; k < VEC; ++k) r[k] = u[k] + r[k];
        }
        pulpo::stv<VEC>(gin + ((((long)b * D + z) * H + y) * W + x) * gips + c, r);
    }
}

// ------------------------------------------------------------------------------------------------ trilinear (planar)
// PyTorch's area_pixel_compute_source_index for align_corners=False: src = scale*(dst+0.5)-0.5, clamped at 0
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& lam) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    lam = s - (float)i0;
}

__global__ __launch_bounds__(256) void resize_fwd_kernel(const float* __restrict__ in, const float* __restrict__ add, float* __restrict__ out,
                                                           long nplanes, int Di, int Hi, int Wi, int Do, int Ho, int Wo, float sd, float sh, float sw,
                                                           float mult) {
    const long total = nplanes * Do * Ho * Wo;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long p = e;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho); p /= Ho;
        const int oz = (int)(p % Do);
        const long pl = p / Do;
        int z0, z1, y0, y1, x0, x1;
        float lz, ly, lx;
        src_index(oz, sd, Di, z0, z1, lz);
        src_index(oy, sh, Hi, y0, y1, ly);
        src_index(ox, sw, Wi, x0, x1, lx);
        const float* s = in + pl * (long)Di * Hi * Wi;
        auto at = [&](int z, int y, int x) { return s[((long)z * Hi + y) * Wi + x]; };
        // same association as ATen's upsample_trilinear3d: w0 = 1 - lambda, sum of 8 weighted taps
        const float wz0 = 1.f - lz, wy0 = 1.f - ly, wx0 = 1.f - lx;
        const float v = wz0 * (wy0 * (wx0 * at(z0, y0, x0) + lx * at(z0, y0, x1)) + ly * (wx0 * at(z0, y1, x0) + lx * at(z0, y1, x1))) +
                        lz * (wy0 * (wx0 * at(z1, y0, x0) + lx * at(z1, y0, x1)) + ly * (wx0 * at(z1, y1, x0) + lx * at(z1, y1, x1)));
        out[e] = add != nullptr ? v * mult + add[e] : v * mult;
    }
}

// exact x2 up-sampling (the pyramid's ResizeTransform, network_blocks.py:146-150, and the level-0 output resize): one thread = TWO outputs of an
// output row (ox = 2 m, 2 m + 1), the z and y taps and weights once for both, 32-bit index arithmetic, one 8-byte store.  Each output is the
// expression of resize_fwd_kernel on the same (i0, i1, lambda) - bit-identical results; 55 -> ~25 us at 3 x 80^3 -> 160^3 (round 5).
__global__ __launch_bounds__(256) void resize_up2_fwd_kernel(const float* __restrict__ in, const float* __restrict__ add, float* __restrict__ out,
                                                               int nplanes, int Di, int Hi, int Wi, float mult) {
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const long total = (long)nplanes * Do * Ho * Wi;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int m = (int)(e % Wi);
        int p = (int)(e / Wi);
        const int oy = p % Ho; p /= Ho;
        const int oz = p % Do;
        const int pl = p / Do;
        int z0, z1, y0, y1, xa0, xa1, xb0, xb1;
        float lz, ly, lxa, lxb;
        src_index(oz, 0.5f, Di, z0, z1, lz);
        src_index(oy, 0.5f, Hi, y0, y1, ly);
        src_index(2 * m, 0.5f, Wi, xa0, xa1, lxa);
        src_index(2 * m + 1, 0.5f, Wi, xb0, xb1, lxb);
        const float* s = in + (long)pl * Di * Hi * Wi;
        const float* r00 = s + ((long)z0 * Hi + y0) * Wi;
        const float* r01 = s + ((long)z0 * Hi + y1) * Wi;
        const float* r10 = s + ((long)z1 * Hi + y0) * Wi;
        const float* r11 = s + ((long)z1 * Hi + y1) * Wi;
        const float wz0 = 1.f - lz, wy0 = 1.f - ly;
        float v[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int x0 = k ? xb0 : xa0, x1 = k ? xb1 : xa1;
            const float lx = k ? lxb : lxa, wx0 = 1.f - lx;
            v[k] = wz0 * (wy0 * (wx0 * r00[x0] + lx * r00[x1]) + ly * (wx0 * r01[x0] + lx * r01[x1])) +
                   lz * (wy0 * (wx0 * r10[x0] + lx * r10[x1]) + ly * (wx0 * r11[x0] + lx * r11[x1]));
        }
        const long o = (((long)pl * Do + oz) * Ho + oy) * Wo + 2 * m;
        float2 r;
        if (add != nullptr) {
            const float2 a2 = *reinterpret_cast<const float2*>(add + o);
            r = make_float2(v[0] * mult + a2.x, v[1] * mult + a2.y);
        } else {
            r = make_float2(v[0] * mult, v[1] * mult);
        }
        *reinterpret_cast<float2*>(out + o) = r;
    }
}

// generic transpose: scatter with float atomics into a zeroed gin
__global__ __launch_bounds__(256) void resize_bwd_atomic_kernel(const float* __restrict__ gout, float* __restrict__ gin, long nplanes, int Di, int Hi,
                                                                  int Wi, int Do, int Ho, int Wo, float sd, float sh, float sw, float mult) {
    const long total = nplanes * Do * Ho * Wo;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long p = e;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho); p /= Ho;
        const int oz = (int)(p % Do);
        const long pl = p / Do;
        int z0, z1, y0, y1, x0, x1;
        float lz, ly, lx;
        src_index(oz, sd, Di, z0, z1, lz);
        src_index(oy, sh, Hi, y0, y1, ly);
        src_index(ox, sw, Wi, x0, x1, lx);
        const float g = gout[e] * mult;
        float* d = gin + pl * (long)Di * Hi * Wi;
        const float wz[2] = {1.f - lz, lz}, wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
        const int zi[2] = {z0, z1}, yi[2] = {y0, y1}, xi[2] = {x0, x1};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int c = 0; c < 2; ++c) atomicAdd(d + ((long)zi[a] * Hi + yi[b2]) * Wi + xi[c], g * wz[a] * wy[b2] * wx[c]);
    }
}

// generic transpose as a GATHER (deterministic form, PULPO_DETERMINISTIC): input voxel m collects, in ascending output order, every output voxel
// whose two taps along each axis include it - the same products as resize_bwd_atomic_kernel, summed in a fixed order instead of arrival order.
// Candidates along an axis: the source coordinate is monotone in the output index, so the outputs with i0 in {m - 1, m} lie in a window of
// about 2 / scale indices around (m + 0.5) / scale.
__device__ __forceinline__ void gather_window(int m, float scale, int out_size, int& lo, int& hi) {
    const float inv = 1.f / scale;
    lo = (int)floorf(((float)m - 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)m + 1.5f) * inv - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > out_size - 1 ? out_size - 1 : hi;
}
__device__ __forceinline__ float gather_weight(int o, int m, float scale, int in_size) {
    int i0, i1;
    float lam;
    src_index(o, scale, in_size, i0, i1, lam);
    return (i0 == m ? 1.f - lam : 0.f) + (i1 == m ? lam : 0.f);
}
__global__ __launch_bounds__(256) void resize_bwd_gather_kernel(const float* __restrict__ gout, float* __restrict__ gin, long nplanes, int Di, int Hi,
                                                                  int Wi, int Do, int Ho, int Wo, float sd, float sh, float sw, float mult) {
    const long total = nplanes * Di * Hi * Wi;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long p = e;
        const int x = (int)(p % Wi); p /= Wi;
        const int y = (int)(p % Hi); p /= Hi;
        const int z = (int)(p % Di);
        const long pl = p / Di;
        int zl, zh, yl, yh, xl, xh;
        gather_window(z, sd, Do, zl, zh);
        gather_window(y, sh, Ho, yl, yh);
        gather_window(x, sw, Wo, xl, xh);
        const float* g = gout + pl * (long)Do * Ho * Wo;
        float acc = 0.f;
        for (int oz = zl; oz <= zh; ++oz) {
            const float wz = gather_weight(oz, z, sd, Di);
            if (wz == 0.f) continue;
            for (int oy = yl; oy <= yh; ++oy) {
                const float wy = gather_weight(oy, y, sh, Hi);
                if (wy == 0.f) continue;
                float row = 0.f;
                for (int ox = xl; ox <= xh; ++ox) {
                    const float wx = gather_weight(ox, x, sw, Wi);
                    if (wx != 0.f) row += g[((long)oz * Ho + oy) * Wo + ox] * mult * wx;
                }
                acc += row * (wz * wy);
            }
        }
        gin[e] = acc;
    }
}

// weight with which coarse index m contributes to fine index o (exact x2 up-sampling), 0 if none
__device__ __forceinline__ float up2_weight(int o, int m, int in_size) {
    int i0, i1;
    float lam;
    src_index(o, 0.5f, in_size, i0, i1, lam);
    return (i0 == m ? 1.f - lam : 0.f) + (i1 == m ? lam : 0.f);
}

// deterministic transpose for Do == 2*Di etc.: every coarse voxel gathers its <= 4x4x4 fine contributors
__global__ __launch_bounds__(256) void resize_up2_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gin, long nplanes, int Di, int Hi,
                                                               int Wi, float mult) {
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const long total = nplanes * Di * Hi * Wi;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long p = e;
        const int x = (int)(p % Wi); p /= Wi;
        const int y = (int)(p % Hi); p /= Hi;
        const int z = (int)(p % Di);
        const long pl = p / Di;
        const float* g = gout + pl * (long)Do * Ho * Wo;
        // the four candidate taps per axis and their weights once per voxel (12 instead of 84 source-index evaluations); same products and
        // the same summation order as the plain triple loop
        float wz4[4], wy4[4], wx4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oz = 2 * z - 1 + k, oy = 2 * y - 1 + k, ox = 2 * x - 1 + k;
            wz4[k] = (oz >= 0 && oz < Do) ? up2_weight(oz, z, Di) : 0.f;
            wy4[k] = (oy >= 0 && oy < Ho) ? up2_weight(oy, y, Hi) : 0.f;
            wx4[k] = (ox >= 0 && ox < Wo) ? up2_weight(ox, x, Wi) : 0.f;
        }
        float acc = 0.f;
#pragma unroll
        for (int kz = 0; kz < 4; ++kz) {
            const int oz = 2 * z - 1 + kz;
            if (oz < 0 || oz >= Do) continue;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int oy = 2 * y - 1 + ky;
                if (oy < 0 || oy >= Ho) continue;
                const float wzy = wz4[kz] * wy4[ky];
                const float* row = g + ((long)oz * Ho + oy) * Wo;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int ox = 2 * x - 1 + kx;
                    if (ox >= 0 && ox < Wo) acc += wzy * wx4[kx] * row[ox];
                }
            }
        }
        gin[e] = acc * mult;
    }
}

// ------------------------------------------------------------------------------------------------ feedback gather
constexpr int kMaxSrc = 8;
struct FeedbackArgs {
    const float* src[kMaxSrc];    // planar (B, ch, Di, Hi, Wi)
    float* gsrc[kMaxSrc];         // backward only
    int ch[kMaxSrc];
    int nsrc, ctot;
    int B, Di, Hi, Wi;
};

// out[b][fine voxel][ctot] (pixel stride ops) = concat_s up2(src_s)
template <typename T = float>
__global__ __launch_bounds__(256) void feedback_fwd_kernel(FeedbackArgs a, T* __restrict__ out, long ops) {
    const int Do = 2 * a.Di, Ho = 2 * a.Hi, Wo = 2 * a.Wi;
    const long Vi = (long)a.Di * a.Hi * a.Wi;
    const long total = (long)a.B * Do * Ho * Wo;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long p = e;
        const int ox = (int)(p % Wo); p /= Wo;
        const int oy = (int)(p % Ho); p /= Ho;
        const int oz = (int)(p % Do);
        const int b = (int)(p / Do);
        int z0, z1, y0, y1, x0, x1;
        float lz, ly, lx;
        src_index(oz, 0.5f, a.Di, z0, z1, lz);
        src_index(oy, 0.5f, a.Hi, y0, y1, ly);
        src_index(ox, 0.5f, a.Wi, x0, x1, lx);
        const long o00 = ((long)z0 * a.Hi + y0) * a.Wi, o01 = ((long)z0 * a.Hi + y1) * a.Wi;
        const long o10 = ((long)z1 * a.Hi + y0) * a.Wi, o11 = ((long)z1 * a.Hi + y1) * a.Wi;
        const float wz0 = 1.f - lz, wy0 = 1.f - ly, wx0 = 1.f - lx;
        T* d = out + e * ops;
        int cc = 0;
        for (int s = 0; s < a.nsrc; ++s) {
            for (int c = 0; c < a.ch[s]; ++c, ++cc) {
                const float* q = a.src[s] + ((long)b * a.ch[s] + c) * Vi;
                float val[1];
                val[0] = wz0 * (wy0 * (wx0 * q[o00 + x0] + lx * q[o00 + x1]) + ly * (wx0 * q[o01 + x0] + lx * q[o01 + x1])) +
                        lz * (wy0 * (wx0 * q[o10 + x0] + lx * q[o10 + x1]) + ly * (wx0 * q[o11 + x0] + lx * q[o11 + x1]));
                pulpo::stv<1>(d + cc, val);
            }
        }
    }
}

// gsrc_s[b][c][coarse voxel] = sum over the <= 4x4x4 fine contributors of w * gout[b][fine][choff_s + c]   (deterministic gather).
// One thread per (coarse voxel, channel): 16 consecutive lanes read one 64-byte channel vector of a fine voxel.
template <typename T = float>
__global__ __launch_bounds__(256) void feedback_bwd_kernel(FeedbackArgs a, const T* __restrict__ gout, long gops) {
    const int Do = 2 * a.Di, Ho = 2 * a.Hi, Wo = 2 * a.Wi;
    const long Vi = (long)a.Di * a.Hi * a.Wi;
    const int CT = a.ctot;
    const long total = (long)a.B * Vi * CT;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(e % CT);
        long p = e / CT;
        const long v = p % Vi;
        const int x = (int)(p % a.Wi); p /= a.Wi;
        const int y = (int)(p % a.Hi); p /= a.Hi;
        const int z = (int)(p % a.Di);
        const int b = (int)(p / a.Di);
        // (tap weights once per voxel, as in resize_up2_bwd_kernel; same products, same summation order)
        float wz4[4], wy4[4], wx4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oz = 2 * z - 1 + k, oy = 2 * y - 1 + k, ox = 2 * x - 1 + k;
            wz4[k] = (oz >= 0 && oz < Do) ? up2_weight(oz, z, a.Di) : 0.f;
            wy4[k] = (oy >= 0 && oy < Ho) ? up2_weight(oy, y, a.Hi) : 0.f;
            wx4[k] = (ox >= 0 && ox < Wo) ? up2_weight(ox, x, a.Wi) : 0.f;
        }
        float acc = 0.f;
#pragma unroll
        for (int kz = 0; kz < 4; ++kz) {
            const int oz = 2 * z - 1 + kz;
            if (oz < 0 || oz >= Do) continue;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int oy = 2 * y - 1 + ky;
                if (oy < 0 || oy >= Ho) continue;
                const float wzy = wz4[kz] * wy4[ky];
                const T* row = gout + (((long)b * Do + oz) * Ho + oy) * Wo * gops + cc;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int ox = 2 * x - 1 + kx;
                    if (ox >= 0 && ox < Wo) {
                        float g1[1];
                        pulpo::ldv<1>(row + (long)ox * gops, g1);
                        acc += wzy * wx4[kx] * g1[0];
                    }
                }
            }
        }
        int s = 0, c = cc;                       // which source / channel within it
        while (c >= a.ch[s]) { c -= a.ch[s]; ++s; }
        if (a.gsrc[s] != nullptr) a.gsrc[s][((long)b * a.ch[s] + c) * Vi + v] = acc;
    }
}

// the same gather with four channels per thread (one 16-byte load per tap; ctot % 4 == 0, 16-byte aligned rows): a wave then covers 16
// coarse voxels instead of 4 and issues a quarter of the loads (40^3 x 16 channels: 104 -> ~40 us)
template <typename T = float>
__global__ __launch_bounds__(256) void feedback_bwd4_kernel(FeedbackArgs a, const T* __restrict__ gout, long gops) {
    const int Do = 2 * a.Di, Ho = 2 * a.Hi, Wo = 2 * a.Wi;
    const long Vi = (long)a.Di * a.Hi * a.Wi;
    const int CQ = a.ctot / 4;
    const long total = (long)a.B * Vi * CQ;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(e % CQ) * 4;
        long p = e / CQ;
        const long v = p % Vi;
        const int x = (int)(p % a.Wi); p /= a.Wi;
        const int y = (int)(p % a.Hi); p /= a.Hi;
        const int z = (int)(p % a.Di);
        const int b = (int)(p / a.Di);
        float wz4[4], wy4[4], wx4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oz = 2 * z - 1 + k, oy = 2 * y - 1 + k, ox = 2 * x - 1 + k;
            wz4[k] = (oz >= 0 && oz < Do) ? up2_weight(oz, z, a.Di) : 0.f;
            wy4[k] = (oy >= 0 && oy < Ho) ? up2_weight(oy, y, a.Hi) : 0.f;
            wx4[k] = (ox >= 0 && ox < Wo) ? up2_weight(ox, x, a.Wi) : 0.f;
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kz = 0; kz < 4; ++kz) {
            const int oz = 2 * z - 1 + kz;
            if (oz < 0 || oz >= Do) continue;
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int oy = 2 * y - 1 + ky;
                if (oy < 0 || oy >= Ho) continue;
                const float wzy = wz4[kz] * wy4[ky];
                const T* row = gout + (((long)b * Do + oz) * Ho + oy) * Wo * gops + cc;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int ox = 2 * x - 1 + kx;
                    if (ox >= 0 && ox < Wo) {
                        const float w = wzy * wx4[kx];
                        float g[4];
                        pulpo::ldv<4>(row + (long)ox * gops, g);
                        acc.x += w * g[0]; acc.y += w * g[1]; acc.z += w * g[2]; acc.w += w * g[3];
                    }
                }
            }
        }
        const float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int s_ = 0, c = cc + k;                  // which source / channel within it
            while (c >= a.ch[s_]) { c -= a.ch[s_]; ++s_; }
            if (a.gsrc[s_] != nullptr) a.gsrc[s_][((long)b * a.ch[s_] + c) * Vi + v] = r[k];
        }
    }
}

inline int eblocks(long items) { return (int)std::max<long>(1, std::min<long>((items + 255) / 256, 8192)); }

}  // namespace

// ---- typed forms: `dt` = dtype code of every activation operand of the call (0 fp32, 1 bf16), strides in elements
namespace {
template <typename T>
int avgpool2_fwd_t(const T* in, long ips, T* out, long ops, int B, int D, int H, int W, int C, hipStream_t st) {
    const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    pulpo::GroupProbe g(C);
    g.add(in, ips, sizeof(T)); g.add(out, ops, sizeof(T));
    if (sizeof(T) == 2 && g.ok8) {
        hipLaunchKernelGGL((avgpool2_fwd_kernel<8, T>), dim3(eblocks((long)B * Do * Ho * Wo * (C / 8))), dim3(256), 0, st, in, ips, out, ops, B, D, H, W, Do, Ho, Wo, C);
    } else if (g.ok4) {
        hipLaunchKernelGGL((avgpool2_fwd_kernel<4, T>), dim3(eblocks((long)B * Do * Ho * Wo * (C / 4))), dim3(256), 0, st, in, ips, out, ops, B, D, H, W, Do, Ho, Wo, C);
    } else {
        hipLaunchKernelGGL((avgpool2_fwd_kernel<1, T>), dim3(eblocks((long)B * Do * Ho * Wo * C)), dim3(256), 0, st, in, ips, out, ops, B, D, H, W, Do, Ho, Wo, C);
    }
    return pulpo::check_launch("avgpool2_fwd");
}

template <typename T>
int avgpool2_bwd_t(const T* gout, long gops, const T* add, long aps, T* gin, long gips, int B, int D, int H, int W, int C, hipStream_t st) {
    const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    pulpo::GroupProbe g(C);
    g.add(gout, gops, sizeof(T)); g.add(gin, gips, sizeof(T)); g.add(add, aps, sizeof(T));
    if (sizeof(T) == 2 && g.ok8) {
        hipLaunchKernelGGL((avgpool2_bwd_kernel<8, T>), dim3(eblocks((long)B * D * H * W * (C / 8))), dim3(256), 0, st, gout, gops, gin, gips, B, D, H, W, Do, Ho, Wo, C, add, aps);
    } else if (g.ok4) {
        hipLaunchKernelGGL((avgpool2_bwd_kernel<4, T>), dim3(eblocks((long)B * D * H * W * (C / 4))), dim3(256), 0, st, gout, gops, gin, gips, B, D, H, W, Do, Ho, Wo, C, add, aps);
    } else {
        hipLaunchKernelGGL((avgpool2_bwd_kernel<1, T>), dim3(eblocks((long)B * D * H * W * C)), dim3(256), 0, st, gout, gops, gin, gips, B, D, H, W, Do, Ho, Wo, C, add, aps);
    }
    return pulpo::check_launch("avgpool2_bwd");
}
}  // namespace

PULPO_API int pulpo_avgpool2_fwd_t(const void* in, int64_t ips, void* out, int64_t ops, int dt, int B, int D, int H, int W, int C, void* stream) {
    PULPO_REQUIRE(in && out && B > 0 && D > 0 && H > 0 && W > 0 && C > 0, "avgpool2_fwd: bad arguments");
    PULPO_REQUIRE_DT(dt, "avgpool2_fwd");
    PULPO_DISPATCH_DT(dt, T, return avgpool2_fwd_t((const T*)in, (long)ips, (T*)out, (long)ops, B, D, H, W, C, (hipStream_t)stream));
    return -1;
}

PULPO_API int pulpo_avgpool2_fwd(const float* in, int64_t ips, float* out, int64_t ops, int B, int D, int H, int W, int C, void* stream) {
    return pulpo_avgpool2_fwd_t(in, ips, out, ops, 0, B, D, H, W, C, stream);
}

// gin = (add +) avgpool2_bwd(gout); add (nullable) = the other gradient of a tensor that is pooled AND used as a skip connection
// (channels-last, voxel stride aps: e.g. a channel slice of a concatenation's gradient), batch stride = D*H*W*aps
PULPO_API int pulpo_avgpool2_bwd_t(const void* gout, int64_t gops, const void* add, int64_t aps, void* gin, int64_t gips, int dt, int B, int D, int H,
                                   int W, int C, void* stream) {
    PULPO_REQUIRE(gout && gin && B > 0 && D > 0 && H > 0 && W > 0 && C > 0, "avgpool2_bwd: bad arguments");
    PULPO_REQUIRE_DT(dt, "avgpool2_bwd");
    PULPO_DISPATCH_DT(dt, T, return avgpool2_bwd_t((const T*)gout, (long)gops, (const T*)add, (long)aps, (T*)gin, (long)gips, B, D, H, W, C, (hipStream_t)stream));
    return -1;
}

PULPO_API int pulpo_avgpool2_bwd(const float* gout, int64_t gops, float* gin, int64_t gips, int B, int D, int H, int W, int C, void* stream) {
    return pulpo_avgpool2_bwd_t(gout, gops, nullptr, 0, gin, gips, 0, B, D, H, W, C, stream);
}

PULPO_API int pulpo_avgpool2_bwd_add(const float* gout, int64_t gops, const float* add, int64_t aps, float* gin, int64_t gips, int B, int D, int H, int W,
                                     int C, void* stream) {
    PULPO_REQUIRE(add != nullptr, "avgpool2_bwd_add: null pointer");
    return pulpo_avgpool2_bwd_t(gout, gops, add, aps, gin, gips, 0, B, D, H, W, C, stream);
}

// planar tensors: in (nplanes, Di, Hi, Wi) -> out (nplanes, Do, Ho, Wo); out = mult * interpolate(in) (+ add, nullable:
// the DFAdder of src/network_blocks.py:152-158 fused into ResizeTransform)
// scale_*: source-coordinate step per output voxel along each axis; <= 0 selects in / out (what F.interpolate(size=...) uses).
// F.interpolate(scale_factor=f) maps with 1 / f instead, whatever floor(in * f) came out as (ResizeTransform, network_blocks.py:138-149).
PULPO_API int pulpo_resize_trilinear_scaled_fwd(const float* in, const float* add, float* out, int64_t nplanes, int Di, int Hi, int Wi, int Do, int Ho,
                                                int Wo, float scale_d, float scale_h, float scale_w, float mult, void* stream) {
    PULPO_REQUIRE(in && out && nplanes > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0, "resize_fwd: bad arguments");
    const float sd = scale_d > 0.f ? scale_d : (float)Di / (float)Do, sh = scale_h > 0.f ? scale_h : (float)Hi / (float)Ho,
                sw = scale_w > 0.f ? scale_w : (float)Wi / (float)Wo;
    if (Do == 2 * Di && Ho == 2 * Hi && Wo == 2 * Wi && sd == 0.5f && sh == 0.5f && sw == 0.5f && nplanes * (long)Do * Ho < (1L << 31) &&
        (((uintptr_t)out | (uintptr_t)add) & 7) == 0) {
        hipLaunchKernelGGL(resize_up2_fwd_kernel, dim3(eblocks(nplanes * Do * Ho * Wi)), dim3(256), 0, (hipStream_t)stream, in, add, out, (int)nplanes, Di, Hi,
                           Wi, mult);
        return pulpo::check_launch("resize_up2_fwd");
    }
    hipLaunchKernelGGL(resize_fwd_kernel, dim3(eblocks(nplanes * Do * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, in, add, out, (long)nplanes, Di,
                       Hi, Wi, Do, Ho, Wo, sd, sh, sw, mult);
    return pulpo::check_launch("resize_fwd");
}

PULPO_API int pulpo_resize_trilinear_fwd(const float* in, const float* add, float* out, int64_t nplanes, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                         float mult, void* stream) {
    return pulpo_resize_trilinear_scaled_fwd(in, add, out, nplanes, Di, Hi, Wi, Do, Ho, Wo, 0.f, 0.f, 0.f, mult, stream);
}

static int resize_bwd_impl(const float* gout, float* gin, int64_t nplanes, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                           float scale_d, float scale_h, float scale_w, float mult, bool det, void* stream) {
    PULPO_REQUIRE(gout && gin && nplanes > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0, "resize_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const float sd = scale_d > 0.f ? scale_d : (float)Di / (float)Do, sh = scale_h > 0.f ? scale_h : (float)Hi / (float)Ho,
                sw = scale_w > 0.f ? scale_w : (float)Wi / (float)Wo;
    if (Do == 2 * Di && Ho == 2 * Hi && Wo == 2 * Wi && sd == 0.5f && sh == 0.5f && sw == 0.5f) {
        hipLaunchKernelGGL(resize_up2_bwd_kernel, dim3(eblocks(nplanes * Di * Hi * Wi)), dim3(256), 0, st, gout, gin, (long)nplanes, Di, Hi, Wi, mult);
        return pulpo::check_launch("resize_up2_bwd");
    }
    if (det) {
        hipLaunchKernelGGL(resize_bwd_gather_kernel, dim3(eblocks(nplanes * Di * Hi * Wi)), dim3(256), 0, st, gout, gin, (long)nplanes, Di, Hi, Wi, Do,
                           Ho, Wo, sd, sh, sw, mult);
        return pulpo::check_launch("resize_bwd_gather");
    }
    hipError_t e = hipMemsetAsync(gin, 0, sizeof(float) * nplanes * Di * Hi * Wi, st);
    if (e != hipSuccess) return pulpo::fail((int)e, "resize_bwd memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(resize_bwd_atomic_kernel, dim3(eblocks(nplanes * Do * Ho * Wo)), dim3(256), 0, st, gout, gin, (long)nplanes, Di, Hi, Wi, Do,
                       Ho, Wo, sd, sh, sw, mult);
    return pulpo::check_launch("resize_bwd_atomic");
}

PULPO_API int pulpo_resize_trilinear_scaled_bwd(const float* gout, float* gin, int64_t nplanes, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                                float scale_d, float scale_h, float scale_w, float mult, void* stream) {
    return resize_bwd_impl(gout, gin, nplanes, Di, Hi, Wi, Do, Ho, Wo, scale_d, scale_h, scale_w, mult, false, stream);
}

// deterministic form (since ABI 4): ratios other than the exact x2 are transposed by a gather in fixed order instead of float atomics
PULPO_API int pulpo_resize_trilinear_scaled_bwd_det(const float* gout, float* gin, int64_t nplanes, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                                    float scale_d, float scale_h, float scale_w, float mult, void* stream) {
    return resize_bwd_impl(gout, gin, nplanes, Di, Hi, Wi, Do, Ho, Wo, scale_d, scale_h, scale_w, mult, true, stream);
}

PULPO_API int pulpo_resize_trilinear_bwd(const float* gout, float* gin, int64_t nplanes, int Di, int Hi, int Wi, int Do, int Ho, int Wo, float mult,
                                         void* stream) {
    return pulpo_resize_trilinear_scaled_bwd(gout, gin, nplanes, Di, Hi, Wi, Do, Ho, Wo, 0.f, 0.f, 0.f, mult, stream);
}

// srcs[i]: planar fp32 (B, chans[i], Di, Hi, Wi); out: channels-last (B, 2Di, 2Hi, 2Wi, sum chans) with pixel stride ops, dtype code dt
PULPO_API int pulpo_feedback_up2_fwd_t(const float* const* srcs, const int* chans, int nsrc, void* out, int dt, int64_t ops, int B, int Di, int Hi,
                                       int Wi, void* stream) {
    PULPO_REQUIRE(srcs && chans && out && nsrc > 0 && nsrc <= kMaxSrc && B > 0, "feedback_up2_fwd: bad arguments");
    PULPO_REQUIRE_DT(dt, "feedback_up2_fwd");
    FeedbackArgs a;
    memset(&a, 0, sizeof(a));
    a.nsrc = nsrc; a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi;
    for (int i = 0; i < nsrc; ++i) { a.src[i] = srcs[i]; a.ch[i] = chans[i]; a.ctot += chans[i]; }
    PULPO_REQUIRE(a.ctot <= 16, "feedback_up2_fwd: more than 16 feedback channels");
    const int nb = eblocks((long)B * 8 * Di * Hi * Wi);
    PULPO_DISPATCH_DT(dt, T, hipLaunchKernelGGL((feedback_fwd_kernel<T>), dim3(nb), dim3(256), 0, (hipStream_t)stream, a, (T*)out, (long)ops));
    return pulpo::check_launch("feedback_up2_fwd");
}

PULPO_API int pulpo_feedback_up2_fwd(const float* const* srcs, const int* chans, int nsrc, float* out, int64_t ops, int B, int Di, int Hi, int Wi,
                                     void* stream) {
    return pulpo_feedback_up2_fwd_t(srcs, chans, nsrc, out, 0, ops, B, Di, Hi, Wi, stream);
}

PULPO_API int pulpo_feedback_up2_bwd_t(const void* gout, int dt, int64_t gops, float* const* gsrcs, const int* chans, int nsrc, int B, int Di, int Hi,
                                       int Wi, void* stream) {
    PULPO_REQUIRE(gout && gsrcs && chans && nsrc > 0 && nsrc <= kMaxSrc && B > 0, "feedback_up2_bwd: bad arguments");
    PULPO_REQUIRE_DT(dt, "feedback_up2_bwd");
    FeedbackArgs a;
    memset(&a, 0, sizeof(a));
    a.nsrc = nsrc; a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi;
    for (int i = 0; i < nsrc; ++i) { a.gsrc[i] = gsrcs[i]; a.ch[i] = chans[i]; a.ctot += chans[i]; }
    PULPO_REQUIRE(a.ctot <= 16, "feedback_up2_bwd: more than 16 feedback channels");
    const bool four = a.ctot % 4 == 0 && gops % 4 == 0 && (((uintptr_t)gout) % (dt ? 8 : 16)) == 0;
    PULPO_DISPATCH_DT(dt, T, {
        if (four) hipLaunchKernelGGL((feedback_bwd4_kernel<T>), dim3(eblocks((long)B * Di * Hi * Wi * (a.ctot / 4))), dim3(256), 0, (hipStream_t)stream, a, (const T*)gout, (long)gops);
        else hipLaunchKernelGGL((feedback_bwd_kernel<T>), dim3(eblocks((long)B * Di * Hi * Wi * a.ctot)), dim3(256), 0, (hipStream_t)stream, a, (const T*)gout, (long)gops);
    });
    return pulpo::check_launch("feedback_up2_bwd");
}

PULPO_API int pulpo_feedback_up2_bwd(const float* gout, int64_t gops, float* const* gsrcs, const int* chans, int nsrc, int B, int Di, int Hi, int Wi,
                                     void* stream) {
    return pulpo_feedback_up2_bwd_t(gout, 0, gops, gsrcs, chans, nsrc, B, Di, Hi, Wi, stream);
}
