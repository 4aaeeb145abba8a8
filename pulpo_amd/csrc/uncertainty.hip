// Streaming per-voxel moments over Monte-Carlo samples of the latent hierarchy.
// evaluate.py:222-251 keeps all N sampled volumes / fields per level and calls torch.std(axis=0) then torch.mean(axis=0) over the
// channels; here each sample is folded into running (mean, M2) images as it is produced (Welford), so N never enters the
// memory footprint.  HBM-bound: one read of the sample, one read-modify-write of the two state images per update.
#include "common.h"

namespace {

inline int eblocks(long items, int cap = 8192) { return (int)std::max<long>(1, std::min<long>((items + 255) / 256, cap)); }

// k = number of samples including this one (k >= 1)
__global__ __launch_bounds__(256) void mc_update_kernel(const float* __restrict__ sample, float* __restrict__ mean, float* __restrict__ m2, long n,
                                                          int k) {
    const float inv = 1.f / (float)k;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float x = sample[e];
        if (k == 1) {
            mean[e] = x;
            m2[e] = 0.f;
        } else {
            const float mu = mean[e];
            const float d = x - mu;
            const float mu2 = mu + d * inv;
            mean[e] = mu2;
            m2[e] += d * (x - mu2);
        }
    }
}

// out[b][v] = mean over channels of sqrt(M2[b][c][v] / (k - 1))  (torch.std is unbiased: k = 1 gives NaN, as in the reference),
// optionally times |scale[b][v]| (a per-voxel mask applied to every sample: std(m * x) = |m| std(x), evaluate.py:249)
__global__ __launch_bounds__(256) void mc_std_kernel(const float* __restrict__ m2, const float* __restrict__ scale, float* __restrict__ out, int C,
                                                       long V, long total, int k) {
    const float inv = 1.f / (float)(k - 1);
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long b = e / V, v = e - b * V;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += sqrtf(m2[(b * C + c) * V + v] * inv);
        s /= (float)C;
        if (scale != nullptr) s *= fabsf(scale[e]);
        out[e] = s;
    }
}

}  // namespace

PULPO_API int pulpo_mc_moments_update(const float* sample, float* mean, float* m2, int64_t n, int k, void* stream) {
    PULPO_REQUIRE(sample && mean && m2 && n > 0 && k >= 1, "mc_moments_update: bad arguments");
    hipLaunchKernelGGL(mc_update_kernel, dim3(eblocks(n)), dim3(256), 0, (hipStream_t)stream, sample, mean, m2, (long)n, k);
    return pulpo::check_launch("mc_moments_update");
}

PULPO_API int pulpo_mc_moments_std(const float* m2, const float* scale, float* out, int B, int C, int64_t V, int k, void* stream) {
    PULPO_REQUIRE(m2 && out && B > 0 && C > 0 && V > 0 && k >= 1, "mc_moments_std: bad arguments");
    const long total = (long)B * V;
    hipLaunchKernelGGL(mc_std_kernel, dim3(eblocks(total)), dim3(256), 0, (hipStream_t)stream, m2, scale, out, C, (long)V, total, k);
    return pulpo::check_launch("mc_moments_std");
}
