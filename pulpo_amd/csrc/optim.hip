// Fused Adam over the flat parameter / gradient arena (reference: src/models.py:398-400, torch.optim.Adam defaults:
// betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad).  One streaming pass: 4 reads + 3 writes per element.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                     long n, float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt, float gscale) {
    const long n4 = n >> 2;
    const float step = lr / bc1;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[e], gg = reinterpret_cast<const float4*>(g)[e];
        float4 mm = reinterpret_cast<float4*>(m)[e], vv = reinterpret_cast<float4*>(v)[e];
        float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gr = ga[k] * gscale;
            ma[k] = beta1 * ma[k] + (1.f - beta1) * gr;
            va[k] = beta2 * va[k] + (1.f - beta2) * gr * gr;
            pa[k] -= step * ma[k] / (sqrtf(va[k]) / bc2_sqrt + eps);
        }
        reinterpret_cast<float4*>(p)[e] = pp;
        reinterpret_cast<float4*>(m)[e] = mm;
        reinterpret_cast<float4*>(v)[e] = vv;
    }
    for (long e = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float gr = g[e] * gscale;
        const float mn = beta1 * m[e] + (1.f - beta1) * gr;
        const float vn = beta2 * v[e] + (1.f - beta2) * gr * gr;
        m[e] = mn; v[e] = vn;
        p[e] -= step * mn / (sqrtf(vn) / bc2_sqrt + eps);
    }
}

}  // namespace

// step >= 1.  gscale multiplies the gradient first (1/world_size after a sum all-reduce).
PULPO_API int pulpo_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step,
                              float gscale, void* stream) {
    PULPO_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
    PULPO_REQUIRE(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0, "adam_step: arenas must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const int nblk = (int)std::max<long>(1, std::min<long>(((n >> 2) + 255) / 256, 4096));
    hipLaunchKernelGGL(adam_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), gscale);
    return pulpo::check_launch("adam_step");
}
