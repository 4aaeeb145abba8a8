// Stand-alone probe (not part of the product): what the matrix pipe of an MI355X sustains for the instruction mixes the convolution kernels
// use or might use.  hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
//   mix 0: fp32 MFMAs only (v_mfma_f32_32x32x2_f32, four independent accumulators per wave)
//   mix 1: fp32 MFMA + one ds_read_b128 per 2 MFMAs + one v_fma per MFMA            (the (y, x) Winograd kernel's inner loop, roughly)
//   mix 2: bf16 MFMAs only (v_mfma_f32_32x32x16_bf16)
//   mix 3: six bf16 MFMAs per six ds_read_b128                                      (fp32 products from three-term bf16 splits)
//   mix 4: eight fp32 MFMAs (eight accumulators) per six ds_read_b128 and twenty v_fma  (Winograd F(2,3) along z as well: operands = y and z
//          combinations of six rows formed at read time, two z points x four k-steps per point step)
// For every mix and 1 / 2 / 4 waves per SIMD on all CUs: MFMA instructions per second, the implied TFLOP/s, and - from s_memtime around the
// loop - shader clocks per MFMA and wave, i.e. pipe clocks per MFMA = that / (waves per SIMD), and the clock the chip held.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

template <int MIX>
__global__ __launch_bounds__(256, 2) void probe(float* out, unsigned long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int j = tid; j < 8192; j += blockDim.x) lds[j] = (float)(j & 15) * 0.001f;
    __syncthreads();
    f32x16 acc[4];
    for (int p = 0; p < 4; ++p)
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    [[maybe_unused]] f32x16 acc8[8];
    if constexpr (MIX == 4)
        for (int p = 0; p < 8; ++p)
            for (int r = 0; r < 16; ++r) acc8[p][r] = 0.f;
    float a = 1.0f + lane * 1e-3f, b = 0.5f;
    const float* rp = lds + lane * 4;
    float4 v = *reinterpret_cast<const float4*>(rp);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MIX == 0) {
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p], 0, 0, 0);
        } else if constexpr (MIX == 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if ((p & 1) == 0) v = *reinterpret_cast<const float4*>(rp + ((it * 4 + p) & 255) * 16);
                const float x = __builtin_fmaf(-1.f, v.x, v.y);
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b, acc[p], 0, 0, 0);
            }
        } else if constexpr (MIX == 4) {
            float4 r6[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) r6[k] = *reinterpret_cast<const float4*>(rp + ((it * 6 + k) & 255) * 16);
            // y combinations of the three planes, then the two z combinations
            float pz[3][4], op[2][4];
#pragma unroll
            for (int z = 0; z < 3; ++z) {
                pz[z][0] = __builtin_fmaf(-1.f, r6[2 * z + 1].x, r6[2 * z].x); pz[z][1] = __builtin_fmaf(-1.f, r6[2 * z + 1].y, r6[2 * z].y);
                pz[z][2] = __builtin_fmaf(-1.f, r6[2 * z + 1].z, r6[2 * z].z); pz[z][3] = __builtin_fmaf(-1.f, r6[2 * z + 1].w, r6[2 * z].w);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { op[0][k] = __builtin_fmaf(-1.f, pz[2][k], pz[0][k]); op[1][k] = __builtin_fmaf(1.f, pz[2][k], pz[1][k]); }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc8[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(op[0][k], b, acc8[k], 0, 0, 0);
                acc8[4 + k] = __builtin_amdgcn_mfma_f32_32x32x2f32(op[1][k], b, acc8[4 + k], 0, 0, 0);
            }
        } else if constexpr (MIX == 2) {
            bf16x8 x, y;
            for (int k = 0; k < 8; ++k) { x[k] = (short)0x3f80; y[k] = (short)0x3f00; }
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[p], 0, 0, 0);
        } else {
            bf16x8 fr[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float4 w = *reinterpret_cast<const float4*>(rp + ((it * 6 + k) & 255) * 16);
                fr[k] = *reinterpret_cast<const bf16x8*>(&w);
            }
            // a1b1, a1b2, a2b1, a2b2, a1b3, a3b1 on two accumulators (two output tiles share nothing here: worst case for LDS)
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0], fr[3], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0], fr[4], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[1], fr[3], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[1], fr[4], acc[3], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[0], fr[5], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[2], fr[3], acc[1], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if constexpr (MIX != 4)
        for (int p = 0; p < 4; ++p)
            for (int r = 0; r < 16; ++r) s += acc[p][r];
    if constexpr (MIX == 4)
        for (int p = 0; p < 8; ++p)
            for (int r = 0; r < 16; ++r) s += acc8[p][r];
    out[blockIdx.x * blockDim.x + tid] = s + v.x;
    if (lane == 0) clk[blockIdx.x * (blockDim.x / 64) + tid / 64] = t1 - t0;
}

template <int MIX>
void run(const char* name, int mfma_per_iter, double flop_per_mfma) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    for (int wps : {1, 2, 4}) {
        const int nblk = ncu * wps;                 // blocks of 4 waves: wps blocks per CU = wps waves per SIMD
        const int iters = 20000;
        float* out; unsigned long long* clk;
        hipMalloc(&out, sizeof(float) * nblk * 256);
        hipMalloc(&clk, sizeof(unsigned long long) * nblk * 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(probe<MIX>, dim3(nblk), dim3(256), 32768, 0, out, clk, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MIX>, dim3(nblk), dim3(256), 32768, 0, out, clk, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nblk * 4);
        hipMemcpy(h.data(), clk, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
        double c = 0;
        for (auto x : h) c += (double)x;
        c /= h.size();
        const double n_mfma = (double)nblk * 4 * iters * mfma_per_iter;
        const double clk_per_mfma_wave = c / ((double)iters * mfma_per_iter);
        printf("%-34s %d wave(s)/SIMD: %7.3f ms  %7.1f TFLOP/s  %6.1f clocks per MFMA and wave = %5.1f pipe clocks per MFMA;  clock held %.2f GHz\n", name, wps, ms,
               n_mfma * flop_per_mfma / (ms * 1e-3) / 1e12, clk_per_mfma_wave, clk_per_mfma_wave / wps, c / (ms * 1e-3) / 1e9);
        hipFree(out); hipFree(clk);
    }
}

int main() {
    run<0>("fp32 32x32x2, MFMA only", 4, 2.0 * 32 * 32 * 2);
    run<1>("fp32 32x32x2 + LDS read + fma", 4, 2.0 * 32 * 32 * 2);
    run<2>("bf16 32x32x16, MFMA only", 4, 2.0 * 32 * 32 * 16);
    run<3>("bf16 x6 per 6 ds_read_b128", 6, 2.0 * 32 * 32 * 16);
    run<4>("fp32 x8 per 6 ds_read_b128 + 20 fma", 8, 2.0 * 32 * 32 * 2);
    return 0;
}
