// Calibration of rocprofv3's FETCH_SIZE for the access pattern of the pipelined (y, x) Winograd convolution kernel (conv3d_wino2p.hip):
// per voxel a 32-byte run (one 8-channel chunk of a 32-channel channels-last tensor, two lanes x 16 B, buffer_load_dwordx4), gathered over a
// 6 x 10 x 10 halo by a 256-thread workgroup, four x taps per staging item, four chunks per tile.  MI355X_MICROARCH.md (HBM) gives the factor
// for wide streaming reads (FETCH_SIZE = 1/2 of the bytes) and calls other widths uncalibrated: this program reads a KNOWN byte count with the
// kernel's own pattern so that scripts/pmc_traffic.py can use a measured factor.
//
// Every workgroup reads a DISJOINT 6 x 10 x 10 block (block origins 6 / 10 / 10 apart), so each voxel's 128 bytes cross the fabric once
// (the two-tap overlap of neighbouring staging items is re-read within the same workgroup, microseconds apart: L1 / L2 hits); the tensor
// (160^3 x 32 channels = 524 MB) is far larger than the 256 MiB Infinity Cache.  A plain 16 B/lane streaming read of the same buffer - the
// guide's reference case - runs beside it as a check of the method.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/build/fetch_calib scripts/probes/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -d <dir> --output-format csv -- scripts/probes/build/fetch_calib      (then scripts/probes/fetch_calib_report.py <dir>)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int S = 160, C = 32, HZ = 6, HY = 10, HXB = 4;        // halo block: 6 x 10 x (4 x-pairs -> 10 voxels)
constexpr int NBZ = S / HZ, NBY = S / HY, NBX = S / HY;          // 26 x 16 x 16 disjoint blocks

template <int VARIANT>      // 0: 512 workgroups (two per CU, the convolution's launch); 1: 32 workgroups - 2.4 MB of halo blocks in flight, inside ONE XCD's 4 MiB L2 share
__global__ __launch_bounds__(256) void halo_gather_kernel(const float* __restrict__ in, float* __restrict__ sink, int nblk) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)((long)S * S * S * C * 4), 0x00020000);
    const int tid = threadIdx.x;
    const unsigned ps = C * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int bx = blk % NBX, by = (blk / NBX) % NBY, bz = blk / (NBX * NBY);
        const unsigned origin = (unsigned)(((bz * HZ) * S + by * HY) * S + bx * HY) * ps;
        for (int chunk = 0; chunk < C / 8; ++chunk) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = tid + u * 256;
                const int q = j & 1, rb = j >> 1, xb = rb & 3, hrow = rb >> 2, hz = hrow / HY, hy = hrow - hz * HY;
                if (j < HZ * HY * HXB * 2) {
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt) {
                        const unsigned off = origin + (unsigned)((hz * S + hy) * S + 2 * xb + tt) * ps + 16u * q;
                        const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, chunk * 32, 0));
                        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                    }
                }
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[tid] = 1.f;
}

__global__ __launch_bounds__(256) void stream_read_kernel(const float4* __restrict__ in, float* __restrict__ sink, long n4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = in[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[threadIdx.x] = 1.f;
}

int main() {
    const size_t bytes = (size_t)S * S * S * C * 4;
    float *buf, *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(buf, 0, bytes));
    const int nblk = NBZ * NBY * NBX;
    const double halo_bytes = (double)nblk * HZ * HY * HY * C * 4;      // every voxel of every block, all 32 channels, once
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(halo_gather_kernel<0>, dim3(512), dim3(256), 0, 0, buf, sink, nblk);
        hipLaunchKernelGGL(halo_gather_kernel<1>, dim3(32), dim3(256), 0, 0, buf, sink, nblk);
        hipLaunchKernelGGL(stream_read_kernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const float4*>(buf), sink, (long)(bytes / 16));
    }
    CHECK(hipDeviceSynchronize());
    printf("halo_gather_kernel expected_bytes %.0f\nstream_read_kernel expected_bytes %.0f\n", halo_bytes, (double)bytes);
    return 0;
}
