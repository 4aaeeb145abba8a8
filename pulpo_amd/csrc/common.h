// Shared helpers for the PULPo HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>

#define PULPO_API extern "C" __attribute__((visibility("default")))

namespace pulpo {

// thread-local last error string (returned by pulpo_last_error)
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (observed round-robin dispatch), so give
// every XCD a contiguous run of logical ids -> neighbouring tiles (which share halos / weight panels) hit the same L2.
// Speed only, never correctness (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace pulpo

#define PULPO_REQUIRE(cond, ...) \
    do {                         \
        if (!(cond)) return pulpo::fail(-1, __VA_ARGS__); \
    } while (0)
