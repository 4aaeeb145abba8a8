// Element-group loads / stores for activation tensors that live in HBM as fp32 or as bf16 (the "bf16 activation storage" of BASELINE
// configs 4-5: conv outputs y, activations z and their gradients are STORED as bf16, every kernel computes in fp32).
// A group is N consecutive channels of one voxel: N = 4 / 8 are single 8- / 16-byte (bf16) or 16- / 2 x 16-byte (fp32) accesses.
#pragma once
#include "common.h"

namespace pulpo {

typedef uint16_t bf16_t;                      // storage type of a bf16 element (the C ABI passes void* + a dtype code)
enum ActDtype { kF32 = 0, kBF16 = 1 };

__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t p = {(__bf16)lo, (__bf16)hi};    // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

// ---- loads
template <int N>
__device__ __forceinline__ void ldv(const float* p, float (&v)[N]) {
    if constexpr (N == 1) {
        v[0] = *p;
    } else {
        static_assert(N % 4 == 0, "groups of 1, 4 or 8");
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(p + 4 * q);
            v[4 * q + 0] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    }
}
template <int N>
__device__ __forceinline__ void ldv(const bf16_t* p, float (&v)[N]) {
    if constexpr (N == 1) {
        v[0] = bf2f(*p);
    } else if constexpr (N == 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = bf2f(t.x & 0xffffu); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = bf2f(t.y & 0xffffu); v[3] = __uint_as_float(t.y & 0xffff0000u);
    } else {
        static_assert(N == 8, "groups of 1, 4 or 8");
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        v[0] = bf2f(t.x & 0xffffu); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = bf2f(t.y & 0xffffu); v[3] = __uint_as_float(t.y & 0xffff0000u);
        v[4] = bf2f(t.z & 0xffffu); v[5] = __uint_as_float(t.z & 0xffff0000u);
        v[6] = bf2f(t.w & 0xffffu); v[7] = __uint_as_float(t.w & 0xffff0000u);
    }
}

// ---- stores (bf16: round to nearest even)
template <int N>
__device__ __forceinline__ void stv(float* p, const float (&v)[N]) {
    if constexpr (N == 1) {
        *p = v[0];
    } else {
#pragma unroll
        for (int q = 0; q < N / 4; ++q) *reinterpret_cast<float4*>(p + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}
template <int N>
__device__ __forceinline__ void stv(bf16_t* p, const float (&v)[N]) {
    if constexpr (N == 1) {
        *p = f2bf(v[0]);
    } else if constexpr (N == 4) {
        *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
    } else {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
    }
}

// what a value becomes when it is stored as T and read back (statistics of a tensor must describe the tensor as stored)
template <typename T>
__device__ __forceinline__ float as_stored(float f) {
    if constexpr (sizeof(T) == 2) return bf2f(f2bf(f));
    else return f;
}

template <typename T>
constexpr int elem_bytes() { return (int)sizeof(T); }

// largest group (8, 4 or 1 channels) every operand of a launch supports: C divisible, strides divisible, base pointers aligned to the
// group's bytes.  ptr/ps pairs are (address, voxel stride in elements, element bytes).
struct GroupProbe {
    int C;
    bool ok4 = true, ok8 = true;
    explicit GroupProbe(int C_) : C(C_) { ok4 = C % 4 == 0; ok8 = C % 8 == 0; }
    void add(const void* p, long ps, int esize) {
        if (p == nullptr) return;
        const uintptr_t a = (uintptr_t)p;
        if (ps % 4 != 0 || (a % (4 * esize)) != 0) ok4 = false;
        if (ps % 8 != 0 || (a % (esize == 2 ? 16 : 16)) != 0) ok8 = false;     // (fp32 groups of 8 are two 16-byte accesses)
    }
};

}  // namespace pulpo

// dispatch helpers: run BODY with TA (and TB) bound to float / pulpo::bf16_t according to the dtype codes
#define PULPO_DISPATCH_DT(dt, TA, ...)                                  \
    do {                                                                \
        if ((dt) == 0) { using TA = float; __VA_ARGS__; }               \
        else { using TA = pulpo::bf16_t; __VA_ARGS__; }                 \
    } while (0)
#define PULPO_REQUIRE_DT(dt, what) PULPO_REQUIRE((dt) == 0 || (dt) == 1, what ": dtype code 0 (fp32) or 1 (bf16) expected")
