// Weight gradient of the 3x3x3 convolution with Winograd F(2x2,3x3) in (y, x), direct over the z taps - the transpose of the forward
// identity (conv3d_wino.hip):  with V = B^T d B (16 transformed points of a 4x4 input patch) and E = A dY A^T (16 combinations of the 2x2
// block of output gradients),
//     M[p][dz][ci][co] = sum over 2x2 blocks and planes z of  V_p[z + dz - 1][block][ci] * E_p[z][block][co],      dw[dz] = G^T M G
// i.e. 3 x 16 matrix products per block of four outputs instead of 4 x 27: 2.25x fewer MFMAs than the direct form, 1.5x fewer than the
// x-only form (conv3d_wgrad.hip).  All arithmetic fp32 on v_mfma_f32_32x32x2_f32.
//
// The GEMM's K index is the BLOCK, so - unlike the forward kernel, whose K is the channel - wide operand reads need block-contiguous rows
// per channel.  The images are therefore register-staged (global float4 of four channels -> x-transformed -> transposed ds_write_b32):
//     VX[px][plane slot][ci][hy 0..9][xb 0..3]   rows of 44 floats (40 + 4 pad)      EX[px][plane slot][co][y 0..7][xb 0..3]   rows of 36 floats
// and one ds_read_b128 hands a lane the four blocks xb = 0..3 of its row = the operands of four MFMA k-steps (k pair = yb 2g, 2g + 1); the y
// combination (two rows, wave-uniform coefficients) is formed in registers as in the forward kernel.  Row strides of 11 and 9 sixteen-byte
// slots are units mod 16: the 16 lanes of every ds_read_b128 group land on 16 distinct slots.
//
// One workgroup of 8 waves per CU (wave = (py, half of px)): 6 accumulator tiles (2 px x 3 dz) of a 32 ci x 32 co pair per wave.  A
// workgroup STREAMS ALONG z through 8x8 (y, x) columns: per plane step it needs the transformed input planes z - 1, z, z + 1 (a ring of four
// slots) and the output-gradient plane z (two slots); the planes of step z + 1 are fetched into registers during step z - 1, transformed and
// written during step z (a quarter behind each group of MFMAs) - one barrier per plane step, no halo re-read along z.  Work = contiguous
// ranges of the linearised (column, z) plane steps, so the load balance is exact to one plane.  Flush: G^T M G (in-lane over the wave's px,
// across the waves through LDS), three float atomics per (dz, ci, co) triple into the packed scratch shared with the other wgrad kernels.
#include "conv_shared.h"
#include <stdlib.h>

namespace {

using namespace pulpo_conv;

constexpr int W2G_VROW = 44, W2G_EROW = 36;                       // floats per (px, slot, channel) row
constexpr int W2G_VSLOT = 32 * W2G_VROW, W2G_ESLOT = 32 * W2G_EROW; // floats per (px, slot)
constexpr int W2G_VX = 4 * 4 * W2G_VSLOT;                         // [px][4 slots][32 ci][44]
constexpr int W2G_EX = 4 * 2 * W2G_ESLOT;                         // [px][2 slots][32 co][36]
constexpr size_t W2G_LDS = (size_t)(W2G_VX + W2G_EX) * sizeof(float);     // 126,976 bytes: one workgroup per CU

struct Wgrad2Args {
    const float* in;
    long in_bs, in_ps;            // channels-last: channel stride 1
    const float* go;              // output gradient, channels-last
    long go_bs, go_ps;
    float* dwp;                   // zero-initialised scratch [27][Cin][NPad], accumulated with float atomics
    int B, D, H, W, Cin, Cout, NPad;
    int nty, ntx, ncit, ncot, nsplit;
};

template <int DUMMY>
__global__ __launch_bounds__(512, 2) void conv3d_k3_wgrad_w2(Wgrad2Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* VX = smem;
    float* EX = smem + W2G_VX;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kk = lane >> 5;
    const int py = wave & 3, pxh = wave >> 2;             // this wave's points: (py, px = 2 pxh, 2 pxh + 1)
    const int lid = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int npair = a.ncit * a.ncot;
    const int pair = lid % npair, split = lid / npair;
    const int cit = pair / a.ncot, cot = pair - cit * a.ncot;
    const int ci0 = cit * 32, co0 = cot * 32;

    // y combinations: V row = X[2 yb + ta] + sa * X[2 yb + tb];   E row = c0 * dY[2 yb] + c1 * dY[2 yb + 1]
    const int ta = py == 0 ? 0 : py == 2 ? 2 : 1;
    const int tb = py == 2 ? 1 : py == 3 ? 3 : 2;
    const float sa = py == 1 ? 1.f : -1.f;
    const float c0 = py == 3 ? 0.f : 1.f;
    const float c1 = py == 0 ? 0.f : py == 1 ? 1.f : -1.f;

    // ---- work: plane steps [p_begin, p_end) of the linearised (column, z) space of this pair's split
    const int ncol = a.B * a.nty * a.ntx;
    const long nstep = (long)ncol * a.D;
    const long p_begin = nstep * split / a.nsplit, p_end = nstep * (split + 1) / a.nsplit;

    // ---- staging items of this thread (column-invariant geometry)
    // X plane: items (hy 0..9, channel quad q 0..7, xb 0..3) = 320 of 512 threads; four x taps (gx = x0 - 1 + 2 xb + t) of four channels
    const bool x_item = tid < 320;
    const int x_q = tid & 7, x_xb = (tid >> 3) & 3, x_hy = tid >> 5;
    // dY plane: items (y 0..7, channel quad q, xb) = 256 threads (the upper half: threads 256..511, so that the two kinds of item are spread)
    const bool e_item = tid >= 256;
    const int e_t = tid - 256;
    const int e_q = e_t & 7, e_xb = (e_t >> 3) & 3, e_y = (e_t >> 5) & 7;
    const bool x_cok = ci0 + 4 * x_q < a.Cin, e_cok = co0 + 4 * e_q < a.Cout;
    float* const x_dst = VX + (4 * x_q) * W2G_VROW + x_hy * 4 + x_xb;          // + (px * 4 + slot) * VSLOT + c * VROW
    float* const e_dst = EX + (4 * e_q) * W2G_EROW + e_y * 4 + e_xb;           // + (px * 2 + slot) * ESLOT + c * EROW

    int cb = 0, y0 = 0, x0 = 0;                           // current column
    unsigned x_ok = 0;                                    // in-volume bits of the four x taps (row in range), e_ok likewise (two taps)
    unsigned e_ok = 0;
    const float* x_src = a.in;                            // tap 0 of this thread's X item in plane 0 of the column
    const float* e_src = a.go;
    auto set_column = [&](int col) {
        int t = col;
        const int tx_ = t % a.ntx; t /= a.ntx;
        const int ty_ = t % a.nty;
        cb = t / a.nty;
        y0 = ty_ * 8; x0 = tx_ * 8;
        const int gy = y0 - 1 + x_hy;
        x_ok = 0;
        if (x_item && x_cok && (unsigned)gy < (unsigned)a.H)
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4)
                if ((unsigned)(x0 - 1 + 2 * x_xb + t4) < (unsigned)a.W) x_ok |= 1u << t4;
        x_src = a.in + (long)cb * a.in_bs + ((long)gy * a.W + (x0 - 1 + 2 * x_xb)) * a.in_ps + ci0 + 4 * x_q;
        const int ey = y0 + e_y;
        e_ok = 0;
        if (e_item && e_cok && ey < a.H)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
                if (x0 + 2 * e_xb + t2 < a.W) e_ok |= 1u << t2;
        e_src = a.go + (long)cb * a.go_bs + ((long)ey * a.W + (x0 + 2 * e_xb)) * a.go_ps + co0 + 4 * e_q;
    };
    const long x_plane = (long)a.H * a.W * a.in_ps, e_plane = (long)a.H * a.W * a.go_ps;

    float4 xr[4], er[2];                                  // raw registers of the plane being fetched
    auto issue_x = [&](int zp) {                          // input plane zp (zeros outside the volume)
        const bool zok = (unsigned)zp < (unsigned)a.D;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            xr[t4] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (zok && ((x_ok >> t4) & 1u)) xr[t4] = *reinterpret_cast<const float4*>(x_src + (long)zp * x_plane + (long)t4 * a.in_ps);
        }
    };
    auto issue_e = [&](int zp) {                          // output-gradient plane zp
        const bool zok = (unsigned)zp < (unsigned)a.D;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            er[t2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (zok && ((e_ok >> t2) & 1u)) er[t2] = *reinterpret_cast<const float4*>(e_src + (long)zp * e_plane + (long)t2 * a.go_ps);
        }
    };
    // point px of the x-transformed item -> LDS (transposed: one ds_write_b32 per channel)
    auto write_x = [&](int px, int slot) {
        if (x_item) {
            const float4 d0 = xr[0], d1 = xr[1], d2 = xr[2], d3 = xr[3];
            float4 v;
            if (px == 0) v = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
            else if (px == 1) v = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
            else if (px == 2) v = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
            else v = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
            float* o = x_dst + (px * 4 + slot) * W2G_VSLOT;
            o[0] = v.x; o[W2G_VROW] = v.y; o[2 * W2G_VROW] = v.z; o[3 * W2G_VROW] = v.w;
        }
    };
    auto write_e = [&](int px, int slot) {
        if (e_item) {
            const float4 d0 = er[0], d1 = er[1];
            float4 v;
            if (px == 0) v = d0;
            else if (px == 1) v = make_float4(d0.x + d1.x, d0.y + d1.y, d0.z + d1.z, d0.w + d1.w);
            else if (px == 2) v = make_float4(d0.x - d1.x, d0.y - d1.y, d0.z - d1.z, d0.w - d1.w);
            else v = make_float4(-d1.x, -d1.y, -d1.z, -d1.w);
            float* o = e_dst + (px * 2 + slot) * W2G_ESLOT;
            o[0] = v.x; o[W2G_EROW] = v.y; o[2 * W2G_EROW] = v.z; o[3 * W2G_EROW] = v.w;
        }
    };
    // every thread "uses" its raw registers unconditionally (see conv3d_wino.hip): the compiler's wait for the loads sits in straight-line code
    auto touch_raw = [&]() {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) asm volatile("" : : "v"(xr[t4].x), "v"(xr[t4].y), "v"(xr[t4].z), "v"(xr[t4].w));
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) asm volatile("" : : "v"(er[t2].x), "v"(er[t2].y), "v"(er[t2].z), "v"(er[t2].w));
    };

    f32x16 acc[2][3];                                     // [px of this wave][dz]
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][dz][r] = 0.f;

    // operand row addresses of this lane (floats): + (px * 4 + slot) * VSLOT + g * 16 for V, + (px * 2 + slot) * ESLOT + g * 16 for E
    // (block row yb = 2 g + kk: V rows hy = 2 yb + ta / tb, E rows y = 2 yb, 2 yb + 1; four floats per row; input plane p sits in slot (p + 1) & 3)
    const float* va = VX + i * W2G_VROW + (2 * kk + ta) * 4;
    const float* vb = VX + i * W2G_VROW + (2 * kk + tb) * 4;
    const float* ea = EX + i * W2G_EROW + (2 * kk) * 4;

    // ---- main loop: column segments [zs, ze) of this split's plane-step range, each entered through three warm-up iterations (j = zs - 3 ..
    // zs - 1: stage only) so that the loads have exactly one definition inside the loop (a second, conditional one costs register copies and a
    // vmcnt(0) on the back edge).  Iteration j: barrier; [j >= zs: plane step j = E[j] x V[j - 1 .. j + 1]]; the registers (input plane j + 2,
    // gradient plane j + 1) are transformed and written into the free slots; the loads of planes j + 3 / j + 2 are issued.
    long p = p_begin;
    while (p < p_end) {
        const int pc = (int)(p / a.D);
        const int zs = (int)(p - (long)pc * a.D);
        const int ze = (int)min((long)a.D, (long)zs + (p_end - p));
        set_column(pc);
        issue_x(zs - 1);
        er[0] = make_float4(0.f, 0.f, 0.f, 0.f); er[1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int j = zs - 3; j < zs; ++j) {               // warm-up: stage only (its own loop: the accumulators must not see a conditional)
            __syncthreads();
            touch_raw();
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                write_x(gi, (j + 3) & 3);
                write_e(gi, (j & 1) ^ 1);
            }
            issue_x(j + 3);
            issue_e(j + 2);
        }
#pragma unroll 1
        for (int j = zs; j < ze; ++j) {
            __syncthreads();                              // staged planes visible; everybody has finished the previous iteration's reads
            touch_raw();
            const int xs_slot = (j + 3) & 3, es = j & 1;
            {
                // four groups (g, pl) of 12 MFMAs; the eight ds_read_b128 of group k + 1 are requested before the MFMAs of group k are issued
                float4 av[2][3], bv[2][3], e0[2], e1[2];
                auto fetch = [&](int gi, int set) {
                    const int g = gi >> 1, px = 2 * pxh + (gi & 1);
                    const float* eb = ea + (px * 2 + es) * W2G_ESLOT + g * 16;
                    e0[set] = *reinterpret_cast<const float4*>(eb);
                    e1[set] = *reinterpret_cast<const float4*>(eb + 4);
#pragma unroll
                    for (int dz = 0; dz < 3; ++dz) {
                        const int off = (px * 4 + ((j + dz) & 3)) * W2G_VSLOT + g * 16;
                        av[set][dz] = *reinterpret_cast<const float4*>(va + off);
                        bv[set][dz] = *reinterpret_cast<const float4*>(vb + off);
                    }
                };
                fetch(0, 0);
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    const int set = gi & 1, pl = gi & 1;
                    if (gi + 1 < 4) fetch(gi + 1, set ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                    const float4 f0 = e0[set], f1 = e1[set];
                    const float ev[4] = {fmaf(c1, f1.x, c0 * f0.x), fmaf(c1, f1.y, c0 * f0.y), fmaf(c1, f1.z, c0 * f0.z), fmaf(c1, f1.w, c0 * f0.w)};
#pragma unroll
                    for (int dz = 0; dz < 3; ++dz) {
                        const float4 pa_ = av[set][dz], pb_ = bv[set][dz];
                        const float v4[4] = {fmaf(sa, pb_.x, pa_.x), fmaf(sa, pb_.y, pa_.y), fmaf(sa, pb_.z, pa_.z), fmaf(sa, pb_.w, pa_.w)};
#pragma unroll
                        for (int s_ = 0; s_ < 4; ++s_) acc[pl][dz] = __builtin_amdgcn_mfma_f32_32x32x2f32(v4[s_], ev[s_], acc[pl][dz], 0, 0, 0);
                        if (dz == 0) {
                            // staging quarter gi: point gi of both items, behind the group's first MFMAs
                            __builtin_amdgcn_sched_barrier(0);
                            write_x(gi, xs_slot);
                            write_e(gi, es ^ 1);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            issue_x(j + 3);
            issue_e(j + 2);
        }
        p += ze - zs;
        __syncthreads();                                  // (the next segment's warm-up overwrites the rings)
    }

    // ---- flush: dw[dz][ky][kx] = sum_py sum_px G[py][ky] G[px][kx] M[py][px][dz], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].  The px sum
    // is in-lane (this wave's two px), the (py, px-half) sum meets in LDS, one dz at a time: X[wave][kx][r][lane] (8 x 3 x 1024 floats = 96 KB)
    __syncthreads();
    float* X = smem;
    const int Cc = min(32, a.Cin - ci0);
#pragma unroll 1
    for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = dz == 0 ? acc[0][0][r] : dz == 1 ? acc[0][1][r] : acc[0][2][r];
            const float m1 = dz == 0 ? acc[1][0][r] : dz == 1 ? acc[1][1][r] : acc[1][2][r];
            float t0, t1, t2;                             // kx = 0, 1, 2 from px = 2 pxh (m0), 2 pxh + 1 (m1)
            if (pxh == 0) { t0 = m0 + 0.5f * m1; t1 = 0.5f * m1; t2 = 0.5f * m1; }               // px 0: G = (1,0,0); px 1: (.5,.5,.5)
            else { t0 = 0.5f * m0; t1 = -0.5f * m0; t2 = 0.5f * m0 + m1; }                          // px 2: (.5,-.5,.5); px 3: (0,0,1)
            X[((wave * 3 + 0) * 16 + r) * 64 + lane] = t0;
            X[((wave * 3 + 1) * 16 + r) * 64 + lane] = t1;
            X[((wave * 3 + 2) * 16 + r) * 64 + lane] = t2;
        }
        __syncthreads();
        // 3 kx x 16 r x 64 lanes = 3072 entries, 6 per thread; each yields the three ky taps
        for (int e = tid; e < 3 * 16 * 64; e += 512) {
            const int l = e & 63, r = (e >> 6) & 15, kx = e >> 10;
            float s[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)                   // py = q: waves q (px half 0) and q + 4 (px half 1)
                s[q] = X[((q * 3 + kx) * 16 + r) * 64 + l] + X[(((q + 4) * 3 + kx) * 16 + r) * 64 + l];
            const int ci = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), co = co0 + (l & 31);
            if (ci < Cc && co < a.Cout) {
                const float hs = 0.5f * (s[1] + s[2]);
                float* d = a.dwp + ((long)((dz * 3 + 0) * 3 + kx) * a.Cin + ci0 + ci) * a.NPad + co;
                const long kystride = 3L * a.Cin * a.NPad;
                atomicAdd(d, s[0] + hs);
                atomicAdd(d + kystride, 0.5f * (s[1] - s[2]));
                atomicAdd(d + 2 * kystride, hs + s[3]);
            }
        }
        __syncthreads();
    }
}

}  // namespace

namespace pulpo_conv {

// launched by pulpo_conv3d_k3_wgrad (conv3d_wgrad.hip) for channels-last operands on large volumes; scratch must be zeroed by the caller
int launch_wgrad_w2(const float* in, long in_bs, long in_ps, const float* go, long go_bs, long go_ps, float* scratch, int B, int D, int H, int W,
                    int Cin, int Cout, hipStream_t st) {
    Wgrad2Args a;
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps;
    a.go = go; a.go_bs = go_bs; a.go_ps = go_ps;
    a.dwp = scratch;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.NPad = npad(Cout);
    a.nty = pulpo::cdiv(H, 8); a.ntx = pulpo::cdiv(W, 8);
    a.ncit = pulpo::cdiv(Cin, 32); a.ncot = pulpo::cdiv(Cout, 32);
    const int npair = a.ncit * a.ncot;
    const long nstep = (long)B * a.nty * a.ntx * D;
    int nsplit = std::max(1, 256 / npair);                 // one 512-thread workgroup per CU
    nsplit = (int)std::min<long>(nsplit, nstep);
    a.nsplit = nsplit;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_w2<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2G_LDS);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(wgrad w2): %s", hipGetErrorString(e));
        attr = true;
    }
    hipLaunchKernelGGL((conv3d_k3_wgrad_w2<0>), dim3(npair * nsplit), dim3(512), W2G_LDS, st, a);
    return pulpo::check_launch("conv3d_k3_wgrad_w2");
}

}  // namespace pulpo_conv
