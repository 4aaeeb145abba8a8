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
// One workgroup of 4 waves per CU (wave = py, ONE wave per SIMD): 12 accumulator tiles (4 px x 3 dz) of a 32 ci x 32 co pair per wave,
// ~260 of the SIMD's 512 registers - the other half stays free, so the HBM-bound kernels of the other stream (BatchNorm passes) keep
// running on the same CUs while this matrix-bound kernel holds them (an 8-wave variant at 2 x 236 registers ran 3 % faster alone and
// starved everything else: the step got slower).  A workgroup STREAMS ALONG z through 8x8 (y, x) columns: per plane step it needs the transformed input planes z - 1, z, z + 1 (a ring of four
// slots) and the output-gradient plane z (two slots); the planes of step z + 1 are fetched into registers during step z - 1, transformed and
// written during step z (a quarter behind each group of MFMAs) - one barrier per plane step, no halo re-read along z.  Work = contiguous
// ranges of the linearised (column, z) plane steps, so the load balance is exact to one plane.  Flush: G^T M G (in-lane over the wave's px,
// across the waves through LDS), three float atomics per (dz, ci, co) triple into the packed scratch shared with the other wgrad kernels.
#include "conv_shared.h"

#ifndef PULPO_ABL
#define PULPO_ABL 0          // diagnostic builds (scripts/ablate.py): 21 no matrix instructions, 22 no staging writes, 23 no global loads, 24 no barrier per plane step
#endif
#ifndef PULPO_ABLX
#define PULPO_ABLX 0         // diagnostic builds of the eight-wave kernel (bit mask): 1 no barrier per plane step, 2 no staging writes, 4 no global loads, 32 / 64 (F(2x2x2) kernel) input / gradient taps from one 64 KB window,
#endif                       // 8 no flush, 16 no operand combinations
#include <stdlib.h>
#include <type_traits>

namespace {

using namespace pulpo_conv;
using f32x2 = __attribute__((ext_vector_type(2))) float;

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
    long split_stride;            // 0, or (deterministic mode) floats between the per-split copies of dwp (see WgradArgs in conv3d_wgrad.hip)
    long in_kb;                   // F(2x2x2) kernel: the same block stride for the INPUT operand (8 = channels-last)
    long go_kb;                   // F(2x2x2) kernel: floats between consecutive 8-channel blocks of a gradient voxel - 8 = channels-last, B * V * 8 (go_ps = 8) = the
                                  // channel-blocked layout [Cout / 8][B][D][H][W][8] of pulpo_bn_lrelu_bwd_apply_kb_t
};

template <int DUMMY>
__global__ __launch_bounds__(256, 1) void conv3d_k3_wgrad_w2(Wgrad2Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* VX = smem;
    float* EX = smem + W2G_VX;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kk = lane >> 5;
    const int py = wave;                                  // this wave's points: (py, px = 0..3)
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
    // X plane: 320 items (hy 0..9, channel quad q 0..7, xb 0..3), four x taps (gx = x0 - 1 + 2 xb + t) of four channels each: item A = tid
    // (hy 0..7) for every thread, item B = 256 + tid (hy 8, 9) for the first 64.  dY plane: 256 items (y 0..7, q, xb), two x taps: item = tid.
    const int it_q = tid & 7, it_xb = (tid >> 3) & 3, it_y = tid >> 5;          // shared by item A (hy = it_y) and the dY item (y = it_y)
    const bool xb_item = tid < 64;                                                // item B: hy = 8 + (tid >> 5)
    const bool x_cok = ci0 + 4 * it_q < a.Cin, e_cok = co0 + 4 * it_q < a.Cout;
    float* const xa_dst = VX + (4 * it_q) * W2G_VROW + it_y * 4 + it_xb;          // + (px * 4 + slot) * VSLOT + c * VROW
    float* const xb_dst = xa_dst + 8 * 4;                                          // hy + 8
    float* const e_dst = EX + (4 * it_q) * W2G_EROW + it_y * 4 + it_xb;           // + (px * 2 + slot) * ESLOT + c * EROW

    unsigned xa_ok = 0, xb_ok = 0, e_ok = 0;              // in-volume bits of the x taps of the three items (row in range)
    const float* xa_src = a.in;                           // tap 0 of item A in plane 0 of the column; item B is 8 rows further down
    const float* e_src = a.go;
    auto set_column = [&](int col) {
        int t = col;
        const int tx_ = t % a.ntx; t /= a.ntx;
        const int ty_ = t % a.nty;
        const int cb = t / a.nty;
        const int y0 = ty_ * 8, x0 = tx_ * 8;
        const int gy = y0 - 1 + it_y;
        xa_ok = 0; xb_ok = 0; e_ok = 0;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const bool xin = (unsigned)(x0 - 1 + 2 * it_xb + t4) < (unsigned)a.W;
            if (x_cok && xin && (unsigned)gy < (unsigned)a.H) xa_ok |= 1u << t4;
            if (xb_item && x_cok && xin && (unsigned)(gy + 8) < (unsigned)a.H) xb_ok |= 1u << t4;
        }
        xa_src = a.in + (long)cb * a.in_bs + ((long)gy * a.W + (x0 - 1 + 2 * it_xb)) * a.in_ps + ci0 + 4 * it_q;
        const int ey = y0 + it_y;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
            if (e_cok && ey < a.H && x0 + 2 * it_xb + t2 < a.W) e_ok |= 1u << t2;
        e_src = a.go + (long)cb * a.go_bs + ((long)ey * a.W + (x0 + 2 * it_xb)) * a.go_ps + co0 + 4 * it_q;
    };
    const long x_plane = (long)a.H * a.W * a.in_ps, e_plane = (long)a.H * a.W * a.go_ps;
    const long x_rows8 = 8L * a.W * a.in_ps;

    float4 xra[4], xrb[4], er[2];                         // raw registers of the planes being fetched
    auto issue_x = [&](int zp) {                          // input plane zp (zeros outside the volume)
        const bool zok = (unsigned)zp < (unsigned)a.D;
        const float* src = xa_src + (long)zp * x_plane;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            xra[t4] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (zok && ((xa_ok >> t4) & 1u)) xra[t4] = *reinterpret_cast<const float4*>(src + (long)t4 * a.in_ps);
            xrb[t4] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (zok && ((xb_ok >> t4) & 1u)) xrb[t4] = *reinterpret_cast<const float4*>(src + x_rows8 + (long)t4 * a.in_ps);
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
    // point px of an x-transformed item -> LDS (transposed: one ds_write_b32 per channel)
    auto write_x = [&](const float4 (&d)[4], float* dst, int px, int slot) {
        const float4 d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
        float4 v;
        if (px == 0) v = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
        else if (px == 1) v = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
        else if (px == 2) v = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
        else v = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
        float* o = dst + (px * 4 + slot) * W2G_VSLOT;
        o[0] = v.x; o[W2G_VROW] = v.y; o[2 * W2G_VROW] = v.z; o[3 * W2G_VROW] = v.w;
    };
    auto write_e = [&](int px, int slot) {
        const float4 d0 = er[0], d1 = er[1];
        float4 v;
        if (px == 0) v = d0;
        else if (px == 1) v = make_float4(d0.x + d1.x, d0.y + d1.y, d0.z + d1.z, d0.w + d1.w);
        else if (px == 2) v = make_float4(d0.x - d1.x, d0.y - d1.y, d0.z - d1.z, d0.w - d1.w);
        else v = make_float4(-d1.x, -d1.y, -d1.z, -d1.w);
        float* o = e_dst + (px * 2 + slot) * W2G_ESLOT;
        o[0] = v.x; o[W2G_EROW] = v.y; o[2 * W2G_EROW] = v.z; o[3 * W2G_EROW] = v.w;
    };
    // staging part k of 4: point k of the three items
    auto stage_part = [&](int k, int xslot, int eslot) {
        write_x(xra, xa_dst, k, xslot);
        write_e(k, eslot);
        if (xb_item) write_x(xrb, xb_dst, k, xslot);
    };
    // every thread "uses" its raw registers unconditionally (see conv3d_wino.hip): the compiler's wait for the loads sits in straight-line code
    auto touch_raw = [&]() {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            asm volatile("" : : "v"(xra[t4].x), "v"(xra[t4].y), "v"(xra[t4].z), "v"(xra[t4].w));
            asm volatile("" : : "v"(xrb[t4].x), "v"(xrb[t4].y), "v"(xrb[t4].z), "v"(xrb[t4].w));
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) asm volatile("" : : "v"(er[t2].x), "v"(er[t2].y), "v"(er[t2].z), "v"(er[t2].w));
    };

    f32x16 acc[4][3];                                     // [px][dz]
#pragma unroll
    for (int p = 0; p < 4; ++p)
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
    // zs - 1: stage only; a loop of their own - a conditional around the MFMAs costs a second copy of the accumulators).  Iteration j:
    // barrier; plane step j = E[j] x V[j - 1 .. j + 1]; the registers (input plane j + 2, gradient plane j + 1) are transformed and written
    // into the free slots, one point behind each of the first four groups' first MFMAs; then the loads of planes j + 3 / j + 2 are issued.
    long p = p_begin;
    while (p < p_end) {
        const int pc = (int)(p / a.D);
        const int zs = (int)(p - (long)pc * a.D);
        const int ze = (int)min((long)a.D, (long)zs + (p_end - p));
        set_column(pc);
        issue_x(zs - 1);
        er[0] = make_float4(0.f, 0.f, 0.f, 0.f); er[1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
        for (int j = zs - 3; j < zs; ++j) {
            __syncthreads();
            touch_raw();
#pragma unroll
            for (int k = 0; k < 4; ++k) stage_part(k, (j + 3) & 3, (j & 1) ^ 1);
            issue_x(j + 3);
            issue_e(j + 2);
        }
#pragma unroll 1
        for (int j = zs; j < ze; ++j) {
#if PULPO_ABL != 24
            __syncthreads();                              // staged planes visible; everybody has finished the previous iteration's reads
#endif
            const int xs_slot = (j + 3) & 3, es = j & 1;
            // eight groups (g, px) of 12 MFMAs; the eight ds_read_b128 of group k + 1 are requested before the MFMAs of group k are issued
            // (two register sets: with one wave per SIMD nothing else hides the LDS round trip).
            // Measured (round 2, scripts/ablate.py w2_*): with ONE wave per SIMD a wave's own MFMAs and its other instructions do not overlap -
            // the kernel takes matrix time + everything-else time (0.73 + 0.50 ms for 32->32 at 160^3; 12 bare MFMAs 724 clocks, the same 12
            // with 8 LDS reads and 20 VALU operations slotted one by one between them 896).  A version with every piece of side work pinned
            // between two MFMAs (volatile-asm arithmetic, branch-free loads and staging so that the loop stays one basic block) gained 3 % for
            // 36 more registers and nothing in the training step; not kept.  The lever left is a second wave per SIMD (an 8-wave workgroup
            // with half the accumulators per wave), see DESIGN.md.
            float4 av[2][3], bv[2][3], e0[2], e1[2];
            auto fetch = [&](int gi, int set) {
                const int g = gi >> 2, px = gi & 3;
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
            for (int gi = 0; gi < 8; ++gi) {
                const int set = gi & 1, px = gi & 3;
                if (gi + 1 < 8) fetch(gi + 1, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                const float4 f0 = e0[set], f1 = e1[set];
                const float ev[4] = {fmaf(c1, f1.x, c0 * f0.x), fmaf(c1, f1.y, c0 * f0.y), fmaf(c1, f1.z, c0 * f0.z), fmaf(c1, f1.w, c0 * f0.w)};
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const float4 pa_ = av[set][dz], pb_ = bv[set][dz];
                    const float v4[4] = {fmaf(sa, pb_.x, pa_.x), fmaf(sa, pb_.y, pa_.y), fmaf(sa, pb_.z, pa_.z), fmaf(sa, pb_.w, pa_.w)};
#pragma unroll
#if PULPO_ABL == 21
                    for (int s_ = 0; s_ < 4; ++s_) acc[px][dz][s_] = fmaf(v4[s_], ev[s_], acc[px][dz][s_]);
#else
                    for (int s_ = 0; s_ < 4; ++s_) acc[px][dz] = __builtin_amdgcn_mfma_f32_32x32x2f32(v4[s_], ev[s_], acc[px][dz], 0, 0, 0);
#endif
                    if (dz == 0 && gi < 4) {
                        // behind the group's first MFMAs: the registers (planes j + 2 / j + 1, requested half an iteration ago) are transformed
                        // and written, one point per group; after the fourth they are free and the next planes are requested, which leaves
                        // those loads groups 4..7 and the barrier to land
                        __builtin_amdgcn_sched_barrier(0);
                        if (gi == 0) touch_raw();
#if PULPO_ABL != 22
                        stage_part(gi, xs_slot, es ^ 1);
#endif
#if PULPO_ABL != 23
                        if (gi == 3) {
                            issue_x(j + 3);
                            issue_e(j + 2);
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        p += ze - zs;
        __syncthreads();                                  // (the next segment's warm-up overwrites the rings)
    }

    // ---- flush: dw[dz][ky][kx] = sum_py sum_px G[py][ky] G[px][kx] M[py][px][dz], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].  The px sum
    // is in-lane, the py sum meets in LDS, one dz at a time: X[py][kx][r][lane] (4 x 3 x 1024 floats = 48 KB)
    __syncthreads();
    float* X = smem;
    const int Cc = min(32, a.Cin - ci0);
#pragma unroll 1
    for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = dz == 0 ? acc[0][0][r] : dz == 1 ? acc[0][1][r] : acc[0][2][r];
            const float m1 = dz == 0 ? acc[1][0][r] : dz == 1 ? acc[1][1][r] : acc[1][2][r];
            const float m2 = dz == 0 ? acc[2][0][r] : dz == 1 ? acc[2][1][r] : acc[2][2][r];
            const float m3 = dz == 0 ? acc[3][0][r] : dz == 1 ? acc[3][1][r] : acc[3][2][r];
            const float hs = 0.5f * (m1 + m2);
            X[((py * 3 + 0) * 16 + r) * 64 + lane] = m0 + hs;
            X[((py * 3 + 1) * 16 + r) * 64 + lane] = 0.5f * (m1 - m2);
            X[((py * 3 + 2) * 16 + r) * 64 + lane] = hs + m3;
        }
        __syncthreads();
        // 3 kx x 16 r x 64 lanes = 3072 entries, 12 per thread; each yields the three ky taps
        for (int e = tid; e < 3 * 16 * 64; e += 256) {
            const int l = e & 63, r = (e >> 6) & 15, kx = e >> 10;
            float s4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) s4[q] = X[((q * 3 + kx) * 16 + r) * 64 + l];
            const int ci = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), co = co0 + (l & 31);
            if (ci < Cc && co < a.Cout) {
                const float hs = 0.5f * (s4[1] + s4[2]);
                float* d = a.dwp + (long)split * a.split_stride + ((long)((dz * 3 + 0) * 3 + kx) * a.Cin + ci0 + ci) * a.NPad + co;
                const long kystride = 3L * a.Cin * a.NPad;
                atomicAdd(d, s4[0] + hs);
                atomicAdd(d + kystride, 0.5f * (s4[1] - s4[2]));
                atomicAdd(d + 2 * kystride, hs + s4[3]);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ the same kernel with eight waves
// wave = (py, half of px): 6 accumulator tiles (2 px x 3 dz) per wave, two waves per SIMD.  The first build of this round's weight gradient
// had this shape at 2 x 236 registers and starved every kernel of the other stream; this one keeps ONE register set of operand rows (the
// second wave of the SIMD hides the LDS round trip) and stages the fetched planes early in the plane step.
template <int DUMMY>
__global__ __launch_bounds__(512, 2) void conv3d_k3_wgrad_w2x(Wgrad2Args a) {
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

    // Operand planes are read through buffer descriptors (as in conv3d_wino2p.hip): per column, every thread holds the byte offsets of its
    // taps in plane 0 of the column's batch element - OOB (beyond num_records: the load returns zeros) where the tap lies outside the volume or
    // the thread has no item - and a plane step adds the plane's offset as the instruction's scalar offset.  No branch, no zero-initialised
    // destination, no 64-bit address arithmetic inside the loop.  (host: volume bytes < 2^31)
    constexpr unsigned OOB = 0x80000000u;
    const int x_bytes = (int)((long)a.D * a.H * a.W * a.in_ps * 4), e_bytes = (int)((long)a.D * a.H * a.W * a.go_ps * 4);
    int cb = 0, y0 = 0, x0 = 0;                           // current column
    unsigned x_off[4], e_off[2];
    __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t e_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.go), 0, e_bytes, 0x00020000);
    auto set_column = [&](int col) {
        int t = col;
        const int tx_ = t % a.ntx; t /= a.ntx;
        const int ty_ = t % a.nty;
        cb = t / a.nty;
        y0 = ty_ * 8; x0 = tx_ * 8;
        x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + (long)cb * a.in_bs), 0, x_bytes, 0x00020000);
        e_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.go + (long)cb * a.go_bs), 0, e_bytes, 0x00020000);
        const int gy = y0 - 1 + x_hy;
        const bool xrow = x_item && x_cok && (unsigned)gy < (unsigned)a.H;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int gx = x0 - 1 + 2 * x_xb + t4;
            x_off[t4] = (xrow && (unsigned)gx < (unsigned)a.W) ? (unsigned)((gy * a.W + gx) * (int)a.in_ps + ci0 + 4 * x_q) * 4u : OOB;
        }
        const int ey = y0 + e_y;
        const bool erow = e_item && e_cok && ey < a.H;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int gx = x0 + 2 * e_xb + t2;
            e_off[t2] = (erow && gx < a.W) ? (unsigned)((ey * a.W + gx) * (int)a.go_ps + co0 + 4 * e_q) * 4u : OOB;
        }
    };
    const unsigned x_plane = (unsigned)((long)a.H * a.W * a.in_ps * 4), e_plane = (unsigned)((long)a.H * a.W * a.go_ps * 4);

    float4 xr[4], er[2];                                  // raw registers of the plane being fetched
    auto issue_x = [&](int zp) {                          // input plane zp (zeros outside the volume)
        const unsigned zmask = (unsigned)zp < (unsigned)a.D ? 0u : OOB;          // (wave-uniform)
        const unsigned zo = zmask ? 0u : (unsigned)zp * x_plane;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
            xr[t4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)(x_off[t4] | zmask), (int)zo, 0));
    };
    auto issue_e = [&](int zp) {                          // output-gradient plane zp
        const unsigned zmask = (unsigned)zp < (unsigned)a.D ? 0u : OOB;
        const unsigned zo = zmask ? 0u : (unsigned)zp * e_plane;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
            er[t2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(e_rs, (int)(e_off[t2] | zmask), (int)zo, 0));
    };
    // point px of the x-transformed item -> LDS (transposed: one ds_write_b32 per channel)
    auto write_x = [&](int px, int slot) {
        if (x_item) {
            // (two-wide vector arithmetic: v_pk_add_f32)
            const float4 pa_ = px == 0 ? xr[0] : px == 2 ? xr[2] : xr[1], pb_ = px == 0 ? xr[2] : px == 1 ? xr[2] : px == 2 ? xr[1] : xr[3];
            const f32x2 sg = px == 1 ? f32x2{1.f, 1.f} : f32x2{-1.f, -1.f};
            const f32x2 lo = f32x2{pa_.x, pa_.y} + sg * f32x2{pb_.x, pb_.y}, hi = f32x2{pa_.z, pa_.w} + sg * f32x2{pb_.z, pb_.w};
            const float4 v = make_float4(lo.x, lo.y, hi.x, hi.y);
            float* o = x_dst + (px * 4 + slot) * W2G_VSLOT;
            o[0] = v.x; o[W2G_VROW] = v.y; o[2 * W2G_VROW] = v.z; o[3 * W2G_VROW] = v.w;
        }
    };
    auto write_e = [&](int px, int slot) {
        if (e_item) {
            const float4 d0 = er[0], d1 = er[1];
            float4 v;
            if (px == 0) v = d0;
            else if (px == 3) v = make_float4(-d1.x, -d1.y, -d1.z, -d1.w);
            else {
                const f32x2 sg = px == 1 ? f32x2{1.f, 1.f} : f32x2{-1.f, -1.f};
                const f32x2 lo = f32x2{d0.x, d0.y} + sg * f32x2{d1.x, d1.y}, hi = f32x2{d0.z, d0.w} + sg * f32x2{d1.z, d1.w};
                v = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
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
#if !(PULPO_ABLX & 1)
            __syncthreads();                              // staged planes visible; everybody has finished the previous iteration's reads
#endif
            const int xs_slot = (j + 3) & 3, es = j & 1;
            // four groups (g, pl) of 12 MFMAs, ONE register set of operand rows (the other wave of the SIMD covers the LDS round trip - see
            // profiles/r2_mfma_probe.md - and the registers saved keep two such waves at 2 x 208).  The planes fetched during the previous step
            // are transformed and written behind the first MFMAs of groups 0 and 1 (two points each); their registers are then free and the
            // next planes are requested, which leaves those loads groups 2, 3 and the barrier to land.
            float4 av[3], bv[3], e0, e1;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const int g = gi >> 1, px = 2 * pxh + (gi & 1), pl = gi & 1;
                const float* eb = ea + (px * 2 + es) * W2G_ESLOT + g * 16;
                e0 = *reinterpret_cast<const float4*>(eb);
                e1 = *reinterpret_cast<const float4*>(eb + 4);
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const int off = (px * 4 + ((j + dz) & 3)) * W2G_VSLOT + g * 16;
                    av[dz] = *reinterpret_cast<const float4*>(va + off);
                    bv[dz] = *reinterpret_cast<const float4*>(vb + off);
                }
                __builtin_amdgcn_sched_barrier(0);
                // the group's sixteen operand combinations first, then its twelve MFMAs back to back (a v_fma -> MFMA dependency in front of
                // every MFMA stalls the issue: conv3d_wino2p.hip, same finding)
                // (two-wide vector arithmetic: v_pk_mul_f32 / v_pk_fma_f32 - ten instructions per group instead of twenty)
                const f32x2 c0v = {c0, c0}, c1v = {c1, c1}, sav = {sa, sa};
                const f32x2 e01 = __builtin_elementwise_fma(c1v, f32x2{e1.x, e1.y}, c0v * f32x2{e0.x, e0.y});
                const f32x2 e23 = __builtin_elementwise_fma(c1v, f32x2{e1.z, e1.w}, c0v * f32x2{e0.z, e0.w});
                const float ev[4] = {e01.x, e01.y, e23.x, e23.y};
                float vv[3][4];
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const float4 pa_ = av[dz], pb_ = bv[dz];
                    const f32x2 v01 = __builtin_elementwise_fma(sav, f32x2{pb_.x, pb_.y}, f32x2{pa_.x, pa_.y});
                    const f32x2 v23 = __builtin_elementwise_fma(sav, f32x2{pb_.z, pb_.w}, f32x2{pa_.z, pa_.w});
                    vv[dz][0] = v01.x; vv[dz][1] = v01.y; vv[dz][2] = v23.x; vv[dz][3] = v23.y;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
#if PULPO_ABLX & 16
                    for (int s_ = 0; s_ < 4; ++s_) acc[pl][dz] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[dz].x, e0.x, acc[pl][dz], 0, 0, 0);      // (no combinations)
#else
                    for (int s_ = 0; s_ < 4; ++s_) acc[pl][dz] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[dz][s_], ev[s_], acc[pl][dz], 0, 0, 0);
#endif
                    if (dz == 0 && gi < 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (gi == 0) touch_raw();
#if !(PULPO_ABLX & 2)
                        write_x(2 * gi, xs_slot);
                        write_x(2 * gi + 1, xs_slot);
                        write_e(2 * gi, es ^ 1);
                        write_e(2 * gi + 1, es ^ 1);
#endif
#if !(PULPO_ABLX & 4)
                        if (gi == 1) {
                            issue_x(j + 3);
                            issue_e(j + 2);
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        p += ze - zs;
        __syncthreads();                                  // (the next segment's warm-up overwrites the rings)
    }

#if PULPO_ABLX & 8
    if (acc[0][0][0] + acc[1][1][3] + acc[0][2][7] + acc[1][0][9] + acc[0][1][11] + acc[1][2][15] != 12345.678f) return;      // (no flush)
#endif
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
                float* d = a.dwp + (long)split * a.split_stride + ((long)((dz * 3 + 0) * 3 + kx) * a.Cin + ci0 + ci) * a.NPad + co;
                const long kystride = 3L * a.Cin * a.NPad;
                atomicAdd(d, s[0] + hs);
                atomicAdd(d + kystride, 0.5f * (s[1] - s[2]));
                atomicAdd(d + 2 * kystride, hs + s[3]);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ F(2x2x2,3x3x3): Winograd along z as well
// The eight-wave kernel with the z axis transformed too (round 4): output planes are taken in PAIRS (2 J, 2 J + 1), and with
//     Vz = B^T over the input planes 2 J - 1 .. 2 J + 2,   Ez = (d0, d0 + d1, d0 - d1, -d1) over the pair's gradient planes
// (the same B^T / A patterns the x axis uses at staging time), M[pz][py][px] = sum over 2x2x2 blocks of V^T E takes 64 matrix products per
// block of eight outputs instead of 3 x 16 x 2 = 96 (and 216 in the direct form); dw = G^T M G along all three axes in the flush.
// A pair step runs as TWO half steps h = 2 J, 2 J + 1, each reading the input planes h - 1, h, h + 1 - exactly the planes the (y, x) kernel's
// plane step h reads, so ring, staging cadence (one input and one gradient plane staged per half step) and barrier count per plane are its own:
//     h even:  acc[pz 0] += (P- - P+) x d0         acc[pz 1] += (P0 + P+) x (d0 + d1)
//     h odd :  acc[pz 2] += (P0 - P-) x (d0 - d1)   acc[pz 3] += (P- - P+) x (-d1)              (P-, P0, P+ = planes h - 1, h, h + 1)
// - 32 MFMAs per wave and plane instead of 48.  The z combinations are formed in registers behind the y combinations (ten ds_read_b128 per group
// of eight MFMAs); both gradient planes of the pair stay resident, so the gradient ring has four slots: 90 + 72 KB = the CU's 160 KB exactly.
// Accumulators: 2 px x 4 pz = eight tiles per wave.  Needs an even depth.
constexpr int W3G_EX = 4 * 4 * W2G_ESLOT;                          // [px][4 slots][32 co][36]
constexpr size_t W3G_LDS = (size_t)(W2G_VX + W3G_EX) * sizeof(float);     // 163,840 bytes

#ifndef PULPO_WG3_SKEW
#define PULPO_WG3_SKEW 1         // 1: the two waves of a SIMD (w, w + 4) stage their planes behind DIFFERENT groups of a half step (see the kernel below)
#endif

// SG: the group of a half step behind whose first MFMAs a wave starts transforming / writing the fetched planes (groups SG, SG + 1; the next planes'
// loads are issued in group SG + 1)
template <int SG>
__device__ __forceinline__ void wgrad_w3x_body(const Wgrad2Args& a) {
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
    // E row = dY[2 yb + ea_] + se * dY[2 yb + eb_] - ONE fused multiply-add per value: py 0: d0 (+ 0 d0), 1: d0 + d1, 2: d0 - d1, 3: -d1 = d1 - 2 d1
    const int ea_ = py == 3 ? 1 : 0, eb_ = py == 0 ? 0 : 1;
    const float se = py == 0 ? 0.f : py == 1 ? 1.f : py == 2 ? -1.f : -2.f;

    // ---- work: plane-PAIR steps [p_begin, p_end) of the linearised (column, z / 2) space of this pair's split
    const int ncol = a.B * a.nty * a.ntx;
    const int DP = a.D >> 1;
    const long nstep = (long)ncol * DP;
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
    // Channel rows sit in PERMUTED order: channel c = 4 q + k lives in row perm(c) = 8 k + (q ^ 4 (k >> 1)).  In channel order the rows of the
    // eight quads q a staging instruction writes (channel k of each quad: rows 4 q + k, 176 floats apart) fall on TWO of the 32 ds_write_b32
    // banks groups - a 4-way conflict on every staging write (SQ_LDS_BANK_CONFLICT: 36 % of the LDS cycles).  Permuted, the eight rows of
    // an instruction are 8 consecutive rows (all residues mod 8: conflict-free) and every ds_read_b128 lane group still covers all 16 residues
    // of its sixteen-byte slots.
    float* const x_dst = VX + x_q * W2G_VROW + x_hy * 4 + x_xb;                // + (px * 4 + slot) * VSLOT + perm-row offsets below
    float* const e_dst = EX + e_q * W2G_EROW + e_y * 4 + e_xb;                 // + (px * 4 + slot) * ESLOT + ...
    const int x_d4 = ((x_q ^ 4) - x_q) * W2G_VROW, e_d4 = ((e_q ^ 4) - e_q) * W2G_EROW;    // rows of channels 4 q + 2, 4 q + 3: 16 / 24 + (q ^ 4)

    // Operand planes are read through buffer descriptors (as in conv3d_wino2p.hip): per column, every thread holds the byte offsets of its
    // taps in plane 0 of the column's batch element - OOB (beyond num_records: the load returns zeros) where the tap lies outside the volume or
    // the thread has no item - and a plane step adds the plane's offset as the instruction's scalar offset.  No branch, no zero-initialised
    // destination, no 64-bit address arithmetic inside the loop.  (host: volume bytes < 2^31)
    constexpr unsigned OOB = 0x80000000u;
    // (num_records: a raw buffer load is out of range when  voffset >= num_records - soffset;  the plane offset rides in soffset, the channel block's
    //  offset of the blocked gradient layout in voffset - the extent reaches the last block's voxels)
    const int x_bytes = (int)((((long)(a.Cin + 7) / 8 - 1) * (a.in_kb == 8 ? 0 : a.in_kb) + (long)a.D * a.H * a.W * a.in_ps) * 4);
    const int e_bytes = (int)((((long)(a.Cout + 7) / 8 - 1) * (a.go_kb == 8 ? 0 : a.go_kb) + (long)a.D * a.H * a.W * a.go_ps) * 4);
    int cb = 0, y0 = 0, x0 = 0;                           // current column
    unsigned x_off[4], e_off[2];
    __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t e_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.go), 0, e_bytes, 0x00020000);
    auto set_column = [&](int col) {
        int t = col;
        const int tx_ = t % a.ntx; t /= a.ntx;
        const int ty_ = t % a.nty;
        cb = t / a.nty;
        y0 = ty_ * 8; x0 = tx_ * 8;
        x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + (long)cb * a.in_bs), 0, x_bytes, 0x00020000);
        e_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.go + (long)cb * a.go_bs), 0, e_bytes, 0x00020000);
        const int gy = y0 - 1 + x_hy;
        const bool xrow = x_item && x_cok && (unsigned)gy < (unsigned)a.H;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int gx = x0 - 1 + 2 * x_xb + t4;
            x_off[t4] = (xrow && (unsigned)gx < (unsigned)a.W) ? (unsigned)((long)((ci0 + 4 * x_q) >> 3) * a.in_kb + (gy * a.W + gx) * (int)a.in_ps + ((4 * x_q) & 7)) * 4u : OOB;
            if ((PULPO_ABLX & 32) && x_off[t4] != OOB) x_off[t4] &= 0xFFF0u;       // (diagnostic: every input tap from one 64 KB window - cache hits)
        }
        const int ey = y0 + e_y;
        const bool erow = e_item && e_cok && ey < a.H;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int gx = x0 + 2 * e_xb + t2;
            e_off[t2] = (erow && gx < a.W) ? (unsigned)((long)((co0 + 4 * e_q) >> 3) * a.go_kb + (ey * a.W + gx) * (int)a.go_ps + ((4 * e_q) & 7)) * 4u : OOB;
            if ((PULPO_ABLX & 64) && e_off[t2] != OOB) e_off[t2] &= 0xFFF0u;       // (diagnostic: the same for the gradient operand)
        }
    };
    const unsigned x_plane = (unsigned)((long)a.H * a.W * a.in_ps * 4), e_plane = (unsigned)((long)a.H * a.W * a.go_ps * 4);

    float4 xr[4], er[2];                                  // raw registers of the plane being fetched
    auto issue_x = [&](int zp) {                          // input plane zp (zeros outside the volume)
        const unsigned zmask = (unsigned)zp < (unsigned)a.D ? 0u : OOB;          // (wave-uniform)
        const unsigned zo = (zmask | (unsigned)(PULPO_ABLX & 32)) ? 0u : (unsigned)zp * x_plane;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
            xr[t4] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (int)(x_off[t4] | zmask), (int)zo, 0));
    };
    auto issue_e = [&](int zp) {                          // output-gradient plane zp
        const unsigned zmask = (unsigned)zp < (unsigned)a.D ? 0u : OOB;
        const unsigned zo = (zmask | (unsigned)(PULPO_ABLX & 64)) ? 0u : (unsigned)zp * e_plane;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
            er[t2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(e_rs, (int)(e_off[t2] | zmask), (int)zo, 0));
    };
    // point px of the x-transformed item -> LDS (transposed: one ds_write_b32 per channel)
    auto write_x = [&](int px, int slot) {
        if (x_item) {
            // (two-wide vector arithmetic: v_pk_add_f32)
            const float4 pa_ = px == 0 ? xr[0] : px == 2 ? xr[2] : xr[1], pb_ = px == 0 ? xr[2] : px == 1 ? xr[2] : px == 2 ? xr[1] : xr[3];
            const f32x2 sg = px == 1 ? f32x2{1.f, 1.f} : f32x2{-1.f, -1.f};
            const f32x2 lo = f32x2{pa_.x, pa_.y} + sg * f32x2{pb_.x, pb_.y}, hi = f32x2{pa_.z, pa_.w} + sg * f32x2{pb_.z, pb_.w};
            const float4 v = make_float4(lo.x, lo.y, hi.x, hi.y);
            float* o = x_dst + (px * 4 + slot) * W2G_VSLOT;
            o[0] = v.x; o[8 * W2G_VROW] = v.y; o[16 * W2G_VROW + x_d4] = v.z; o[24 * W2G_VROW + x_d4] = v.w;
        }
    };
    auto write_e = [&](int px, int slot) {
        if (e_item) {
            const float4 d0 = er[0], d1 = er[1];
            float4 v;
            if (px == 0) v = d0;
            else if (px == 3) v = make_float4(-d1.x, -d1.y, -d1.z, -d1.w);
            else {
                const f32x2 sg = px == 1 ? f32x2{1.f, 1.f} : f32x2{-1.f, -1.f};
                const f32x2 lo = f32x2{d0.x, d0.y} + sg * f32x2{d1.x, d1.y}, hi = f32x2{d0.z, d0.w} + sg * f32x2{d1.z, d1.w};
                v = make_float4(lo.x, lo.y, hi.x, hi.y);
            }
            float* o = e_dst + (px * 4 + slot) * W2G_ESLOT;
            o[0] = v.x; o[8 * W2G_EROW] = v.y; o[16 * W2G_EROW + e_d4] = v.z; o[24 * W2G_EROW + e_d4] = v.w;
        }
    };
    // every thread "uses" its raw registers unconditionally (see conv3d_wino.hip): the compiler's wait for the loads sits in straight-line code
    auto touch_raw = [&]() {
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) asm volatile("" : : "v"(xr[t4].x), "v"(xr[t4].y), "v"(xr[t4].z), "v"(xr[t4].w));
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) asm volatile("" : : "v"(er[t2].x), "v"(er[t2].y), "v"(er[t2].z), "v"(er[t2].w));
    };

    f32x16 acc[2][4];                                     // [px of this wave][pz]
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int pz = 0; pz < 4; ++pz)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][pz][r] = 0.f;

    // operand row addresses of this lane (floats): + (px * 4 + slot) * VSLOT + g * 16 for V, + (px * 4 + slot) * ESLOT + g * 16 for E (gradient plane q in slot q & 3)
    // (block row yb = 2 g + kk: V rows hy = 2 yb + ta / tb, E rows y = 2 yb, 2 yb + 1; four floats per row; input plane p sits in slot (p + 1) & 3)
    const int irow = 8 * (i & 3) + ((i >> 2) ^ (4 * ((i >> 1) & 1)));            // perm(i): the row of this lane's channel
    const float* va = VX + irow * W2G_VROW + (2 * kk + ta) * 4;
    const float* vb = VX + irow * W2G_VROW + (2 * kk + tb) * 4;
    const float* ea = EX + irow * W2G_EROW + (2 * kk) * 4;

#ifndef PULPO_W3_SETPRIO
#define PULPO_W3_SETPRIO 0       // 1: waves 4-7 raised for good, 2: the partners of a SIMD (w, w + 4) alternate per group (conv3d_wino3.hip)
#endif
    if (PULPO_W3_SETPRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);       // (the second-dispatched half loses every issue arbitration otherwise: MI355X_MICROARCH.md)
    // ---- main loop: column segments of pair steps [Js, Je) of this split's range, i.e. half steps h = 2 Js .. 2 Je - 1, each segment entered
    // through three warm-up half steps (stage only).  Half step h: barrier; its 32 MFMAs; the registers (input plane h + 2, gradient plane
    // h + 2) are transformed and written into the free slots; the loads of planes h + 3 are issued.
    long p = p_begin;
    while (p < p_end) {
        const int pc = (int)(p / DP);
        const int Js = (int)(p - (long)pc * DP);
        const int Je = (int)min((long)DP, (long)Js + (p_end - p));
        set_column(pc);
        const int hs = 2 * Js;
        issue_x(hs - 1);
        issue_e(hs);
#pragma unroll 1
        for (int h = hs - 3; h < hs; ++h) {               // warm-up: input planes hs - 1, hs, hs + 1 and gradient planes hs, hs + 1 (the first e write is a dummy)
            __syncthreads();
            touch_raw();
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                write_x(gi, (h + 3) & 3);
                write_e(gi, (h + 2) & 3);
            }
            issue_x(h + 3);
            issue_e(h + 3 < hs ? hs : h + 3);             // (h = hs - 3: plane hs again, h = hs - 2: hs + 1, h = hs - 1: hs + 2)
        }
        // one half step; HM = h mod 4 as a compile-time constant (every ring slot is then an immediate offset of the operand reads - computed at
        // run time they cost a dozen address additions per group of eight MFMAs); HB = h odd (pz 2, 3), else pz 0, 1
        auto half_step = [&](int h, auto hm_tag) {
            constexpr int HM = decltype(hm_tag)::value;
            constexpr bool HB = (HM & 1) != 0;
#if !(PULPO_ABLX & 1)
            __syncthreads();                              // staged planes visible; everybody has finished the previous half step's reads
#endif
            constexpr int xs_slot = (HM + 3) & 3, ew_slot = (HM + 2) & 3;
            constexpr int d0s = (HB ? HM - 1 : HM) & 3, d1s = (HB ? HM : HM + 1) & 3;      // slots of the pair's gradient planes 2 J, 2 J + 1
            float4 av[3], bv[3], e0[2], e1[2];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const int g = gi >> 1, px = 2 * pxh + (gi & 1), pl = gi & 1;
                if (PULPO_W3_SETPRIO == 2) { if (((gi + pxh) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
                const float* eb0 = ea + (px * 4 + d0s) * W2G_ESLOT + g * 16;
                const float* eb1 = ea + (px * 4 + d1s) * W2G_ESLOT + g * 16;
                e0[0] = *reinterpret_cast<const float4*>(eb0 + 4 * ea_); e1[0] = *reinterpret_cast<const float4*>(eb0 + 4 * eb_);
                e0[1] = *reinterpret_cast<const float4*>(eb1 + 4 * ea_); e1[1] = *reinterpret_cast<const float4*>(eb1 + 4 * eb_);
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const int off = (px * 4 + ((HM + dz) & 3)) * W2G_VSLOT + g * 16;
                    av[dz] = *reinterpret_cast<const float4*>(va + off);
                    bv[dz] = *reinterpret_cast<const float4*>(vb + off);
                }
                __builtin_amdgcn_sched_barrier(0);
                // y combinations of the three input planes and the two gradient planes, then the z combinations (two-wide vector arithmetic)
                const f32x2 sev = {se, se}, sav = {sa, sa};
                f32x2 P[3][2], E[2][2];
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const float4 pa_ = av[dz], pb_ = bv[dz];
                    P[dz][0] = __builtin_elementwise_fma(sav, f32x2{pb_.x, pb_.y}, f32x2{pa_.x, pa_.y});
                    P[dz][1] = __builtin_elementwise_fma(sav, f32x2{pb_.z, pb_.w}, f32x2{pa_.z, pa_.w});
                }
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    E[d][0] = __builtin_elementwise_fma(sev, f32x2{e1[d].x, e1[d].y}, f32x2{e0[d].x, e0[d].y});
                    E[d][1] = __builtin_elementwise_fma(sev, f32x2{e1[d].z, e1[d].w}, f32x2{e0[d].z, e0[d].w});
                }
                f32x2 VA[2], VB[2], EA[2], EB[2];         // operands of the half step's two z points
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (!HB) { VA[q] = P[0][q] - P[2][q]; VB[q] = P[1][q] + P[2][q]; EA[q] = E[0][q]; EB[q] = E[0][q] + E[1][q]; }
                    else { VA[q] = P[1][q] - P[0][q]; VB[q] = P[2][q] - P[0][q]; EA[q] = E[0][q] - E[1][q]; EB[q] = E[1][q]; }       // (pz 3: (P- - P+)(-d1) = (P+ - P-) d1)
                }
                const float va4[4] = {VA[0].x, VA[0].y, VA[1].x, VA[1].y}, vb4[4] = {VB[0].x, VB[0].y, VB[1].x, VB[1].y};
                const float ea4[4] = {EA[0].x, EA[0].y, EA[1].x, EA[1].y}, eb4[4] = {EB[0].x, EB[0].y, EB[1].x, EB[1].y};
                __builtin_amdgcn_sched_barrier(0);
                constexpr int PZA = HB ? 2 : 0, PZB = HB ? 3 : 1;
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) {
                    acc[pl][PZA] = __builtin_amdgcn_mfma_f32_32x32x2f32(va4[s_], ea4[s_], acc[pl][PZA], 0, 0, 0);
                    acc[pl][PZB] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb4[s_], eb4[s_], acc[pl][PZB], 0, 0, 0);
#ifndef PULPO_W3_STAGE
#define PULPO_W3_STAGE 0                                  // 0: two points behind the first MFMAs of groups 0 and 1 (the (y, x) kernel's placement), 1: all four in group 0
#endif
                    if (s_ == 0 && gi >= SG && gi < SG + (PULPO_W3_STAGE ? 1 : 2)) {
                        // the planes fetched during the previous half step are transformed and written behind the group's first MFMAs; their
                        // registers are then free and the next planes are requested.  (All four points in ONE group - 30 instead of 24 of the half
                        // step's 32 MFMAs for the loads to land - measured 1.5 - 2 % SLOWER although a build without any loads is 14 - 18 % faster:
                        // what the loads cost is not their latency.)
                        __builtin_amdgcn_sched_barrier(0);
                        if (gi == SG) touch_raw();
#if !(PULPO_ABLX & 2)
                        if (PULPO_W3_STAGE) {
                            write_x(0, xs_slot); write_x(1, xs_slot); write_x(2, xs_slot); write_x(3, xs_slot);
                            write_e(0, ew_slot); write_e(1, ew_slot); write_e(2, ew_slot); write_e(3, ew_slot);
                        } else {
                            write_x(2 * (gi - SG), xs_slot);
                            write_x(2 * (gi - SG) + 1, xs_slot);
                            write_e(2 * (gi - SG), ew_slot);
                            write_e(2 * (gi - SG) + 1, ew_slot);
                        }
#endif
#if !(PULPO_ABLX & 4)
                        if (gi == SG + (PULPO_W3_STAGE ? 0 : 1)) {
                            issue_x(h + 3);
                            issue_e(h + 3);
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        {
            using H0 = std::integral_constant<int, 0>; using H1 = std::integral_constant<int, 1>;
            using H2 = std::integral_constant<int, 2>; using H3 = std::integral_constant<int, 3>;
            int J = Js;
            if (J < Je && (J & 1)) { half_step(2 * J, H2{}); half_step(2 * J + 1, H3{}); ++J; }
#pragma unroll 1
            for (; J + 1 < Je; J += 2) {
                half_step(2 * J, H0{}); half_step(2 * J + 1, H1{});
                half_step(2 * J + 2, H2{}); half_step(2 * J + 3, H3{});
            }
            if (J < Je) { half_step(2 * J, H0{}); half_step(2 * J + 1, H1{}); }
        }
        p += Je - Js;
        __syncthreads();                                  // (the next segment's warm-up overwrites the rings)
    }

#if PULPO_ABLX & 8
    if (acc[0][0][0] + acc[1][1][3] + acc[0][2][7] + acc[1][0][9] + acc[0][1][11] + acc[1][2][15] != 12345.678f) return;      // (no flush)
#endif
    // ---- flush: first the z axis in-lane, M[dz] = sum_pz G[pz][dz] acc[pz] (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]); then, as in the
    // (y, x) kernel, dw[dz][ky][kx] = sum_py sum_px G[py][ky] G[px][kx] M[py][px][dz]: the px sum in-lane (this wave's two px), the (py, px-half)
    // sum through LDS, one dz at a time: X[wave][kx][r][lane] (8 x 3 x 1024 floats = 96 KB)
    __syncthreads();
    float* X = smem;
    const int Cc = min(32, a.Cin - ci0);
#pragma unroll 1
    for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h0 = 0.5f * (acc[0][1][r] + acc[0][2][r]), h1 = 0.5f * (acc[1][1][r] + acc[1][2][r]);
            const float m0 = dz == 0 ? acc[0][0][r] + h0 : dz == 1 ? 0.5f * (acc[0][1][r] - acc[0][2][r]) : h0 + acc[0][3][r];
            const float m1 = dz == 0 ? acc[1][0][r] + h1 : dz == 1 ? 0.5f * (acc[1][1][r] - acc[1][2][r]) : h1 + acc[1][3][r];
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
                float* d = a.dwp + (long)split * a.split_stride + ((long)((dz * 3 + 0) * 3 + kx) * a.Cin + ci0 + ci) * a.NPad + co;
                const long kystride = 3L * a.Cin * a.NPad;
                atomicAdd(d, s[0] + hs);
                atomicAdd(d + kystride, 0.5f * (s[1] - s[2]));
                atomicAdd(d + 2 * kystride, hs + s[3]);
            }
        }
        __syncthreads();
    }
}

// Two copies of the body that differ in ONE constant (as conv3d_k3_wino3_mfma, conv3d_wino3.hip): waves 0-3 stage behind groups 0 / 1 of a half
// step, their SIMD partners 4-7 behind groups 2 / 3 - one wave's staging arithmetic and LDS writes fall into the other's matrix instructions.
template <int DUMMY>
__global__ __launch_bounds__(512, 2) void conv3d_k3_wgrad_w3x(Wgrad2Args a) {
    if (PULPO_WG3_SKEW && __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) != 0) wgrad_w3x_body<2>(a);
    else wgrad_w3x_body<0>(a);
}

}  // namespace

namespace pulpo_conv {

// PULPO_WGRAD_W3=0: the (y, x) kernel for every shape (A/B switch).  Default: the F(2x2x2,3x3x3) kernel where the depth is even
bool wgrad_w3_depth_ok(int D) {
    static int w3 = -1;
    if (w3 < 0) { const char* e = getenv("PULPO_WGRAD_W3"); w3 = e ? atoi(e) : 1; }
    return w3 && D % 2 == 0 && D >= 4;
}

// launched by pulpo_conv3d_k3_wgrad (conv3d_wgrad.hip) for channels-last operands on large volumes; scratch must be zeroed by the caller
int launch_wgrad_w2(const float* in, long in_bs, long in_ps, const float* go, long go_bs, long go_ps, float* scratch, int B, int D, int H, int W,
                    int Cin, int Cout, hipStream_t st, float* slabs, int nslab, int* used_slabs, long go_kb, long in_kb) {
    Wgrad2Args a;
    a.split_stride = 0;
    a.go_kb = go_kb; a.in_kb = in_kb;
    // deterministic mode: the splits of the grid add into their own zeroed copies of the packed sums (zeroed here, once the split count is known)
    auto use_slabs = [&](int nsplit_) -> int {
        if (!slabs) return 0;
        const size_t base = (size_t)27 * Cin * npad(Cout);
        a.dwp = slabs; a.split_stride = (long)base;
        if (used_slabs) *used_slabs = nsplit_;
        hipError_t e = hipMemsetAsync(slabs, 0, (size_t)nsplit_ * base * sizeof(float), st);
        return e == hipSuccess ? 0 : pulpo::fail((int)e, "wgrad slab memset: %s", hipGetErrorString(e));
    };
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps;
    a.go = go; a.go_bs = go_bs; a.go_ps = go_ps;
    a.dwp = scratch;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.NPad = npad(Cout);
    a.nty = pulpo::cdiv(H, 8); a.ntx = pulpo::cdiv(W, 8);
    a.ncit = pulpo::cdiv(Cin, 32); a.ncot = pulpo::cdiv(Cout, 32);
    const int npair = a.ncit * a.ncot;
    const long nstep = (long)B * a.nty * a.ntx * D;
    int nsplit = std::max(1, 256 / npair);                 // one workgroup per CU
    nsplit = (int)std::min<long>(nsplit, nstep);
    if (slabs) nsplit = std::min(nsplit, nslab);
    a.nsplit = nsplit;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_w2<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2G_LDS);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(wgrad w2): %s", hipGetErrorString(e));
        attr = true;
    }
    const bool small32 = (((long)(Cin + 7) / 8 - 1) * (in_kb == 8 ? 0 : in_kb) + (long)D * H * W * in_ps) * 4 < (1L << 31) &&
                         (((long)(Cout + 7) / 8 - 1) * (go_kb == 8 ? 0 : go_kb) + (long)D * H * W * go_ps) * 4 < (1L << 31);
    if ((go_kb != 8 || in_kb != 8) && !(wgrad_w3_depth_ok(D) && small32))
        return pulpo::fail(1, "conv3d_k3_wgrad: channel-blocked operands are read by the F(2x2x2,3x3x3) kernel only (even depth, operands below 2 GiB)");
    if (wgrad_w3_depth_ok(D) && small32) {
        static bool attr3 = false;
        if (!attr3) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_w3x<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3G_LDS);
            if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(wgrad w3x): %s", hipGetErrorString(e));
            attr3 = true;
        }
        // workgroups: one per CU, each with the CU's whole LDS and 2 x 234 of a SIMD's 512 registers - on a CU it holds, no kernel of the main
        // stream that needs LDS (the BatchNorm finalize / column-sum kernels between two data-gradient launches) starts until it retires.
        // PULPO_WGRAD_W3_WGS (default 256) leaves CUs free for them.
        static int wgs3 = -1;
        if (wgs3 < 0) { const char* e = getenv("PULPO_WGRAD_W3_WGS"); wgs3 = e ? atoi(e) : 256; }
        const long nstep3 = (long)B * a.nty * a.ntx * (D / 2);
        a.nsplit = (int)std::min<long>(std::max(1, wgs3 / npair), nstep3);
        if (slabs) a.nsplit = std::min(a.nsplit, nslab);
        if (int rc = use_slabs(a.nsplit)) return rc;
        hipLaunchKernelGGL((conv3d_k3_wgrad_w3x<0>), dim3(npair * a.nsplit), dim3(512), W3G_LDS, st, a);
        return pulpo::check_launch("conv3d_k3_wgrad_w3x");
    }
    // PULPO_WGRAD_WAVES8=0: the four-wave build (one wave per SIMD, 405 registers).  Default: eight waves, two per SIMD at 2 x 198 registers -
    // 11 % faster alone (step-weighted 9.24 against 10.26 ms) and 0.6 ms per 160^3 training step (37.5 -> 36.9 ms)
    static int waves8 = -1;
    if (waves8 < 0) { const char* e = getenv("PULPO_WGRAD_WAVES8"); waves8 = e ? atoi(e) : 1; }
    // (the eight-wave kernel addresses its operands with 32-bit buffer offsets: volumes of 2 GiB and more take the four-wave kernel)
    const bool small = (long)D * H * W * in_ps * 4 < (1L << 31) && (long)D * H * W * go_ps * 4 < (1L << 31);
    if (waves8 && small) {
        static bool attr8 = false;
        if (!attr8) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_w2x<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2G_LDS);
            if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(wgrad w2x): %s", hipGetErrorString(e));
            attr8 = true;
        }
        if (int rc = use_slabs(nsplit)) return rc;
        hipLaunchKernelGGL((conv3d_k3_wgrad_w2x<0>), dim3(npair * nsplit), dim3(512), W2G_LDS, st, a);
        return pulpo::check_launch("conv3d_k3_wgrad_w2x");
    }
    if (int rc = use_slabs(nsplit)) return rc;
    hipLaunchKernelGGL((conv3d_k3_wgrad_w2<0>), dim3(npair * nsplit), dim3(256), W2G_LDS, st, a);
    return pulpo::check_launch("conv3d_k3_wgrad_w2");
}

}  // namespace pulpo_conv
