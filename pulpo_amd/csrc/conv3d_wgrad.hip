// Weight gradient of the 3x3x3 convolution:  D[(tap,cin)][cout] += A[(tap,cin)][voxel] * B[voxel][cout]  (the GEMM's K is the voxel index).
// Persistent workgroups (one per CU) fed by LDS-DMA, direct form (conv3d_k3_wgrad_mfma) and Winograd F(2,3)-along-x form
// (conv3d_k3_wgrad_wino); atomic flush into a packed scratch, then unpack (optionally accumulating into the parameter's .grad).
#include "conv_shared.h"
#include "../../include/pulpo_hip.h"
#include <stdlib.h>

#ifndef PULPO_ABL
#define PULPO_ABL 0          // diagnostic ablation builds (scripts/ablate.py): 11 no in-loop DMA, 12 no matrix instructions, 13 no flush
#endif

namespace {

using namespace pulpo_conv;

// 64 bytes of zeros: source address of out-of-volume / out-of-channel lanes of an LDS-DMA piece
__device__ float4 g_zero_page[4];

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS starting at the WAVE-UNIFORM address lds_piece
__device__ __forceinline__ void dma16(const float* src, float* lds_piece) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct WgradArgs {
    const float* in;
    long in_bs, in_ps, in_cs;
    const float* dy;
    long dy_bs, dy_ps, dy_cs;
    float* dwp;                   // zero-initialised scratch [27][Cin][NPad], accumulated with float atomics
    int B, D, H, W, Cin, Cout, NPad;
    int ntz, nty, ntx, ncit, ncot, nsplit;
    long split_stride;            // 0, or (deterministic mode) floats between the per-split copies of dwp: split s adds into dwp + s * split_stride,
                                  // which no other workgroup of the same (ci tile, co tile) touches - every element receives ONE add
    // conv3d_k3_wgrad_smallc with the BatchNorm / LeakyReLU backward of its own ConvUnit fused into the operand staging (bn_y != nullptr):
    // `dy` then holds dz (the gradient of the unit's OUTPUT, fp32 or bf16: dz_bf16) and the kernel forms
    //     dy = scale * lrelu'(scale * y + shift) * dz + B * (y - m32) + C      (bn_lrelu_bwd_apply_kernel's arithmetic, norm_act.hip)
    // per element while it stages the tile - the pass that writes dy (and reads dz and y once more) disappears where nothing else needs dy:
    // the input layer, whose data gradient nobody asks for.  bn_part2[split][Cout] receives the column sums of dy (the conv bias gradient).
    const float* bn_y;
    long bn_y_bs, bn_y_ps;
    const float* bn_coef;         // the unit's coefficient block (pulpo_bn_fwd_finalize)
    const double* bn_totd;        // [2][Cout]: mean(dbn), mean(dbn * xhat) (pulpo_bn_bwd_finalize)
    float* bn_part2;
    float slope;
    int dz_bf16;
    long dz_kb;                   // floats between consecutive 8-channel blocks of a dz voxel: 8 = channels-last, else the channel-blocked layout (fp32 dz only)
};

constexpr int WG_CH = 32, WG_NT = 32, WG_CP = WG_CH + 1;

// NTW = row tiles (32 (tap,ci) pairs each) per wave: 7 for a full 32-channel ci tile (27 tiles over 4 waves), fewer for
// narrow inputs.  The MFMAs of the hot loop are unconditional (rows beyond the matrix compute garbage that is never flushed), so the loop
// body is one basic block and the compiler can run the LDS reads ahead of the matrix pipe.
template <bool VEC, int NTW>
__global__ __launch_bounds__(256, VEC ? 1 : 2) void conv3d_k3_wgrad_mfma(WgradArgs a) {
    constexpr int XS = (HV * WG_CP + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int npair = a.ncit * a.ncot;
    const int pair = lid % npair, split = lid / npair;
    const int cit = pair / a.ncot, cot = pair - cit * a.ncot;
    const int ci0 = cit * WG_CH, co0 = cot * WG_NT;
    const int Cc = min(WG_CH, a.Cin - ci0);
    const int rows = 27 * Cc;
    const int nrt = (rows + 31) >> 5;                 // row tiles of 32 (tap,ci) pairs; host guarantees nrt <= 4 * NTW
    const int i = lane & 31, kk = lane >> 5;
    constexpr int STR = VEC ? 32 : WG_CP;             // voxel stride of the halo image (DMA image is unpadded)

    int rowoff[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int r = 32 * (wave + 4 * u) + i;
        const int tap = r < rows ? r / Cc : 0, ci = r < rows ? r - tap * Cc : 0;
        rowoff[u] = tap_halo_offset(tap) * STR + ci;
    }
    f32x16 acc[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    const int ntile = a.B * a.ntz * a.nty * a.ntx;
    const int per = (ntile + a.nsplit - 1) / a.nsplit;
    const int t_begin = split * per, t_end = min(ntile, t_begin + per);

    auto decode = [&](int tl, int& b, int& z0, int& y0, int& x0) {
        int t = tl;
        const int tx_ = t % a.ntx; t /= a.ntx;
        const int ty_ = t % a.nty; t /= a.nty;
        const int tz_ = t % a.ntz;
        b = t / a.ntz;
        z0 = tz_ * TZ; y0 = ty_ * TY; x0 = tx_ * TX;
    };
    // One "body" = 4 voxel-pair steps.  Rows beyond the matrix (last partial row tile, or a whole spare tile) read valid
    // LDS words of tap 0 and accumulate garbage into accumulator rows that the flush never writes: MFMA rows are independent,
    // so no masking is needed.
    auto load_body = [&](float (&A)[4][NTW], float (&Bv)[4], const float* xs_c, const float* dys_c, int sb) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int vox = 2 * (sb * 4 + q4) + kk;
            const int hbk = (((vox >> 6) * HY + ((vox >> 3) & 7)) * HX + (vox & 7)) * STR;
            Bv[q4] = dys_c[vox * WG_NT + i];
#pragma unroll
            for (int u = 0; u < NTW; ++u) A[q4][u] = xs_c[rowoff[u] + hbk];
        }
    };
    auto mma_body = [&](const float (&A)[4][NTW], const float (&Bv)[4]) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int u = 0; u < NTW; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[q4][u], Bv[q4], acc[u], 0, 0, 0);
    };
    float A0[4][NTW], A1[4][NTW], B0[4], B1[4];       // register double buffer of the fragments

    if constexpr (VEC) {
        // LDS-DMA pipeline.  The A-operand lanes index consecutive (tap, ci) rows, so the halo image needs no padding
        // ([halo voxel][32 ci], 128 B per voxel) and is filled by global_load_lds: no staging registers.  Two image sets
        // (X 50 KiB + dY 16 KiB each) ping-pong: the 17 DMA pieces of tile t+1 are issued one per 4 voxel-pair steps inside
        // the MFMA loop of tile t (their address arithmetic hides behind the matrix pipe) and are drained by the
        // s_waitcnt vmcnt(0) + barrier at the tile boundary.
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const float* zero = reinterpret_cast<const float*>(g_zero_page);
        constexpr int XIMG = 13 * 4 * 256;            // halo image padded to 52 DMA pieces (50 used): every piece is unconditional
        constexpr int SET = XIMG + MV * WG_NT;        // floats per image set
        int nb = 0, nz0 = 0, ny0 = 0, nx0 = 0;        // next tile
        bool more = false;
        // piece pc in [0, 17): 0..12 = halo (piece 12 of waves 2,3 is padding), 13..16 = dY.  Branch-free: lanes with nothing
        // to fetch (out of volume / channel range / no next tile / padding) source the zero page.
        auto issue_piece = [&](int pc, float* xd) {
            if (pc < 13) {
                const int j = tid + pc * 256;
                const int hv = j >> 3, q = j & 7;
                const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
                const int gz = nz0 - 1 + hz, gy = ny0 - 1 + hy, gx = nx0 - 1 + hx;
                const bool ok = more && hv < HV && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W &&
                                ci0 + 4 * q < a.Cin;
                const float* src = ok ? a.in + (long)nb * a.in_bs + ((long)(gz * a.H + gy) * a.W + gx) * a.in_ps + ci0 + 4 * q : zero;
                dma16(src, xd + (wave_u + 4 * pc) * 256);
            } else {
                const int u = pc - 13;
                const int j = tid + u * 256;
                const int vv = j >> 3, q = j & 7;
                const int gz = nz0 + (vv >> 6), gy = ny0 + ((vv >> 3) & 7), gx = nx0 + (vv & 7);
                const bool ok = more && gz < a.D && gy < a.H && gx < a.W && co0 + 4 * q < a.Cout;
                const float* src = ok ? a.dy + (long)nb * a.dy_bs + ((long)(gz * a.H + gy) * a.W + gx) * a.dy_ps + co0 + 4 * q : zero;
                dma16(src, xd + XIMG + (wave_u + 4 * u) * 256);
            }
        };
        if (t_begin < t_end) {
            more = true;
            decode(t_begin, nb, nz0, ny0, nx0);
#pragma unroll
            for (int pc = 0; pc < 17; ++pc) issue_piece(pc, smem);
        }
        int cur = 0;
        for (int tl = t_begin; tl < t_end; ++tl, cur ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of tile tl have landed
            __syncthreads();      // (a) everybody's pieces have landed  (b) everybody left the other image set
            more = tl + 1 < t_end;
            if (more) decode(tl + 1, nb, nz0, ny0, nx0);
            const float* xs_c = smem + cur * SET;
            const float* dys_c = xs_c + XIMG;
            float* xnext = smem + (cur ^ 1) * SET;
            load_body(A0, B0, xs_c, dys_c, 0);
#pragma unroll
            for (int sb = 0; sb < 16; sb += 2) {            // fully unrolled: 16 bodies of 4 steps, fragments one body ahead
                load_body(A1, B1, xs_c, dys_c, sb + 1);
                __builtin_amdgcn_sched_barrier(0);
                // the next tile's 17 DMA pieces go out during the first nine bodies, so they have half a tile of MFMAs to land
                if (sb < 8) { issue_piece(2 * sb, xnext); issue_piece(2 * sb + 1, xnext); }
                if (sb == 8) issue_piece(16, xnext);
                mma_body(A0, B0);
                load_body(A0, B0, xs_c, dys_c, (sb + 2) & 15);   // (wraps to body 0 on the last trip: harmless re-read)
                __builtin_amdgcn_sched_barrier(0);
                if (sb < 8) { issue_piece(2 * sb + 2, xnext); issue_piece(2 * sb + 3, xnext); }
                mma_body(A1, B1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the (zero-sourced) pieces issued during the last tile
    } else {
        // scalar-staging path (planar / narrow inputs: the 2- and 3-channel first layers).  Only the Cc real channels of the
        // halo are staged (rows of other channels are never flushed); dY goes through float4 when it is channels-last.
        float* xs = smem;
        float* dys = smem + XS;
        const bool dyvec = (a.dy_cs == 1) && (a.dy_ps % 4 == 0) && (a.dy_bs % 4 == 0) && (a.Cout % 4 == 0) && (((uintptr_t)a.dy & 15) == 0);
        for (int tl = t_begin; tl < t_end; ++tl) {
            int b, z0, y0, x0;
            decode(tl, b, z0, y0, x0);
            __syncthreads();
            const float* in_b = a.in + (long)b * a.in_bs;
            for (int j = tid; j < HV * Cc; j += 256) {
                const int c = j / HV, hv = j - c * HV;              // voxel fastest: coalesced for planar inputs
                const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
                const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                float v = 0.f;
                if ((unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                    v = in_b[((long)(gz * a.H + gy) * a.W + gx) * a.in_ps + (long)(ci0 + c) * a.in_cs];
                xs[hv * WG_CP + c] = v;
            }
            const float* dyb = a.dy + (long)b * a.dy_bs;
            if (dyvec) {
                float4 val[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = tid + u * 256;
                    const int vv = j >> 3, q = j & 7;
                    const int gz = z0 + (vv >> 6), gy = y0 + ((vv >> 3) & 7), gx = x0 + (vv & 7);
                    val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (gz < a.D && gy < a.H && gx < a.W && co0 + 4 * q < a.Cout)
                        val[u] = *reinterpret_cast<const float4*>(dyb + ((long)(gz * a.H + gy) * a.W + gx) * a.dy_ps + co0 + 4 * q);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = tid + u * 256;
                    *reinterpret_cast<float4*>(dys + (j >> 3) * WG_NT + 4 * (j & 7)) = val[u];
                }
            } else {
                for (int j = tid; j < MV * WG_NT; j += 256) {
                    const int vv = j >> 5, c = j & 31;
                    const int gz = z0 + (vv >> 6), gy = y0 + ((vv >> 3) & 7), gx = x0 + (vv & 7);
                    float val = 0.f;
                    if (gz < a.D && gy < a.H && gx < a.W && co0 + c < a.Cout)
                        val = dyb[((long)(gz * a.H + gy) * a.W + gx) * a.dy_ps + (long)(co0 + c) * a.dy_cs];
                    dys[vv * WG_NT + c] = val;
                }
            }
            __syncthreads();
            load_body(A0, B0, xs, dys, 0);
#pragma unroll 1
            for (int sb = 0; sb < 16; sb += 2) {
                load_body(A1, B1, xs, dys, sb + 1);
                __builtin_amdgcn_sched_barrier(0);
                mma_body(A0, B0);
                load_body(A0, B0, xs, dys, (sb + 2) & 15);
                __builtin_amdgcn_sched_barrier(0);
                mma_body(A1, B1);
            }
        }
    }

    // flush: one 128-byte run of couts per (tap, ci) row -> float atomics at full rate
    const int co = co0 + i;
    if (co < a.Cout) {
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            if (wave + 4 * u < nrt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rg = 32 * (wave + 4 * u) + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    if (rg < rows) {
                        const int tap = rg / Cc, ci = rg - tap * Cc;
                        atomicAdd(a.dwp + (long)split * a.split_stride + ((long)tap * a.Cin + ci0 + ci) * a.NPad + co, acc[u][r]);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient, 2-/3-channel inputs
// The input layers (image pair: 2 channels, latent sample: 3): the GEMM has only 27 * Cin <= 108 rows, i.e. NRT <= 4 row tiles, too few
// to give each wave row tiles of its own (conv3d_k3_wgrad_mfma<false, 1> left two of four waves multiplying padding).  Here every wave
// owns ALL row tiles and one z-plane (64 voxels = 32 k-steps) of a 4 x 8 x 8 voxel tile - the waves split the GEMM's K -, the next tile's
// operands are fetched into registers underneath the current tile's MFMAs, and the four partial sums meet in LDS before one atomic flush
// per workgroup.  x: planar or strided (scalar loads, one channel plane per run of lanes); dy: channels-last, 16-byte aligned.
template <int NRT, bool BN = false>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wgrad_smallc(WgradArgs a) {
    constexpr int HV4 = 6 * HY * HX, XP = HV4 + 4;          // halo voxels of a 4 x 8 x 8 tile; plane stride of the planar halo image
    constexpr int NIX = (4 * HV4 + 255) / 256;              // scalar halo loads per thread (<= 4 channels)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                        // [Cin][XP]
    float* dys = smem + 4 * XP;                              // [256 voxels][32 couts]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kk = lane >> 5;
    const int lid = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int cot = lid % a.ncot, split = lid / a.ncot;
    const int co0 = cot * WG_NT;
    const int rows = 27 * a.Cin;

    int rowoff[NRT];
#pragma unroll
    for (int u = 0; u < NRT; ++u) {
        const int r = 32 * u + i;
        const int tap = r < rows ? r / a.Cin : 0, ci = r < rows ? r - tap * a.Cin : 0;
        rowoff[u] = ci * XP + tap_halo_offset(tap);
    }
    f32x16 acc[NRT];
#pragma unroll
    for (int u = 0; u < NRT; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    const int ntile = a.B * a.ntz * a.nty * a.ntx;            // (ntz counts 4-plane tiles here)
    const int per = (ntile + a.nsplit - 1) / a.nsplit;
    const int t_begin = split * per, t_end = min(ntile, t_begin + per);

    float xr[NIX];
    float4 dr[8];
    // ---- BN: this thread's four channels (co0 + 4 q .. + 3, q = tid & 7 for every piece it stages) and their constants
    float4 yr[BN ? 8 : 1];
    unsigned vmask = 0;                                     // piece u of the fetched tile lies inside the volume
    float k_sc[4], k_sh[4], k_m[4], k_b[4], k_chi[4], k_clo[4], bsum[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BN) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = co0 + 4 * (tid & 7) + k;
            k_sc[k] = k_sh[k] = k_m[k] = k_b[k] = k_chi[k] = k_clo[k] = 0.f;
            if (ch < a.Cout) {
                const int C = a.Cout;
                const double* cd = reinterpret_cast<const double*>(a.bn_coef + 4 * C);
                const float sc_ = a.bn_coef[2 * C + ch], m32_ = a.bn_coef[ch];
                const double mean = cd[ch], rstd = cd[C + ch], c1 = a.bn_totd[ch], c2 = a.bn_totd[C + ch];
                const double b = -(double)sc_ * c2 * rstd;
                const double cc = -(double)sc_ * (c1 + c2 * rstd * ((double)m32_ - mean));
                k_sc[k] = sc_; k_sh[k] = a.bn_coef[3 * C + ch]; k_m[k] = m32_; k_b[k] = (float)b; k_chi[k] = (float)cc;
                k_clo[k] = (float)(cc - (double)k_chi[k]);
            }
        }
    }
    auto fetch = [&](int tl) {
        int t = tl;
        const int tx_ = t % a.ntx; t /= a.ntx;
        const int ty_ = t % a.nty; t /= a.nty;
        const int tz_ = t % a.ntz;
        const int b = t / a.ntz;
        const int z0 = tz_ * 4, y0 = ty_ * TY, x0 = tx_ * TX;
        const float* in_b = a.in + (long)b * a.in_bs;
#pragma unroll
        for (int u = 0; u < NIX; ++u) {
            const int j = tid + u * 256;
            const int c = j / HV4, hv = j - c * HV4;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            xr[u] = 0.f;
            if (c < a.Cin && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
                xr[u] = in_b[((long)(gz * a.H + gy) * a.W + gx) * a.in_ps + (long)c * a.in_cs];
        }
        const float* dyb = a.dy + (long)b * a.dy_bs;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = tid + u * 256;
            const int vv = j >> 3, q = j & 7;
            const int gz = z0 + (vv >> 6), gy = y0 + ((vv >> 3) & 7), gx = x0 + (vv & 7);
            dr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool ok = gz < a.D && gy < a.H && gx < a.W && co0 + 4 * q < a.Cout;
            if constexpr (BN) {
                yr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                vmask = u == 0 ? 0u : vmask;
                if (ok) {
                    const long vox = (long)(gz * a.H + gy) * a.W + gx;
                    vmask |= 1u << u;
                    yr[u] = *reinterpret_cast<const float4*>(a.bn_y + (long)b * a.bn_y_bs + vox * a.bn_y_ps + co0 + 4 * q);
                    if (a.dz_bf16) {
                        const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(a.dy) + (long)b * a.dy_bs + vox * a.dy_ps + co0 + 4 * q);
                        dr[u] = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xffff0000u));
                    } else {
                        dr[u] = *reinterpret_cast<const float4*>(dyb + (long)((co0 + 4 * q) >> 3) * a.dz_kb + vox * a.dy_ps + ((4 * q) & 7));
                    }
                }
            } else if (ok) {
                dr[u] = *reinterpret_cast<const float4*>(dyb + ((long)(gz * a.H + gy) * a.W + gx) * a.dy_ps + co0 + 4 * q);
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < NIX; ++u) {
            const int j = tid + u * 256;
            const int c = j / HV4, hv = j - c * HV4;
            if (c < 4) xs[c * XP + hv] = xr[u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = tid + u * 256;
            if constexpr (BN) {
                // dy = scale * dbn + B * (y - m32) + C (C as a (hi, lo) pair), zero outside the volume; its column sums ride along
                const float g_[4] = {dr[u].x, dr[u].y, dr[u].z, dr[u].w}, v_[4] = {yr[u].x, yr[u].y, yr[u].z, yr[u].w};
                float o_[4];
                const bool in = (vmask >> u) & 1u;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float bn = v_[k] * k_sc[k] + k_sh[k];
                    const float sg = k_sc[k] * (bn > 0.f ? g_[k] : g_[k] * a.slope);
                    const float lin = fmaf(k_b[k], v_[k] - k_m[k], k_chi[k]);
                    o_[k] = in ? (sg + lin) + k_clo[k] : 0.f;
                    bsum[k] += o_[k];
                }
                dr[u] = make_float4(o_[0], o_[1], o_[2], o_[3]);
            }
            *reinterpret_cast<float4*>(dys + (j >> 3) * WG_NT + 4 * (j & 7)) = dr[u];
        }
    };

    if (t_begin < t_end) fetch(t_begin);
    for (int tl = t_begin; tl < t_end; ++tl) {
        __syncthreads();                                  // every wave has left the images of the previous tile
        stash();
        __syncthreads();
        if (tl + 1 < t_end) fetch(tl + 1);                // in flight underneath this tile's MFMAs
        const float* xw = xs + wave * (HY * HX);          // this wave's z-plane of the tile
        const float* dw_ = dys + wave * 64 * WG_NT + i;
#pragma unroll 4
        for (int s2 = 0; s2 < 32; ++s2) {
            const int vox = 2 * s2 + kk;                  // voxel of the plane: (y, x) = (vox >> 3, vox & 7)
            const int hb = (vox >> 3) * HX + (vox & 7);
            const float bv = dw_[vox * WG_NT];
#pragma unroll
            for (int u = 0; u < NRT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(xw[rowoff[u] + hb], bv, acc[u], 0, 0, 0);
        }
    }

    // the four waves' partial sums meet in LDS (one wave after the other), then one atomic per (row, cout) and workgroup
    float* red = dys;                                     // [NRT * 32 rows][32 couts]
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int u = 0; u < NRT; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * u + (r & 3) + 8 * (r >> 2) + 4 * kk;
                    float* q = red + row * WG_NT + i;
                    *q = (w == 0 ? 0.f : *q) + acc[u][r];
                }
        }
    }
    __syncthreads();
    for (int j = tid; j < rows * WG_NT; j += 256) {
        const int row = j >> 5, co = co0 + (j & 31);
        if (co < a.Cout) {
            const int tap = row / a.Cin, ci = row - tap * a.Cin;
            atomicAdd(a.dwp + (long)split * a.split_stride + ((long)tap * a.Cin + ci) * a.NPad + co, red[j]);
        }
    }
    if constexpr (BN) {
        // column sums of dy over this workgroup's tiles: the 32 threads that share a channel quad, in thread order (deterministic)
        __syncthreads();
        float* bs = dys;                                  // [256 threads][4]
        *reinterpret_cast<float4*>(bs + tid * 4) = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
        __syncthreads();
        if (tid < WG_NT && co0 + tid < a.Cout) {
            const int q = tid >> 2, k = tid & 3;
            float t = 0.f;
            for (int r = 0; r < 32; ++r) t += bs[(q + 8 * r) * 4 + k];
            a.bn_part2[(long)split * a.Cout + co0 + tid] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient, Winograd along x
// The transpose of the forward identity: with V = B^T d (four transformed inputs per x-pair) and E = A dy (four combinations of the
// pair's two output gradients), M_p[(dz,dy,ci)][co] = sum over x-pairs V_p * E_p and dw[.., t] = G^T M - 36 instead of 54 matrix
// products per x-pair.  Wave p of the workgroup owns transformed point p (all nine (dz, dy) row tiles of its 32-channel slice), so
// both operand transforms are wave-uniform two-term combinations formed from the raw LDS-DMA images as the fragments are read
// (A: two ds_read + one fma, B: two ds_read + two fma per nine MFMAs), and every wave adds its share of G^T M at the flush.
// Same persistent one-workgroup-per-CU LDS-DMA pipeline as conv3d_k3_wgrad_mfma<true, NTW>.
template <int NTW>
__global__ __launch_bounds__(256, 1) void conv3d_k3_wgrad_wino(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lid = pulpo::xcd_remap(blockIdx.x, gridDim.x);
    const int npair = a.ncit * a.ncot;
    const int pair = lid % npair, split = lid / npair;
    const int cit = pair / a.ncot, cot = pair - cit * a.ncot;
    const int ci0 = cit * WG_CH, co0 = cot * WG_NT;
    const int Cc = min(WG_CH, a.Cin - ci0);
    const int rows = 9 * Cc;
    const int nrt = (rows + 31) >> 5;                 // host guarantees nrt <= NTW
    const int i = lane & 31, kk = lane >> 5;
    const int pt = __builtin_amdgcn_readfirstlane(wave);
    // A = X[x + ta] + sa * X[x + tb];  B = c0 * dY[x] + c1 * dY[x + 1]
    const int ta = pt == 0 ? 0 : pt == 2 ? 2 : 1;
    const int tb = pt == 2 ? 1 : pt == 3 ? 3 : 2;
    const float sa = pt == 1 ? 1.f : -1.f;
    const float c0 = pt == 3 ? 0.f : 1.f;
    const float c1 = pt == 0 ? 0.f : pt == 1 ? 1.f : -1.f;

    int rowoff[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int r = 32 * u + i;
        const int zy = r < rows ? r / Cc : 0, ci = r < rows ? r - zy * Cc : 0;
        rowoff[u] = ((zy / 3) * HY + zy % 3) * HX * 32 + ci;
    }
    f32x16 acc[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;

    const int ntile = a.B * a.ntz * a.nty * a.ntx;
    const int per = (ntile + a.nsplit - 1) / a.nsplit;
    const int t_begin = split * per, t_end = min(ntile, t_begin + per);
    auto decode = [&](int tl, int& b, int& z0, int& y0, int& x0) {
        int t = tl;
        const int tx_ = t % a.ntx; t /= a.ntx;
        const int ty_ = t % a.nty; t /= a.nty;
        const int tz_ = t % a.ntz;
        b = t / a.ntz;
        z0 = tz_ * TZ; y0 = ty_ * TY; x0 = tx_ * TX;
    };
    // one body = 2 K-steps of two x-pairs each; x-pair b = (z, y, xb) with xb fastest: 64 per 2x8x8 tile = 16 bodies.
    // The RAW operand pairs are fetched one body ahead; the two-term combinations are formed right in front of the MFMA that
    // consumes them (a VALU op in the shadow of the previous MFMA), so no LDS latency is exposed between bodies.
    constexpr int NQ = 2;
    // per-lane LDS addresses are loop invariant (row offset, the lane's half of the x-pair couple, the point's two taps); the position
    // of the K-step inside the tile is a compile-time constant of the unrolled body => every ds_read is base register + immediate
    int offa[NTW], offb[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        offa[u] = rowoff[u] + kk * 64 + ta * 32;
        offb[u] = rowoff[u] + kk * 64 + tb * 32;
    }
    const int offy = kk * 2 * WG_NT + i;
    auto load_body = [&](float (&Ra)[NQ][NTW], float (&Rb)[NQ][NTW], float (&Y0)[NQ], float (&Y1)[NQ], const float* xs_c, const float* dys_c, int sb) {
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) {
            const int blk = 2 * (sb * NQ + q4);                 // even x-pair of the couple; the odd one is 2 voxels further along x
            const int z = blk >> 5, y = (blk >> 2) & 7, xb = blk & 3;
            const int hbk = ((z * HY + y) * HX + 2 * xb) * 32;
            const int v0 = ((z * 8 + y) * 8 + 2 * xb) * WG_NT;
            Y0[q4] = dys_c[offy + v0];
            Y1[q4] = dys_c[offy + v0 + WG_NT];
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                Ra[q4][u] = xs_c[offa[u] + hbk];
                Rb[q4][u] = xs_c[offb[u] + hbk];
            }
        }
    };
    auto mma_body = [&](const float (&Ra)[NQ][NTW], const float (&Rb)[NQ][NTW], const float (&Y0)[NQ], const float (&Y1)[NQ]) {
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) {
            const float bv = fmaf(c1, Y1[q4], c0 * Y0[q4]);
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const float av = fmaf(sa, Rb[q4][u], Ra[q4][u]);
#if PULPO_ABL == 12
                acc[u][0] += av * bv;
#else
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u], 0, 0, 0);
#endif
            }
        }
    };
    float Ra0[NQ][NTW], Rb0[NQ][NTW], Ra1[NQ][NTW], Rb1[NQ][NTW], Ya0[NQ], Yb0[NQ], Ya1[NQ], Yb1[NQ];

    const int wave_u = pt;
    const float* zero = reinterpret_cast<const float*>(g_zero_page);
    constexpr int XIMG = 13 * 4 * 256;
    constexpr int SET = XIMG + MV * WG_NT;
    int nb = 0, nz0 = 0, ny0 = 0, nx0 = 0;
    bool more = false;
    auto issue_piece = [&](int pc, float* xd) {
        if (pc < 13) {
            const int j = tid + pc * 256;
            const int hv = j >> 3, q = j & 7;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = nz0 - 1 + hz, gy = ny0 - 1 + hy, gx = nx0 - 1 + hx;
            const bool ok = more && hv < HV && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W &&
                            ci0 + 4 * q < a.Cin;
            const float* src = ok ? a.in + (long)nb * a.in_bs + ((long)(gz * a.H + gy) * a.W + gx) * a.in_ps + ci0 + 4 * q : zero;
            dma16(src, xd + (wave_u + 4 * pc) * 256);
        } else {
            const int u = pc - 13;
            const int j = tid + u * 256;
            const int vv = j >> 3, q = j & 7;
            const int gz = nz0 + (vv >> 6), gy = ny0 + ((vv >> 3) & 7), gx = nx0 + (vv & 7);
            const bool ok = more && gz < a.D && gy < a.H && gx < a.W && co0 + 4 * q < a.Cout;
            const float* src = ok ? a.dy + (long)nb * a.dy_bs + ((long)(gz * a.H + gy) * a.W + gx) * a.dy_ps + co0 + 4 * q : zero;
            dma16(src, xd + XIMG + (wave_u + 4 * u) * 256);
        }
    };
    if (t_begin < t_end) {
        more = true;
        decode(t_begin, nb, nz0, ny0, nx0);
#pragma unroll
        for (int pc = 0; pc < 17; ++pc) issue_piece(pc, smem);
    }
    int cur = 0;
    for (int tl = t_begin; tl < t_end; ++tl, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        more = tl + 1 < t_end;
        if (more) decode(tl + 1, nb, nz0, ny0, nx0);
        const float* xs_c = smem + cur * SET;
        const float* dys_c = xs_c + XIMG;
        float* xnext = smem + (cur ^ 1) * SET;
        load_body(Ra0, Rb0, Ya0, Yb0, xs_c, dys_c, 0);
#pragma unroll
        for (int sb = 0; sb < 16; sb += 2) {            // 16 bodies; one DMA piece of the next tile per body (+ the 17th on the last trip)
            // the raw reads of the next body are threaded through the MFMAs of the current one (1 MFMA : 3 LDS reads : 2 VALU), so
            // neither their issue slots nor their latency stall the matrix pipe of this single-wave-per-SIMD kernel
            load_body(Ra1, Rb1, Ya1, Yb1, xs_c, dys_c, sb + 1);
#if PULPO_ABL != 11
            if (sb < 8) { issue_piece(2 * sb, xnext); issue_piece(2 * sb + 1, xnext); }     // all 17 pieces in the first nine bodies
            if (sb == 8) issue_piece(16, xnext);
#endif
            mma_body(Ra0, Rb0, Ya0, Yb0);
#pragma unroll
            for (int g = 0; g < NQ * NTW; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_body(Ra0, Rb0, Ya0, Yb0, xs_c, dys_c, (sb + 2) & 15);
#if PULPO_ABL != 11
            if (sb < 8) { issue_piece(2 * sb + 2, xnext); issue_piece(2 * sb + 3, xnext); }
#endif
            mma_body(Ra1, Rb1, Ya1, Yb1);
#pragma unroll
            for (int g = 0; g < NQ * NTW; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // flush: dw[.., t] = G^T M needs the four points of a (dz, dy, ci, co) entry, which live in the four waves.  They meet in LDS (the image
    // sets are free now), eight row tiles at a time, and each entry leaves as three float atomics (tap 0: M0 + (M1 + M2)/2, tap 1:
    // (M1 - M2)/2, tap 2: (M1 + M2)/2 + M3) in 128-byte runs of couts - 2.7x fewer atomics than flushing every wave's share separately.
#if PULPO_ABL == 13
    if (a.Cin != -7) return;
#endif
    __syncthreads();
    float* X = smem;                                       // [point][tile slot 0..7][r][lane]
    const int co = co0 + i;
    constexpr int GRP = 8;
    for (int g0 = 0; g0 < nrt; g0 += GRP) {
        const int ng = min(GRP, nrt - g0);
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            if (u >= g0 && u < g0 + ng) {
#pragma unroll
                for (int r = 0; r < 16; ++r) X[((pt * GRP + (u - g0)) * 16 + r) * 64 + lane] = acc[u][r];
            }
        }
        __syncthreads();
        for (int j = wave; j < ng * 16; j += 4) {
            const int tl = j >> 4, r = j & 15;
            const float m0 = X[((0 * GRP + tl) * 16 + r) * 64 + lane], m1 = X[((1 * GRP + tl) * 16 + r) * 64 + lane];
            const float m2 = X[((2 * GRP + tl) * 16 + r) * 64 + lane], m3 = X[((3 * GRP + tl) * 16 + r) * 64 + lane];
            const int rg = 32 * (g0 + tl) + (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (rg < rows && co < a.Cout) {
                const int zy = rg / Cc, ci = rg - zy * Cc;
                float* d = a.dwp + (long)split * a.split_stride + ((long)(zy * 3) * a.Cin + ci0 + ci) * a.NPad + co;
                const float hs = 0.5f * (m1 + m2);
                atomicAdd(d, m0 + hs);
                atomicAdd(d + (long)a.Cin * a.NPad, 0.5f * (m1 - m2));
                atomicAdd(d + 2L * a.Cin * a.NPad, hs + m3);
            }
        }
        __syncthreads();
    }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cin, int Cout, int NPad, long total, int accumulate) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(e % 27);
        const long r = e / 27;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        const float val = dwp[((long)tap * Cin + ci) * NPad + co];
        dw[e] = accumulate ? dw[e] + val : val;
    }
}

// deferred parameter-gradient jobs of a whole backward pass in ONE launch (grid = (blocks per job, jobs)):
//   kind 0: dw[co][ci][27] += packed[tap][ci][co] and packed := 0   (a = Cin, b = Cout, c = NPad)     - weight gradients of pulpo_conv3d_k3_wgrad(accumulate = 2)
//   kind 1: dst[col] += sum_r src[r * stride + col]                (a = rows, b = columns, c = row stride or 0 = b)  - conv-bias gradients from the
//           BatchNorm backward partials; column ranges of the 1x1x1 heads' partial rows (weights and biases of mu / sigma / velocity heads)
__global__ __launch_bounds__(256) void grad_finish_multi_kernel(const PulpoGradJob* __restrict__ jobs) {
    const PulpoGradJob j = jobs[blockIdx.y];
    if (j.kind == 0) {
        // transposed through LDS: a block takes one (ci, 32 couts) strip = 27 x 32 packed values read in 128-byte runs (and zeroed), written
        // to dw[co][ci][0..26] in 108-byte runs
        __shared__ float tile[27][33];
        float* packed = const_cast<float*>(j.src);
        const int ncog = (j.b + 31) / 32;
        const long nstrip = (long)j.a * ncog;
        for (long sidx = blockIdx.x; sidx < nstrip; sidx += gridDim.x) {
            const int ci = (int)(sidx % j.a), co0 = (int)(sidx / j.a) * 32;
            for (int e = threadIdx.x; e < 27 * 32; e += blockDim.x) {
                const int tap = e >> 5, c = e & 31;
                float v = 0.f;
                if (co0 + c < j.b) {
                    const long src = ((long)tap * j.a + ci) * j.c + co0 + c;
                    v = packed[src];
                    packed[src] = 0.f;
                }
                tile[tap][c] = v;
            }
            __syncthreads();
            for (int e = threadIdx.x; e < 27 * 32; e += blockDim.x) {
                const int c = e / 27, tap = e - c * 27;
                if (co0 + c < j.b) j.dst[((long)(co0 + c) * j.a + ci) * 27 + tap] += tile[tap][c];
            }
            __syncthreads();
        }
    } else {
        // 8 columns x 32 row lanes per workgroup (up to 2048 rows over 32 - 288 columns: with 32 columns x 8 row lanes a single workgroup
        // per 32 columns walked 256 dependent passes - most of this launch's 170 us)
        __shared__ double red[32][9];
        const int cx = threadIdx.x & 7, ry = threadIdx.x >> 3;
        const long rs = j.c > 0 ? j.c : j.b;           // row stride of src (c = 0: the rows are dense)
        for (int c0 = blockIdx.x * 8; c0 < j.b; c0 += gridDim.x * 8) {
            const int c = c0 + cx;
            double s_ = 0.0;
            if (c < j.b) {
#pragma unroll 4
                for (int r = ry; r < j.a; r += 32) s_ += (double)j.src[(long)r * rs + c];
            }
            red[ry][cx] = s_;
            __syncthreads();
            if (ry == 0 && c < j.b) {
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < 32; ++k) t += red[k][cx];
                j.dst[c] += (float)t;
            }
            __syncthreads();
        }
    }
}

}  // namespace

PULPO_API int pulpo_grad_finish_multi(const PulpoGradJob* jobs, int njobs, void* stream) {
    PULPO_REQUIRE(jobs && njobs > 0, "grad_finish_multi: bad arguments");
    hipLaunchKernelGGL(grad_finish_multi_kernel, dim3(192, njobs), dim3(256), 0, (hipStream_t)stream, jobs);
    return pulpo::check_launch("grad_finish_multi");
}

int pulpo_conv::launch_unpack_wgrad(const float* packed, float* dw, int Cin, int Cout, int accumulate, hipStream_t st) {
    const long total = (long)Cout * Cin * 27;
    const int ub = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(ub), dim3(256), 0, st, packed, dw, Cin, Cout, npad(Cout), total, accumulate);
    return pulpo::check_launch("unpack_wgrad");
}

// ================================================================================================ C ABI
PULPO_API size_t pulpo_conv3d_k3_wgrad_scratch_floats(int Cin, int Cout) { return (size_t)27 * Cin * npad(Cout); }

// weight-gradient kernel for a shape with 16-byte-vectorisable channels-last operands (vec != 0) or not: 3 = Winograd F(2x2x2,3x3x3) (even depths),
// 2 = Winograd F(2x2,3x3) in (y, x) (both conv3d_wgrad_w2.hip), 1 = Winograd F(2,3) along x, 0 = direct; PULPO_WGRAD_WINOGRAD = 0 / 1 caps
// the choice, PULPO_WGRAD_W3=0 keeps the (y, x) kernel (A/B runs)
PULPO_API int pulpo_conv3d_k3_wgrad_algo(int B, int D, int H, int W, int Cin, int Cout, int vec) {
    static int cap = -1;
    if (cap < 0) { const char* e = getenv("PULPO_WGRAD_WINOGRAD"); cap = e ? atoi(e) : 2; }
    static long minvox = -1;                                                   // PULPO_WGRAD_MIN_VOXELS: smallest volume for the Winograd forms (A/B runs)
    if (minvox < 0) { const char* e = getenv("PULPO_WGRAD_MIN_VOXELS"); minvox = e ? atol(e) : 1000; }
    const bool big = vec && Cin >= 8 && (long)D * H * W >= minvox;             // (measured: the (y, x) form also wins on the 20^3 and 10^3 pyramid levels)
    const int algo = big ? std::min(cap, 2) : 0;
    return algo == 2 && pulpo_conv::wgrad_w3_depth_ok(D) ? 3 : algo;           // (operands of 2 GiB and more still take the (y, x) kernel)
}

// dw[Cout][Cin][27] (+)= sum_vox in[vox+tap-1][ci] * dy[vox][co]  (accumulate != 0 adds to dw, e.g. a parameter's .grad storage).
// scratch: pulpo_conv3d_k3_wgrad_scratch_floats floats.
static int wgrad_impl(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dy, int64_t dy_bs,
                      int64_t dy_ps, int64_t dy_cs, float* dw, int accumulate, float* scratch, float* slabs, int nslab, int B, int D, int H, int W,
                      int Cin, int Cout, void* stream, const WgradArgs* bnf = nullptr, int64_t dy_kb = 8, int64_t in_kb = 8) {
    PULPO_REQUIRE(in && dy && scratch && (dw || accumulate == 2), "conv3d_k3_wgrad: null pointer");
    PULPO_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3d_k3_wgrad: bad dims");
    hipStream_t st = (hipStream_t)stream;
    WgradArgs a;
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps; a.in_cs = in_cs;
    a.dy = dy; a.dy_bs = dy_bs; a.dy_ps = dy_ps; a.dy_cs = dy_cs;
    a.dwp = scratch;
    a.bn_y = nullptr; a.bn_y_bs = a.bn_y_ps = 0; a.bn_coef = nullptr; a.bn_totd = nullptr; a.bn_part2 = nullptr; a.slope = 0.f; a.dz_bf16 = 0; a.dz_kb = 8;
    if (bnf) {
        a.bn_y = bnf->bn_y; a.bn_y_bs = bnf->bn_y_bs; a.bn_y_ps = bnf->bn_y_ps; a.bn_coef = bnf->bn_coef; a.bn_totd = bnf->bn_totd;
        a.bn_part2 = bnf->bn_part2; a.slope = bnf->slope; a.dz_bf16 = bnf->dz_bf16; a.dz_kb = bnf->dz_kb ? bnf->dz_kb : 8;
    }
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.NPad = npad(Cout);
    a.ntz = pulpo::cdiv(D, TZ); a.nty = pulpo::cdiv(H, TY); a.ntx = pulpo::cdiv(W, TX);
    a.ncit = pulpo::cdiv(Cin, WG_CH); a.ncot = pulpo::cdiv(Cout, WG_NT);
    const int ntile = B * a.ntz * a.nty * a.ntx;
    const int npair = a.ncit * a.ncot;
    const bool vec = !a.dz_bf16 && (in_cs == 1) && (in_ps % 4 == 0) && (in_bs % 4 == 0) && (Cin % 4 == 0) && (((uintptr_t)in & 15) == 0) &&
                     (dy_cs == 1) && (dy_ps % 4 == 0) && (dy_bs % 4 == 0) && (Cout % 4 == 0) && (((uintptr_t)dy & 15) == 0);
    // DMA variant: one resident workgroup per CU -> one round of <= 256 persistent workgroups (fewest atomic flushes);
    // scalar variant: two per CU
    int nsplit = std::max(1, (vec ? 256 : 512) / npair);
    nsplit = std::min(nsplit, ntile);
    if (slabs) nsplit = std::min(nsplit, nslab);
    a.nsplit = nsplit;
    const bool deferred = accumulate == 2;                 // scratch arrives zeroed and keeps the packed sums: the caller unpacks later
    if (!deferred) {
        hipError_t e = hipMemsetAsync(scratch, 0, pulpo_conv3d_k3_wgrad_scratch_floats(Cin, Cout) * sizeof(float), st);
        if (e != hipSuccess) return pulpo::fail((int)e, "wgrad memset: %s", hipGetErrorString(e));
    }
    // Deterministic mode: the kernels' float atomics land in per-split copies of the packed sums (one add per element and copy: order-free),
    // which finish() adds to `scratch` in fixed order.
    const long base = (long)pulpo_conv3d_k3_wgrad_scratch_floats(Cin, Cout);
    a.split_stride = 0;
    if (slabs) { a.dwp = slabs; a.split_stride = base; }
    int used = nsplit;                                     // copies the launch below writes (the other TU's launcher reports its own count)
    auto prepare = [&](int n) -> int {
        if (!slabs) return 0;
        hipError_t e = hipMemsetAsync(slabs, 0, (size_t)n * base * sizeof(float), st);
        return e == hipSuccess ? 0 : pulpo::fail((int)e, "wgrad slab memset: %s", hipGetErrorString(e));
    };
    auto finish = [&](int rc0) -> int {
        if (rc0) return rc0;
        if (slabs) {
            const int r = pulpo_conv::launch_wgrad_slab_reduce(scratch, slabs, used, base, st, npad(Cout), Cout);
            if (r) return r;
        }
        if (deferred) return 0;
        return pulpo_conv::launch_unpack_wgrad(scratch, dw, Cin, Cout, accumulate, st);
    };
    const int nrt_max = (27 * std::min(Cin, WG_CH) + 31) / 32;
    const int ntw = (nrt_max + 3) / 4;                 // 1..7
    constexpr size_t lds = (size_t)(((HV * WG_CP + 3) & ~3) + MV * WG_NT) * sizeof(float);
    constexpr size_t lds_dma = (size_t)2 * (13 * 4 * 256 + MV * WG_NT) * sizeof(float);
    const int nblk = npair * nsplit;
    int rc = 0;
#define PULPO_WGRAD(VECV, NTWV)                                                                                                   \
    {                                                                                                                             \
        static bool attr = false;                                                                                                 \
        const size_t bytes = VECV ? lds_dma : lds;                                                                                \
        if (!attr) {                                                                                                              \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_mfma<VECV, NTWV>),                 \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);                        \
            if (ea != hipSuccess) return pulpo::fail((int)ea, "hipFuncSetAttribute(wgrad): %s", hipGetErrorString(ea));           \
            attr = true;                                                                                                          \
        }                                                                                                                         \
        hipLaunchKernelGGL((conv3d_k3_wgrad_mfma<VECV, NTWV>), dim3(nblk), dim3(256), bytes, st, a);                              \
    }
    const int algo = bnf ? 0 : pulpo_conv3d_k3_wgrad_algo(B, D, H, W, Cin, Cout, (int)vec);
    if (algo >= 2) {
        // F(2x2,3x3) in (y, x) / F(2x2x2,3x3x3), register-staged transposed operand images, z-streaming workgroups (conv3d_wgrad_w2.hip)
        // (its launcher picks its own split count: it zeroes the copies it will use and reports how many)
        rc = pulpo_conv::launch_wgrad_w2(in, in_bs, in_ps, dy, dy_bs, dy_ps, scratch, B, D, H, W, Cin, Cout, st, slabs, nslab, &used, (long)dy_kb, (long)in_kb);
        return finish(rc);
    }
    PULPO_REQUIRE(dy_kb == 8 && in_kb == 8, "conv3d_k3_wgrad_kb: %dx%dx%d, %d -> %d channels does not run the F(2x2x2,3x3x3) kernel (pulpo_conv3d_k3_wgrad_algo != 3), the only reader of channel-blocked operands", D, H, W, Cin, Cout);
    if ((bnf || (Cin <= 4 && (dy_cs == 1) && (dy_ps % 4 == 0) && (dy_bs % 4 == 0) && (Cout % 4 == 0) && (((uintptr_t)dy & 15) == 0))) && algo != 1) {
        const int ntile4 = B * pulpo::cdiv(D, 4) * a.nty * a.ntx;
        used = std::min(std::max(1, 512 / a.ncot), ntile4);
        if (slabs) used = std::min(used, nslab);
    }
    if ((rc = prepare(used))) return rc;
    if (algo == 1) {
        // Winograd-x variant: wave = transformed point, nine (dz, dy) row tiles of <= 32 channels
        const int nrt9 = (9 * std::min(Cin, WG_CH) + 31) / 32;
#define PULPO_WGRAD_W(NTWV)                                                                                                       \
    {                                                                                                                             \
        static bool attr = false;                                                                                                 \
        if (!attr) {                                                                                                              \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_wino<NTWV>),                       \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma);                      \
            if (ea != hipSuccess) return pulpo::fail((int)ea, "hipFuncSetAttribute(wgrad wino): %s", hipGetErrorString(ea));      \
            attr = true;                                                                                                          \
        }                                                                                                                         \
        hipLaunchKernelGGL((conv3d_k3_wgrad_wino<NTWV>), dim3(nblk), dim3(256), lds_dma, st, a);                                  \
    }
        if (nrt9 <= 3) PULPO_WGRAD_W(3) else if (nrt9 <= 5) PULPO_WGRAD_W(5) else PULPO_WGRAD_W(9)
#undef PULPO_WGRAD_W
    } else if (bnf || (Cin <= 4 && (dy_cs == 1) && (dy_ps % 4 == 0) && (dy_bs % 4 == 0) && (Cout % 4 == 0) && (((uintptr_t)dy & 15) == 0))) {
        // the 2-/3-channel input layers: waves split K, every wave all row tiles (conv3d_k3_wgrad_smallc)
        WgradArgs b = a;
        b.ntz = pulpo::cdiv(D, 4);
        const int ntile4 = B * b.ntz * b.nty * b.ntx;
        b.ncit = 1;
        b.nsplit = std::min(std::max(1, 512 / b.ncot), ntile4);
        if (slabs) b.nsplit = std::min(b.nsplit, nslab);
        const int nrt = (27 * Cin + 31) / 32;
        constexpr size_t lds_s = (size_t)(4 * (6 * HY * HX + 4) + 256 * WG_NT) * sizeof(float);
        static_assert(4 * 32 * WG_NT <= 256 * WG_NT, "the partial sums of four row tiles fit the gradient image");
#define PULPO_WGRAD_S(NRTV)                                                                                                       \
    {                                                                                                                             \
        static bool attr = false;                                                                                                 \
        if (!attr) {                                                                                                              \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_smallc<NRTV>),                     \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);                        \
            if (ea != hipSuccess) return pulpo::fail((int)ea, "hipFuncSetAttribute(wgrad smallc): %s", hipGetErrorString(ea));    \
            attr = true;                                                                                                          \
        }                                                                                                                         \
        if (bnf) {                                                                                                                \
            static bool attrb = false;                                                                                            \
            if (!attrb) {                                                                                                         \
                hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_smallc<NRTV, true>),           \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s);                    \
                if (ea != hipSuccess) return pulpo::fail((int)ea, "hipFuncSetAttribute(wgrad smallc bn): %s", hipGetErrorString(ea)); \
                attrb = true;                                                                                                     \
            }                                                                                                                     \
            hipLaunchKernelGGL((conv3d_k3_wgrad_smallc<NRTV, true>), dim3(b.ncot * b.nsplit), dim3(256), lds_s, st, b);           \
        } else                                                                                                                    \
            hipLaunchKernelGGL((conv3d_k3_wgrad_smallc<NRTV>), dim3(b.ncot * b.nsplit), dim3(256), lds_s, st, b);                 \
    }
        if (nrt <= 2) PULPO_WGRAD_S(2) else if (nrt <= 3) PULPO_WGRAD_S(3) else PULPO_WGRAD_S(4)
#undef PULPO_WGRAD_S
    } else if (vec) {
        if (ntw <= 1) PULPO_WGRAD(true, 1) else if (ntw <= 2) PULPO_WGRAD(true, 2) else if (ntw <= 4) PULPO_WGRAD(true, 4) else PULPO_WGRAD(true, 7)
    } else {
        if (ntw <= 1) PULPO_WGRAD(false, 1) else if (ntw <= 2) PULPO_WGRAD(false, 2) else if (ntw <= 4) PULPO_WGRAD(false, 4) else PULPO_WGRAD(false, 7)
    }
#undef PULPO_WGRAD
    return finish(pulpo::check_launch("conv3d_k3_wgrad_mfma"));
}

PULPO_API int pulpo_conv3d_k3_wgrad(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dy, int64_t dy_bs,
                                    int64_t dy_ps, int64_t dy_cs, float* dw, int accumulate, float* scratch, int B, int D, int H, int W,
                                    int Cin, int Cout, void* stream) {
    return wgrad_impl(in, in_bs, in_ps, in_cs, dy, dy_bs, dy_ps, dy_cs, dw, accumulate, scratch, nullptr, 0, B, D, H, W, Cin, Cout, stream);
}

// ---- the input layer's weight gradient with its ConvUnit's BatchNorm / LeakyReLU backward fused into the operand staging (round 5):
// dw (+)= wgrad(in, dy) with dy = bn_lrelu_bwd_apply(dz, y, coef, totd) formed per element while the tile is staged; part2 [rows][Cout]
// (rows = pulpo_conv3d_k3_wgrad_bn_rows) receives the column sums of dy (the conv-bias gradient's partials).  Cin <= 4 only (the layers whose
// data gradient nobody needs: nothing else reads dy), dz fp32 or bf16 (dz_dt), y fp32, both channels-last with Cout % 4 == 0.
PULPO_API int pulpo_conv3d_k3_wgrad_bn_rows(int B, int D, int H, int W, int Cout) {
    const int ntile4 = B * pulpo::cdiv(D, 4) * pulpo::cdiv(H, TY) * pulpo::cdiv(W, TX);
    return std::min(std::max(1, 512 / pulpo::cdiv(Cout, WG_NT)), ntile4);
}

static int wgrad_bn_impl(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const void* dz, int dz_dt, int64_t dz_bs, int64_t dz_ps, int64_t dz_kb,
                         const float* y, int64_t y_bs, int64_t y_ps, const float* coef, const double* totd, float slope, float* dw, int accumulate,
                         float* scratch, float* part2, int B, int D, int H, int W, int Cin, int Cout, void* stream) {
    PULPO_REQUIRE(dz && y && coef && totd && part2, "conv3d_k3_wgrad_bn: null pointer");
    PULPO_REQUIRE(Cin >= 1 && Cin <= 4 && Cout % 4 == 0, "conv3d_k3_wgrad_bn: %d -> %d channels (input layers only: Cin <= 4, Cout %% 4 == 0)", Cin, Cout);
    PULPO_REQUIRE(dz_dt == 0 || dz_dt == 1, "conv3d_k3_wgrad_bn: dtype code %d", dz_dt);
    const int g = dz_dt ? 8 : 16;
    PULPO_REQUIRE(dz_ps % 4 == 0 && dz_bs % 4 == 0 && (((uintptr_t)dz) % g) == 0 && y_ps % 4 == 0 && y_bs % 4 == 0 && (((uintptr_t)y) & 15) == 0,
                  "conv3d_k3_wgrad_bn: dz and y must be channels-last with aligned four-channel pieces");
    WgradArgs f{};
    PULPO_REQUIRE(dz_kb >= 8 && dz_kb % 4 == 0 && (dz_kb == 8 || (dz_dt == 0 && Cout % 8 == 0)), "conv3d_k3_wgrad_bn: a channel-blocked dz is fp32 with whole 8-channel blocks");
    f.bn_y = y; f.bn_y_bs = y_bs; f.bn_y_ps = y_ps; f.bn_coef = coef; f.bn_totd = totd; f.bn_part2 = part2; f.slope = slope; f.dz_bf16 = dz_dt; f.dz_kb = dz_kb;
    return wgrad_impl(in, in_bs, in_ps, in_cs, (const float*)dz, dz_bs, dz_ps, 1, dw, accumulate, scratch, nullptr, 0, B, D, H, W, Cin, Cout, stream, &f);
}

PULPO_API int pulpo_conv3d_k3_wgrad_bn(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const void* dz, int dz_dt, int64_t dz_bs,
                                       int64_t dz_ps, const float* y, int64_t y_bs, int64_t y_ps, const float* coef, const double* totd, float slope,
                                       float* dw, int accumulate, float* scratch, float* part2, int B, int D, int H, int W, int Cin, int Cout,
                                       void* stream) {
    return wgrad_bn_impl(in, in_bs, in_ps, in_cs, dz, dz_dt, dz_bs, dz_ps, 8, y, y_bs, y_ps, coef, totd, slope, dw, accumulate, scratch, part2, B, D, H, W, Cin,
                         Cout, stream);
}

// (since ABI 5) dz in the channel-blocked layout: element (b, v, c) at dz + b * dz_bs + (c / 8) * dz_kb + v * dz_ps + c % 8 (fp32) - the gradient of a
// blocked activation (pulpo_bn_lrelu_apply_kb) as the next unit's data-gradient kernel writes it (pulpo_conv3d_k3_fwd_wino3_kb, out_kb)
PULPO_API int pulpo_conv3d_k3_wgrad_bn_kb(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dz, int64_t dz_bs, int64_t dz_ps,
                                          int64_t dz_kb, const float* y, int64_t y_bs, int64_t y_ps, const float* coef, const double* totd, float slope,
                                          float* dw, int accumulate, float* scratch, float* part2, int B, int D, int H, int W, int Cin, int Cout,
                                          void* stream) {
    return wgrad_bn_impl(in, in_bs, in_ps, in_cs, dz, 0, dz_bs, dz_ps, dz_kb, y, y_bs, y_ps, coef, totd, slope, dw, accumulate, scratch, part2, B, D, H, W, Cin,
                         Cout, stream);
}

// ---- deterministic form (PULPO_DETERMINISTIC): the same kernels, but workgroups that share a (ci tile, co tile) add their partial sums into
// separate zeroed copies ("slabs") of the packed scratch, and an ordered pass adds the copies up: results are bit-identical run to run (the plain
// form's float atomics add the workgroups' sums in arrival order).  slabs: nslab * pulpo_conv3d_k3_wgrad_scratch_floats(Cin, Cout) floats with
// nslab = pulpo_conv3d_k3_wgrad_det_slabs(Cin, Cout) (fewer are accepted: the spatial splits are capped at nslab).
PULPO_API int pulpo_conv3d_k3_wgrad_det_slabs(int Cin, int Cout) {
    const int npair = pulpo::cdiv(Cin, WG_CH) * pulpo::cdiv(Cout, WG_NT);
    return std::max(1, 512 / std::max(1, Cin <= 4 ? pulpo::cdiv(Cout, WG_NT) : npair));
}

PULPO_API int pulpo_conv3d_k3_wgrad_det(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dy, int64_t dy_bs,
                                        int64_t dy_ps, int64_t dy_cs, float* dw, int accumulate, float* scratch, float* slabs, int nslab, int B,
                                        int D, int H, int W, int Cin, int Cout, void* stream) {
    PULPO_REQUIRE(slabs && nslab >= 1, "conv3d_k3_wgrad_det: slabs of nslab >= 1 copies of the packed scratch required");
    return wgrad_impl(in, in_bs, in_ps, in_cs, dy, dy_bs, dy_ps, dy_cs, dw, accumulate, scratch, slabs, nslab, B, D, H, W, Cin, Cout, stream);
}

// The gradient in the channel-BLOCKED layout of pulpo_bn_lrelu_bwd_apply_kb_t: element (b, voxel v, channel c) at dy + b * dy_bs + (c / 8) * dy_kb + v * dy_ps + c % 8
// (dy_ps = 8, dy_bs = V * 8, dy_kb = B * V * 8).  Shapes with pulpo_conv3d_k3_wgrad_algo(...) == 3 only; slabs == NULL: the atomic form, else the
// deterministic one (as pulpo_conv3d_k3_wgrad_det).
PULPO_API int pulpo_conv3d_k3_wgrad_kb(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_kb, const float* dy, int64_t dy_bs, int64_t dy_ps, int64_t dy_kb,
                                       float* dw, int accumulate, float* scratch, float* slabs, int nslab, int B, int D, int H, int W, int Cin, int Cout,
                                       void* stream) {
    PULPO_REQUIRE(dy_kb >= 8 && dy_kb % 4 == 0 && in_kb >= 8 && in_kb % 4 == 0 && (slabs == nullptr || nslab >= 1), "conv3d_k3_wgrad_kb: bad block stride / slab count");
    PULPO_REQUIRE((in_kb == 8 || Cin % 8 == 0) && (dy_kb == 8 || Cout % 8 == 0), "conv3d_k3_wgrad_kb: a channel-blocked operand has whole 8-channel blocks");
    return wgrad_impl(in, in_bs, in_ps, 1, dy, dy_bs, dy_ps, 1, dw, accumulate, scratch, slabs, slabs ? nslab : 0, B, D, H, W, Cin, Cout, stream, nullptr, dy_kb, in_kb);
}

namespace {
// (rows of npad floats of which the first `cols` are ever written: the padding columns of a 32-channel layer - half of every slab - are skipped)
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(float* __restrict__ scratch, const float* __restrict__ slabs, int nslab, long n, int npad,
                                                                  int cols) {
    const int q4 = (cols + 3) >> 2;                       // four-float groups per row
    const long rows = n / npad, items = rows * q4;
    for (long it = blockIdx.x * (long)blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
        const long e = (it / q4) * npad + (it % q4) * 4;
        float4 t = *reinterpret_cast<const float4*>(scratch + e);
        for (int s_ = 0; s_ < nslab; ++s_) {
            const float4 u = *reinterpret_cast<const float4*>(slabs + (long)s_ * n + e);
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<float4*>(scratch + e) = t;
    }
}
}  // namespace

int pulpo_conv::launch_wgrad_slab_reduce(float* scratch, const float* slabs, int nslab, long n, hipStream_t st, int npad_, int cols) {
    // (n = 27 * Cin * npad(Cout) is a multiple of 64 floats; scratch and slabs come from the caller's allocator: 16-byte aligned)
    if ((n & 3) || (((uintptr_t)scratch | (uintptr_t)slabs) & 15)) return pulpo::fail(1, "wgrad slab reduce: operands must be 16-byte aligned");
    if (npad_ <= 0 || n % npad_ != 0) { npad_ = 64; cols = 64; }
    const long items = (n / npad_) * ((cols + 3) / 4);
    const int nb = (int)std::min<long>((items + 255) / 256, 4096);
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(nb), dim3(256), 0, st, scratch, slabs, nslab, n, npad_, cols);
    return pulpo::check_launch("wgrad_slab_reduce");
}
