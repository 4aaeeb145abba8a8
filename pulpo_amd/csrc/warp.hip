// Deformable warp (SpatialTransformer) and scaling-and-squaring integration (VecInt).
// Reference: src/network_blocks.py:88-121 (grid + df -> normalise by (S-1) -> grid_sample(bilinear, border,
// align_corners=False)) and :160-177 (v/2^n, then n times v <- v + warp(v, v)).
// The coordinate arithmetic repeats the reference's operation order (divide by S-1, -0.5, *2, then ATen's
// ((n+1)*S-1)/2 un-normalisation and clamp) so results agree to rounding, including the reference's quirk that a
// zero field is not the identity.
// HBM/L2-bound gather kernels: one thread per output voxel, displacement planes read coalesced, 8-corner gather
// served by L1/L2; the backward scatters with float atomics (memory-side adds, MI355X_MICROARCH.md).
#include "common.h"
#include <stdlib.h>

namespace {

struct Corner {
    int i0, i1;
    float f;       // fraction towards i1
    float dscale;  // d(coord)/d(displacement); 0 where the coordinate was clamped
};

__device__ __forceinline__ Corner sample_coord(float pos, float disp, int Sg, int Si) {
    if (Sg == 1) {            // a depth-1 grid is the reference's 2-D case (bilinear grid_sample over H, W): no coordinate along this axis
        Corner r;
        r.i0 = 0; r.i1 = 0; r.f = 0.f; r.dscale = 0.f;
        return r;
    }
    float t = pos + disp;
    t = t / (float)(Sg - 1);
    t = t - 0.5f;
    t = 2.f * t;
    float c = ((t + 1.f) * (float)Si - 1.f) / 2.f;
    Corner r;
    const float hi = (float)(Si - 1);
    r.dscale = (c > 0.f && c < hi) ? (float)Si / (float)(Sg - 1) : 0.f;   // ATen clip_coordinates_set_grad
    c = fminf(hi, fmaxf(c, 0.f));
    const float fl = floorf(c);
    r.i0 = (int)fl;
    r.i1 = min(r.i0 + 1, Si - 1);
    r.f = c - fl;
    return r;
}

// out[b][c][v] = trilinear(img[b][c], grid position v displaced by df[b][:, v]);  optional residual: out += add[b][c][v]
template <int C>
__global__ __launch_bounds__(256) void warp_fwd_kernel(const float* __restrict__ df, const float* __restrict__ img, const float* __restrict__ add,
                                                         float* __restrict__ out, int B, int Dg, int Hg, int Wg, int Di, int Hi, int Wi, int Cr) {
    const long Vg = (long)Dg * Hg * Wg, Vi = (long)Di * Hi * Wi;
    const long total = (long)B * Vg;
    const int nch = C > 0 ? C : Cr;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long b = B == 1 ? 0 : e / Vg, v = e - b * Vg;          // (one 64-bit division per voxel at most; the voxel's coordinates in 32 bits -
        const int vi = (int)v;                                        //  three 64-bit divisions per voxel were a third of these kernels' time)
        const int x = vi % Wg, y = (vi / Wg) % Hg, z = vi / (Wg * Hg);
        const float* d = df + b * 3 * Vg + v;
        const Corner cz = sample_coord((float)z, d[0], Dg, Di);
        const Corner cy = sample_coord((float)y, d[Vg], Hg, Hi);
        const Corner cx = sample_coord((float)x, d[2 * Vg], Wg, Wi);
        const long o00 = ((long)cz.i0 * Hi + cy.i0) * Wi, o01 = ((long)cz.i0 * Hi + cy.i1) * Wi;
        const long o10 = ((long)cz.i1 * Hi + cy.i0) * Wi, o11 = ((long)cz.i1 * Hi + cy.i1) * Wi;
        const float wz0 = 1.f - cz.f, wy0 = 1.f - cy.f, wx0 = 1.f - cx.f;
        for (int c = 0; c < nch; ++c) {
            const float* s = img + (b * nch + c) * Vi;
            float val = wz0 * wy0 * wx0 * s[o00 + cx.i0] + wz0 * wy0 * cx.f * s[o00 + cx.i1] + wz0 * cy.f * wx0 * s[o01 + cx.i0] +
                        wz0 * cy.f * cx.f * s[o01 + cx.i1] + cz.f * wy0 * wx0 * s[o10 + cx.i0] + cz.f * wy0 * cx.f * s[o10 + cx.i1] +
                        cz.f * cy.f * wx0 * s[o11 + cx.i0] + cz.f * cy.f * cx.f * s[o11 + cx.i1];
            const long oi = (b * nch + c) * Vg + v;
            if (add != nullptr) val += add[oi];
            out[oi] = val;
        }
    }
}

// gdf (nullable) and gimg (nullable) are ACCUMULATED with float atomics (caller zero-fills or pre-loads them).
// gdf_direct: write gdf with plain stores instead (only legal when nothing else adds to gdf concurrently).
template <bool GDF_ATOMIC>
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ df, const float* __restrict__ img, const float* __restrict__ gout,
                                                         float* __restrict__ gdf, float* __restrict__ gimg, int B, int Dg, int Hg, int Wg, int Di,
                                                         int Hi, int Wi, int nch) {
    const long Vg = (long)Dg * Hg * Wg, Vi = (long)Di * Hi * Wi;
    const long total = (long)B * Vg;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long b = B == 1 ? 0 : e / Vg, v = e - b * Vg;          // (one 64-bit division per voxel at most; the voxel's coordinates in 32 bits -
        const int vi = (int)v;                                        //  three 64-bit divisions per voxel were a third of these kernels' time)
        const int x = vi % Wg, y = (vi / Wg) % Hg, z = vi / (Wg * Hg);
        const float* d = df + b * 3 * Vg + v;
        const Corner cz = sample_coord((float)z, d[0], Dg, Di);
        const Corner cy = sample_coord((float)y, d[Vg], Hg, Hi);
        const Corner cx = sample_coord((float)x, d[2 * Vg], Wg, Wi);
        const long o00 = ((long)cz.i0 * Hi + cy.i0) * Wi, o01 = ((long)cz.i0 * Hi + cy.i1) * Wi;
        const long o10 = ((long)cz.i1 * Hi + cy.i0) * Wi, o11 = ((long)cz.i1 * Hi + cy.i1) * Wi;
        const float wz0 = 1.f - cz.f, wy0 = 1.f - cy.f, wx0 = 1.f - cx.f;
        float gz = 0.f, gy = 0.f, gx = 0.f;
        for (int c = 0; c < nch; ++c) {
            const float g = gout[(b * nch + c) * Vg + v];
            const float* s = img + (b * nch + c) * Vi;
            const float s000 = s[o00 + cx.i0], s001 = s[o00 + cx.i1], s010 = s[o01 + cx.i0], s011 = s[o01 + cx.i1];
            const float s100 = s[o10 + cx.i0], s101 = s[o10 + cx.i1], s110 = s[o11 + cx.i0], s111 = s[o11 + cx.i1];
            // d/dfz, d/dfy, d/dfx of the trilinear interpolant (same expansion as ATen's grid_sampler_3d_backward)
            gz += g * (wy0 * wx0 * (s100 - s000) + wy0 * cx.f * (s101 - s001) + cy.f * wx0 * (s110 - s010) + cy.f * cx.f * (s111 - s011));
            gy += g * (wz0 * wx0 * (s010 - s000) + wz0 * cx.f * (s011 - s001) + cz.f * wx0 * (s110 - s100) + cz.f * cx.f * (s111 - s101));
            gx += g * (wz0 * wy0 * (s001 - s000) + wz0 * cy.f * (s011 - s010) + cz.f * wy0 * (s101 - s100) + cz.f * cy.f * (s111 - s110));
            if (gimg != nullptr) {
                float* q = gimg + (b * nch + c) * Vi;
                atomicAdd(q + o00 + cx.i0, g * wz0 * wy0 * wx0);
                atomicAdd(q + o00 + cx.i1, g * wz0 * wy0 * cx.f);
                atomicAdd(q + o01 + cx.i0, g * wz0 * cy.f * wx0);
                atomicAdd(q + o01 + cx.i1, g * wz0 * cy.f * cx.f);
                atomicAdd(q + o10 + cx.i0, g * cz.f * wy0 * wx0);
                atomicAdd(q + o10 + cx.i1, g * cz.f * wy0 * cx.f);
                atomicAdd(q + o11 + cx.i0, g * cz.f * cy.f * wx0);
                atomicAdd(q + o11 + cx.i1, g * cz.f * cy.f * cx.f);
            }
        }
        if (gdf != nullptr) {
            float* q = gdf + b * 3 * Vg + v;
            if constexpr (GDF_ATOMIC) {
                atomicAdd(q, gz * cz.dscale);
                atomicAdd(q + Vg, gy * cy.dscale);
                atomicAdd(q + 2 * Vg, gx * cx.dscale);
            } else {
                q[0] = gz * cz.dscale;
                q[Vg] = gy * cy.dscale;
                q[2 * Vg] = gx * cx.dscale;
            }
        }
    }
}

// One backward squaring step of VecInt (v' = v + warp(v, v): the field is image and displacement at once) with the scatter collected in
// LDS: a workgroup owns a 4 x 8 x 8 voxel tile and keeps a (4+2R) x (8+2R) x (8+2R) x 3 accumulation box around it; a step's displacement
// is a fraction of the final field (v / 2^(nsteps-k)), so almost every corner a voxel scatters to lies inside the box of its own tile -
// those adds are LDS atomics, the rest go to memory as before.  The box then leaves with one memory atomic per touched cell: 4-7 memory
// atomics per voxel instead of 27 (24 corner adds + 3 displacement-gradient adds), which is what bounded warp_bwd_kernel here.  Round 3:
// the LDS atomics themselves (about one lane per two clocks) had become the bound - neighbouring lanes now merge their contributions to
// shared cells before the add (along x, then along y: see below), 607 -> 335 us per seven steps at 80^3.
// gp must be all zero on entry: the identity path (g itself) is added to the voxel's own cell of the box and leaves with it (no copy of g
// into gp in front of every step; pulpo_vecint_bwd zeroes the buffers of all steps with one fill).
template <int R>
__global__ __launch_bounds__(256) void vecint_bwd_tile_kernel(const float* __restrict__ cur, const float* __restrict__ gout, float* __restrict__ gp,
                                                                int B, int D, int H, int W, int ntz, int nty, int ntx) {
    constexpr int TZ_ = 4, TY_ = 8, TX_ = 8;
    constexpr int BZ = TZ_ + 2 * R, BY = TY_ + 2 * R, BX = TX_ + 2 * R, BV = BZ * BY * BX;
    __shared__ float box[3 * BV];
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tx_ = t % ntx; t /= ntx;
    const int ty_ = t % nty; t /= nty;
    const int tz_ = t % ntz;
    const int b = t / ntz;
    const int z0 = tz_ * TZ_, y0 = ty_ * TY_, x0 = tx_ * TX_;
    const long V = (long)D * H * W;
    for (int j = tid; j < 3 * BV; j += 256) box[j] = 0.f;
    __syncthreads();
    const int z = z0 + (tid >> 6), y = y0 + ((tid >> 3) & 7), x = x0 + (tid & 7);
    // (straight-line code with a `valid` predicate instead of a branch around the voxel's work: the lanes of a wave - one z-plane of the
    //  tile, 8 rows of 8 voxels - exchange contributions below)
    const bool valid = z < D && y < H && x < W;
    const long v = valid ? ((long)z * H + y) * W + x : 0;
    const float* d = cur + (long)b * 3 * V + v;
    const Corner cz = sample_coord((float)z, valid ? d[0] : 0.f, D, D);
    const Corner cy = sample_coord((float)y, valid ? d[V] : 0.f, H, H);
    const Corner cx = sample_coord((float)x, valid ? d[2 * V] : 0.f, W, W);
    const float wz0 = 1.f - cz.f, wy0 = 1.f - cy.f, wx0 = 1.f - cx.f;
    const long o00 = ((long)cz.i0 * H + cy.i0) * W, o01 = ((long)cz.i0 * H + cy.i1) * W;
    const long o10 = ((long)cz.i1 * H + cy.i0) * W, o11 = ((long)cz.i1 * H + cy.i1) * W;
    // box coordinates of the corners (negative / too large = outside the box)
    const int lz0 = cz.i0 - z0 + R, lz1 = cz.i1 - z0 + R, ly0 = cy.i0 - y0 + R, ly1 = cy.i1 - y0 + R, lx0 = cx.i0 - x0 + R, lx1 = cx.i1 - x0 + R;
    const bool inbox = valid && lz0 >= 0 && lz1 < BZ && ly0 >= 0 && ly1 < BY && lx0 >= 0 && lx1 < BX;
    // Neighbouring lanes of a row mostly scatter to overlapping cells: for a smooth field the "x1" corners of voxel x are the "x0" corners of
    // voxel x + 1.  Where that holds (same z / y cells, both in the box) the right neighbour TAKES the left one's four x1 contributions per
    // channel into its own x0 adds, and the left one skips them: 12 + 3 instead of 24 + 3 LDS atomics per voxel - which retire about one lane
    // per two clocks and are what bounds this kernel (scripts/vecint_probe.py).  Sums are the same up to the order of the additions.
    const int key = inbox ? (((lz0 * 32 + lz1) * 32 + ly0) * 32 + ly1) : -1;
    const int key_l = __shfl_up(key, 1, 64), lx1_l = __shfl_up(lx1, 1, 64);
    const bool take = inbox && (tid & 7) != 0 && key_l == key && lx1_l == lx0;
    const int take_r = __shfl_down((int)take, 1, 64);          // (every lane takes part in the exchange: no short-circuit in front of it)
    const bool give = (tid & 7) != 7 && take_r != 0;
    // ... and the same along y (lanes 8 apart), separately for the x0 corners and for the x1 corners a lane still holds: the lane of row
    // y + 1 takes the dy = 1 contributions of row y into its dy = 0 adds.  An interior voxel of a smooth field is left with 2 of its 8
    // corner adds per channel.
    const int row = (tid >> 3) & 7;
    const int ky0 = inbox ? ((lz0 * 32 + lz1) * 32 + lx0) : -1, ky1 = (inbox && !give) ? ((lz0 * 32 + lz1) * 32 + lx1) : -1;
    const int ky0_u = __shfl_up(ky0, 8, 64), ky1_u = __shfl_up(ky1, 8, 64), ly1_u = __shfl_up(ly1, 8, 64);
    const bool take_y0 = row != 0 && ky0 >= 0 && ky0_u == ky0 && ly1_u == ly0;
    const bool take_y1 = row != 0 && ky1 >= 0 && ky1_u == ky1 && ly1_u == ly0;
    const int ty0_d = __shfl_down((int)take_y0, 8, 64), ty1_d = __shfl_down((int)take_y1, 8, 64);
    const bool give_y0 = row != 7 && ty0_d != 0, give_y1 = row != 7 && ty1_d != 0;
    float gz = 0.f, gy = 0.f, gx = 0.f;
    float gid[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = valid ? gout[((long)b * 3 + c) * V + v] : 0.f;
        gid[c] = g;
        if (valid) {
            const float* s = cur + ((long)b * 3 + c) * V;
            const float s000 = s[o00 + cx.i0], s001 = s[o00 + cx.i1], s010 = s[o01 + cx.i0], s011 = s[o01 + cx.i1];
            const float s100 = s[o10 + cx.i0], s101 = s[o10 + cx.i1], s110 = s[o11 + cx.i0], s111 = s[o11 + cx.i1];
            gz += g * (wy0 * wx0 * (s100 - s000) + wy0 * cx.f * (s101 - s001) + cy.f * wx0 * (s110 - s010) + cy.f * cx.f * (s111 - s011));
            gy += g * (wz0 * wx0 * (s010 - s000) + wz0 * cx.f * (s011 - s001) + cz.f * wx0 * (s110 - s100) + cz.f * cx.f * (s111 - s101));
            gx += g * (wz0 * wy0 * (s001 - s000) + wz0 * cy.f * (s011 - s010) + cz.f * wy0 * (s101 - s100) + cz.f * cy.f * (s111 - s110));
        }
        float a000 = g * wz0 * wy0 * wx0, a010 = g * wz0 * cy.f * wx0, a100 = g * cz.f * wy0 * wx0, a110 = g * cz.f * cy.f * wx0;       // x0 corners
        const float a001 = g * wz0 * wy0 * cx.f, a011 = g * wz0 * cy.f * cx.f, a101 = g * cz.f * wy0 * cx.f, a111 = g * cz.f * cy.f * cx.f;   // x1 corners
        const float t00 = __shfl_up(a001, 1, 64), t01 = __shfl_up(a011, 1, 64), t10 = __shfl_up(a101, 1, 64), t11 = __shfl_up(a111, 1, 64);
        if (take) { a000 += t00; a010 += t01; a100 += t10; a110 += t11; }
        float b001 = a001, b101 = a101;
        const float u010 = __shfl_up(a010, 8, 64), u110 = __shfl_up(a110, 8, 64), u011 = __shfl_up(a011, 8, 64), u111 = __shfl_up(a111, 8, 64);
        if (take_y0) { a000 += u010; a100 += u110; }
        if (take_y1) { b001 += u011; b101 += u111; }
        if (inbox) {
            float* q = box + c * BV;
            const int p00 = (lz0 * BY + ly0) * BX, p01 = (lz0 * BY + ly1) * BX, p10 = (lz1 * BY + ly0) * BX, p11 = (lz1 * BY + ly1) * BX;
            atomicAdd(q + p00 + lx0, a000);
            atomicAdd(q + p10 + lx0, a100);
            if (!give_y0) {
                atomicAdd(q + p01 + lx0, a010);
                atomicAdd(q + p11 + lx0, a110);
            }
            if (!give) {
                atomicAdd(q + p00 + lx1, b001);
                atomicAdd(q + p10 + lx1, b101);
                if (!give_y1) {
                    atomicAdd(q + p01 + lx1, a011);
                    atomicAdd(q + p11 + lx1, a111);
                }
            }
        } else if (valid) {
            float* q = gp + ((long)b * 3 + c) * V;
            atomicAdd(q + o00 + cx.i0, a000);
            atomicAdd(q + o00 + cx.i1, a001);
            atomicAdd(q + o01 + cx.i0, a010);
            atomicAdd(q + o01 + cx.i1, a011);
            atomicAdd(q + o10 + cx.i0, a100);
            atomicAdd(q + o10 + cx.i1, a101);
            atomicAdd(q + o11 + cx.i0, a110);
            atomicAdd(q + o11 + cx.i1, a111);
        }
    }
    if (valid) {
        // the displacement-gradient term and the identity path land on the voxel itself: into its (always in-box) cell
        const int own = ((z - z0 + R) * BY + (y - y0 + R)) * BX + (x - x0 + R);
        atomicAdd(box + own, gid[0] + gz * cz.dscale);
        atomicAdd(box + BV + own, gid[1] + gy * cy.dscale);
        atomicAdd(box + 2 * BV + own, gid[2] + gx * cx.dscale);
    }
    __syncthreads();
    for (int j = tid; j < 3 * BV; j += 256) {
        const float val = box[j];
        if (val == 0.f) continue;
        const int c = j / BV, r = j - c * BV;
        const int gz_ = z0 - R + r / (BY * BX), gy_ = y0 - R + (r / BX) % BY, gx_ = x0 - R + r % BX;
        if (gz_ >= 0 && gz_ < D && gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W)
            atomicAdd(gp + ((long)b * 3 + c) * V + ((long)gz_ * H + gy_) * W + gx_, val);
    }
}

__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ in, float* __restrict__ out, float s, long n) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) out[e] = in[e] * s;
}

// VecInt forward for small fields (3 * D*H*W floats <= 96 KB fit; used up to 2048 voxels - the 10^3 pyramid level - see pulpo_vecint_fwd): ALL squaring steps in ONE launch, one
// workgroup of 1024 threads per batch element.  The field lives in LDS (image and displacement at once: every gather is an LDS read); a
// step's new values are formed in registers (up to 8 voxels per thread), then written back over the field and to work[k + 1] (the backward
// pass needs every intermediate field).  Same arithmetic as scale_kernel + nsteps x warp_fwd_kernel<3> with add = cur (agreement to fp32
// rounding: 1e-6 relative after seven squarings).
constexpr int VI_THREADS = 1024, VI_MAXV = 8 * VI_THREADS;
__global__ __launch_bounds__(VI_THREADS) void vecint_fwd_lds_kernel(const float* __restrict__ v, float* __restrict__ work, int B, int D, int H, int W,
                                                                      int nsteps, float scale) {
    extern __shared__ float fld[];                     // [3][V]
    const int V = D * H * W, tid = threadIdx.x;
    const long b = blockIdx.x, n = (long)B * 3 * V;
    for (int i = tid; i < 3 * V; i += VI_THREADS) {
        const float val = v[b * 3 * V + i] * scale;
        fld[i] = val;
        work[b * 3 * V + i] = val;
    }
    __syncthreads();
    for (int k = 0; k < nsteps; ++k) {
        float nv[8][3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int vox = tid + j * VI_THREADS;
            if (vox < V) {
                const int x = vox % W, y = (vox / W) % H, z = vox / (W * H);
                const Corner cz = sample_coord((float)z, fld[vox], D, D);
                const Corner cy = sample_coord((float)y, fld[V + vox], H, H);
                const Corner cx = sample_coord((float)x, fld[2 * V + vox], W, W);
                const int o00 = (cz.i0 * H + cy.i0) * W, o01 = (cz.i0 * H + cy.i1) * W;
                const int o10 = (cz.i1 * H + cy.i0) * W, o11 = (cz.i1 * H + cy.i1) * W;
                const float wz0 = 1.f - cz.f, wy0 = 1.f - cy.f, wx0 = 1.f - cx.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* s = fld + c * V;
                    float val = wz0 * wy0 * wx0 * s[o00 + cx.i0] + wz0 * wy0 * cx.f * s[o00 + cx.i1] + wz0 * cy.f * wx0 * s[o01 + cx.i0] +
                                wz0 * cy.f * cx.f * s[o01 + cx.i1] + cz.f * wy0 * wx0 * s[o10 + cx.i0] + cz.f * wy0 * cx.f * s[o10 + cx.i1] +
                                cz.f * cy.f * wx0 * s[o11 + cx.i0] + cz.f * cy.f * cx.f * s[o11 + cx.i1];
                    val += s[vox];
                    nv[j][c] = val;
                }
            }
        }
        __syncthreads();                               // every gather of this step is done: the field may be overwritten
        float* out = work + (long)(k + 1) * n + b * 3 * V;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int vox = tid + j * VI_THREADS;
            if (vox < V) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    fld[c * V + vox] = nv[j][c];
                    out[c * V + vox] = nv[j][c];
                }
            }
        }
        __syncthreads();
    }
}

inline int eblocks(long items) { return (int)std::max<long>(1, std::min<long>((items + 255) / 256, 8192)); }

// ------------------------------------------------------------------------------------------------ deterministic backward (PULPO_DETERMINISTIC)
// The scatter of the image gradient (grid_sampler_3d_backward's atomics; network_blocks.py:120, :175) is what makes these backward passes differ
// in the last bits from run to run: float atomics add in arrival order.  Here every contribution enters a 64-bit FIXED-POINT accumulator
// - integer adds commute, so the sum does not depend on the order - as llrint(v * 2^S) with S taken from the largest |gradient| of the pass
// (2^46 units per that maximum: 22 bits below fp32's last place of the largest value, 17 bits of head-room for the sums); a second pass turns the
// accumulators back into floats (and returns them to zero).  Same contributions as the plain kernels, summed exactly instead of in fp32.
__device__ __forceinline__ double fx_unit(const unsigned* maxbits) {          // 2^S for the pass whose largest |gradient| has these bits
    const float m = __uint_as_float(*maxbits);
    if (!(m > 0.f) || !(m < INFINITY)) return 1.0;                           // all zero (or not finite: the result is not finite either way)
    int ex;
    frexpf(m, &ex);                                                           // m = f * 2^ex, f in [0.5, 1)
    return ldexp(1.0, 46 - ex);
}
__device__ __forceinline__ void fx_add(long long* p, float v, double unit) {
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double2ll_rn((double)v * unit));
}

// largest value of a workgroup -> ONE atomic per workgroup (thousands of waves hitting one address serialise at the L2: the first version, one
// atomic per wave, spent 70 us per launch on a 1.5 M-element field); bit patterns of non-negative floats order like the values
__device__ __forceinline__ void block_absmax_to(float m, unsigned* out) {
    __shared__ float wm[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (t > 0.f) atomicMax(out, __float_as_uint(t));
    }
}
inline int fx_blocks(long items) { return (int)std::max<long>(1, std::min<long>((items + 255) / 256, 1024)); }

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ g, long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(g[e]));
    block_absmax_to(m, out);
}

// SELF = a VecInt squaring step (v' = v + warp(v, v): image and displacement are the same 3-channel field; the identity path and the
// displacement gradient join the voxel's own accumulator).  !SELF = SpatialTransformer: acc collects the image gradient, gdf is written plainly.
template <bool SELF>
__global__ __launch_bounds__(256) void warp_bwd_fx_kernel(const float* __restrict__ df, const float* __restrict__ img, const float* __restrict__ gout,
                                                            float* __restrict__ gdf, long long* __restrict__ acc, const unsigned* __restrict__ maxbits,
                                                            int B, int Dg, int Hg, int Wg, int Di, int Hi, int Wi, int nch) {
    const long Vg = (long)Dg * Hg * Wg, Vi = (long)Di * Hi * Wi;
    const long total = (long)B * Vg;
    const double unit = fx_unit(maxbits);
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long b = B == 1 ? 0 : e / Vg, v = e - b * Vg;          // (one 64-bit division per voxel at most; the voxel's coordinates in 32 bits -
        const int vi = (int)v;                                        //  three 64-bit divisions per voxel were a third of these kernels' time)
        const int x = vi % Wg, y = (vi / Wg) % Hg, z = vi / (Wg * Hg);
        const float* d = df + b * 3 * Vg + v;
        const Corner cz = sample_coord((float)z, d[0], Dg, Di);
        const Corner cy = sample_coord((float)y, d[Vg], Hg, Hi);
        const Corner cx = sample_coord((float)x, d[2 * Vg], Wg, Wi);
        const long o00 = ((long)cz.i0 * Hi + cy.i0) * Wi, o01 = ((long)cz.i0 * Hi + cy.i1) * Wi;
        const long o10 = ((long)cz.i1 * Hi + cy.i0) * Wi, o11 = ((long)cz.i1 * Hi + cy.i1) * Wi;
        const float wz0 = 1.f - cz.f, wy0 = 1.f - cy.f, wx0 = 1.f - cx.f;
        float gz = 0.f, gy = 0.f, gx = 0.f;
        float gid[3] = {0.f, 0.f, 0.f};
        for (int c = 0; c < nch; ++c) {
            const float g = gout[(b * nch + c) * Vg + v];
            if (SELF && c < 3) gid[c] = g;
            const float* s = img + (b * nch + c) * Vi;
            const float s000 = s[o00 + cx.i0], s001 = s[o00 + cx.i1], s010 = s[o01 + cx.i0], s011 = s[o01 + cx.i1];
            const float s100 = s[o10 + cx.i0], s101 = s[o10 + cx.i1], s110 = s[o11 + cx.i0], s111 = s[o11 + cx.i1];
            gz += g * (wy0 * wx0 * (s100 - s000) + wy0 * cx.f * (s101 - s001) + cy.f * wx0 * (s110 - s010) + cy.f * cx.f * (s111 - s011));
            gy += g * (wz0 * wx0 * (s010 - s000) + wz0 * cx.f * (s011 - s001) + cz.f * wx0 * (s110 - s100) + cz.f * cx.f * (s111 - s101));
            gx += g * (wz0 * wy0 * (s001 - s000) + wz0 * cy.f * (s011 - s010) + cz.f * wy0 * (s101 - s100) + cz.f * cy.f * (s111 - s110));
            if (acc != nullptr) {
                long long* q = acc + (b * nch + c) * Vi;
                fx_add(q + o00 + cx.i0, g * wz0 * wy0 * wx0, unit);
                fx_add(q + o00 + cx.i1, g * wz0 * wy0 * cx.f, unit);
                fx_add(q + o01 + cx.i0, g * wz0 * cy.f * wx0, unit);
                fx_add(q + o01 + cx.i1, g * wz0 * cy.f * cx.f, unit);
                fx_add(q + o10 + cx.i0, g * cz.f * wy0 * wx0, unit);
                fx_add(q + o10 + cx.i1, g * cz.f * wy0 * cx.f, unit);
                fx_add(q + o11 + cx.i0, g * cz.f * cy.f * wx0, unit);
                fx_add(q + o11 + cx.i1, g * cz.f * cy.f * cx.f, unit);
            }
        }
        if (SELF) {
            long long* q = acc + b * 3 * Vg + v;
            fx_add(q, gid[0] + gz * cz.dscale, unit);
            fx_add(q + Vg, gid[1] + gy * cy.dscale, unit);
            fx_add(q + 2 * Vg, gid[2] + gx * cx.dscale, unit);
        } else if (gdf != nullptr) {
            float* q = gdf + b * 3 * Vg + v;
            q[0] = gz * cz.dscale;
            q[Vg] = gy * cy.dscale;
            q[2 * Vg] = gx * cx.dscale;
        }
    }
}

// out = float(acc / 2^S) * post, acc <- 0; next_max (nullable) receives the largest |out| (the next pass's scale)
__global__ __launch_bounds__(256) void fx_to_float_kernel(long long* __restrict__ acc, float* __restrict__ out, long n, const unsigned* __restrict__ maxbits,
                                                            float post, unsigned* __restrict__ next_max) {
    const double inv = 1.0 / fx_unit(maxbits);
    float m = 0.f;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float val = (float)((double)acc[e] * inv) * post;
        acc[e] = 0;
        out[e] = val;
        m = fmaxf(m, fabsf(val));
    }
    if (next_max != nullptr) block_absmax_to(m, next_max);          // (uniform branch: every thread of the workgroup takes it)
}

}  // namespace

// df: (B,3,Dg,Hg,Wg) planar; img: (B,C,Di,Hi,Wi) planar; out: (B,C,Dg,Hg,Wg)
PULPO_API int pulpo_warp3d_fwd(const float* df, const float* img, float* out, int B, int C, int Dg, int Hg, int Wg, int Di, int Hi, int Wi,
                               void* stream) {
    PULPO_REQUIRE(df && img && out && B > 0 && C > 0, "warp3d_fwd: bad arguments");
    PULPO_REQUIRE(Dg >= 1 && Hg > 1 && Wg > 1 && Di > 0 && Hi > 0 && Wi > 0 && (Dg > 1 || Di == 1), "warp3d_fwd: grid H, W must be > 1 (depth 1 = 2-D form, with a depth-1 image)");
    PULPO_REQUIRE((long)Dg * Hg * Wg < (1L << 31), "warp3d_fwd: grids of 2^31 voxels and more are not supported");
    const long total = (long)B * Dg * Hg * Wg;
    hipStream_t st = (hipStream_t)stream;
    if (C == 1) hipLaunchKernelGGL(warp_fwd_kernel<1>, dim3(eblocks(total)), dim3(256), 0, st, df, img, nullptr, out, B, Dg, Hg, Wg, Di, Hi, Wi, C);
    else if (C == 3) hipLaunchKernelGGL(warp_fwd_kernel<3>, dim3(eblocks(total)), dim3(256), 0, st, df, img, nullptr, out, B, Dg, Hg, Wg, Di, Hi, Wi, C);
    else hipLaunchKernelGGL(warp_fwd_kernel<0>, dim3(eblocks(total)), dim3(256), 0, st, df, img, nullptr, out, B, Dg, Hg, Wg, Di, Hi, Wi, C);
    return pulpo::check_launch("warp3d_fwd");
}

// gdf: (B,3,grid) written; gimg: (B,C,img) zero-filled here then accumulated.  Either may be null.
PULPO_API int pulpo_warp3d_bwd(const float* df, const float* img, const float* gout, float* gdf, float* gimg, int B, int C, int Dg, int Hg, int Wg,
                               int Di, int Hi, int Wi, void* stream) {
    PULPO_REQUIRE(df && img && gout && B > 0 && C > 0, "warp3d_bwd: bad arguments");
    PULPO_REQUIRE(Dg >= 1 && Hg > 1 && Wg > 1 && (long)Dg * Hg * Wg < (1L << 31), "warp3d_bwd: grid H, W must be > 1 (and fewer than 2^31 voxels)");
    hipStream_t st = (hipStream_t)stream;
    if (gimg != nullptr) {
        hipError_t e = hipMemsetAsync(gimg, 0, sizeof(float) * (size_t)B * C * Di * Hi * Wi, st);
        if (e != hipSuccess) return pulpo::fail((int)e, "warp3d_bwd memset: %s", hipGetErrorString(e));
    }
    const long total = (long)B * Dg * Hg * Wg;
    hipLaunchKernelGGL(warp_bwd_kernel<false>, dim3(eblocks(total)), dim3(256), 0, st, df, img, gout, gdf, gimg, B, Dg, Hg, Wg, Di, Hi, Wi, C);
    return pulpo::check_launch("warp3d_bwd");
}

// work: (nsteps+1) buffers of B*3*D*H*W floats; work[k] is the field after k squarings, work[nsteps] the result.
PULPO_API int pulpo_vecint_fwd(const float* v, float* work, int B, int D, int H, int W, int nsteps, void* stream) {
    PULPO_REQUIRE(v && work && B > 0 && D >= 1 && H > 1 && W > 1 && nsteps >= 0 && (long)D * H * W < (1L << 31), "vecint_fwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)B * 3 * D * H * W, total = (long)B * D * H * W;
    // the one-launch form: ONE workgroup per batch element does every gather of every step - 16.6 us at 10^3 against 36 for seven launches, but 76 us
    // at 20^3 against 38 (scripts/vecint_probe.py): up to 2048 voxels by default (PULPO_VECINT_LDS_MAXV moves it; it fits up to 8192)
    static long lds_maxv = -1;
    if (lds_maxv < 0) { const char* e = getenv("PULPO_VECINT_LDS_MAXV"); lds_maxv = e ? atol(e) : 2048; }
    if (total / B <= std::min<long>(VI_MAXV, lds_maxv) && nsteps > 0) {
        const size_t lds = sizeof(float) * 3 * (size_t)(total / B);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vecint_fwd_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(sizeof(float) * 3 * VI_MAXV));
            if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(vecint_fwd_lds): %s", hipGetErrorString(e));
            attr_set = true;
        }
        hipLaunchKernelGGL(vecint_fwd_lds_kernel, dim3(B), dim3(VI_THREADS), lds, st, v, work, B, D, H, W, nsteps, 1.0f / (float)(1 << nsteps));
        return pulpo::check_launch("vecint_fwd_lds");
    }
    hipLaunchKernelGGL(scale_kernel, dim3(eblocks(n)), dim3(256), 0, st, v, work, 1.0f / (float)(1 << nsteps), n);
    int rc = pulpo::check_launch("vecint scale");
    if (rc) return rc;
    for (int k = 0; k < nsteps; ++k) {
        const float* cur = work + (long)k * n;
        hipLaunchKernelGGL(warp_fwd_kernel<3>, dim3(eblocks(total)), dim3(256), 0, st, cur, cur, cur, work + (long)(k + 1) * n, B, D, H, W, D, H, W, 3);
        rc = pulpo::check_launch("vecint step");
        if (rc) return rc;
    }
    return 0;
}

// fields the LDS-boxed backward step takes.  Round 5: from 8^3 up (16^3 before) - the 10^3 level of the metric's pyramid ran the plain scatter with a
// device copy in front of every squaring step (14 launches of ~5 + ~10 us); PULPO_VECINT_TILED_MIN moves the threshold (A/B switch)
static bool vecint_bwd_tiled(int D, int H, int W) {
    static int lo = -1;
    if (lo < 0) { const char* e = getenv("PULPO_VECINT_TILED_MIN"); lo = e ? atoi(e) : 8; }
    return D >= lo && H >= lo && W >= lo && D >= 2;
}

// floats of scratch pulpo_vecint_bwd needs: one buffer per step for the tiled scatter (zeroed by one fill), two for the plain scatter.
// (A one-launch variant with the gradient in LDS, as for the forward pass, was measured at 20^3: one compute unit takes 135 us per step
// for the 27 LDS atomics and 24 gathers per voxel that the tiled kernel spreads over 45 workgroups in 15 us.)
PULPO_API size_t pulpo_vecint_bwd_tmp_floats(int B, int D, int H, int W, int nsteps) {
    if (B <= 0 || D < 1 || H < 1 || W < 1 || nsteps <= 0) return 0;
    const size_t n = (size_t)B * 3 * D * H * W;
    return vecint_bwd_tiled(D, H, W) ? (size_t)nsteps * n : 2 * n;
}

// gin = d loss / d v given gout = d loss / d work[nsteps].  tmp: pulpo_vecint_bwd_tmp_floats() floats (NULL when that is 0).
PULPO_API int pulpo_vecint_bwd(const float* work, const float* gout, float* gin, float* tmp, int B, int D, int H, int W, int nsteps, void* stream) {
    PULPO_REQUIRE(work && gout && gin && B > 0 && D >= 1 && H > 1 && W > 1 && nsteps >= 0 && (long)D * H * W < (1L << 31), "vecint_bwd: bad arguments");
    PULPO_REQUIRE(tmp || pulpo_vecint_bwd_tmp_floats(B, D, H, W, nsteps) == 0, "vecint_bwd: scratch of pulpo_vecint_bwd_tmp_floats() floats required");
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)B * 3 * D * H * W, total = (long)B * D * H * W;
    const float scale = 1.0f / (float)(1 << nsteps);
    const bool tiled = vecint_bwd_tiled(D, H, W);
    if (tiled && nsteps > 0) {
        hipError_t e = hipMemsetAsync(tmp, 0, sizeof(float) * n * nsteps, st);
        if (e != hipSuccess) return pulpo::fail((int)e, "vecint_bwd fill: %s", hipGetErrorString(e));
    }
    const float* g = gout;
    for (int k = nsteps - 1; k >= 0; --k) {
        // v_{k+1} = v_k + warp(v_k, v_k):  g_k = g_{k+1} (identity) + scatter (image role) + d/d field
        const float* cur = work + (long)k * n;
        float* gp;
        if (tiled) {
            gp = tmp + (long)k * n;
            const int ntz = pulpo::cdiv(D, 4), nty = pulpo::cdiv(H, 8), ntx = pulpo::cdiv(W, 8);
            // the LAST squaring steps move by the largest fraction of the field (v / 2 at k = nsteps - 1): a box with a two-voxel rim keeps their
            // corners in LDS where the one-voxel rim sent them to memory atomics (80^3: 88 against ~40 us for the other steps).
            // PULPO_VECINT_R2_STEPS: how many of the last steps take the wider box (default 1; 0 = none, A/B switch)
            static int r2 = -1;
            if (r2 < 0) { const char* e = getenv("PULPO_VECINT_R2_STEPS"); r2 = e ? atoi(e) : 1; }
            if (k >= nsteps - r2 && (long)D * H * W >= 64L * 64 * 64)     // (below 64^3 the wider box costs more than the fallbacks it saves: 14 against 12 us)
                hipLaunchKernelGGL(vecint_bwd_tile_kernel<2>, dim3((unsigned)((long)B * ntz * nty * ntx)), dim3(256), 0, st, cur, g, gp, B, D, H, W, ntz, nty, ntx);
            else
                hipLaunchKernelGGL(vecint_bwd_tile_kernel<1>, dim3((unsigned)((long)B * ntz * nty * ntx)), dim3(256), 0, st, cur, g, gp, B, D, H, W, ntz, nty, ntx);
        } else {
            gp = tmp + (long)(k & 1) * n;
            hipError_t e = hipMemcpyAsync(gp, g, sizeof(float) * n, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return pulpo::fail((int)e, "vecint_bwd copy: %s", hipGetErrorString(e));
            hipLaunchKernelGGL(warp_bwd_kernel<true>, dim3(eblocks(total)), dim3(256), 0, st, cur, cur, g, gp, gp, B, D, H, W, D, H, W, 3);
        }
        int rc = pulpo::check_launch("vecint_bwd step");
        if (rc) return rc;
        g = gp;
    }
    hipLaunchKernelGGL(scale_kernel, dim3(eblocks(n)), dim3(256), 0, st, g, gin, scale, n);
    return pulpo::check_launch("vecint_bwd scale");
}

// ------------------------------------------------------------------------------------------------ deterministic forms (since ABI 4)
// Workspaces (caller-owned, any content on entry): see the kernels above.  Layout: [256 bytes of scale slots][int64 accumulators][float buffers].
PULPO_API size_t pulpo_warp3d_bwd_det_ws_bytes(int B, int C, int Di, int Hi, int Wi) {
    if (B <= 0 || C <= 0 || Di < 1 || Hi < 1 || Wi < 1) return 0;
    return 256 + sizeof(long long) * (size_t)B * C * Di * Hi * Wi;
}

PULPO_API int pulpo_warp3d_bwd_det(const float* df, const float* img, const float* gout, float* gdf, float* gimg, void* ws, int B, int C, int Dg, int Hg,
                                   int Wg, int Di, int Hi, int Wi, void* stream) {
    PULPO_REQUIRE(df && img && gout && B > 0 && C > 0, "warp3d_bwd_det: bad arguments");
    PULPO_REQUIRE(Dg >= 1 && Hg > 1 && Wg > 1 && (long)Dg * Hg * Wg < (1L << 31), "warp3d_bwd_det: grid H, W must be > 1 (and fewer than 2^31 voxels)");
    PULPO_REQUIRE(ws != nullptr || gimg == nullptr, "warp3d_bwd_det: workspace of pulpo_warp3d_bwd_det_ws_bytes() bytes required for the image gradient");
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)B * Dg * Hg * Wg, ni = (long)B * C * Di * Hi * Wi;
    unsigned* slots = (unsigned*)ws;
    long long* acc = gimg ? (long long*)((char*)ws + 256) : nullptr;
    if (gimg != nullptr) {
        hipError_t e = hipMemsetAsync(ws, 0, 256 + sizeof(long long) * (size_t)ni, st);
        if (e != hipSuccess) return pulpo::fail((int)e, "warp3d_bwd_det memset: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(absmax_kernel, dim3(fx_blocks((long)B * C * Dg * Hg * Wg)), dim3(256), 0, st, gout, (long)B * C * Dg * Hg * Wg, slots);
    }
    hipLaunchKernelGGL(warp_bwd_fx_kernel<false>, dim3(eblocks(total)), dim3(256), 0, st, df, img, gout, gdf, acc, slots, B, Dg, Hg, Wg, Di, Hi, Wi, C);
    if (gimg != nullptr) hipLaunchKernelGGL(fx_to_float_kernel, dim3(fx_blocks(ni)), dim3(256), 0, st, acc, gimg, ni, slots, 1.0f, (unsigned*)nullptr);
    return pulpo::check_launch("warp3d_bwd_det");
}

PULPO_API size_t pulpo_vecint_bwd_det_ws_bytes(int B, int D, int H, int W, int nsteps) {
    if (B <= 0 || D < 1 || H < 1 || W < 1 || nsteps <= 0) return 0;
    const size_t n = (size_t)B * 3 * D * H * W;
    return 256 + sizeof(long long) * n + 2 * sizeof(float) * n;
}

PULPO_API int pulpo_vecint_bwd_det(const float* work, const float* gout, float* gin, void* ws, int B, int D, int H, int W, int nsteps, void* stream) {
    PULPO_REQUIRE(work && gout && gin && B > 0 && D >= 1 && H > 1 && W > 1 && nsteps >= 0 && nsteps < 60 && (long)D * H * W < (1L << 31), "vecint_bwd_det: bad arguments");
    PULPO_REQUIRE(ws || nsteps == 0, "vecint_bwd_det: workspace of pulpo_vecint_bwd_det_ws_bytes() bytes required");
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)B * 3 * D * H * W, total = (long)B * D * H * W;
    const float scale = 1.0f / (float)(1 << nsteps);
    if (nsteps == 0) {
        hipLaunchKernelGGL(scale_kernel, dim3(eblocks(n)), dim3(256), 0, st, gout, gin, scale, n);
        return pulpo::check_launch("vecint_bwd_det scale");
    }
    unsigned* slots = (unsigned*)ws;                       // slot k: largest |gradient| entering step k's scatter
    long long* acc = (long long*)((char*)ws + 256);
    float* buf[2] = {(float*)(acc + n), (float*)(acc + n) + n};
    hipError_t e = hipMemsetAsync(ws, 0, 256 + sizeof(long long) * (size_t)n, st);
    if (e != hipSuccess) return pulpo::fail((int)e, "vecint_bwd_det memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(absmax_kernel, dim3(fx_blocks(n)), dim3(256), 0, st, gout, n, slots + (nsteps - 1));
    const float* g = gout;
    for (int k = nsteps - 1; k >= 0; --k) {
        const float* cur = work + (long)k * n;
        hipLaunchKernelGGL(warp_bwd_fx_kernel<true>, dim3(eblocks(total)), dim3(256), 0, st, cur, cur, g, (float*)nullptr, acc, slots + k, B, D, H, W, D, H, W, 3);
        float* out = k == 0 ? gin : buf[k & 1];
        hipLaunchKernelGGL(fx_to_float_kernel, dim3(fx_blocks(n)), dim3(256), 0, st, acc, out, n, slots + k, k == 0 ? scale : 1.0f, k == 0 ? (unsigned*)nullptr : slots + (k - 1));
        g = out;
    }
    return pulpo::check_launch("vecint_bwd_det");
}
