// Winograd F(2x2,3x3) in (y, x), direct over the z taps - the forward / data-gradient kernel of volumes tiled 4x8x8 with channels-last
// 16-byte-aligned operands (every ConvUnit of the BASELINE configurations from the second layer on).  Same arithmetic, weight packing,
// tile order, statistics rows and epilogue as conv3d_k3_wino2_mfma (conv3d_wino.hip), which stays the kernel of planar / odd-channel
// operands; what differs is the pipeline around the matrix loop:
//
//   * TWO halo images in LDS.  The x-transformed halo of chunk c + 1 (or of the next tile's chunk 0) is staged into the image that is not
//     being read while the MFMAs of chunk c run: raw loads in the chunk's first point steps, transform + ds_write a few steps later, all of
//     it between MFMAs.  ONE barrier per 8-channel chunk (96 MFMAs per wave) instead of four, and no staging phase in which a workgroup
//     issues no MFMA.
//   * Weights never touch LDS.  Wave py is the only consumer of the points (py, 0..3) of a cout tile, so a slab gave no reuse between
//     waves - it was a latency buffer.  A lane's B operands of four k-steps are 16 contiguous bytes of the packed weights: they are loaded
//     global -> registers one dz iteration ahead (the register a point step has just used is re-loaded for the next iteration), counted
//     by the compiler's own vmcnt bookkeeping; no LDS-DMA, no vmcnt(0), no weight ds_reads, 32 KB of LDS per workgroup returned.
//   * Image = [channel quad][row] float4s (rows = (hz, px, hy, x-pair), planes of 164 rows): the 16 lanes of every ds_read_b128 lane group
//     fall on 16 different 16-byte slots without pad floats (rows {0-3, 24-27} of one plane and {8-11, 16-19} of the next, plane stride 4 mod 16).
#include "conv_shared.h"
#include <stdlib.h>

#ifndef PULPO_W2P_FMA_BATCH
#define PULPO_W2P_FMA_BATCH 1
#endif
#ifndef PULPO_ABL
#define PULPO_ABL 0          // diagnostic builds (scripts/ablate.py): timings only, results are garbage.  Bits: 1 no epilogue, 2 no halo staging,
#endif                       // 4 no weight re-loads, 8 no chunk barrier, 16 no MFMAs (one v_fma each), 32 no operand-row reads inside the loop, 64 phase stamps,
                             // 128 no output stores (fast path), 256 no y combination (one operand row per row tile, no v_fma: the loop of a 2-D-staged image),
                             // 512 staging without transform + LDS writes (loads and their waits only), 1024 staging without loads (transform + writes only)

#if PULPO_ABL & 64
// g_stamps[block][0] = HW_REG_HW_ID, [1] = HW_REG_XCC_ID, [2] = start clock, [3 + 2k] / [4 + 2k] = main-loop end / tile end of the block's k-th tile
__device__ unsigned long long g_stamps[512 * 80];
#define STAMP(slot, val) do { if (threadIdx.x == 0 && (slot) < 80) g_stamps[blockIdx.x * 80 + (slot)] = (val); } while (0)
PULPO_API int pulpo_debug_read_stamps(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#else
#define STAMP(slot, val) do {} while (0)
#endif

namespace {

using namespace pulpo_conv;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int P_CH = 8, P_NT = 32;
constexpr int P_PL = HY * 4;                     // rows of one px slice of a plane: (hy, x-pair)
constexpr int P_PLROWS = 4 * P_PL + 4;           // rows per hz plane (4 mod 16: see above)
constexpr int P_QROWS = 6 * P_PLROWS + 4;        // rows per channel quad (stride 16 dwords mod 32: the two quads of a ds_write_b128 group use different banks)
constexpr int P_IMGF = 2 * P_QROWS * 4;          // floats of one halo image
constexpr int P_RH = 4 * 2 * 16 * 64;            // floats of the cross-wave exchange buffer of one row tile: [py][ox][r][lane]
constexpr int P_IMG = P_RH + 4 * 2 * P_NT;       // floats of one image region (halo image, or exchange buffer + statistics rows)
static_assert(P_IMGF <= P_IMG, "halo image must fit its region");
constexpr int P_NITEM = 6 * HY * 4 * 2;          // staging items of a chunk: (hz, hy, x-pair, channel quad), two per thread
constexpr int P_TAB_MAX = 3 * 1024;               // floats of the per-channel table [3][ncot * 32]: up to 1024 output channels
constexpr size_t P_LDS = (size_t)(2 * P_IMG + P_TAB_MAX) * sizeof(float);
static_assert(2 * P_LDS <= 160 * 1024, "two workgroups per CU");

// side-work schedule of a chunk's 12 point steps (s = dz * 4 + px): raw loads of staging item u at steps P_LD[u], P_LD[u] + 1 (two taps
// each), its transform + four ds_write_b128 at step P_XF[u]
constexpr int P_LD0 = 0, P_XF0 = 4, P_LD1 = 5, P_XF1 = 10;      // (one item's raw registers live at a time)

template <bool BNR>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wino2p_mfma(ConvArgs a) {
    constexpr int CH = P_CH, NT = P_NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kk = lane >> 5;
    const int nchunk = (a.Cin + CH - 1) / CH;
    const int ksplit = a.ksplit;                        // > 1: the chunks of a (tile, cout tile) pair are split over ksplit work items, each storing a partial slab
    const int nwork = a.B * a.ntz * a.nty * a.ntx * a.ncot * ksplit;
    const int nwg = gridDim.x;
    const unsigned ps_bytes = (unsigned)a.in_ps * 4u;

    // ---- tile-invariant per-thread data: the two staging items (hz, hy, x-pair, quad) of this thread
    unsigned roff[2];                                   // byte offset of tap 0 relative to the tile's halo origin
    int lofs[2];                                        // float offset of the item's px = 0 row in an image
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = tid + u * 256;
        const int q = j & 1, rb = j >> 1;
        const int xb = rb & 3, hrow = rb >> 2;
        const int hz = hrow / HY, hy = hrow - hz * HY;
        roff[u] = ((unsigned)((hz * a.H + hy) * a.W + 2 * xb) * (unsigned)a.in_ps + 4u * q) * 4u;
        lofs[u] = (q * P_QROWS + hz * P_PLROWS + hy * 4 + xb) * 4;
    }
    const bool item1 = tid + 256 < P_NITEM;
    // y combination of this wave's point row: v = X[2 yb + ta] + sa * X[2 yb + tb]   (same table as the x transform)
    const int py = wave;
    const int ta = py == 0 ? 0 : py == 2 ? 2 : 1;
    const int tb = py == 2 ? 1 : py == 3 ? 3 : 2;
    const float sa = py == 1 ? 1.f : -1.f;
    // MFMA row i of row tile m = block (z = 2 m + (i >> 4), yb = (i >> 2) & 3, xb = i & 3)
    const int lrow = (i >> 4) * P_PLROWS + ((i >> 2) & 3) * 8 + (i & 3);
    const int pa_off = (kk * P_QROWS + lrow + ta * 4) * 4;
    const int pb_off = (kk * P_QROWS + lrow + tb * 4) * 4;

    // Operands are read through buffer descriptors: 32-bit offsets instead of 64-bit per-lane pointers, and a halo tap outside the volume is
    // a load at an offset beyond num_records, which returns zeros - no branch inside the matrix loop.  (host: volume bytes < 2^31)
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, -1, 0x00020000);
    const int in_bytes = (int)((long)a.D * a.H * a.W * a.in_ps * 4);
    struct Tile {
        int tile_lin, b, z0, y0, x0, co0;
        unsigned wbase;              // byte offset of (chunk c0, dz 0, point (py, 0), cout co0) in the packed weights
        int c0, c1, ks;              // chunk range [c0, c1) of this work item and its split index (0 .. ksplit - 1)
    };
    auto describe = [&](int work) {
        Tile t;
        t.ks = work % ksplit;                           // (splits of a pair are neighbours: they share the halo in L2)
        work /= ksplit;
        t.c0 = nchunk * t.ks / ksplit; t.c1 = nchunk * (t.ks + 1) / ksplit;
        const int cot = work % a.ncot;
        int q = work / a.ncot;
        int tx_, ty_, tz_;
        if (a.tile_order == 1) {
            // blocked order: consecutive work items walk 4 x 4 x 4 blocks of tiles (host: ntx, nty, ntz multiples of 4).  The 64 workgroups of
            // an XCD hold 64 consecutive tiles at any time: as one compact block their halos (18 x 34 x 34 voxels, 2.7 MB at 32 channels)
            // fit the XCD's 4 MiB L2, so the four chunk passes over a voxel's 128-byte line find it there instead of fetching it again
            const int nt = a.ntx * a.nty * a.ntz;
            t.b = q / nt;
            q -= t.b * nt;
            const int blk = q >> 6, w = q & 63;
            const int nbx = a.ntx >> 2, nby = a.nty >> 2;
            const int bx = blk % nbx, by = (blk / nbx) % nby, bz = blk / (nbx * nby);
            tx_ = bx * 4 + (w & 3); ty_ = by * 4 + ((w >> 2) & 3); tz_ = bz * 4 + (w >> 4);
        } else {
            tx_ = q % a.ntx; q /= a.ntx;
            ty_ = q % a.nty; q /= a.nty;
            tz_ = q % a.ntz;
            t.b = q / a.ntz;
        }
        t.tile_lin = ((t.b * a.ntz + tz_) * a.nty + ty_) * a.ntx + tx_;        // row of the statistics buffer: the canonical order either way
        t.z0 = tz_ * 4; t.y0 = ty_ * TY; t.x0 = tx_ * TX;
        t.co0 = cot * NT;
        t.wbase = (unsigned)((py * 4) * a.NPad + t.co0) * (CH * 4u) + (unsigned)t.c0 * 3u * (16u * CH * a.NPad * 4u);
        return t;
    };
    // byte offsets of this thread's eight (item, tap) loads of a tile's halo, channel 4 rq of chunk 0, relative to the batch element; OOB where
    // the tap lies outside the volume (or the item does not exist).  The chunk's channel offset rides in the instruction's scalar offset.
    unsigned hoff[2][4];
    auto halo_offsets = [&](const Tile& t) {
        const unsigned origin = (unsigned)(((t.z0 - 1) * a.H + (t.y0 - 1)) * a.W + (t.x0 - 1)) * ps_bytes;      // modulo 2^32: may be "negative"
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = tid + u * 256;
            const int rb = j >> 1;
            const int xb = rb & 3, hrow = rb >> 2;
            const int hz = hrow / HY, hy = hrow - hz * HY;
            const bool rowok = j < P_NITEM && (unsigned)(t.z0 - 1 + hz) < (unsigned)a.D && (unsigned)(t.y0 - 1 + hy) < (unsigned)a.H;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
                hoff[u][tt] = (rowok && (unsigned)(t.x0 - 1 + 2 * xb + tt) < (unsigned)a.W) ? origin + roff[u] + tt * ps_bytes : OOB;
        }
    };
    auto in_rsrc = [&](int b) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + (long)b * a.in_bs), 0, in_bytes, 0x00020000);
    };

    float4 raw[4];                                      // the four x taps of ONE staging item
    // (host: Cin % 8 == 0, so every chunk has both channel quads)
    auto load_raw = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned c0_bytes, int u, int tt) {
        raw[tt] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)hoff[u][tt], (int)c0_bytes, 0));
    };
    auto store_item = [&](float* img, int u) {
        if (u == 0 || item1) {
            float* o = img + lofs[u];
            const float4 d0 = raw[0], d1 = raw[1], d2 = raw[2], d3 = raw[3];
            *reinterpret_cast<float4*>(o) = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
            *reinterpret_cast<float4*>(o + P_PL * 4) = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
            *reinterpret_cast<float4*>(o + 2 * P_PL * 4) = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
            *reinterpret_cast<float4*>(o + 3 * P_PL * 4) = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
        }
    };
    const unsigned w_it_stride = 16u * CH * a.NPad * 4u;    // bytes between consecutive (chunk, dz) slabs
    const unsigned w_px_stride = (unsigned)CH * a.NPad * 4u;    // bytes between consecutive points
    const int wl_off = (i * CH + 4 * kk) * 4;               // this lane's 16 bytes inside a point's [32 n][8 k] piece
    auto load_w = [&](unsigned wofs, int px) {
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wl_off, (int)(wofs + px * w_px_stride), 0));
    };

    if (a.stagger > 0 && (int)blockIdx.x >= (nwg >> 1)) {       // start-up offset of the second half of the grid (blocks b and b + nwg / 2 share a CU)
        for (int s_ = 0; s_ < a.stagger; ++s_) __builtin_amdgcn_s_sleep(127);
    }
    // ---- per-channel constants of the epilogue (bias | BatchNorm mean, scale, shift), once per workgroup into LDS: the epilogue reads them
    // with ds_read_b128 instead of waiting for global loads at the head of every tile
    float* const tab = smem + 2 * P_IMG;                // [3][ctab]
    const int ctab = a.ncot * NT;
    for (int c = tid; c < ctab; c += 256) {
        const bool in = c < a.Cout;
        float t0 = 0.f, t1 = 1.f, t2 = 0.f;
        if (BNR) {
            if (in) { t0 = a.bn_coef[c]; t1 = a.bn_coef[2 * a.Cout + c]; t2 = a.bn_coef[3 * a.Cout + c]; }
        } else {
            if (in && a.bias != nullptr) t0 = a.bias[c];
            if (in && a.coef != nullptr) { t1 = a.coef[2 * a.Cout + c]; t2 = a.coef[3 * a.Cout + c]; }
        }
        tab[c] = t0; tab[ctab + c] = t1; tab[2 * ctab + c] = t2;
    }
    STAMP(0, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));        // HW_REG_HW_ID
    STAMP(1, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));       // HW_REG_XCC_ID
    STAMP(2, __builtin_amdgcn_s_memtime());
    [[maybe_unused]] int tile_no = 0;
    int work = pulpo::xcd_remap(blockIdx.x, nwg);
    Tile cur = describe(work);
    int cb = 0;                                         // image being read
    // ---- prologue: chunk 0 of the first tile, the first weight rows
    float4 wr[4];
#pragma unroll
    for (int px = 0; px < 4; ++px) wr[px] = load_w(cur.wbase, px);
    {
        const __amdgpu_buffer_rsrc_t rs0 = in_rsrc(cur.b);
        halo_offsets(cur);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) load_raw(rs0, (unsigned)cur.c0 * CH * 4u, u, tt);
            store_item(smem, u);
        }
    }
    __syncthreads();

    float4 ra[2][2], rb[2][2];                          // two register sets of operand rows ((ta, tb) x two row tiles)

    for (;;) {
        // (the next tile is described once, here: inside the chunk loop - under `last_chunk` - the compiler if-converted the description into
        //  every chunk's instruction stream, scalar divisions and spilled-register reloads included; conv3d_wino3.hip, same finding)
        const int next_work = work + nwg;
        const bool has_next = next_work < nwork;
        const Tile nxt = has_next ? describe(next_work) : cur;

        f32x16 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][p][r] = 0.f;

        unsigned wnext = cur.wbase + w_it_stride;       // weights of iteration it + 1
        int it = 0;
        for (int chunk = cur.c0; chunk < cur.c1; ++chunk) {
            const bool last_chunk = chunk + 1 == cur.c1;
            constexpr bool stage = !(PULPO_ABL & 2);
            // what is staged underneath this chunk's MFMAs: the tile's next chunk, or chunk 0 of the next tile (after the last tile: the
            // tile's own chunk 0 again, into an image nobody reads - cheaper than a branch around every piece of the side work)
            if (last_chunk) halo_offsets(has_next ? nxt : cur);
            const unsigned st_c0 = (unsigned)(last_chunk ? (has_next ? nxt.c0 : cur.c0) : chunk + 1) * CH * 4u;
            const __amdgpu_buffer_rsrc_t st_rs = in_rsrc(last_chunk ? nxt.b : cur.b);
            const float* img_r = smem + cb * P_IMG;
            float* img_w = smem + (cb ^ 1) * P_IMG;
            const float* pa = img_r + pa_off;
            const float* pb = img_r + pb_off;
            auto fetch_a = [&](int dz, int px, int slot) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int off = ((2 * m + dz) * P_PLROWS + px * P_PL) * 4;
                    ra[slot][m] = *reinterpret_cast<const float4*>(pa + off);
#if !(PULPO_ABL & 256)
                    rb[slot][m] = *reinterpret_cast<const float4*>(pb + off);
#endif
                }
            };
            fetch_a(0, 0, 0);
#pragma unroll
            for (int dz = 0; dz < 3; ++dz, ++it) {
                const bool tile_end = dz == 2 && last_chunk;
                // (after the last tile the re-load fetches the tile's own first rows again: no branch around a load inside the loop, so the
                //  compiler's vmcnt counts stay exact)
                constexpr bool more_w = !(PULPO_ABL & 4);
                const unsigned wsrc = tile_end ? (has_next ? nxt.wbase : cur.wbase) : wnext;
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    const int s = dz * 4 + px;
                    if (!(PULPO_ABL & 32)) {
                        if (px + 1 < 4) fetch_a(dz, px + 1, (px + 1) & 1);
                        else if (dz < 2) fetch_a(dz + 1, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int sl = px & 1;
                    const float wv[4] = {wr[px].x, wr[px].y, wr[px].z, wr[px].w};
#if PULPO_W2P_FMA_BATCH
                    // the step's eight y combinations first, then eight MFMAs back to back (no VALU -> MFMA dependency stall between them)
                    float av[4][2];
#if PULPO_ABL & 256
#pragma unroll
                    for (int m = 0; m < 2; ++m) { av[0][m] = ra[sl][m].x; av[1][m] = ra[sl][m].y; av[2][m] = ra[sl][m].z; av[3][m] = ra[sl][m].w; }
#else
                    // (two-wide vector arithmetic: four v_pk_fma_f32 per step instead of eight v_fma_f32)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const f32x2 sav = {sa, sa};
                        const f32x2 lo = __builtin_elementwise_fma(sav, f32x2{rb[sl][m].x, rb[sl][m].y}, f32x2{ra[sl][m].x, ra[sl][m].y});
                        const f32x2 hi = __builtin_elementwise_fma(sav, f32x2{rb[sl][m].z, rb[sl][m].w}, f32x2{ra[sl][m].z, ra[sl][m].w});
                        av[0][m] = lo.x; av[1][m] = lo.y; av[2][m] = hi.x; av[3][m] = hi.y;
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
#if PULPO_W2P_FMA_BATCH
                            const float v_ = av[s2][m];
#else
                            const float a_ = s2 == 0 ? ra[sl][m].x : s2 == 1 ? ra[sl][m].y : s2 == 2 ? ra[sl][m].z : ra[sl][m].w;
                            const float b_ = s2 == 0 ? rb[sl][m].x : s2 == 1 ? rb[sl][m].y : s2 == 2 ? rb[sl][m].z : rb[sl][m].w;
                            const float v_ = fmaf(sa, b_, a_);
#endif
#if PULPO_ABL & 16
                            acc[m][px][s2] = fmaf(v_, wv[s2], acc[m][px][s2]);
#else
                            acc[m][px] = __builtin_amdgcn_mfma_f32_32x32x2f32(v_, wv[s2], acc[m][px], 0, 0, 0);
#endif
                        }
                        if (s2 == 0) {                  // behind the step's first MFMAs: their 128 pipe clocks cover the issue of the side work
                            __builtin_amdgcn_sched_barrier(0);
                            if (stage) {
#if !(PULPO_ABL & 1024)
                                if (s == P_LD0) { load_raw(st_rs, st_c0, 0, 0); load_raw(st_rs, st_c0, 0, 1); }
                                if (s == P_LD0 + 1) { load_raw(st_rs, st_c0, 0, 2); load_raw(st_rs, st_c0, 0, 3); }
                                if (s == P_LD1) { load_raw(st_rs, st_c0, 1, 0); load_raw(st_rs, st_c0, 1, 1); }
                                if (s == P_LD1 + 1) { load_raw(st_rs, st_c0, 1, 2); load_raw(st_rs, st_c0, 1, 3); }
#endif
#if PULPO_ABL & 512
                                if (s == P_XF0 || s == P_XF1) asm volatile("" : : "v"(raw[0].x), "v"(raw[1].x), "v"(raw[2].x), "v"(raw[3].x));      // (wait only)
#else
                                if (s == P_XF0) store_item(img_w, 0);
                                if (s == P_XF1) store_item(img_w, 1);
#endif
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    // the weights of this point are consumed (issued): fetch the same point of the next iteration into the register
                    if (more_w) wr[px] = load_w(wsrc, px);
                    __builtin_amdgcn_sched_barrier(0);
                }
                wnext += w_it_stride;
            }
            // The first three weight rows of the next iteration (re-loaded 1 - 3 point steps ago) are "used" in front of the barrier: the chunk's
            // first point steps then need no vmcnt wait on ANY path into them.  (A wait there is one static count for both the chunk and the
            // tile back edge; behind an epilogue it would stand for "all but the youngest few" of that tile's output stores.)
#pragma unroll
            for (int px = 0; px < 3; ++px) asm volatile("" : : "v"(wr[px].x), "v"(wr[px].y), "v"(wr[px].z), "v"(wr[px].w));
            if (!(PULPO_ABL & 8)) __syncthreads();      // image cb ^ 1 complete and visible; every wave has left image cb
            cb ^= 1;
        }

        STAMP(3 + 6 * tile_no, __builtin_amdgcn_s_memtime());
        // ---- epilogue: x inverse transform in registers, y inverse transform across the four waves through LDS, one row tile (two z-planes) at
        // a time; the exchange buffer is the image the last chunk was read from (the next tile's chunk 0 already sits in the other one)
        float* R = smem + (cb ^ 1) * P_IMG;                 // [py][ox][r][lane]
        float* red = R + P_RH;                              // [4 waves][2][NT]
        // split-K work items store their partial sums into slab `ks` of a.part ([ksplit][B * V][Cout], dense channels-last; bias from split 0
        // only); pulpo_conv::launch_splitk_reduce adds the slabs up in fixed order and produces the BatchNorm partials / the eval-mode store
        const bool split = ksplit > 1;
        const long o_ps = split ? (long)a.Cout : a.out_ps, o_cs = split ? 1L : a.out_cs;
        float* out_b = split ? a.part + ((long)cur.ks * a.B + cur.b) * ((long)a.D * a.H * a.W) * a.Cout : a.out + (long)cur.b * a.out_bs;
        const int z0 = cur.z0, y0 = cur.y0, x0 = cur.x0, co0 = cur.co0;
        // Fast path (all 32 couts real, channels-last 16-byte aligned output - every tile of the BASELINE layers): lane = (channel quad q, row
        // half, row group): the four waves' partial rows are fetched with ds_read_b128, the y inverse transform is done on float4s, and each
        // voxel leaves as one 128-byte line written by 8 lanes x 16 bytes; voxels of a ragged tile outside the volume are masked per store.
        // (BNR: the host launches this instantiation only when every tile is whole; without bias and without the eval-mode store)
        const bool whole = BNR || (z0 + 4 <= a.D && y0 + TY <= a.H && x0 + TX <= a.W);
        const bool fast = BNR || (o_cs == 1 && (o_ps & 3) == 0 && (split || (a.out_bs & 3) == 0) && (((uintptr_t)out_b) & 15) == 0 && co0 + NT <= a.Cout);
        // (the lane id passes through an opaque asm: what the epilogue derives from it is computed here and not hoisted above the main loop)
        int elane = lane;
        asm volatile("" : "+v"(elane));
        const int q = elane & 7, kh = (elane >> 3) & 1, g = elane >> 4;
        const int ei = elane & 31, ekk = elane >> 5;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 s4 = zero4, q4 = zero4;
        float ssum = 0.f, ssq = 0.f;
        const bool fuse = !BNR && !split && a.coef != nullptr;
        const bool cok = co0 + ei < a.Cout;
        const bool with_bias = !split || cur.ks == 0;
        float* const stats = split ? nullptr : a.stats;
        const bool bnr = BNR && fast;
        // fast path: four channels per lane; general path: channel co0 + i.  From the workgroup's table in LDS: [0] the bias (BNR: the channel
        // means rounded to fp32 - pulpo_bn_bwd_finalize corrects for the rounding), [1] / [2] scale and shift of the eval-mode store (BNR: of
        // the unit whose BatchNorm-backward sums this launch delivers)
        const float4 b4 = (BNR || !with_bias) ? zero4 : *reinterpret_cast<const float4*>(tab + co0 + 4 * q);
        const float4 bm4 = *reinterpret_cast<const float4*>(tab + co0 + 4 * q);
        const float4 sc4 = *reinterpret_cast<const float4*>(tab + ctab + co0 + 4 * q);
        const float4 sh4 = *reinterpret_cast<const float4*>(tab + 2 * ctab + co0 + 4 * q);
        const float bias1 = with_bias ? tab[co0 + ei] : 0.f, fsc1 = tab[ctab + co0 + ei], fsh1 = tab[2 * ctab + co0 + ei];
        const float* bn_b = bnr ? a.bn_y + (long)cur.b * a.bn_y_bs + co0 + 4 * q : nullptr;
        STAMP(4 + 6 * tile_no, __builtin_amdgcn_s_memtime());
#pragma unroll
        for (int m = 0; m < ((PULPO_ABL & 1) ? 0 : 2); ++m) {
            if (m > 0) { STAMP(6 + 6 * tile_no, __builtin_amdgcn_s_memtime()); __syncthreads(); }                     // every wave has left the exchange buffer (previous row tile)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m0 = acc[m][0][r], m1 = acc[m][1][r], m2 = acc[m][2][r], m3 = acc[m][3][r];
                R[((py * 2 + 0) * 16 + r) * 64 + elane] = m0 + m1 + m2;
                R[((py * 2 + 1) * 16 + r) * 64 + elane] = m1 - m2 - m3;
            }
            float4 yv[2][2];                                // (bnr) the pre-norm activations of this lane's four voxels
            if (bnr) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int combo = h * 16 + wave * 4 + g;
                    const int ox = combo >> 4, r = combo & 15;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int gz = z0 + 2 * m + (row >> 4), gy = y0 + 2 * ((row >> 2) & 3), gx = x0 + 2 * (row & 3) + ox;
                    const long vox = (long)(gz * a.H + gy) * a.W + gx;
                    yv[h][0] = *reinterpret_cast<const float4*>(bn_b + vox * a.bn_y_ps);
                    yv[h][1] = *reinterpret_cast<const float4*>(bn_b + (vox + a.W) * a.bn_y_ps);
                }
            }
            __syncthreads();
            if (m == 0) STAMP(5 + 6 * tile_no, __builtin_amdgcn_s_memtime());
            if (fast) {
                float4 tq[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int combo = h * 16 + wave * 4 + g;               // (ox, r) = (combo >> 4, combo & 15)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        tq[h][p] = *reinterpret_cast<const float4*>(R + ((p * 2 + (combo >> 4)) * 16 + (combo & 15)) * 64 + kh * 32 + 4 * q);
                }
                float* obase = out_b + co0 + 4 * q;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int combo = h * 16 + wave * 4 + g;
                    const int ox = combo >> 4, r = combo & 15;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int gz = z0 + 2 * m + (row >> 4), gy = y0 + 2 * ((row >> 2) & 3), gx = x0 + 2 * (row & 3) + ox;
                    const long vox = (long)(gz * a.H + gy) * a.W + gx;
                    const bool in0 = whole || (gz < a.D && gy < a.H && gx < a.W), in1 = whole || (gz < a.D && gy + 1 < a.H && gx < a.W);
                    const float4 t0 = tq[h][0], t1 = tq[h][1], t2 = tq[h][2], t3 = tq[h][3];
                    float4 v0 = make_float4(t0.x + t1.x + t2.x + b4.x, t0.y + t1.y + t2.y + b4.y, t0.z + t1.z + t2.z + b4.z, t0.w + t1.w + t2.w + b4.w);
                    float4 v1 = make_float4(t1.x - t2.x - t3.x + b4.x, t1.y - t2.y - t3.y + b4.y, t1.z - t2.z - t3.z + b4.z, t1.w - t2.w - t3.w + b4.w);
                    if (bnr) {
                        // dbn = dz * lrelu'(bn(y));  sums of dbn and of dbn * (y - fp32 mean): all fp32 (what the ROUNDED mean leaves out is
                        // added back, in double, by the finalize kernel: sum dbn * xhat = rstd * (sum dbn * (y - m32) - (mean - m32) * sum dbn))
                        auto red1 = [&](float dzv, float yy, float sc, float sh, float m32, float& s_, float& q_) {
                            const float bn = yy * sc + sh;
                            const float d = bn > 0.f ? dzv : dzv * a.slope;
                            s_ += d;
                            q_ = fmaf(d, yy - m32, q_);
                        };
                        red1(v0.x, yv[h][0].x, sc4.x, sh4.x, bm4.x, s4.x, q4.x);
                        red1(v0.y, yv[h][0].y, sc4.y, sh4.y, bm4.y, s4.y, q4.y);
                        red1(v0.z, yv[h][0].z, sc4.z, sh4.z, bm4.z, s4.z, q4.z);
                        red1(v0.w, yv[h][0].w, sc4.w, sh4.w, bm4.w, s4.w, q4.w);
                        red1(v1.x, yv[h][1].x, sc4.x, sh4.x, bm4.x, s4.x, q4.x);
                        red1(v1.y, yv[h][1].y, sc4.y, sh4.y, bm4.y, s4.y, q4.y);
                        red1(v1.z, yv[h][1].z, sc4.z, sh4.z, bm4.z, s4.z, q4.z);
                        red1(v1.w, yv[h][1].w, sc4.w, sh4.w, bm4.w, s4.w, q4.w);
                    } else if (whole) {
                        s4.x += v0.x + v1.x; s4.y += v0.y + v1.y; s4.z += v0.z + v1.z; s4.w += v0.w + v1.w;
                        q4.x += v0.x * v0.x + v1.x * v1.x; q4.y += v0.y * v0.y + v1.y * v1.y; q4.z += v0.z * v0.z + v1.z * v1.z; q4.w += v0.w * v0.w + v1.w * v1.w;
                    } else {
                        if (in0) { s4.x += v0.x; s4.y += v0.y; s4.z += v0.z; s4.w += v0.w; q4.x += v0.x * v0.x; q4.y += v0.y * v0.y; q4.z += v0.z * v0.z; q4.w += v0.w * v0.w; }
                        if (in1) { s4.x += v1.x; s4.y += v1.y; s4.z += v1.z; s4.w += v1.w; q4.x += v1.x * v1.x; q4.y += v1.y * v1.y; q4.z += v1.z * v1.z; q4.w += v1.w * v1.w; }
                    }
                    if (fuse) {
                        auto act = [&](float v, float sc, float sh) { const float tt = v * sc + sh; return tt > 0.f ? tt : tt * a.slope; };
                        v0 = make_float4(act(v0.x, sc4.x, sh4.x), act(v0.y, sc4.y, sh4.y), act(v0.z, sc4.z, sh4.z), act(v0.w, sc4.w, sh4.w));
                        v1 = make_float4(act(v1.x, sc4.x, sh4.x), act(v1.y, sc4.y, sh4.y), act(v1.z, sc4.z, sh4.z), act(v1.w, sc4.w, sh4.w));
                    }
#if PULPO_ABL & 128
                    if (v0.x == 12345.678f && v1.y == 9876.54f)
#endif
                    {
                        if (in0) *reinterpret_cast<float4*>(obase + vox * o_ps) = v0;
                        if (in1) *reinterpret_cast<float4*>(obase + (vox + a.W) * o_ps) = v1;
                    }
                }
            } else {
                // general path: ragged tiles (volume edge), partial cout tiles, planar / strided outputs.  Wave w finishes x parity w & 1, rows
                // 8 (w >> 1) .. 8 (w >> 1) + 7 of the row tile; lane = (row half kk, channel i)
                const int fox = wave & 1;
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = 8 * (wave >> 1) + rr;
                    float tq[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) tq[p] = R[((p * 2 + fox) * 16 + r) * 64 + elane];
                    float v0 = tq[0] + tq[1] + tq[2] + bias1, v1 = tq[1] - tq[2] - tq[3] + bias1;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * ekk;
                    const int gz = z0 + 2 * m + (row >> 4), gy = y0 + 2 * ((row >> 2) & 3), gx = x0 + 2 * (row & 3) + fox;
                    if (cok && gz < a.D && gx < a.W) {
                        const long vox = (long)(gz * a.H + gy) * a.W + gx;
                        if (gy < a.H) {
                            ssum += v0; ssq += v0 * v0;
                            if (fuse) { const float tt = v0 * fsc1 + fsh1; v0 = tt > 0.f ? tt : tt * a.slope; }
                            out_b[vox * o_ps + (long)(co0 + ei) * o_cs] = v0;
                        }
                        if (gy + 1 < a.H) {
                            ssum += v1; ssq += v1 * v1;
                            if (fuse) { const float tt = v1 * fsc1 + fsh1; v1 = tt > 0.f ? tt : tt * a.slope; }
                            out_b[(vox + a.W) * o_ps + (long)(co0 + ei) * o_cs] = v1;
                        }
                    }
                }
            }
        }
        STAMP(7 + 6 * tile_no, __builtin_amdgcn_s_memtime());
        // per-tile BatchNorm partial sums: reduce over the lanes that hold the same channel(s), then over the four waves.  The barrier also
        // separates the exchange buffer's last reads from the next tile's staging into the same image, so it is taken without statistics too.
        if (stats != nullptr) {
            if (fast) {
#pragma unroll
                for (int o = 8; o <= 32; o <<= 1) {
                    s4.x += __shfl_xor(s4.x, o, 64); s4.y += __shfl_xor(s4.y, o, 64); s4.z += __shfl_xor(s4.z, o, 64); s4.w += __shfl_xor(s4.w, o, 64);
                    q4.x += __shfl_xor(q4.x, o, 64); q4.y += __shfl_xor(q4.y, o, 64); q4.z += __shfl_xor(q4.z, o, 64); q4.w += __shfl_xor(q4.w, o, 64);
                }
                if (elane < 8) {
                    *reinterpret_cast<float4*>(red + (wave * 2 + 0) * NT + 4 * q) = s4;
                    *reinterpret_cast<float4*>(red + (wave * 2 + 1) * NT + 4 * q) = q4;
                }
            } else {
                ssum += __shfl_xor(ssum, 32, 64);
                ssq += __shfl_xor(ssq, 32, 64);
                if (elane < 32) {
                    red[(wave * 2 + 0) * NT + ei] = ssum;
                    red[(wave * 2 + 1) * NT + ei] = ssq;
                }
            }
        }
        __syncthreads();
        if (stats != nullptr && tid < 2 * NT) {
            const int which = tid / NT, c = tid - which * NT;
            if (co0 + c < a.Cout) {
                const float tot = red[(0 * 2 + which) * NT + c] + red[(1 * 2 + which) * NT + c] + red[(2 * 2 + which) * NT + c] +
                                  red[(3 * 2 + which) * NT + c];
                stats[((long)cur.tile_lin * 2 + which) * a.Cout + co0 + c] = tot;
            }
        }
#if PULPO_ABL & 1
        if (acc[0][0][0] + acc[1][1][1] + acc[0][2][2] + acc[1][3][3] + acc[0][1][5] + acc[1][0][7] + acc[0][3][9] + acc[1][2][11] == 12345.678f) out_b[tid] = 1.f;
#endif
        STAMP(8 + 6 * tile_no, __builtin_amdgcn_s_memtime());
        ++tile_no;
        if (!has_next) break;
        cur = nxt;
        work = next_work;
    }
}

}  // namespace

namespace pulpo_conv {

bool wino2p_ok(const ConvArgs& a) {
    return a.Cin % P_CH == 0 && (long)a.D * a.H * a.W * a.in_ps * 4 < (1L << 31) && 3 * ((a.Cout + P_NT - 1) / P_NT) * P_NT <= P_TAB_MAX;
}

// launch of the pipelined (y, x) Winograd kernel; operands channels-last, 16-byte aligned, K % 8 == 0, volume bytes < 2^31 (wino2p_ok)
int launch_wino2p(const ConvArgs& a, int nblk, bool bnr, hipStream_t st) {
    static bool attr_set[2] = {false, false};
    const void* fn = bnr ? reinterpret_cast<const void*>(&conv3d_k3_wino2p_mfma<true>) : reinterpret_cast<const void*>(&conv3d_k3_wino2p_mfma<false>);
    if (!attr_set[bnr]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_LDS);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(conv3d wino2p): %s", hipGetErrorString(e));
        attr_set[bnr] = true;
    }
    {
        static int pct = -1;                            // PULPO_W2P_STAGGER = start-up offset of the second workgroup of a CU, % of a tile's time
        if (pct < 0) { const char* e = getenv("PULPO_W2P_STAGGER"); pct = e ? atoi(e) : 0; }
        const long clocks = ((long)((a.Cin + P_CH - 1) / P_CH) * 96 * 64 * 2 + 6000) * pct / 100;
        const_cast<ConvArgs&>(a).stagger = nblk >= 512 ? (int)(clocks / (64 * 127)) : 0;
    }
    {
        // PULPO_W2P_TILE_ORDER: 1 (default) blocked 4 x 4 x 4 where the tile grid allows (the 160^3 layers), 0 linear (x fastest).  Measured with
        // rocprofv3 --pmc FETCH_SIZE on 32 -> 32 @ 160^3 / 128 -> 128 @ 40^3: 28 % fewer bytes over the fabric, 1 % less time stand-alone.
        static int order = -1;
        if (order < 0) { const char* e = getenv("PULPO_W2P_TILE_ORDER"); order = e ? atoi(e) : 1; }
        const_cast<ConvArgs&>(a).tile_order = (order == 1 && a.ntx % 4 == 0 && a.nty % 4 == 0 && a.ntz % 4 == 0) ? 1 : 0;
    }
    // persistent workgroups: two per CU
    if (bnr) hipLaunchKernelGGL((conv3d_k3_wino2p_mfma<true>), dim3(std::min(nblk, 512)), dim3(256), P_LDS, st, a);
    else hipLaunchKernelGGL((conv3d_k3_wino2p_mfma<false>), dim3(std::min(nblk, 512)), dim3(256), P_LDS, st, a);
    return pulpo::check_launch("conv3d_k3_wino2p_mfma");
}

}  // namespace pulpo_conv
