// Winograd F(2x2x2, 3x3x3) - minimal filtering in ALL three axes - for the forward / data-gradient convolution: 64 transformed points per
// 2 x 2 x 2 output block instead of the 216 multiply-adds of the direct form (the (y, x) kernel of conv3d_wino2p.hip issues 96): 1.5x fewer
// matrix instructions than that kernel, 3.375x fewer than the direct one, all fp32 (coefficients +-1 and 1/2; measured error against fp64
// 2.8e-7 relative L2 at 32 channels, the (y, x) form 2.1e-7, the direct form 1.2e-7 - same 2e-6 test bound).
//
//   tile 4 x 8 x 8 voxels = 2 x 4 x 4 blocks = ONE 32-row MFMA tile per point; 32 output channels per work item; 8-channel chunks.
//   512 threads = 8 waves, ONE workgroup per CU: wave (py, pzh) owns the eight points (pz in {2 pzh, 2 pzh + 1}, py, px 0..3) - eight
//   32 x 32 accumulator tiles, the register budget of the (y, x) kernel's wave.
//   LDS image of a chunk = the (y, x) kernel's: the halo x-transformed while it is staged (one item per thread: four taps -> four px rows,
//   six raw z planes).  The matrix loop reads four rows per point step - rows ta / tb of planes za / zb - and forms the y AND z
//   combinations in registers (six two-wide fmas); point steps run in pairs whose MFMAs alternate between two accumulator tiles.
//   Staging runs two chunks ahead: the taps a thread requests during chunk c stay in registers for a whole chunk (2000 matrix clocks, more
//   than a trip to HBM) and are transformed into the image of chunk c + 2 at the start of chunk c + 1; two images, one barrier per chunk.
//   Weights global -> registers through a ring of four point rows (a row is re-loaded for the point four steps on as soon as it is used).
//   Epilogue: x inverse transform in registers, the z pair of a wave combined in registers, y (four waves) and the two z halves summed
//   through a 64 KB exchange buffer, one output z parity at a time; stores, bias, BatchNorm partials, the eval-mode store and the fused
//   BatchNorm-backward sums (BNR) as in the (y, x) kernel's fast path.
//
// Whole tiles only (D % 4 == 0, H % 8 == 0, W % 8 == 0), channels-last 16-byte aligned operands, Cin % 8 == 0, Cout % 32 == 0, at least 256 work
// items: every such layer from 16 reduction channels up (pulpo_conv3d_k3_algo = 3; 32 until the end of round 4 - with the leaner tile head and epilogue
// two chunks per tile pay too: 16 -> 96 at 80^3 0.242 -> 0.213 ms; PULPO_CONV_WINO3=0 / PULPO_CONV_WINO3_MINK=<k> move the policy).
// What was measured while it was built: an image transformed along x AND z at staging time (two rows per step instead of
// four) multiplies faster (bare loop 0.31 against 0.34 ms at 64 -> 64 / 80^3) but its staging - eight loads, 32 combinations and a second
// item for a quarter of the threads - cost 20 - 24 % against 8 - 10 % here; wave-uniform branches around staging loads cost 15 % (every
// vmcnt behind them becomes a worst-case guess); the plane offset must ride in the VECTOR offset of a buffer load (the bounds check that
// zeroes out-of-volume planes ignores the scalar offset).
#include "conv_shared.h"
#include "wino3_pack.h"
#include <stdlib.h>

#ifndef PULPO_W3_STAMPS
#define PULPO_W3_STAMPS 0        // diagnostic build (scripts/stamps_w3.py): s_memtime stamps of one tile's phases per wave, kept in registers until the tile's end
#endif
#if PULPO_W3_STAMPS
__device__ unsigned g_w3_stamps[256 * 8 * 32];
PULPO_API int pulpo_debug_read_stamps_w3(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_w3_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#define W3_CLK() ((unsigned)__builtin_amdgcn_s_memtime())                        // (waits for lgkmcnt(0) where it is consumed: the stamps sit where that wait is due anyway)
// scalar stamps (SGPRs): T = now; the phase accumulators of the tile are wave-uniform sums
#define W3_NOW(var) do { __builtin_amdgcn_sched_barrier(0); var = W3_CLK(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define W3_STAMP(i) do {} while (0)
#else
#define W3_STAMP(i) do {} while (0)
#endif

#ifndef PULPO_W3_PK
#define PULPO_W3_PK 1        // the y / z combinations in two-wide vector arithmetic (v_pk_fma_f32); 0: scalar v_fma_f32 - measured 3-4 % slower here (64-clock fp32 MFMAs leave room)
#endif
#ifndef PULPO_ABL
#define PULPO_ABL 0          // diagnostic builds (scripts/ablate.py): timings only, results are garbage.  Bits: 1 no epilogue, 2 no halo staging, 64 every tap from one 64 KB window (cache hits),
#endif                       // 4 no weight re-loads, 8 no chunk barrier, 32 no operand-row reads inside the loop

namespace {

using namespace pulpo_conv;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int Q_CH = 8, Q_NT = 32;
constexpr int Q_PL = HY * 4;                     // rows of one px slice of a plane: (hy, x-pair)
constexpr int Q_PLROWS = 4 * Q_PL + 2;           // rows per hz plane: the two z blocks of an MFMA row tile lie TWO planes apart = 4 rows mod 16
constexpr int Q_QROWS = 6 * Q_PLROWS;            // rows per channel quad (x 4 dwords = 16 mod 32: the two quads of a ds_write_b128 group use different banks)
constexpr int Q_IMG = 2 * Q_QROWS * 4;           // floats of one image
static_assert((2 * Q_PLROWS) % 16 == 4 && (Q_QROWS * 4) % 32 == 16, "bank layout");
constexpr int Q_NITEM = 6 * HY * 4 * 2;          // staging items of a chunk: (hz, hy, x-pair, channel quad) = 480, one per thread
constexpr int Q_TAB = 3 * 512;                   // per-channel table [3][ncot * 32]: up to 512 output channels
constexpr int Q_R = 8 * 2 * 16 * 64;             // floats of the exchange buffer of one output z parity: [wave][ox][r][lane]
constexpr int Q_RED = 8 * 2 * Q_NT;              // statistics rows of the eight waves
constexpr int Q_PART = 8 * 64 * 8;               // a wave's per-lane statistics before its own (in-order, barrier-free) reduction: [wave][lane][8]
constexpr size_t Q_LDS = (size_t)(2 * Q_IMG + Q_TAB + Q_R + Q_RED + Q_PART) * sizeof(float);
static_assert(Q_LDS <= 160 * 1024, "one workgroup per CU");

#ifndef PULPO_W3_YEARLY
#define PULPO_W3_YEARLY 1          // BNR epilogue: y requested early (see the epilogue)
#endif
#ifndef PULPO_W3_SKEW
#define PULPO_W3_SKEW 1          // 1: the two waves of a SIMD (w and w + 4) do their halo staging in DIFFERENT pairs of a chunk (see the kernel below)
#endif

// SP: the pair of a chunk behind whose first MFMAs a wave transforms and stores its staging item (SP) and requests the next taps (SP + 1)
template <bool BNR, int SP>
__device__ __forceinline__ void wino3_body(const ConvArgs& a) {
    constexpr int CH = Q_CH, NT = Q_NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const tab = smem + 2 * Q_IMG;                // [3][ctab]
    float* const R = tab + Q_TAB;                       // exchange buffer
    float* const red = R + Q_R;                         // [8 waves][2][NT]
    float* const part = red + Q_RED;                    // [8 waves][64 lanes][8]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kk = lane >> 5;
    const int py = wave & 3, pzh = wave >> 2;
    const int nchunk = a.Cin / CH;                      // (host: >= 2)
    const int nwork = a.B * a.ntz * a.nty * a.ntx * a.ncot;
    const int nwg = gridDim.x;
    const unsigned ps_bytes = (unsigned)a.in_ps * 4u;
    const unsigned kb_bytes = (unsigned)a.in_kb * 4u;        // bytes between consecutive chunks of a voxel (32: channels-last)

    // ---- the staging item of this thread (the (y, x) kernel's): (hz, hy, x-pair xb, channel quad q) -> the four x-transformed rows px = 0..3
    unsigned roff;                                      // byte offset of tap 0 relative to the tile's halo origin
    int lofs;                                           // float offset of the item's px = 0 row in an image
    {
        const int q = tid & 1, rb = tid >> 1;
        const int xb = rb & 3, hrow = rb >> 2;
        const int hz = hrow / HY, hy = hrow - hz * HY;
        roff = ((unsigned)((hz * a.H + hy) * a.W + 2 * xb) * (unsigned)a.in_ps + 4u * q) * 4u;
        lofs = (q * Q_QROWS + hz * Q_PLROWS + hy * 4 + xb) * 4;
    }
    const bool item = tid < Q_NITEM;
    // combination tables of B^T (rows t_a + s * t_b): point 0: (0, 2, -), 1: (1, 2, +), 2: (2, 1, -), 3: (1, 3, -)
    auto tab_a = [](int p) { return p == 0 ? 0 : p == 2 ? 2 : 1; };
    auto tab_b = [](int p) { return p == 2 ? 1 : p == 3 ? 3 : 2; };
    auto tab_s = [](int p) { return p == 1 ? 1.f : -1.f; };
    // y combination of this wave's point row, z combination of its two point planes pz = 2 pzh + (0, 1): per-wave scalars
    const int ta = tab_a(py), tb = tab_b(py);
    const float sa = tab_s(py);
    // the wave's two z points share THREE planes U, V, W of a z block: pzh = 0: pz 0 = P0 - P2, pz 1 = P1 + P2 (U, V, W = P0, P1, P2);
    // pzh = 1: pz 2 = P2 - P1, pz 3 = P1 - P3 = -(P3 - P1) (U, V, W = P2, P3, P1).  Both waves form  v0 = U - W  and  v1 = V + bw * W  (bw = +1 / -1):
    // the second accumulator of a pzh = 1 wave therefore holds MINUS its point, which the epilogue's z inverse transform takes into account.
    const int zu = (pzh == 0 ? 0 : 2) * Q_PLROWS * 4, zv = (pzh == 0 ? 1 : 3) * Q_PLROWS * 4, zw = (pzh == 0 ? 2 : 1) * Q_PLROWS * 4;
    const float bw = pzh == 0 ? 1.f : -1.f;
    // MFMA row i = block (zb = i >> 4, yb = (i >> 2) & 3, xb = i & 3): halo plane 2 zb + (0..3), halo row 2 yb + (0..3)
    const int lrow = (i >> 4) * 2 * Q_PLROWS + ((i >> 2) & 3) * 8 + (i & 3);
    const int pa_off = (kk * Q_QROWS + lrow + ta * 4) * 4;
    const int pb_off = (kk * Q_QROWS + lrow + tb * 4) * 4;

    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, -1, 0x00020000);
    // num_records of the operand's descriptor.  The range check of a raw buffer load is  voffset >= num_records - soffset  (the chunk's offset rides in
    // soffset): the extent must therefore reach the LAST chunk's voxels - in the channel-blocked layout (in_kb = V * 8) that is the whole operand
    const int in_bytes = (int)((((long)a.Cin / CH - 1) * (a.in_kb == CH ? 0 : a.in_kb) + (long)a.D * a.H * a.W * a.in_ps) * 4);
    struct Tile {
        int tile_lin, b, z0, y0, x0, co0;
        unsigned wbase;              // byte offset of (chunk 0, point (2 pzh, py, 0), cout co0) in the packed weights
    };
    auto describe = [&](int work) {
        Tile t;
        const int cot = work % a.ncot;
        int q = work / a.ncot;
        int tx_, ty_, tz_;
        if (a.tile_order == 1) {                        // blocks of 4 x 4 x 4 tiles, as in the (y, x) kernel
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
        t.tile_lin = ((t.b * a.ntz + tz_) * a.nty + ty_) * a.ntx + tx_;
        t.z0 = tz_ * 4; t.y0 = ty_ * TY; t.x0 = tx_ * TX;
        t.co0 = cot * NT;
        t.wbase = (unsigned)(((2 * pzh) * 16 + py * 4) * a.NPad + t.co0) * (CH * 4u);
        return t;
    };
    // byte offsets of this thread's four tap loads of a tile's halo at chunk 0, relative to the batch element; OOB (beyond num_records: the load
    // returns zeros) where the tap lies outside the volume or the item does not exist.  The chunk's channel offset rides in the scalar offset.
    unsigned hoff[4];
    auto halo_offsets = [&](const Tile& t) {
        const unsigned origin = (unsigned)(((t.z0 - 1) * a.H + (t.y0 - 1)) * a.W + (t.x0 - 1)) * ps_bytes;      // modulo 2^32: may be "negative"
        const int rb = tid >> 1;
        const int xb = rb & 3, hrow = rb >> 2;
        const int hz = hrow / HY, hy = hrow - hz * HY;
        const bool rowok = item && (unsigned)(t.z0 - 1 + hz) < (unsigned)a.D && (unsigned)(t.y0 - 1 + hy) < (unsigned)a.H;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            hoff[tt] = (rowok && (unsigned)(t.x0 - 1 + 2 * xb + tt) < (unsigned)a.W) ? ((PULPO_ABL & 64) ? ((origin + roff + tt * ps_bytes) & 0xFFE0u) : origin + roff + tt * ps_bytes) : OOB;
    };
    auto in_rsrc = [&](int b) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in + (long)b * a.in_bs), 0, in_bytes, 0x00020000);
    };
    float4 raw[4];                                      // the four x taps of the staging item: loaded one chunk before they are transformed
    auto load_raw = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned c0_bytes, int tt) {
        raw[tt] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)hoff[tt], (int)c0_bytes, 0));
    };
    auto store_item = [&](float* img) {
        if (item) {
            float* o = img + lofs;
            const float4 d0 = raw[0], d1 = raw[1], d2 = raw[2], d3 = raw[3];
            *reinterpret_cast<float4*>(o) = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
            *reinterpret_cast<float4*>(o + Q_PL * 4) = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
            *reinterpret_cast<float4*>(o + 2 * Q_PL * 4) = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
            *reinterpret_cast<float4*>(o + 3 * Q_PL * 4) = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
        }
    };
    const unsigned w_chunk_stride = 64u * CH * a.NPad * 4u;     // bytes between consecutive chunks
    const unsigned w_pt_stride = (unsigned)CH * a.NPad * 4u;    // bytes between consecutive points (px fastest, then py, then pz)
    const int wl_off = (i * CH + 4 * kk) * 4;                   // this lane's 16 bytes inside a point's [32 n][8 k] piece
    auto load_w = [&](unsigned wofs, int s) {                   // point step s = pz_local * 4 + px
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wl_off, (int)(wofs + ((s >> 2) * 16 + (s & 3)) * w_pt_stride), 0));
    };

    // ---- per-channel constants of the epilogue, once per workgroup
    const int ctab = a.ncot * NT;
    for (int c = tid; c < ctab; c += 512) {
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
    int work = pulpo::xcd_remap(blockIdx.x, nwg);
    Tile cur = describe(work);
    int cb = 0;                                         // image being read
    // ---- prologue: chunk 0 of the first tile staged, chunk 1 requested; the first weight rows
    float4 wr[2][2];                                    // ring of two pairs of point rows: pair pp uses wr[pp & 1], which then takes pair pp + 2
    {
        const __amdgpu_buffer_rsrc_t rs0 = in_rsrc(cur.b);
        halo_offsets(cur);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) load_raw(rs0, 0u, tt);
        store_item(smem);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) load_raw(rs0, kb_bytes, tt);
    }
    // (the weight rows are requested AFTER the taps, as in the loop: the compiler's vmcnt bookkeeping merges this path with the loop's back
    //  edge, and with the taps as the youngest loads here every chunk's transform waited for vmcnt(0) - the weight rows of the pair after next)
    __builtin_amdgcn_sched_barrier(0);                  // (... and the scheduler would hoist them back in front)
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) { wr[pp][0] = load_w(cur.wbase, pp); wr[pp][1] = load_w(cur.wbase, 4 + pp); }
    __syncthreads();

#ifndef PULPO_W3_SETPRIO
#define PULPO_W3_SETPRIO 2       // 0: equal priorities, 1: waves 4-7 raised for good (no gain), 2: the partners of a SIMD alternate per pair (+1.5 %)
#endif
    if (PULPO_W3_SETPRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);       // (the second-dispatched half loses every issue arbitration otherwise: MI355X_MICROARCH.md)
    float4 ra[3], rb[3];                                // the operand rows of a PAIR of point steps (pz local 0 / 1 at one px): rows ta / tb of planes U, V, W

    // The tile's statistics: the waves' partial sums wait in `red` and are added behind the NEXT barrier every wave passes anyway (the end of the
    // next tile's first chunk, or the one after the loop): the epilogue needs no barrier of its own at its end - the exchange buffer and `red`
    // are not written again before the next tile's epilogue, nchunk barriers away.
    int pend_tile = -1, pend_co0 = 0;
    auto flush_stats = [&]() {
        if (a.stats != nullptr && pend_tile >= 0 && tid < 2 * NT) {
            const int which = tid / NT, c = tid - which * NT;
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) tot += red[(w * 2 + which) * NT + c];
            if (pend_co0 + c < a.Cout) a.stats[((long)pend_tile * 2 + which) * a.Cout + pend_co0 + c] = tot;
        }
        pend_tile = -1;
    };
#if PULPO_W3_STAMPS
    int tile_no = 0;
    unsigned w3_tile0 = 0, w3_rows = 0, w3_mfma = 0, w3_bar = 0;
#endif
    for (;;) {
        // the next tile is known from the start (its description inside the chunk loop, under `chunk + 2 == nchunk`, was if-converted by the
        // compiler: five integer divisions' worth of scalar instructions and a dozen spilled-register reloads in EVERY chunk)
        const int next_work = work + nwg;
        const bool has_next = next_work < nwork;
        const Tile nxt = has_next ? describe(next_work) : cur;

        f32x16 acc[2][4];                               // [pz local][px]
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][x][r] = 0.f;

        unsigned wcur = cur.wbase;                      // weights of the chunk being multiplied
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            // Staging runs TWO chunks ahead of the matrix loop: step 0 transforms the taps requested during the previous chunk into the image
            // of chunk + 1 and steps 1 - 2 request those of chunk + 2 (or of the next tile's chunk 0 / 1), which the registers hold for a whole
            // chunk - 2000 matrix clocks, more than a trip to HBM.
            const bool ahead = chunk + 2 >= nchunk;     // (after the last tile: the tile's own first chunks again, into images nobody reads)
            const unsigned st_c0 = (unsigned)(ahead ? chunk + 2 - nchunk : chunk + 2) * kb_bytes;
            const __amdgpu_buffer_rsrc_t st_rs = in_rsrc(ahead ? nxt.b : cur.b);
            const float* img_r = smem + cb * Q_IMG;
            float* img_w = smem + (cb ^ 1) * Q_IMG;
            const float* pa = img_r + pa_off;
            const float* pb = img_r + pb_off;
            auto fetch_a = [&](int px) {
                const int o = px * Q_PL * 4;
                ra[0] = *reinterpret_cast<const float4*>(pa + zu + o); rb[0] = *reinterpret_cast<const float4*>(pb + zu + o);
                ra[1] = *reinterpret_cast<const float4*>(pa + zv + o); rb[1] = *reinterpret_cast<const float4*>(pb + zv + o);
                ra[2] = *reinterpret_cast<const float4*>(pa + zw + o); rb[2] = *reinterpret_cast<const float4*>(pb + zw + o);
            };
            const bool last_chunk = chunk + 1 == nchunk;
            const unsigned wnext = last_chunk ? (has_next ? nxt.wbase : cur.wbase) : wcur + w_chunk_stride;
            // Point steps run in PAIRS - the wave's two z points at one px: they share the three planes U, V, W (six rows instead of eight), and
            // their MFMAs alternate between two accumulator tiles, so that no matrix instruction waits for the result of the one in front of
            // it.  The rows of the next pair are requested as soon as the combinations have been formed.
            // y combination of each plane (Y = A + sa B), then the two z combinations; two-wide vector arithmetic
            auto combine = [&](float (&av0)[4], float (&av1)[4]) {
#if PULPO_W3_PK
                const f32x2 sav = {sa, sa}, m1 = {-1.f, -1.f}, bwv = {bw, bw};
                f32x2 ylo[3], yhi[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    ylo[p] = __builtin_elementwise_fma(sav, f32x2{rb[p].x, rb[p].y}, f32x2{ra[p].x, ra[p].y});
                    yhi[p] = __builtin_elementwise_fma(sav, f32x2{rb[p].z, rb[p].w}, f32x2{ra[p].z, ra[p].w});
                }
                const f32x2 lo0 = __builtin_elementwise_fma(m1, ylo[2], ylo[0]), hi0 = __builtin_elementwise_fma(m1, yhi[2], yhi[0]);
                const f32x2 lo1 = __builtin_elementwise_fma(bwv, ylo[2], ylo[1]), hi1 = __builtin_elementwise_fma(bwv, yhi[2], yhi[1]);
                av0[0] = lo0.x; av0[1] = lo0.y; av0[2] = hi0.x; av0[3] = hi0.y;
                av1[0] = lo1.x; av1[1] = lo1.y; av1[2] = hi1.x; av1[3] = hi1.y;
#else
                // scalar fmas (the guide prices v_pk_fma_f32 above two v_fma_f32 beside 32-clock bf16 MFMAs; beside these 64-clock fp32 MFMAs the packed form won)
                float y[3][4];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    y[p][0] = fmaf(sa, rb[p].x, ra[p].x); y[p][1] = fmaf(sa, rb[p].y, ra[p].y);
                    y[p][2] = fmaf(sa, rb[p].z, ra[p].z); y[p][3] = fmaf(sa, rb[p].w, ra[p].w);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    av0[k] = y[0][k] - y[2][k];
                    av1[k] = fmaf(bw, y[2][k], y[1][k]);
                }
#endif
            };
#ifndef PULPO_W3_SPREAD
#define PULPO_W3_SPREAD 0        // 1: the next pair's combinations spread over the gaps behind the fourth, sixth and eighth MFMA instead of one block behind the fourth
#endif
            f32x2 sy_lo[3], sy_hi[3];
            auto comb_y = [&](int p) {
                const f32x2 sav = {sa, sa};
                sy_lo[p] = __builtin_elementwise_fma(sav, f32x2{rb[p].x, rb[p].y}, f32x2{ra[p].x, ra[p].y});
                sy_hi[p] = __builtin_elementwise_fma(sav, f32x2{rb[p].z, rb[p].w}, f32x2{ra[p].z, ra[p].w});
            };
            auto comb_z = [&](float (&av0)[4], float (&av1)[4]) {
                const f32x2 m1 = {-1.f, -1.f}, bwv = {bw, bw};
                const f32x2 lo0 = __builtin_elementwise_fma(m1, sy_lo[2], sy_lo[0]), hi0 = __builtin_elementwise_fma(m1, sy_hi[2], sy_hi[0]);
                const f32x2 lo1 = __builtin_elementwise_fma(bwv, sy_lo[2], sy_lo[1]), hi1 = __builtin_elementwise_fma(bwv, sy_hi[2], sy_hi[1]);
                av0[0] = lo0.x; av0[1] = lo0.y; av0[2] = hi0.x; av0[3] = hi0.y;
                av1[0] = lo1.x; av1[1] = lo1.y; av1[2] = hi1.x; av1[3] = hi1.y;
            };
            // Software pipeline over the chunk's four pairs: the operand rows of pair pp + 1 are requested in front of pair pp's MFMAs and
            // COMBINED between them (behind the fourth of the eight), the rows of pair pp + 2 requested right after - a wave never stands in a
            // vector-only phase while it has matrix instructions to issue, except in front of a chunk's first pair.
            float avn0[4], avn1[4];                     // the operands of the pair in flight / of the next pair
#if PULPO_W3_STAMPS
            unsigned t_a, t_b, t_c, t_d;
            W3_NOW(t_a);
            if (chunk == 0) { w3_tile0 = t_a; w3_rows = 0; w3_mfma = 0; w3_bar = 0; }
#endif
            if (!(PULPO_ABL & 32)) fetch_a(0);
            combine(avn0, avn1);
#if PULPO_W3_STAMPS
            W3_NOW(t_b);
#endif
            if (!(PULPO_ABL & 32)) fetch_a(1);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int s0 = pp, s1 = 4 + pp;         // point steps (pz local 0, px = pp) and (pz local 1, px = pp)
                // PULPO_W3_SETPRIO = 2: the two waves of a SIMD (w and w + 4: pzh 0 / 1) take the higher issue priority in ALTERNATING pairs.  At
                // equal priority the older wave wins every arbitration: it runs its 32 MFMAs of a chunk in 3 350 clocks, the younger one gets
                // the leftover slots and finishes alone 1 300 clocks later (scripts/stamps_w3.py), with the pipe idle in its gaps.
                if (PULPO_W3_SETPRIO == 2) { if (((pp + pzh) & 1) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
                const float av0[4] = {avn0[0], avn0[1], avn0[2], avn0[3]}, av1[4] = {avn1[0], avn1[1], avn1[2], avn1[3]};
                const float wv0[4] = {wr[pp & 1][0].x, wr[pp & 1][0].y, wr[pp & 1][0].z, wr[pp & 1][0].w};
                const float wv1[4] = {wr[pp & 1][1].x, wr[pp & 1][1].y, wr[pp & 1][1].z, wr[pp & 1][1].w};
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    acc[s0 >> 2][s0 & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[s2], wv0[s2], acc[s0 >> 2][s0 & 3], 0, 0, 0);
                    acc[s1 >> 2][s1 & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[s2], wv1[s2], acc[s1 >> 2][s1 & 3], 0, 0, 0);
                    if (s2 == 0) {                      // behind the pair's first MFMAs: the staging work of the pair
                        __builtin_amdgcn_sched_barrier(0);
                        if (!(PULPO_ABL & 2)) {
                            if (pp == SP) {
                                store_item(img_w);
                                if (chunk + 2 == nchunk) halo_offsets(has_next ? nxt : cur);
                            }
                            if (pp == SP + 1) { load_raw(st_rs, st_c0, 0); load_raw(st_rs, st_c0, 1); load_raw(st_rs, st_c0, 2); load_raw(st_rs, st_c0, 3); }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (!PULPO_W3_SPREAD && s2 == 1 && pp + 1 < 4) {        // behind the fourth MFMA: the next pair's combinations, then the request for the pair after it
                        __builtin_amdgcn_sched_barrier(0);
                        combine(avn0, avn1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (!(PULPO_ABL & 32) && pp + 2 < 4) fetch_a(pp + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (PULPO_W3_SPREAD && pp + 1 < 4 && s2 >= 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (s2 == 1) { comb_y(0); comb_y(1); }
                        if (s2 == 2) { comb_y(2); comb_z(avn0, avn1); }
                        if (s2 == 3 && !(PULPO_ABL & 32) && pp + 2 < 4) fetch_a(pp + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // the pair's weights are consumed: a ring of two pairs - the rows take the points of the pair two on (the same chunk's, or the
                // next chunk's first two)
                if (!(PULPO_ABL & 4)) {
                    wr[pp & 1][0] = pp < 2 ? load_w(wcur, s0 + 2) : load_w(wnext, s0 - 2);
                    wr[pp & 1][1] = pp < 2 ? load_w(wcur, s1 + 2) : load_w(wnext, s1 - 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            wcur += w_chunk_stride;
#if PULPO_W3_STAMPS
            W3_NOW(t_c);
#endif
            if (!(PULPO_ABL & 8)) __syncthreads();      // image cb ^ 1 complete and visible; every wave has left image cb
#if PULPO_W3_STAMPS
            W3_NOW(t_d);
            w3_rows += t_b - t_a; w3_mfma += t_c - t_b; w3_bar += t_d - t_c;
#endif
            cb ^= 1;
            if (chunk == 0) flush_stats();              // (the previous tile's, see above)
        }

        // ---- epilogue
        float* out_b = a.out + (long)cur.b * a.out_bs;
        const int z0 = cur.z0, y0 = cur.y0, x0 = cur.x0, co0 = cur.co0;
        int elane = lane;
        asm volatile("" : "+v"(elane));
        const int q = elane & 7, kh = (elane >> 3) & 1, g = elane >> 4;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 s4 = zero4, q4 = zero4;
        const bool fuse = !BNR && a.coef != nullptr;
        const float4 b4 = BNR ? zero4 : *reinterpret_cast<const float4*>(tab + co0 + 4 * q);
        const float4 bm4 = *reinterpret_cast<const float4*>(tab + co0 + 4 * q);
        const float4 sc4 = *reinterpret_cast<const float4*>(tab + ctab + co0 + 4 * q);
        const float4 sh4 = *reinterpret_cast<const float4*>(tab + 2 * ctab + co0 + 4 * q);
        const float* bn_b = BNR ? a.bn_y + (long)cur.b * a.bn_y_bs + co0 + 4 * q : nullptr;
        // this lane finishes (ox, r) = combo of the exchange buffer, rows of half kh, channels 4 q .. 4 q + 3
        const int combo = wave * 4 + g;
        const int ox = combo >> 4, rr = combo & 15;
        const int row = (rr & 3) + 8 * (rr >> 2) + 4 * kh;          // = MFMA row = block (zb, yb, xb)
        const int vzb = row >> 4, vyb = (row >> 2) & 3, vxb = row & 3;
#if PULPO_W3_STAMPS
        unsigned t_e0, t_e1 = 0, t_e2 = 0, t_e3;
        W3_NOW(t_e0);
#endif
        // x inverse transform (4 px -> 2 ox), once and in place: acc[p][0] <- out x0 = q0 + q1 + q2, acc[p][1] <- out x1 = q1 - q2 - q3
        // BNR: the pre-norm values of this lane's two output voxels of a parity are requested EARLY - parity 0 in front of the x inverse transform,
        // parity 1 as soon as parity 0's have been used - so that a miss (y is read once per step: HBM) lands under the transform and the exchange
        // instead of in front of the sums that need it (PULPO_W3_YEARLY=0: requested right in front of the exchange barrier of their parity)
        float4 yv0 = zero4, yv1 = zero4;
        const bool qok = co0 + 4 * q < a.Cout;         // (a partly empty cout tile: channel quads beyond the tensor are neither read nor stored)
        auto load_y = [&](int oz_) {
            if (BNR && qok) {
                const long vox_ = (long)((z0 + 2 * vzb + oz_) * a.H + y0 + 2 * vyb) * a.W + x0 + 2 * vxb + ox;
                yv0 = *reinterpret_cast<const float4*>(bn_b + vox_ * a.bn_y_ps);
                yv1 = *reinterpret_cast<const float4*>(bn_b + (vox_ + a.W) * a.bn_y_ps);
            }
        };
        if (PULPO_W3_YEARLY) { load_y(0); __builtin_amdgcn_sched_barrier(0); }
        // a - b below is fma(m1, b, a) with m1 = -1 the compiler cannot see through: exact (the product is), and it stays ONE two-wide instruction
        // (v_pk_fma_f32) - a two-wide subtraction is expanded into two scalar ones by the backend
        f32x2 m1 = {-1.f, -1.f};
        asm volatile("" : "+s"(m1));
        // (two-wide: the accumulator registers of rows r, r + 1 are an aligned pair - v_pk_add_f32 with the association of the scalar form,
        //  (a0 + a1) + a2 and (a1 - a2) - a3: 64 instead of 128 vector instructions per wave, which the SIMD's two waves issue at the same time)
        if (!(PULPO_ABL & 1)) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 a0 = {acc[p][0][r], acc[p][0][r + 1]}, a1 = {acc[p][1][r], acc[p][1][r + 1]};
                    const f32x2 a2 = {acc[p][2][r], acc[p][2][r + 1]}, a3 = {acc[p][3][r], acc[p][3][r + 1]};
                    const f32x2 o0 = (a0 + a1) + a2, o1 = __builtin_elementwise_fma(m1, a3, __builtin_elementwise_fma(m1, a2, a1));
                    acc[p][0][r] = o0.x; acc[p][0][r + 1] = o0.y;
                    acc[p][1][r] = o1.x; acc[p][1][r + 1] = o1.y;
                }
        }
#pragma unroll
        for (int oz = 0; oz < ((PULPO_ABL & 1) ? 0 : 2); ++oz) {
            if (oz > 0) __syncthreads();                // every wave has left the exchange buffer (previous parity)
            // this wave's share of the z inverse transform: out z0 = q0 + q1 + q2, out z1 = q1 - q2 - q3; wave pzh = 0 holds (q0, q1), pzh = 1
            // holds (q2, -q3) (the second accumulator of those waves holds MINUS its point, see the matrix loop).  A branch per wave (pzh is
            // wave-uniform) instead of two selects per value.
            float* const r0 = R + ((wave * 2 + 0) * 16) * 64 + elane;
            float* const r1 = R + ((wave * 2 + 1) * 16) * 64 + elane;
            if (pzh == 0) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 u0 = f32x2{acc[0][0][r], acc[0][0][r + 1]} + f32x2{acc[1][0][r], acc[1][0][r + 1]};
                    const f32x2 u1 = f32x2{acc[0][1][r], acc[0][1][r + 1]} + f32x2{acc[1][1][r], acc[1][1][r + 1]};
                    r0[r * 64] = oz == 0 ? u0.x : acc[1][0][r]; r0[(r + 1) * 64] = oz == 0 ? u0.y : acc[1][0][r + 1];
                    r1[r * 64] = oz == 0 ? u1.x : acc[1][1][r]; r1[(r + 1) * 64] = oz == 0 ? u1.y : acc[1][1][r + 1];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 u0 = __builtin_elementwise_fma(m1, f32x2{acc[0][0][r], acc[0][0][r + 1]}, f32x2{acc[1][0][r], acc[1][0][r + 1]});
                    const f32x2 u1 = __builtin_elementwise_fma(m1, f32x2{acc[0][1][r], acc[0][1][r + 1]}, f32x2{acc[1][1][r], acc[1][1][r + 1]});
                    r0[r * 64] = oz == 0 ? acc[0][0][r] : u0.x; r0[(r + 1) * 64] = oz == 0 ? acc[0][0][r + 1] : u0.y;
                    r1[r * 64] = oz == 0 ? acc[0][1][r] : u1.x; r1[(r + 1) * 64] = oz == 0 ? acc[0][1][r + 1] : u1.y;
                }
            }
            const int gz = z0 + 2 * vzb + oz, gy = y0 + 2 * vyb, gx = x0 + 2 * vxb + ox;
            const long vox = (long)(gz * a.H + gy) * a.W + gx;
            if (BNR && !PULPO_W3_YEARLY && qok) {
                yv0 = *reinterpret_cast<const float4*>(bn_b + vox * a.bn_y_ps);
                yv1 = *reinterpret_cast<const float4*>(bn_b + (vox + a.W) * a.bn_y_ps);
            }
            __syncthreads();
#if PULPO_W3_STAMPS
            if (oz == 0) W3_NOW(t_e1);
#endif
            float4 t[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {               // point row p: the two z halves summed
                const float4 u0 = *reinterpret_cast<const float4*>(R + (((0 * 4 + p) * 2 + ox) * 16 + rr) * 64 + kh * 32 + 4 * q);
                const float4 u1 = *reinterpret_cast<const float4*>(R + (((1 * 4 + p) * 2 + ox) * 16 + rr) * 64 + kh * 32 + 4 * q);
                t[p] = make_float4(u0.x + u1.x, u0.y + u1.y, u0.z + u1.z, u0.w + u1.w);
            }
            float4 v0 = make_float4(t[0].x + t[1].x + t[2].x + b4.x, t[0].y + t[1].y + t[2].y + b4.y, t[0].z + t[1].z + t[2].z + b4.z, t[0].w + t[1].w + t[2].w + b4.w);
            float4 v1 = make_float4(t[1].x - t[2].x - t[3].x + b4.x, t[1].y - t[2].y - t[3].y + b4.y, t[1].z - t[2].z - t[3].z + b4.z, t[1].w - t[2].w - t[3].w + b4.w);
            if (BNR) {
                auto red1 = [&](float dzv, float yy, float sc, float sh, float m32, float& s_, float& q_) {
                    const float bn = yy * sc + sh;
                    const float d = bn > 0.f ? dzv : dzv * a.slope;
                    s_ += d;
                    q_ = fmaf(d, yy - m32, q_);
                };
                red1(v0.x, yv0.x, sc4.x, sh4.x, bm4.x, s4.x, q4.x); red1(v0.y, yv0.y, sc4.y, sh4.y, bm4.y, s4.y, q4.y);
                red1(v0.z, yv0.z, sc4.z, sh4.z, bm4.z, s4.z, q4.z); red1(v0.w, yv0.w, sc4.w, sh4.w, bm4.w, s4.w, q4.w);
                red1(v1.x, yv1.x, sc4.x, sh4.x, bm4.x, s4.x, q4.x); red1(v1.y, yv1.y, sc4.y, sh4.y, bm4.y, s4.y, q4.y);
                red1(v1.z, yv1.z, sc4.z, sh4.z, bm4.z, s4.z, q4.z); red1(v1.w, yv1.w, sc4.w, sh4.w, bm4.w, s4.w, q4.w);
                if (PULPO_W3_YEARLY && oz == 0) { __builtin_amdgcn_sched_barrier(0); load_y(1); __builtin_amdgcn_sched_barrier(0); }
            } else {
                s4.x += v0.x + v1.x; s4.y += v0.y + v1.y; s4.z += v0.z + v1.z; s4.w += v0.w + v1.w;
                q4.x += v0.x * v0.x + v1.x * v1.x; q4.y += v0.y * v0.y + v1.y * v1.y; q4.z += v0.z * v0.z + v1.z * v1.z; q4.w += v0.w * v0.w + v1.w * v1.w;
            }
            if (fuse) {
                auto act = [&](float v, float sc, float sh) { const float tt = v * sc + sh; return tt > 0.f ? tt : tt * a.slope; };
                v0 = make_float4(act(v0.x, sc4.x, sh4.x), act(v0.y, sc4.y, sh4.y), act(v0.z, sc4.z, sh4.z), act(v0.w, sc4.w, sh4.w));
                v1 = make_float4(act(v1.x, sc4.x, sh4.x), act(v1.y, sc4.y, sh4.y), act(v1.z, sc4.z, sh4.z), act(v1.w, sc4.w, sh4.w));
            }
            float* obase = out_b + (long)((co0 + 4 * q) >> 3) * a.out_kb + ((4 * q) & 7);      // (out_kb = 8: channels-last, co0 + 4 q)
            if (qok) {
                *reinterpret_cast<float4*>(obase + vox * a.out_ps) = v0;
                *reinterpret_cast<float4*>(obase + (vox + a.W) * a.out_ps) = v1;
            }
#if PULPO_W3_STAMPS
            if (oz == 0) W3_NOW(t_e2);
#endif
        }
        // per-tile BatchNorm partial sums: over the lanes that hold the same channels, then over the eight waves
        // (through the wave's own LDS rows - DS operations of one wave execute in order, no barrier - instead of three rounds of cross-lane
        //  shuffles: lane l then owns ONE of the wave's 64 sums (which = l / 32, channel l % 32) and adds the eight lanes that hold its channel quad)
        if (a.stats != nullptr) {
            float* pw_ = part + (wave * 64 + elane) * 8;
            *reinterpret_cast<float4*>(pw_) = s4;
            *reinterpret_cast<float4*>(pw_ + 4) = q4;
            const int c_ = elane & 31, wh_ = elane >> 5;
            const float* pr_ = part + (wave * 64 + (c_ >> 2)) * 8 + wh_ * 4 + (c_ & 3);
            float t_ = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t_ += pr_[j * 64];          // lanes q + 8 j
            red[(wave * 2 + wh_) * NT + c_] = t_;
        }
        pend_tile = cur.tile_lin; pend_co0 = co0;
#if PULPO_W3_STAMPS
        W3_NOW(t_e3);
        if (tile_no == 1 && lane == 0) {
            unsigned* o_ = g_w3_stamps + (blockIdx.x * 8 + wave) * 32;
            o_[0] = w3_rows; o_[1] = w3_mfma; o_[2] = w3_bar; o_[3] = t_e0 - w3_tile0; o_[4] = t_e1 - t_e0; o_[5] = t_e2 - t_e1; o_[6] = t_e3 - t_e2;
            o_[7] = t_e3 - w3_tile0;
        }
        ++tile_no;
#endif
#if PULPO_ABL & 1
        {
            float t_ = 0.f;
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int x = 0; x < 4; ++x) t_ += acc[p][x][(p * 4 + x) & 15];
            if (t_ == 12345.678f) out_b[tid] = 1.f;
        }
#endif
        if (!has_next) break;
        cur = nxt;
        work = next_work;
    }
    __syncthreads();
    flush_stats();
}

// The kernel body exists in two copies that differ in ONE constant, SP.  Waves 0-3 stage behind pairs 0 / 1 of a chunk, their SIMD partners 4-7
// behind pairs 2 / 3: the kernel's phases ADD UP (ablations, DESIGN 3d: two waves per SIMD in lock step hide nothing of each other) - staggered,
// one wave's vector and LDS-write work falls into the other's matrix phase.  Two whole copies behind ONE wave-uniform branch at the top (no state
// joins behind it: the copies share only the kernel arguments and the LDS), and a compile-time constant per copy, because a wave-uniform RUNTIME
// branch around the tap loads makes every vmcnt behind it a worst-case guess (measured in round 4: -15 %).  Both copies pass the same barriers.
template <bool BNR>
__global__ __launch_bounds__(512, 1) void conv3d_k3_wino3_mfma(ConvArgs a) {
    if (PULPO_W3_SKEW && __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) != 0) wino3_body<BNR, 2>(a);
    else wino3_body<BNR, 0>(a);
}

__global__ void pack_weight_wino3_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int NPad, int dgrad, long total) {
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
        pulpo_conv::pack_wino3_one(w, wp, Cin, Cout, NPad, dgrad, e);
}

int wino3_enabled() {                                   // PULPO_CONV_WINO3=0: the (y, x) kernel everywhere (A/B switch)
    static int on = -1;
    if (on < 0) { const char* e = getenv("PULPO_CONV_WINO3"); on = e ? atoi(e) : 1; }
    return on;
}
int wino3_min_k() {                                     // PULPO_CONV_WINO3_MINK: smallest reduction-channel count that takes this kernel
    static int k = -1;
    if (k < 0) { const char* e = getenv("PULPO_CONV_WINO3_MINK"); k = e ? atoi(e) : 16; }
    return k < 16 ? 16 : k;                             // (at least two chunks per tile: the statistics' deferred flush counts on a second chunk barrier)
}

}  // namespace

namespace pulpo_conv {

// shapes the F(2x2x2,3x3x3) kernel takes: whole 4 x 8 x 8 tiles, K % 8 == 0, N % 32 == 0, at least 256 work items (one per CU)
int wino3_shape_ok(int B, int D, int H, int W, int K, int N) {
    // (output channels: a multiple of 4; a cout tile of 32 may be partly empty - the 16-channel data gradient of the feedback layer runs at half
    //  the tile's columns here as it did in the (y, x) kernel, on 1.5x fewer matrix instructions)
    const int ncot = (N + Q_NT - 1) / Q_NT;
    if (!wino3_enabled() || K < wino3_min_k() || K % Q_CH != 0 || N % 4 != 0 || (N % Q_NT != 0 && N < 16) || 3 * ncot * Q_NT > Q_TAB) return 0;
    if (D % 4 != 0 || H % TY != 0 || W % TX != 0) return 0;
    // the BatchNorm statistics rows are counted by pulpo_conv3d_k3_stat_tiles(), i.e. with conv_tz(): this kernel writes one row per 4-deep tile
    if (conv_tz(D, H, W) != 4) return 0;
    const long items = (long)B * (D / 4) * (H / TY) * (W / TX) * ncot;
    return items >= 256;
}

}  // namespace pulpo_conv

PULPO_API size_t pulpo_conv3d_k3_packed_wino3_floats(int K, int N) { return (size_t)((K + Q_CH - 1) / Q_CH) * 64 * Q_CH * npad(N); }

PULPO_API int pulpo_conv3d_k3_pack_weight_wino3(const float* w, float* wp, int Cin, int Cout, int dgrad, void* stream) {
    PULPO_REQUIRE(w && wp && Cin > 0 && Cout > 0, "conv3d_k3_pack_weight_wino3: bad arguments");
    const int K = dgrad ? Cout : Cin, N = dgrad ? Cin : Cout;
    const long total = (long)((K + Q_CH - 1) / Q_CH) * Q_CH * npad(N);          // threads: one per (chunk, k, n)
    const int nb = (int)std::min<long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(pack_weight_wino3_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wp, Cin, Cout, npad(N), dgrad, total);
    return pulpo::check_launch("pack_weight_wino3");
}

static int fwd_wino3_impl(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias, const float* coef, float slope,
                          float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats, const float* bn_y, int64_t bn_y_bs, int64_t bn_y_ps,
                          const float* bn_coef, int B, int D, int H, int W, int K, int N, void* stream, int64_t in_kb = Q_CH, int64_t out_kb = Q_CH) {
    PULPO_REQUIRE(in && wp && out, "conv3d_k3_fwd_wino3: null pointer");
    PULPO_REQUIRE(B > 0 && K > 0 && N > 0 && D > 0 && D % 4 == 0 && H > 0 && H % TY == 0 && W > 0 && W % TX == 0 && K % Q_CH == 0 && N % 4 == 0 && 3 * ((N + Q_NT - 1) / Q_NT) * Q_NT <= Q_TAB,
                  "conv3d_k3_fwd_wino3: shape %dx%dx%d, %d -> %d channels is not whole 4x8x8 tiles of 8 / 32 channels (see pulpo_conv3d_k3_algo)", D, H, W, K, N);
    PULPO_REQUIRE(K >= 2 * Q_CH, "conv3d_k3_fwd_wino3: at least two 8-channel chunks per tile (%d reduction channels given): the staging runs two chunks ahead", K);
    PULPO_REQUIRE(!stats || conv_tz(D, H, W) == 4, "conv3d_k3_fwd_wino3: %dx%dx%d is tiled 2-deep by pulpo_conv3d_k3_stat_tiles(); this kernel writes 4-deep statistics rows", D, H, W);
    PULPO_REQUIRE(!(coef && stats), "conv3d_k3_fwd_wino3: batch statistics are not available from the fused eval-mode epilogue");
    PULPO_REQUIRE(in_cs == 1 && in_ps % 4 == 0 && in_bs % 4 == 0 && (((uintptr_t)in) & 15) == 0 && (long)D * H * W * in_ps * 4 < (1L << 31) &&
                      in_kb >= Q_CH && in_kb % 4 == 0 && ((long)(K / Q_CH - 1) * in_kb + (long)D * H * W * in_ps) * 4 < (1L << 31),
                  "conv3d_k3_fwd_wino3: the operand must be channels-last (or channel-blocked), 16-byte aligned and smaller than 2 GiB per batch element");
    PULPO_REQUIRE(out_cs == 1 && out_ps % 4 == 0 && out_bs % 4 == 0 && (((uintptr_t)out) & 15) == 0, "conv3d_k3_fwd_wino3: the result must be channels-last, 16-byte aligned");
    PULPO_REQUIRE((((uintptr_t)wp) & 15) == 0, "conv3d_k3_fwd_wino3: packed weights must be 16-byte aligned");
    PULPO_REQUIRE(out_kb >= Q_CH && out_kb % 4 == 0 && (out_kb == Q_CH || N % Q_CH == 0), "conv3d_k3_fwd_wino3: a channel-blocked result needs whole 8-channel blocks");
    ConvArgs a{};
    a.in = in; a.in_bs = in_bs; a.in_ps = in_ps; a.in_cs = in_cs; a.in_kb = in_kb; a.out_kb = out_kb;
    a.wp = wp; a.bias = bias;
    a.out = out; a.out_bs = out_bs; a.out_ps = out_ps; a.out_cs = out_cs;
    a.stats = stats;
    a.coef = coef; a.slope = slope;
    a.bn_y = bn_y; a.bn_y_bs = bn_y_bs; a.bn_y_ps = bn_y_ps; a.bn_coef = bn_coef;
    a.B = B; a.D = D; a.H = H; a.W = W; a.Cin = K; a.Cout = N; a.NPad = npad(N);
    a.ntz = D / 4; a.nty = H / TY; a.ntx = W / TX;
    a.ncot = (N + Q_NT - 1) / Q_NT;
    a.ksplit = 1; a.part = nullptr;
    a.tile_order = (a.ntx % 4 == 0 && a.nty % 4 == 0 && a.ntz % 4 == 0) ? 1 : 0;
    const long nwork = (long)B * a.ntz * a.nty * a.ntx * a.ncot;
    PULPO_REQUIRE(nwork < (1L << 31), "conv3d_k3_fwd_wino3: grid too large");
    hipStream_t st = (hipStream_t)stream;
    const bool bnr = bn_y != nullptr;
    static bool attr_set[2] = {false, false};
    const void* fn = bnr ? reinterpret_cast<const void*>(&conv3d_k3_wino3_mfma<true>) : reinterpret_cast<const void*>(&conv3d_k3_wino3_mfma<false>);
    if (!attr_set[bnr]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q_LDS);
        if (e != hipSuccess) return pulpo::fail((int)e, "hipFuncSetAttribute(conv3d wino3): %s", hipGetErrorString(e));
        attr_set[bnr] = true;
    }
    const int nwg = (int)std::min<long>(nwork, 256);    // persistent workgroups: one per CU
    if (bnr) hipLaunchKernelGGL((conv3d_k3_wino3_mfma<true>), dim3(nwg), dim3(512), Q_LDS, st, a);
    else hipLaunchKernelGGL((conv3d_k3_wino3_mfma<false>), dim3(nwg), dim3(512), Q_LDS, st, a);
    return pulpo::check_launch("conv3d_k3_wino3_mfma");
}

PULPO_API int pulpo_conv3d_k3_fwd_wino3(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias,
                                        const float* coef, float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats, int B,
                                        int D, int H, int W, int K, int N, void* stream) {
    return fwd_wino3_impl(in, in_bs, in_ps, in_cs, wp, bias, coef, slope, out, out_bs, out_ps, out_cs, stats, nullptr, 0, 0, nullptr, B, D, H, W, K, N, stream);
}

PULPO_API int pulpo_conv3d_k3_dgrad_wino3_bnred(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, float* out, int64_t out_bs,
                                                int64_t out_ps, const float* bn_y, int64_t bn_y_bs, int64_t bn_y_ps, const float* bn_coef, float slope,
                                                float* part, int B, int D, int H, int W, int K, int N, void* stream) {
    PULPO_REQUIRE(bn_y && bn_coef && part, "conv3d_k3_dgrad_wino3_bnred: null pointer");
    PULPO_REQUIRE(bn_y_ps % 4 == 0 && bn_y_bs % 4 == 0 && (((uintptr_t)bn_y) & 15) == 0 && (((uintptr_t)bn_coef) & 15) == 0,
                  "conv3d_k3_dgrad_wino3_bnred: pre-norm tensor and coefficients must be channels-last and 16-byte aligned");
    return fwd_wino3_impl(in, in_bs, in_ps, in_cs, wp, nullptr, nullptr, slope, out, out_bs, out_ps, 1, part, bn_y, bn_y_bs, bn_y_ps, bn_coef, B, D, H, W, K, N,
                          stream);
}

// The same two kernels on a channel-BLOCKED operand: element (voxel v, channel c) at  in + b * in_bs + (c / 8) * in_kb + v * in_ps + c % 8  (floats).
// in_kb = 8, in_ps = row pitch is the channels-last tensor of the entries above; in_kb = D * H * W * 8, in_ps = 8 is the layout [K / 8][D][H][W][8], in
// which the four taps of a staging item are 128 consecutive bytes and a wave's tap load touches a fifth of the cache lines.
PULPO_API int pulpo_conv3d_k3_fwd_wino3_kb(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_kb, const float* wp, const float* bias,
                                           const float* coef, float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_kb, float* stats, int B,
                                           int D, int H, int W, int K, int N, void* stream) {
    return fwd_wino3_impl(in, in_bs, in_ps, 1, wp, bias, coef, slope, out, out_bs, out_ps, 1, stats, nullptr, 0, 0, nullptr, B, D, H, W, K, N, stream, in_kb,
                          out_kb);
}

PULPO_API int pulpo_conv3d_k3_dgrad_wino3_bnred_kb(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_kb, const float* wp, float* out, int64_t out_bs,
                                                   int64_t out_ps, int64_t out_kb, const float* bn_y, int64_t bn_y_bs, int64_t bn_y_ps, const float* bn_coef, float slope,
                                                   float* part, int B, int D, int H, int W, int K, int N, void* stream) {
    PULPO_REQUIRE(bn_y && bn_coef && part, "conv3d_k3_dgrad_wino3_bnred_kb: null pointer");
    PULPO_REQUIRE(bn_y_ps % 4 == 0 && bn_y_bs % 4 == 0 && (((uintptr_t)bn_y) & 15) == 0 && (((uintptr_t)bn_coef) & 15) == 0,
                  "conv3d_k3_dgrad_wino3_bnred_kb: pre-norm tensor and coefficients must be channels-last and 16-byte aligned");
    return fwd_wino3_impl(in, in_bs, in_ps, 1, wp, nullptr, nullptr, slope, out, out_bs, out_ps, 1, part, bn_y, bn_y_bs, bn_y_ps, bn_coef, B, D, H, W, K, N,
                          stream, in_kb, out_kb);
}
