// Pieces of the 3x3x3 convolution shared by the fp32 (conv3d.hip) and bf16-operand (conv3d_bf16.hip) kernels.
#pragma once
#include "common.h"

namespace pulpo_conv {

constexpr int TY = 8, TX = 8;                    // y/x extent of a voxel tile; the z extent is conv_tz() (2 or 4)
constexpr int HY = TY + 2, HX = TX + 2;

// z extent of the forward voxel tile for a volume (both precisions use the same tiling, so the BatchNorm partial-statistics
// rows written by either kernel are pulpo_conv3d_k3_stat_tiles() many)
int conv_tz(int D, int H, int W);
// (y, x) Winograd kernel on a volume whose depth is NOT a multiple of 4 (the 10^3 level): pipelined kernel with split-K work items only
int wino2_ragged_depth_ok(int B, int D, int H, int W, int K, int N);

__host__ __device__ inline int npad(int N) { return (N + 63) & ~63; }
// Cin chunk of the direct kernel's weight packing (conv3d.hip pick_ch)
__host__ __device__ inline int direct_ch(int K) { return K <= 2 ? 2 : K <= 4 ? 4 : 16; }

// F(2x2x2,3x3x3) kernel (conv3d_wino3.hip): 1 when pulpo_conv3d_k3_algo may answer 3 for the shape
int wino3_shape_ok(int B, int D, int H, int W, int K, int N);

// out = sum over the ksplit partial slabs (fixed order) + per-row BatchNorm partials; see splitk_reduce_kernel in conv3d.hip
// (coef != nullptr: eval-mode BatchNorm + LeakyReLU applied to the reduced value, see ConvArgs::coef)
// (out_dt: dtype code of `out`, 0 fp32 / 1 bf16 - rounded on the store, the statistics describe the stored values)
int launch_splitk_reduce(const float* part, int ksplit, void* out, long obs, long ops, long ocs, int B, long V, int C, int nrow, float* stats,
                         const float* coef, float slope, hipStream_t st, int out_dt = 0);

// dw[Cout][Cin][27] (+)= packed[27][Cin][NPad]; see unpack_wgrad_kernel in conv3d.hip
int launch_unpack_wgrad(const float* packed, float* dw, int Cin, int Cout, int accumulate, hipStream_t st);

// weight gradient, Winograd F(2x2,3x3) form (conv3d_wgrad_w2.hip): channels-last operands, accumulates into the zeroed packed scratch
// slabs / nslab (deterministic mode, see pulpo_conv3d_k3_wgrad_det): non-null -> split s of the grid accumulates into its own zeroed copy
// slabs + s * 27 * Cin * npad(Cout) of the packed sums instead of `scratch` (the caller adds the copies up in fixed order), at most nslab splits
int launch_wgrad_w2(const float* in, long in_bs, long in_ps, const float* go, long go_bs, long go_ps, float* scratch, int B, int D, int H, int W,
                    int Cin, int Cout, hipStream_t st, float* slabs = nullptr, int nslab = 0, int* used_slabs = nullptr, long go_kb = 8, long in_kb = 8);
// scratch[e] += slabs[0][e] + slabs[1][e] + ... (fixed order, e < n): the ordered second stage of the deterministic weight gradient
int launch_wgrad_slab_reduce(float* scratch, const float* slabs, int nslab, long n, hipStream_t st, int npad_ = 64, int cols = 64);
bool wgrad_w3_depth_ok(int D);                      // the F(2x2x2,3x3x3) weight-gradient kernel takes this depth (even, PULPO_WGRAD_W3 != 0)

// forward / data gradient, pipelined F(2x2,3x3) kernel (conv3d_wino2p.hip): channels-last 16-byte-aligned operands, K % 4 == 0; bnr = with the
// BatchNorm-backward sums of the unit in front in the epilogue (ConvArgs::bn_y)
struct ConvArgs;
int launch_wino2p(const ConvArgs& a, int nblk, bool bnr, hipStream_t st);
bool wino2p_ok(const ConvArgs& a);                 // Cin % 8 == 0 and 32-bit halo offsets

// ---- shared by the direct (conv3d.hip), Winograd (conv3d_wino.hip) and weight-gradient (conv3d_wgrad.hip) translation units
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TZ = 2, MV = TZ * TY * TX;          // base output voxel tile (weight gradient; the forward kernels use conv_tz())
constexpr int HZ = TZ + 2, HV = HZ * HY * HX;     // its halo tile

struct ConvArgs {
    const float* in;
    long in_bs, in_ps, in_cs;     // batch / pixel / channel strides in floats
    const float* wp;              // packed [nchunk][27][CH][NPad]
    const float* bias;            // nullable
    float* out;
    long out_bs, out_ps, out_cs;
    float* stats;                 // nullable: [voxel tile][2][Cout]  (sum, sum of squares of conv+bias)
    int B, D, H, W, Cin, Cout, NPad;
    int ntz, nty, ntx, ncot;
    int ksplit;                   // > 1: the Cin chunks are split over ksplit workgroups, each storing a partial slab into `part`
    float* part;                  // [ksplit][B*V][Cout] dense partial outputs (reduced in fixed order by splitk_reduce_kernel)
    const float* coef;            // nullable: eval-mode BatchNorm coefficients (scale at [2C], shift at [3C]) + LeakyReLU fused into the store
    float slope;
    // Winograd (y, x) kernel used as a data-gradient convolution whose result is the gradient dz of a ConvUnit's output z = lrelu(bn(y)):
    // bn_y / bn_coef non-null -> `stats` receives, instead of (sum, sum of squares), that unit's BatchNorm-backward partial sums
    // (sum dbn, sum dbn * xhat) per voxel tile - what bn_lrelu_bwd_reduce_kernel would compute in a pass of its own over dz and y
    const float* bn_y;
    long bn_y_bs, bn_y_ps;
    const float* bn_coef;
    int tile_order;               // pipelined (y, x) kernel: 0 = tiles in linear order (x fastest), 1 = in 4 x 4 x 4 blocks (see describe())
    int stagger;                  // pipelined (y, x) kernel: diagnostic start-up delay (units of 64 x 127 clocks) of the second half of the grid, 0 = none
    long out_kb;                  // F(2x2x2) kernel: the same block stride for the RESULT (8 = channels-last; 0 is read as 8)
    long in_kb;                   // F(2x2x2) kernel: floats between consecutive 8-channel blocks of a voxel - 8 = channels-last; V * 8 (with in_ps = 8) = the channel-BLOCKED layout
                                  // [C / 8][D][H][W][8] in which the BatchNorm backward writes the gradient it hands to the data- / weight-gradient kernels (0 is read as 8)
};


__device__ __forceinline__ int tap_halo_offset(int tap) {
    return ((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3;
}

}  // namespace pulpo_conv
