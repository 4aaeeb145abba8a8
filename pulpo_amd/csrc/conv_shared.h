// Pieces of the 3x3x3 convolution shared by the fp32 (conv3d.hip) and bf16-operand (conv3d_bf16.hip) kernels.
#pragma once
#include "common.h"

namespace pulpo_conv {

constexpr int TY = 8, TX = 8;                    // y/x extent of a voxel tile; the z extent is conv_tz() (2 or 4)
constexpr int HY = TY + 2, HX = TX + 2;

// z extent of the forward voxel tile for a volume (both precisions use the same tiling, so the BatchNorm partial-statistics
// rows written by either kernel are pulpo_conv3d_k3_stat_tiles() many)
int conv_tz(int D, int H, int W);

inline int npad(int N) { return (N + 63) & ~63; }

// out = sum over the ksplit partial slabs (fixed order) + per-row BatchNorm partials; see splitk_reduce_kernel in conv3d.hip
// (coef != nullptr: eval-mode BatchNorm + LeakyReLU applied to the reduced value, see ConvArgs::coef)
int launch_splitk_reduce(const float* part, int ksplit, float* out, long obs, long ops, long ocs, int B, long V, int C, int nrow, float* stats,
                         const float* coef, float slope, hipStream_t st);

// dw[Cout][Cin][27] (+)= packed[27][Cin][NPad]; see unpack_wgrad_kernel in conv3d.hip
int launch_unpack_wgrad(const float* packed, float* dw, int Cin, int Cout, int accumulate, hipStream_t st);

}  // namespace pulpo_conv
