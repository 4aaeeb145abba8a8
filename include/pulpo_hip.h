/* pulpo_hip.h — C ABI of libpulpo_hip.so: PULPo's registration hot path as hand-written HIP kernels for
 * AMD Instinct MI355X (gfx950, CDNA4).
 *
 * The reference (leonardsiegert/PULPo) has no native/FFI layer: its hot path is a sequence of ATen operators issued
 * from Python.  Each entry point below replaces one of those operator call sites; the citation names
 * the reference line that issues it.  The Python host (pulpo_amd/ops.py) binds these with ctypes inside
 * torch.autograd.Function wrappers; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer to fp32 unless stated; the library never allocates, frees or synchronises:
 *    outputs, scratch and saved-for-backward buffers are owned by the caller.  Work is only enqueued on `stream`
 *    (a hipStream_t passed as void*; NULL = the default stream).  No global mutable state: safe to call from the
 *    main thread and the autograd thread concurrently, and from one process per GPU.
 *  - Return value: 0 on success, otherwise a non-zero code (hipError_t value, or -1 for an argument error);
 *    pulpo_last_error() then returns a thread-local message.  Nothing throws.
 *  - "channels-last" activations: element (b, voxel v, channel c) lives at  base + b*bs + v*ps + c*cs  (strides in
 *    floats).  A contiguous (B,D,H,W,C) tensor has ps = C, cs = 1, bs = D*H*W*C; a channel slice of a wider
 *    concatenation buffer keeps the buffer's ps.  Where only `ps` is passed, cs == 1 and bs == D*H*W*ps are implied.
 *  - "planar": contiguous (B, C, D, H, W), the reference's own layout, used for 1- and 3-channel images and fields.
 *  - voxel order is (D,H,W) = ('ij' indexing of dims 0,1,2), channel i of a displacement field moves along dim i
 *    (src/network_blocks.py:94-98).
 */
#ifndef PULPO_HIP_H
#define PULPO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------- runtime
 * PULPO_ABI_VERSION is bumped whenever a prototype below changes its argument list or a buffer contract, or an entry point is removed
 * (history: INTEGRATION.md "ABI history").  pulpo_abi_version() returns the value the library was built with: a client compares it with
 * the header it was compiled against before the first call (pulpo_amd/_lib.py does). */
#define PULPO_ABI_VERSION 5
int pulpo_abi_version(void);
const char* pulpo_last_error(void);

/* ------------------------------------------------------------------------- ConvUnit: Conv3d(k=3, pad=1, bias)
 * replaces nn.Conv3d inside ConvUnit (src/network_blocks.py:23) = aten::convolution / convolution_backward.
 * Implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32).  K = reduction channels, N = output channels.
 *   forward : K = Cin,  N = Cout, weights packed with dgrad = 0
 *   dgrad   : K = Cout, N = Cin,  weights packed with dgrad = 1 (transposed + tap-flipped), bias = NULL
 * `stats` (nullable): [pulpo_conv3d_k3_stat_tiles()][2][N] per-tile (sum, sum of squares) of the outputs, the
 * BatchNorm batch statistics partials (src/network_blocks.py:24). */
size_t pulpo_conv3d_k3_packed_floats(int K, int N);
int pulpo_conv3d_k3_pack_weight(const float* w /*[Cout][Cin][3][3][3]*/, float* wp, int Cin, int Cout, int dgrad, void* stream);
int pulpo_conv3d_k3_stat_tiles(int B, int D, int H, int W);
int pulpo_conv3d_k3_tile_config(int K, int N); /* CH*1000 + NT of the kernel instantiation used (for profiling) */
size_t pulpo_conv3d_k3_fwd_scratch_floats(int B, int D, int H, int W, int K, int N); /* > 0 for small volumes (deterministic split-K) */
int pulpo_conv3d_k3_fwd(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias, float* out,
                        int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats, float* scratch /*nullable if the query is 0*/, int B,
                        int D, int H, int W, int K, int N, void* stream);
/* eval-mode ConvUnit in ONE kernel (inference: predict / predict_deterministic / evaluate.py): out = LeakyReLU_slope(conv * scale + shift)
 * with coef from pulpo_bn_eval_coef (running statistics folded) - the store of the convolution applies what pulpo_bn_lrelu_apply would */
int pulpo_conv3d_k3_fwd_bn_lrelu(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias,
                                 const float* coef, float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* scratch, int B,
                                 int D, int H, int W, int K, int N, void* stream);
/* Winograd F(2x2,3x3) in (y, x) of the same convolution for volumes tiled 4x8x8 (2.25x fewer matrix instructions, all fp32; results differ
 * from the direct kernel by fp32 rounding only).  pulpo_conv3d_k3_algo() says which kernel family to use for a shape (0 direct, 2 this one);
 * each has its own weight packing; coef (nullable) selects the fused eval-mode BatchNorm + LeakyReLU store, stats (nullable) the BatchNorm
 * partials (same tiles).  (Family 1, F(2,3) along x only, was retired in round 3: measured slower on every shape.) */
int pulpo_conv3d_k3_algo(int B, int D, int H, int W, int K, int N);
size_t pulpo_conv3d_k3_packed_wino2_floats(int K, int N);
int pulpo_conv3d_k3_pack_weight_wino2(const float* w /*[Cout][Cin][3][3][3]*/, float* wp, int Cin, int Cout, int dgrad, void* stream);
/* scratch: pulpo_conv3d_k3_fwd_wino2_scratch_floats() floats, > 0 for volumes with fewer (voxel tile, channel tile) pairs than resident
 * workgroups (the 20^3 level): the reduction channels are split over several work items (partial slabs + ordered reduce, deterministic) */
size_t pulpo_conv3d_k3_fwd_wino2_scratch_floats(int B, int D, int H, int W, int K, int N);
int pulpo_conv3d_k3_fwd_wino2(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias, const float* coef,
                              float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats,
                              float* scratch /*nullable if the query is 0*/, int B, int D, int H, int W, int K, int N, void* stream);
/* 1 when the _wino2 entry points run the pipelined kernel (conv3d_wino2p.hip: double-buffered halo images, weights global -> registers) for a
 * channels-last 16-byte-aligned operand: K % 8 == 0 and D*H*W*in_ps*4 < 2^31.  Otherwise (and with PULPO_W2_PIPE=0) the round-2 kernel runs.
 * Same results either way (network_blocks.py:23). */
int pulpo_conv3d_k3_wino2_pipelined(int D, int H, int W, int K, int64_t in_ps);
/* The data-gradient convolution of a ConvUnit (in = dy of that unit, wp packed with dgrad = 1, N = the unit's input channels) with the
 * FIRST pass of the BatchNorm/LeakyReLU backward of the ConvUnit in front of it fused into the store (the reference runs these as
 * separate autograd nodes: ConvolutionBackward of src/network_blocks.py:23, then LeakyReluBackward / NativeBatchNormBackward of :24-25):
 * part[tile][2][N], tile < pulpo_conv3d_k3_stat_tiles(), receives sum(dbn) and sum(dbn * (bn_y - fp32 batch mean)) per voxel tile, where
 * dbn = out * lrelu'(bn_y * scale + shift); the layout pulpo_bn_bwd_finalize takes.  bn_y / bn_coef: pre-norm tensor (channels-last,
 * N channels) and coefficient block (pulpo_bn_fwd_finalize) of the unit in front.  _ok() = 1 when the shape is accepted. */
int pulpo_conv3d_k3_dgrad_wino2_bnred_ok(int B, int D, int H, int W, int K, int N);
int pulpo_conv3d_k3_dgrad_wino2_bnred(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, float* out, int64_t out_bs,
                                      int64_t out_ps, const float* bn_y, int64_t bn_y_bs, int64_t bn_y_ps, const float* bn_coef, float slope,
                                      float* part, int B, int D, int H, int W, int K, int N, void* stream);
/* Re-packing of many weights in one launch (after an optimizer step wrote the parameters through raw pointers): jobs is a DEVICE array,
 * wp buffers as sized by pulpo_conv3d_k3_packed_floats (kind 0, the layout of pulpo_conv3d_k3_pack_weight) or
 * pulpo_conv3d_k3_packed_wino2_floats (kind 2, the layout of pulpo_conv3d_k3_pack_weight_wino2) or pulpo_conv3d_k3_packed_bf16_elems
 * bf16 elements (kind 3, the layout of pulpo_conv3d_k3_pack_weight_bf16; wp then points to uint16_t). */
typedef struct PulpoPackJob {
    const float* w;   /* [Cout][Cin][3][3][3] */
    float* wp;
    int Cin, Cout, dgrad, kind;
} PulpoPackJob;
int pulpo_conv3d_k3_pack_weights_multi(const PulpoPackJob* jobs, int njobs, void* stream);
/* weight gradient: dw[Cout][Cin][27] = sum_voxels in[v + tap - 1][ci] * dy[v][co]; scratch is overwritten */
size_t pulpo_conv3d_k3_wgrad_scratch_floats(int Cin, int Cout);
/* which weight-gradient kernel pulpo_conv3d_k3_wgrad runs for a shape: 3 = Winograd F(2x2x2,3x3x3) (even depths), 2 = Winograd F(2x2,3x3) in
 * (y, x), 1 = Winograd F(2,3) along x, 0 = direct.  vec != 0: both operands channels-last, 16-byte aligned, channel counts multiples of 4 (diagnostics / roofline accounting) */
int pulpo_conv3d_k3_wgrad_algo(int B, int D, int H, int W, int Cin, int Cout, int vec);
/* accumulate: 0 dw = result, 1 dw += result, 2 DEFERRED - `scratch` must arrive all zero, keeps the packed sums [27][Cin][NPad] and dw (may
 * be NULL) is not touched: the caller finishes every deferred gradient of a backward pass with one pulpo_grad_finish_multi launch (which
 * adds into dw and returns the scratch to all zero).  The same holds for the bf16-operand variant below. */
int pulpo_conv3d_k3_wgrad(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dy, int64_t dy_bs, int64_t dy_ps,
                          int64_t dy_cs, float* dw, int accumulate, float* scratch, int B, int D, int H, int W, int Cin,
                          int Cout, void* stream);
/* One launch for the parameter-gradient epilogues of a whole backward pass (the reference has no counterpart: autograd's per-parameter
 * AccumulateGrad adds, torch.optim's view of .grad).  jobs: DEVICE array of njobs entries.
 *   kind 0  dst[co][ci][27] += src[tap][ci][co of NPad] and src := 0     a = Cin, b = Cout, c = NPad   (deferred pulpo_conv3d_k3_wgrad)
 *   kind 1  dst[col] += sum_r src[r][col]                                a = rows, b = columns         (conv-bias gradient from the
 *           pulpo_bn_lrelu_bwd_apply partials) */
typedef struct PulpoGradJob {
    const float* src;
    float* dst;
    int kind, a, b, c;
} PulpoGradJob;
int pulpo_grad_finish_multi(const PulpoGradJob* jobs, int njobs, void* stream);

/* bf16-operand variant (BASELINE configs 4-5; no counterpart in the reference, whose arithmetic is fp32 throughout - SURVEY.md 8(d)):
 * the same convolution with both operands rounded to bf16 (round-to-nearest-even) as they are staged, products and sums in
 * fp32 on v_mfma_f32_32x32x16_bf16.  Activations, gradients, bias, outputs and `stats` stay fp32; `stats` has
 * pulpo_conv3d_k3_fwd_bf16_stat_tiles() rows.  wp holds bf16 bit patterns. */
size_t pulpo_conv3d_k3_packed_bf16_elems(int K, int N);
int pulpo_conv3d_k3_pack_weight_bf16(const float* w /*[Cout][Cin][3][3][3]*/, uint16_t* wp, int Cin, int Cout, int dgrad, void* stream);
size_t pulpo_conv3d_k3_fwd_bf16_scratch_floats(int B, int D, int H, int W, int K, int N);
int pulpo_conv3d_k3_fwd_bf16_stat_tiles(int B, int D, int H, int W); /* rows of `stats` this kernel writes (its own tiling policy) */
int pulpo_conv3d_k3_fwd_bf16(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const uint16_t* wp, const float* bias, float* out,
                             int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats, float* scratch, int B, int D, int H, int W, int K,
                             int N, void* stream);
int pulpo_conv3d_k3_fwd_bn_lrelu_bf16(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const uint16_t* wp, const float* bias,
                                      const float* coef, float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* scratch,
                                      int B, int D, int H, int W, int K, int N, void* stream);
int pulpo_conv3d_k3_wgrad_bf16(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dy, int64_t dy_bs, int64_t dy_ps,
                               int64_t dy_cs, float* dw, int accumulate, float* scratch /*pulpo_conv3d_k3_wgrad_scratch_floats*/, int B, int D,
                               int H, int W, int Cin, int Cout, void* stream);

/* ------------------------------------------------------------- ConvUnit: BatchNorm3d + LeakyReLU(0.2, inplace)
 * replaces nn.BatchNorm3d / nn.LeakyReLU (src/network_blocks.py:24-25) = aten::native_batch_norm(+_backward),
 * aten::leaky_relu_(+_backward).  coef = 8*C floats: [4][C] floats (mean, rstd, scale = gamma*rstd, shift = beta - mean*scale)
 * followed by [2][C] doubles (mean, rstd): the backward carries the per-channel means in double, as ATen's CPU kernels do. */
int pulpo_colsum(const float* partials, int nrow, int ncol, float* out, float scale, int accumulate, void* stream);
size_t pulpo_bn_fwd_finalize_scratch_doubles(int ntile, int C);
int pulpo_bn_fwd_finalize(const float* stats, int ntile, int C, double count, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked /*nullable, += 1*/, float momentum, float eps, float* coef,
                          double* scratch, void* stream);
int pulpo_bn_eval_coef(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C,
                       float* coef, void* stream);
int pulpo_bn_lrelu_apply(const float* y, int64_t yps, float* z, int64_t zps, const float* coef, int64_t npix, int C, float slope, void* stream);
/* the same pass for the last ConvUnit of an encoder level, whose output is pooled for the next level (components/pulpo.py:58): writes z AND
 * pooled = AvgPool3d(2, 2, ceil_mode=True)(z) while y is read once; _ok() = 1 when the float4 path applies (else: the two separate passes) */
int pulpo_bn_lrelu_apply_pool2_ok(int C, int64_t yps, int64_t zps, int64_t pps);
int pulpo_bn_lrelu_apply_pool2(const float* y, int64_t yps, float* z, int64_t zps, float* pooled, int64_t pps, const float* coef, int B, int D, int H,
                               int W, int C, float slope, void* stream);
int pulpo_bn_bwd_blocks(int64_t npix, int C);
/* gin = (add +) avgpool2_bwd(gout) AND the partial rows of pulpo_bn_lrelu_bwd_reduce for the ConvUnit whose output was pooled, in one pass
 * (the last unit of every encoder level: pooling backward, autograd's accumulation add and the BatchNorm-backward reduction read its
 * gradient three times otherwise); partial: pulpo_bn_bwd_blocks(B*D*H*W, C) rows of 2C floats; add nullable */
int pulpo_avgpool2_bwd_bnred(const float* gout, int64_t gops, const float* add, int64_t aps, float* gin, int64_t gips, const float* y, int64_t yps,
                             const float* coef, float slope, float* partial, int B, int D, int H, int W, int C, void* stream);
int pulpo_bn_lrelu_bwd_reduce(const float* dz, int64_t dzps, const float* y, int64_t yps, const float* coef, int64_t npix, int C, float slope,
                              float* partial /*[blocks][2C]*/, void* stream);
/* rows [nrow][2][C] = (sum dbn, sum dbn * (y - fp32 batch mean)): the block partials of pulpo_bn_lrelu_bwd_reduce (nrow = pulpo_bn_bwd_blocks)
 * or the per-voxel-tile rows of pulpo_conv3d_k3_dgrad_wino2_bnred (nrow = pulpo_conv3d_k3_stat_tiles).  coef: the unit's coefficient block.
 * scratch: pulpo_bn_bwd_finalize_scratch_doubles(nrow, C) doubles (NULL when that is 0). */
size_t pulpo_bn_bwd_finalize_scratch_doubles(int nrow, int C);
int pulpo_bn_bwd_finalize(const float* rows, int nrow, int C, const float* coef, double count, int use_means, float* dbeta, float* dgamma,
                          int accumulate, double* totd /*[2C]: mean(dbn) | mean(dbn*xhat)*/, double* scratch, void* stream);
int pulpo_bn_lrelu_bwd_apply(const float* dz, int64_t dzps, const float* y, int64_t yps, const float* coef, const double* totd, float* dy,
                             int64_t dyps, int64_t npix, int C, float slope, float* partial2 /*[blocks][C]*/, void* stream);

/* ---------------------------------------------------------------------------- 1x1x1 heads (channel mixing C -> 3)
 * nout = 6: MuSigmaBlock + gauss_sampler (src/network_blocks.py:54-60, :7-8; eps = injected N(0,1) noise, NULL -> z = mu)
 * nout = 3: VelocityField's last conv (src/network_blocks.py:81).   Wt = [nout][C], bias = [nout]; outputs planar (B,3,V). */
int pulpo_heads_fwd(const float* h, int64_t ps, const float* Wt, const float* bias, const float* eps, float* o0, float* o1, float* o2, int nout,
                    int B, int64_t V, int C, void* stream);
int pulpo_heads_bwd_blocks(int B, int64_t V, int C);
int pulpo_heads_bwd(const float* h, int64_t ps, const float* Wt, const float* g0, const float* g1, const float* g2, const float* eps,
                    const float* sigma, float* dh, int64_t dps, float* partial /*[blocks][nout*C+nout]*/, int nout, int B, int64_t V, int C,
                    void* stream);

/* ------------------------------------------------------------------------------------------------- resampling
 * avgpool2: nn.AvgPool3d(2,2,ceil_mode=True) (src/components/pulpo.py:33,59,174-177), channels-last, any C (C = 1: images)
 * resize_trilinear: F.interpolate(trilinear, align_corners=False) on planar tensors (pulpo.py:202; network_blocks.py:141-147
 *   with `mult` = ResizeTransform's factor and `add` = DFAdder's second operand, network_blocks.py:156-157; losses.py:313)
 * feedback_up2: the x2 up-sampling + torch.cat of the feedback tensors (pulpo.py:195-206), planar sources -> channels-last out */
int pulpo_avgpool2_fwd(const float* in, int64_t ips, float* out, int64_t ops, int B, int D, int H, int W, int C, void* stream);
int pulpo_avgpool2_bwd(const float* gout, int64_t gops, float* gin, int64_t gips, int B, int D, int H, int W, int C, void* stream);
/* gin = add + avgpool2_bwd(gout): the gradient of a tensor that is pooled AND used as a skip connection (components/pulpo.py:58, 77), both
 * contributions in one pass instead of a pooling backward and autograd's accumulation add; add: channels-last with voxel stride aps */
int pulpo_avgpool2_bwd_add(const float* gout, int64_t gops, const float* add, int64_t aps, float* gin, int64_t gips, int B, int D, int H, int W,
                           int C, void* stream);
int pulpo_resize_trilinear_fwd(const float* in, const float* add, float* out, int64_t nplanes, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                               float mult, void* stream);
int pulpo_resize_trilinear_bwd(const float* gout, float* gin, int64_t nplanes, int Di, int Hi, int Wi, int Do, int Ho, int Wo, float mult,
                               void* stream);
/* the same with explicit source-coordinate steps per output voxel (scale_* <= 0: in / out, as above).  F.interpolate(scale_factor = f) - what
 * ResizeTransform calls, src/network_blocks.py:141-147 - maps coordinates with 1 / f whatever floor(in * f) is, so for sizes where in * f is
 * not an integer (a factor < 1 on odd sizes, non-integer factors) it differs from the size-ratio mapping of F.interpolate(size = ...)
 * (components/pulpo.py:202, losses.py:313).  Since ABI 3. */
int pulpo_resize_trilinear_scaled_fwd(const float* in, const float* add, float* out, int64_t nplanes, int Di, int Hi, int Wi, int Do, int Ho,
                                      int Wo, float scale_d, float scale_h, float scale_w, float mult, void* stream);
int pulpo_resize_trilinear_scaled_bwd(const float* gout, float* gin, int64_t nplanes, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                      float scale_d, float scale_h, float scale_w, float mult, void* stream);
int pulpo_feedback_up2_fwd(const float* const* srcs /*host array of device ptrs*/, const int* chans /*host*/, int nsrc, float* out, int64_t ops,
                           int B, int Di, int Hi, int Wi, void* stream);
int pulpo_feedback_up2_bwd(const float* gout, int64_t gops, float* const* gsrcs /*host array, entries may be NULL*/, const int* chans, int nsrc,
                           int B, int Di, int Hi, int Wi, void* stream);

/* ------------------------------------------------------------------------------ warp and scaling-and-squaring
 * warp3d: SpatialTransformer.forward (src/network_blocks.py:101-121) = grid_sample(bilinear, border, align_corners=False)
 *   on coordinates normalised by (S-1); image size may differ from the grid size (src/models.py:330).
 * vecint: VecInt.forward (src/network_blocks.py:173-177).  work = (nsteps+1) field buffers, result = the last one. */
int pulpo_warp3d_fwd(const float* df, const float* img, float* out, int B, int C, int Dg, int Hg, int Wg, int Di, int Hi, int Wi, void* stream);
int pulpo_warp3d_bwd(const float* df, const float* img, const float* gout, float* gdf, float* gimg, int B, int C, int Dg, int Hg, int Wg, int Di,
                     int Hi, int Wi, void* stream);
int pulpo_vecint_fwd(const float* v, float* work, int B, int D, int H, int W, int nsteps, void* stream);
/* tmp: pulpo_vecint_bwd_tmp_floats() floats (one zero-filled buffer per step for fields of 16^3 and up, two otherwise) */
size_t pulpo_vecint_bwd_tmp_floats(int B, int D, int H, int W, int nsteps);
int pulpo_vecint_bwd(const float* work, const float* gout, float* gin, float* tmp /*nullable if the query is 0*/, int B, int D, int H, int W,
                     int nsteps, void* stream);

/* ------------------------------------------------------------------------------------------------------ losses
 * ncc:   NCC_loss (src/losses.py:85-135); I = y_true, J = y_pred, planar (B,1,D,H,W); N = B*D*H*W.
 * kl:    KL_two_gauss_with_diag_cov (src/losses.py:47-76); mu1/sigma1 NULL = the N(0,1) prior (src/components/pulpo.py:330-341)
 * l2reg: L2_reg (src/losses.py:208-222).
 * Forward kernels write pulpo_loss_blocks(n) partial sums; pulpo_colsum(partial, blocks, 1, out, scale) finishes the
 * scalar.  Backward kernels take the upstream scalar gradient as a device pointer `gscale` (nullable = 1). */
int pulpo_loss_blocks(int64_t n);
int pulpo_ncc_fwd(const float* I, const float* J, float* S /*5N, saved*/, float* T /*10N scratch*/, float* partial, int B, int D, int H, int W,
                  int win, void* stream);
int pulpo_ncc_bwd(const float* I, const float* J, const float* S, float* T /*6N scratch*/, const float* gscale, float coef, float* gJ, int B, int D,
                  int H, int W, int win, void* stream);
int pulpo_kl_fwd(const float* mu, const float* sigma, const float* mu1 /*nullable: 0*/, const float* sigma1 /*nullable: 1*/, int64_t n,
                 float* partial, void* stream);
int pulpo_kl_bwd(const float* mu, const float* sigma, const float* mu1, const float* sigma1, const float* gscale, float coef, float* gmu,
                 float* gsigma, int64_t n, void* stream);
int pulpo_l2reg_fwd(const float* df, int64_t nplanes, int D, int H, int W, float* partial, void* stream);
int pulpo_l2reg_bwd(const float* df, const float* gscale, float coef, float* gdf, int64_t nplanes, int D, int H, int W, void* stream);
/* sum_l w_l * term_l of the Hierarchical* losses (src/losses.py:262-276, 305-325, 343-355) and models.py:161-162's `kl * beta` in one launch:
 * levels[i] = w[i] * terms[i] (* scale if scaled), total[0] = (sum_i w[i] * terms[i]) (* scale if scaled), summed in index order like the
 * reference's `loss = loss + all_levels[l]`.  Device arrays of n floats (total: 1); n <= 64.  Backward: gterms[i] = (gtotal[0] + glevels[i]) *
 * scale * w[i]; gtotal / glevels nullable (= 0). */
int pulpo_weighted_sum_fwd(const float* terms, const float* weights, int n, float scale, int scaled, float* levels, float* total, void* stream);
int pulpo_weighted_sum_bwd(const float* gtotal, const float* glevels, const float* weights, int n, float scale, float* gterms, void* stream);

/* ------------------------------------------------------------------- alternative losses / evaluation metrics
 *   sqdiff: L2_loss (src/losses.py:79-83); dice: Soft_dice_loss (:137-145);
 * jacdet: jacobian_det (:172-199, 3-D); jdetstd: JDetStd (:202-204).  Scalars finish on the device. */
int pulpo_metric_blocks(int64_t n);
int pulpo_sqdiff_fwd(const float* a, const float* b, int64_t n, float* partial, void* stream);
int pulpo_sqdiff_bwd(const float* a, const float* b, const float* gscale, float coef, float* ga, int64_t n, void* stream);
int pulpo_dice_blocks(int64_t V);
int pulpo_dice_fwd(const float* inp, const float* tgt, int nplanes, int64_t V, float dice_factor, float* partial, double* numden, float* loss,
                   void* stream);
int pulpo_dice_bwd(const float* inp, const float* tgt, const double* numden, const float* gscale, int nplanes, int64_t V, float dice_factor,
                   float* ginp, void* stream);
int pulpo_jacdet_fwd(const float* df, float* out, float* partial /*nullable*/, int B, int D, int H, int W, int normalize, void* stream);
int pulpo_jdetstd_finalize(const float* partial, int64_t n, float lamb, double* stat, float* loss, void* stream);
/* KL_nondiagonal.loss (src/losses.py:8-44, 3-D): mu, sigma planar (B,3,D,H,W), nplanes = B*3 */
int pulpo_kl_nondiag_fwd(const float* mu, const float* sigma, int64_t nplanes, int D, int H, int W, float prior_lambda, float* partial, float* loss,
                         void* stream);
int pulpo_kl_nondiag_bwd(const float* mu, const float* sigma, const float* gscale, int64_t nplanes, int D, int H, int W, float prior_lambda,
                         float* gmu, float* gsigma, void* stream);
int pulpo_jdetstd_bwd(const float* df, const float* jdet, const double* stat, const float* gscale, float lamb, float* gdf, int B, int D, int H, int W,
                      int normalize, void* stream);

/* Evaluation scalars of the reference's harness, finished on the device:
 *   pulpo_rmse           Evaluate.rmse, evaluate.py:315-319   sqrt(MSELoss(input, target))
 *   pulpo_dsc            Evaluate.dsc, evaluate.py:321-327    mean over (B, C) of (2 mean(t i) + 1e-6) / (mean(t^2) + mean(i^2) + 1e-6)
 *   pulpo_percent_leq0   evaluate.py:1441-1446 ("JDetLeq0")   100 * count(jacobian_det <= 0) / numel
 *   pulpo_warp_landmarks Evaluate.warp_landmarks, evaluate.py:410-423 = src/components/utils.py:15-25
 *                        out[s][k][:] = long(lm[k]) - df[s, :, lm[k][0], lm[k][1], lm[k][2]]; *flag = 1 if an index is out of range */
int pulpo_rmse(const float* a, const float* b, int64_t n, float* partial /* pulpo_metric_blocks(n) */, float* out, void* stream);
int pulpo_dsc(const float* inp, const float* tgt, int nplanes, int64_t V, float* partial /* nplanes*pulpo_dice_blocks(V)*3 */, float* out,
              void* stream);
int pulpo_percent_leq0(const float* x, int64_t n, float* partial /* pulpo_metric_blocks(n) */, float* out, void* stream);
int pulpo_warp_landmarks(const float* lm /*(nlm,nd)*/, const float* df /*(nsamp,nd,D,H,W)*/, float* out /*(nsamp,nlm,nd)*/, int nlm, int nsamp,
                         int nd, int D, int H, int W, int* flag, void* stream);

/* ------------------------------------------------------------------------------- Monte-Carlo uncertainty statistics
 * evaluate.py:222-251 stacks N sampled volumes / fields per level and takes torch.std(axis=0) (unbiased) then torch.mean over the
 * channel axis.  Streaming form: fold sample k (1-based count) into running (mean, M2) images of the sample's shape, then
 * out[b][v] = mean_c sqrt(M2[b][c][v] / (k - 1)) (* |scale[b][v]| when scale != NULL: the warped mask of evaluate.py:249). */
int pulpo_mc_moments_update(const float* sample, float* mean, float* m2, int64_t n, int k, void* stream);
int pulpo_mc_moments_std(const float* m2, const float* scale /*nullable (B,V)*/, float* out /*(B,V)*/, int B, int C, int64_t V, int k, void* stream);

/* --------------------------------------------------------------------------------------------------- optimizer
 * torch.optim.Adam(lr) defaults (src/models.py:398-400) over a flat fp32 arena; gscale pre-multiplies the gradient. */
int pulpo_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step, float gscale,
                    void* stream);

/* ------------------------------------------------------------------------- Winograd F(2x2x2,3x3x3): the deep layers (since ABI 3)
 * The same convolution (src/network_blocks.py:23, forward and data gradient) with minimal filtering along z, y AND x: 64 products per 2x2x2
 * output block instead of 216 (1.5x fewer matrix instructions than the (y, x) form above), all fp32, coefficients +-1 and 1/2 (results differ
 * from the direct kernel by fp32 rounding only).  pulpo_conv3d_k3_algo() answers 3 for the shapes it takes: whole 4x8x8 tiles (D % 4 == 0,
 * H % 8 == 0, W % 8 == 0) of a volume of at least 20^3 voxels (where pulpo_conv3d_k3_stat_tiles() counts 4-deep tiles too), K % 8 == 0 and
 * K >= 16 (two 8-channel chunks: the entry point refuses fewer), N % 4 == 0 and N >= 16 (cout tiles of 32, the last one may be partly empty), at
 * least 256 (tile, cout tile) work items; operands channels-last and 16-byte aligned (the caller guarantees that; the entry point refuses
 * anything else).  stats / part: pulpo_conv3d_k3_stat_tiles(B, D, H, W) rows of 2 * N floats, one row per 4x8x8 tile.  Own weight packing
 * (kind 4 of PulpoPackJob); stats / coef / bias and the _bnred form as for the (y, x) kernel. */
size_t pulpo_conv3d_k3_packed_wino3_floats(int K, int N);
int pulpo_conv3d_k3_pack_weight_wino3(const float* w /*[Cout][Cin][3][3][3]*/, float* wp, int Cin, int Cout, int dgrad, void* stream);
int pulpo_conv3d_k3_fwd_wino3(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, const float* bias, const float* coef,
                              float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, float* stats, int B, int D, int H, int W, int K,
                              int N, void* stream);
int pulpo_conv3d_k3_dgrad_wino3_bnred(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* wp, float* out, int64_t out_bs,
                                      int64_t out_ps, const float* bn_y, int64_t bn_y_bs, int64_t bn_y_ps, const float* bn_coef, float slope, float* part,
                                      int B, int D, int H, int W, int K, int N, void* stream);

/* ------------------------------------------------------------------------- bf16 ACTIVATION STORAGE (BASELINE configs 4-5; since ABI 3)
 * No counterpart in the reference (fp32 throughout).  On top of the bf16-operand convolutions the multi-channel activation
 * tensors of a ConvUnit - the pre-norm output y (src/network_blocks.py:23), the output z (:25), pooled / concatenated / up-sampled feature
 * maps (components/pulpo.py:58, 195-206, 250) - and their gradients may live in HBM as bf16: every kernel below computes in fp32 and
 * rounds to nearest even on the store; BatchNorm statistics describe the tensor AS STORED.  These are the typed forms of the entry points
 * above: activation operands are `void*` with a dtype code (0 = fp32, 1 = bf16), strides in ELEMENTS, everything else (coefficients,
 * partial sums, planar fields, parameters) stays fp32; with code 0 they are the fp32 entry points.  The convolution and weight-gradient
 * kernels take operand and result in ONE storage type (dt). */
int pulpo_conv3d_k3_fwd_bf16_t(const void* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const uint16_t* wp, const float* bias, void* out,
                               int64_t out_bs, int64_t out_ps, int64_t out_cs, int dt, float* stats, float* scratch, int B, int D, int H, int W,
                               int K, int N, void* stream);
int pulpo_conv3d_k3_fwd_bn_lrelu_bf16_t(const void* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const uint16_t* wp, const float* bias,
                                        const float* coef, float slope, void* out, int64_t out_bs, int64_t out_ps, int64_t out_cs, int dt,
                                        float* scratch, int B, int D, int H, int W, int K, int N, void* stream);
int pulpo_conv3d_k3_wgrad_bf16_t(const void* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const void* dy, int64_t dy_bs, int64_t dy_ps,
                                 int64_t dy_cs, int dt, float* dw, int accumulate, float* scratch, int B, int D, int H, int W, int Cin, int Cout,
                                 void* stream);
int pulpo_bn_lrelu_apply_t(const void* y, int y_dt, int64_t yps, void* z, int z_dt, int64_t zps, const float* coef, int64_t npix, int C, float slope,
                           void* stream);
int pulpo_bn_lrelu_apply_pool2_t(const void* y, int y_dt, int64_t yps, void* z /* nullable since ABI 4: only the pooled tensor is written */, int z_dt, int64_t zps,
                                 void* pooled /* z_dt */, int64_t pps,
                                 const float* coef, int B, int D, int H, int W, int C, float slope, void* stream);
int pulpo_bn_lrelu_bwd_reduce_t(const void* dz, int dz_dt, int64_t dzps, const void* y, int y_dt, int64_t yps, const float* coef, int64_t npix, int C,
                                float slope, float* partial, void* stream);
int pulpo_bn_lrelu_bwd_apply_t(const void* dz, int dz_dt, int64_t dzps, const void* y, int y_dt, int64_t yps, const float* coef, const double* totd,
                               void* dy /* y's dtype */, int64_t dyps, int64_t npix, int C, float slope, float* partial2, void* stream);
int pulpo_avgpool2_bwd_bnred_t(const void* gout, int64_t gops, const void* add, int64_t aps, void* gin, int64_t gips, int g_dt /* gout, add, gin */,
                               const void* y, int y_dt, int64_t yps, const float* coef, float slope, float* partial, int B, int D, int H, int W,
                               int C, void* stream);
int pulpo_heads_fwd_t(const void* h, int h_dt, int64_t ps, const float* Wt, const float* bias, const float* eps, float* o0, float* o1, float* o2,
                      int nout, int B, int64_t V, int C, void* stream);
int pulpo_heads_bwd_t(const void* h, int h_dt, int64_t ps, const float* Wt, const float* g0, const float* g1, const float* g2, const float* eps,
                      const float* sigma, void* dh /* h's dtype */, int64_t dps, float* partial, int nout, int B, int64_t V, int C, void* stream);
int pulpo_avgpool2_fwd_t(const void* in, int64_t ips, void* out, int64_t ops, int dt, int B, int D, int H, int W, int C, void* stream);
int pulpo_avgpool2_bwd_t(const void* gout, int64_t gops, const void* add /* nullable */, int64_t aps, void* gin, int64_t gips, int dt, int B, int D,
                         int H, int W, int C, void* stream);
int pulpo_feedback_up2_fwd_t(const float* const* srcs /*host array of device ptrs, planar fp32*/, const int* chans /*host*/, int nsrc, void* out,
                             int dt, int64_t ops, int B, int Di, int Hi, int Wi, void* stream);
int pulpo_feedback_up2_bwd_t(const void* gout, int dt, int64_t gops, float* const* gsrcs /*host array, entries may be NULL*/, const int* chans,
                             int nsrc, int B, int Di, int Hi, int Wi, void* stream);

/* ------------------------------------------------------------------------- input-layer weight gradient with the unit's BatchNorm backward fused (since ABI 4)
 * The ConvUnit of the image pair (src/network_blocks.py:22-26 as down_blocks[0]._op[0], components/pulpo.py:26): nobody needs its data gradient, so
 * the gradient dy of its pre-norm tensor has ONE reader - this weight gradient.  The kernel forms dy = batch_norm_backward(leaky_relu_backward(dz))
 * per element while it stages its tiles (pulpo_bn_lrelu_bwd_apply's arithmetic from the same coefficient block and totals), so the pass that
 * writes dy is not run.  part2: pulpo_conv3d_k3_wgrad_bn_rows() rows of Cout floats - the column sums of dy (conv-bias gradient partials).
 * Cin <= 4, Cout % 4 == 0; dz fp32 or bf16 (dz_dt), y fp32, channels-last.  scratch / dw / accumulate as pulpo_conv3d_k3_wgrad. */
int pulpo_conv3d_k3_wgrad_bn_rows(int B, int D, int H, int W, int Cout);
int pulpo_conv3d_k3_wgrad_bn(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const void* dz, int dz_dt, int64_t dz_bs, int64_t dz_ps,
                             const float* y, int64_t y_bs, int64_t y_ps, const float* coef, const double* totd, float slope, float* dw,
                             int accumulate, float* scratch, float* part2, int B, int D, int H, int W, int Cin, int Cout, void* stream);

/* ------------------------------------------------------------------------- BatchNorm backward of a POOLED ConvUnit output without its gradient tensor (since ABI 4)
 * DownPath pools the output z of every level's last ConvUnit (components/pulpo.py:58: avg_pool3d, ceil mode) and may also use it as a skip
 * connection.  dz = (skip gradient +) avg_pool3d_backward(gradient of the pooled tensor) is formed per element inside BOTH passes of the unit's
 * batch_norm_backward + leaky_relu_backward: pulpo_avgpool2_bwd_bnred_t with gin == NULL (first pass) and this entry point (second pass). */
int pulpo_bn_lrelu_bwd_apply_pooled_t(const void* gout, int64_t gops, const void* add /*nullable*/, int64_t aps, int g_dt, const void* y, int y_dt,
                                      int64_t yps, const float* coef, const double* totd, void* dy /* y's dtype */, int64_t dyps, float slope,
                                      float* partial2, int B, int D, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------------- CHANNEL-BLOCKED gradient of the pre-norm tensor (since ABI 5)
 * Replaces nothing in the reference: inside aten::native_batch_norm_backward -> aten::convolution_backward (src/network_blocks.py:23-25) the gradient
 * dy of a ConvUnit's pre-norm tensor has exactly two readers, the unit's data- and weight-gradient convolution, so its layout is this library's own
 * business.  Channels-last, a staging item of the F(2x2x2,3x3x3) data-gradient kernel gathers 32 useful bytes from each of four 128-byte voxel
 * lines per 8-channel chunk; blocked - element (b, voxel v, channel c) at  p + b * bs + (c / 8) * kb + v * ps + c % 8  floats with ps = 8,
 * bs = D*H*W * 8, kb = B * D*H*W * 8, i.e. [C / 8][B][D][H][W][8] - the four taps are 128 consecutive bytes (32 -> 32 at 160^3: 0.94 -> 0.80 ms,
 * profiles/r5_blocked_probe.txt).  kb = 8 with ps = the row pitch addresses an ordinary channels-last tensor through the same entry points.
 *  - pulpo_bn_lrelu_bwd_apply_kb_t / _pooled_kb_t: the second BatchNorm-backward pass (pulpo_bn_lrelu_bwd_apply_t / _pooled_t) writing dy blocked;
 *    fp32 y / dy, C % 8 == 0.
 *  The same holds for the activation z BETWEEN two ConvUnits of a ConvSequence (src/network_blocks.py:40-46: read by the next unit's convolution,
 *  forward and weight gradient, and by nothing else) and for its gradient dz:
 *  - pulpo_bn_lrelu_apply_kb: the BatchNorm + LeakyReLU pass writing z blocked; the *_kb convolution entries take it (in_kb) and write the gradient dz
 *    blocked (out_kb); pulpo_bn_lrelu_bwd_apply_kb_t (dzkb) and, for the input layer, pulpo_conv3d_k3_wgrad_bn_kb (dz_kb) read it.
 *  - pulpo_conv3d_k3_fwd_wino3_kb / pulpo_conv3d_k3_dgrad_wino3_bnred_kb: pulpo_conv3d_k3_fwd_wino3 / _dgrad_wino3_bnred on a blocked operand
 *    (result channels-last); same shape contract (pulpo_conv3d_k3_algo == 3).
 *  - pulpo_conv3d_k3_wgrad_kb: pulpo_conv3d_k3_wgrad (slabs == NULL) / pulpo_conv3d_k3_wgrad_det on a blocked dy; shapes with
 *    pulpo_conv3d_k3_wgrad_algo(...) == 3 only (the entry point refuses others). */
int pulpo_bn_lrelu_apply_kb(const float* y, int64_t yps, float* z, int64_t zps, int64_t zkb, const float* coef, int64_t npix, int C, float slope, void* stream);
int pulpo_bn_lrelu_bwd_apply_kb_t(const void* dz, int dz_dt, int64_t dzps, int64_t dzkb, const float* y, int64_t yps, const float* coef, const double* totd,
                                  float* dy, int64_t dyps, int64_t dykb, int64_t npix, int C, float slope, float* partial2, void* stream);
int pulpo_conv3d_k3_wgrad_bn_kb(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dz, int64_t dz_bs, int64_t dz_ps, int64_t dz_kb,
                                const float* y, int64_t y_bs, int64_t y_ps, const float* coef, const double* totd, float slope, float* dw, int accumulate,
                                float* scratch, float* part2, int B, int D, int H, int W, int Cin, int Cout, void* stream);
int pulpo_bn_lrelu_bwd_apply_pooled_kb_t(const void* gout, int64_t gops, const void* add /*nullable*/, int64_t aps, int g_dt, const float* y, int64_t yps,
                                         const float* coef, const double* totd, float* dy, int64_t dyps, int64_t dykb, float slope, float* partial2,
                                         int B, int D, int H, int W, int C, void* stream);
int pulpo_conv3d_k3_fwd_wino3_kb(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_kb, const float* wp, const float* bias, const float* coef,
                                 float slope, float* out, int64_t out_bs, int64_t out_ps, int64_t out_kb, float* stats, int B, int D, int H, int W, int K,
                                 int N, void* stream);
int pulpo_conv3d_k3_dgrad_wino3_bnred_kb(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_kb, const float* wp, float* out, int64_t out_bs,
                                         int64_t out_ps, int64_t out_kb, const float* bn_y, int64_t bn_y_bs, int64_t bn_y_ps, const float* bn_coef, float slope,
                                         float* part, int B, int D, int H, int W, int K, int N, void* stream);
int pulpo_conv3d_k3_wgrad_kb(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_kb, const float* dy, int64_t dy_bs, int64_t dy_ps, int64_t dy_kb,
                             float* dw, int accumulate, float* scratch, float* slabs /*nullable*/, int nslab, int B, int D, int H, int W, int Cin, int Cout,
                             void* stream);

/* ------------------------------------------------------------------------- DETERMINISTIC forms of the backward kernels that add with float atomics (since ABI 4)
 * The reference's CPU backward is run-to-run deterministic; the plain entry points above add the weight-gradient partial sums
 * of concurrent workgroups (aten::convolution_backward, src/network_blocks.py:23) and the image-gradient scatter of grid_sampler_3d_backward
 * (src/network_blocks.py:120, :175) with float atomics, i.e. in arrival order: gradients differ in their last bits from run to run.  These forms
 * give bit-identical results on every run of the same build (pulpo_amd.ops.set_deterministic / PULPO_DETERMINISTIC=1 routes the operators here):
 *  - weight gradient: the same kernels, but workgroups that share a (ci tile, co tile) accumulate into separate zeroed copies ("slabs") of the
 *    packed scratch, which an ordered pass adds up.  slabs: nslab * pulpo_conv3d_k3_wgrad_scratch_floats(Cin, Cout) floats, 16-byte aligned, any
 *    content; nslab = pulpo_conv3d_k3_wgrad_det_slabs(Cin, Cout) (a smaller count is accepted and caps the spatial splits of the grid).
 *  - warp / VecInt backward: every scattered contribution enters a 64-bit fixed-point accumulator (integer adds commute) scaled by the largest
 *    |upstream gradient| of the pass (2^46 units per that maximum), a second pass converts back.  ws: caller-owned workspace of the queried
 *    size, any content.
 *  - F.interpolate backward at ratios other than the exact x2 (src/network_blocks.py:146-150 with other factors, losses.py:313): a gather in fixed
 *    order instead of the scatter. */
int pulpo_conv3d_k3_wgrad_det_slabs(int Cin, int Cout);
int pulpo_conv3d_k3_wgrad_det(const float* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const float* dy, int64_t dy_bs, int64_t dy_ps,
                              int64_t dy_cs, float* dw, int accumulate, float* scratch, float* slabs, int nslab, int B, int D, int H, int W,
                              int Cin, int Cout, void* stream);
int pulpo_conv3d_k3_wgrad_bf16_det_t(const void* in, int64_t in_bs, int64_t in_ps, int64_t in_cs, const void* dy, int64_t dy_bs, int64_t dy_ps,
                                     int64_t dy_cs, int dt, float* dw, int accumulate, float* scratch, float* slabs, int nslab, int B, int D,
                                     int H, int W, int Cin, int Cout, void* stream);
int pulpo_resize_trilinear_scaled_bwd_det(const float* gout, float* gin, int64_t nplanes, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                          float scale_d, float scale_h, float scale_w, float mult, void* stream);
size_t pulpo_warp3d_bwd_det_ws_bytes(int B, int C, int Di, int Hi, int Wi);
int pulpo_warp3d_bwd_det(const float* df, const float* img, const float* gout, float* gdf /*nullable*/, float* gimg /*nullable*/,
                         void* ws /*nullable when gimg is*/, int B, int C, int Dg, int Hg, int Wg, int Di, int Hi, int Wi, void* stream);
size_t pulpo_vecint_bwd_det_ws_bytes(int B, int D, int H, int W, int nsteps);
int pulpo_vecint_bwd_det(const float* work, const float* gout, float* gin, void* ws /*nullable if the query is 0*/, int B, int D, int H, int W,
                         int nsteps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PULPO_HIP_H */
