/*
 * pcrl_hip.h -- C ABI of libpcrl_hip.so: the MI355X (gfx950) native operator layer of the
 * PCRLv2 3D pre-training hot path.
 *
 * The reference (RL4M/PCRLv2) has no FFI of its own: its operator boundary is torch.nn /
 * autograd.  Each entry point below replaces the ATen operator that the cited
 * reference line dispatches.  Paths are relative to the reference repository.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch types.  All pointers are DEVICE pointers.
 *   - Activations are NDHWC ("channels last 3d"): element (n,d,h,w,c) at (((n*D+d)*H+h)*W+w)*C+c.
 *     `dtype` selects the activation / packed-weight storage type: PCRL_F32 or PCRL_BF16.
 *     Accumulation, statistics, 1-channel maps, head tensors, gradients of parameters and the
 *     parameters themselves ("*_ref" pointers, reference layout) are always float32.
 *   - Every call is asynchronous on `stream`; nothing synchronises, allocates or frees.
 *     The caller owns all memory, including workspaces (sizes from the *_ws_bytes helpers).
 *   - Returns 0 on success, a negative PCRL_E* code on failure; pcrl_last_error() returns a
 *     thread-local message.  Nothing throws across the ABI.  Re-entrant: the entry points keep no mutable state of their own;
 *     the only process-wide state is the set of kernel-selection switches of the "Test hooks" section at the end of this
 *     header (std::atomic<int>, defaults = the product path; tests and probes flip them to A/B the kernels behind one entry point).
 *   - Reductions are deterministic (two-stage, fixed order; no floating-point atomics).
 */
#ifndef PCRL_HIP_H
#define PCRL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pcrl_stream_t; /* hipStream_t */

enum { PCRL_F32 = 0, PCRL_BF16 = 1 };
enum { PCRL_ACT_NONE = 0, PCRL_ACT_RELU = 1, PCRL_ACT_SIGMOID = 2, PCRL_ACT_SILU = 3 /* optional extra, not used by the reference path */,
       PCRL_ACT_ELU = 4 /* LUConv(act='elu'), models/pcrlv2_model_3d.py:24-25: a constructor variant train_3d.py never instantiates */ };
enum { PCRL_OK = 0, PCRL_EINVAL = -1, PCRL_ELAUNCH = -2, PCRL_EWORKSPACE = -3 };

#define PCRL_CONV_BM 128 /* rows (voxels) per conv tile == rows per BN-statistics partial */

const char* pcrl_version(void);
const char* pcrl_last_error(void);
/* zero-fill `bytes` bytes at p on `stream` (hipMemsetAsync; aten::zero_ on a freshly allocated gradient tensor) */
int pcrl_zero(void* p, size_t bytes, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Weight packing (once per optimizer step).  Reference layouts: Conv3d weight [Co][Ci][3][3][3]
 * (models/pcrlv2_model_3d.py:9), ConvTranspose3d weight [Ci][Co][2][2][2] (:52).
 *   conv3 : w_fwd[co][t][ci]           (t = kd*9+kh*3+kw)         -- B^T operand of the forward GEMM
 *           w_dgrad[ci][26-t][co]                                   -- B^T operand of the data-gradient GEMM
 *   convT : w_fwd[t][co][ci]           (t = i*4+j*2+k)
 *           w_dgrad[ci][t][co]
 * Either output pointer may be NULL. */
int pcrl_pack_conv3_weight(const float* w_ref, void* w_fwd, void* w_dgrad, int Co, int Ci, int dtype, pcrl_stream_t stream);
int pcrl_pack_convt_weight(const float* w_ref, void* w_fwd, void* w_dgrad, int Ci, int Co, int dtype, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * 3x3x3 convolution, pad 1, stride 1 -- aten::convolution at pcrlv2_model_3d.py:9,33 (LUConv.conv1).
 * Implicit GEMM on MFMA: M = N*D*H*W voxels, N = Co, K = 27*Ci.  Ci % 32 == 0, Co % 32 == 0.
 *   y[m][co] = bias[co] + sum_{t,ci} x[m+delta_t][ci] * wp[co][t][ci]      (zero outside the volume)
 * `stats_partial` (optional): [pcrl_conv3d_k3_stats_rows(...)][Co][2] float; row r receives (sum y, sum y^2) of
 * output tile r, taken from the fp32 accumulators, for the training-mode BatchNorm that follows (:12,33).
 * Three kernels sit behind this entry point: two LDS-halo "brick" kernels (bf16, D%4 == 0, H%8 == 0: 4x8x16-voxel tiles when
 * W%16 == 0, 4x8x8-voxel tiles when W%8 == 0) and a gather kernel (any shape, both dtypes: 128-voxel tiles); the helper tells
 * which tiling the given shape gets.
 * The data gradient (aten::convolution_backward, input half) is the same call with wp = w_dgrad,
 * Ci/Co exchanged, bias = NULL, stats_partial = NULL. */
int64_t pcrl_conv3d_k3_stats_rows(int N, int D, int H, int W, int Ci, int Co, int dtype);
int64_t pcrl_conv3d_k3_fwd_kernel(int N, int D, int H, int W, int Ci, int Co, int dtype);   /* informational: 2 wide-brick, 1 brick, 0 gather kernel */
int pcrl_conv3d_k3_fwd(const void* x, const void* wp, const float* bias, void* y, float* stats_partial,
                       int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream);
/* Same operation with a caller-provided workspace: volumes too small to fill the chip with 128-voxel tiles (the 8x8x4 bottleneck
 * level at b=32; the 4^3 / 2^3 levels of the 16^3 local views, train_3d.py:118-121) are then split along K = 27 taps x Ci/32
 * into float partial sums in `ws`, combined in fixed order by a second pass that also adds the bias, rounds and emits
 * `stats_partial` (same layout and row count as the one-pass kernel).  `pcrl_conv3d_k3_fwd_ws_bytes` returns the size needed
 * (0: the shape is not split; `ws` may then be NULL). */
int64_t pcrl_conv3d_k3_fwd_ws_bytes(int N, int D, int H, int W, int Ci, int Co, int dtype);
int pcrl_conv3d_k3_fwd_ws(const void* x, const void* wp, const float* bias, void* y, float* stats_partial, void* ws, int64_t ws_bytes,
                          int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream);

/* Data gradient of a 3x3x3 convolution (convolution_backward(input) of LUConv.conv1, models/pcrlv2_model_3d.py:9,33) WITH the first pass of the
 * BatchNorm backward of the layer BELOW in its epilogue -- for the case that layer's activation has this convolution as its only consumer
 * (ops.0 -> ops.1 inside nn.Sequential(LUConv, LUConv), :37-45): dx IS the gradient of a = relu(scale * bn_y + shift), and rows of
 * (sum dz, sum dz * xhat), dz = [scale * bn_y + shift > 0] * dx (as stored), xhat = (bn_y - mean) * rstd, come out of the convolution's tiles
 * instead of a separate pass over dx and bn_y (pcrl_bn_act_bwd_reduce; aten::threshold_backward + native_batch_norm_backward's reduction).
 * dy: [N][D][H][W][Ci]; wp_dgrad: the packed data-gradient weights of pcrl_conv3d_k3_fwd; dx, bn_y: [N][D][H][W][Co]; scale .. rstd: Co floats
 * (pcrl_bn_finalize's outputs for the layer below); partial: [rows][Co][2] float for pcrl_bn_bwd_finalize(partial, rows, Co, count = N D H W, ...).
 * `pcrl_conv3d_k3_dgrad_bnred_rows` returns the row count, or 0 when the shape / activation / dtype has no such kernel (wide-brick bf16 shapes
 * behind ReLU only): the caller then runs pcrl_conv3d_k3_fwd_ws and pcrl_bn_act_bwd_reduce as two passes. */
int64_t pcrl_conv3d_k3_dgrad_bnred_rows(int N, int D, int H, int W, int Ci, int Co, int act, int dtype);
int pcrl_conv3d_k3_dgrad_bnred(const void* dy, const void* wp_dgrad, void* dx, const void* bn_y, const float* scale, const float* shift,
                               const float* mean, const float* rstd, float* partial, int N, int D, int H, int W, int Ci, int Co, int act,
                               int dtype, pcrl_stream_t stream);

/* Weight gradient (aten::convolution_backward, weight half).  dw_ref[co][ci][27] float32, reference layout.
 * Split-K over voxels with a fixed-order second pass.  ws: pcrl_conv3d_k3_wgrad_ws_bytes(). */
size_t pcrl_conv3d_k3_wgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Co);
int pcrl_conv3d_k3_wgrad(const void* x, const void* dy, float* dw_ref, void* ws, size_t ws_bytes,
                         int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream);

/* First layer, Ci == 1 (pcrlv2_model_3d.py:101 -> down_tr64.ops.0, K = 27: HBM-bound, no MFMA).
 * x: float32 scalar field [M]; w_ref [Co][1][27] float32; y: dtype [M][Co]; Co in {16, 32, 64}.
 * `stats_partial`: [pcrl_conv3d_k3_c1_stats_rows(...)][Co][2] (bf16 on D%4 == 0, H%8 == 0, W%8 == 0 volumes: an MFMA brick kernel, one
 * row per 4x8x8 brick, x and w enter the MFMA as bf16; otherwise one row per 128 voxels, float32 FMAs).
 * The weight gradient is an MFMA GEMM dy^T . im2col(x) (im2col built in LDS on brick volumes, else staged in the workspace). */
int64_t pcrl_conv3d_k3_c1_stats_rows(int N, int D, int H, int W, int Co, int dtype);
int pcrl_conv3d_k3_c1_fwd(const float* x, const float* w_ref, const float* bias, void* y, float* stats_partial,
                          int N, int D, int H, int W, int Co, int dtype, pcrl_stream_t stream);
size_t pcrl_conv3d_k3_c1_wgrad_ws_bytes(int N, int D, int H, int W, int Co);
int pcrl_conv3d_k3_c1_wgrad(const float* x, const void* dy, float* dw_ref, void* ws, size_t ws_bytes,
                            int N, int D, int H, int W, int Co, int dtype, pcrl_stream_t stream);

/* Convolutions with ONE output channel: deep-supervision head conv3x3x3 C->1 (pcrlv2_model_3d.py:60,71)
 * and OutputTransition.final_conv 1x1x1 64->1 (:78).  taps = 27 or 1.  y, dy: float32 [M].
 * w_ref: [1][C][taps] float32.  `stats_partial`: [pcrl_conv3d_to1_stats_rows(...)][1][2] or NULL.  fwd, 27 taps, three kernels:
 * an LDS-halo brick kernel (bf16, D%4 == 0, H%8 == 0, W%8 == 0, C%32 == 0: z = x.w for every halo voxel on MFMA, gathered per
 * output voxel inside the block; one statistics row per 4x8x8 brick); else a pointwise MFMA product z[t][m] = sum_c x[m][c] w[c][t]
 * (x read once) + a shifted sum of the 27 planes, staged in `ws` (pcrl_conv3d_to1_fwd_ws_bytes; one row per 1024 voxels);
 * ws = NULL falls back to the direct 27-tap gather kernel.
 * dgrad: dx[m][c] = add_src[m][c] + sum_t dy[m-delta_t] * w[c][t]; add_src may be NULL (zero), dx itself
 * (in-place accumulation) or another tensor of the same shape (an upstream gradient to fold in). */
int64_t pcrl_conv3d_to1_stats_rows(int N, int D, int H, int W, int C, int taps, int dtype);
size_t pcrl_conv3d_to1_fwd_ws_bytes(int N, int D, int H, int W, int C, int taps);
int pcrl_conv3d_to1_fwd(const void* x, const float* w_ref, const float* bias, float* y, float* stats_partial,
                        void* ws, size_t ws_bytes, int N, int D, int H, int W, int C, int taps, int dtype,
                        pcrl_stream_t stream);
int pcrl_conv3d_to1_dgrad(const float* dy, const float* w_ref, const void* add_src, void* dx,
                          int N, int D, int H, int W, int C, int taps, int dtype, pcrl_stream_t stream);
size_t pcrl_conv3d_to1_wgrad_ws_bytes(int N, int D, int H, int W, int C, int taps);
int pcrl_conv3d_to1_wgrad(const void* x, const float* dy, float* dw_ref, float* db, void* ws, size_t ws_bytes,
                          int N, int D, int H, int W, int C, int taps, int dtype, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * ConvTranspose3d kernel 2 stride 2 -- aten::convolution (transposed) at pcrlv2_model_3d.py:52,64.
 * N,D,H,W are the INPUT dims; the output is [N][2D][2H][2W][Co].  Eight independent GEMMs with a
 * scatter store (fwd) / one K = 8*Co gather GEMM (dgrad).  Ci % 32 == 0, Co % 32 == 0. */
int pcrl_convt3d_k2s2_fwd(const void* x, const void* wp_fwd, const float* bias, void* y,
                          int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream);
int pcrl_convt3d_k2s2_dgrad(const void* dy, const void* wp_dgrad, void* dx,
                            int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream);
size_t pcrl_convt3d_k2s2_wgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Co);
int pcrl_convt3d_k2s2_wgrad(const void* x, const void* dy, float* dw_ref, void* ws, size_t ws_bytes,
                            int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream);

/* Per-channel column sum of a [M][C] activation (bias gradients): out[c] = sum_m v[m][c].
 * ws: pcrl_colsum_ws_bytes(M, C). */
size_t pcrl_colsum_ws_bytes(int64_t M, int C);
int pcrl_colsum(const void* v, float* out, void* ws, size_t ws_bytes, int64_t M, int C, int dtype, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Training-mode BatchNorm3d + activation -- aten::native_batch_norm(+_backward), relu_, sigmoid
 * at pcrlv2_model_3d.py:12,21,27,33.  Statistics arrive as per-tile partials from the conv epilogue.
 *   finalize : mean, biased var over `count` elements per channel (fp64 fixed-order reduction);
 *              running_mean/var updated in place (momentum, unbiased var), num_batches_tracked is the
 *              caller's; emits mean, rstd and the fused apply coefficients scale = gamma*rstd,
 *              shift = beta - mean*scale.
 *   apply    : a = act(scale[c]*y + shift[c])
 *   bwd      : dz = da*act'(z);  reduce -> partial (sum dz, sum dz*xhat) per 1024-row tile;
 *              bwd_finalize -> dgamma, dbeta and coefficients k1,kB,kA;  bwd_apply: dy = k1*dz + kB*y + kA. */
int pcrl_bn_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps,
                     float* mean, float* rstd, float* scale, float* shift, pcrl_stream_t stream);
int pcrl_bn_act_apply(const void* y, void* a, const float* scale, const float* shift,
                      int64_t M, int C, int act, int dtype, pcrl_stream_t stream);
/* rows of `partial` for bwd_reduce: ceil(M / tile), tile = 1024 rows for M >= 2^20, halved down to 32 for smaller M */
int64_t pcrl_bn_bwd_partial_rows(int64_t M);
int pcrl_bn_act_bwd_reduce(const void* da, const void* y, const float* scale, const float* shift,
                           const float* mean, const float* rstd, float* partial,
                           int64_t M, int C, int act, int dtype, pcrl_stream_t stream);
/* The two passes for a LUConv whose activation is consumed through nn.MaxPool3d(2) ONLY (pcrlv2_model_3d.py:115-117: the second LUConv
 * of an encoder stage; the skip tensors are never used): max_pool3d_backward folded in.  dp: dtype [N][D/2][H/2][W/2][C] gradient of the
 * POOLED tensor; y: dtype [N][D][H][W][C] the layer's pre-normalisation output.  The activation is recomputed from y exactly as the
 * forward stored it; the first maximum of a window in scan order (d, h, w) receives the gradient, NaN wins (aten's rule).  The
 * full-resolution gradient of the activation is never materialised.  Available when pcrl_bn_act_bwd_pool_ok() != 0 (even D, H, W and
 * pcrl_bn_act_bwd_rowadd_ok(C, dtype)); partial: [pcrl_bn_act_bwd_pool_partial_rows(N, D, H, W)][C][2] -> pcrl_bn_bwd_finalize with
 * count = N*D*H*W. */
int64_t pcrl_bn_act_bwd_pool_ok(int D, int H, int W, int C, int dtype);   /* 1 / 0 */
/* forward of the same pair: a = act(scale*y + shift) and p = MaxPool3d(2)(a) in one pass (same availability).  a may be NULL: only p is stored --
   the encoder stages' unpooled outputs have no reader in a training step (models/pcrlv2_model_3d.py:114-117 stashes them as attributes, nothing
   consumes them), and the backward of the pair reads y, not a: 47 % of the pass's bytes */
int pcrl_bn_act_apply_pool(const void* y, void* a, void* p, const float* scale, const float* shift, int N, int D, int H, int W, int C,
                           int act, int dtype, pcrl_stream_t stream);
int64_t pcrl_bn_act_bwd_pool_partial_rows(int N, int D, int H, int W);
int pcrl_bn_act_bwd_reduce_pool(const void* dp, const void* y, const float* scale, const float* shift, const float* mean,
                                const float* rstd, float* partial, int N, int D, int H, int W, int C, int act, int dtype,
                                pcrl_stream_t stream);
int pcrl_bn_act_bwd_apply_pool(const void* dp, const void* y, void* dy, const float* scale, const float* shift, const float* k1,
                               const float* kB, const float* kA, int N, int D, int H, int W, int C, int act, int dtype,
                               pcrl_stream_t stream);
int pcrl_bn_bwd_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* mean,
                         const float* rstd, float* dgamma, float* dbeta, float* k1, float* kB, float* kA,
                         pcrl_stream_t stream);
int pcrl_bn_act_bwd_apply(const void* da, const void* y, void* dy, const float* scale, const float* shift,
                          const float* k1, const float* kB, const float* kA,
                          int64_t M, int C, int act, int dtype, pcrl_stream_t stream);
/* The same two passes with a per-(sample, channel) term folded into the incoming gradient: da_eff[n][s][c] = da[n][s][c] + row_g[n][c] / S
 * (M = N * S rows; da may be NULL: the term alone).  This is the global-average-pool branch of UpTransition.forward
 * (pcrlv2_model_3d.py:67; autograd: adaptive_avg_pool3d_backward + the add of the two gradients of x) without materialising the
 * broadcast (pcrl_gap_bwd).  Available when pcrl_bn_act_bwd_rowadd_ok(C, dtype) != 0 (C a multiple of the 16-byte vector that
 * divides 256 vectors). */
int64_t pcrl_bn_act_bwd_rowadd_ok(int C, int dtype);   /* 1 / 0 */
int pcrl_bn_act_bwd_reduce_rowadd(const void* da, const float* row_g, int N, int64_t S, const void* y, const float* scale,
                                  const float* shift, const float* mean, const float* rstd, float* partial,
                                  int64_t M, int C, int act, int dtype, pcrl_stream_t stream);
int pcrl_bn_act_bwd_apply_rowadd(const void* da, const float* row_g, int N, int64_t S, const void* y, void* dy, const float* scale,
                                 const float* shift, const float* k1, const float* kB, const float* kA,
                                 int64_t M, int C, int act, int dtype, pcrl_stream_t stream);

/* The same two passes when the incoming gradient is a SUM of up to three parts: da + da2 + row_g[n][c] / S (each may be NULL, at least
 * one given; N, S only read with row_g).  Replaces the aten::add launches autograd issues where one activation has several consumers:
 * a 2D decoder block's output feeds the next block, its own deep-supervision head and the pooled projection head
 * (models/pcrlv2_model.py:119-127), a BasicBlock's input feeds conv1 and the identity branch.  Same availability as the row term. */
int pcrl_bn_act_bwd_reduce_sum(const void* da, const void* da2, const float* row_g, int N, int64_t S, const void* y, const float* scale,
                               const float* shift, const float* mean, const float* rstd, float* partial,
                               int64_t M, int C, int act, int dtype, pcrl_stream_t stream);
int pcrl_bn_act_bwd_apply_sum(const void* da, const void* da2, const float* row_g, int N, int64_t S, const void* y, void* dy,
                              const float* scale, const float* shift, const float* k1, const float* kB, const float* kA,
                              int64_t M, int C, int act, int dtype, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * MaxPool3d(2) -- aten::max_pool3d_with_indices(+backward) at pcrlv2_model_3d.py:100,115-117.
 * N,D,H,W are the INPUT dims (even).  Backward recomputes the argmax from the saved input
 * (first maximum in d,h,w scan order takes the gradient, like ATen). */
int pcrl_maxpool3d_2_fwd(const void* x, void* y, int N, int D, int H, int W, int C, int dtype, pcrl_stream_t stream);
int pcrl_maxpool3d_2_bwd(const void* x, const void* dy, void* dx, int N, int D, int H, int W, int C, int dtype, pcrl_stream_t stream);

/* Global average pool -- aten::adaptive_avg_pool3d -> (1,1,1) at pcrlv2_model_3d.py:67.
 * fwd: g[n][c] = mean_s a[n][s][c] (float32 out).  bwd: da[n][s][c] = add_src[n][s][c] + dg[n][c]/S
 * (add_src: NULL, da itself, or another tensor, as for pcrl_conv3d_to1_dgrad). */
size_t pcrl_gap_ws_bytes(int N, int64_t S, int C);
int pcrl_gap_fwd(const void* a, float* g, void* ws, size_t ws_bytes, int N, int64_t S, int C, int dtype, pcrl_stream_t stream);
int pcrl_gap_bwd(const float* dg, const void* add_src, void* da, int N, int64_t S, int C, int dtype, pcrl_stream_t stream);
/* pcrl_bn_act_apply and pcrl_gap_fwd of its result in one pass over y (UpTransition: ops.1's activation feeds the pool, :64-67):
 * a = act(scale*y + shift) (dtype [N][S][C]), g[n][c] = mean_s a (the stored, rounded values).  ws: pcrl_gap_ws_bytes(N, S, C).
 * Available when pcrl_bn_act_bwd_rowadd_ok(C, dtype) != 0. */
int pcrl_bn_act_apply_gap(const void* y, void* a, float* g, const float* scale, const float* shift, void* ws, size_t ws_bytes,
                          int N, int64_t S, int C, int act, int dtype, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Projection / predictor heads on [rows][C] float32 -- BatchNorm1d, Linear, ReLU at
 * pcrlv2_model_3d.py:55-59,69-70.  Tiny (latency-bound) kernels. */
int pcrl_bn1d_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                  float momentum, float eps, float* mean, float* rstd, int rows, int C, int relu, pcrl_stream_t stream);
int pcrl_bn1d_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* mean, const float* rstd,
                  float* dx, float* dgamma, float* dbeta, int rows, int C, int relu, pcrl_stream_t stream);
int pcrl_linear_fwd(const float* x, const float* w, const float* b, float* y, int rows, int Cin, int Cout, pcrl_stream_t stream);
int pcrl_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                    int rows, int Cin, int Cout, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Trilinear upsampling of 1-channel float32 maps, align_corners=False, integer scale --
 * aten::upsample_trilinear3d(+backward) at pcrlv2_model_3d.py:125-126.  N,D,H,W: INPUT dims. */
int pcrl_upsample_trilinear_fwd(const float* x, float* y, int N, int D, int H, int W, int scale, pcrl_stream_t stream);
int pcrl_upsample_trilinear_bwd(const float* dy, float* dx, int N, int D, int H, int W, int scale, pcrl_stream_t stream);

/* Elementwise sigmoid on float32 maps (OutputTransition, pcrlv2_model_3d.py:79,82) and its backward
 * dpre = dout * out * (1 - out). */
int pcrl_sigmoid_fwd(const float* x, float* y, int64_t n, pcrl_stream_t stream);
int pcrl_sigmoid_bwd(const float* dout, const float* out, float* dpre, int64_t n, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Losses -- aten::mse_loss (train_3d.py:56,135,137) and aten::cosine_similarity(dim=1, eps=1e-8).mean()
 * (train_3d.py:57,90-91).  ws: pcrl_reduce_ws_bytes(n). */
size_t pcrl_reduce_ws_bytes(int64_t n);
int pcrl_mse_fwd(const float* p, const float* gt, float* loss, void* ws, size_t ws_bytes, int64_t n, pcrl_stream_t stream);
int pcrl_mse_bwd(const float* p, const float* gt, const float* dloss, float* dp, int64_t n, pcrl_stream_t stream);
/* out[0] = mean_r cos(x_r, y_r); saved[r][3] = (dot, |x|, |y|).  bwd: gradient w.r.t. x only (y is detached). */
int pcrl_cosine_mean_fwd(const float* x, const float* y, float* out, float* saved, int rows, int C, float eps, pcrl_stream_t stream);
int pcrl_cosine_mean_bwd(const float* x, const float* y, const float* saved, const float* dout, float* dx,
                         int rows, int C, float eps, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * OPTIONAL EXTRA, not part of the reference: NT-Xent (SimCLR) contrastive loss over z = [z1; z2], R = 2N
 * rows (row i's positive is row (i + N) mod R), temperature tau:
 *   zn = z / max(|z|, eps);  loss = mean_i ( logsumexp_{k != i} zn_i.zn_k / tau  -  zn_i.zn_pos(i) / tau ).
 * fwd leaves zn, the softmax and the norms in `ws` (pcrl_ntxent_ws_bytes) for bwd, which must get the same workspace. */
size_t pcrl_ntxent_ws_bytes(int R, int C);
int pcrl_ntxent_fwd(const float* z, float* loss, void* ws, size_t ws_bytes, int R, int C, float tau, float eps, pcrl_stream_t stream);
int pcrl_ntxent_bwd(const float* z, const float* dloss, float* dz, void* ws, size_t ws_bytes, int R, int C, float tau, float eps,
                    pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * OPTIONAL EXTRA, not part of the reference: GroupNorm(G) + activation on NDHWC activations, built from the
 * BatchNorm streaming kernels applied per sample with per-(sample, channel) coefficients:
 *   gn_stats:        partial[n][tile][c][2] = (sum, sum^2) of y over the tile's voxels        (tiles = pcrl_gn_stats_tiles(S))
 *   gn_finalize:     per (n, group): mean, rstd (biased variance, eps) -> mean_c, rstd_c, scale, shift as [N][C] arrays
 *   apply:           pcrl_bn_act_apply on sample n with scale + n*C, shift + n*C   (act = PCRL_ACT_SILU for GroupNorm+SiLU)
 *   backward:        pcrl_bn_act_bwd_reduce per sample (mean_c/rstd_c rows) -> gn_bwd_finalize (k1, kB, kA as [N][C]; per-sample
 *                    dgamma/dbeta parts [N][C]) -> pcrl_bn_act_bwd_apply per sample. */
int64_t pcrl_gn_stats_tiles(int64_t S);
int pcrl_gn_stats(const void* y, float* partial, int N, int64_t S, int C, int dtype, pcrl_stream_t stream);
int pcrl_gn_finalize(const float* partial, int tiles, int N, int64_t S, int C, int G, const float* gamma, const float* beta, float eps,
                     float* mean_c, float* rstd_c, float* scale, float* shift, pcrl_stream_t stream);
int pcrl_gn_bwd_finalize(const float* partial_b, int rows_b, int N, int64_t S, int C, int G, const float* gamma, const float* mean_c,
                         const float* rstd_c, float* k1, float* kB, float* kA, float* dgamma_n, float* dbeta_n, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * LUConv constructor variants (models/pcrlv2_model_3d.py:15-16,22-25; accepted by the reference's constructor, never instantiated by
 * train_3d.py:45): act='elu' is PCRL_ACT_ELU of the fused BatchNorm kernels; norm='in' (InstanceNorm3d, affine) is the GroupNorm
 * machinery above with G == C; act='prelu' (nn.PReLU(out_chan), aten::prelu / prelu_backward) is a streaming pass behind the
 * normalisation, which then runs with PCRL_ACT_NONE:
 *   prelu_fwd:  a[m][c] = z > 0 ? z : w[c] * z                                   (z, a: [M][C] activations; w: float32 [C])
 *   prelu_bwd:  dz = da * (z > 0 ? 1 : w[c]);  partial[tile][c] = sum_m da * z * [z <= 0] over the tile's rows
 *               (pcrl_prelu_bwd_partial_rows(M) tiles; dw = column sums of `partial`, e.g. pcrl_colsum). */
int pcrl_prelu_fwd(const void* z, const float* w, void* a, int64_t M, int C, int dtype, pcrl_stream_t stream);
int64_t pcrl_prelu_bwd_partial_rows(int64_t M);
int pcrl_prelu_bwd(const void* da, const void* z, const float* w, void* dz, float* partial, int64_t M, int C, int dtype, pcrl_stream_t stream);

/* =======================================================================================
 * 2D path: the PCRLv2 ResNet-18 U-Net of models/pcrlv2_model.py:68-209 (decoder) and the smp/torchvision ResNet-18
 * encoder it wraps (pcrlv2_model.py:200).  Activations NHWC ([N][H][W][C], C-contiguous), float32 or bf16.
 *
 * Convolution KHxKW, stride 1|2, zero padding `pad`, optional fused nearest x2 upsample of the input (F.interpolate(scale_factor=2,
 * mode="nearest") at pcrlv2_model.py:114 followed by conv1): replaces aten::convolution / convolution_backward for nn.Conv2d.
 * Source channel counts must be powers of two >= 8: the caller zero-pads 3-channel tensors (image, 3-channel gradients) to 8.
 *   pack  : reference weight float32 [Co][Ci][KH][KW] -> K-contiguous rows in `dtype`; mode 0 = forward ([round32(Co)][Kpad],
 *           k = tap*CsP + ci), mode 1 = data gradient ([round32(Ci)][Kpad], k = tap*CsP + co); Kpad = round32(KH*KW*CsP);
 *           pcrl_conv2d_packed_elems(rows, taps, CsP) elements.
 *   fwd   : y[N][Ho][Wo][Co] (+bias), `out_f32` stores float32 whatever `dtype`; stats_partial = pcrl_conv2d_stats_rows(N,Ho,Wo)
 *           rows of [Co][2] (sum, sum^2) for the BatchNorm2d that follows, or NULL.  Hi/Wi = stored input dims (before `up`).
 *   dgrad : dx[N][Hi][Wi][Ci] from dy[N][Ho][Wo][CoP]  (for a fused-upsample forward Hi/Wi are the UPSAMPLED dims; follow with
 *           pcrl_upsample2d_nearest2_bwd)
 *   wgrad : dw float32 [CoP][Ci_out][KH][KW]; ws: pcrl_conv2d_wgrad_ws_bytes */
int64_t pcrl_conv2d_packed_elems(int rows, int taps, int CsP);
int pcrl_conv2d_pack(const float* w_ref, void* out, int Co, int Ci, int KH, int KW, int CsP, int mode, int dtype, pcrl_stream_t stream);
int64_t pcrl_conv2d_stats_rows(int N, int Ho, int Wo);     /* upper bound for any kernel: one row per 128 output pixels */
/* rows of statistics pcrl_conv2d_fwd WRITES for this geometry (depends on the kernel its dispatcher picks): allocate that many, pass the
 * count as `stats_rows` and to pcrl_bn_finalize; more rows may be passed -- the surplus is zero-filled */
int64_t pcrl_conv2d_fwd_stats_rows(int N, int Hi, int Wi, int CiP, int Co, int KH, int KW, int stride, int pad, int up, int out_f32, int dtype);
int pcrl_conv2d_fwd(const void* x, const void* wp, const float* bias, void* y, float* stats_partial, int64_t stats_rows, int N, int Hi, int Wi,
                    int CiP, int Co, int KH, int KW, int stride, int pad, int up, int out_f32, int dtype, pcrl_stream_t stream);
/* y may be NULL where pcrl_conv2d_fwd_stats_only_ok(...) == 1: the statistics rows only -- the 3x3 convolution of a deep-supervision head whose map
 * nothing reads (14 of the 15 heads of a step, pcrlv2_model.py:103-106 / train_2d.py:143-168) runs for its BatchNorm's running statistics alone; its
 * output (537 MB at 512 x 512 x 16 channels, b = 64) is not written.  The statistics are taken from the float32 accumulators either way. */
int64_t pcrl_conv2d_fwd_stats_only_ok(int N, int Hi, int Wi, int CiP, int Co, int KH, int KW, int stride, int pad, int up, int out_f32, int dtype);
/* data gradient of a 3x3 / stride 1 / pad 1 convolution that read its input through the nearest x2 upsample (decoder conv1,
 * models/pcrlv2_model.py:114) INCLUDING the upsample's backward (aten::convolution_backward's input gradient + aten::upsample_nearest2d_backward):
 * dx[N][Hc][Wc][Ci] = 2 x 2 block sums of the fine-resolution gradient, which is never stored.  dy: [N][2Hc][2Wc][CoP]; wp_dgrad as for
 * pcrl_conv2d_dgrad.  Only where pcrl_conv2d_dgrad_up_ok() != 0 (both channel counts <= 32, bf16, fine extents multiples of 8 x 32). */
/* which kernel the dispatcher runs for a geometry (no launch): 0 gather implicit GEMM, 1 LDS-halo brick kernel, 2 right-sized narrow kernel */
int64_t pcrl_conv2d_fwd_kind(int N, int Hi, int Wi, int CiP, int Co, int KH, int KW, int stride, int pad, int up, int out_f32, int dtype);
int64_t pcrl_conv2d_dgrad_kind(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int CoP, int KH, int KW, int stride, int pad, int dtype);
int64_t pcrl_conv2d_dgrad_up_ok(int N, int Hc, int Wc, int Ci, int CoP, int dtype);
int pcrl_conv2d_dgrad_up(const void* dy, const void* wp_dgrad, void* dx, int N, int Hc, int Wc, int Ci, int CoP, int dtype, pcrl_stream_t stream);
int pcrl_conv2d_dgrad(const void* dy, const void* wp_dgrad, void* dx, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int CoP, int KH,
                      int KW, int stride, int pad, int dtype, pcrl_stream_t stream);
/* 3x3 / stride 1 / pad 1 data gradient WITH the first pass of the BatchNorm backward of the layer below (the 2D counterpart of
 * pcrl_conv3d_k3_dgrad_bnred): conv2 of a torchvision BasicBlock under relu(bn1(conv1(x))), conv2 of a DecoderBlock under conv1's
 * BatchNorm + ReLU (models/pcrlv2_model.py:113-128) -- activations with one consumer.  dy: [N][H][W][CoP]; dx, bn_y: [N][H][W][Ci];
 * partial: [rows][Ci][2] for pcrl_bn_bwd_finalize(partial, rows, Ci, count = N H W, ...).  _rows == 0: no fused kernel for the shape (two passes). */
int64_t pcrl_conv2d_dgrad_bnred_rows(int N, int H, int W, int Ci, int CoP, int act, int dtype);
int pcrl_conv2d_dgrad_bnred(const void* dy, const void* wp_dgrad, void* dx, const void* bn_y, const float* scale, const float* shift,
                            const float* mean, const float* rstd, float* partial, int N, int H, int W, int Ci, int CoP, int act, int dtype,
                            pcrl_stream_t stream);
/* Stride-2 data gradient without idle taps: the parity classes (a, b) = (ih & 1, iw & 1) of dx are four stride-1 gathers over dy
 * (3x3/pad 1: 1, 2, 2, 4 taps; 1x1/pad 0: class (0,0) only, the caller zero-fills dx).  Hi, Wi even.  pack_s2: rows round32(Ci),
 * K = taps(a) * taps(b) * CoP -> pcrl_conv2d_packed_elems(Ci, taps(a) * taps(b), CoP) elements, taps(0) = 1, taps(1) = 2 (3x3). */
int pcrl_conv2d_pack_s2(const float* w_ref, void* out, int Co, int Ci, int KH, int KW, int CoP, int a, int b, int dtype, pcrl_stream_t stream);
int pcrl_conv2d_dgrad_s2(const void* dy, const void* wp_class, void* dx, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int CoP, int KH, int KW,
                         int a, int b, int dtype, pcrl_stream_t stream);
size_t pcrl_conv2d_wgrad_ws_bytes(int N, int Ho, int Wo, int CiP, int CoP, int KH, int KW);
int pcrl_conv2d_wgrad(const void* x, const void* dy, float* dw_ref, void* ws, size_t ws_bytes, int N, int Hi, int Wi, int CiP, int Ci_out,
                      int Ho, int Wo, int CoP, int KH, int KW, int stride, int pad, int up, int dtype, pcrl_stream_t stream);

/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the ResNet stem; idx: uint8 [N][Ho][Wo][C] window position of the maximum. */
int pcrl_maxpool2d_3s2_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int dtype, pcrl_stream_t stream);
int pcrl_maxpool2d_3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int dtype, pcrl_stream_t stream);
/* backward of the nearest x2 upsample: dx[N][H][W][C] = sum of the 2x2 block of dy[N][2H][2W][C] */
int pcrl_upsample2d_nearest2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, pcrl_stream_t stream);
/* F.interpolate(scale_factor=s, mode="bilinear", align_corners=False) on float32 NHWC maps (pcrlv2_model.py:190); N,H,W: INPUT dims */
int pcrl_upsample2d_bilinear_fwd(const float* x, float* y, int N, int H, int W, int C, int scale, pcrl_stream_t stream);
int pcrl_upsample2d_bilinear_bwd(const float* dy, float* dx, int N, int H, int W, int C, int scale, pcrl_stream_t stream);
/* BasicBlock tail: a = relu(t + r);  backward mask: g = da where a > 0 */
int pcrl_add_relu_fwd(const void* t, const void* r, void* a, int64_t n, int dtype, pcrl_stream_t stream);
int pcrl_relu_mask_bwd(const void* da, const void* a, void* g, int64_t n, int dtype, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * UpTransition's first two linear layers as ONE operator: ConvTranspose3d(Ci -> Cm, k=2, s=2) followed directly by
 * Conv3d(Cm -> Co, 3x3x3, pad 1) -- `self.ops(self.up_conv(x))`, pcrlv2_model_3d.py:64 (layers :52 and :9/33), nothing in between.
 * Same function, same parameters (w_up [Ci][Cm][2][2][2], b_up [Cm], w0 [Co][Cm][3][3][3], b0 [Co], reference layouts, float32) and the
 * same four parameter gradients as the two aten calls, computed on the COARSE grid: a fine voxel 2v+p sees only 2x2x2 coarse voxels
 * through its 3x3x3 window, so the composed operator has 8 taps per output phase (0.30 of the multiply-adds) and the Cm-channel
 * 2x-upsampled tensor (and its gradient) never exists.  See csrc/upconv_fused.hip for the algebra.
 *   compose : wf (dtype [8][Co][8][Ci]), wd (dtype [Ci][64][Co]), optionally w3f (dtype [8*Co][27][Ci]: the same composed weights zero-embedded
 *             in 3x3x3 form, read by the wide-brick kernel on the shapes it tiles: coarse D % 4 == H % 8 == W % 16 == 0, bf16, Co % 64 == 0;
 *             NULL: not produced), optionally wd3 (dtype [Ci][27][8*Co]: the data-gradient weights zero-embedded in 3x3x3 form over the
 *             space-to-depth view of dy0 -- channel = parity * Co + co -- read by the wide-brick kernel where pcrl_upconv_dgrad_uses_brick();
 *             NULL: not produced), bias_tab (float32 [27][Co]: border classes of the fine voxel) from the four parameters; once per
 *             optimizer step.  ws: pcrl_upconv_compose_ws_bytes().
 *   fwd     : x dtype [N][D][H][W][Ci] -> y0 dtype [N][2D][2H][2W][Co];  stats_partial: [pcrl_upconv_stats_rows()][Co][2] (sum, sum^2)
 *             for the BatchNorm that follows (or NULL).
 *   dgrad   : dy0 -> dx (dtype [N][D][H][W][Ci]); wd3 may be NULL unless pcrl_upconv_dgrad_uses_brick() (bf16, coarse D % 4 == H % 8 ==
 *             W % 16 == 0, Ci % 64 == 0, Co = 32 * 2^k).
 *   wgrad   : x, dy0 -> dw_up, db_up, dw0 (float32, reference layouts; db0 is the column sum of dy0 -- identically cancelled by the
 *             BatchNorm that follows in the reference model and not produced here).  ws: pcrl_upconv_wgrad_ws_bytes().
 * Channels are multiples of 32.
 *   wgrad_accum / wgrad_finish: the same in two stages, both linear in their intermediates -- `accum` (once per backward pass) adds the pass's
 *             gradient of the composed weights and border-class sums of dy0 (box_acc, float32 [27][Co]) to caller-owned buffers; `finish` runs the
 *             chain rule to dw_up / db_up / dw0 once after the last pass.  dweff_acc: float32 [Co][Ci][64] (the gradient of the composed weights,
 *             index p*8+q; from the brick weight-gradient kernel where it tiles the coarse grid, else from the gather kernel); flags bit 0: store
 *             instead of add (first pass); bit 1: the caller states that dy0 sums to ZERO over all voxels per channel -- true when it is the output of
 *             the training-mode BatchNorm backward that follows conv1 in the reference -- so the border-class sums read only the border voxels and the
 *             interior class is minus the rest (exact arithmetic's value, free of the rounding of dy0). */
size_t pcrl_upconv_compose_ws_bytes(int Ci, int Cm, int Co, int dtype);
int pcrl_upconv_compose(const float* w_up, const float* b_up, const float* w0, const float* b0, void* wf, void* wd, void* w3f, void* wd3,
                        float* bias_tab, void* ws, size_t ws_bytes, int Ci, int Cm, int Co, int dtype, pcrl_stream_t stream);
int64_t pcrl_upconv_fwd_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype);   /* informational: which kernel a shape gets */
int64_t pcrl_upconv_stats_rows(int N, int D, int H, int W, int Ci, int Co, int dtype);
int pcrl_upconv_fwd(const void* x, const void* wf, const void* w3f, const float* bias_tab, void* y0, float* stats_partial, int N, int D, int H, int W,
                    int Ci, int Co, int dtype, pcrl_stream_t stream);
int64_t pcrl_upconv_dgrad_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype);   /* informational: which kernel a shape gets */
int pcrl_upconv_dgrad(const void* dy0, const void* wd, const void* wd3, void* dx, int N, int D, int H, int W, int Ci, int Co, int dtype,
                      pcrl_stream_t stream);
/* The same with a caller-owned workspace (convolution_backward(input) of :9,33 through :52,64 on SMALL coarse grids -- up_tr256's 8x8x4 and
 * the local views' 2^3 / 4^3): the 64-tap reduction of the gather form is split over the grid's third dimension into float partial sums and
 * summed by a fixed-order finish pass (deterministic), like pcrl_conv3d_k3_fwd_ws.  ws_bytes() == 0: no workspace needed, ws may be NULL. */
int64_t pcrl_upconv_dgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Co, int dtype);
int pcrl_upconv_dgrad_ws(const void* dy0, const void* wd, const void* wd3, void* dx, void* ws, int64_t ws_bytes, int N, int D, int H, int W,
                         int Ci, int Co, int dtype, pcrl_stream_t stream);
int64_t pcrl_upconv_wgrad_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype);   /* informational: which kernel a shape gets */
size_t pcrl_upconv_wgrad_accum_ws_bytes(int N, int D, int H, int W, int Ci, int Co, int dtype);
int pcrl_upconv_wgrad_accum(const void* x, const void* dy0, float* dweff_acc, float* box_acc, int flags, void* ws, size_t ws_bytes, int N, int D,
                            int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream);
size_t pcrl_upconv_wgrad_finish_ws_bytes(int Ci, int Cm, int Co, int dtype);
int pcrl_upconv_wgrad_finish(const float* dweff_acc, const float* box_acc, const float* w_up, const float* b_up, const float* w0, float* dw_up,
                             float* db_up, float* dw0, void* ws, size_t ws_bytes, int Ci, int Cm, int Co, int dtype, pcrl_stream_t stream);
size_t pcrl_upconv_wgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Cm, int Co, int dtype);
int pcrl_upconv_wgrad(const void* x, const void* dy0, const float* w_up, const float* b_up, const float* w0, float* dw_up, float* db_up,
                      float* dw0, void* ws, size_t ws_bytes, int N, int D, int H, int W, int Ci, int Cm, int Co, int dtype,
                      pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * All cosine terms of one step in one call -- train_3d.py:119-134 (13 cos_loss calls = 26 cosine means, :86-92).
 *   out[g] = sum_{t : group[t] == g} w[t] * mean_r cos(x[t][r], y[t][r]),  x[t], y[t]: float32 [rows][C[t]] on the device
 * `x`, `y`, `dx`, `w`, `C`, `group`, `first` are HOST arrays of `nterms` <= 32 entries (device pointers / scalars); they are copied into
 * the kernel arguments.  bwd: dx[t] (device float32 [rows][C[t]]; several terms may name the same buffer) receives
 * the sum, in term order (deterministic), of dout[group[t]] * w[t] * d(mean cos)/dx[t] over the terms that name it; first[t] != 0
 * must mark exactly the first term of every buffer (checked), and terms that share a buffer share C.
 * y is the reference's detached operand: no gradient.  fwd: `ws` = pcrl_cosine_terms_ws_bytes(nterms) bytes of device scratch. */
size_t pcrl_cosine_terms_ws_bytes(int nterms);
int pcrl_cosine_terms_fwd(const void* const* x, const void* const* y, const float* w, const int* C, const int* group, int nterms, int rows,
                          int ngroups, float eps, float* out, void* ws, size_t ws_bytes, pcrl_stream_t stream);
int pcrl_cosine_terms_bwd(const void* const* x, const void* const* y, void* const* dx, const float* w, const int* C, const int* group,
                          const int* first, int nterms, int rows, int ngroups, float eps, const float* dout, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * torch.optim.SGD (momentum, weight decay, dampening 0, no nesterov) over a flat parameter arena --
 * train_3d.py:48-51,151.  `offsets`: int64[ntensors+1] element offsets of each tensor in the arena;
 * `flags`: int32[ntensors], bit0 = tensor has a gradient this step (others are skipped, like
 * params whose .grad is None), bit1 = momentum buffer already initialised (else buf = g).
 * g is multiplied by grad_scale first (1/world_size after an all-reduce sum). */
int pcrl_sgd_step(float* p, const float* g, float* buf, const int64_t* offsets, const int32_t* flags, int ntensors,
                  int64_t total, float lr, float momentum, float weight_decay, float grad_scale, pcrl_stream_t stream);
/* The sum of the gradients several forward passes produced for each parameter, written into the flat gradient arena -- autograd's
 * AccumulateGrad behind `loss.backward()` (train_3d.py:149) for the tensors t0 .. t0 + cnt - 1 of the arena in one launch (per 480 / nsrc
 * tensors): dst[offsets[t] + e] = ((src[t][0][e] + src[t][1][e]) + ...), e < numels[t].  offsets / numels: device arrays (int64[ntensors + 1]
 * / int64[ntensors], slots padded to 4 floats); offsets_host: the same offsets on the host; srcs: HOST array [cnt][nsrc] of device pointers
 * (contiguous float32; 16-byte aligned ones are read with 16-byte loads; NULL = no term; a tensor with no term is left untouched), read
 * before the call returns. */
int pcrl_grad_sum(float* dst, const int64_t* offsets, const int64_t* numels, const int64_t* offsets_host, const void* const* srcs, int t0, int cnt, int nsrc,
                  pcrl_stream_t stream);
/* torch.cat(local_views, dim=0) (train_3d.py:121) as one launch: dst = the n <= 8 contiguous pieces src[k] (nbytes[k] bytes each, multiples
 * of 16, 16-byte aligned) one after the other.  src / nbytes are HOST arrays, read before the call returns. */
int pcrl_concat(const void* const* src, const int64_t* nbytes, int n, void* dst, pcrl_stream_t stream);
/* loss = loss1 + loss2 + loss4 + local_loss with loss4 = beta * l4 (train_3d.py:136-138) from four device scalars in one launch:
 * out[0] = total (the reference's order of additions), out[1] = beta * l4. */
int pcrl_loss_total(const float* l1, const float* l2, const float* l4, const float* l5, float beta, float* out, pcrl_stream_t stream);
/* Its backward (autograd's mul / select_backward / add_ on four scalars, train_3d.py:138,144): out[0..3] = g, beta * g, g, g -- the gradients
 * of loss1, l4 and of the (global, local) pair of cosine groups, which pcrl_cosine_terms_bwd takes as out + 2. */
int pcrl_loss_total_bwd(const float* g, float beta, float* out, pcrl_stream_t stream);
/* The divergence guard of train_3d.py:140-142 (`if loss > 1000 and epoch > 10: continue`) decided ON THE DEVICE, so that epochs 11..240 run
 * without a forward -> backward host synchronisation: pcrl_guard_flag writes out[0] = (loss[0] > threshold) ? 1 : 0 (under data parallelism
 * the caller MAX-all-reduces it: one process, one decision in the reference), pcrl_sgd_step_guarded is pcrl_sgd_step that does NOTHING when
 * skip[0] != 0 -- parameters and momentum buffers stay bit-unchanged, which is what the reference's `continue` leaves behind. */
int pcrl_guard_flag(const float* loss, float threshold, float* out, pcrl_stream_t stream);
int pcrl_sgd_step_guarded(float* p, const float* g, float* buf, const int64_t* offsets, const int32_t* flags, int ntensors,
                          int64_t total, float lr, float momentum, float weight_decay, float grad_scale, const float* skip,
                          pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Input pipeline: the torchio transforms of data.py:73-89 / datasets/lunaDataset.py:28-81 on float32 volumes
 * [B][D][H][W] resident on the device, one random parameter set per volume (drawn by the caller).  PARITY UNPINNED against torchio
 * (absent from the image; the reference holds no vectors); each entry point states the definition it implements.
 *   volume_min : vmin[b] = min over the volume (RandomAffine's default_pad_value='minimum')
 *   affine     : RandomFlip(axes=0) then RandomAffine: y(o) = trilinear sample of flip_d^{flip[b]}(x) at centre + inv[b] (o - centre),
 *                inv[b] row-major 3x3 in (d,h,w) order and isotropic voxel units, samples outside the volume = fill[b]
 *   blur_axis  : one axis (0=d,1=h,2=w) of RandomBlur: Gaussian of std sigma[b] voxels, taps -radius..radius, symmetric borders
 *   noise_gamma: RandomNoise then RandomGamma: v = x + noise_std[b] * n(0,1) (counter-based generator keyed by seed, b, index),
 *                y = sign(v) |v|^gamma[b]
 *   meanstd / znorm : ZNormalization: (x - mean[b]) * rstd[b], rstd = 1 / unbiased standard deviation
 *   swap       : RandomSwap, in place: for it < iters, exchange patch origins[it][b][0] with patch origins[it][b][1] ([d,h,w] corners,
 *                patch pd x ph x pw); the caller makes the two corners of a skipped (overlapping) draw equal */
int pcrl_aug_volume_min(const float* x, float* vmin, int B, int64_t S, pcrl_stream_t stream);
int pcrl_aug_affine(const float* x, float* y, const float* inv, const int* flip, const float* fill, int B, int D, int H, int W, pcrl_stream_t stream);
int pcrl_aug_blur_axis(const float* x, float* y, const float* sigma, int B, int D, int H, int W, int axis, int radius, pcrl_stream_t stream);
int pcrl_aug_noise_gamma(const float* x, float* y, const float* noise_std, const float* gamma, int B, int64_t S, int64_t seed, pcrl_stream_t stream);
int pcrl_aug_meanstd(const float* x, float* mean, float* rstd, int B, int64_t S, pcrl_stream_t stream);
int pcrl_aug_znorm(const float* x, float* y, const float* mean, const float* rstd, int B, int64_t S, pcrl_stream_t stream);
int pcrl_aug_swap(float* x, const int* origins, int B, int D, int H, int W, int pd, int ph, int pw, int iters, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Test hooks (NOT part of the drop-in surface; process-wide atomics, default 0 / tr 1 = the product path).  They select which
 * of the kernels behind one entry point runs, so that tests can check every kernel against the same reference and probes can
 * time them against each other inside one process (tools/conv_probe.py).
 *   conv  impl: 0 auto (LDS-halo brick kernel where eligible, co-located launch), 1 gather kernel, 2 gather kernel without
 *               split-K, 3 brick kernel on its plain 2-D grid, 4 the 4x8x8-brick kernel also where the 4x8x16-brick one is eligible,
 *               5 / 6 auto with the wide-brick kernel on 4 x 8 x 16 bricks only / on 8 x 8 x 16 bricks wherever they tile
 *   wgrad impl: 0 auto (brick kernel where eligible, XCD co-located launch), 1 gather kernel, 2 brick kernel on its plain 2-D grid,
 *               4 / 5 co-located launch with the old walk order / plain grid with the new walk order (experiments),
 *               6 co-located launch with 64 x 64 tiles only (0 also uses 128 x 64, 64 x 128 and 64 x 32 tiles)
 *   wgrad tr  : bf16 fragment fetch of the gather weight-gradient kernel: 1 ds_read_b64_tr_b16, 0 scalar LDS reads
 *   conv2d impl: 0 auto (brick / narrow kernels where eligible), 1 gather kernel */
void pcrl_debug_set_conv_impl(int impl);
void pcrl_debug_set_wgrad_impl(int impl);
void pcrl_debug_set_wgrad_tr(int on);
void pcrl_debug_set_conv2d_impl(int impl);
/* Timing ablation (tools/double_ablation.py): the non-accumulating fixed-order second passes of the weight gradients are launched n times (idempotent);
 * the step-time difference between n = 2 and n = 1 is what those launches cost inside the multi-stream step.  Default 1. */
void pcrl_debug_set_reduce_repeat(int n);

/* ---------------------------------------------------------------------------------------
 * Backward of one half of the projection / predictor heads (csrc/heads_fused.hip; models/pcrlv2_model_3d.py:55-59,67-70 and
 * models/pcrlv2_model.py:108-111,124-127) in ONE launch: the data gradient of a Linear, the BatchNorm1d (+ReLU) backward of the tensor that
 * gradient belongs to, and the Linear's weight / bias gradients -- aten::mm x 2 + sum (addmm backward), native_batch_norm_backward,
 * threshold_backward.  All float32, row-major:
 *   t[n][c] = add[n][c] + sum_k dy[n][k] * W[k][c]      dy [N][K] (NULL: t = add), W [K][C] (the Linear's weight [out][in]), add [N][C] or NULL
 *   dx, dgamma, dbeta = BatchNorm1d backward of t through xbn [N][C] (the normalisation's input), gamma, mean, rstd [C];
 *                       relu != 0: t is masked where ybn [N][C] (the normalisation's ReLU'd output) is <= 0 first
 *   dW[k][c] = sum_n dy[n][k] * xin[n][c], db[k] = sum_n dy[n][k]      xin [N][C]: the Linear's input in forward
 * N <= 512 rows, C % 4 == 0, K % 4 == 0. */
int pcrl_head_bwd_stage(const float* dy, const float* W, const float* add, const float* xin, const float* xbn, const float* ybn,
                        const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta, float* dW,
                        float* db, int N, int K, int C, int relu, pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * 2D path, the ResNet stem (csrc/stem2d.hip): conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False) of the ResNet-18 encoder
 * smp.Unet('resnet18', in_channels=3) builds (models/pcrlv2_model.py:200) -- aten::convolution / convolution_backward (weight branch; the
 * image needs no gradient) reading the float32 NCHW image as the loader delivers it (train_2d.py:139-141).  bf16 MFMA, float32 accumulation.
 * Available where pcrl_stem7_ok() != 0: bf16, even H, W with H/2 a multiple of 8 and W/2 a multiple of 32 (else pcrl_conv2d_* on the image
 * padded to 8 channels).
 *   pack  : float32 [64][3][7][7] -> bf16 [64][7][32] (k = kw * 4 + c; zeros at c = 3 and kw = 7): pcrl_stem7_packed_elems() elements
 *   fwd   : y bf16 NHWC [N][H/2][W/2][64]; stats: pcrl_stem7_stats_rows(N, H, W) rows of [64][2] (sum, sum^2) for bn1, or NULL
 *   wgrad : dw_ref float32 [64][3][7][7] from dy bf16 [N][H/2][W/2][64]; ws: pcrl_stem7_wgrad_ws_bytes */
int64_t pcrl_stem7_ok(int N, int H, int W, int dtype);
int64_t pcrl_stem7_packed_elems(void);
int64_t pcrl_stem7_stats_rows(int N, int H, int W);
size_t pcrl_stem7_wgrad_ws_bytes(int N, int H, int W);
int pcrl_stem7_pack(const float* w_ref, void* out, pcrl_stream_t stream);
int pcrl_stem7_fwd(const float* x, const void* wp, void* y, float* stats, int N, int H, int W, int dtype, pcrl_stream_t stream);
int pcrl_stem7_wgrad(const float* x, const void* dy, float* dw_ref, void* ws, size_t ws_bytes, int N, int H, int W, int dtype,
                     pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * 2D path, fused passes of the ResNet-18 encoder (csrc/encoder2d.hip): each replaces a chain of separate passes over one tensor and
 * reproduces the chain's values bit for bit (intermediates the chain stored are rounded to `dtype` here too).
 *   bn_add_relu_fwd      : out = relu(T(scale*y + shift) + i), i = r (identity) or T(rscale*r + rshift) (the downsample branch's
 *                          BatchNorm2d; rscale / rshift both NULL or both given) -- torchvision BasicBlock.forward's
 *                          `out = self.bn2(out); out += identity; out = self.relu(out)` (aten::native_batch_norm's apply half, add_, relu_)
 *   relu_mask_sum_bwd    : g = T(da + db) where a > 0, else 0 -- the two gradients that reach a BasicBlock's output (next block's conv1
 *                          branch and identity branch) summed and masked (aten::add + aten::threshold_backward)
 *   bn_relu_maxpool2d_3s2_fwd : p, idx = MaxPool2d(3, 2, 1)(T(relu(scale*y + shift))) from one pass over the stem convolution's output
 *                          (the full-resolution activation has no other consumer on the training path); idx as pcrl_maxpool2d_3s2_fwd
 *   maxpool2d_3s2_bwd_sum: pcrl_maxpool2d_3s2_bwd of T(dy + dy2) (the pooled tensor feeds layer1.0's conv1 and its identity branch) */
int pcrl_bn_add_relu_fwd(const void* y, const float* scale, const float* shift, const void* r, const float* rscale, const float* rshift,
                         void* out, int64_t M, int C, int dtype, pcrl_stream_t stream);
int pcrl_relu_mask_sum_bwd(const void* da, const void* db, const void* a, void* g, int64_t n, int dtype, pcrl_stream_t stream);
int pcrl_bn_relu_maxpool2d_3s2_fwd(const void* y, const float* scale, const float* shift, void* p, uint8_t* idx, int N, int H, int W, int C,
                                   int dtype, pcrl_stream_t stream);
int pcrl_maxpool2d_3s2_bwd_sum(const void* dy, const void* dy2, const uint8_t* idx, void* dx, int N, int H, int W, int C, int dtype,
                               pcrl_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * 2D path, the 3-channel ends of the step (csrc/heads2d.hip).
 * pcrl_mse2d_fwd / _bwd_pad -- nn.MSELoss()(masks, gt) at train_2d.py:165,167 with the prediction in NHWC memory (float32 [N][HW][C], what
 *   the segmentation / deep-supervision heads write) and the target as the loader delivers it (float32 NCHW [N][C][HW]): no layout copy of
 *   the image.  The backward writes d loss / d masks as `dtype` [N*HW][CP] with zeros in channels C..CP-1 -- the form the convolution
 *   backward kernels take (CP = 8), or unpadded float32 (CP = C) for the bilinear backward -- and leaves colpart[pcrl_rows1024(N*HW)][CP],
 *   per-block column sums whose total (pcrl_colsum) is the bias gradient of the convolution that produced the masks
 *   (aten::mse_loss_backward + the zero-pad copy + the sum over pixels of aten::convolution_backward's bias branch).
 * pcrl_conv2d_1x1_small_bwd -- backward of deep_supervision_head[3] = nn.Conv2d(C, 3, kernel_size=1) (models/pcrlv2_model.py:106;
 *   aten::convolution_backward): x `dtype` [M][Ci], dy float32 [M][3], w float32 [3][Ci] -> dx `dtype` [M][Ci] and
 *   part[pcrl_rows1024(M)][PW] (per block: 3*Ci dw entries, 3 db entries, zeros up to the row pitch PW >= 3*Ci + 3; pcrl_colsum
 *   finishes) in one pass over x and dy. */
int64_t pcrl_rows1024(int64_t M);
size_t pcrl_mse2d_ws_bytes(int64_t M);
int pcrl_mse2d_fwd(const float* p, const float* gt, float* loss, void* ws, size_t ws_bytes, int N, int64_t HW, int C, pcrl_stream_t stream);
int pcrl_mse2d_bwd_pad(const float* p, const float* gt, const float* dloss, void* dy, float* colpart, int N, int64_t HW, int C, int CP,
                       int dtype, pcrl_stream_t stream);
int pcrl_conv2d_1x1_small_bwd(const void* x, const float* dy, const float* w, void* dx, float* part, int64_t M, int Ci, int Co, int PW,
                              int dtype, pcrl_stream_t stream);
/* the network input (train_2d.py:139-141: `.float().cuda()` images, NCHW float32 [N][C][HW], C <= 8) as the `dtype` NHWC tensor [N*HW][CP]
 * with zero padding channels the ResNet stem's convolution reads (aten::zeros + aten::copy_ of a permuted view) */
int pcrl_nchw_to_nhwc_pad(const float* x, void* out, int N, int C, int64_t HW, int CP, int dtype, pcrl_stream_t stream);
/* out = a + b on float32 vectors: the sum of the two gradients that reach x_pro = bn(avgpool(x)) -- directly (the projection is a cosine
 * operand) and through the predictor head (pcrlv2_model.py:125-127, pcrlv2_model_3d.py:69-70) -- autograd's aten::add on a [N, C] matrix */
int pcrl_add_f32(const float* a, const float* b, float* out, int64_t n, pcrl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PCRL_HIP_H */
