// Data gradient of a 3x3x3 convolution on the wide-brick kernel (conv_brick16.h) WITH the first pass of the BatchNorm backward of the layer below in its
// epilogue (BNR instantiations; a translation unit of their own so the plain instantiations' register allocation is untouched).
//
// Where it applies: the activation of the layer below has ONE consumer, this convolution -- ops.0 -> ops.1 inside every DownTransition / UpTransition
// (models/pcrlv2_model_3d.py:37-45: nn.Sequential(LUConv, LUConv)).  The gradient of that activation is then exactly this kernel's output, and the two sums
// the BatchNorm backward needs (sum dz, sum dz * xhat over the batch; aten::native_batch_norm_backward behind ReLU's threshold_backward) are taken from the
// block's 512 x BN output tile while it is still in registers: one two-byte load of the layer's saved pre-normalisation value per output element instead of a
// separate pass that re-reads both tensors (bn_bwd_reduce_kernel, norm_pool.hip: 1.07 GB per pass at 64 x 64 x 32 x 32, 64 channels).  The summands are the
// values AS STORED (bf16), the arithmetic per element is bn_bwd_reduce_kernel's; only the summation order differs (per brick, then bn_bwd_finalize's
// fixed-order fp64 sum over the brick rows).
#include "conv_brick16.h"

bool pcrl_brick16_conv_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype);   // conv_brick16.hip

// partial: [pcrl_brick16_conv_rows(N, D, H, W)][Co][2]; scale, shift, mean, rstd: Co floats each (the layer below's pcrl_bn_finalize outputs)
int pcrl_brick16_dgrad_bnred_launch(const void* dy, const void* wp, void* dx, const void* bn_y, const float* scale, const float* shift, const float* mean,
                                    const float* rstd, float* partial, int N, int D, int H, int W, int Ci, int Co, hipStream_t stream) {
  Brick16Params p{(const bf16*)dy, (const bf16*)wp, nullptr, (bf16*)dx, partial, N, D, H, W, Ci, Co, 0, 0, nullptr, 0, (const bf16*)bn_y, scale, shift, mean, rstd};
  const int BN = Co % 64 == 0 ? 64 : 32, ny = Co / BN;
  const int64_t bricks = (int64_t)N * D * H * W / (TD * TH * TW);
  if (bricks * ny >= ((int64_t)1 << 31)) return pcrl_fail(PCRL_EINVAL, "brick16_dgrad_bnred: grid too large");
  dim3 grid((unsigned)bricks, ny);
  if (ny > 1) {
    p.ny = ny;
    grid = dim3((unsigned)(bricks * ny));
  }
  return BN == 64 ? launch16<64, 0, 4, true>(p, grid, stream, "brick16_dgrad_bnred") : launch16<32, 0, 4, true>(p, grid, stream, "brick16_dgrad_bnred");
}

// The 2D path's 3x3 / stride 1 / pad 1 data gradient (MODE 3: the image index is the brick's d axis) with the same epilogue: conv2 -> bn1 inside a
// BasicBlock (torchvision ResNet-18: relu(bn1(conv1(x))) feeds conv2 only) and conv2 -> conv1's BatchNorm inside a DecoderBlock (models/pcrlv2_model.py:113-128).
// dy: [N][H][W][Ci]; dx, bn_y: [N][H][W][Co]; partial: [N H W / 512][Co][2].
int pcrl_brick16_dgrad2d_bnred_launch(const void* dy, const void* wp, void* dx, const void* bn_y, const float* scale, const float* shift, const float* mean,
                                      const float* rstd, float* partial, int N, int H, int W, int Ci, int Co, hipStream_t stream) {
  Brick16Params p{(const bf16*)dy, (const bf16*)wp, nullptr, (bf16*)dx, partial, 1, N, H, W, Ci, Co, 0, 0, nullptr, 0, (const bf16*)bn_y, scale, shift, mean, rstd};
  const int BN = Co % 64 == 0 ? 64 : 32, ny = Co / BN;
  const int64_t bricks = (int64_t)N * H * W / (TD * TH * TW);
  if (bricks * ny >= ((int64_t)1 << 31)) return pcrl_fail(PCRL_EINVAL, "brick16_dgrad2d_bnred: grid too large");
  dim3 grid((unsigned)bricks, ny);
  if (ny > 1) {
    p.ny = ny;
    grid = dim3((unsigned)(bricks * ny));
  }
  return BN == 64 ? launch16<64, 3, 4, true>(p, grid, stream, "brick16_dgrad2d_bnred") : launch16<32, 3, 4, true>(p, grid, stream, "brick16_dgrad2d_bnred");
}
