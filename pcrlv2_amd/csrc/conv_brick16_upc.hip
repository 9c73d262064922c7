// Composed ConvTranspose3d -> Conv3d instantiations (forward: MODE 1, data gradient: MODE 2) of the wide-brick convolution kernel (conv_brick16.h).
#include "conv_brick16.h"

// (Measured in round 4, not kept: the composed modes on 8 x 8 x 16 bricks -- no gain over the 4-plane form; the instantiations are gone.)

// ---- forward of the composed ConvTranspose3d -> Conv3d operator on the wide-brick kernel (see Brick16Params::upc) ----
// x: coarse [N][D][H][W][Ci]; w3: zero-embedded weights [8 * Co][27][Ci]; y0: fine [N][2D][2H][2W][Co]; stats [bricks * 8][Co][2]
bool pcrl_brick16_upc_fwd_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return pcrl_brick16_conv_eligible(N, D, H, W, Ci, 8 * Co, dtype) && Co % 64 == 0 && (int64_t)N * D * H * W * 8 < ((int64_t)1 << 29);
}
int pcrl_brick16_upc_fwd_launch(const void* x, const void* w3, const float* bias_tab, void* y0, float* stats, int N, int D, int H, int W, int Ci, int Co,
                                hipStream_t stream) {
  Brick16Params p{(const bf16*)x, (const bf16*)w3, nullptr, (bf16*)y0, stats, N, D, H, W, Ci, 8 * Co, 0, Co, bias_tab, 0};
  const int64_t bricks = pcrl_brick16_conv_rows(N, D, H, W);
  const int ny = 8 * Co / 64;
  if (bricks * ny >= ((int64_t)1 << 31)) return pcrl_fail(PCRL_EINVAL, "brick16 (composed up-conv): grid too large");
  p.ny = ny;
  return launch16<64, 1>(p, dim3((unsigned)(bricks * ny)), stream, "brick16_conv (composed up-conv forward)");
}

// ---- data gradient of the composed operator on the wide-brick kernel (see Brick16Params::cshift) ----
// dy0: fine [N][2D][2H][2W][Co]; wd3: zero-embedded weights [Ci][27][8 * Co]; dx: coarse [N][D][H][W][Ci]
static int upc_cshift(int Co) {
  for (int k = 0; k < 8; ++k)
    if (Co == (32 << k)) return k;
  return -1;
}
bool pcrl_brick16_upc_dgrad_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return upc_cshift(Co) >= 0 && Ci % 64 == 0 && pcrl_brick16_conv_eligible(N, D, H, W, 8 * Co, Ci, dtype) && (int64_t)N * D * H * W * 8 < ((int64_t)1 << 29);
}
int pcrl_brick16_upc_dgrad_launch(const void* dy0, const void* wd3, void* dx, int N, int D, int H, int W, int Ci, int Co, hipStream_t stream) {
  Brick16Params p{(const bf16*)dy0, (const bf16*)wd3, nullptr, (bf16*)dx, nullptr, N, D, H, W, 8 * Co, Ci, 0, Co, nullptr, upc_cshift(Co)};
  const int64_t bricks = pcrl_brick16_conv_rows(N, D, H, W);
  const int ny = Ci / 64;
  if (bricks * ny >= ((int64_t)1 << 31)) return pcrl_fail(PCRL_EINVAL, "brick16 (composed up-conv data gradient): grid too large");
  p.ny = ny;
  return launch16<64, 2>(p, dim3((unsigned)(bricks * ny)), stream, "brick16_conv (composed up-conv data gradient)");
}
