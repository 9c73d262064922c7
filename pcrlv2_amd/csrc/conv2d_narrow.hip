// Forward / data gradient of the NARROW stride-1 convolutions of the 2D path (source and output channels <= 32: decoder block 4 of
// models/pcrlv2_model.py at full resolution, its deep-supervision head, the 16 -> 3 heads and their 3 -> 16 data gradients), bf16.
//
//   y[px][co] = b[co] + sum_{tap, c} x[px + delta_tap][c] * w[co][tap][c]        (x optionally read through a nearest x2 upsample)
//
// The gather kernel (conv2d.hip) re-reads every input pixel once per tap from L2 (9x amplification: 5.4 GB of L2 traffic per
// 512^2 x 64-image launch) and runs 131 072 blocks of a few hundred MFMA cycles each: 0.75 ms where HBM needs 0.2 ms.  Here a
// block owns an 8 x 32-pixel patch: the patch plus its one-pixel halo (10 x 34 pixels x CS channels) is staged in LDS once,
// the whole weight matrix lives in REGISTERS as B fragments (<= 2 x 9 fragments), and an A fragment is a plain 16-byte LDS read
// per lane: lane (pixel lr, k-group lg) of K-step s reads channels c..c+7 of tap (s*32 + lg*8) / CS at its pixel's shifted halo
// position -- K-steps straddle taps for CS = 8 / 16 exactly as in the gather kernel (same packed weights, k = tap * CS + c).
// Wave = 64 pixels (two patch rows) x NF * 16 output channels.  Epilogue: + bias, bf16 or float32 store (columns >= Nc masked),
// one (sum, sum^2) statistics row per BLOCK for the BatchNorm2d that follows (see the kernel's comment for the round-4 form).  The stride-1 data gradient is the same kernel
// on the tap-flipped packed weights.
#include "common.h"

#ifndef NARROW_TRANSPOSED
#define NARROW_TRANSPOSED 1
#endif

namespace {

constexpr int PH = 8, PW = 32, HPH = PH + 2, HPW = PW + 2;
constexpr int NPX = PH * PW, NHP = HPH * HPW;   // 256 patch pixels, 340 halo pixels

struct NarrowConvParams {
  const bf16* x;      // [N][Hs][Ws][CS]
  const bf16* w;      // packed [32][Kpad], k = tap * CS + c
  const float* bias;  // [Nc] or null
  void* y;            // [N][H][W][Nc] bf16, or float when out_f32; RED2: [N][H/2][W/2][Nc]
  float* stats;       // [blocks][Nc][2] or null
  int N, H, W, up;
  int Nc, out_f32;
  int ks;             // 1 (pad 0) or 3 (pad 1)
  int Ktot, Kpad;
  int npatch, per;    // patches in all, patches per block (a block walks a contiguous range)
};

// Round 4 form.  What changed against the one-patch-per-block kernel (351 us for 16 -> 16 channels at 512^2 x 64 images, where HBM needs 180):
//   * a block WALKS a contiguous range of patches with two LDS halo buffers: the next patch's halo is in flight (global -> registers) while
//     the current one is multiplied and stored, the weights' B fragments are loaded once per block instead of once per patch;
//   * the MFMA runs TRANSPOSED (D = W * X^T: the weight fragment is the A operand): a lane then holds FOUR CONSECUTIVE CHANNELS of one pixel --
//     one 8-byte (bf16) or 16-byte (float32) store per fragment instead of four 2-byte stores; 16 lanes cover 16 consecutive pixels;
//   * BatchNorm statistics are accumulated in registers over the block's whole range: ONE (sum, sum^2) row per block (<= 2 048 rows for the
//     finalize instead of 65 536);
//   * RED2: the data gradient of a convolution that read its input through the nearest x2 upsample (decoder conv1, pcrlv2_model.py:114): the 2 x 2
//     sum of F.interpolate's backward is taken on the float accumulators (vertical pair = two fragments of the lane, horizontal pair = the
//     neighbouring lane) and the COARSE tensor is stored -- the fine-resolution gradient (1 GB at block 4) is never written or read.
template <int CS, int NF, int NS, bool RED2>   // source channels (8/16/32), output fragments (Nc <= 16 * NF), K-steps (Kpad / 32)
__global__ void __launch_bounds__(256, PCRL_OCC2) conv2d_narrow_kernel(const NarrowConvParams p) {
  constexpr int VPC = CS / 8;
  constexpr int XPIECES = NHP * VPC, XP = (XPIECES + 255) / 256;
  constexpr int CSH = CS == 8 ? 3 : (CS == 16 ? 4 : 5);
  constexpr int XBYTES = NHP * CS * 2;
  __shared__ __attribute__((aligned(16))) char xS[2][XBYTES];
  __shared__ float red[4][NF * 16][2];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int pw = p.W / PW, ph = p.H / PH;
  const int Hs = p.up ? p.H >> 1 : p.H, Ws = p.up ? p.W >> 1 : p.W;
  const int b_beg = blockIdx.x * p.per, b_end = min(b_beg + p.per, p.npatch);

  // ---- weights: fragments straight into registers (row co = nf * 16 + lr, k = s * 32 + lg * 8), once per block ----
  bf16x8 fb[NF][NS];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int s = 0; s < NS; ++s) fb[nf][s] = *reinterpret_cast<const bf16x8*>(p.w + (int64_t)(nf * 16 + lr) * p.Kpad + s * 32 + lg * 8);
  // ---- per-lane byte offsets: K-step part (tap of this lane's k-group) and pixel part ----
  int koff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int k = s * 32 + lg * 8;
    int tap = k >> CSH;
    const int c = k & (CS - 1);
    const int ntap = p.ks * p.ks;
    if (tap >= ntap) tap = ntap - 1;                       // K padding: the weights are zero there, any finite operand will do
    const int kh = p.ks == 3 ? tap / 3 : 1, kw = p.ks == 3 ? tap - (tap / 3) * 3 : 1;   // 1x1: the centre of the halo
    koff[s] = ((kh * HPW + kw) * CS + c) * 2;
  }
  int poff[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
    const int px = wid * 64 + mf * 16 + lr;
    poff[mf] = (((px >> 5) * HPW) + (px & 31)) * CS * 2;
  }
  // channels of this lane's accumulator rows: c0 + r, c0 = nf * 16 + 4 * lg
  float bv[NF][4], s1[NF][4], s2[NF][4], bvu[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) bvu[nf] = (p.bias && nf * 16 + lr < p.Nc) ? p.bias[nf * 16 + lr] : 0.f;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = nf * 16 + 4 * lg + r;
      bv[nf][r] = (p.bias && co < p.Nc) ? p.bias[co] : 0.f;
      s1[nf][r] = s2[nf][r] = 0.f;
    }

  u32x4 rx[XP];
  uint32_t xok = 0;
#define NC_LOAD(pb_)                                                                                        \
  do {                                                                                                      \
    int t_ = (pb_);                                                                                         \
    const int w0_ = (t_ % pw) * PW; t_ /= pw;                                                               \
    const int h0_ = (t_ % ph) * PH; t_ /= ph;                                                               \
    xok = 0;                                                                                                \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                        \
      const int q = tid + 256 * i;                                                                          \
      const int hp = q / VPC, pc = q % VPC;                                                                 \
      const int hr = hp / HPW, hc = hp % HPW;                                                               \
      int h = h0_ + hr - 1, w = w0_ + hc - 1;                                                               \
      const bool ok = q < XPIECES && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;            \
      if (p.up) {                                                                                           \
        h >>= 1;                                                                                            \
        w >>= 1;                                                                                            \
      }                                                                                                     \
      const int64_t row = ok ? ((int64_t)t_ * Hs + h) * Ws + w : (int64_t)0;                                \
      rx[i] = *reinterpret_cast<const u32x4*>(p.x + row * CS + (ok ? pc * 8 : 0));                          \
      xok |= (uint32_t)ok << i;                                                                             \
    }                                                                                                       \
  } while (0)
#define NC_STORE(buf_)                                                                                      \
  do {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                        \
      const int q = tid + 256 * i;                                                                          \
      if (q < XPIECES) *reinterpret_cast<u32x4*>(xS[buf_] + q * 16) = keep_if((xok >> i) & 1u, rx[i]);      \
    }                                                                                                       \
  } while (0)

  if (b_beg < b_end) {
    NC_LOAD(b_beg);
    NC_STORE(0);
  }
  __syncthreads();
  bf16* __restrict__ Y = reinterpret_cast<bf16*>(p.y);
  float* __restrict__ Yf = reinterpret_cast<float*>(p.y);
  const bool vec_ok = (p.Nc & 3) == 0;
  int cur = 0;
  for (int pb = b_beg; pb < b_end; ++pb) {
    const bool more = pb + 1 < b_end;
    if (more) NC_LOAD(pb + 1);             // the next patch's halo lands while this one is multiplied and stored
    __builtin_amdgcn_sched_barrier(0);
    const char* xs = xS[cur];
    f32x4 acc[4][NF];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bf16x8 fa[4];
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) fa[mf] = *reinterpret_cast<const bf16x8*>(xs + poff[mf] + koff[s]);
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          acc[mf][nf] = NARROW_TRANSPOSED ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nf][s], fa[mf], acc[mf][nf], 0, 0, 0)
                                          : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mf], fb[nf][s], acc[mf][nf], 0, 0, 0);
    }
    // ---- epilogue: lane holds pixel wid*64 + mf*16 + lr, channels nf*16 + 4*lg + r ----
    int t_ = pb;
    const int w0 = (t_ % pw) * PW; t_ /= pw;
    const int h0 = (t_ % ph) * PH; t_ /= ph;
    const int n = t_;
#if NARROW_TRANSPOSED
    if (RED2) {
      // rows 2*wid (mf 0, 1) and 2*wid + 1 (mf 2, 3) of the patch are one coarse row; columns lr, lr ^ 1 one coarse column
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = acc[mf][nf][r] + acc[mf + 2][nf][r];
            v[r] = a + __shfl_xor(a, 1, 64);
          }
          const int c0 = nf * 16 + 4 * lg;
          if ((lr & 1) == 0 && c0 < p.Nc) {
            const int64_t row = ((int64_t)n * (p.H >> 1) + (h0 >> 1) + wid) * (p.W >> 1) + ((w0 + mf * 16 + lr) >> 1);
            if (vec_ok) {
              *reinterpret_cast<bf16x4*>(Y + row * p.Nc + c0) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (c0 + r < p.Nc) Y[row * p.Nc + c0 + r] = (bf16)v[r];
            }
          }
        }
    } else {
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        const int px = wid * 64 + mf * 16 + lr;
        const int64_t row = ((int64_t)n * p.H + h0 + (px >> 5)) * p.W + w0 + (px & 31);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const int c0 = nf * 16 + 4 * lg;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[mf][nf][r] + bv[nf][r];
            s1[nf][r] += v[r];
            s2[nf][r] += v[r] * v[r];
          }
          if (c0 < p.Nc && p.y) {      // p.y == nullptr (kernel argument: uniform): statistics only, the output has no reader (pcrl_conv2d_fwd with y = NULL)
            const int64_t o = row * p.Nc + c0;
            if (vec_ok) {
              if (p.out_f32) *reinterpret_cast<f32x4*>(Yf + o) = f32x4{v[0], v[1], v[2], v[3]};
              else *reinterpret_cast<bf16x4*>(Y + o) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (c0 + r < p.Nc) {
                  if (p.out_f32) Yf[o + r] = v[r];
                  else Y[o + r] = (bf16)v[r];
                }
            }
          }
        }
      }
    }
#else
    // untransposed: lane holds pixels wid*64 + mf*16 + 4*lg + r (four consecutive pixels of a row), channel nf*16 + lr
    if (RED2) {
#pragma unroll
      for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const int co = nf * 16 + lr;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float v = (acc[mf][nf][2 * q] + acc[mf + 2][nf][2 * q]) + (acc[mf][nf][2 * q + 1] + acc[mf + 2][nf][2 * q + 1]);
            const int64_t row = ((int64_t)n * (p.H >> 1) + (h0 >> 1) + wid) * (p.W >> 1) + ((w0 + mf * 16 + 4 * lg) >> 1) + q;
            if (co < p.Nc) Y[row * p.Nc + co] = (bf16)v;
          }
        }
    } else {
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int px = wid * 64 + mf * 16 + lg * 4 + r;
          const int64_t row = ((int64_t)n * p.H + h0 + (px >> 5)) * p.W + w0 + (px & 31);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            const int co = nf * 16 + lr;
            if (co < p.Nc) {
              const float v = acc[mf][nf][r] + bvu[nf];
              if (!p.y) {
              } else if (p.out_f32) Yf[row * p.Nc + co] = v;
              else Y[row * p.Nc + co] = (bf16)v;
              s1[nf][0] += v;
              s2[nf][0] += v * v;
            }
          }
        }
    }
#endif
    if (more) NC_STORE(cur ^ 1);
    __syncthreads();      // the other buffer is complete; everybody is done reading this one
    cur ^= 1;
  }
#undef NC_LOAD
#undef NC_STORE
  if (!RED2 && p.stats) {
#if NARROW_TRANSPOSED
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[nf][r], b = s2[nf][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        if (lr == 0) {
          red[wid][nf * 16 + 4 * lg + r][0] = a;
          red[wid][nf * 16 + 4 * lg + r][1] = b;
        }
      }
#else
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      float a = s1[nf][0], b = s2[nf][0];
      a += __shfl_xor(a, 16, 64);
      b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 32, 64);
      if (lg == 0) {
        red[wid][nf * 16 + lr][0] = a;
        red[wid][nf * 16 + lr][1] = b;
      }
    }
#endif
    __syncthreads();
    if (tid < NF * 16 && tid < p.Nc) {
      float* o = p.stats + ((int64_t)blockIdx.x * p.Nc + tid) * 2;
      o[0] = (red[0][tid][0] + red[1][tid][0]) + (red[2][tid][0] + red[3][tid][0]);
      o[1] = (red[0][tid][1] + red[1][tid][1]) + (red[2][tid][1] + red[3][tid][1]);
    }
  }
}

struct NarrowConvPlan {
  int blocks, per;
};
NarrowConvPlan narrow_conv_plan(int64_t npatch) {
  const int64_t nb = npatch < 2048 ? npatch : 2048;      // eight blocks per CU
  const int per = (int)((npatch + nb - 1) / nb);
  return NarrowConvPlan{(int)((npatch + per - 1) / per), per};
}

template <int CS, int NF> int launch_ns(const NarrowConvParams& p, unsigned blocks, bool red2, hipStream_t st) {
  constexpr int NS3 = CS == 8 ? 3 : (CS == 16 ? 5 : 9);   // K-steps of the 3x3 kernel; the 1x1 kernel has one
  const int ns = p.Kpad / 32;
  if (red2) {
    if (ns != NS3) return pcrl_fail(PCRL_EINVAL, "conv2d_narrow: the upsample-backward form is 3x3 only");
    hipLaunchKernelGGL((conv2d_narrow_kernel<CS, NF, NS3, true>), dim3(blocks), dim3(256), 0, st, p);
  } else if (ns == 1) hipLaunchKernelGGL((conv2d_narrow_kernel<CS, NF, 1, false>), dim3(blocks), dim3(256), 0, st, p);
  else if (ns == NS3) hipLaunchKernelGGL((conv2d_narrow_kernel<CS, NF, NS3, false>), dim3(blocks), dim3(256), 0, st, p);
  else return pcrl_fail(PCRL_EINVAL, "conv2d_narrow: unsupported K (%d steps)", ns);
  return pcrl_check_launch("conv2d_narrow");
}

}  // namespace

// ---- internal interface used by conv2d.hip ---------------------------------------------------------------------------------------
// H, W: output (= logical input) dims; Cs: source channels as stored; Nc: output channels.
bool pcrl_conv2d_narrow_eligible(int N, int H, int W, int Cs, int Nc, int ks, int dtype) {
  if (dtype != PCRL_BF16 || !(Cs == 8 || Cs == 16 || Cs == 32) || Nc < 1 || Nc > 32 || !(ks == 1 || ks == 3)) return false;
  if (H % PH || W % PW || (int64_t)N * H * W * 32 >= ((int64_t)1 << 40)) return false;
  return true;
}
int64_t pcrl_conv2d_narrow_rows(int N, int H, int W) { return narrow_conv_plan((int64_t)N * (H / PH) * (W / PW)).blocks; }

// red2: the output is the 2 x 2 block sum (nearest x2 upsample backward) at [N][H/2][W/2][Nc]; no bias, no statistics, bf16 only
int pcrl_conv2d_narrow_launch(const void* x, const void* wp, const float* bias, void* y, float* stats, int N, int H, int W, int Cs, int Nc, int ks,
                              int up, int out_f32, int red2, hipStream_t stream) {
  const int64_t npatch = (int64_t)N * (H / PH) * (W / PW);
  const NarrowConvPlan pl = narrow_conv_plan(npatch);
  if (red2 && (bias || stats || out_f32 || up || ks != 3)) return pcrl_fail(PCRL_EINVAL, "conv2d_narrow: bad arguments for the upsample-backward form");
  NarrowConvParams p{(const bf16*)x, (const bf16*)wp, bias, y, stats, N, H, W, up, Nc, out_f32, ks, ks * ks * Cs, (ks * ks * Cs + 31) / 32 * 32,
                     (int)npatch, pl.per};
  const unsigned blocks = (unsigned)pl.blocks;
  const bool two = Nc > 16;
  if (Cs == 8) return two ? launch_ns<8, 2>(p, blocks, red2, stream) : launch_ns<8, 1>(p, blocks, red2, stream);
  if (Cs == 16) return two ? launch_ns<16, 2>(p, blocks, red2, stream) : launch_ns<16, 1>(p, blocks, red2, stream);
  return two ? launch_ns<32, 2>(p, blocks, red2, stream) : launch_ns<32, 1>(p, blocks, red2, stream);
}
