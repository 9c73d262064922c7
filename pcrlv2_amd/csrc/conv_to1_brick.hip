// LDS-halo ("brick") forward of the 3x3x3 convolution with ONE output channel, bf16 -- the throughput path of
// pcrl_conv3d_to1_fwd for the deep-supervision heads (LUConv(C -> 1), models/pcrlv2_model_3d.py:60,71).
//
//   y[m] = b + sum_t sum_c x[m + delta_t][c] * w[c][t]
//
// One output channel leaves nothing for the matrix unit in the direct form (every product is used once).  The two-pass form
// (conv_c1.hip) computes the pointwise product z[t][m] = sum_c x[m][c] w[c][t] on MFMA and then sums 27 shifted float planes:
// x is read once, but 27 planes of float32 (4.5 x the bf16 input at C = 64) go to HBM and come back.  Here a block owns a
// 4x8x8 brick: it stages the brick's halo (6x10x16 rows of 32 channels, same layout as conv_brick.hip) chunk by chunk,
// computes z for EVERY halo row (A = halo rows as they lie in LDS, B = [32 taps][32 channels] weight tile) and keeps it in
// registers; after the last chunk z goes to LDS as float32 [600 halo voxels][27 taps] (over the dead halo buffer) and each
// thread gathers the 27 entries of its output voxel.  HBM traffic: the input once (+ halo overlap, mostly L2 hits) and 4 bytes
// per voxel out.  The statistics row of the BatchNorm that follows is one (sum, sum^2) pair per brick.
#include "common.h"
#include <mutex>

namespace {

constexpr int TD = 4, TH = 8, TW = 8;
constexpr int HD = TD + 2, HH = TH + 2, HWU = TW + 2, HW = 16;
constexpr int HLINES = HD * HH;                       // 60 (d,h) lines of 16 rows
constexpr int HALO_BYTES = HLINES * HW * 64;          // 60 KiB
constexpr int HLT = 240, HPT = HLINES / 6;            // staging: 240 threads x 10 rounds of 6 lines (see conv_brick.hip)
constexpr int ZROWS = HD * HH * HWU;                  // 600 halo voxels
constexpr int Z_BYTES = ZROWS * 27 * 4;               // 64800

struct To1Params {
  const bf16* x;        // [M][C]
  const float* w_ref;   // [C][27]
  const float* bias;    // [1] or null
  float* y;             // [M]
  float* stats;         // [bricks][2] or null
  int N, D, H, W, C;
};

__device__ __forceinline__ int hoff_w(int row, int slot) {   // weight tile rows of 64 B, see conv_brick.hip
  const int key = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
  return row * 64 + ((slot ^ key) << 4);
}
__device__ __forceinline__ int hoff_h(int row, int slot) { return row * 64 + ((slot ^ (((row >> 2) & 1) << 1)) << 4); }

__global__ void __launch_bounds__(256, 2) to1_brick_fwd_kernel(const To1Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  const int C = p.C, nchunk = C / 32;
  char* wl = smem + HALO_BYTES;                         // [nchunk][32 taps][32 k] bf16, 2 KiB per chunk
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;

  // ---- brick origin (XCD-contiguous ranges, as in conv_brick.hip) ----
  const int bw = p.W / TW, bh = p.H / TH, bd = p.D / TD;
  int b = blockIdx.x;
  if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
  const int brick_id = b;
  const int w0 = (b % bw) * TW; b /= bw;
  const int h0 = (b % bh) * TH; b /= bh;
  const int d0 = (b % bd) * TD; b /= bd;
  const int n = b;

  // ---- weights: float [C][27] -> bf16 tiles [chunk][tap][k], taps 27..31 zero ----
  for (int idx = tid; idx < C * 32; idx += 256) {
    const int c = idx >> 5, t = idx & 31;               // channel, tap
    const float v = t < 27 ? p.w_ref[c * 27 + t] : 0.f;
    const int chunk = c >> 5, k = c & 31;
    *reinterpret_cast<bf16*>(wl + chunk * 2048 + hoff_w(t, k >> 3) + (k & 7) * 2) = (bf16)v;
  }

  // ---- halo staging roles (conv_brick.hip) ----
  const int htid = tid < HLT ? tid : tid - HLT;
  const int hq = htid % 40, hl0 = htid / 40;
  const int hslot = hq & 3;
  int grow[HPT];
  uint32_t hvalid = 0;
#pragma unroll
  for (int i = 0; i < HPT; ++i) {
    const int line = hl0 + 6 * i;
    const int hd = line / HH, hh = line % HH, hw = hq >> 2;
    const int d = d0 + hd - 1, h = h0 + hh - 1, w = w0 + hw - 1;
    const bool ok = (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
    grow[i] = ok ? ((n * p.D + d) * p.H + h) * p.W + w : ((n * p.D + d0) * p.H + h0) * p.W + w0;
    hvalid |= (uint32_t)ok << i;
  }
  const int hdst0 = hoff_h(hl0 * HW + (hq >> 2), hslot);
  u32x4 rh[HPT];
#define T1_LOAD_HALO(c_)                                                                                  \
  do {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < HPT; ++i)                                                       \
      rh[i] = *reinterpret_cast<const u32x4*>(p.x + (int64_t)grow[i] * C + (c_)*32 + hslot * 8);          \
  } while (0)
#define T1_STORE_HALO()                                                                                   \
  do {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < HPT; ++i)                                                       \
      *reinterpret_cast<u32x4*>(halo + hdst0 + i * (6 * HW * 64)) = keep_if((hvalid >> i) & 1u, rh[i]);   \
  } while (0)

  // ---- z accumulators: this wave's 15 lines (row fragments) x 2 tap fragments ----
  constexpr int FPW = HLINES / 4;   // 15
  f32x4 acc[FPW][2];
#pragma unroll
  for (int f = 0; f < FPW; ++f)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int aoff = hoff_h(wid * FPW * HW + lr, lg);   // + f * HW * 64 (the swizzle key only looks at row bits 2, i.e. lr)
  const int boff = hoff_w(lr, lg);                    // + j * 1024

  T1_LOAD_HALO(0);
  T1_STORE_HALO();
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    if (c + 1 < nchunk) T1_LOAD_HALO(c + 1);
    const char* wt = wl + c * 2048;
    const bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(wt + boff);
    const bf16x8 fb1 = *reinterpret_cast<const bf16x8*>(wt + boff + 1024);
#pragma unroll
    for (int f = 0; f < FPW; ++f) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(halo + aoff + f * (HW * 64));
      acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb0, acc[f][0], 0, 0, 0);
      acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb1, acc[f][1], 0, 0, 0);
    }
    __syncthreads();                 // every wave is done with this chunk's halo
    if (c + 1 < nchunk) {
      T1_STORE_HALO();
      __syncthreads();
    }
  }
#undef T1_LOAD_HALO
#undef T1_STORE_HALO

  // ---- z -> LDS [600 halo voxels][27 taps] float (the halo and weight tiles are dead) ----
  float* zl = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int f = 0; f < FPW; ++f) {
    const int line = wid * FPW + f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int hw = 4 * lg + r;
      if (hw < HWU) {
        float* row = zl + (line * HWU + hw) * 27;
        row[lr] = acc[f][0][r];
        if (lr < 11) row[16 + lr] = acc[f][1][r];
      }
    }
  }
  __syncthreads();
  // ---- gather: thread = output voxel ----
  const int vd = tid >> 6, vh = (tid >> 3) & 7, vw = tid & 7;
  float out = p.bias ? p.bias[0] : 0.f;
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
        out += zl[((((vd + kd) * HH + (vh + kh)) * HWU) + (vw + kw)) * 27 + (kd * 9 + kh * 3 + kw)];
  p.y[(((int64_t)n * p.D + d0 + vd) * p.H + h0 + vh) * p.W + w0 + vw] = out;
  if (p.stats) {
    __shared__ float red[8];
    const float s1 = wave_sum(out), s2 = wave_sum(out * out);
    if (lane == 0) {
      red[wid * 2 + 0] = s1;
      red[wid * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid == 0) {
      p.stats[(int64_t)brick_id * 2 + 0] = (red[0] + red[2]) + (red[4] + red[6]);
      p.stats[(int64_t)brick_id * 2 + 1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
  }
}

}  // namespace

// ---- internal interface used by conv_c1.hip ------------------------------------------------------------------------
bool pcrl_to1_brick_eligible(int N, int D, int H, int W, int C, int taps, int dtype) {
  return dtype == PCRL_BF16 && taps == 27 && D % TD == 0 && H % TH == 0 && W % TW == 0 && C % 32 == 0 && C <= 512 &&
         (int64_t)N * D * H * W < (int64_t)1 << 31;
}
int64_t pcrl_to1_brick_rows(int N, int D, int H, int W) { return (int64_t)N * (D / TD) * (H / TH) * (W / TW); }

int pcrl_to1_brick_launch(const void* x, const float* w_ref, const float* bias, float* y, float* stats, int N, int D, int H, int W, int C,
                          hipStream_t stream) {
  const int lds_w = HALO_BYTES + (C / 32) * 2048;
  const int lds = lds_w > Z_BYTES ? lds_w : Z_BYTES;
  static std::once_flag attr_once;   // hipFuncSetAttribute once per process, race-free
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(to1_brick_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HALO_BYTES + 16 * 2048);
  });
  To1Params p{(const bf16*)x, w_ref, bias, y, stats, N, D, H, W, C};
  hipLaunchKernelGGL(to1_brick_fwd_kernel, dim3((unsigned)pcrl_to1_brick_rows(N, D, H, W)), dim3(256), lds, stream, p);
  return pcrl_check_launch("to1_brick_fwd");
}
