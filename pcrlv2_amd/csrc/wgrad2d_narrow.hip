// Weight gradient of the NARROW 3x3 / stride 1 / pad 1 convolutions of the 2D path (16 or 32 channels on either side: decoder blocks 3
// and 4 of models/pcrlv2_model.py at 256^2 / 512^2, their deep-supervision convolutions), bf16, gfx950.
//
//   dW[co][ci][kh][kw] = sum_px dy[px][co] * x[px + (kh-1, kw-1)][ci]          (x optionally read through a nearest x2 upsample)
//
// The general kernel (conv_wgrad.hip) works on 64 x 64 result tiles; with 16 channels 15/16 of every tile is padding and the
// kernel is bound by LDS traffic and MFMA issue on zeros (1.2 ms per launch at 512^2 x 64 images where HBM needs 0.2 ms).  Here
// the result is right-sized: a block walks over 8 x 32-pixel patches of the images; per patch it stages dy[256 px][CU] and the
// x halo [10 x 34 px][CV] in LDS ONCE, and each wave multiplies two 32-pixel rows (the K dimension) for all nine taps:
// CU/16 A fragments (dy, transpose reads) x 9 taps x CV/16 B fragments (x at the tap's shifted position, transpose reads).
// Accumulators (CU/16 x 9 x CV/16 fragments) stay in registers over the block's whole patch range; every WAVE writes its own
// partials in the block (fixed order) and the block writes one slab, summed in fixed order by wgrad2d_reduce_kernel (conv_wgrad.hip).
#include "common.h"

namespace {

constexpr int PH = 8, PW = 32, HPH = PH + 2, HPW = PW + 2;   // patch and halo extents
constexpr int NPX = PH * PW, NHP = HPH * HPW;                // 256 patch pixels, 340 halo pixels

struct NarrowParams {
  const bf16* dy;   // [N][H][W][CU]
  const bf16* x;    // [N][Hs][Ws][CV]   (Hs = H, or H/2 when up)
  float* ws;        // [blocks][CU][9 * CV]
  int N, H, W, up;
  int npatch, per;
  int xpitch, cbase;   // x row pitch in channels (CiP) and the first of the CV input channels this launch handles (CiP = 64: two launches)
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
// 32 consecutive LDS rows (the K dimension) x 16 channels starting at channel cb * 16 -> the canonical MFMA operand fragment
__device__ __forceinline__ bf16x8 tr_frag(const char* base, int row0, int cb, int pitch, int lane) {
  const int g = lane >> 4, jr = (lane & 15) >> 2, cq = lane & 3;
  const char* p0 = base + (row0 + 8 * g + jr) * pitch + (cb * 16 + 4 * cq) * 2;
  union { struct { s16x4 a, b; } s; bf16x8 f; } u;
  u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * pitch));
  return u.f;
}

template <int CU, int CV>
__global__ void __launch_bounds__(256, (CU == 32 && CV == 32) ? 1 : PCRL_OCC2) wgrad2d_narrow_kernel(const NarrowParams p) {   // (32 x 32: 61 spilled registers at two waves per SIMD)
  constexpr int UPC = CU / 8, VPC = CV / 8;                   // 16-byte pieces per pixel
  constexpr int DYP = NPX * UPC / 256;                        // dy pieces per thread: 2 / 4
  constexpr int XPIECES = NHP * VPC, XP = (XPIECES + 255) / 256;   // 680 / 1360 -> 3 / 6 per thread
  constexpr int FA = CU >= 16 ? CU / 16 : 1, FB = CV / 16;   // CU = 8 (3-channel heads, padded): one fragment whose rows 8..15 are never stored
  constexpr int DY_BYTES = NPX * CU * 2, X_BYTES = NHP * CV * 2;
  __shared__ __attribute__((aligned(16))) char smem[DY_BYTES + X_BYTES];
  char* dyS = smem;
  char* xS = smem + DY_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int pw = p.W / PW, ph = p.H / PH;
  const int Hs = p.up ? p.H >> 1 : p.H, Ws = p.up ? p.W >> 1 : p.W;
  const int b_beg = blockIdx.x * p.per, b_end = min(b_beg + p.per, p.npatch);

  f32x4 acc[FA][9][FB];
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int b = 0; b < FB; ++b) acc[a][t][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 rdy[DYP], rx[XP];
  uint32_t xok = 0;
#define NW_LOAD(pb_)                                                                                        \
  do {                                                                                                      \
    int t_ = (pb_);                                                                                         \
    const int w0 = (t_ % pw) * PW; t_ /= pw;                                                                \
    const int h0 = (t_ % ph) * PH; t_ /= ph;                                                                \
    const int n = t_;                                                                                       \
    _Pragma("unroll") for (int i = 0; i < DYP; ++i) {                                                       \
      const int q = tid + 256 * i, px = q / UPC, pc = q % UPC;                                              \
      const int64_t row = ((int64_t)n * p.H + h0 + (px >> 5)) * p.W + w0 + (px & 31);                       \
      rdy[i] = *reinterpret_cast<const u32x4*>(p.dy + row * CU + pc * 8);                                   \
    }                                                                                                       \
    xok = 0;                                                                                                \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                        \
      const int q = tid + 256 * i;                                                                          \
      const int hp = q / VPC, pc = q % VPC;                                                                 \
      const int hr = hp / HPW, hc = hp % HPW;                                                               \
      int h = h0 + hr - 1, w = w0 + hc - 1;                                                                 \
      const bool ok = q < XPIECES && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;            \
      if (p.up) {                                                                                           \
        h >>= 1;                                                                                            \
        w >>= 1;                                                                                            \
      }                                                                                                     \
      const int64_t row = ok ? ((int64_t)n * Hs + h) * Ws + w : (int64_t)0;                                 \
      rx[i] = *reinterpret_cast<const u32x4*>(p.x + row * p.xpitch + p.cbase + (ok ? pc * 8 : 0));         \
      xok |= (uint32_t)ok << i;                                                                             \
    }                                                                                                       \
  } while (0)
#define NW_STORE()                                                                                          \
  do {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < DYP; ++i)                                                         \
      *reinterpret_cast<u32x4*>(dyS + (tid + 256 * i) * 16) = rdy[i];                                       \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                        \
      const int q = tid + 256 * i;                                                                          \
      if (q < XPIECES) *reinterpret_cast<u32x4*>(xS + q * 16) = keep_if((xok >> i) & 1u, rx[i]);            \
    }                                                                                                       \
  } while (0)

  if (b_beg < b_end) NW_LOAD(b_beg);
  for (int pb = b_beg; pb < b_end; ++pb) {
    NW_STORE();
    __syncthreads();
    NW_LOAD(pb + 1 < b_end ? pb + 1 : pb);   // the next patch lands while this one is multiplied
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const int r = 2 * wid + kc;             // patch row = one 32-pixel K chunk
      bf16x8 fa[FA];
#pragma unroll
      for (int a = 0; a < FA; ++a) fa[a] = tr_frag(dyS, r * PW, a, CU * 2, lane);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int R0 = (r + t / 3) * HPW + t % 3;
#pragma unroll
        for (int b = 0; b < FB; ++b) {
          const bf16x8 fb = tr_frag(xS, R0, b, CV * 2, lane);
#pragma unroll
          for (int a = 0; a < FA; ++a) acc[a][t][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb, acc[a][t][b], 0, 0, 0);
        }
      }
    }
    __syncthreads();   // every wave is done with this patch's tiles
  }
#undef NW_LOAD
#undef NW_STORE

  // The four waves' accumulators are summed inside the block -- ((w0 + w1) + w2) + w3, through LDS one wave at a time (the staging tiles are
  // free now) -- and the block leaves ONE partial slab: the second pass reads a quarter of the bytes (it was 54-138 us per full-resolution layer,
  // latency-bound on 4 096 slabs).
  constexpr int NACC = FA * 9 * FB * 4;
  static_assert(NACC * 64 * 4 <= DY_BYTES + X_BYTES, "the wave-combine buffer must fit the staging tiles");
  float* comb = reinterpret_cast<float*>(smem);
#pragma unroll 1
  for (int src = 1; src < 4; ++src) {
    if (wid == src) {
      int e = 0;
#pragma unroll
      for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int b = 0; b < FB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r, ++e) comb[e * 64 + lane] = acc[a][t][b][r];
    }
    __syncthreads();
    if (wid == 0) {
      int e = 0;
#pragma unroll
      for (int a = 0; a < FA; ++a)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int b = 0; b < FB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r, ++e) acc[a][t][b][r] += comb[e * 64 + lane];
    }
    __syncthreads();
  }
  if (wid != 0) return;
  // D[i][j]: lane holds i = 16 a + 4 (lane >> 4) + r (co), j = 16 b + (lane & 15) (ci) of tap t
  float* out = p.ws + (int64_t)blockIdx.x * (CU * 9 * p.xpitch) + p.cbase;
#pragma unroll
  for (int a = 0; a < FA; ++a)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int b = 0; b < FB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = a * 16 + (lane >> 4) * 4 + r;
          if (i < CU) out[i * (9 * p.xpitch) + t * p.xpitch + b * 16 + (lane & 15)] = acc[a][t][b][r];
        }
}

struct NarrowPlan {
  int blocks, per;
};
NarrowPlan narrow_plan(int npatch) {
  const int nb = npatch < 1024 ? npatch : 1024;
  const int per = (npatch + nb - 1) / nb;
  return NarrowPlan{(npatch + per - 1) / per, per};
}

}  // namespace

// ---- internal interface used by conv_wgrad.hip ---------------------------------------------------------------------------------
// H, W: OUTPUT (= logical input) dims.
bool pcrl_wgrad2d_narrow_eligible(int N, int H, int W, int CiP, int CoP, int dtype) {
  return dtype == PCRL_BF16 && (CiP == 16 || CiP == 32 || CiP == 64) && (CoP == 8 || CoP == 16 || CoP == 32) && H % PH == 0 && W % PW == 0 &&
         (int64_t)N * H * W * 32 < ((int64_t)1 << 40);
}
int pcrl_wgrad2d_narrow_slabs(int N, int H, int W) { return narrow_plan((int)((int64_t)N * (H / PH) * (W / PW))).blocks; }

int pcrl_wgrad2d_narrow_launch(const void* x, const void* dy, float* ws, int N, int H, int W, int CiP, int CoP, int up, hipStream_t stream) {
  const int npatch = (int)((int64_t)N * (H / PH) * (W / PW));
  const NarrowPlan pl = narrow_plan(npatch);
  const int blocks = pl.blocks;
  for (int cbase = 0; cbase < CiP; cbase += 32) {   // 64 input channels: two launches of the 32-channel kernel on disjoint columns of the same slabs
    const int cv = CiP - cbase < 32 ? CiP - cbase : 32;
    NarrowParams p{(const bf16*)dy, (const bf16*)x, ws, N, H, W, up, npatch, pl.per, CiP, cbase};
    if (CoP == 8 && cv == 16) hipLaunchKernelGGL((wgrad2d_narrow_kernel<8, 16>), dim3(blocks), dim3(256), 0, stream, p);
    else if (CoP == 8) hipLaunchKernelGGL((wgrad2d_narrow_kernel<8, 32>), dim3(blocks), dim3(256), 0, stream, p);
    else if (CoP == 16 && cv == 16) hipLaunchKernelGGL((wgrad2d_narrow_kernel<16, 16>), dim3(blocks), dim3(256), 0, stream, p);
    else if (CoP == 16) hipLaunchKernelGGL((wgrad2d_narrow_kernel<16, 32>), dim3(blocks), dim3(256), 0, stream, p);
    else if (cv == 16) hipLaunchKernelGGL((wgrad2d_narrow_kernel<32, 16>), dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((wgrad2d_narrow_kernel<32, 32>), dim3(blocks), dim3(256), 0, stream, p);
  }
  return pcrl_check_launch("wgrad2d_narrow");
}
