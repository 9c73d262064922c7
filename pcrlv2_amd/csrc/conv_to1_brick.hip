// LDS-halo ("brick") forward of the 3x3x3 convolution with ONE output channel, bf16 -- the throughput path of
// pcrl_conv3d_to1_fwd for the deep-supervision heads (LUConv(C -> 1), models/pcrlv2_model_3d.py:60,71).
//
//   y[m] = b + sum_t sum_c x[m + delta_t][c] * w[c][t]
//
// One output channel leaves nothing for the matrix unit in the direct form (every product is used once).  The two-pass form
// (conv_c1.hip) computes the pointwise product z[t][m] = sum_c x[m][c] w[c][t] on MFMA and then sums 27 shifted float planes:
// x is read once, but 27 planes of float32 (4.5 x the bf16 input at C = 64) go to HBM and come back.  Here a block owns a
// 4x8x8 brick: it stages the brick's halo (6 x 10 x 10 = 600 voxels, COMPACT: 608 rows of 32 channels, any 16 consecutive rows are
// one A fragment -- z is computed per row, so fragments need not follow the (d, h) lines) chunk by chunk, computes z for every halo
// row (B = [32 taps][32 channels] weight tile) and keeps it in registers; after the last chunk z goes to LDS as float32 in two
// phases (taps 0..15, then 16..26: [600 rows][17] and [600 rows][13] over the dead halo buffer) and each thread gathers the 27
// entries of its output voxel in tap order.  HBM traffic: the input once (+ halo overlap, mostly L2 hits) and 4 bytes per voxel
// out.  The statistics row of the BatchNorm that follows is one (sum, sum^2) pair per brick.
//
// Round 3: the first version kept the halo as 60 lines of 16 rows (6 of them padding: 58 % more MFMAs, LDS reads and staging
// slots), held all 27 taps of z in LDS at once (64.8 KB: two blocks per CU) and converted the float32 weights to bf16 tiles in every
// block (2 048 scattered loads per block of 256 voxels; 15-30 % of the kernel by ablation).  Now: compact rows, z in two phases
// (43 KB: three blocks per CU), and the weight tiles packed once per call into the workspace as the LDS image (one 16-byte load
// per thread and chunk pair).
#include "common.h"
#include <mutex>

namespace {

constexpr int TD = 4, TH = 8, TW = 8;
constexpr int HD = TD + 2, HH = TH + 2, HWU = TW + 2;
constexpr int ZROWS = HD * HH * HWU;                  // 600 halo voxels
constexpr int NFRAG = (ZROWS + 15) / 16;              // 38 A fragments of 16 consecutive rows
constexpr int HROWS = NFRAG * 16;                     // 608 rows of 64 B
constexpr int HALO_BYTES = HROWS * 64;                // 38 912
constexpr int HPT = (HROWS * 4 + 255) / 256;          // 10 staging pieces (16 B) per thread and chunk
constexpr int FPW = (NFRAG + 3) / 4;                  // 10 fragments per wave (wave w: fragments w, w + 4, ...)
constexpr int ZA = 17, ZB = 13;                       // TO1_ZT = 0: row pitches (floats) of the two z phases: odd, conflict-free for the writes and the gather
// TO1_ZT = 1 (round 5): z tap-major, zl[tap][row] with a plane pitch of ZP floats.  The four accumulator elements of a lane are four CONSECUTIVE rows of one
// tap: one ds_write_b128 instead of four ds_write_b32 -- 20 instead of 80 LDS writes per lane and brick.  (profiles/r05y_to1_dma_ab.txt, r05z_to1_ab.txt: 203 -> 191 us at 64 x 64 x 32 x 32, 64 channels.)  ZP = 612: 16-byte aligned planes, and the eight lanes of a write phase (taps lr .. lr + 7) land 144 bytes
// apart mod 256 -- distinct bank quads.  Same sums in the same order: bit-identical.
#ifndef TO1_ZT
#define TO1_ZT 1
#endif
constexpr int ZP = 612;
constexpr int Z_BYTES = TO1_ZT ? 16 * ZP * 4 : ZROWS * ZA * 4;   // 39 168 / 40 800

struct To1Params {
  const bf16* x;        // [M][C]
  const float* w_ref;   // [C][27]
  const bf16* w_img;    // packed LDS image [C / 32][2 KiB] (to1_pack_kernel) or null: converted in every block
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

// w_ref float [C][27] -> the LDS image of the weight tiles: [chunk][32 taps][32 k] bf16 with the tile swizzle, taps 27..31 zero
__global__ void __launch_bounds__(256) to1_pack_kernel(const float* __restrict__ w_ref, bf16* __restrict__ img, int C) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= C * 32) return;
  const int c = idx >> 5, t = idx & 31;
  const float v = t < 27 ? w_ref[c * 27 + t] : 0.f;
  const int chunk = c >> 5, k = c & 31;
  *reinterpret_cast<bf16*>(reinterpret_cast<char*>(img) + chunk * 2048 + hoff_w(t, k >> 3) + (k & 7) * 2) = (bf16)v;
}

__global__ void __launch_bounds__(256, 3) to1_brick_fwd_kernel(const To1Params p) {
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

  // ---- weight tiles ----
  if (p.w_img) {
    for (int i = tid; i < nchunk * 128; i += 256)
      *reinterpret_cast<u32x4*>(wl + i * 16) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.w_img) + i * 16);
  } else {
    for (int idx = tid; idx < C * 32; idx += 256) {
      const int c = idx >> 5, t = idx & 31;               // channel, tap
      const float v = t < 27 ? p.w_ref[c * 27 + t] : 0.f;
      const int chunk = c >> 5, k = c & 31;
      *reinterpret_cast<bf16*>(wl + chunk * 2048 + hoff_w(t, k >> 3) + (k & 7) * 2) = (bf16)v;
    }
  }

  // ---- halo staging: piece q = tid + 256 i -> compact row q >> 2 = (hd * 10 + hh) * 10 + hw, 16-byte slot q & 3 ----
  int grow[HPT];
  uint32_t hvalid = 0;
#pragma unroll
  for (int i = 0; i < HPT; ++i) {
    const int row = (tid + 256 * i) >> 2;
    const int hd = row / (HH * HWU), rem = row % (HH * HWU), hh = rem / HWU, hw = rem % HWU;
    const int d = d0 + hd - 1, h = h0 + hh - 1, w = w0 + hw - 1;
    const bool ok = row < ZROWS && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
    grow[i] = ok ? ((n * p.D + d) * p.H + h) * p.W + w : ((n * p.D + d0) * p.H + h0) * p.W + w0;
    hvalid |= (uint32_t)ok << i;
  }
  const int hslot = tid & 3;
  u32x4 rh[HPT];
#define T1_LOAD_HALO(c_)                                                                                  \
  do {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < HPT; ++i)                                                       \
      rh[i] = *reinterpret_cast<const u32x4*>(p.x + (int64_t)grow[i] * C + (c_)*32 + hslot * 8);          \
  } while (0)
#define T1_STORE_HALO()                                                                                   \
  do {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < HPT; ++i)                                                       \
      if (tid + 256 * i < HROWS * 4)                                                                      \
        *reinterpret_cast<u32x4*>(halo + hoff_h((tid + 256 * i) >> 2, hslot)) = keep_if((hvalid >> i) & 1u, rh[i]); \
  } while (0)

  // ---- z accumulators: this wave's fragments wid, wid + 4, ... (16 consecutive compact rows each) x 2 tap fragments ----
  f32x4 acc[FPW][2];
#pragma unroll
  for (int f = 0; f < FPW; ++f)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int aoff = hoff_h(wid * 16 + lr, lg);          // + f * 4 * 16 * 64 (the swizzle key only looks at row bit 2, i.e. lr)
  const int boff = hoff_w(lr, lg);                     // + j * 1024

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
      if (wid + 4 * f < NFRAG) {                         // wave-uniform: the last round has fragments for waves 0 and 1 only
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(halo + aoff + f * (64 * 64));
        acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb0, acc[f][0], 0, 0, 0);
        acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb1, acc[f][1], 0, 0, 0);
      }
    }
    __syncthreads();                 // every wave is done with this chunk's halo
    if (c + 1 < nchunk) {
      T1_STORE_HALO();
      __syncthreads();
    }
  }
#undef T1_LOAD_HALO
#undef T1_STORE_HALO

  // ---- z -> LDS (the halo and weight tiles are dead) and gather, thread = output voxel, taps in order ----
  float* zl = reinterpret_cast<float*>(smem);
  const int vd = tid >> 6, vh = (tid >> 3) & 7, vw = tid & 7;
  const int vrow = (vd * HH + vh) * HWU + vw;            // halo row of tap (0, 0, 0)
  float out = p.bias ? p.bias[0] : 0.f;
  // phase A: taps 0 .. 15 (C layout of the 16x16 MFMA: acc[f][j][r] = row 16 (wid + 4 f) + 4 lg + r, tap 16 j + lr)
#if TO1_ZT
#pragma unroll
  for (int f = 0; f < FPW; ++f)
    if (wid + 4 * f < NFRAG) *reinterpret_cast<f32x4*>(zl + lr * ZP + (wid + 4 * f) * 16 + 4 * lg) = acc[f][0];   // rows 600 .. 607 of the last fragment: inside the plane, never read
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    out += zl[t * ZP + vrow + (kd * HH + kh) * HWU + kw];
  }
  __syncthreads();
  // phase B: taps 16 .. 26
#pragma unroll
  for (int f = 0; f < FPW; ++f)
    if (wid + 4 * f < NFRAG && lr < 11) *reinterpret_cast<f32x4*>(zl + lr * ZP + (wid + 4 * f) * 16 + 4 * lg) = acc[f][1];
  __syncthreads();
#pragma unroll
  for (int t = 16; t < 27; ++t) {
    const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    out += zl[(t - 16) * ZP + vrow + (kd * HH + kh) * HWU + kw];
  }
#else
#pragma unroll
  for (int f = 0; f < FPW; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (wid + 4 * f) * 16 + 4 * lg + r;
      if (row < ZROWS) zl[row * ZA + lr] = acc[f][0][r];
    }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    out += zl[(vrow + (kd * HH + kh) * HWU + kw) * ZA + t];
  }
  __syncthreads();
  // phase B: taps 16 .. 26
#pragma unroll
  for (int f = 0; f < FPW; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (wid + 4 * f) * 16 + 4 * lg + r;
      if (row < ZROWS && lr < 11) zl[row * ZB + lr] = acc[f][1][r];
    }
  __syncthreads();
#pragma unroll
  for (int t = 16; t < 27; ++t) {
    const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
    out += zl[(vrow + (kd * HH + kh) * HWU + kw) * ZB + (t - 16)];
  }
#endif
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

size_t pcrl_to1_brick_ws_bytes(int C) { return (size_t)(C / 32) * 2048; }

// ws (optional, pcrl_to1_brick_ws_bytes(C)): receives the packed weight tiles; without it every block converts the weights itself
int pcrl_to1_brick_launch(const void* x, const float* w_ref, const float* bias, float* y, float* stats, void* ws, size_t ws_bytes, int N, int D, int H,
                          int W, int C, hipStream_t stream) {
  const int lds_w = HALO_BYTES + (C / 32) * 2048;
  const int lds = lds_w > Z_BYTES ? lds_w : Z_BYTES;
  static std::once_flag attr_once;   // hipFuncSetAttribute once per process, race-free
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(to1_brick_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HALO_BYTES + 16 * 2048);
  });
  const bf16* img = nullptr;
  if (ws && ws_bytes >= pcrl_to1_brick_ws_bytes(C)) {
    img = (const bf16*)ws;
    hipLaunchKernelGGL(to1_pack_kernel, dim3((unsigned)((C * 32 + 255) / 256)), dim3(256), 0, stream, w_ref, (bf16*)ws, C);
    if (int e = pcrl_check_launch("to1_pack")) return e;
  }
  To1Params p{(const bf16*)x, w_ref, img, bias, y, stats, N, D, H, W, C};
  hipLaunchKernelGGL(to1_brick_fwd_kernel, dim3((unsigned)pcrl_to1_brick_rows(N, D, H, W)), dim3(256), lds, stream, p);
  return pcrl_check_launch("to1_brick_fwd");
}
