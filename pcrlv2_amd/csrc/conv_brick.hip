// LDS-halo ("brick") implicit-GEMM 3x3x3 convolution for gfx950, bf16 -- the throughput path of
// pcrl_conv3d_k3_fwd (forward and data-gradient) for volumes whose D,H,W are multiples of the brick.
//
// Replaces aten::convolution / convolution_backward(input) of LUConv.conv1 (models/pcrlv2_model_3d.py:9,33).
//
// Why: the gather kernel (conv_igemm.hip) fetches the A operand of every (tap, 32-channel) K-step from L2: 16 KB per
// 256 MFMA-cycles and CU, above what one CU gets from its XCD's L2 (~56 B/clk).  Here a block owns a spatial brick of
// TD x 8 x 8 output voxels; per 32-channel chunk it stages the brick PLUS its one-voxel halo ((TD+2) x 10 x 10 rows of
// 64 B, zero-filled outside the volume) in LDS once, and all 27 taps read their shifted rows from LDS.  Global traffic
// per chunk: the halo once (2.3x the brick) + 27 weight tiles, ~4x less than the gather kernel; MFMA operands are
// 16-byte LDS reads (A: halo row of the lane's voxel + tap offset; B: weight row), 8 x 2 fragments of
// v_mfma_f32_16x16x32_bf16 per wave and tap, three taps (one kw row) per barrier.
//
// Block = 256 threads = 4 waves as 2 (voxels) x 2 (channels): wave tile 128 voxels x 32 output channels.
// LDS: halo 600 x 64 B = 37.5 KiB (single buffer; the next chunk is prefetched into registers during the last tap row)
//      + weights 2 x (3 taps x 64 x 64 B) = 24 KiB  -> two blocks per CU.
#include "common.h"

namespace {

constexpr int TD = 4, TH = 8, TW = 8;
constexpr int BRICK = TD * TH * TW;                    // 256 voxels
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;   // halo extents
constexpr int HROWS = HD * HH * HW;                    // 600
constexpr int BN = 64;
constexpr int HPIECES = HROWS * 4;                     // 16-byte pieces of one 32-channel halo chunk
constexpr int HPT = (HPIECES + 255) / 256;             // pieces per thread (10)
constexpr int HALO_BYTES = HROWS * 64;
constexpr int WT_BYTES = 3 * BN * 64;                  // one weight stage: 3 taps

struct BrickParams {
  const bf16* x;
  const bf16* w;      // packed [Nc][27][K]
  const float* bias;
  bf16* y;
  float* stats;       // [bricks][Nc][2] or null
  int N, D, H, W;
  int K, Nc;
};

// same slot swizzle as conv_igemm.hip's Tile<bf16> (64-byte rows, keys f = [0,2,3,1] per row quad)
__device__ __forceinline__ int hoff(int row, int slot) {
  const int key = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
  return row * 64 + ((slot ^ key) << 4);
}

__global__ void __launch_bounds__(256, 2) brick_conv_kernel(const BrickParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wbuf = smem + HALO_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int K = p.K, nchunk = K / 32;
  const int n0 = blockIdx.y * BN;

  // ---- brick origin ----
  const int bw = p.W / TW, bh = p.H / TH, bd = p.D / TD;
  int b = blockIdx.x;
  const int w0 = (b % bw) * TW; b /= bw;
  const int h0 = (b % bh) * TH; b /= bh;
  const int d0 = (b % bd) * TD; b /= bd;
  const int n = b;

  // ---- halo pieces of this thread: global row (clamped to a valid one) + validity ----
  int grow[HPT];
  uint32_t hvalid = 0;
#pragma unroll
  for (int i = 0; i < HPT; ++i) {
    const int pc = tid + 256 * i;
    const int row = pc >> 2;
    const int hd = row / (HH * HW), hh = (row / HW) % HH, hw = row % HW;
    const int d = d0 + hd - 1, h = h0 + hh - 1, w = w0 + hw - 1;
    const bool ok = pc < HPIECES && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
    grow[i] = ok ? ((n * p.D + d) * p.H + h) * p.W + w : ((n * p.D + d0) * p.H + h0) * p.W + w0;
    hvalid |= (uint32_t)ok << i;
  }
  const int hslot = tid & 3;

  // ---- A-fragment base rows in the halo (tap offset added per tap) ----
  int abase[8];
#pragma unroll
  for (int fm = 0; fm < 8; ++fm) {
    const int v = wm * 128 + fm * 16 + lr;
    abase[fm] = ((v >> 6) * HH + ((v >> 3) & 7)) * HW + (v & 7);
  }
  // ---- weight staging: 3 pieces per thread (tap kw = 0,1,2 of the current (kd,kh)), row co = tid>>2, slot tid&3 ----
  const bf16* wrow = p.w + ((int64_t)(n0 + (tid >> 2)) * 27) * K + (tid & 3) * 8;
  const int wdst = hoff(tid >> 2, tid & 3);   // within one tap tile [64][32]

  f32x4 acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 rh[HPT], rw[3];

#define LOAD_HALO(c_)                                                                                     \
  do {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < HPT; ++i)                                                       \
      rh[i] = *reinterpret_cast<const u32x4*>(p.x + (int64_t)grow[i] * K + (c_)*32 + hslot * 8);          \
  } while (0)
#define STORE_HALO()                                                                                      \
  do {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < HPT; ++i) {                                                     \
      const int pc = tid + 256 * i;                                                                       \
      if (pc < HPIECES) *reinterpret_cast<u32x4*>(halo + hoff(pc >> 2, pc & 3)) = keep_if((hvalid >> i) & 1u, rh[i]); \
    }                                                                                                     \
  } while (0)
#define LOAD_W(c_, s9_)                                                                                   \
  do {                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                         \
      rw[j] = *reinterpret_cast<const u32x4*>(wrow + (int64_t)((s9_)*3 + j) * K + (c_)*32);               \
  } while (0)
#define STORE_W(buf_)                                                                                     \
  do {                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                         \
      *reinterpret_cast<u32x4*>(wbuf + (buf_)*WT_BYTES + j * (BN * 64) + wdst) = rw[j];                   \
  } while (0)

  LOAD_HALO(0);
  LOAD_W(0, 0);
  STORE_HALO();
  STORE_W(0);
  __syncthreads();

  int cur = 0;
  for (int c = 0; c < nchunk; ++c) {
    for (int s9 = 0; s9 < 9; ++s9) {
      // next weight stage (wraps into the next chunk; the very last one re-loads itself)
      int cn = c, sn = s9 + 1;
      if (sn == 9) { sn = 0; cn = c + 1; }
      const bool last = (cn == nchunk);
      if (last) { cn = c; sn = s9; }
      LOAD_W(cn, sn);
      const bool halo_next = (s9 == 8) && !last;   // block-uniform
      if (halo_next) LOAD_HALO(c + 1);
      __builtin_amdgcn_sched_barrier(0);

      const int kd = s9 / 3, kh = s9 % 3;
      const int tapoff = (kd * HH + kh) * HW;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const char* wt = wbuf + cur * WT_BYTES + kw * (BN * 64);
        bf16x8 fb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(wt + hoff(wn * 32 + j * 16 + lr, lg));
#pragma unroll
        for (int fm = 0; fm < 8; ++fm) {
          const bf16x8 fa = *reinterpret_cast<const bf16x8*>(halo + hoff(abase[fm] + tapoff + kw, lg));
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[fm][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[j], acc[fm][j], 0, 0, 0);
        }
      }

      __builtin_amdgcn_sched_barrier(0);
      STORE_W(cur ^ 1);
      if (halo_next) {
        __syncthreads();   // every wave has finished reading this chunk's halo
        STORE_HALO();
      }
      __syncthreads();
      cur ^= 1;
    }
  }
#undef LOAD_HALO
#undef STORE_HALO
#undef LOAD_W
#undef STORE_W

  // ---- epilogue: bias, store, BatchNorm partial statistics (one row per brick) ----
  float s1[2], s2[2], bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    s1[j] = 0.f;
    s2[j] = 0.f;
    bv[j] = p.bias ? p.bias[n0 + wn * 32 + j * 16 + lr] : 0.f;
  }
#pragma unroll
  for (int fm = 0; fm < 8; ++fm) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int v = wm * 128 + fm * 16 + lg * 4 + r;
      const int64_t row = (((int64_t)n * p.D + d0 + (v >> 6)) * p.H + h0 + ((v >> 3) & 7)) * p.W + w0 + (v & 7);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float val = acc[fm][j][r] + bv[j];
        p.y[row * p.Nc + n0 + wn * 32 + j * 16 + lr] = (bf16)val;
        s1[j] += val;
        s2[j] += val * val;
      }
    }
  }
  if (p.stats) {
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][2][16][2]; the loop ended with a barrier
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float a = s1[j], c2 = s2[j];
      a += __shfl_xor(a, 16, 64);
      c2 += __shfl_xor(c2, 16, 64);
      a += __shfl_xor(a, 32, 64);
      c2 += __shfl_xor(c2, 32, 64);
      if (lg == 0) {
        red[((wid * 2 + j) * 16 + lr) * 2 + 0] = a;
        red[((wid * 2 + j) * 16 + lr) * 2 + 1] = c2;
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int wn_ = tid >> 5, j = (tid >> 4) & 1, l = tid & 15;
      const float* r0 = red + (((0 * 2 + wn_) * 2 + j) * 16 + l) * 2;
      const float* r1 = red + (((1 * 2 + wn_) * 2 + j) * 16 + l) * 2;
      float* o = p.stats + ((int64_t)blockIdx.x * p.Nc + n0 + tid) * 2;
      o[0] = r0[0] + r1[0];
      o[1] = r0[1] + r1[1];
    }
  }
}

}  // namespace

// ---- internal interface used by conv_igemm.hip's dispatcher -------------------------------------------------------
bool pcrl_brick_conv_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return dtype == PCRL_BF16 && D % TD == 0 && H % TH == 0 && W % TW == 0 && Ci % 32 == 0 && Co % BN == 0 &&
         (int64_t)N * D * H * W < (int64_t)1 << 31;
}
int64_t pcrl_brick_conv_rows(int N, int D, int H, int W) { return (int64_t)N * (D / TD) * (H / TH) * (W / TW); }

int pcrl_brick_conv_launch(const void* x, const void* wp, const float* bias, void* y, float* stats,
                           int N, int D, int H, int W, int Ci, int Co, hipStream_t stream) {
  static bool attr_set = false;
  const size_t lds = HALO_BYTES + 2 * WT_BYTES;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(brick_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  BrickParams p{(const bf16*)x, (const bf16*)wp, bias, (bf16*)y, stats, N, D, H, W, Ci, Co};
  dim3 grid((unsigned)pcrl_brick_conv_rows(N, D, H, W), (unsigned)(Co / BN));
  hipLaunchKernelGGL(brick_conv_kernel, grid, dim3(256), lds, stream, p);
  return pcrl_check_launch("brick_conv");
}
