// Streaming forward kernel of the 2x2x2 / stride-2 transposed convolution, bf16, gfx950 -- the throughput path of
// pcrl_convt3d_k2s2_fwd.  Replaces aten::convolution (transposed) of UpTransition.up_conv (models/pcrlv2_model_3d.py:52,64).
//
//   y[n, 2d+kd, 2h+kh, 2w+kw, co] = b[co] + sum_ci x[n, d, h, w, ci] * w[ci, co, kd, kh, kw]
//
// No reduction over taps: every output voxel has ONE source voxel, so the layer is a GEMM [M voxels] x [8 * Co] with the
// short reduction K = Ci (128..512) and an output 8 * Co / Ci times the size of the input -- HBM-write bound
// (up_tr64: 0.13 GB in, 1.07 GB out).  The generic implicit-GEMM kernel (conv_igemm.hip) spends a block per (128 rows,
// tap) with 4..16 K-steps, re-stages x for each of the 8 taps and stores 2 bytes per lane: 0.82 ms for up_tr64, 1.3 TB/s.
//
// Here a block owns 128 input voxels and walks over the (tap, 64-channel) column tiles:
//   * x is read ONCE, straight into registers, already in MFMA operand layout (wave = 32 voxels, K/32 x 2 fragments);
//   * the weight tile [64 co][128 ci] of the next step is prefetched into registers and double-buffered in LDS;
//   * the product is computed TRANSPOSED (A operand = weights, B operand = x), and the weight rows are permuted on their
//     way into LDS so that D row 4*lg + r of column fragment j is channel 16*lg + 4*j + r: a lane ends up with 16
//     CONSECUTIVE channels of one output voxel = two 16-byte stores (instead of sixteen 2-byte ones).
#include "common.h"

namespace {

constexpr int UM = 128;          // input voxels per block (4 waves x 32)
constexpr int KC = 128;          // channels of one staged weight tile
constexpr int WT_BYTES = 64 * KC * 2;   // 16 KiB per buffer

struct Up2Params {
  const bf16* x;      // [M][K]
  const bf16* w;      // packed [8][Nc][K]
  const float* bias;  // [Nc] or null
  bf16* y;            // [N][2D][2H][2W][Nc]
  Dims g;
  int64_t M;
  int K, Nc;
  int tiles_per_block;   // column tiles (tap, 64 channels) handled by one block (grid.y splits the 8 * Nc / 64 tiles)
  int nt;                // non-temporal stores: the output is a stream far larger than the caches
};

// Weight tile rows are 256 B = 16 slots of 16 B; a fragment read takes 16 consecutive rows at slots {s, s+1}: XOR the slot
// with the row index (mod 16) and the 16 lanes of every ds_read_b128 group land in 16 different bank quads.
__device__ __forceinline__ int wt_off(int row, int slot) { return row * 256 + ((slot ^ (row & 15)) << 4); }

template <int KCN>   // K / 128
__global__ void __launch_bounds__(256) convt_up2_fwd_kernel(const Up2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // two weight tiles
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int K = p.K, Nc = p.Nc;
  const int64_t m0 = (int64_t)blockIdx.x * UM;
  const int ct_per_tap = Nc / 64;
  const int tile0 = blockIdx.y * p.tiles_per_block;
  const int ntile = p.tiles_per_block;

  // ---- x fragments of this wave's 32 voxels, all of K, straight from global memory (B operand: column = voxel) ----
  bf16x8 xa[KCN * 4][2];
  int64_t orow[2];   // output row of tap 0 for the lane's voxel of fragment f
  bool vok[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int64_t m = m0 + wid * 32 + f * 16 + lr;
    vok[f] = m < p.M;
    const int64_t mc = vok[f] ? m : p.M - 1;
    int n, d, h, w;
    decode_voxel(mc, p.g, n, d, h, w);
    orow[f] = up2_row(n, d, h, w, 0, p.g);
    const bf16* src = p.x + mc * K + lg * 8;
#pragma unroll
    for (int kk = 0; kk < KCN * 4; ++kk) xa[kk][f] = *reinterpret_cast<const bf16x8*>(src + kk * 32);
  }

  // ---- weight staging: piece i of the thread = row (tid >> 4) + 16 i of the tile, 16-byte slot tid & 15 ----
  int wdst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = (tid >> 4) + 16 * i;
    const int row = ((co >> 2) & 3) * 16 + (co >> 4) * 4 + (co & 3);   // fragment j = (co>>2)&3, D row = 4*(co>>4) + (co&3)
    wdst[i] = wt_off(row, tid & 15);
  }
  u32x4 rw[4];
#define UP2_LOAD_W(tile_, kc_)                                                                              \
  do {                                                                                                      \
    const int t_ = (tile_) / ct_per_tap, ct_ = (tile_) % ct_per_tap;                                        \
    const bf16* src_ = p.w + ((int64_t)(t_ * Nc + ct_ * 64 + (tid >> 4)) * K) + (kc_)*KC + (tid & 15) * 8;  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) rw[i] = *reinterpret_cast<const u32x4*>(src_ + (int64_t)(16 * i) * K); \
  } while (0)
#define UP2_STORE_W(buf_)                                                                                   \
  do {                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>((buf_) + wdst[i]) = rw[i];      \
  } while (0)

  // fragment read offsets: rows j*16 + lr (j = immediate), slot ks*4 + lg
  int woff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) woff[ks] = wt_off(lr, ks * 4 + lg);

  f32x4 acc[2][4];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  UP2_LOAD_W(tile0, 0);
  UP2_STORE_W(smem);
  __syncthreads();

  int buf = 0;
  for (int ti = 0; ti < ntile; ++ti) {
    const int tile = tile0 + ti;
#pragma unroll
    for (int kc = 0; kc < KCN; ++kc) {
      // prefetch the next weight tile (next K-chunk, or the first chunk of the next column tile; the last one re-loads itself)
      int ntile_i = tile, nkc = kc + 1;
      if (nkc == KCN) { nkc = 0; ntile_i = tile + 1; }
      if (ti + 1 == ntile && kc == KCN - 1) { ntile_i = tile; nkc = kc; }
      UP2_LOAD_W(ntile_i, nkc);
      const char* wt = smem + buf * WT_BYTES;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 fw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const bf16x8*>(wt + woff[ks] + j * 4096);
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], xa[kc * 4 + ks][f], acc[f][j], 0, 0, 0);
      }
      UP2_STORE_W(smem + (buf ^ 1) * WT_BYTES);
      __syncthreads();   // next tile visible; everyone is done with the current one
      buf ^= 1;
    }
    // ---- column tile finished: bias, bf16, two 16-byte stores per voxel fragment ----
    const int t = tile / ct_per_tap, ct = tile % ct_per_tap;
    const int64_t tdelta = ((int64_t)(t >> 2) * (2 * p.g.H) + ((t >> 1) & 1)) * (2 * p.g.W) + (t & 1);
    const int c0 = ct * 64 + lg * 16;
    float bv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bv[q] = p.bias ? p.bias[c0 + q] : 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      union { bf16 h[16]; u32x4 v[2]; } o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) o.h[j * 4 + r] = (bf16)(acc[f][j][r] + bv[j * 4 + r]);
      if (vok[f]) {
        bf16* dst = p.y + (orow[f] + tdelta) * Nc + c0;
        if (p.nt) {   // block-uniform: y is written once and far larger than L2 + MALL (common.h: pcrl_streaming)
          __builtin_nontemporal_store(o.v[0], reinterpret_cast<u32x4*>(dst));
          __builtin_nontemporal_store(o.v[1], reinterpret_cast<u32x4*>(dst + 8));
        } else {
          *reinterpret_cast<u32x4*>(dst) = o.v[0];
          *reinterpret_cast<u32x4*>(dst + 8) = o.v[1];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
#undef UP2_LOAD_W
#undef UP2_STORE_W
}

template <int KCN> int launch(const Up2Params& p, dim3 grid, hipStream_t stream) {
  hipLaunchKernelGGL((convt_up2_fwd_kernel<KCN>), grid, dim3(256), 2 * WT_BYTES, stream, p);
  return pcrl_check_launch("convt_up2_fwd");
}

}  // namespace

// ---- internal interface used by conv_igemm.hip's pcrl_convt3d_k2s2_fwd --------------------------------------------
bool pcrl_convt_up2_eligible(int Ci, int Co, int dtype) {
  return dtype == PCRL_BF16 && (Ci == 128 || Ci == 256 || Ci == 512) && Co % 64 == 0;
}

int pcrl_convt_up2_launch(const void* x, const void* wp, const float* bias, void* y, int N, int D, int H, int W, int Ci, int Co,
                          hipStream_t stream) {
  const int64_t M = (int64_t)N * D * H * W;
  const int ntiles = 8 * (Co / 64);
  const int64_t mblocks = (M + UM - 1) / UM;
  int split = 1;   // small volumes: spread the column tiles over grid.y (x is then read `split` times -- it is small)
  while (mblocks * split < 768 && split * 2 <= ntiles && ntiles % (split * 2) == 0) split *= 2;
  Up2Params p{(const bf16*)x, (const bf16*)wp, bias, (bf16*)y, Dims{N, D, H, W}, M, Ci, Co, ntiles / split, pcrl_streaming(M * 8 * Co * 2) ? 1 : 0};
  const dim3 grid((unsigned)mblocks, (unsigned)split);
  if (Ci == 128) return launch<1>(p, grid, stream);
  if (Ci == 256) return launch<2>(p, grid, stream);
  return launch<4>(p, grid, stream);
}
