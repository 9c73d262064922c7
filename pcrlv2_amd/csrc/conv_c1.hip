// HBM/L2-bound convolutions with ONE input or ONE output channel (no MFMA: K or N of the GEMM is 1).
//
//   c1_fwd            : first layer, Conv3d(1 -> Co, 3x3x3)   models/pcrlv2_model_3d.py:101 (down_tr64.ops.0)
//   to1_fwd / to1_dgrad : Conv3d(C -> 1), 27 taps (deep_supervision_head.conv1, :60,71) or
//                                     1 tap (OutputTransition.final_conv, :78)
// Scalar fields (x of the first layer, y/dy of the heads) are float32 [M]; C-channel tensors are NDHWC
// in the activation dtype, accessed as 16-byte channel vectors.  The weight gradients of these layers are MFMA GEMMs
// (conv_wgrad.hip, via an im2col of the scalar operand).
#include "common.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ bool tap_ok(int d, int h, int w, int kd, int kh, int kw, const Dims& g) {
  return (unsigned)(d + kd) < (unsigned)g.D && (unsigned)(h + kh) < (unsigned)g.H && (unsigned)(w + kw) < (unsigned)g.W;
}

// ------------------------------------------------------------------------------------------------
// First layer forward: y[m][c] = b[c] + sum_t x[m+delta_t] * w[c][t].  128 voxels per block (one BN-statistics
// partial row, same granularity as the MFMA conv), 16 channels per thread, blockDim = 8*Co.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w_ref, const float* __restrict__ bias,
                              T* __restrict__ y, float* __restrict__ stats, Dims g, int64_t M, int Co) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;             // [27][Co]
  float* red = sm + 27 * Co;  // [128][Co + 1]
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int i = tid; i < 27 * Co; i += nthr) {
    const int t = i / Co, c = i % Co;
    wl[i] = w_ref[c * 27 + t];
  }
  __syncthreads();
  const int vox = tid % 128, cg = tid / 128;  // channel group of 16
  const int64_t m = (int64_t)blockIdx.x * 128 + vox;
  float acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[cg * 16 + j] : 0.f;
  const bool live = m < M;
  if (live) {
    int n, d, h, w;
    decode_voxel(m, g, n, d, h, w);
    const uint32_t mask = tap_mask27(d, h, w, g);
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      if ((mask >> t) & 1u) {
        const float xv = x[m + tap_delta27(t, g)];
        const float* wr = wl + t * Co + cg * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] += xv * wr[j];
      }
    }
    T* yo = y + m * Co + cg * 16;
    constexpr int VEC = 16 / (int)sizeof(T);
#pragma unroll
    for (int v0 = 0; v0 < 16; v0 += VEC) {
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(acc[v0 + j]);
      st16(yo + v0, o);
    }
  }
  if (stats) {
    const int ld = Co + 1;
#pragma unroll
    for (int j = 0; j < 16; ++j) red[vox * ld + cg * 16 + j] = live ? acc[j] : 0.f;
    __syncthreads();
    // all 8 * Co threads reduce: thread (channel c, part q of 8) sums 16 voxel rows, then Co threads combine the 8 parts
    const int c = tid % Co, q = tid / Co;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
    for (int r = q * 16; r < q * 16 + 16; ++r) {
      const float v = red[r * ld + c];
      s1 += v;
      s2 += v * v;
    }
    __syncthreads();   // everyone has read its rows: the tile can be reused for the partials
    red[(q * Co + c) * 2 + 0] = s1;
    red[(q * Co + c) * 2 + 1] = s2;
    __syncthreads();
    if (tid < Co) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a += red[(k * Co + tid) * 2 + 0];
        b += red[(k * Co + tid) * 2 + 1];
      }
      stats[((int64_t)blockIdx.x * Co + tid) * 2 + 0] = a;
      stats[((int64_t)blockIdx.x * Co + tid) * 2 + 1] = b;
    }
  }
}

// First layer forward on 4x8x8 bricks, bf16: one K = 32 MFMA step covers all 27 taps.  The A operand (16 voxels x 32 taps) is
// the im2col of the scalar field, built per lane from eight LDS reads of the brick's halo (6x10x10 floats, zero outside the
// volume); the B operand (taps x 16 channels) is the weight row w_ref[c][8 lg .. 8 lg + 7], contiguous in the reference layout.
// A wave = one d-plane of the brick (4 fragments of 16 voxels).  x and w enter the MFMA rounded to bf16 (like every other
// convolution in bf16 mode); bias, the bf16 store and the BatchNorm partials (one row per brick) come from the float sums.
// six blocks per CU for the narrow forms (<= 80 registers): the kernel is a chain of latencies per block -- stage, im2col, MFMA, store -- (130 -> 116 us
// at 64x64x32 with the batched halo loads; eight blocks per CU spill: 273 us)
// Round 4: a block WALKS bricks (blockIdx.x, + gridDim.x, ...): the next brick's halo is requested before the current one is multiplied and stored
// and lands under its MFMAs and stores (two LDS halo buffers), the weight fragments and tap offsets are built once per block -- the per-brick chain
// stage -> im2col -> MFMA -> store becomes a pipeline.  Same arithmetic per brick, same statistics rows (one per brick): bit-identical.
template <int CO>
__global__ void __launch_bounds__(256, CO <= 32 ? 4 : 2) c1_brick_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w_ref,
                                                           const float* __restrict__ bias, bf16* __restrict__ y,
                                                           float* __restrict__ stats, Dims g, int nbricks) {
  constexpr int FN = CO / 16;
  __shared__ float sh[2][600];
  __shared__ float red[4 * CO * 2];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int bw = g.W / 8, bh = g.H / 8, bd = g.D / 4;
  const uint32_t yl = (uint32_t)((((lg >> 1) * g.W + (lg & 1) * 4) * CO + lr) * 2);   // lane part of every store address of the epilogue
  // halo staging: the (up to) three loads of a thread are issued together (one global-memory latency per brick instead of three)
  float hv[3];
  bool hok[3];
  // Per piece, ONCE per thread: the halo voxel's offset from the brick's first voxel and the faces of the halo it lies on (bit: d-, d+, h-, h+, w-, w+;
  // bit 6 = not a halo voxel).  Per brick a block-uniform mask of the faces that stick out of the volume decides validity with one AND, and the load is
  // a block-uniform base pointer + a 32-bit offset (was: per piece and brick two divisions of the piece number by constants, three range checks and a
  // 64-bit voxel index -- 12 quarter-rate multiplies and 6 64-bit multiply-adds per lane and brick on a vector unit that is the kernel's limiter).
  int hrel[3];
  uint32_t hedge[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int q = tid + 256 * i;
    const int hd = q / 100, hh = (q / 10) % 10, hw = q % 10;
    hrel[i] = ((hd - 1) * g.H + (hh - 1)) * g.W + (hw - 1);
    hedge[i] = (hd == 0 ? 1u : 0u) | (hd == 5 ? 2u : 0u) | (hh == 0 ? 4u : 0u) | (hh == 9 ? 8u : 0u) | (hw == 0 ? 16u : 0u) | (hw == 9 ? 32u : 0u) |
               (q < 600 ? 0u : 64u);
  }
#define C1_LOAD(b_)                                                                                       \
  do {                                                                                                    \
    int t_ = (b_);                                                                                        \
    const int w0_ = (t_ % bw) * 8; t_ /= bw;                                                              \
    const int h0_ = (t_ % bh) * 8; t_ /= bh;                                                              \
    const int d0_ = (t_ % bd) * 4; t_ /= bd;                                                              \
    const float* const xb_ = x + ((((int64_t)t_ * g.D + d0_) * g.H + h0_) * g.W + w0_);                   \
    const uint32_t out_ = (d0_ == 0 ? 1u : 0u) | (d0_ + 4 == g.D ? 2u : 0u) | (h0_ == 0 ? 4u : 0u) | (h0_ + 8 == g.H ? 8u : 0u) |   \
                          (w0_ == 0 ? 16u : 0u) | (w0_ + 8 == g.W ? 32u : 0u) | 64u;                       \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                       \
      hok[i] = (hedge[i] & out_) == 0;                                                                    \
      hv[i] = xb_[hok[i] ? hrel[i] : 0];                                                                  \
    }                                                                                                     \
  } while (0)
#define C1_STORE(buf_)                                                                                    \
  do {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                         \
      if (tid + 256 * i < 600) sh[buf_][tid + 256 * i] = hok[i] ? hv[i] : 0.f;                            \
  } while (0)
  // weights: fragment j, lane (lr = channel, lg = taps 8 lg .. 8 lg + 7)
  bf16x8 fb[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int t = 8 * lg + e;
      fb[j][e] = t < 27 ? (bf16)w_ref[(j * 16 + lr) * 27 + t] : (bf16)0.f;
    }
  int toff[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int t = 8 * lg + e;
    toff[e] = t < 27 ? ((t / 9) * 10 + (t / 3) % 3) * 10 + t % 3 : 0;   // K slots 27 .. 31: any tap of the voxel's own window -- their weight rows are zero
  }
  float bv[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) bv[j] = bias ? bias[j * 16 + lr] : 0.f;
  int b = blockIdx.x, cur = 0;
  if (b < nbricks) {
    C1_LOAD(b);
    C1_STORE(0);
  }
  __syncthreads();
  for (; b < nbricks; b += gridDim.x, cur ^= 1) {
    const bool more = b + (int)gridDim.x < nbricks;
    if (more) C1_LOAD(b + (int)gridDim.x);
    int t = b;
    const int w0 = (t % bw) * 8; t /= bw;
    const int h0 = (t % bh) * 8; t /= bh;
    const int d0 = (t % bd) * 4; t /= bd;
    const int64_t base0 = (((int64_t)t * g.D + d0) * g.H + h0) * g.W + w0;
    const float* shc = sh[cur];
    f32x4 acc[4][FN];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int hb = ((wid * 10 + 2 * f + (lr >> 3)) * 10) + (lr & 7);   // halo index of the lane's voxel (tap 0,0,0 corner)
      bf16x8 fa;
#pragma unroll
      for (int e = 0; e < 8; ++e) fa[e] = (bf16)shc[hb + toff[e]];
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    float s1[FN], s2[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      s1[j] = 0.f;
      s2[j] = 0.f;
    }
    // Store addresses (round 5, as in conv_brick16.h): voxel v = 16 f + 4 lg + r of the wave's plane sits at h = 2 f + (lg >> 1), w = 4 (lg & 1) + r, so a row
    // is a wave-uniform SCALAR base (brick origin + the wave's plane + 2 f lines + r voxels) + one loop-invariant lane offset + the store's immediate
    // (32 j bytes).  The 64-bit voxel index per (f, r) on the vector unit (22 quarter-rate 32-bit multiplies, 11 64-bit multiply-adds and ~60 shifts /
    // adds per lane and brick: ~40 % of the loop's vector issue time, with the SIMDs 65 % busy on vector instructions -- rocprofv3 counters +
    // tools/isa_mix.py, profiles/r05z_c1_addr_ab.txt) is gone.
    char* const ybase = reinterpret_cast<char*>(y) + (base0 + (int64_t)wid * g.H * g.W) * (CO * 2);
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        char* const yrow = ybase + ((int64_t)(2 * f) * g.W + r) * (CO * 2);   // scalar
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const float val = acc[f][j][r] + bv[j];
          *reinterpret_cast<bf16*>(yrow + yl + j * 32) = (bf16)val;
          s1[j] += val;
          s2[j] += val * val;
        }
      }
    if (stats) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        float a = s1[j], c2 = s2[j];
        a += __shfl_xor(a, 16, 64);
        c2 += __shfl_xor(c2, 16, 64);
        a += __shfl_xor(a, 32, 64);
        c2 += __shfl_xor(c2, 32, 64);
        if (lg == 0) {
          red[(wid * CO + j * 16 + lr) * 2 + 0] = a;
          red[(wid * CO + j * 16 + lr) * 2 + 1] = c2;
        }
      }
      __syncthreads();
      if (tid < CO) {
        float a = 0.f, c2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a += red[(q * CO + tid) * 2 + 0];
          c2 += red[(q * CO + tid) * 2 + 1];
        }
        stats[((int64_t)b * CO + tid) * 2 + 0] = a;
        stats[((int64_t)b * CO + tid) * 2 + 1] = c2;
      }
    }
    if (more) C1_STORE(cur ^ 1);
    __syncthreads();   // the other halo buffer is complete; everybody has read this one and this brick's `red` (one barrier with a double-buffered `red`: measured equal, profiles/r05z_c1_addr_ab.txt)
  }
#undef C1_LOAD
#undef C1_STORE
}

// ------------------------------------------------------------------------------------------------
// C -> 1 forward: y[m] = b + sum_t sum_c x[m+delta_t][c] * w[c][t].  LPV lanes share one voxel (one 16-byte
// channel vector each, strided if C is wider), shuffle-reduce, 1024 voxels per block.
// ------------------------------------------------------------------------------------------------
constexpr int TO1_VOX = 1024;
template <typename T>
__global__ void __launch_bounds__(256) to1_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w_ref,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      float* __restrict__ stats, Dims g, int64_t M, int C, int taps, int lpv) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;  // [taps][C]
  __shared__ float red[4];
  const int tid = threadIdx.x;
  for (int i = tid; i < taps * C; i += 256) {
    const int t = i / C, c = i % C;
    wl[i] = w_ref[c * taps + t];
  }
  __syncthreads();
  const int sub = tid % lpv, vslot = tid / lpv, vpp = 256 / lpv;
  const int nvec = C / VEC;
  const float b0 = bias ? bias[0] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  const int64_t mbeg = (int64_t)blockIdx.x * TO1_VOX;
  if (taps == 1 && nvec == lpv) {
    // Pointwise head (out_tr: 1x1x1, 64 -> 1) with one channel vector per lane: no voxel decode (two 64-bit divisions per voxel), the lane's weights in
    // registers, four voxels' loads in flight.  The general loop below ran the SIMDs' vector issue 85 % busy at 4.2 TB/s (profiles/r05z_3d_valu_table.txt:
    // 4 120 vector instructions per wave); same products in the same order.
    float wreg[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) wreg[j] = wl[sub * VEC + j];
    constexpr int U = 4;
    for (int it = 0; it < TO1_VOX / vpp; it += U) {
      Vec16<T> xv[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t m = mbeg + (it + u) * vpp + vslot;
        ok[u] = m < M;
        xv[u] = ld16(x + (ok[u] ? m : mbeg) * C + sub * VEC);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc += to_f(xv[u].v[j]) * wreg[j];
        if (!ok[u]) acc = 0.f;
        for (int o = lpv >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (ok[u] && sub == 0) {
          const float v = acc + b0;
          y[mbeg + (it + u) * vpp + vslot] = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
  } else
  for (int it = 0; it < TO1_VOX / vpp; ++it) {
    const int64_t m = mbeg + it * vpp + vslot;
    float acc = 0.f;
    if (m < M) {
      int n, d, h, w;
      decode_voxel(m, g, n, d, h, w);
      for (int t = 0; t < taps; ++t) {
        int64_t src = m;
        if (taps == 27) {
          const int kd = t / 9 - 1, kh = (t / 3) % 3 - 1, kw = t % 3 - 1;
          if (!tap_ok(d, h, w, kd, kh, kw, g)) continue;
          src = m + ((int64_t)kd * g.H + kh) * g.W + kw;
        }
        for (int cv = sub; cv < nvec; cv += lpv) {
          const Vec16<T> xv = ld16(x + src * C + cv * VEC);
          const float* wr = wl + t * C + cv * VEC;
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc += to_f(xv.v[j]) * wr[j];
        }
      }
    }
    for (int o = lpv >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (m < M && sub == 0) {
      const float v = acc + b0;
      y[m] = v;
      s1 += v;
      s2 += v * v;
    }
  }
  if (stats) {
    const float a = block_sum_256(s1, red);
    const float b = block_sum_256(s2, red);
    if (tid == 0) {
      stats[(int64_t)blockIdx.x * 2 + 0] = a;
      stats[(int64_t)blockIdx.x * 2 + 1] = b;
    }
  }
}

// C -> 1 data gradient: dx[m][c] = add[m][c] + sum_t dy[m - delta_t] * w[c][t].  One 16-byte channel vector per thread.
template <typename T>
__global__ void __launch_bounds__(256) to1_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w_ref,
                                                        const T* add, T* dx, Dims g, int64_t M, int C, int taps) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* wl = sm;  // [taps][C]
  const int tid = threadIdx.x;
  for (int i = tid; i < taps * C; i += 256) {
    const int t = i / C, c = i % C;
    wl[i] = w_ref[c * taps + t];
  }
  __syncthreads();
  const int nvec = C / VEC;
  const int64_t total = M * nvec;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + tid; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / nvec;
    const int cv = (int)(idx % nvec);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if (taps == 27) {
      int n, d, h, w;
      decode_voxel(m, g, n, d, h, w);
      for (int t = 0; t < 27; ++t) {
        const int kd = t / 9 - 1, kh = (t / 3) % 3 - 1, kw = t % 3 - 1;
        if (!tap_ok(d, h, w, -kd, -kh, -kw, g)) continue;
        const float dv = dy[m - (((int64_t)kd * g.H + kh) * g.W + kw)];
        const float* wr = wl + t * C + cv * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += dv * wr[j];
      }
    } else {
      const float dv = dy[m];
      const float* wr = wl + cv * VEC;
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] = dv * wr[j];
    }
    T* o = dx + m * C + cv * VEC;
    Vec16<T> ov;
    if (add) {
      const Vec16<T> old = ld16(add + m * C + cv * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) ov.v[j] = from_f<T>(to_f(old.v[j]) + acc[j]);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) ov.v[j] = from_f<T>(acc[j]);
    }
    st16(o, ov);
  }
}

// wt[t][c] = w_ref[c][t] (t < 27), zero rows 27..31: the B^T operand of the pointwise product.
template <typename T>
__global__ void __launch_bounds__(256) pack_to1_kernel(const float* __restrict__ w_ref, T* __restrict__ wt, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 32 * C) return;
  const int t = i / C, c = i % C;
  wt[i] = from_f<T>(t < 27 ? w_ref[c * 27 + t] : 0.f);
}

// y[m] = bias + sum_t z[t][m + delta_t] over the taps whose neighbour lies inside the volume; z is plane-major, so every
// tap is one coalesced read.  1024 voxels per block = one BatchNorm partial row.
__global__ void __launch_bounds__(256) shift_sum27_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                          float* __restrict__ y, float* __restrict__ stats, Dims g, int64_t M) {
  __shared__ float red[4];
  const float b0 = bias ? bias[0] : 0.f;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int64_t m = (int64_t)blockIdx.x * TO1_VOX + it * 256 + threadIdx.x;
    if (m < M) {
      int n, d, h, w;
      decode_voxel(m, g, n, d, h, w);
      const uint32_t mask = tap_mask27(d, h, w, g);
      float acc = b0;
#pragma unroll
      for (int t = 0; t < 27; ++t)
        if ((mask >> t) & 1u) acc += z[(int64_t)t * M + m + tap_delta27(t, g)];
      y[m] = acc;
      s1 += acc;
      s2 += acc * acc;
    }
  }
  if (stats) {
    const float a = block_sum_256(s1, red);
    const float c = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
      stats[(int64_t)blockIdx.x * 2 + 0] = a;
      stats[(int64_t)blockIdx.x * 2 + 1] = c;
    }
  }
}

int pow2_floor(int v) {
  int p = 1;
  while (p * 2 <= v) p *= 2;
  return p;
}

}  // namespace

int pcrl_debug_conv_impl();   // conv_igemm.hip: 0 = auto
static bool c1_brick_ok(int D, int H, int W, int dtype) {
  return pcrl_debug_conv_impl() == 0 && dtype == PCRL_BF16 && D % 4 == 0 && H % 8 == 0 && W % 8 == 0;
}
extern "C" int64_t pcrl_conv3d_k3_c1_stats_rows(int N, int D, int H, int W, int Co, int dtype) {
  (void)Co;
  if (c1_brick_ok(D, H, W, dtype)) return (int64_t)N * (D / 4) * (H / 8) * (W / 8);
  return ((int64_t)N * D * H * W + 127) / 128;
}

extern "C" int pcrl_conv3d_k3_c1_fwd(const float* x, const float* w_ref, const float* bias, void* y, float* stats_partial,
                                     int N, int D, int H, int W, int Co, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && w_ref && y, "conv3d_k3_c1_fwd: null pointer");
  PCRL_REQUIRE(Co == 16 || Co == 32 || Co == 64, "conv3d_k3_c1_fwd: Co must be 16, 32 or 64 (got %d)", Co);
  const Dims g{N, D, H, W};
  const int64_t M = (int64_t)N * D * H * W;
  if (c1_brick_ok(D, H, W, dtype)) {
    const int64_t nb64 = pcrl_conv3d_k3_c1_stats_rows(N, D, H, W, Co, dtype);
    PCRL_REQUIRE(nb64 < ((int64_t)1 << 31), "conv3d_k3_c1_fwd: too many bricks");
    const int nbr = (int)nb64;
    // persistent blocks: four per CU for the narrow forms (128 registers: the prefetch does not fit the 80 of six blocks per CU), two otherwise
    const int cap = 256 * (Co <= 32 ? 4 : 2);
    const unsigned bricks = (unsigned)(nbr < cap ? nbr : cap);
    if (Co == 16) hipLaunchKernelGGL(c1_brick_fwd_kernel<16>, dim3(bricks), dim3(256), 0, as_stream(stream), x, w_ref, bias, (bf16*)y, stats_partial, g, nbr);
    else if (Co == 32) hipLaunchKernelGGL(c1_brick_fwd_kernel<32>, dim3(bricks), dim3(256), 0, as_stream(stream), x, w_ref, bias, (bf16*)y, stats_partial, g, nbr);
    else hipLaunchKernelGGL(c1_brick_fwd_kernel<64>, dim3(bricks), dim3(256), 0, as_stream(stream), x, w_ref, bias, (bf16*)y, stats_partial, g, nbr);
    return pcrl_check_launch("c1_brick_fwd");
  }
  const unsigned blocks = (unsigned)((M + 127) / 128);
  const size_t lds = (size_t)(27 * Co + 128 * (Co + 1)) * sizeof(float);
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(c1_fwd_kernel<bf16>, dim3(blocks), dim3(8 * Co), lds, as_stream(stream), x, w_ref, bias, (bf16*)y, stats_partial, g, M, Co);
  else if (dtype == PCRL_F32)
    hipLaunchKernelGGL(c1_fwd_kernel<float>, dim3(blocks), dim3(8 * Co), lds, as_stream(stream), x, w_ref, bias, (float*)y, stats_partial, g, M, Co);
  else
    return pcrl_fail(PCRL_EINVAL, "conv3d_k3_c1_fwd: bad dtype %d", dtype);
  return pcrl_check_launch("c1_fwd");
}

static int to1_check(const char* what, int C, int taps, int dtype) {
  if (taps != 27 && taps != 1) return pcrl_fail(PCRL_EINVAL, "%s: taps must be 27 or 1 (got %d)", what, taps);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  if (dtype != PCRL_BF16 && dtype != PCRL_F32) return pcrl_fail(PCRL_EINVAL, "%s: bad dtype %d", what, dtype);
  if (C <= 0 || C % vec != 0 || C > 1024) return pcrl_fail(PCRL_EINVAL, "%s: C=%d must be a multiple of %d, <= 1024", what, C, vec);
  return 0;
}

int pcrl_pointwise_planes_launch(const void* x, const void* wt, float* z, int64_t M, int C, int dtype, hipStream_t stream);
// LDS-halo brick kernel (conv_to1_brick.hip)
bool pcrl_to1_brick_eligible(int N, int D, int H, int W, int C, int taps, int dtype);
int64_t pcrl_to1_brick_rows(int N, int D, int H, int W);
int pcrl_to1_brick_launch(const void* x, const float* w_ref, const float* bias, float* y, float* stats, void* ws, size_t ws_bytes, int N, int D, int H,
                          int W, int C, hipStream_t stream);

extern "C" int64_t pcrl_conv3d_to1_stats_rows(int N, int D, int H, int W, int C, int taps, int dtype) {
  if (pcrl_debug_conv_impl() == 0 && pcrl_to1_brick_eligible(N, D, H, W, C, taps, dtype)) return pcrl_to1_brick_rows(N, D, H, W);
  return ((int64_t)N * D * H * W + TO1_VOX - 1) / TO1_VOX;
}

extern "C" size_t pcrl_conv3d_to1_fwd_ws_bytes(int N, int D, int H, int W, int C, int taps) {
  if (taps != 27 || C % 32 != 0) return 0;
  return (size_t)32 * C * 4 + (size_t)32 * N * D * H * W * sizeof(float);
}

extern "C" int pcrl_conv3d_to1_fwd(const void* x, const float* w_ref, const float* bias, float* y, float* stats_partial,
                                   void* ws, size_t ws_bytes, int N, int D, int H, int W, int C, int taps, int dtype,
                                   pcrl_stream_t stream) {
  if (int e = to1_check("conv3d_to1_fwd", C, taps, dtype)) return e;
  PCRL_REQUIRE(x && w_ref && y, "conv3d_to1_fwd: null pointer");
  if (pcrl_debug_conv_impl() == 0 && pcrl_to1_brick_eligible(N, D, H, W, C, taps, dtype))
    return pcrl_to1_brick_launch(x, w_ref, bias, y, stats_partial, ws, ws_bytes, N, D, H, W, C, as_stream(stream));
  const Dims g{N, D, H, W};
  const int64_t M = (int64_t)N * D * H * W;
  if (taps == 27 && C % 32 == 0 && M % 4 == 0 && ws && ws_bytes >= pcrl_conv3d_to1_fwd_ws_bytes(N, D, H, W, C, taps)) {
    // Two passes instead of a 27-fold gather of x through L2: (1) pointwise MFMA product z[t][m] = sum_c x[m][c] w[c][t]
    // (x is read ONCE), (2) shifted sum of the 27 float32 planes.
    char* wt = (char*)ws;
    float* z = (float*)(wt + (size_t)32 * C * 4);
    if (dtype == PCRL_BF16) hipLaunchKernelGGL(pack_to1_kernel<bf16>, dim3((32 * C + 255) / 256), dim3(256), 0, as_stream(stream), w_ref, (bf16*)wt, C);
    else hipLaunchKernelGGL(pack_to1_kernel<float>, dim3((32 * C + 255) / 256), dim3(256), 0, as_stream(stream), w_ref, (float*)wt, C);
    if (int e = pcrl_check_launch("pack_to1")) return e;
    if (int e = pcrl_pointwise_planes_launch(x, wt, z, M, C, dtype, as_stream(stream))) return e;
    hipLaunchKernelGGL(shift_sum27_kernel, dim3((unsigned)((M + TO1_VOX - 1) / TO1_VOX)), dim3(256), 0, as_stream(stream),
                       (const float*)z, bias, y, stats_partial, g, M);
    return pcrl_check_launch("shift_sum27");
  }
  const unsigned blocks = (unsigned)((M + TO1_VOX - 1) / TO1_VOX);
  const size_t lds = (size_t)taps * C * sizeof(float);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  int lpv = pow2_floor(C / vec);
  if (lpv > 64) lpv = 64;
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(to1_fwd_kernel<bf16>, dim3(blocks), dim3(256), lds, as_stream(stream), (const bf16*)x, w_ref, bias, y, stats_partial, g, M, C, taps, lpv);
  else
    hipLaunchKernelGGL(to1_fwd_kernel<float>, dim3(blocks), dim3(256), lds, as_stream(stream), (const float*)x, w_ref, bias, y, stats_partial, g, M, C, taps, lpv);
  return pcrl_check_launch("to1_fwd");
}

extern "C" int pcrl_conv3d_to1_dgrad(const float* dy, const float* w_ref, const void* add_src, void* dx,
                                     int N, int D, int H, int W, int C, int taps, int dtype, pcrl_stream_t stream) {
  if (int e = to1_check("conv3d_to1_dgrad", C, taps, dtype)) return e;
  PCRL_REQUIRE(dy && w_ref && dx, "conv3d_to1_dgrad: null pointer");
  const Dims g{N, D, H, W};
  const int64_t M = (int64_t)N * D * H * W;
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const int64_t total = M * (C / vec);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  const size_t lds = (size_t)taps * C * sizeof(float);
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(to1_dgrad_kernel<bf16>, dim3((unsigned)blocks), dim3(256), lds, as_stream(stream), dy, w_ref, (const bf16*)add_src, (bf16*)dx, g, M, C, taps);
  else
    hipLaunchKernelGGL(to1_dgrad_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, as_stream(stream), dy, w_ref, (const float*)add_src, (float*)dx, g, M, C, taps);
  return pcrl_check_launch("to1_dgrad");
}

