// HBM-bound NDHWC kernels: training-mode BatchNorm3d (+ReLU / sigmoid) forward & backward, MaxPool3d(2),
// global average pool, per-channel column sums.  All accesses are 16-byte channel vectors; every
// reduction is two-stage with a fixed order (no atomics), second stage in fp64.
//
// Replaces aten::native_batch_norm(+_backward), relu_/threshold_backward, sigmoid(+_backward)
// (models/pcrlv2_model_3d.py:12,21,27,33), max_pool3d_with_indices(+backward) (:100,115-117) and
// adaptive_avg_pool3d(+backward) (:67).
#include "common.h"

namespace {

constexpr int TILE_ROWS = 1024;  // rows per first-stage partial

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + __expf(-z)); }

template <int ACT> __device__ __forceinline__ float act_fwd(float z) {
  if (ACT == PCRL_ACT_RELU) return z > 0.f ? z : 0.f;
  if (ACT == PCRL_ACT_SIGMOID) return 1.f / (1.f + expf(-z));
  if (ACT == PCRL_ACT_SILU) return z / (1.f + expf(-z));
  if (ACT == PCRL_ACT_ELU) return z > 0.f ? z : expm1f(z);   // nn.ELU(): alpha = 1 (models/pcrlv2_model_3d.py:25)
  return z;
}
template <int ACT> __device__ __forceinline__ float act_bwd(float z, float da) {
  if (ACT == PCRL_ACT_RELU) return z > 0.f ? da : 0.f;
  if (ACT == PCRL_ACT_SIGMOID) {
    const float a = 1.f / (1.f + expf(-z));
    return da * a * (1.f - a);
  }
  if (ACT == PCRL_ACT_SILU) {   // d/dz [z sigma(z)] = sigma (1 + z (1 - sigma))
    const float a = 1.f / (1.f + expf(-z));
    return da * a * (1.f + z * (1.f - a));
  }
  if (ACT == PCRL_ACT_ELU) return z > 0.f ? da : da * expf(z);
  return da;
}

// A thread's share (rows threadIdx.x, +256, ...) of the (sum, sum^2) pairs of channel c: 8-byte loads, four rows in flight with their
// own accumulators (a fixed tree: deterministic) -- with up to 16 384 statistics rows the walk is a chain of dependent round trips.
__device__ __forceinline__ void partial_pair_sum(const float* __restrict__ partial, int rows, int C, int c, double& s1, double& s2) {
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  const float2* src = reinterpret_cast<const float2*>(partial) + c;
  int r = threadIdx.x;
  for (; r + 768 < rows; r += 1024) {
    float2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[(int64_t)(r + 256 * u) * C];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] += (double)v[u].x; b[u] += (double)v[u].y; }
  }
  for (; r < rows; r += 256) {
    const float2 v = src[(int64_t)r * C];
    a[0] += (double)v.x;
    b[0] += (double)v.y;
  }
  s1 = (a[0] + a[1]) + (a[2] + a[3]);
  s2 = (b[0] + b[1]) + (b[2] + b[3]);
}

// ---------------------------------------------------------------------------------------------
// Statistics finalize: one block per channel.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* __restrict__ partial, int rows, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, float momentum, float eps,
                                                          float* mean, float* rstd, float* scale, float* shift) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double s1, s2;
  partial_pair_sum(partial, rows, C, c, s1, s2);
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) {
    const double mu = s1 / count;
    double var = s2 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const double rs = 1.0 / sqrt(var + (double)eps);
    mean[c] = (float)mu;
    rstd[c] = (float)rs;
    const double sc = (double)gamma[c] * rs;
    scale[c] = (float)sc;
    shift[c] = (float)((double)beta[c] - mu * sc);
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    if (running_var) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
  }
}

__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const float* __restrict__ partial, int rows, int C, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* dgamma, float* dbeta,
                                                              float* k1, float* kB, float* kA) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double s1, s2;
  partial_pair_sum(partial, rows, C, c, s1, s2);
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) {
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    const double g1 = (double)gamma[c] * (double)rstd[c];
    const double b = -g1 * (double)rstd[c] * s2 / count;
    k1[c] = (float)g1;
    kB[c] = (float)b;
    kA[c] = (float)(-g1 * s1 / count - b * (double)mean[c]);
  }
}

// ---------------------------------------------------------------------------------------------
// Elementwise apply (forward and backward).  C % VEC == 0, or C == 1.
// ---------------------------------------------------------------------------------------------
template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ y, T* __restrict__ a, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int64_t nvec, int C) {
  constexpr int VEC = 16 / (int)sizeof(T);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int c0 = (C == 1) ? 0 : (int)((i * VEC) % C);
    const Vec16<T> v = ld16(y + i * VEC);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = (C == 1) ? 0 : c0 + j;
      o.v[j] = from_f<T>(act_fwd<ACT>(scale[c] * to_f(v.v[j]) + shift[c]));
    }
    st16(a + i * VEC, o);
  }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ da, const T* __restrict__ y, T* __restrict__ dy,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ k1, const float* __restrict__ kB,
                                                           const float* __restrict__ kA, int64_t nvec, int C) {
  constexpr int VEC = 16 / (int)sizeof(T);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const int c0 = (C == 1) ? 0 : (int)((i * VEC) % C);
    const Vec16<T> g = ld16(da + i * VEC);
    const Vec16<T> v = ld16(y + i * VEC);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = (C == 1) ? 0 : c0 + j;
      const float yv = to_f(v.v[j]);
      const float dz = act_bwd<ACT>(scale[c] * yv + shift[c], to_f(g.v[j]));
      o.v[j] = from_f<T>(k1[c] * dz + kB[c] * yv + kA[c]);
    }
    st16(dy + i * VEC, o);
  }
}

// Fast variants for C % VEC == 0 with (C/VEC) dividing 256: a thread owns ONE channel vector for its whole life, so the
// per-channel coefficients are loaded once into registers and the streaming loop touches only the activations (the generic
// kernels above re-load 2-5 coefficients per ELEMENT, which bounds them by VMEM issue at ~1.2 TB/s instead of HBM).
template <typename T, int ACT, bool NT>
__device__ __forceinline__ void bn_apply_rc_kernel_body(const T* __restrict__ y, T* __restrict__ a, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int64_t M, int C) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC, cv = threadIdx.x % nvec, slot = threadIdx.x / nvec, nslots = 256 / nvec;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sc[j] = scale[cv * VEC + j]; sh[j] = shift[cv * VEC + j]; }
  const int64_t stride = (int64_t)gridDim.x * nslots;
#pragma unroll 4
  for (int64_t r = (int64_t)blockIdx.x * nslots + slot; r < M; r += stride) {
    const int64_t off = (r * nvec + cv) * VEC;
    const Vec16<T> v = ld16_sel<NT>(y + off);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(act_fwd<ACT>(sc[j] * to_f(v.v[j]) + sh[j]));
    st16_sel<NT>(a + off, o);
  }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_apply_rc_kernel(const T* __restrict__ y, T* __restrict__ a, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int64_t M, int C, bool nt) {
  if (nt) bn_apply_rc_kernel_body<T, ACT, true>(y, a, scale, shift, M, C);
  else bn_apply_rc_kernel_body<T, ACT, false>(y, a, scale, shift, M, C);
}

// The gradient that enters the BatchNorm backward may carry a per-(sample, channel) term on top of (or instead of) the tensor `da`:
// the global-average-pool branch of UpTransition (pcrlv2_model_3d.py:67: d a1 += dg[n][c] / S).  `RowAdd` folds it into both passes, so
// the broadcast is never materialised (pcrl_gap_bwd: one read + one write of the full-resolution gradient, and a write + two reads
// of a tensor that is ONLY the broadcast on the passes whose reconstruction output is unused).  da == nullptr: the term alone.
struct RowAdd {
  const float* g;   // [N][C] or null
  int64_t S;        // rows per sample
  float inv_s;
  const void* da2;  // a SECOND gradient tensor of da's shape and dtype added to da (two consumers of one activation: a 2D decoder block's
                    // output feeds the next block and its own deep-supervision head, pcrlv2_model.py:119-127), or null
};

template <typename T, int ACT, bool NT>
__device__ __forceinline__ void bn_bwd_apply_rc_kernel_body(const T* __restrict__ da, const T* __restrict__ y, T* __restrict__ dy,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ k1, const float* __restrict__ kB,
                                                              const float* __restrict__ kA, int64_t M, int C, RowAdd ra) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC, cv = threadIdx.x % nvec, slot = threadIdx.x / nvec, nslots = 256 / nvec;
  float sc[VEC], sh[VEC], c1[VEC], cB[VEC], cA[VEC], add[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = cv * VEC + j;
    sc[j] = scale[c]; sh[j] = shift[c]; c1[j] = k1[c]; cB[j] = kB[c]; cA[j] = kA[c];
    add[j] = 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * nslots;
  const int64_t r0 = (int64_t)blockIdx.x * nslots + slot;
  // sample of the row, kept incrementally (the stride is fixed): no division per row
  int64_t n = 0, rem = 0, n_have = -1;
  const int64_t dn = ra.g ? stride / ra.S : 0, drem = ra.g ? stride % ra.S : 0;
  if (ra.g) { n = r0 / ra.S; rem = r0 % ra.S; }
#pragma unroll 4
  for (int64_t r = r0; r < M; r += stride) {
    // (walking the rows from the END of the tensor, where the reduce pass that ran just before left the Infinity Cache, was measured in round 4:
    // 464.8 -> 460.5 us per pair at 537 MB -- these passes run at the fabric's rate, cache residency does not help; removed in round 6)
    const int64_t off = (r * nvec + cv) * VEC;
    Vec16<T> g, g2;
    if (da) g = ld16_sel<NT>(da + off);
    if (ra.da2) g2 = ld16_sel<NT>(reinterpret_cast<const T*>(ra.da2) + off);
    const Vec16<T> v = ld16_sel<NT>(y + off);
    if (ra.g) {
      if (n != n_have) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) add[j] = ra.g[n * C + cv * VEC + j] * ra.inv_s;
        n_have = n;
      }
      n += dn; rem += drem;
      if (rem >= ra.S) { rem -= ra.S; ++n; }
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float yv = to_f(v.v[j]);
      const float gin = ((da ? to_f(g.v[j]) : 0.f) + (ra.da2 ? to_f(g2.v[j]) : 0.f)) + add[j];
      const float dz = act_bwd<ACT>(sc[j] * yv + sh[j], gin);
      o.v[j] = from_f<T>(c1[j] * dz + cB[j] * yv + cA[j]);
    }
    st16_sel<NT>(dy + off, o);
  }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_bwd_apply_rc_kernel(const T* __restrict__ da, const T* __restrict__ y, T* __restrict__ dy,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ k1, const float* __restrict__ kB,
                                                              const float* __restrict__ kA, int64_t M, int C, int nt, RowAdd ra) {
  if (nt & 1) bn_bwd_apply_rc_kernel_body<T, ACT, true>(da, y, dy, scale, shift, k1, kB, kA, M, C, ra);
  else bn_bwd_apply_rc_kernel_body<T, ACT, false>(da, y, dy, scale, shift, k1, kB, kA, M, C, ra);
}

#ifndef BN_RED_U
#define BN_RED_U 4
#endif
// First-stage reduction of the backward: per 1024-row tile, per channel: (sum dz, sum dz*xhat).
// Thread = (channel vector, row slot); LDS combine over row slots.  nvec = C/VEC divides 256.
template <typename T, int ACT, bool NT>
__device__ __forceinline__ void bn_bwd_reduce_kernel_body(const T* __restrict__ da, const T* __restrict__ y,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ partial, int64_t M, int C, int tile_rows, RowAdd ra) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [slots][C'][2], C' = max(C, VEC)
  const int tid = threadIdx.x;
  const int nvec = (C == 1) ? 1 : C / VEC;
  const int cv = tid % nvec, slot = tid / nvec, nslots = 256 / nvec;
  // C == 1: a "row" is VEC consecutive voxels of the single channel
  const int64_t rows_total = (C == 1) ? M / VEC : M;
  const int tile = (C == 1) ? tile_rows / VEC : tile_rows;
  const int64_t rbeg = (int64_t)blockIdx.x * tile;
  const int64_t rend = (rbeg + tile < rows_total) ? rbeg + tile : rows_total;
  float s1[VEC], s2[VEC], sc[VEC], sh[VEC], mu[VEC], rs[VEC], add[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = (C == 1) ? 0 : cv * VEC + j;
    s1[j] = 0.f; s2[j] = 0.f;
    sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; rs[j] = rstd[c];
    add[j] = 0.f;
  }
  // row term (C > 1 only, checked by the host): one sample per tile when the tile divides the sample, else looked up per row
  const bool ra_tile = ra.g && ra.S % tile == 0;
  int64_t n_have = -1;
  if (ra_tile) {
    const int64_t n = rbeg / ra.S;
#pragma unroll
    for (int j = 0; j < VEC; ++j) add[j] = ra.g[n * C + cv * VEC + j] * ra.inv_s;
  }
#define BR_ROWADD(r_)                                                                            \
  if (ra.g && !ra_tile) {                                                                        \
    const int64_t n_ = (r_) / ra.S;                                                              \
    if (n_ != n_have) {                                                                          \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) add[j] = ra.g[n_ * C + cv * VEC + j] * ra.inv_s; \
      n_have = n_;                                                                               \
    }                                                                                            \
  }
  // BN_RED_U rows in flight per thread (2 x BN_RED_U 16-byte loads): the tile is streamed once and nothing else hides HBM latency
  constexpr int U = BN_RED_U;
  int64_t r = rbeg + slot;
  if (!ra.g || ra_tile) {
    for (; r + (U - 1) * (int64_t)nslots < rend; r += U * (int64_t)nslots) {
      Vec16<T> g[U], g2[U], v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t off = ((r + u * (int64_t)nslots) * nvec + cv) * VEC;
        if (da) g[u] = ld16_sel<NT>(da + off);
        if (ra.da2) g2[u] = ld16_sel<NT>(reinterpret_cast<const T*>(ra.da2) + off);
        v[u] = ld16_sel<NT>(y + off);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float yv = to_f(v[u].v[j]);
          const float dz = act_bwd<ACT>(sc[j] * yv + sh[j], ((da ? to_f(g[u].v[j]) : 0.f) + (ra.da2 ? to_f(g2[u].v[j]) : 0.f)) + add[j]);
          s1[j] += dz;
          s2[j] += dz * (yv - mu[j]) * rs[j];
        }
    }
  }
  for (; r < rend; r += nslots) {
    const int64_t off = (r * nvec + cv) * VEC;
    Vec16<T> g, g2;
    if (da) g = ld16_sel<NT>(da + off);
    if (ra.da2) g2 = ld16_sel<NT>(reinterpret_cast<const T*>(ra.da2) + off);
    const Vec16<T> v = ld16_sel<NT>(y + off);
    BR_ROWADD(r)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float yv = to_f(v.v[j]);
      const float dz = act_bwd<ACT>(sc[j] * yv + sh[j], ((da ? to_f(g.v[j]) : 0.f) + (ra.da2 ? to_f(g2.v[j]) : 0.f)) + add[j]);
      s1[j] += dz;
      s2[j] += dz * (yv - mu[j]) * rs[j];
    }
  }
#undef BR_ROWADD
  const int Cp = nvec * VEC;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sm[(slot * Cp + cv * VEC + j) * 2 + 0] = s1[j];
    sm[(slot * Cp + cv * VEC + j) * 2 + 1] = s2[j];
  }
  __syncthreads();
  if (C == 1) {
    if (tid == 0) {
      float a = 0.f, b = 0.f;
      for (int s = 0; s < nslots; ++s)
        for (int j = 0; j < VEC; ++j) { a += sm[(s * Cp + j) * 2]; b += sm[(s * Cp + j) * 2 + 1]; }
      partial[(int64_t)blockIdx.x * 2 + 0] = a;
      partial[(int64_t)blockIdx.x * 2 + 1] = b;
    }
  } else {
    for (int c = tid; c < C; c += 256) {
      float a = 0.f, b = 0.f;
      for (int s = 0; s < nslots; ++s) { a += sm[(s * Cp + c) * 2]; b += sm[(s * Cp + c) * 2 + 1]; }
      partial[((int64_t)blockIdx.x * C + c) * 2 + 0] = a;
      partial[((int64_t)blockIdx.x * C + c) * 2 + 1] = b;
    }
  }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const T* __restrict__ da, const T* __restrict__ y,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ partial, int64_t M, int C, int tile_rows, bool nt, RowAdd ra) {
  if (nt) bn_bwd_reduce_kernel_body<T, ACT, true>(da, y, scale, shift, mean, rstd, partial, M, C, tile_rows, ra);
  else bn_bwd_reduce_kernel_body<T, ACT, false>(da, y, scale, shift, mean, rstd, partial, M, C, tile_rows, ra);
}

// ---------------------------------------------------------------------------------------------
// Column sums over row tiles (used by colsum and the global average pool).
//   ws[(blockIdx.y * gridDim.x + blockIdx.x) * C + c] = sum over the tile's rows of v[n=blockIdx.y][row][c]
// ---------------------------------------------------------------------------------------------
// WEIGHTED: every row is multiplied by a float scalar first (rowscale[n][row]): the weight gradient of a 1x1x1 convolution to ONE channel,
// dw[c] = sum_m x[m][c] * dy[m] (OutputTransition.final_conv, models/pcrlv2_model_3d.py:78), as one pass over x.
template <typename T, bool WEIGHTED = false>
__global__ void __launch_bounds__(256) coltile_sum_kernel(const T* __restrict__ v, float* __restrict__ ws, int64_t rows_per_n, int C,
                                                         int tile_rows, const float* __restrict__ rowscale = nullptr) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [slots][C]
  const int tid = threadIdx.x;
  const int nvec = C / VEC;
  const int cv = tid % nvec, slot = tid / nvec, nslots = 256 / nvec;
  const T* base = v + (int64_t)blockIdx.y * rows_per_n * C;
  const int64_t rbeg = (int64_t)blockIdx.x * tile_rows;
  const int64_t rend = (rbeg + tile_rows < rows_per_n) ? rbeg + tile_rows : rows_per_n;
  float s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
  for (int64_t r = rbeg + slot; r < rend; r += nslots) {
    const Vec16<T> x = ld16(base + (r * nvec + cv) * VEC);
    const float wr = WEIGHTED ? rowscale[(int64_t)blockIdx.y * rows_per_n + r] : 1.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] += WEIGHTED ? to_f(x.v[j]) * wr : to_f(x.v[j]);
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) sm[slot * C + cv * VEC + j] = s[j];
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int q = 0; q < nslots; ++q) a += sm[q * C + c];
    ws[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * C + c] = a;
  }
}

// out[n][c] = scale * sum_t ws[n][t][c].  Block = 32 channels x 8 tile slices (fp64 partials, LDS combine, fixed order).
__global__ void __launch_bounds__(256) coltile_finish_kernel(const float* __restrict__ ws, float* __restrict__ out, int tiles, int C,
                                                             int N, double scale) {
  __shared__ double sm[8][33];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, n = blockIdx.y;
  // four independent partial sums (a fixed tree: deterministic): the column sum over the 4 096 tiles of a full-resolution tensor was a
  // chain of 512 dependent loads per thread on two blocks (145 us on the backward's critical chain; OutputTransition's weight gradient)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (c < C) {
    const float* src = ws + (int64_t)n * tiles * C + c;
    int t = sl;
    for (; t + 24 < tiles; t += 32) {
      const float v0 = src[(int64_t)t * C], v1 = src[(int64_t)(t + 8) * C], v2 = src[(int64_t)(t + 16) * C], v3 = src[(int64_t)(t + 24) * C];
      a0 += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
    }
    for (; t < tiles; t += 8) a0 += (double)src[(int64_t)t * C];
  }
  const double s = (a0 + a1) + (a2 + a3);
  sm[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && c < C) {
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) a += sm[q][cl];
    out[(int64_t)n * C + c] = (float)(a * scale);
  }
}

// a = act(bn(y)) and the per-sample column sums of the ROUNDED a in one pass: the global average pool of UpTransition
// (pcrlv2_model_3d.py:67) re-read the full-resolution activation that bn_act_apply had just written.  Block = (row tile, sample) as in
// coltile_sum_kernel; ws[(n * tiles + tile) * C + c], finished by coltile_finish_kernel.
template <typename T, int ACT, bool NT>
__device__ __forceinline__ void bn_apply_gap_body(const T* __restrict__ y, T* __restrict__ a, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, float* __restrict__ ws, int64_t rows_per_n, int C, int tile_rows) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [slots][C]
  const int tid = threadIdx.x;
  const int nvec = C / VEC, cv = tid % nvec, slot = tid / nvec, nslots = 256 / nvec;
  const int64_t base = (int64_t)blockIdx.y * rows_per_n;
  const int64_t rbeg = (int64_t)blockIdx.x * tile_rows;
  const int64_t rend = (rbeg + tile_rows < rows_per_n) ? rbeg + tile_rows : rows_per_n;
  float sc[VEC], sh[VEC], acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sc[j] = scale[cv * VEC + j]; sh[j] = shift[cv * VEC + j]; acc[j] = 0.f; }
#pragma unroll 4
  for (int64_t r = rbeg + slot; r < rend; r += nslots) {
    const int64_t off = ((base + r) * nvec + cv) * VEC;
    const Vec16<T> v = ld16_sel<NT>(y + off);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      o.v[j] = from_f<T>(act_fwd<ACT>(sc[j] * to_f(v.v[j]) + sh[j]));
      acc[j] += to_f(o.v[j]);
    }
    st16_sel<NT>(a + off, o);
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) sm[slot * C + cv * VEC + j] = acc[j];
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float t = 0.f;
    for (int q = 0; q < nslots; ++q) t += sm[q * C + c];
    ws[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * C + c] = t;
  }
}
template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_apply_gap_kernel(const T* __restrict__ y, T* __restrict__ a, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ ws, int64_t rows_per_n, int C,
                                                           int tile_rows, bool nt) {
  if (nt) bn_apply_gap_body<T, ACT, true>(y, a, scale, shift, ws, rows_per_n, C, tile_rows);
  else bn_apply_gap_body<T, ACT, false>(y, a, scale, shift, ws, rows_per_n, C, tile_rows);
}

template <typename T, bool NT>
__global__ void __launch_bounds__(256) gap_bwd_kernel(const float* __restrict__ dg, const T* add, T* da, int64_t S,
                                                      int C, int64_t nvec_total, float inv_s) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec_total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % nvec);
    const int64_t n = (i / nvec) / S;
    const float* gr = dg + n * C + cv * VEC;
    Vec16<T> o;
    if (add) {
      const Vec16<T> old = ld16_sel<NT>(add + i * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(to_f(old.v[j]) + gr[j] * inv_s);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(gr[j] * inv_s);
    }
    st16_sel<NT>(da + i * VEC, o);
  }
}

// ---------------------------------------------------------------------------------------------
// MaxPool3d(2): thread = (output voxel, channel vector)
// ---------------------------------------------------------------------------------------------
template <typename T, bool NT>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, Dims g, int C, int64_t total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC;
  const Dims go{g.N, g.D / 2, g.H / 2, g.W / 2};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % nvec);
    int n, d, h, w;
    decode_voxel(i / nvec, go, n, d, h, w);
    float m[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int64_t row = (((int64_t)n * g.D + 2 * d + (t >> 2)) * g.H + 2 * h + ((t >> 1) & 1)) * g.W + 2 * w + (t & 1);
      const Vec16<T> v = ld16_sel<NT>(x + row * C + cv * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float f = to_f(v.v[j]);
        if (f > m[j] || f != f) m[j] = f;
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(m[j]);
    st16(y + i * VEC, o);
  }
}

template <typename T, bool NT>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, Dims g,
                                                          int C, int64_t total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC;
  const Dims go{g.N, g.D / 2, g.H / 2, g.W / 2};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % nvec);
    int n, d, h, w;
    decode_voxel(i / nvec, go, n, d, h, w);
    float m[VEC];
    int arg[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { m[j] = -INFINITY; arg[j] = 0; }
    int64_t rows[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      rows[t] = (((int64_t)n * g.D + 2 * d + (t >> 2)) * g.H + 2 * h + ((t >> 1) & 1)) * g.W + 2 * w + (t & 1);
      const Vec16<T> v = ld16_sel<NT>(x + rows[t] * C + cv * VEC);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float f = to_f(v.v[j]);
        if (f > m[j] || f != f) { m[j] = f; arg[j] = t; }  // strict '>' keeps the FIRST maximum in scan order
      }
    }
    const Vec16<T> gy = ld16(dy + i * VEC);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.v[j] = (arg[j] == t) ? gy.v[j] : from_f<T>(0.f);
      st16_sel<NT>(dx + rows[t] * C + cv * VEC, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm backward of a LUConv whose activation a = act(bn(y)) is consumed through MaxPool3d(2) only (the second LUConv of an
// encoder stage: pcrlv2_model_3d.py:114-117, the skip tensors are never used).  autograd runs max_pool3d_backward
// (read a, write the full-resolution gradient), then the two BatchNorm passes read that gradient and y again: 7 full-resolution
// tensor passes.  Here thread = (pooled voxel, channel vector): the eight y vectors of the window are loaded, a is RECOMPUTED from
// them exactly as the forward stored it (same expression, same rounding), the first maximum in scan order takes the pooled
// gradient (max_pool3d_backward's rule; NaN wins like in aten), and the passes use that directly -- the full-resolution gradient
// is never written or read: 3 full-resolution passes (y twice, dy once) plus the pooled gradient.
// ---------------------------------------------------------------------------------------------
template <typename T, int ACT, bool NT>
__device__ __forceinline__ void pool_window(const T* __restrict__ y, const Dims& g, int C, int cv, int64_t pv, Vec16<T> (&v)[8], int64_t (&rows)[8]) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const Dims go{g.N, g.D / 2, g.H / 2, g.W / 2};
  int n, d, h, w;
  decode_voxel(pv, go, n, d, h, w);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    rows[t] = (((int64_t)n * g.D + 2 * d + (t >> 2)) * g.H + 2 * h + ((t >> 1) & 1)) * g.W + 2 * w + (t & 1);
    v[t] = ld16_sel<NT>(y + rows[t] * C + cv * VEC);
  }
}
// channel j of the window: -> index of the first maximum of a = round_T(act(sc * y + sh)), and dz of that element for pooled gradient gp
template <typename T, int ACT>
__device__ __forceinline__ int pool_argmax(const Vec16<T> (&v)[8], int j, float sc, float sh, float gp, float& dz, float& ybest) {
  float m = -INFINITY, zb = 0.f;
  int arg = 0;
  ybest = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float yv = to_f(v[t].v[j]);
    const float z = sc * yv + sh;
    const float a = to_f(from_f<T>(act_fwd<ACT>(z)));
    if (a > m || a != a) { m = a; arg = t; zb = z; ybest = yv; }
  }
  dz = act_bwd<ACT>(zb, gp);
  return arg;
}

// Forward counterpart: a = act(bn(y)) and p = MaxPool3d(2)(a) in one pass (thread = pooled voxel x channel vector): the separate
// pool re-read the full-resolution activation that bn_act_apply had just written.
template <typename T, int ACT, bool NT>
__device__ __forceinline__ void bn_apply_pool_body(const T* __restrict__ y, T* __restrict__ a, T* __restrict__ p, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, Dims g, int C) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC, cv = threadIdx.x % nvec;
  const int64_t Mp = (int64_t)g.N * (g.D / 2) * (g.H / 2) * (g.W / 2);
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sc[j] = scale[cv * VEC + j]; sh[j] = shift[cv * VEC + j]; }
  const int64_t total = Mp * nvec;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    Vec16<T> v[8];
    int64_t rows[8];
    pool_window<T, ACT, NT>(y, g, C, cv, i / nvec, v, rows);
    float m[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        o.v[j] = from_f<T>(act_fwd<ACT>(sc[j] * to_f(v[t].v[j]) + sh[j]));
        const float f = to_f(o.v[j]);
        if (f > m[j] || f != f) m[j] = f;
      }
      if (a) st16_sel<NT>(a + rows[t] * C + cv * VEC, o);      // a == nullptr (kernel argument: uniform): only the pooled tensor is wanted
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(m[j]);
    st16(p + i * VEC, o);
  }
}
template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_apply_pool_kernel(const T* __restrict__ y, T* __restrict__ a, T* __restrict__ p,
                                                            const float* __restrict__ scale, const float* __restrict__ shift, Dims g, int C, bool nt) {
  if (nt) bn_apply_pool_body<T, ACT, true>(y, a, p, scale, shift, g, C);
  else bn_apply_pool_body<T, ACT, false>(y, a, p, scale, shift, g, C);
}

template <typename T, int ACT, bool NT>
__device__ __forceinline__ void bn_bwd_reduce_pool_body(const T* __restrict__ dp, const T* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ partial, Dims g, int C, int tile_p) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [slots][C][2]
  const int tid = threadIdx.x;
  const int nvec = C / VEC, cv = tid % nvec, slot = tid / nvec, nslots = 256 / nvec;
  const int64_t Mp = (int64_t)g.N * (g.D / 2) * (g.H / 2) * (g.W / 2);
  const int64_t pbeg = (int64_t)blockIdx.x * tile_p;
  const int64_t pend = (pbeg + tile_p < Mp) ? pbeg + tile_p : Mp;
  float s1[VEC], s2[VEC], sc[VEC], sh[VEC], mu[VEC], rs[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = cv * VEC + j;
    s1[j] = 0.f; s2[j] = 0.f;
    sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; rs[j] = rstd[c];
  }
  for (int64_t pv = pbeg + slot; pv < pend; pv += nslots) {
    Vec16<T> v[8];
    int64_t rows[8];
    pool_window<T, ACT, NT>(y, g, C, cv, pv, v, rows);
    const Vec16<T> gp = ld16(dp + (pv * nvec + cv) * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float dz, yb;
      pool_argmax<T, ACT>(v, j, sc[j], sh[j], to_f(gp.v[j]), dz, yb);
      s1[j] += dz;
      s2[j] += dz * (yb - mu[j]) * rs[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sm[(slot * C + cv * VEC + j) * 2 + 0] = s1[j];
    sm[(slot * C + cv * VEC + j) * 2 + 1] = s2[j];
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int q = 0; q < nslots; ++q) { a += sm[(q * C + c) * 2]; b += sm[(q * C + c) * 2 + 1]; }
    partial[((int64_t)blockIdx.x * C + c) * 2 + 0] = a;
    partial[((int64_t)blockIdx.x * C + c) * 2 + 1] = b;
  }
}
template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_bwd_reduce_pool_kernel(const T* __restrict__ dp, const T* __restrict__ y, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, float* __restrict__ partial, Dims g, int C,
                                                                 int tile_p, bool nt) {
  if (nt) bn_bwd_reduce_pool_body<T, ACT, true>(dp, y, scale, shift, mean, rstd, partial, g, C, tile_p);
  else bn_bwd_reduce_pool_body<T, ACT, false>(dp, y, scale, shift, mean, rstd, partial, g, C, tile_p);
}

template <typename T, int ACT, bool NT>
__device__ __forceinline__ void bn_bwd_apply_pool_body(const T* __restrict__ dp, const T* __restrict__ y, T* __restrict__ dy,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ k1, const float* __restrict__ kB,
                                                         const float* __restrict__ kA, Dims g, int C) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC, cv = threadIdx.x % nvec;   // 256 % nvec == 0: the channel vector of a thread never changes
  const int64_t Mp = (int64_t)g.N * (g.D / 2) * (g.H / 2) * (g.W / 2);
  float sc[VEC], sh[VEC], c1[VEC], cB[VEC], cA[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = cv * VEC + j;
    sc[j] = scale[c]; sh[j] = shift[c]; c1[j] = k1[c]; cB[j] = kB[c]; cA[j] = kA[c];
  }
  const int64_t total = Mp * nvec;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t pv = i / nvec;
    Vec16<T> v[8];
    int64_t rows[8];
    pool_window<T, ACT, NT>(y, g, C, cv, pv, v, rows);
    const Vec16<T> gp = ld16(dp + i * VEC);
    int arg[VEC];
    float dzb[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float yb;
      arg[j] = pool_argmax<T, ACT>(v, j, sc[j], sh[j], to_f(gp.v[j]), dzb[j], yb);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      Vec16<T> o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float dz = (arg[j] == t) ? dzb[j] : 0.f;
        o.v[j] = from_f<T>(c1[j] * dz + cB[j] * to_f(v[t].v[j]) + cA[j]);
      }
      st16_sel<NT>(dy + rows[t] * C + cv * VEC, o);
    }
  }
}
template <typename T, int ACT>
__global__ void __launch_bounds__(256) bn_bwd_apply_pool_kernel(const T* __restrict__ dp, const T* __restrict__ y, T* __restrict__ dy,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ k1, const float* __restrict__ kB,
                                                                const float* __restrict__ kA, Dims g, int C, bool nt) {
  if (nt) bn_bwd_apply_pool_body<T, ACT, true>(dp, y, dy, scale, shift, k1, kB, kA, g, C);
  else bn_bwd_apply_pool_body<T, ACT, false>(dp, y, dy, scale, shift, k1, kB, kA, g, C);
}

inline unsigned grid_for(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// grid for the register-cached streaming kernels: enough blocks to fill the chip (8 per CU), never more rows than exist
inline unsigned rc_grid(int64_t M, int nvec) {
  const int nslots = (nvec > 0 && nvec <= 256) ? 256 / nvec : 1;
  int64_t b = (M + nslots - 1) / nslots;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}

int check_vec(const char* what, int C, int dtype, bool allow_c1) {
  if (dtype != PCRL_F32 && dtype != PCRL_BF16) return pcrl_fail(PCRL_EINVAL, "%s: bad dtype %d", what, dtype);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  if (C == 1 && allow_c1 && dtype == PCRL_F32) return 0;
  if (C <= 0 || C % vec != 0) return pcrl_fail(PCRL_EINVAL, "%s: C=%d must be a multiple of %d", what, C, vec);
  return 0;
}
int check_tilevec(const char* what, int C, int dtype, bool allow_c1) {
  if (int e = check_vec(what, C, dtype, allow_c1)) return e;
  if (C == 1) return 0;
  const int nvec = C / (dtype == PCRL_BF16 ? 8 : 4);
  if (nvec > 256 || 256 % nvec != 0) return pcrl_fail(PCRL_EINVAL, "%s: C=%d: channel vectors (%d) must divide 256", what, C, nvec);
  return 0;
}

}  // namespace

extern "C" int pcrl_bn_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps,
                                float* mean, float* rstd, float* scale, float* shift, pcrl_stream_t stream) {
  PCRL_REQUIRE(partial && gamma && beta && mean && rstd && scale && shift, "bn_finalize: null pointer");
  PCRL_REQUIRE(rows > 0 && C > 0 && count > 0, "bn_finalize: bad sizes rows=%d C=%d", rows, C);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, as_stream(stream), partial, rows, C, count, gamma, beta,
                     running_mean, running_var, momentum, eps, mean, rstd, scale, shift);
  return pcrl_check_launch("bn_finalize");
}

#define DISPATCH_ACT_T(KERNEL, GRID, LDS, ...)                                                                                   \
  do {                                                                                                                           \
    if (act == PCRL_ACT_RELU) hipLaunchKernelGGL((KERNEL<T, PCRL_ACT_RELU>), GRID, dim3(256), LDS, as_stream(stream), __VA_ARGS__); \
    else if (act == PCRL_ACT_SIGMOID) hipLaunchKernelGGL((KERNEL<T, PCRL_ACT_SIGMOID>), GRID, dim3(256), LDS, as_stream(stream), __VA_ARGS__); \
    else if (act == PCRL_ACT_SILU) hipLaunchKernelGGL((KERNEL<T, PCRL_ACT_SILU>), GRID, dim3(256), LDS, as_stream(stream), __VA_ARGS__); \
    else if (act == PCRL_ACT_ELU) hipLaunchKernelGGL((KERNEL<T, PCRL_ACT_ELU>), GRID, dim3(256), LDS, as_stream(stream), __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<T, PCRL_ACT_NONE>), GRID, dim3(256), LDS, as_stream(stream), __VA_ARGS__);                    \
  } while (0)

extern "C" int pcrl_bn_act_apply(const void* y, void* a, const float* scale, const float* shift,
                                 int64_t M, int C, int act, int dtype, pcrl_stream_t stream) {
  if (int e = check_vec("bn_act_apply", C, dtype, true)) return e;
  PCRL_REQUIRE(y && a && scale && shift, "bn_act_apply: null pointer");
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  PCRL_REQUIRE((M * C) % vec == 0, "bn_act_apply: M*C must be a multiple of %d", vec);
  const int64_t nvec = M * C / vec;
  const dim3 grid(grid_for(nvec));
  const bool rc = C % vec == 0 && (C / vec) <= 256 && 256 % (C / vec) == 0;
  const dim3 grid_rc(rc_grid(M, C / (C % vec == 0 ? vec : 1)));
  if (dtype == PCRL_BF16) {
    using T = bf16;
    if (rc) DISPATCH_ACT_T(bn_apply_rc_kernel, grid_rc, 0, (const T*)y, (T*)a, scale, shift, M, C, pcrl_streaming(M * C * (int64_t)sizeof(T)));
    else DISPATCH_ACT_T(bn_apply_kernel, grid, 0, (const T*)y, (T*)a, scale, shift, nvec, C);
  } else {
    using T = float;
    if (rc) DISPATCH_ACT_T(bn_apply_rc_kernel, grid_rc, 0, (const T*)y, (T*)a, scale, shift, M, C, pcrl_streaming(M * C * (int64_t)sizeof(T)));
    else DISPATCH_ACT_T(bn_apply_kernel, grid, 0, (const T*)y, (T*)a, scale, shift, nvec, C);
  }
  return pcrl_check_launch("bn_act_apply");
}

// Rows per first-stage partial of the BatchNorm backward: 1024 for the big volumes, fewer for small ones so that the
// reduction still spreads over >= ~1000 blocks (M = 65536 used to run on 64 of the 256 CUs).
static int bn_bwd_tile_rows(int64_t M) {
  int t = TILE_ROWS;
  while (t > 32 && M / t < 1024) t >>= 1;
  return t;
}
extern "C" int64_t pcrl_bn_bwd_partial_rows(int64_t M) {
  const int t = bn_bwd_tile_rows(M);
  return (M + t - 1) / t;
}

static bool rowadd_ok(int C, int dtype) {
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  return C > 1 && C % vec == 0 && (C / vec) <= 256 && 256 % (C / vec) == 0;
}
extern "C" int64_t pcrl_bn_act_bwd_rowadd_ok(int C, int dtype) { return rowadd_ok(C, dtype) ? 1 : 0; }

static int bn_bwd_reduce_impl(const void* da, const void* y, const float* scale, const float* shift, const float* mean, const float* rstd,
                              float* partial, int64_t M, int C, int act, int dtype, RowAdd ra, pcrl_stream_t stream) {
  if (int e = check_tilevec("bn_act_bwd_reduce", C, dtype, true)) return e;
  PCRL_REQUIRE((da || ra.g || ra.da2) && y && scale && shift && mean && rstd && partial, "bn_act_bwd_reduce: null pointer");
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  PCRL_REQUIRE(C != 1 || M % vec == 0, "bn_act_bwd_reduce: M must be a multiple of %d for C == 1", vec);
  const dim3 grid((unsigned)pcrl_bn_bwd_partial_rows(M));
  const int nvec = C == 1 ? 1 : C / vec;
  const size_t lds = (size_t)(256 / nvec) * (nvec * vec) * 2 * sizeof(float);
  if (dtype == PCRL_BF16) {
    using T = bf16;
    DISPATCH_ACT_T(bn_bwd_reduce_kernel, grid, lds, (const T*)da, (const T*)y, scale, shift, mean, rstd, partial, M, C, bn_bwd_tile_rows(M), pcrl_streaming(M * C * (int64_t)sizeof(T)), ra);
  } else {
    using T = float;
    DISPATCH_ACT_T(bn_bwd_reduce_kernel, grid, lds, (const T*)da, (const T*)y, scale, shift, mean, rstd, partial, M, C, bn_bwd_tile_rows(M), pcrl_streaming(M * C * (int64_t)sizeof(T)), ra);
  }
  return pcrl_check_launch("bn_act_bwd_reduce");
}
extern "C" int pcrl_bn_act_bwd_reduce(const void* da, const void* y, const float* scale, const float* shift,
                                      const float* mean, const float* rstd, float* partial,
                                      int64_t M, int C, int act, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(da, "bn_act_bwd_reduce: null pointer");
  return bn_bwd_reduce_impl(da, y, scale, shift, mean, rstd, partial, M, C, act, dtype, RowAdd{nullptr, 1, 0.f, nullptr}, stream);
}
static int rowadd_check(const char* what, const float* row_g, int N, int64_t S, int64_t M, int C, int dtype, RowAdd& ra) {
  PCRL_REQUIRE(row_g && N > 0 && S > 0 && (int64_t)N * S == M, "%s: the row term needs N * S == M (N=%d S=%lld M=%lld)", what, N, (long long)S, (long long)M);
  PCRL_REQUIRE(rowadd_ok(C, dtype), "%s: the row term is not available for C=%d (pcrl_bn_act_bwd_rowadd_ok)", what, C);
  ra = RowAdd{row_g, S, (float)(1.0 / (double)S), nullptr};
  return PCRL_OK;
}
extern "C" int pcrl_bn_act_bwd_reduce_rowadd(const void* da, const float* row_g, int N, int64_t S, const void* y, const float* scale,
                                             const float* shift, const float* mean, const float* rstd, float* partial,
                                             int64_t M, int C, int act, int dtype, pcrl_stream_t stream) {
  RowAdd ra;
  if (int e = rowadd_check("bn_act_bwd_reduce_rowadd", row_g, N, S, M, C, dtype, ra)) return e;
  return bn_bwd_reduce_impl(da, y, scale, shift, mean, rstd, partial, M, C, act, dtype, ra, stream);
}

extern "C" int pcrl_bn_bwd_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* mean,
                                    const float* rstd, float* dgamma, float* dbeta, float* k1, float* kB, float* kA,
                                    pcrl_stream_t stream) {
  PCRL_REQUIRE(partial && gamma && mean && rstd && dgamma && dbeta && k1 && kB && kA, "bn_bwd_finalize: null pointer");
  PCRL_REQUIRE(rows > 0 && C > 0 && count > 0, "bn_bwd_finalize: bad sizes rows=%d C=%d", rows, C);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, as_stream(stream), partial, rows, C, count, gamma, mean, rstd,
                     dgamma, dbeta, k1, kB, kA);
  return pcrl_check_launch("bn_bwd_finalize");
}

static int bn_bwd_apply_impl(const void* da, const void* y, void* dy, const float* scale, const float* shift,
                             const float* k1, const float* kB, const float* kA,
                             int64_t M, int C, int act, int dtype, RowAdd ra, pcrl_stream_t stream) {
  if (int e = check_vec("bn_act_bwd_apply", C, dtype, true)) return e;
  PCRL_REQUIRE((da || ra.g || ra.da2) && y && dy && scale && shift && k1 && kB && kA, "bn_act_bwd_apply: null pointer");
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  PCRL_REQUIRE((M * C) % vec == 0, "bn_act_bwd_apply: M*C must be a multiple of %d", vec);
  const int64_t nvec = M * C / vec;
  const dim3 grid(grid_for(nvec));
  const bool rc = C % vec == 0 && (C / vec) <= 256 && 256 % (C / vec) == 0;
  PCRL_REQUIRE(rc || (!ra.g && !ra.da2), "bn_act_bwd_apply: the row term / second gradient need the channel-vector kernel (C=%d)", C);
  const dim3 grid_rc(rc_grid(M, C / (C % vec == 0 ? vec : 1)));
  if (dtype == PCRL_BF16) {
    using T = bf16;
    if (rc) DISPATCH_ACT_T(bn_bwd_apply_rc_kernel, grid_rc, 0, (const T*)da, (const T*)y, (T*)dy, scale, shift, k1, kB, kA, M, C, (int)pcrl_streaming(M * C * (int64_t)sizeof(T)), ra);
    else DISPATCH_ACT_T(bn_bwd_apply_kernel, grid, 0, (const T*)da, (const T*)y, (T*)dy, scale, shift, k1, kB, kA, nvec, C);
  } else {
    using T = float;
    if (rc) DISPATCH_ACT_T(bn_bwd_apply_rc_kernel, grid_rc, 0, (const T*)da, (const T*)y, (T*)dy, scale, shift, k1, kB, kA, M, C, (int)pcrl_streaming(M * C * (int64_t)sizeof(T)), ra);
    else DISPATCH_ACT_T(bn_bwd_apply_kernel, grid, 0, (const T*)da, (const T*)y, (T*)dy, scale, shift, k1, kB, kA, nvec, C);
  }
  return pcrl_check_launch("bn_act_bwd_apply");
}
extern "C" int pcrl_bn_act_bwd_apply(const void* da, const void* y, void* dy, const float* scale, const float* shift,
                                     const float* k1, const float* kB, const float* kA,
                                     int64_t M, int C, int act, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(da, "bn_act_bwd_apply: null pointer");
  return bn_bwd_apply_impl(da, y, dy, scale, shift, k1, kB, kA, M, C, act, dtype, RowAdd{nullptr, 1, 0.f, nullptr}, stream);
}
extern "C" int pcrl_bn_act_bwd_apply_rowadd(const void* da, const float* row_g, int N, int64_t S, const void* y, void* dy, const float* scale,
                                            const float* shift, const float* k1, const float* kB, const float* kA,
                                            int64_t M, int C, int act, int dtype, pcrl_stream_t stream) {
  RowAdd ra;
  if (int e = rowadd_check("bn_act_bwd_apply_rowadd", row_g, N, S, M, C, dtype, ra)) return e;
  return bn_bwd_apply_impl(da, y, dy, scale, shift, k1, kB, kA, M, C, act, dtype, ra, stream);
}

// ---- the incoming gradient is a SUM: da (+ da2) (+ row_g[n] / S broadcast over the sample) -- any subset, at least one ----
// A 2D decoder block's output has three consumers (the next block, its own deep-supervision head, the pooled projection head:
// pcrlv2_model.py:119-127); autograd would materialise their sum with two element-wise adds over the full-resolution tensor.
static int sum_check(const char* what, const void* da, const void* da2, const float* row_g, int N, int64_t S, int64_t M, int C, int dtype, RowAdd& ra) {
  PCRL_REQUIRE(da || da2 || row_g, "%s: no gradient at all", what);
  PCRL_REQUIRE(rowadd_ok(C, dtype), "%s: not available for C=%d (pcrl_bn_act_bwd_rowadd_ok)", what, C);
  ra = RowAdd{nullptr, 1, 0.f, da2};
  if (row_g) {
    PCRL_REQUIRE(N > 0 && S > 0 && (int64_t)N * S == M, "%s: the row term needs N * S == M (N=%d S=%lld M=%lld)", what, N, (long long)S, (long long)M);
    ra.g = row_g; ra.S = S; ra.inv_s = (float)(1.0 / (double)S);
  }
  return PCRL_OK;
}
extern "C" int pcrl_bn_act_bwd_reduce_sum(const void* da, const void* da2, const float* row_g, int N, int64_t S, const void* y, const float* scale,
                                          const float* shift, const float* mean, const float* rstd, float* partial,
                                          int64_t M, int C, int act, int dtype, pcrl_stream_t stream) {
  RowAdd ra;
  if (int e = sum_check("bn_act_bwd_reduce_sum", da, da2, row_g, N, S, M, C, dtype, ra)) return e;
  return bn_bwd_reduce_impl(da, y, scale, shift, mean, rstd, partial, M, C, act, dtype, ra, stream);
}
extern "C" int pcrl_bn_act_bwd_apply_sum(const void* da, const void* da2, const float* row_g, int N, int64_t S, const void* y, void* dy,
                                         const float* scale, const float* shift, const float* k1, const float* kB, const float* kA,
                                         int64_t M, int C, int act, int dtype, pcrl_stream_t stream) {
  RowAdd ra;
  if (int e = sum_check("bn_act_bwd_apply_sum", da, da2, row_g, N, S, M, C, dtype, ra)) return e;
  return bn_bwd_apply_impl(da, y, dy, scale, shift, k1, kB, kA, M, C, act, dtype, ra, stream);
}

// ---- BatchNorm backward with MaxPool3d(2) backward folded in (see bn_bwd_reduce_pool_body) ----
static int bn_pool_tile(int64_t Mp) {
  int t = TILE_ROWS / 8;   // 128 pooled voxels = 1024 rows
  while (t > 4 && Mp / t < 1024) t >>= 1;
  return t;
}
extern "C" int64_t pcrl_bn_act_bwd_pool_ok(int D, int H, int W, int C, int dtype) {
  return (D > 0 && H > 0 && W > 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && rowadd_ok(C, dtype)) ? 1 : 0;
}
extern "C" int64_t pcrl_bn_act_bwd_pool_partial_rows(int N, int D, int H, int W) {
  const int64_t Mp = (int64_t)N * (D / 2) * (H / 2) * (W / 2);
  const int t = bn_pool_tile(Mp);
  return (Mp + t - 1) / t;
}
extern "C" int pcrl_bn_act_apply_pool(const void* y, void* a, void* p, const float* scale, const float* shift, int N, int D, int H, int W, int C,
                                      int act, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(y && p && scale && shift && N > 0, "bn_act_apply_pool: bad arguments");   // a may be null: the full-resolution activation is not stored
  PCRL_REQUIRE(pcrl_bn_act_bwd_pool_ok(D, H, W, C, dtype), "bn_act_apply_pool: not available for %dx%dx%d, C=%d (pcrl_bn_act_bwd_pool_ok)", D, H, W, C);
  const int64_t Mp = (int64_t)N * (D / 2) * (H / 2) * (W / 2);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const dim3 grid(grid_for(Mp * (C / vec)));
  const Dims g{N, D, H, W};
  const bool nt = pcrl_streaming(Mp * 8 * C * (dtype == PCRL_BF16 ? 2 : 4));
  if (dtype == PCRL_BF16) {
    using T = bf16;
    DISPATCH_ACT_T(bn_apply_pool_kernel, grid, 0, (const T*)y, (T*)a, (T*)p, scale, shift, g, C, nt);
  } else {
    using T = float;
    DISPATCH_ACT_T(bn_apply_pool_kernel, grid, 0, (const T*)y, (T*)a, (T*)p, scale, shift, g, C, nt);
  }
  return pcrl_check_launch("bn_act_apply_pool");
}
extern "C" int pcrl_bn_act_bwd_reduce_pool(const void* dp, const void* y, const float* scale, const float* shift, const float* mean,
                                           const float* rstd, float* partial, int N, int D, int H, int W, int C, int act, int dtype,
                                           pcrl_stream_t stream) {
  PCRL_REQUIRE(dp && y && scale && shift && mean && rstd && partial && N > 0, "bn_act_bwd_reduce_pool: bad arguments");
  PCRL_REQUIRE(pcrl_bn_act_bwd_pool_ok(D, H, W, C, dtype), "bn_act_bwd_reduce_pool: not available for %dx%dx%d, C=%d (pcrl_bn_act_bwd_pool_ok)", D, H, W, C);
  const int64_t Mp = (int64_t)N * (D / 2) * (H / 2) * (W / 2);
  const dim3 grid((unsigned)pcrl_bn_act_bwd_pool_partial_rows(N, D, H, W));
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const size_t lds = (size_t)(256 / (C / vec)) * C * 2 * sizeof(float);
  const Dims g{N, D, H, W};
  const bool nt = pcrl_streaming(Mp * 8 * C * (dtype == PCRL_BF16 ? 2 : 4));
  if (dtype == PCRL_BF16) {
    using T = bf16;
    DISPATCH_ACT_T(bn_bwd_reduce_pool_kernel, grid, lds, (const T*)dp, (const T*)y, scale, shift, mean, rstd, partial, g, C, bn_pool_tile(Mp), nt);
  } else {
    using T = float;
    DISPATCH_ACT_T(bn_bwd_reduce_pool_kernel, grid, lds, (const T*)dp, (const T*)y, scale, shift, mean, rstd, partial, g, C, bn_pool_tile(Mp), nt);
  }
  return pcrl_check_launch("bn_act_bwd_reduce_pool");
}
extern "C" int pcrl_bn_act_bwd_apply_pool(const void* dp, const void* y, void* dy, const float* scale, const float* shift, const float* k1,
                                          const float* kB, const float* kA, int N, int D, int H, int W, int C, int act, int dtype,
                                          pcrl_stream_t stream) {
  PCRL_REQUIRE(dp && y && dy && scale && shift && k1 && kB && kA && N > 0, "bn_act_bwd_apply_pool: bad arguments");
  PCRL_REQUIRE(pcrl_bn_act_bwd_pool_ok(D, H, W, C, dtype), "bn_act_bwd_apply_pool: not available for %dx%dx%d, C=%d (pcrl_bn_act_bwd_pool_ok)", D, H, W, C);
  const int64_t Mp = (int64_t)N * (D / 2) * (H / 2) * (W / 2);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const dim3 grid(grid_for(Mp * (C / vec)));
  const Dims g{N, D, H, W};
  const bool nt = pcrl_streaming(Mp * 8 * C * (dtype == PCRL_BF16 ? 2 : 4));
  if (dtype == PCRL_BF16) {
    using T = bf16;
    DISPATCH_ACT_T(bn_bwd_apply_pool_kernel, grid, 0, (const T*)dp, (const T*)y, (T*)dy, scale, shift, k1, kB, kA, g, C, nt);
  } else {
    using T = float;
    DISPATCH_ACT_T(bn_bwd_apply_pool_kernel, grid, 0, (const T*)dp, (const T*)y, (T*)dy, scale, shift, k1, kB, kA, g, C, nt);
  }
  return pcrl_check_launch("bn_act_bwd_apply_pool");
}

extern "C" int pcrl_maxpool3d_2_fwd(const void* x, void* y, int N, int D, int H, int W, int C, int dtype, pcrl_stream_t stream) {
  if (int e = check_vec("maxpool3d_2_fwd", C, dtype, false)) return e;
  PCRL_REQUIRE(x && y, "maxpool3d_2_fwd: null pointer");
  PCRL_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && D > 0 && H > 0 && W > 0, "maxpool3d_2_fwd: dims must be even (%d %d %d)", D, H, W);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const int64_t total = (int64_t)N * (D / 2) * (H / 2) * (W / 2) * (C / vec);
  const Dims g{N, D, H, W};
  const bool nt = pcrl_streaming((int64_t)N * D * H * W * C * (dtype == PCRL_BF16 ? 2 : 4));
#define MP_FWD(T_, NT_) hipLaunchKernelGGL((maxpool_fwd_kernel<T_, NT_>), dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const T_*)x, (T_*)y, g, C, total)
  if (dtype == PCRL_BF16) { if (nt) MP_FWD(bf16, true); else MP_FWD(bf16, false); }
  else { if (nt) MP_FWD(float, true); else MP_FWD(float, false); }
#undef MP_FWD
  return pcrl_check_launch("maxpool_fwd");
}

extern "C" int pcrl_maxpool3d_2_bwd(const void* x, const void* dy, void* dx, int N, int D, int H, int W, int C, int dtype, pcrl_stream_t stream) {
  if (int e = check_vec("maxpool3d_2_bwd", C, dtype, false)) return e;
  PCRL_REQUIRE(x && dy && dx, "maxpool3d_2_bwd: null pointer");
  PCRL_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0 && D > 0 && H > 0 && W > 0, "maxpool3d_2_bwd: dims must be even (%d %d %d)", D, H, W);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const int64_t total = (int64_t)N * (D / 2) * (H / 2) * (W / 2) * (C / vec);
  const Dims g{N, D, H, W};
  const bool nt = pcrl_streaming((int64_t)N * D * H * W * C * (dtype == PCRL_BF16 ? 2 : 4));
#define MP_BWD(T_, NT_) hipLaunchKernelGGL((maxpool_bwd_kernel<T_, NT_>), dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const T_*)x, (const T_*)dy, (T_*)dx, g, C, total)
  if (dtype == PCRL_BF16) { if (nt) MP_BWD(bf16, true); else MP_BWD(bf16, false); }
  else { if (nt) MP_BWD(float, true); else MP_BWD(float, false); }
#undef MP_BWD
  return pcrl_check_launch("maxpool_bwd");
}

// Rows per first-stage tile of the column sums: 1024, halved (down to 32) while the grid would have fewer than 512 blocks.
static int coltile_rows(int N, int64_t S) {
  int t = TILE_ROWS;
  while (t > 32 && (int64_t)N * ((S + t - 1) / t) < 512) t >>= 1;
  return t;
}
static int64_t coltile_tiles(int N, int64_t S) {
  const int t = coltile_rows(N, S);
  return (S + t - 1) / t;
}
extern "C" size_t pcrl_colsum_ws_bytes(int64_t M, int C) { return (size_t)coltile_tiles(1, M) * C * sizeof(float); }

static int coltile_launch(const void* v, float* out, void* ws, size_t ws_bytes, int N, int64_t S, int C, int dtype, double scale,
                          hipStream_t stream, const char* what) {
  if (int e = check_tilevec(what, C, dtype, false)) return e;
  const int64_t tiles = coltile_tiles(N, S);
  const size_t need = (size_t)N * tiles * C * sizeof(float);
  if (!ws || ws_bytes < need) return pcrl_fail(PCRL_EWORKSPACE, "%s: workspace %zu < %zu", what, ws_bytes, need);
  const size_t lds = (size_t)(256 / (C / (dtype == PCRL_BF16 ? 8 : 4))) * C * sizeof(float);
  const dim3 grid((unsigned)tiles, (unsigned)N);
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(coltile_sum_kernel<bf16>, grid, dim3(256), lds, stream, (const bf16*)v, (float*)ws, S, C, coltile_rows(N, S));
  else hipLaunchKernelGGL(coltile_sum_kernel<float>, grid, dim3(256), lds, stream, (const float*)v, (float*)ws, S, C, coltile_rows(N, S));
  if (int e = pcrl_check_launch(what)) return e;
  hipLaunchKernelGGL(coltile_finish_kernel, dim3((C + 31) / 32, N), dim3(256), 0, stream, (const float*)ws, out, (int)tiles, C, N, scale);
  return pcrl_check_launch(what);
}

extern "C" int pcrl_colsum(const void* v, float* out, void* ws, size_t ws_bytes, int64_t M, int C, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(v && out, "colsum: null pointer");
  return coltile_launch(v, out, ws, ws_bytes, 1, M, C, dtype, 1.0, as_stream(stream), "colsum");
}

// out[c] = sum_m v[m][c] * rowscale[m]  (internal: the taps == 1 path of pcrl_conv3d_to1_wgrad, conv_wgrad.hip)
size_t pcrl_weighted_colsum_ws_bytes(int64_t M, int C) { return pcrl_colsum_ws_bytes(M, C); }
int pcrl_weighted_colsum(const void* v, const float* rowscale, float* out, void* ws, size_t ws_bytes, int64_t M, int C, int dtype, hipStream_t stream) {
  if (int e = check_tilevec("weighted_colsum", C, dtype, false)) return e;
  const int64_t tiles = coltile_tiles(1, M);
  const size_t need = (size_t)tiles * C * sizeof(float);
  if (!ws || ws_bytes < need) return pcrl_fail(PCRL_EWORKSPACE, "weighted_colsum: workspace %zu < %zu", ws_bytes, need);
  const size_t lds = (size_t)(256 / (C / (dtype == PCRL_BF16 ? 8 : 4))) * C * sizeof(float);
  const dim3 grid((unsigned)tiles, 1);
  if (dtype == PCRL_BF16) hipLaunchKernelGGL((coltile_sum_kernel<bf16, true>), grid, dim3(256), lds, stream, (const bf16*)v, (float*)ws, M, C, coltile_rows(1, M), rowscale);
  else hipLaunchKernelGGL((coltile_sum_kernel<float, true>), grid, dim3(256), lds, stream, (const float*)v, (float*)ws, M, C, coltile_rows(1, M), rowscale);
  if (int e = pcrl_check_launch("weighted_colsum")) return e;
  hipLaunchKernelGGL(coltile_finish_kernel, dim3((C + 31) / 32, 1), dim3(256), 0, stream, (const float*)ws, out, (int)tiles, C, 1, 1.0);
  return pcrl_check_launch("weighted_colsum");
}

extern "C" size_t pcrl_gap_ws_bytes(int N, int64_t S, int C) { return (size_t)N * coltile_tiles(N, S) * C * sizeof(float); }

extern "C" int pcrl_gap_fwd(const void* a, float* g, void* ws, size_t ws_bytes, int N, int64_t S, int C, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(a && g && N > 0 && S > 0, "gap_fwd: bad arguments");
  return coltile_launch(a, g, ws, ws_bytes, N, S, C, dtype, 1.0 / (double)S, as_stream(stream), "gap_fwd");
}

// a = act(scale*y + shift) and g[n][c] = mean over the sample's voxels of a (the rounded values), one pass over y.  ws: pcrl_gap_ws_bytes.
extern "C" int pcrl_bn_act_apply_gap(const void* y, void* a, float* g, const float* scale, const float* shift, void* ws, size_t ws_bytes,
                                     int N, int64_t S, int C, int act, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(y && a && g && scale && shift && N > 0 && S > 0, "bn_act_apply_gap: bad arguments");
  PCRL_REQUIRE(rowadd_ok(C, dtype), "bn_act_apply_gap: not available for C=%d (pcrl_bn_act_bwd_rowadd_ok)", C);
  const int64_t tiles = coltile_tiles(N, S);
  const size_t need = (size_t)N * tiles * C * sizeof(float);
  if (!ws || ws_bytes < need) return pcrl_fail(PCRL_EWORKSPACE, "bn_act_apply_gap: workspace %zu < %zu", ws_bytes, need);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const size_t lds = (size_t)(256 / (C / vec)) * C * sizeof(float);
  const dim3 grid((unsigned)tiles, (unsigned)N);
  const bool nt = pcrl_streaming((int64_t)N * S * C * (dtype == PCRL_BF16 ? 2 : 4));
  if (dtype == PCRL_BF16) {
    using T = bf16;
    DISPATCH_ACT_T(bn_apply_gap_kernel, grid, lds, (const T*)y, (T*)a, scale, shift, (float*)ws, S, C, coltile_rows(N, S), nt);
  } else {
    using T = float;
    DISPATCH_ACT_T(bn_apply_gap_kernel, grid, lds, (const T*)y, (T*)a, scale, shift, (float*)ws, S, C, coltile_rows(N, S), nt);
  }
  if (int e = pcrl_check_launch("bn_act_apply_gap")) return e;
  hipLaunchKernelGGL(coltile_finish_kernel, dim3((C + 31) / 32, N), dim3(256), 0, as_stream(stream), (const float*)ws, g, (int)tiles, C, N, 1.0 / (double)S);
  return pcrl_check_launch("bn_act_apply_gap");
}

extern "C" int pcrl_gap_bwd(const float* dg, const void* add_src, void* da, int N, int64_t S, int C, int dtype, pcrl_stream_t stream) {
  if (int e = check_vec("gap_bwd", C, dtype, false)) return e;
  PCRL_REQUIRE(dg && da && N > 0 && S > 0, "gap_bwd: bad arguments");
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const int64_t nvt = (int64_t)N * S * (C / vec);
  const float inv = (float)(1.0 / (double)S);
  const bool nt = pcrl_streaming(nvt * 16);
#define GAP_BWD(T_, NT_) hipLaunchKernelGGL((gap_bwd_kernel<T_, NT_>), dim3(grid_for(nvt)), dim3(256), 0, as_stream(stream), dg, (const T_*)add_src, (T_*)da, S, C, nvt, inv)
  if (dtype == PCRL_BF16) { if (nt) GAP_BWD(bf16, true); else GAP_BWD(bf16, false); }
  else { if (nt) GAP_BWD(float, true); else GAP_BWD(float, false); }
#undef GAP_BWD
  return pcrl_check_launch("gap_bwd");
}
