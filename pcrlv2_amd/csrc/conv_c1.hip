// HBM/L2-bound convolutions with ONE input or ONE output channel (no MFMA: K or N of the GEMM is 1).
//
//   c1_fwd / c1_wgrad : first layer, Conv3d(1 -> Co, 3x3x3)   models/pcrlv2_model_3d.py:101 (down_tr64.ops.0)
//   to1_fwd / to1_dgrad / to1_wgrad : Conv3d(C -> 1), 27 taps (deep_supervision_head.conv1, :60,71) or
//                                     1 tap (OutputTransition.final_conv, :78)
// Scalar fields (x of the first layer, y/dy of the heads) are float32 [M]; C-channel tensors are NDHWC
// in the activation dtype, accessed as 16-byte channel vectors.
#include "common.h"

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
    if (tid < Co) {
      float s1 = 0.f, s2 = 0.f;
      for (int r = 0; r < 128; ++r) {
        const float v = red[r * ld + tid];
        s1 += v;
        s2 += v * v;
      }
      stats[((int64_t)blockIdx.x * Co + tid) * 2 + 0] = s1;
      stats[((int64_t)blockIdx.x * Co + tid) * 2 + 1] = s2;
    }
  }
}

// First layer weight gradient: dw[c][t] = sum_m dy[m][c] * x[m+delta_t].  Thread = (channel, tap group of 8);
// block = 2048 voxels; partials ws[block][Co*27] reduced in fixed order.
constexpr int C1W_VOX = 2048;
template <typename T>
__global__ void __launch_bounds__(256) c1_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                       float* __restrict__ ws, Dims g, int64_t M, int Co) {
  const int tid = threadIdx.x;
  const int c = tid % Co, tg = tid / Co;
  const int ngroups = 256 / Co;  // Co in {16,32,64} -> 16, 8, 4 tap groups
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const int64_t mbeg = (int64_t)blockIdx.x * C1W_VOX;
  const int64_t mend = (mbeg + C1W_VOX < M) ? mbeg + C1W_VOX : M;
  for (int64_t m = mbeg; m < mend; ++m) {
    int n, d, h, w;
    decode_voxel(m, g, n, d, h, w);
    const float dv = to_f(dy[m * Co + c]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = tg + j * ngroups;
      if (t < 27) {
        const int kd = t / 9 - 1, kh = (t / 3) % 3 - 1, kw = t % 3 - 1;
        if (tap_ok(d, h, w, kd, kh, kw, g)) acc[j] += dv * x[m + ((int64_t)kd * g.H + kh) * g.W + kw];
      }
    }
  }
  float* o = ws + (int64_t)blockIdx.x * Co * 27;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int t = tg + j * ngroups;
    if (t < 27) o[c * 27 + t] = acc[j];
  }
}

// out[i] = sum_r ws[r*stride + i], i < n: fixed-order fp64 accumulation.
__global__ void __launch_bounds__(256) rows_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int rows, int n,
                                                          int64_t stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int r = 0; r < rows; ++r) s += (double)ws[(int64_t)r * stride + i];
  out[i] = (float)s;
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

// C -> 1 weight gradient: dw[c][t] = sum_m dy[m] * x[m+delta_t][c];  db = sum_m dy[m].
// Thread = (channel vector, tap group); block = 1024 voxels; partials ws[block][taps*C + 1].
template <typename T>
__global__ void __launch_bounds__(256) to1_wgrad_kernel(const T* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ ws, Dims g, int64_t M, int C, int taps) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int TPT = 8;  // max taps per thread
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int nvec = C / VEC;         // <= 64 vectors handled per pass of the block
  const int cvs = nvec < 256 ? nvec : 256;
  const int ngroups = 256 / cvs;    // tap groups
  const int cv0 = tid % cvs, tg = tid / cvs;
  const int64_t mbeg = (int64_t)blockIdx.x * TO1_VOX;
  const int64_t mend = (mbeg + TO1_VOX < M) ? mbeg + TO1_VOX : M;
  float* o = ws + (int64_t)blockIdx.x * ((int64_t)taps * C + 1);
  for (int cv = cv0; cv < nvec; cv += cvs) {
    float acc[TPT][VEC];
#pragma unroll
    for (int a = 0; a < TPT; ++a)
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[a][j] = 0.f;
    if (tg < ngroups) {
      for (int64_t m = mbeg; m < mend; ++m) {
        const float dv = dy[m];
        int n, d, h, w;
        decode_voxel(m, g, n, d, h, w);
#pragma unroll
        for (int a = 0; a < TPT; ++a) {
          const int t = tg + a * ngroups;
          if (t < taps) {
            int64_t src = m;
            bool ok = true;
            if (taps == 27) {
              const int kd = t / 9 - 1, kh = (t / 3) % 3 - 1, kw = t % 3 - 1;
              ok = tap_ok(d, h, w, kd, kh, kw, g);
              src = m + ((int64_t)kd * g.H + kh) * g.W + kw;
            }
            if (ok) {
              const Vec16<T> xv = ld16(x + src * C + cv * VEC);
#pragma unroll
              for (int j = 0; j < VEC; ++j) acc[a][j] += dv * to_f(xv.v[j]);
            }
          }
        }
      }
#pragma unroll
      for (int a = 0; a < TPT; ++a) {
        const int t = tg + a * ngroups;
        if (t < taps) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[(int64_t)(cv * VEC + j) * taps + t] = acc[a][j];
        }
      }
    }
  }
  float s = 0.f;
  for (int64_t m = mbeg + tid; m < mend; m += 256) s += dy[m];
  s = block_sum_256(s, red);
  if (tid == 0) o[(int64_t)taps * C] = s;
}

int pow2_floor(int v) {
  int p = 1;
  while (p * 2 <= v) p *= 2;
  return p;
}

}  // namespace

extern "C" int pcrl_conv3d_k3_c1_fwd(const float* x, const float* w_ref, const float* bias, void* y, float* stats_partial,
                                     int N, int D, int H, int W, int Co, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && w_ref && y, "conv3d_k3_c1_fwd: null pointer");
  PCRL_REQUIRE(Co == 16 || Co == 32 || Co == 64, "conv3d_k3_c1_fwd: Co must be 16, 32 or 64 (got %d)", Co);
  const Dims g{N, D, H, W};
  const int64_t M = (int64_t)N * D * H * W;
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

extern "C" size_t pcrl_conv3d_k3_c1_wgrad_ws_bytes(int N, int D, int H, int W, int Co) {
  const int64_t M = (int64_t)N * D * H * W;
  return (size_t)((M + C1W_VOX - 1) / C1W_VOX) * Co * 27 * sizeof(float);
}

extern "C" int pcrl_conv3d_k3_c1_wgrad(const float* x, const void* dy, float* dw_ref, void* ws, size_t ws_bytes,
                                       int N, int D, int H, int W, int Co, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && dy && dw_ref, "conv3d_k3_c1_wgrad: null pointer");
  PCRL_REQUIRE(Co == 16 || Co == 32 || Co == 64, "conv3d_k3_c1_wgrad: Co must be 16, 32 or 64 (got %d)", Co);
  const size_t need = pcrl_conv3d_k3_c1_wgrad_ws_bytes(N, D, H, W, Co);
  if (!ws || ws_bytes < need) return pcrl_fail(PCRL_EWORKSPACE, "conv3d_k3_c1_wgrad: workspace %zu < %zu", ws_bytes, need);
  const Dims g{N, D, H, W};
  const int64_t M = (int64_t)N * D * H * W;
  const unsigned blocks = (unsigned)((M + C1W_VOX - 1) / C1W_VOX);
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(c1_wgrad_kernel<bf16>, dim3(blocks), dim3(256), 0, as_stream(stream), x, (const bf16*)dy, (float*)ws, g, M, Co);
  else if (dtype == PCRL_F32)
    hipLaunchKernelGGL(c1_wgrad_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), x, (const float*)dy, (float*)ws, g, M, Co);
  else
    return pcrl_fail(PCRL_EINVAL, "conv3d_k3_c1_wgrad: bad dtype %d", dtype);
  if (int e = pcrl_check_launch("c1_wgrad")) return e;
  const int n = Co * 27;
  hipLaunchKernelGGL(rows_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), (const float*)ws, dw_ref, (int)blocks, n, (int64_t)n);
  return pcrl_check_launch("c1_wgrad_reduce");
}

static int to1_check(const char* what, int C, int taps, int dtype) {
  if (taps != 27 && taps != 1) return pcrl_fail(PCRL_EINVAL, "%s: taps must be 27 or 1 (got %d)", what, taps);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  if (dtype != PCRL_BF16 && dtype != PCRL_F32) return pcrl_fail(PCRL_EINVAL, "%s: bad dtype %d", what, dtype);
  if (C <= 0 || C % vec != 0 || C > 1024) return pcrl_fail(PCRL_EINVAL, "%s: C=%d must be a multiple of %d, <= 1024", what, C, vec);
  return 0;
}

extern "C" int pcrl_conv3d_to1_fwd(const void* x, const float* w_ref, const float* bias, float* y, float* stats_partial,
                                   int N, int D, int H, int W, int C, int taps, int dtype, pcrl_stream_t stream) {
  if (int e = to1_check("conv3d_to1_fwd", C, taps, dtype)) return e;
  PCRL_REQUIRE(x && w_ref && y, "conv3d_to1_fwd: null pointer");
  const Dims g{N, D, H, W};
  const int64_t M = (int64_t)N * D * H * W;
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

extern "C" size_t pcrl_conv3d_to1_wgrad_ws_bytes(int N, int D, int H, int W, int C, int taps) {
  const int64_t M = (int64_t)N * D * H * W;
  return (size_t)((M + TO1_VOX - 1) / TO1_VOX) * ((size_t)taps * C + 1) * sizeof(float);
}

extern "C" int pcrl_conv3d_to1_wgrad(const void* x, const float* dy, float* dw_ref, float* db, void* ws, size_t ws_bytes,
                                     int N, int D, int H, int W, int C, int taps, int dtype, pcrl_stream_t stream) {
  if (int e = to1_check("conv3d_to1_wgrad", C, taps, dtype)) return e;
  PCRL_REQUIRE(x && dy && dw_ref && db, "conv3d_to1_wgrad: null pointer");
  const size_t need = pcrl_conv3d_to1_wgrad_ws_bytes(N, D, H, W, C, taps);
  if (!ws || ws_bytes < need) return pcrl_fail(PCRL_EWORKSPACE, "conv3d_to1_wgrad: workspace %zu < %zu", ws_bytes, need);
  const Dims g{N, D, H, W};
  const int64_t M = (int64_t)N * D * H * W;
  const unsigned blocks = (unsigned)((M + TO1_VOX - 1) / TO1_VOX);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const int nvec = C / vec;
  const int cvs = nvec < 256 ? nvec : 256;
  if (256 % cvs != 0 || (taps + (256 / cvs) - 1) / (256 / cvs) > 8)
    return pcrl_fail(PCRL_EINVAL, "conv3d_to1_wgrad: unsupported C=%d (vectors per voxel %d)", C, nvec);
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(to1_wgrad_kernel<bf16>, dim3(blocks), dim3(256), 0, as_stream(stream), (const bf16*)x, dy, (float*)ws, g, M, C, taps);
  else
    hipLaunchKernelGGL(to1_wgrad_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)x, dy, (float*)ws, g, M, C, taps);
  if (int e = pcrl_check_launch("to1_wgrad")) return e;
  // partial row = [taps*C] weight sums followed by 1 bias sum
  const int n = taps * C;
  hipLaunchKernelGGL(rows_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), (const float*)ws, dw_ref, (int)blocks, n, (int64_t)n + 1);
  if (int e = pcrl_check_launch("to1_wgrad_reduce")) return e;
  hipLaunchKernelGGL(rows_reduce_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const float*)ws + n, db, (int)blocks, 1, (int64_t)n + 1);
  return pcrl_check_launch("to1_wgrad_reduce_b");
}
