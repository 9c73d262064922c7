// OPTIONAL EXTRA -- GroupNorm(G) statistics / coefficient kernels ("GroupNorm+SiLU" is named in north_star but
// the reference only ever instantiates BatchNorm3d + ReLU; its own norm='gn' path crashes at construction).  Nothing on the
// reference-parity path uses this file.
//
// GroupNorm is the BatchNorm machinery applied per SAMPLE with statistics pooled over a channel group: the streaming kernels of
// norm_pool.hip (apply, backward reduce, backward apply; activation templated, PCRL_ACT_SILU added for this) are launched on one
// sample at a time with per-(sample, channel) coefficient rows produced here.
//   forward :  (sum, sum^2) per (sample, tile, channel)  ->  per (sample, group) mean, rstd  ->  scale, shift  [N][C]
//   backward:  per (sample, channel) A = sum dz, B = sum dz * xhat (bn_bwd_reduce_kernel with the sample's mean_c / rstd_c rows)
//              dx = r g_c dz - r m1 - r xhat m2,  m1 = sum_{c in group} g_c A_c / cnt,  m2 = sum g_c B_c / cnt
//              written as dx = k1 dz + kB y + kA  with  k1 = r g_c,  kB = -r^2 m2,  kA = r^2 m2 mu - r m1
#include "common.h"

namespace {

constexpr int GN_TILE_ROWS = 512;

// partial[((n * tiles + tile) * C + c) * 2 + {0,1}] = (sum, sum^2) of y[n][rows of the tile][c]; thread = channel vector x row slot
template <typename T>
__global__ void __launch_bounds__(256) gn_stats_kernel(const T* __restrict__ y, float* __restrict__ partial, int64_t S, int C) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [slots][C][2]
  const int tid = threadIdx.x, nvec = C / VEC;
  const int cv = tid % nvec, slot = tid / nvec, nslots = 256 / nvec;
  const T* base = y + (int64_t)blockIdx.y * S * C;
  const int64_t rbeg = (int64_t)blockIdx.x * GN_TILE_ROWS;
  const int64_t rend = (rbeg + GN_TILE_ROWS < S) ? rbeg + GN_TILE_ROWS : S;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.f;
  for (int64_t r = rbeg + slot; r < rend; r += nslots) {
    const Vec16<T> x = ld16(base + (r * nvec + cv) * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float v = to_f(x.v[j]);
      s1[j] += v;
      s2[j] += v * v;
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
    for (int q = 0; q < nslots; ++q) {
      a += sm[(q * C + c) * 2 + 0];
      b += sm[(q * C + c) * 2 + 1];
    }
    float* o = partial + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * C + c) * 2;
    o[0] = a;
    o[1] = b;
  }
}

// one block per sample; C <= 1024
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ partial, int tiles, int64_t S, int C, int G,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float* __restrict__ mean_c, float* __restrict__ rstd_c, float* __restrict__ scale,
                                                          float* __restrict__ shift) {
  __shared__ double cs[1024][2];
  __shared__ float gm[1024], gr[1024];
  const int n = blockIdx.x, cpg = C / G;
  for (int c = threadIdx.x; c < C; c += 256) {
    double a = 0.0, b = 0.0;
    for (int t = 0; t < tiles; ++t) {
      const float* p = partial + (((int64_t)n * tiles + t) * C + c) * 2;
      a += (double)p[0];
      b += (double)p[1];
    }
    cs[c][0] = a;
    cs[c][1] = b;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 256) {   // G == C: InstanceNorm3d (models/pcrlv2_model_3d.py:15-16)
    double a = 0.0, b = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      a += cs[c][0];
      b += cs[c][1];
    }
    const double cnt = (double)S * cpg, mu = a / cnt;
    double var = b / cnt - mu * mu;   // biased, as torch.nn.GroupNorm / InstanceNorm3d
    if (var < 0.0) var = 0.0;
    gm[g] = (float)mu;
    gr[g] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float sc = gamma[c] * gr[g];
    mean_c[(int64_t)n * C + c] = gm[g];
    rstd_c[(int64_t)n * C + c] = gr[g];
    scale[(int64_t)n * C + c] = sc;
    shift[(int64_t)n * C + c] = beta[c] - gm[g] * sc;
  }
}

__global__ void __launch_bounds__(256) gn_bwd_finalize_kernel(const float* __restrict__ partial_b, int rows_b, int64_t S, int C, int G,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean_c,
                                                              const float* __restrict__ rstd_c, float* __restrict__ k1, float* __restrict__ kB,
                                                              float* __restrict__ kA, float* __restrict__ dgamma_n, float* __restrict__ dbeta_n) {
  __shared__ double cs[1024][2];
  __shared__ float m1[1024], m2[1024];
  const int n = blockIdx.x, cpg = C / G;
  for (int c = threadIdx.x; c < C; c += 256) {
    double a = 0.0, b = 0.0;
    for (int t = 0; t < rows_b; ++t) {
      const float* p = partial_b + (((int64_t)n * rows_b + t) * C + c) * 2;
      a += (double)p[0];
      b += (double)p[1];
    }
    cs[c][0] = a;   // sum dz
    cs[c][1] = b;   // sum dz * xhat
    dbeta_n[(int64_t)n * C + c] = (float)a;
    dgamma_n[(int64_t)n * C + c] = (float)b;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 256) {
    double a = 0.0, b = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      a += (double)gamma[c] * cs[c][0];
      b += (double)gamma[c] * cs[c][1];
    }
    const double cnt = (double)S * cpg;
    m1[g] = (float)(a / cnt);
    m2[g] = (float)(b / cnt);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    const float r = rstd_c[(int64_t)n * C + c], mu = mean_c[(int64_t)n * C + c];
    k1[(int64_t)n * C + c] = r * gamma[c];
    kB[(int64_t)n * C + c] = -r * r * m2[g];
    kA[(int64_t)n * C + c] = r * r * m2[g] * mu - r * m1[g];
  }
}

int gn_check(const char* what, int N, int64_t S, int C, int G) {
  if (N <= 0 || S <= 0 || C <= 0 || C > 1024 || G <= 0 || C % G != 0)
    return pcrl_fail(PCRL_EINVAL, "%s: bad sizes N=%d S=%lld C=%d G=%d (C <= 1024, G | C)", what, N, (long long)S, C, G);
  return 0;
}

}  // namespace

extern "C" int64_t pcrl_gn_stats_tiles(int64_t S) { return (S + GN_TILE_ROWS - 1) / GN_TILE_ROWS; }

extern "C" int pcrl_gn_stats(const void* y, float* partial, int N, int64_t S, int C, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(y && partial && N > 0 && S > 0, "gn_stats: bad arguments");
  PCRL_REQUIRE(dtype == PCRL_F32 || dtype == PCRL_BF16, "gn_stats: bad dtype %d", dtype);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  PCRL_REQUIRE(C > 0 && C % vec == 0 && (C / vec) <= 256 && 256 % (C / vec) == 0, "gn_stats: C=%d: channel vectors must divide 256", C);
  const dim3 grid((unsigned)pcrl_gn_stats_tiles(S), (unsigned)N);
  const size_t lds = (size_t)(256 / (C / vec)) * C * 2 * sizeof(float);
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(gn_stats_kernel<bf16>, grid, dim3(256), lds, as_stream(stream), (const bf16*)y, partial, S, C);
  else hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), lds, as_stream(stream), (const float*)y, partial, S, C);
  return pcrl_check_launch("gn_stats");
}

extern "C" int pcrl_gn_finalize(const float* partial, int tiles, int N, int64_t S, int C, int G, const float* gamma, const float* beta, float eps,
                                float* mean_c, float* rstd_c, float* scale, float* shift, pcrl_stream_t stream) {
  PCRL_REQUIRE(partial && gamma && beta && mean_c && rstd_c && scale && shift && tiles > 0, "gn_finalize: bad arguments");
  if (int e = gn_check("gn_finalize", N, S, C, G)) return e;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(N), dim3(256), 0, as_stream(stream), partial, tiles, S, C, G, gamma, beta, eps, mean_c, rstd_c, scale, shift);
  return pcrl_check_launch("gn_finalize");
}

extern "C" int pcrl_gn_bwd_finalize(const float* partial_b, int rows_b, int N, int64_t S, int C, int G, const float* gamma, const float* mean_c,
                                    const float* rstd_c, float* k1, float* kB, float* kA, float* dgamma_n, float* dbeta_n, pcrl_stream_t stream) {
  PCRL_REQUIRE(partial_b && gamma && mean_c && rstd_c && k1 && kB && kA && dgamma_n && dbeta_n && rows_b > 0, "gn_bwd_finalize: bad arguments");
  if (int e = gn_check("gn_bwd_finalize", N, S, C, G)) return e;
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(N), dim3(256), 0, as_stream(stream), partial_b, rows_b, S, C, G, gamma, mean_c, rstd_c, k1, kB, kA,
                     dgamma_n, dbeta_n);
  return pcrl_check_launch("gn_bwd_finalize");
}
