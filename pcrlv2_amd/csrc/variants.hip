// Constructor variants of LUConv that the reference accepts but train_3d.py:45 never instantiates (models/pcrlv2_model_3d.py:20-30):
// nn.PReLU(out_chan) as a separate streaming pass behind the normalisation (which then runs with PCRL_ACT_NONE).  ELU lives in
// norm_pool.hip as an activation code of the fused kernels; InstanceNorm3d is GroupNorm with one channel per group (extras_groupnorm.hip).
// Not on the pre-training hot path: plain coalesced 16-byte streaming, two-stage deterministic reduction for the slope gradient.
#include "common.h"

namespace {

constexpr int PRELU_TILE_ROWS = 256;

// a[m][c] = z > 0 ? z : w[c] * z       -- aten::prelu, models/pcrlv2_model_3d.py:23,33
template <typename T>
__global__ void __launch_bounds__(256) prelu_fwd_kernel(const T* __restrict__ z, const float* __restrict__ w, T* __restrict__ a, int64_t nvec_total, int C) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec_total; i += (int64_t)gridDim.x * 256) {
    const int c0 = (int)(i % nvec) * VEC;
    const Vec16<T> v = ld16(z + i * VEC);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float x = to_f(v.v[j]);
      o.v[j] = from_f<T>(x > 0.f ? x : w[c0 + j] * x);
    }
    st16(a + i * VEC, o);
  }
}

// dz = da * (z > 0 ? 1 : w[c]);  partial[tile][c] = sum over the tile's rows of da * z * [z <= 0]     -- aten::prelu_backward
// thread = (channel vector, row slot) like the BatchNorm reductions; nvec = C / VEC divides 256
template <typename T>
__global__ void __launch_bounds__(256) prelu_bwd_kernel(const T* __restrict__ da, const T* __restrict__ z, const float* __restrict__ w,
                                                        T* __restrict__ dz, float* __restrict__ partial, int64_t M, int C) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [slots][C]
  const int nvec = C / VEC, cv = threadIdx.x % nvec, slot = threadIdx.x / nvec, nslots = 256 / nvec;
  const int64_t rbeg = (int64_t)blockIdx.x * PRELU_TILE_ROWS;
  const int64_t rend = rbeg + PRELU_TILE_ROWS < M ? rbeg + PRELU_TILE_ROWS : M;
  float sw[VEC], acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sw[j] = w[cv * VEC + j];
    acc[j] = 0.f;
  }
  for (int64_t r = rbeg + slot; r < rend; r += nslots) {
    const int64_t off = (r * nvec + cv) * VEC;
    const Vec16<T> g = ld16(da + off), v = ld16(z + off);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float x = to_f(v.v[j]), d = to_f(g.v[j]);
      o.v[j] = from_f<T>(x > 0.f ? d : sw[j] * d);
      acc[j] += x > 0.f ? 0.f : d * x;
    }
    st16(dz + off, o);
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) sm[slot * C + cv * VEC + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f;
    for (int q = 0; q < nslots; ++q) a += sm[q * C + c];
    partial[(int64_t)blockIdx.x * C + c] = a;
  }
}

int prelu_check(const char* what, int64_t M, int C, int dtype) {
  if (dtype != PCRL_F32 && dtype != PCRL_BF16) return pcrl_fail(PCRL_EINVAL, "%s: bad dtype %d", what, dtype);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  if (M <= 0 || C <= 0 || C % vec != 0 || C / vec > 256 || 256 % (C / vec) != 0)
    return pcrl_fail(PCRL_EINVAL, "%s: M=%lld C=%d: channel vectors of %d must divide 256", what, (long long)M, C, vec);
  return 0;
}

}  // namespace

extern "C" int pcrl_prelu_fwd(const void* z, const float* w, void* a, int64_t M, int C, int dtype, pcrl_stream_t stream) {
  if (int e = prelu_check("prelu_fwd", M, C, dtype)) return e;
  PCRL_REQUIRE(z && w && a, "prelu_fwd: null pointer");
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const int64_t nv = M * C / vec;
  const unsigned grid = (unsigned)((nv + 255) / 256 < 4096 ? (nv + 255) / 256 : 4096);
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(prelu_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16*)z, w, (bf16*)a, nv, C);
  else hipLaunchKernelGGL(prelu_fwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)z, w, (float*)a, nv, C);
  return pcrl_check_launch("prelu_fwd");
}

extern "C" int64_t pcrl_prelu_bwd_partial_rows(int64_t M) { return (M + PRELU_TILE_ROWS - 1) / PRELU_TILE_ROWS; }

extern "C" int pcrl_prelu_bwd(const void* da, const void* z, const float* w, void* dz, float* partial, int64_t M, int C, int dtype, pcrl_stream_t stream) {
  if (int e = prelu_check("prelu_bwd", M, C, dtype)) return e;
  PCRL_REQUIRE(da && z && w && dz && partial, "prelu_bwd: null pointer");
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const dim3 grid((unsigned)pcrl_prelu_bwd_partial_rows(M));
  const size_t lds = (size_t)(256 / (C / vec)) * C * sizeof(float);
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(prelu_bwd_kernel<bf16>, grid, dim3(256), lds, as_stream(stream), (const bf16*)da, (const bf16*)z, w, (bf16*)dz, partial, M, C);
  else
    hipLaunchKernelGGL(prelu_bwd_kernel<float>, grid, dim3(256), lds, as_stream(stream), (const float*)da, (const float*)z, w, (float*)dz, partial, M, C);
  return pcrl_check_launch("prelu_bwd");
}
