// Shared device/host helpers for libpcrl_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

// Second argument of __launch_bounds__ for the kernels whose default allocation (no bound: the compiler assumes one wave per SIMD is acceptable)
// came out at 150-218 VGPRs PLUS 40-144 AGPRs, i.e. ONE 256-thread block per CU: 2 = at most 256 registers per lane in all, two waves per SIMD
// (found by rocprofv3 counters on the 2D stride-2 gather convolution: 0.23 waves per SIMD on average, 203 us for 38.7 GFLOP).  -DPCRL_OCC2=1
// rebuilds the old allocation for A/B runs (tools/build_variant.sh).
#ifndef PCRL_OCC2
#define PCRL_OCC2 2
#endif

#include "../../include/pcrl_hip.h"

typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // one 16-byte staging register

// By-VALUE zero-select of a staged 16-byte vector.  (`cond ? arr[i] : zero` on lvalues is a select between ADDRESSES and
// forces the staging array into scratch memory.)
__device__ __forceinline__ u32x4 keep_if(bool ok, u32x4 v) {
  const unsigned m = ok ? 0xFFFFFFFFu : 0u;
  return u32x4{v.x & m, v.y & m, v.z & m, v.w & m};
}

// ---- error reporting (thread-local, never throws across the ABI) ----------------------
int pcrl_fail(int code, const char* fmt, ...);
int pcrl_check_launch(const char* what);

#define PCRL_REQUIRE(cond, ...)                                  \
  do {                                                           \
    if (!(cond)) return pcrl_fail(PCRL_EINVAL, __VA_ARGS__);     \
  } while (0)

static inline hipStream_t as_stream(pcrl_stream_t s) { return (hipStream_t)s; }

// ---- element type helpers ---------------------------------------------------------------
__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }

// A 16-byte vector of T (4 floats or 8 bf16): the unit of every coalesced activation access.
template <typename T> struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
  T v[N];
};
template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
  union { uint4 u; Vec16<T> v; } x;
  x.u = *reinterpret_cast<const uint4*>(p);
  return x.v;
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& v) {
  union { uint4 u; Vec16<T> v; } x;
  x.v = v;
  *reinterpret_cast<uint4*>(p) = x.u;
}
// streaming variants: non-temporal hint for tensors far larger than L2 + MALL (measured on the BatchNorm kernels: +6..9 % on
// 0.5-1 GB tensors, -10 % on 64 MB ones that the neighbouring kernels still find in cache) -- selected by size (compile-time copies of the loop behind one uniform branch)
template <typename T> __device__ __forceinline__ Vec16<T> ld16_nt(const T* p) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  union { u32x4_t u; Vec16<T> v; } x;
  x.u = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
  return x.v;
}
template <bool NT, typename T> __device__ __forceinline__ Vec16<T> ld16_sel(const T* p) { return NT ? ld16_nt(p) : ld16(p); }
template <typename T> __device__ __forceinline__ void st16_nt(T* p, const Vec16<T>& v) {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  union { u32x4_t u; Vec16<T> v; } x;
  x.v = v;
  __builtin_nontemporal_store(x.u, reinterpret_cast<u32x4_t*>(p));
}
template <bool NT, typename T> __device__ __forceinline__ void st16_sel(T* p, const Vec16<T>& v) {
  if (NT) st16_nt(p, v);
  else st16(p, v);
}
inline bool pcrl_streaming(int64_t bytes) {
  return bytes >= ((int64_t)192 << 20);   // non-temporal accesses from 192 MB per tensor on (tools/bn_probe.py, round 3)
}

// ---- wave / block reductions (wave = 64) ------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over a 256-thread block; result valid in thread 0.  `red` = 4 values of LDS scratch.
template <typename A> __device__ __forceinline__ A block_sum_256(A v, A* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---- voxel index helpers ----------------------------------------------------------------
struct Dims {
  int N, D, H, W;
};
__host__ __device__ __forceinline__ void decode_voxel(int64_t m, const Dims& g, int& n, int& d, int& h, int& w) {
  w = (int)(m % g.W);
  int64_t t = m / g.W;
  h = (int)(t % g.H);
  t /= g.H;
  d = (int)(t % g.D);
  n = (int)(t / g.D);
}
// 27-bit mask: bit t = kd*9+kh*3+kw set iff neighbour (d+kd-1, h+kh-1, w+kw-1) is inside the volume.
__host__ __device__ __forceinline__ uint32_t tap_mask27(int d, int h, int w, const Dims& g) {
  uint32_t md = (d > 0 ? 1u : 0u) | 2u | (d < g.D - 1 ? 4u : 0u);
  uint32_t mh = (h > 0 ? 1u : 0u) | 2u | (h < g.H - 1 ? 4u : 0u);
  uint32_t mw = (w > 0 ? 1u : 0u) | 2u | (w < g.W - 1 ? 4u : 0u);
  uint32_t m = 0;
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
      if (((md >> kd) & 1u) && ((mh >> kh) & 1u)) m |= mw << (kd * 9 + kh * 3);
  return m;
}
// Row offset (in voxels) of tap t of a 3x3x3 stencil.
__host__ __device__ __forceinline__ int64_t tap_delta27(int t, const Dims& g) {
  int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
  return ((int64_t)(kd - 1) * g.H + (kh - 1)) * g.W + (kw - 1);
}
// Row in the 2x-upsampled volume [N][2D][2H][2W] of input voxel (n,d,h,w), tap t = i*4+j*2+k.
__host__ __device__ __forceinline__ int64_t up2_row(int n, int d, int h, int w, int t, const Dims& g) {
  int i = t >> 2, j = (t >> 1) & 1, k = t & 1;
  return (((int64_t)n * (2 * g.D) + (2 * d + i)) * (2 * g.H) + (2 * h + j)) * (2 * g.W) + (2 * w + k);
}
