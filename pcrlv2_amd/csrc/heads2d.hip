// The 3-channel ends of the 2D PCRLv2 step for gfx950: restoration / deep-supervision MSE against the image and the
// backward of the 1x1 convolution to 3 channels -- all HBM-bound single passes.
//
//   nn.MSELoss()(masks, gt)                      train_2d.py:165,167   (masks: NHWC float32 prediction, gt: the loader's NCHW image)
//   its backward, written as the zero-padded `dtype` tensor the convolution backward kernels consume (the 3-channel gradient used to be
//   padded by two ATen launches, summed for the bias by two more and converted by a fifth)
//   deep_supervision_head[3] = Conv2d(C, 3, 1)   models/pcrlv2_model.py:106: backward (dx, dw, db) in ONE pass over x and dy
//
// Every reduction is two-stage with a fixed order (no atomics): a block leaves one partial row, pcrl_colsum finishes.
#include "common.h"

namespace {

constexpr int ROWS_PER_BLOCK = 1024;   // pixels per first-stage block

// ---- MSE between an NHWC prediction and an NCHW target --------------------------------------------------------------------------
// p: float32 [N][HW][C], gt: float32 [N][C][HW].  A thread owns one pixel: C contiguous floats of p, C plane reads of gt (coalesced
// across the threads of a wave).
template <int C>
__global__ void __launch_bounds__(256) mse2d_partial_kernel(const float* __restrict__ p, const float* __restrict__ gt, double* __restrict__ ws,
                                                            int64_t HW, int64_t M) {
  __shared__ double red[4];
  const int64_t beg = (int64_t)blockIdx.x * ROWS_PER_BLOCK;
  const int64_t end = (beg + ROWS_PER_BLOCK < M) ? beg + ROWS_PER_BLOCK : M;
  float s = 0.f;
  for (int64_t m = beg + threadIdx.x; m < end; m += 256) {
    const int64_t n = m / HW, q = m - n * HW;
    const float* pp = p + m * C;
    const float* gg = gt + n * C * HW + q;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float d = pp[c] - gg[(int64_t)c * HW];
      s += d * d;
    }
  }
  const double t = block_sum_256((double)s, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = t;
}
__global__ void __launch_bounds__(256) mse2d_finish_kernel(const double* __restrict__ ws, float* __restrict__ loss, int blocks, double inv_n) {
  __shared__ double red[4];
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  int i = threadIdx.x;
  for (; i + 768 < blocks; i += 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] += ws[i + 256 * u];
  }
  for (; i < blocks; i += 256) a[0] += ws[i];
  const double s = block_sum_256((a[0] + a[1]) + (a[2] + a[3]), red);
  if (threadIdx.x == 0) loss[0] = (float)(s * inv_n);
}

// dy[m][0..CP) = 2/n * dloss * (p - gt) in channels < C, zero in the padding; colpart[block][CP] = the block's column sums (the bias
// gradient of the convolution that produced p is their total).  T = output type (bf16: what the MFMA backward kernels read).
template <typename T, int C, int CP>
__global__ void __launch_bounds__(256) mse2d_bwd_pad_kernel(const float* __restrict__ p, const float* __restrict__ gt, const float* __restrict__ dloss,
                                                            T* __restrict__ dy, float* __restrict__ colpart, int64_t HW, int64_t M, float two_over_n) {
  __shared__ float sm[4][C];
  const float g = dloss[0] * two_over_n;
  const int64_t beg = (int64_t)blockIdx.x * ROWS_PER_BLOCK;
  const int64_t end = (beg + ROWS_PER_BLOCK < M) ? beg + ROWS_PER_BLOCK : M;
  float cs[C];
#pragma unroll
  for (int c = 0; c < C; ++c) cs[c] = 0.f;
  for (int64_t m = beg + threadIdx.x; m < end; m += 256) {
    const int64_t n = m / HW, q = m - n * HW;
    const float* pp = p + m * C;
    const float* gg = gt + n * C * HW + q;
    alignas(16) T o[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) o[c] = from_f<T>(0.f);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float d = g * (pp[c] - gg[(int64_t)c * HW]);
      o[c] = from_f<T>(d);
      cs[c] += to_f(o[c]);        // the sum of what the weight / data gradient kernels will read (the rounded values)
    }
    T* dst = dy + m * CP;
    if (CP * sizeof(T) % 16 == 0) {
#pragma unroll
      for (int v = 0; v < (int)(CP * sizeof(T) / 16); ++v) reinterpret_cast<uint4*>(dst)[v] = reinterpret_cast<const uint4*>(o)[v];
    } else {
#pragma unroll
      for (int c = 0; c < CP; ++c) dst[c] = o[c];
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float w = wave_sum(cs[c]);
    if (lane == 0) sm[wid][c] = w;
  }
  __syncthreads();
  if (threadIdx.x < CP) {
    const int c = threadIdx.x;
    colpart[(int64_t)blockIdx.x * CP + c] = c < C ? (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]) : 0.f;
  }
}

// ---- backward of a 1x1 convolution to CO <= 4 channels -------------------------------------------------------------------------
// x: T [M][Ci] (the layer's input), dy: float32 [M][CO], w: float32 [CO][Ci].
//   dx[m][ci] = sum_co dy[m][co] * w[co][ci]                (T [M][Ci])
//   part[block][co][ci] = sum over the block's pixels of dy[m][co] * x[m][ci];  part[block][CO*Ci + co] = sum dy[m][co]   (row pitch PW, zero padded)
// A thread owns ONE 16-byte channel vector (like the BatchNorm streaming kernels): its 8 (4) weights per output channel and its
// CO x VEC accumulators live in registers; 256 / nvec pixels per block iteration.
template <typename T, int CO>
__global__ void __launch_bounds__(256) conv1x1_small_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w,
                                                                T* __restrict__ dx, float* __restrict__ part, int64_t M, int Ci, int PW) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sm[];     // [nslots][CO][Ci] then reused
  const int nvec = Ci / VEC, cv = threadIdx.x % nvec, slot = threadIdx.x / nvec, nslots = 256 / nvec;
  float wr[CO][VEC], acc[CO][VEC], dsum[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    dsum[co] = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      wr[co][j] = w[co * Ci + cv * VEC + j];
      acc[co][j] = 0.f;
    }
  }
  const int64_t beg = (int64_t)blockIdx.x * ROWS_PER_BLOCK;
  const int64_t end = (beg + ROWS_PER_BLOCK < M) ? beg + ROWS_PER_BLOCK : M;
#pragma unroll 2
  for (int64_t m = beg + slot; m < end; m += nslots) {
    const Vec16<T> xv = ld16(x + (m * nvec + cv) * VEC);
    float g[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) g[co] = dy[m * CO + co];
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float s = 0.f;
      const float xf = to_f(xv.v[j]);
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        s += g[co] * wr[co][j];
        acc[co][j] += g[co] * xf;
      }
      o.v[j] = from_f<T>(s);
    }
    st16(dx + (m * nvec + cv) * VEC, o);
    if (cv == 0) {
#pragma unroll
      for (int co = 0; co < CO; ++co) dsum[co] += g[co];
    }
  }
  // combine the row slots (fixed order)
#pragma unroll
  for (int co = 0; co < CO; ++co)
#pragma unroll
    for (int j = 0; j < VEC; ++j) sm[(slot * CO + co) * Ci + cv * VEC + j] = acc[co][j];
  __syncthreads();
  float* out = part + (int64_t)blockIdx.x * PW;
  for (int e = CO * Ci + CO + threadIdx.x; e < PW; e += 256) out[e] = 0.f;   // row pitch padding (pcrl_colsum wants 4 * 2^k columns)
  for (int e = threadIdx.x; e < CO * Ci; e += 256) {
    float a = 0.f;
    for (int q = 0; q < nslots; ++q) a += sm[q * CO * Ci + e];
    out[e] = a;
  }
  __syncthreads();
  if (cv == 0) {
#pragma unroll
    for (int co = 0; co < CO; ++co) sm[slot * CO + co] = dsum[co];
  }
  __syncthreads();
  if (threadIdx.x < CO) {
    float a = 0.f;
    for (int q = 0; q < nslots; ++q) a += sm[q * CO + threadIdx.x];
    out[CO * Ci + threadIdx.x] = a;
  }
}

// out[n][q][0..CP) = (T)x[n][c][q] for c < C, zero above: the 3-channel image as the stem's gather reads it (one pass, coalesced plane reads)
template <typename T, int CP>
__global__ void __launch_bounds__(256) nchw_to_nhwc_pad_kernel(const float* __restrict__ x, T* __restrict__ out, int C, int64_t HW, int64_t M) {
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    const int64_t n = m / HW, q = m - n * HW;
    alignas(16) T o[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) o[c] = from_f<T>(c < C ? x[(n * C + c) * HW + q] : 0.f);
    T* dst = out + m * CP;
#pragma unroll
    for (int v = 0; v < (int)(CP * sizeof(T) / 16); ++v) reinterpret_cast<uint4*>(dst)[v] = reinterpret_cast<const uint4*>(o)[v];
  }
}

__global__ void __launch_bounds__(256) add_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] + b[i];
}

}  // namespace

extern "C" int64_t pcrl_rows1024(int64_t M) { return (M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK; }

extern "C" size_t pcrl_mse2d_ws_bytes(int64_t M) { return (size_t)pcrl_rows1024(M) * sizeof(double); }

extern "C" int pcrl_mse2d_fwd(const float* p, const float* gt, float* loss, void* ws, size_t ws_bytes, int N, int64_t HW, int C,
                              pcrl_stream_t stream) {
  PCRL_REQUIRE(p && gt && loss && N > 0 && HW > 0, "mse2d_fwd: bad arguments");
  PCRL_REQUIRE(C == 3 || C == 1, "mse2d_fwd: C=%d (1 or 3 channels)", C);
  const int64_t M = (int64_t)N * HW;
  if (!ws || ws_bytes < pcrl_mse2d_ws_bytes(M)) return pcrl_fail(PCRL_EWORKSPACE, "mse2d_fwd: workspace too small");
  const int blocks = (int)pcrl_rows1024(M);
  if (C == 3) hipLaunchKernelGGL((mse2d_partial_kernel<3>), dim3(blocks), dim3(256), 0, as_stream(stream), p, gt, (double*)ws, HW, M);
  else hipLaunchKernelGGL((mse2d_partial_kernel<1>), dim3(blocks), dim3(256), 0, as_stream(stream), p, gt, (double*)ws, HW, M);
  if (int e = pcrl_check_launch("mse2d_partial")) return e;
  hipLaunchKernelGGL(mse2d_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const double*)ws, loss, blocks, 1.0 / ((double)M * C));
  return pcrl_check_launch("mse2d_finish");
}

extern "C" int pcrl_mse2d_bwd_pad(const float* p, const float* gt, const float* dloss, void* dy, float* colpart, int N, int64_t HW, int C, int CP,
                                  int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(p && gt && dloss && dy && colpart && N > 0 && HW > 0, "mse2d_bwd_pad: bad arguments");
  PCRL_REQUIRE(C == 3 && (CP == 8 || CP == 3), "mse2d_bwd_pad: C=%d CP=%d (3 channels, padded to 8 or not at all)", C, CP);
  PCRL_REQUIRE(dtype == PCRL_F32 || dtype == PCRL_BF16, "mse2d_bwd_pad: bad dtype %d", dtype);
  const int64_t M = (int64_t)N * HW;
  const dim3 grid((unsigned)pcrl_rows1024(M));
  const float k = (float)(2.0 / ((double)M * C));
  if (dtype == PCRL_BF16 && CP == 8) hipLaunchKernelGGL((mse2d_bwd_pad_kernel<bf16, 3, 8>), grid, dim3(256), 0, as_stream(stream), p, gt, dloss, (bf16*)dy, colpart, HW, M, k);
  else if (dtype == PCRL_F32 && CP == 8) hipLaunchKernelGGL((mse2d_bwd_pad_kernel<float, 3, 8>), grid, dim3(256), 0, as_stream(stream), p, gt, dloss, (float*)dy, colpart, HW, M, k);
  else if (dtype == PCRL_F32 && CP == 3) hipLaunchKernelGGL((mse2d_bwd_pad_kernel<float, 3, 3>), grid, dim3(256), 0, as_stream(stream), p, gt, dloss, (float*)dy, colpart, HW, M, k);
  else return pcrl_fail(PCRL_EINVAL, "mse2d_bwd_pad: the unpadded form is float32 only");
  return pcrl_check_launch("mse2d_bwd_pad");
}

extern "C" int pcrl_conv2d_1x1_small_bwd(const void* x, const float* dy, const float* w, void* dx, float* part, int64_t M, int Ci, int Co, int PW,
                                         int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && dy && w && dx && part && M > 0, "conv2d_1x1_small_bwd: bad arguments");
  PCRL_REQUIRE(Co == 3, "conv2d_1x1_small_bwd: Co=%d (the deep-supervision map has 3 channels)", Co);
  PCRL_REQUIRE(PW >= Co * Ci + Co, "conv2d_1x1_small_bwd: row pitch %d < %d", PW, Co * Ci + Co);
  PCRL_REQUIRE(dtype == PCRL_F32 || dtype == PCRL_BF16, "conv2d_1x1_small_bwd: bad dtype %d", dtype);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  PCRL_REQUIRE(Ci % vec == 0 && Ci / vec <= 256 && 256 % (Ci / vec) == 0, "conv2d_1x1_small_bwd: Ci=%d", Ci);
  const dim3 grid((unsigned)pcrl_rows1024(M));
  const size_t lds = (size_t)(256 / (Ci / vec)) * 3 * Ci * sizeof(float);
  PCRL_REQUIRE(lds <= 64 * 1024, "conv2d_1x1_small_bwd: Ci=%d needs %zu bytes of LDS", Ci, lds);
  if (dtype == PCRL_BF16) hipLaunchKernelGGL((conv1x1_small_bwd_kernel<bf16, 3>), grid, dim3(256), lds, as_stream(stream), (const bf16*)x, dy, w, (bf16*)dx, part, M, Ci, PW);
  else hipLaunchKernelGGL((conv1x1_small_bwd_kernel<float, 3>), grid, dim3(256), lds, as_stream(stream), (const float*)x, dy, w, (float*)dx, part, M, Ci, PW);
  return pcrl_check_launch("conv2d_1x1_small_bwd");
}

extern "C" int pcrl_nchw_to_nhwc_pad(const float* x, void* out, int N, int C, int64_t HW, int CP, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && out && N > 0 && HW > 0 && C > 0 && C <= CP && CP == 8, "nchw_to_nhwc_pad: bad arguments (C=%d CP=%d)", C, CP);
  PCRL_REQUIRE(dtype == PCRL_F32 || dtype == PCRL_BF16, "nchw_to_nhwc_pad: bad dtype %d", dtype);
  const int64_t M = (int64_t)N * HW;
  int64_t b = (M + 255) / 256;
  if (b > 65536) b = 65536;
  if (dtype == PCRL_BF16) hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<bf16, 8>), dim3((unsigned)b), dim3(256), 0, as_stream(stream), x, (bf16*)out, C, HW, M);
  else hipLaunchKernelGGL((nchw_to_nhwc_pad_kernel<float, 8>), dim3((unsigned)b), dim3(256), 0, as_stream(stream), x, (float*)out, C, HW, M);
  return pcrl_check_launch("nchw_to_nhwc_pad");
}

extern "C" int pcrl_add_f32(const float* a, const float* b, float* out, int64_t n, pcrl_stream_t stream) {
  PCRL_REQUIRE(a && b && out && n > 0, "add_f32: bad arguments");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a, b, out, n);
  return pcrl_check_launch("add_f32");
}
