// Backward of the projection / predictor heads -- x_pro = bn(avgpool(x)); x_pre = Linear(ReLU(BatchNorm1d(Linear(x_pro))))
// (models/pcrlv2_model_3d.py:55-59,67-70; models/pcrlv2_model.py:108-111,124-127) -- in TWO launches instead of nine.
//
// Each half of the chain is "Linear backward, then BatchNorm1d backward of what it produced":
//     t[n][c]  = add[n][c] + sum_k dy[n][k] * W[k][c]          (data gradient of y = x W^T, W float32 [K][C]; add: a second gradient of the same
//                                                               tensor -- x_pro is also a cosine operand -- or NULL)
//     dx[:, c] = BatchNorm1d backward of t[:, c] through (xbn, [ybn: ReLU mask], gamma, mean, rstd) -> dgamma[c], dbeta[c]
//     dW[k][c] = sum_n dy[n][k] * xin[n][c],   db[k] = sum_n dy[n][k]
// A BatchNorm1d column needs ALL rows of its column of t and nothing else, and a block that computes a column of t has them: the three
// products and the normalisation's backward are roles of ONE launch (aten: addmm backward x 2, native_batch_norm_backward, threshold_backward,
// sum).  Rows <= 512 (6 local views x 64 crops = 384 at most in BASELINE's configurations).  Deterministic: fixed-order sums, no atomics.
#include "common.h"

namespace {

constexpr int RMAX = 8;     // rows per lane: N <= 512

// role A: BLOCK = one column c; the four waves split the K range of the product (a quarter each: the walk over K is the kernel's critical
// chain -- with a whole column per wave a launch took 36 us), lanes own rows n = lane + 64 i; wave 0 combines through LDS (fixed order) and
// runs the BatchNorm1d backward of the column.
__device__ __forceinline__ void role_column(const float* __restrict__ dy, const float* __restrict__ W, const float* __restrict__ add,
                                            const float* __restrict__ xbn, const float* __restrict__ ybn, const float* __restrict__ gamma,
                                            const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int K, int C, int relu, int c, int lane, int wv,
                                            float* __restrict__ sm) {
  float t[RMAX];
#pragma unroll
  for (int i = 0; i < RMAX; ++i) t[i] = 0.f;
  if (dy) {
    const int kq = ((K / 4 + 3) / 4) * 4;                 // a multiple of 4 per wave
    const int k0 = wv * kq, k1 = min(K, k0 + kq);
    for (int k = k0; k < k1; k += 4) {
      const float w0 = W[(int64_t)k * C + c], w1 = W[(int64_t)(k + 1) * C + c], w2 = W[(int64_t)(k + 2) * C + c], w3 = W[(int64_t)(k + 3) * C + c];   // wave-uniform
#pragma unroll
      for (int i = 0; i < RMAX; ++i) {
        const int n = lane + 64 * i;
        if (n < N) {
          const float4 d = *reinterpret_cast<const float4*>(dy + (int64_t)n * K + k);
          t[i] = fmaf(d.x, w0, fmaf(d.y, w1, fmaf(d.z, w2, fmaf(d.w, w3, t[i]))));
        }
      }
    }
    if (wv > 0) {
#pragma unroll
      for (int i = 0; i < RMAX; ++i) sm[((wv - 1) * RMAX + i) * 64 + lane] = t[i];
    }
  }
  __syncthreads();
  if (wv != 0) return;
#pragma unroll
  for (int i = 0; i < RMAX; ++i) {
    const int n = lane + 64 * i;
    if (dy) t[i] = ((t[i] + sm[(0 * RMAX + i) * 64 + lane]) + sm[(1 * RMAX + i) * 64 + lane]) + sm[(2 * RMAX + i) * 64 + lane];
    if (add && n < N) t[i] += add[(int64_t)n * C + c];
  }
  const double mu = mean[c], rs = rstd[c];
  double s1 = 0.0, s2 = 0.0;
  float xh[RMAX];
#pragma unroll
  for (int i = 0; i < RMAX; ++i) {
    const int n = lane + 64 * i;
    xh[i] = 0.f;
    if (n < N) {
      const int64_t o = (int64_t)n * C + c;
      if (relu && ybn[o] <= 0.f) t[i] = 0.f;
      xh[i] = (float)(((double)xbn[o] - mu) * rs);
      s1 += (double)t[i];
      s2 += (double)t[i] * (double)xh[i];
    }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
  }
  const double g1 = (double)gamma[c] * rs;
#pragma unroll
  for (int i = 0; i < RMAX; ++i) {
    const int n = lane + 64 * i;
    if (n < N) dx[(int64_t)n * C + c] = (float)(g1 * ((double)t[i] - s1 / N - (double)xh[i] * s2 / N));
  }
}

__global__ void __launch_bounds__(256) head_bwd_stage_kernel(const float* __restrict__ dy, const float* __restrict__ W, const float* __restrict__ add,
                                                             const float* __restrict__ xin, const float* __restrict__ xbn, const float* __restrict__ ybn,
                                                             const float* __restrict__ gamma, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dW, float* __restrict__ db, int N, int K, int C,
                                                             int relu, int nA, int nBx) {
  __shared__ float sm[3 * RMAX * 64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int b = blockIdx.x;
  if (b < nA) {                                   // ---- role A: one column of t and its BatchNorm1d backward
    role_column(dy, W, add, xbn, ybn, gamma, mean, rstd, dx, dgamma, dbeta, N, K, C, relu, b, lane, wv, sm);
    return;
  }
  b -= nA;
  const int nB = nBx * ((K + 3) / 4);
  if (b < nB) {                                   // ---- role B: dW[k][c] = sum_n dy[n][k] xin[n][c]; wave = one k, lane = 4 consecutive c
    const int k = (b / nBx) * 4 + wv, c = ((b % nBx) * 64 + lane) * 4;
    if (k >= K || c >= C) return;
    float4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int n = 0; n < N; ++n) {
      const float d = dy[(int64_t)n * K + k];     // wave-uniform
      const float4 x = *reinterpret_cast<const float4*>(xin + (int64_t)n * C + c);
      acc.x = fmaf(d, x.x, acc.x);
      acc.y = fmaf(d, x.y, acc.y);
      acc.z = fmaf(d, x.z, acc.z);
      acc.w = fmaf(d, x.w, acc.w);
    }
    *reinterpret_cast<float4*>(dW + (int64_t)k * C + c) = acc;
    return;
  }
  b -= nB;                                        // ---- role C: db[k] = sum_n dy[n][k]
  const int k = b * 4 + wv;
  if (k >= K) return;
  float s = 0.f;
  for (int n = lane; n < N; n += 64) s += dy[(int64_t)n * K + k];
  s = wave_sum(s);
  if (lane == 0) db[k] = s;
}

}  // namespace

extern "C" int pcrl_head_bwd_stage(const float* dy, const float* W, const float* add, const float* xin, const float* xbn, const float* ybn,
                                   const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta, float* dW,
                                   float* db, int N, int K, int C, int relu, pcrl_stream_t stream) {
  PCRL_REQUIRE(xbn && gamma && mean && rstd && dx && dgamma && dbeta && (dy || add), "head_bwd_stage: null pointer");
  PCRL_REQUIRE(N > 1 && N <= 64 * RMAX && C > 0 && C % 4 == 0, "head_bwd_stage: N=%d (2..%d rows) C=%d (a multiple of 4)", N, 64 * RMAX, C);
  PCRL_REQUIRE(!relu || ybn, "head_bwd_stage: the ReLU mask needs the normalisation's output");
  PCRL_REQUIRE(!dy || (W && xin && dW && db && K > 0 && K % 4 == 0), "head_bwd_stage: the Linear part needs W, xin, dW, db and K %% 4 == 0 (K=%d)", K);
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  PCRL_REQUIRE(!dy || (al(dy) && al(xin) && al(dW)), "head_bwd_stage: dy, xin and dW must be 16-byte aligned");
  const int nA = C, nBx = (C + 255) / 256;
  const int nB = dy ? nBx * ((K + 3) / 4) : 0, nC = dy ? (K + 3) / 4 : 0;
  hipLaunchKernelGGL(head_bwd_stage_kernel, dim3(nA + nB + nC), dim3(256), 0, as_stream(stream), dy, W, add, xin, xbn, ybn, gamma, mean, rstd, dx, dgamma,
                     dbeta, dW, db, N, dy ? K : 0, C, relu, nA, nBx);
  return pcrl_check_launch("head_bwd_stage");
}
