// Latency-bound tail of the step: projection/predictor heads on [rows][C] float32 (BatchNorm1d, Linear, ReLU;
// models/pcrlv2_model_3d.py:55-59,69-70), trilinear upsampling of the 1-channel deep-supervision maps (:125-126),
// sigmoid (:79,82), MSE and cosine losses (train_3d.py:56-57,86-92,135-137), the fused SGD update
// (train_3d.py:48-51,151) and the once-per-step weight packing for the MFMA convolutions.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// BatchNorm1d (training mode) on [rows][C]: one WAVE per channel (lanes stride over the rows, fp64 wave reductions).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bn1d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* running_mean, float* running_var,
                                                       float momentum, float eps, float* mean, float* rstd, int rows, int C, int relu) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  double s1 = 0.0;
  for (int r = lane; r < rows; r += 64) s1 += (double)x[(int64_t)r * C + c];
  const double mu = wave_sum(s1) / rows;
  double s2 = 0.0;
  for (int r = lane; r < rows; r += 64) {
    const double d = (double)x[(int64_t)r * C + c] - mu;
    s2 += d * d;
  }
  const double var = wave_sum(s2) / rows;
  const double rs = 1.0 / sqrt(var + (double)eps);
  const float g = gamma[c], b = beta[c];
  for (int r = lane; r < rows; r += 64) {
    float v = (float)(((double)x[(int64_t)r * C + c] - mu) * rs) * g + b;
    if (relu && v < 0.f) v = 0.f;
    y[(int64_t)r * C + c] = v;
  }
  if (lane == 0) {
    mean[c] = (float)mu;
    rstd[c] = (float)rs;
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
    if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * (rows > 1 ? var * rows / (rows - 1.0) : var));
  }
}

__global__ void __launch_bounds__(256) bn1d_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, float* __restrict__ dx, float* dgamma, float* dbeta,
                                                       int rows, int C, int relu) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const double mu = mean[c], rs = rstd[c];
  double s1 = 0.0, s2 = 0.0;
  for (int r = lane; r < rows; r += 64) {
    const int64_t i = (int64_t)r * C + c;
    const double dz = (relu && y[i] <= 0.f) ? 0.0 : (double)dy[i];
    s1 += dz;
    s2 += dz * ((double)x[i] - mu) * rs;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
  }
  const double g1 = (double)gamma[c] * rs;
  for (int r = lane; r < rows; r += 64) {
    const int64_t i = (int64_t)r * C + c;
    const double dz = (relu && y[i] <= 0.f) ? 0.0 : (double)dy[i];
    const double xh = ((double)x[i] - mu) * rs;
    dx[i] = (float)(g1 * (dz - s1 / rows - xh * s2 / rows));
  }
}

// ---------------------------------------------------------------------------------------------
// Linear layers of the heads: one small LDS-tiled SGEMM, C[m][n] = sum_k A(m,k) * B(k,n) (+ bias[n]), operands addressed
// through (row, col) strides so the three products of a Linear (y = x W^T, dx = dy W, dW = dy^T x) share the kernel.
// 16x16 threads, 32x32 tile, K chunks of 64.  Sizes here: M, N, K <= 512: a handful of blocks, so a launch costs its chain of
// global round trips.  The staging index therefore runs along whichever axis of an operand is contiguous (k for x W^T, the
// row/column index for the transposed products: full 256-byte runs instead of 32 scattered words per wave), and chunk c+1 is
// in flight in registers while chunk c is multiplied out of LDS.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sgemm_small_kernel(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                          const float* __restrict__ B, int64_t sbk, int64_t sbn,
                                                          const float* __restrict__ bias, float* __restrict__ Cm, int M, int N, int K) {
  constexpr int KC = 64;
  __shared__ float As[KC][33], Bs[KC][33];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const bool a_kfast = sak == 1, b_kfast = sbk == 1;   // block-uniform
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  float ra[KC / 8], rb[KC / 8];
#define SG_LOAD(k0_)                                                                                   \
  _Pragma("unroll") for (int h = 0; h < KC / 8; ++h) {                                                 \
    const int idx = threadIdx.x + 256 * h;                                                             \
    const int ka = a_kfast ? (idx & (KC - 1)) : (idx >> 5), ma = a_kfast ? (idx >> 6) : (idx & 31);    \
    const int kb = b_kfast ? (idx & (KC - 1)) : (idx >> 5), nb = b_kfast ? (idx >> 6) : (idx & 31);    \
    ra[h] = (m0 + ma < M && (k0_) + ka < K) ? A[(m0 + ma) * sam + ((k0_) + ka) * sak] : 0.f;           \
    rb[h] = (n0 + nb < N && (k0_) + kb < K) ? B[((k0_) + kb) * sbk + (n0 + nb) * sbn] : 0.f;           \
  }
  SG_LOAD(0)
  for (int k0 = 0; k0 < K; k0 += KC) {
#pragma unroll
    for (int h = 0; h < KC / 8; ++h) {
      const int idx = threadIdx.x + 256 * h;
      const int ka = a_kfast ? (idx & (KC - 1)) : (idx >> 5), ma = a_kfast ? (idx >> 6) : (idx & 31);
      const int kb = b_kfast ? (idx & (KC - 1)) : (idx >> 5), nb = b_kfast ? (idx >> 6) : (idx & 31);
      As[ka][ma] = ra[h];
      Bs[kb][nb] = rb[h];
    }
    __syncthreads();
    if (k0 + KC < K) { SG_LOAD(k0 + KC) }
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      const float a0 = As[kk][ty], a1 = As[kk][ty + 16], b0 = Bs[kk][tx], b1 = Bs[kk][tx + 16];
      acc[0][0] = fmaf(a0, b0, acc[0][0]);
      acc[0][1] = fmaf(a0, b1, acc[0][1]);
      acc[1][0] = fmaf(a1, b0, acc[1][0]);
      acc[1][1] = fmaf(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
#undef SG_LOAD
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
      if (m < M && n < N) Cm[(int64_t)m * N + n] = acc[i][j] + (bias ? bias[n] : 0.f);
    }
}
// ---- the same three products for the usual head sizes (rows = 32 .. 192, channels 64 .. 512, all multiples of 4): the tiled kernel above
// has 16 .. 32 blocks walking K in serial LDS chunks (30 us a launch, 54 launches a step); here every output group owns its whole
// contraction with independent 16-byte loads -- one global round trip -- and the parallelism comes from the outputs (128+ blocks).
// y[r][o] = sum_k x[r][k] w[o][k] + b[o].  Wave = one output column o and 32 rows; lanes split k in float4 groups; 32 wave reductions.
__global__ void __launch_bounds__(256) linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ y, int rows, int Cin, int Cout) {
  const int lane = threadIdx.x & 63, o = blockIdx.x * 4 + (threadIdx.x >> 6), r0 = blockIdx.y * 32;
  if (o >= Cout) return;
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.f;
  for (int k = lane * 4; k < Cin; k += 256) {
    const float4 wv = *reinterpret_cast<const float4*>(w + (int64_t)o * Cin + k);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (r0 + i < rows) {   // wave-uniform
        const float4 xv = *reinterpret_cast<const float4*>(x + (int64_t)(r0 + i) * Cin + k);
        acc[i] = fmaf(xv.x, wv.x, fmaf(xv.y, wv.y, fmaf(xv.z, wv.z, fmaf(xv.w, wv.w, acc[i]))));
      }
    }
  }
  // 32 sums over 64 lanes as a halving butterfly (32 exchanges instead of 32 x 6): at distance 32, 16, .., 2 a lane keeps the half of its
  // values selected by that bit of its index and adds the partner's copy of them; lane l ends with row l >> 1 (distance 1: same row twice).
#define HALVE(n_, dist_)                                                                   \
  _Pragma("unroll") for (int i = 0; i < (n_); ++i) {                                       \
    const bool hi = (lane & (dist_)) != 0;                                                 \
    const float keep = hi ? acc[(n_) + i] : acc[i], send = hi ? acc[i] : acc[(n_) + i];    \
    acc[i] = keep + __shfl_xor(send, (dist_), 64);                                         \
  }
  HALVE(16, 32)
  HALVE(8, 16)
  HALVE(4, 8)
  HALVE(2, 4)
  HALVE(1, 2)
#undef HALVE
  const float tot = acc[0] + __shfl_xor(acc[0], 1, 64);
  const int r = r0 + (lane >> 1);
  if ((lane & 1) == 0 && r < rows) y[(int64_t)r * Cout + o] = tot + (bias ? bias[o] : 0.f);
}
// dx[r][k] = sum_o dy[r][o] w[o][k].  Block = 4 rows x 256 columns; lane owns 4 consecutive k, the four waves split o, LDS combine.
__global__ void __launch_bounds__(256) linear_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int rows, int Cin,
                                                        int Cout) {
  __shared__ float4 red[3][4][64];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = (blockIdx.x * 64 + lane) * 4, r0 = blockIdx.y * 4;
  const bool kin = k < Cin;
  float4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = float4{0.f, 0.f, 0.f, 0.f};
  const int per = (Cout + 3) / 4, o0 = wv * per, o1 = min(Cout, o0 + per);
  const float* dyr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dyr[i] = dy + (int64_t)min(r0 + i, rows - 1) * Cout;
#pragma unroll 4
  for (int o = o0; o < o1; ++o) {
    const float4 t = kin ? *reinterpret_cast<const float4*>(w + (int64_t)o * Cin + k) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float d = dyr[i][o];   // wave-uniform
      acc[i].x = fmaf(d, t.x, acc[i].x);
      acc[i].y = fmaf(d, t.y, acc[i].y);
      acc[i].z = fmaf(d, t.z, acc[i].z);
      acc[i].w = fmaf(d, t.w, acc[i].w);
    }
  }
  if (wv > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wv - 1][i][lane] = acc[i];
  }
  __syncthreads();
  if (wv == 0 && kin) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 a = acc[i];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float4 t = red[q][i][lane];
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      if (r0 + i < rows) *reinterpret_cast<float4*>(dx + (int64_t)(r0 + i) * Cin + k) = a;
    }
  }
}
// dw[o][k] = sum_r dy[r][o] x[r][k].  Wave = one output row o, lane owns 4 consecutive k; the rows are walked with independent loads.
__global__ void __launch_bounds__(256) linear_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, int rows, int Cin,
                                                        int Cout) {
  const int lane = threadIdx.x & 63, o = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = (blockIdx.x * 64 + lane) * 4;
  if (o >= Cout || k >= Cin) return;
  float4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int r = 0; r < rows; ++r) {
    const float d = dy[(int64_t)r * Cout + o];   // wave-uniform
    const float4 t = *reinterpret_cast<const float4*>(x + (int64_t)r * Cin + k);
    acc.x = fmaf(d, t.x, acc.x);
    acc.y = fmaf(d, t.y, acc.y);
    acc.z = fmaf(d, t.z, acc.z);
    acc.w = fmaf(d, t.w, acc.w);
  }
  *reinterpret_cast<float4*>(dw + (int64_t)o * Cin + k) = acc;
}
__global__ void __launch_bounds__(256) colsum_small_kernel(const float* __restrict__ v, float* __restrict__ out, int rows, int C) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float s = 0.f;
  for (int r = lane; r < rows; r += 64) s += v[(int64_t)r * C + c];
  s = wave_sum(s);
  if (lane == 0) out[c] = s;
}

// ---------------------------------------------------------------------------------------------
// Trilinear upsample, align_corners=False, integer scale s: src = (dst + 0.5)/s - 0.5, clamped at 0;
// i0 = floor(src), i1 = min(i0+1, n-1), lambda = src - i0  (ATen area_pixel_compute_source_index).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void tri_src(int o, int n_in, float inv_s, int& i0, int& i1, float& l1) {
  float src = ((float)o + 0.5f) * inv_s - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + ((i0 < n_in - 1) ? 1 : 0);
  l1 = src - (float)i0;
}

__global__ void __launch_bounds__(256) tri_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, Dims g, int s, int64_t total) {
  const Dims go{g.N, g.D * s, g.H * s, g.W * s};
  const float inv = 1.f / (float)s;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int n, d, h, w;
    decode_voxel(i, go, n, d, h, w);
    int d0, d1, h0, h1, w0, w1;
    float ld, lh, lw;
    tri_src(d, g.D, inv, d0, d1, ld);
    tri_src(h, g.H, inv, h0, h1, lh);
    tri_src(w, g.W, inv, w0, w1, lw);
    const float* b = x + (int64_t)n * g.D * g.H * g.W;
#define X_(dd, hh, ww) b[((int64_t)(dd)*g.H + (hh)) * g.W + (ww)]
    const float v = (1.f - ld) * ((1.f - lh) * ((1.f - lw) * X_(d0, h0, w0) + lw * X_(d0, h0, w1)) +
                                  lh * ((1.f - lw) * X_(d0, h1, w0) + lw * X_(d0, h1, w1))) +
                    ld * ((1.f - lh) * ((1.f - lw) * X_(d1, h0, w0) + lw * X_(d1, h0, w1)) +
                          lh * ((1.f - lw) * X_(d1, h1, w0) + lw * X_(d1, h1, w1)));
#undef X_
    y[i] = v;
  }
}

// Backward as a gather: input voxel i collects from every output voxel whose stencil touches it.
// Per dimension at most 3*s-1 < 16 candidate outputs (s <= 4 keeps the weight tables in registers).
constexpr int TRI_MAXW = 16;
__device__ __forceinline__ int tri_weights(int i, int n_in, int s, float inv_s, int& obeg, float* wts) {
  int lo = s * (i - 1);
  if (lo < 0) lo = 0;
  int hi = s * (i + 2) - 1;
  const int n_out = n_in * s;
  if (hi > n_out - 1) hi = n_out - 1;
  obeg = lo;
  int cnt = hi - lo + 1;
  if (cnt > TRI_MAXW) cnt = TRI_MAXW;
  for (int k = 0; k < cnt; ++k) {
    int i0, i1;
    float l1;
    tri_src(lo + k, n_in, inv_s, i0, i1, l1);
    float wgt = 0.f;
    if (i0 == i) wgt += 1.f - l1;
    if (i1 == i) wgt += l1;
    wts[k] = wgt;
  }
  return cnt;
}

__global__ void __launch_bounds__(256) tri_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, Dims g, int s, int64_t total) {
  const float inv = 1.f / (float)s;
  const int Do = g.D * s, Ho = g.H * s, Wo = g.W * s;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int n, d, h, w;
    decode_voxel(i, g, n, d, h, w);
    float wd[TRI_MAXW], wh[TRI_MAXW], ww[TRI_MAXW];
    int od, oh, ow;
    const int nd = tri_weights(d, g.D, s, inv, od, wd);
    const int nh = tri_weights(h, g.H, s, inv, oh, wh);
    const int nw = tri_weights(w, g.W, s, inv, ow, ww);
    const float* b = dy + (int64_t)n * Do * Ho * Wo;
    float acc = 0.f;
    for (int a = 0; a < nd; ++a) {
      if (wd[a] == 0.f) continue;
      for (int c = 0; c < nh; ++c) {
        if (wh[c] == 0.f) continue;
        const float wdh = wd[a] * wh[c];
        const float* row = b + ((int64_t)(od + a) * Ho + (oh + c)) * Wo + ow;
        float r = 0.f;
        for (int e = 0; e < nw; ++e) r += ww[e] * row[e];
        acc += wdh * r;
      }
    }
    dx[i] = acc;
  }
}

// The same backward, separable: the trilinear weights factor per axis, so a block takes one input d-plane (n, d), reduces the (at most 11)
// output planes that touch it along d into an [Ho][Wo] image in LDS (coalesced reads: every output plane is read by its two or three input
// planes only), then along h, then along w.  Per input voxel 11 + 11 + 11 products instead of up to 11^3 gathered ones: the x4 upsampling of
// the 16 x 16 x 8 map took 91 us alone on the chip (110 - 470 us next to other streams) at the head of the backward's critical chain.
__global__ void __launch_bounds__(256) tri_bwd_planes_kernel(const float* __restrict__ dy, float* __restrict__ dx, Dims g, int s) {
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  const int Ho = g.H * s, Wo = g.W * s, Do = g.D * s;
  float* p1 = tsm;                 // [Ho][Wo]
  float* p2 = tsm + Ho * Wo;       // [H][Wo]
  const float inv = 1.f / (float)s;
  const int n = blockIdx.x / g.D, d = blockIdx.x % g.D, tid = threadIdx.x;
  float wt[TRI_MAXW];
  int o0;
  const int nd = tri_weights(d, g.D, s, inv, o0, wt);
  const float* b = dy + ((int64_t)n * Do + o0) * Ho * Wo;
  for (int i = tid; i < Ho * Wo; i += 256) {
    float a = 0.f;
    for (int k = 0; k < nd; ++k) a += wt[k] * b[(int64_t)k * Ho * Wo + i];
    p1[i] = a;
  }
  __syncthreads();
  for (int i = tid; i < g.H * Wo; i += 256) {
    const int h = i / Wo, wo = i - h * Wo;
    const int nh = tri_weights(h, g.H, s, inv, o0, wt);
    float a = 0.f;
    for (int k = 0; k < nh; ++k) a += wt[k] * p1[(o0 + k) * Wo + wo];
    p2[i] = a;
  }
  __syncthreads();
  for (int i = tid; i < g.H * g.W; i += 256) {
    const int h = i / g.W, w = i - h * g.W;
    const int nw = tri_weights(w, g.W, s, inv, o0, wt);
    float a = 0.f;
    for (int k = 0; k < nw; ++k) a += wt[k] * p2[h * Wo + o0 + k];
    dx[(((int64_t)n * g.D + d) * g.H + h) * g.W + w] = a;
  }
}

// ---------------------------------------------------------------------------------------------
// sigmoid / MSE / cosine
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = 1.f / (1.f + expf(-x[i]));
}
__global__ void __launch_bounds__(256) sigmoid_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                          float* __restrict__ dpre, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float a = out[i];
    dpre[i] = dout[i] * a * (1.f - a);
  }
}

constexpr int RED_CHUNK = 4096;  // elements per first-stage block
__global__ void __launch_bounds__(256) mse_partial_kernel(const float* __restrict__ p, const float* __restrict__ gt, double* __restrict__ ws, int64_t n) {
  __shared__ double red[4];
  const int64_t beg = (int64_t)blockIdx.x * RED_CHUNK;
  const int64_t end = (beg + RED_CHUNK < n) ? beg + RED_CHUNK : n;
  double s = 0.0;
  for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
    const float d = p[i] - gt[i];
    s += (double)(d * d);
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) mse_finish_kernel(const double* __restrict__ ws, float* __restrict__ loss, int blocks, double inv_n) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < blocks; i += 256) s += ws[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) loss[0] = (float)(s * inv_n);
}
__global__ void __launch_bounds__(256) mse_bwd_kernel(const float* __restrict__ p, const float* __restrict__ gt, const float* __restrict__ dloss,
                                                      float* __restrict__ dp, int64_t n, float two_over_n) {
  const float g = dloss[0] * two_over_n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dp[i] = g * (p[i] - gt[i]);
}

// One block; one wave per row (round-robin): dot, |x|, |y| by wave-shuffle reduction; mean over rows in fp64.
__global__ void __launch_bounds__(256) cosine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                                                         float* __restrict__ saved, int rows, int C, float eps) {
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  double part = 0.0;
  for (int r = wid; r < rows; r += 4) {
    float dot = 0.f, xx = 0.f, yy = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float a = x[(int64_t)r * C + c], b = y[(int64_t)r * C + c];
      dot += a * b;
      xx += a * a;
      yy += b * b;
    }
    dot = wave_sum(dot);
    xx = wave_sum(xx);
    yy = wave_sum(yy);
    const float nx = sqrtf(xx), ny = sqrtf(yy);
    if (lane == 0) {
      saved[r * 3 + 0] = dot;
      saved[r * 3 + 1] = nx;
      saved[r * 3 + 2] = ny;
      part += (double)(dot / (fmaxf(nx, eps) * fmaxf(ny, eps)));
    }
  }
  __syncthreads();
  if (lane == 0) red[wid] = part;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)((red[0] + red[1] + red[2] + red[3]) / rows);
}
// d/dx [x.y / (max(|x|,eps) max(|y|,eps))] = y/(nx' ny') - (x.y) x / (nx'^3 ny')   (second term only where |x| > eps)
__global__ void __launch_bounds__(256) cosine_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ saved,
                                                         const float* __restrict__ dout, float* __restrict__ dx, int rows, int C, float eps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C) return;
  const int r = i / C;
  const float dot = saved[r * 3], nx = saved[r * 3 + 1], ny = saved[r * 3 + 2];
  const float nxc = fmaxf(nx, eps), nyc = fmaxf(ny, eps);
  const float g = dout[0] / (float)rows;
  float v = y[i] / (nxc * nyc);
  if (nx > eps) v -= dot * x[i] / (nxc * nxc * nxc * nyc);
  dx[i] = g * v;
}

// ---------------------------------------------------------------------------------------------
// All cosine terms of one training step in one call (train_3d.py:119-134: the global pair and the twelve (global, local_i) pairs, two
// cosine means each = 26 terms over [rows, C_t] matrices at randomly drawn scales).  out[g] = sum over the terms of group g of
// w_t * mean_r cos(x_t[r], y_t[r]); the gradient flows to the x operands only (the reference detaches y).  A step has
// 26 x 32 rows of <= 512 channels; separate launches (26 forward + 26 backward + ~150 elementwise kernels for the negations, halves,
// sums and stacks) cost ~1 ms per step in launch latency.  Descriptors travel in the kernel arguments (<= 32 terms).
// ---------------------------------------------------------------------------------------------
constexpr int COS_MAX_TERMS = 32;
struct CosTerms {
  const float* x[COS_MAX_TERMS];
  const float* y[COS_MAX_TERMS];
  float* dx[COS_MAX_TERMS];     // backward: gradient buffer of the term's x (several terms may share one)
  float w[COS_MAX_TERMS];
  int C[COS_MAX_TERMS];
  int group[COS_MAX_TERMS];
  int first[COS_MAX_TERMS];     // backward: 1 = first term that writes its dx buffer (store), 0 = accumulate
  int n, rows, ngroups;
  float eps;
};
// Forward, launch 1 of 2: block k = term k (16 waves, a wave per row at a time) -> vals[k] = w_k * mean_r cos(x_k[r], y_k[r]).
// One block for all terms walked them one after the other: 84 us of dependent loads for 26 terms (rocprofv3 r02c).
__global__ void __launch_bounds__(1024) cosine_terms_fwd_kernel(const CosTerms t, double* __restrict__ vals) {
  __shared__ double part[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int k = blockIdx.x;
  const float* __restrict__ x = t.x[k];
  const float* __restrict__ y = t.y[k];
  const int C = t.C[k];
  double s = 0.0;
  for (int r = wid; r < t.rows; r += nw) {
    float dot = 0.f, xx = 0.f, yy = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float a = x[(int64_t)r * C + c], b = y[(int64_t)r * C + c];
      dot += a * b;
      xx += a * a;
      yy += b * b;
    }
    dot = wave_sum(dot);
    xx = wave_sum(xx);
    yy = wave_sum(yy);
    s += (double)(dot / (fmaxf(sqrtf(xx), t.eps) * fmaxf(sqrtf(yy), t.eps)));
  }
  if (lane == 0) part[wid] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = 0.0;
    for (int w = 0; w < nw; ++w) v += part[w];   // fixed order
    vals[k] = v * (double)t.w[k] / (double)t.rows;
  }
}
// launch 2 of 2: out[g] = sum of the terms of group g, in term order
__global__ void __launch_bounds__(64) cosine_terms_sum_kernel(const CosTerms t, const double* __restrict__ vals, float* __restrict__ out) {
  const int g = threadIdx.x;
  if (g >= t.ngroups) return;
  double v = 0.0;
  for (int k = 0; k < t.n; ++k)
    if (t.group[k] == g) v += vals[k];
  out[g] = (float)v;
}
// Backward: dx_t = sum over the terms that share the buffer dx_t, in term order, of dout[group] * w / rows * d cos(x, y) / dx.
// Block = (4 rows, term k): only the FIRST term of a buffer does work -- it walks the later terms with the same dx, keeps the
// row's gradient in registers and stores it once (no read-modify-write, no ordering between blocks; the summation order is the
// term order, as in the one-block form this replaces: 148 us -> a few us).
__global__ void __launch_bounds__(256) cosine_terms_bwd_kernel(const CosTerms t, const float* __restrict__ dout) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int k0 = blockIdx.y;
  if (!t.first[k0] || r >= t.rows) return;
  float* __restrict__ dx = t.dx[k0];
  const int C = t.C[k0];
  for (int cb = 0; cb < C; cb += 512) {
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    for (int k = k0; k < t.n; ++k) {
      if (t.dx[k] != dx) continue;
      const float* __restrict__ x = t.x[k];
      const float* __restrict__ y = t.y[k];
      const float g = dout[t.group[k]] * t.w[k] / (float)t.rows;
      float dot = 0.f, xx = 0.f, yy = 0.f;
      for (int c = lane; c < C; c += 64) {
        const float a = x[(int64_t)r * C + c], b = y[(int64_t)r * C + c];
        dot += a * b;
        xx += a * a;
        yy += b * b;
      }
      dot = wave_sum(dot);
      xx = wave_sum(xx);
      yy = wave_sum(yy);
      const float nx = sqrtf(xx), ny = sqrtf(yy), nxc = fmaxf(nx, t.eps), nyc = fmaxf(ny, t.eps);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = cb + q * 64 + lane;
        if (c < C) {
          float v = y[(int64_t)r * C + c] / (nxc * nyc);
          if (nx > t.eps) v -= dot * x[(int64_t)r * C + c] / (nxc * nxc * nxc * nyc);
          acc[q] += g * v;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = cb + q * 64 + lane;
      if (c < C) dx[(int64_t)r * C + c] = acc[q];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NT-Xent (SimCLR) contrastive loss -- OPTIONAL EXTRA, not in the reference (the reference's contrastive
// term is the negative cosine similarity above).  z = [z1; z2] is [R = 2N][C]; row i's positive is row (i + N) mod R.
//   zn = z / max(|z|, eps);  S = zn zn^T / tau;  loss = mean_i ( logsumexp_{k != i} S[i][k] - S[i][pos_i] )
// fwd: normalise (wave per row) -> S by the small SGEMM -> per-row log-softmax (S is overwritten by the softmax P, diagonal 0).
// bwd: G = (P - onehot(pos)) * dloss / R;  dzn = (G + G^T) zn / tau;  dz = (dzn - zn (zn . dzn)) / max(|z|, eps).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ntx_normalize_kernel(const float* __restrict__ z, float* __restrict__ zn, float* __restrict__ nrm,
                                                            int R, int C, float eps) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float ss = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = z[(int64_t)r * C + c];
    ss += v * v;
  }
  const float n = fmaxf(sqrtf(wave_sum(ss)), eps);
  for (int c = lane; c < C; c += 64) zn[(int64_t)r * C + c] = z[(int64_t)r * C + c] / n;
  if (lane == 0) nrm[r] = n;
}
// one wave per row: S[i][:] (already divided by tau) -> P[i][:], li[i]
__global__ void __launch_bounds__(256) ntx_rows_kernel(float* __restrict__ S, float* __restrict__ li, int R) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= R) return;
  float* row = S + (int64_t)i * R;
  const int pos = (i + R / 2) % R;
  float m = -INFINITY;
  for (int k = lane; k < R; k += 64)
    if (k != i) m = fmaxf(m, row[k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float se = 0.f;
  for (int k = lane; k < R; k += 64)
    if (k != i) se += expf(row[k] - m);
  se = wave_sum(se);
  const float lse = m + logf(se);
  const float spos = row[pos];
  for (int k = lane; k < R; k += 64) row[k] = (k == i) ? 0.f : expf(row[k] - lse);
  if (lane == 0) li[i] = lse - spos;
}
__global__ void __launch_bounds__(256) scale_kernel(float* __restrict__ v, int64_t n, float a) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v[i] *= a;
}
__global__ void __launch_bounds__(256) ntx_mean_kernel(const float* __restrict__ li, float* __restrict__ loss, int R) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < R; i += 256) s += (double)li[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) loss[0] = (float)(s / R);
}
// Gs[i][k] = (G[i][k] + G[k][i]) / tau with G = (P - onehot(pos)) * dloss / R   (in place over P is not possible: needs P^T)
__global__ void __launch_bounds__(256) ntx_gsym_kernel(const float* __restrict__ P, const float* __restrict__ dloss, float* __restrict__ Gs,
                                                       int R, float inv_tau) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= R * R) return;
  const int i = idx / R, k = idx % R;
  const float g = dloss[0] / (float)R;
  const float a = P[idx] - ((k == (i + R / 2) % R) ? 1.f : 0.f);
  const float b = P[(int64_t)k * R + i] - ((i == (k + R / 2) % R) ? 1.f : 0.f);
  Gs[idx] = (i == k) ? 0.f : (a + b) * g * inv_tau;
}
__global__ void __launch_bounds__(256) ntx_denorm_kernel(const float* __restrict__ z, const float* __restrict__ zn, const float* __restrict__ dzn,
                                                         const float* __restrict__ nrm, float* __restrict__ dz, int R, int C, float eps) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) dot += zn[(int64_t)r * C + c] * dzn[(int64_t)r * C + c];
  dot = wave_sum(dot);
  const float n = nrm[r];
  const bool clamped = n <= eps;   // |z| below eps: zn = z / eps, no projection term
  for (int c = lane; c < C; c += 64) {
    const int64_t q = (int64_t)r * C + c;
    dz[q] = clamped ? dzn[q] / n : (dzn[q] - zn[q] * dot) / n;
  }
  (void)z;
}

// ---------------------------------------------------------------------------------------------
// SGD over a flat arena; per-tensor flags looked up by binary search on the offsets table.
// ---------------------------------------------------------------------------------------------
// `skip` (may be null): device flag of the divergence guard (train_3d.py:140-142 decided on the device); non-zero = the whole update is
// skipped -- parameters and momentum buffers stay bit-unchanged, exactly what the reference's `continue` leaves behind.
__global__ void __launch_bounds__(256) sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  const int64_t* __restrict__ offsets, const int32_t* __restrict__ flags, int ntensors,
                                                  int64_t total, float lr, float momentum, float wd, float gscale, const float* __restrict__ skip) {
  if (skip && *skip != 0.f) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int lo = 0, hi = ntensors;  // find t with offsets[t] <= i < offsets[t+1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    const int f = flags[lo];
    if (!(f & 1)) continue;
    const float pv = p[i];
    const float gv = g[i] * gscale + wd * pv;
    const float b = (f & 2) ? momentum * buf[i] + gv : gv;
    buf[i] = b;
    p[i] = pv - lr * b;
  }
}

// Four elements per thread: one table search and 16-byte accesses per group when the group lies inside one tensor (FusedSGD pads its slots to
// 4 floats, so every group does); a group that straddles tensors falls back to the scalar update.  0.16 -> 0.07 ms for 17.1 M parameters.
__global__ void __launch_bounds__(256) sgd4_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                   const int64_t* __restrict__ offsets, const int32_t* __restrict__ flags, int ntensors,
                                                   int64_t total, float lr, float momentum, float wd, float gscale, const float* __restrict__ skip) {
  if (skip && *skip != 0.f) return;
  const int64_t ngroups = (total + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < ngroups; q += (int64_t)gridDim.x * 256) {
    const int64_t i = q << 2;
    int lo = 0, hi = ntensors;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    if (i + 4 <= total && offsets[lo + 1] >= i + 4) {
      const int f = flags[lo];
      if (!(f & 1)) continue;
      float4 pv = *reinterpret_cast<const float4*>(p + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      float4 bv = (f & 2) ? *reinterpret_cast<const float4*>(buf + i) : float4{0.f, 0.f, 0.f, 0.f};
#define SGD1(c_)                                        \
      {                                                 \
        const float gg = gv.c_ * gscale + wd * pv.c_;   \
        const float b = (f & 2) ? momentum * bv.c_ + gg : gg; \
        bv.c_ = b;                                      \
        pv.c_ = pv.c_ - lr * b;                         \
      }
      SGD1(x) SGD1(y) SGD1(z) SGD1(w)
#undef SGD1
      *reinterpret_cast<float4*>(buf + i) = bv;
      *reinterpret_cast<float4*>(p + i) = pv;
    } else {
      for (int64_t e = i; e < i + 4 && e < total; ++e) {
        int t = lo;
        while (t + 1 < ntensors && offsets[t + 1] <= e) ++t;
        const int f = flags[t];
        if (!(f & 1)) continue;
        const float pv = p[e];
        const float gv = g[e] * gscale + wd * pv;
        const float b = (f & 2) ? momentum * buf[e] + gv : gv;
        buf[e] = b;
        p[e] = pv - lr * b;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight packing
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) pack_conv3_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Co, int Ci) {
  const int64_t total = (int64_t)Co * Ci * 27;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // i enumerates the reference layout [co][ci][t]
    const int t = (int)(i % 27);
    const int ci = (int)((i / 27) % Ci);
    const int co = (int)(i / (27 * (int64_t)Ci));
    const T v = from_f<T>(w[i]);
    if (wf) wf[((int64_t)co * 27 + t) * Ci + ci] = v;
    if (wd) wd[((int64_t)ci * 27 + (26 - t)) * Co + co] = v;
  }
}
template <typename T>
__global__ void __launch_bounds__(256) pack_convt_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Ci, int Co) {
  const int64_t total = (int64_t)Ci * Co * 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // reference layout [ci][co][t]
    const int t = (int)(i % 8);
    const int co = (int)((i / 8) % Co);
    const int ci = (int)(i / (8 * (int64_t)Co));
    const T v = from_f<T>(w[i]);
    if (wf) wf[((int64_t)t * Co + co) * Ci + ci] = v;
    if (wd) wd[((int64_t)ci * 8 + t) * Co + co] = v;
  }
}

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int pcrl_bn1d_fwd(const float* x, float* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float momentum, float eps, float* mean, float* rstd, int rows, int C, int relu, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && y && gamma && beta && mean && rstd, "bn1d_fwd: null pointer");
  PCRL_REQUIRE(rows > 1, "bn1d_fwd: Expected more than 1 value per channel when training, got rows=%d", rows);
  hipLaunchKernelGGL(bn1d_fwd_kernel, dim3((C + 3) / 4), dim3(256), 0, as_stream(stream), x, y, gamma, beta, running_mean, running_var,
                     momentum, eps, mean, rstd, rows, C, relu);
  return pcrl_check_launch("bn1d_fwd");
}
extern "C" int pcrl_bn1d_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* mean, const float* rstd,
                             float* dx, float* dgamma, float* dbeta, int rows, int C, int relu, pcrl_stream_t stream) {
  PCRL_REQUIRE(dy && x && y && gamma && mean && rstd && dx && dgamma && dbeta, "bn1d_bwd: null pointer");
  hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((C + 3) / 4), dim3(256), 0, as_stream(stream), dy, x, y, gamma, mean, rstd, dx, dgamma, dbeta, rows, C, relu);
  return pcrl_check_launch("bn1d_bwd");
}
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
extern "C" int pcrl_linear_fwd(const float* x, const float* w, const float* b, float* y, int rows, int Cin, int Cout, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && w && y && rows > 0 && Cin > 0 && Cout > 0, "linear_fwd: bad arguments");
  // y[r][o] = sum_k x[r][k] * w[o][k] + b[o]
  if (Cin % 4 == 0 && al16(x) && al16(w))
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((Cout + 3) / 4, (rows + 31) / 32), dim3(256), 0, as_stream(stream), x, w, b, y, rows, Cin, Cout);
  else
    hipLaunchKernelGGL(sgemm_small_kernel, dim3((Cout + 31) / 32, (rows + 31) / 32), dim3(256), 0, as_stream(stream),
                       x, (int64_t)Cin, (int64_t)1, w, (int64_t)1, (int64_t)Cin, b, y, rows, Cout, Cin);
  return pcrl_check_launch("linear_fwd");
}
extern "C" int pcrl_linear_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                               int rows, int Cin, int Cout, pcrl_stream_t stream) {
  PCRL_REQUIRE(dy && x && w && dx && dw, "linear_bwd: null pointer");
  // dx[r][k] = sum_o dy[r][o] * w[o][k]
  const bool fast = Cin % 4 == 0 && al16(x) && al16(w) && al16(dx) && al16(dw);
  if (fast)
    hipLaunchKernelGGL(linear_dx_kernel, dim3((Cin + 255) / 256, (rows + 3) / 4), dim3(256), 0, as_stream(stream), dy, w, dx, rows, Cin, Cout);
  else
    hipLaunchKernelGGL(sgemm_small_kernel, dim3((Cin + 31) / 32, (rows + 31) / 32), dim3(256), 0, as_stream(stream),
                       dy, (int64_t)Cout, (int64_t)1, w, (int64_t)Cin, (int64_t)1, (const float*)nullptr, dx, rows, Cin, Cout);
  if (int e = pcrl_check_launch("linear_dx")) return e;
  // dw[o][k] = sum_r dy[r][o] * x[r][k]
  if (fast)
    hipLaunchKernelGGL(linear_dw_kernel, dim3((Cin + 255) / 256, (Cout + 3) / 4), dim3(256), 0, as_stream(stream), dy, x, dw, rows, Cin, Cout);
  else
    hipLaunchKernelGGL(sgemm_small_kernel, dim3((Cin + 31) / 32, (Cout + 31) / 32), dim3(256), 0, as_stream(stream),
                       dy, (int64_t)1, (int64_t)Cout, x, (int64_t)Cin, (int64_t)1, (const float*)nullptr, dw, Cout, Cin, rows);
  if (int e = pcrl_check_launch("linear_dw")) return e;
  if (db) {
    hipLaunchKernelGGL(colsum_small_kernel, dim3((Cout + 3) / 4), dim3(256), 0, as_stream(stream), dy, db, rows, Cout);
    return pcrl_check_launch("linear_db");
  }
  return PCRL_OK;
}

extern "C" int pcrl_upsample_trilinear_fwd(const float* x, float* y, int N, int D, int H, int W, int scale, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && y && scale >= 1 && scale <= 4, "upsample_trilinear_fwd: scale must be 1..4 (got %d)", scale);
  const int64_t total = (int64_t)N * D * H * W * scale * scale * scale;
  hipLaunchKernelGGL(tri_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, y, Dims{N, D, H, W}, scale, total);
  return pcrl_check_launch("tri_fwd");
}
extern "C" int pcrl_upsample_trilinear_bwd(const float* dy, float* dx, int N, int D, int H, int W, int scale, pcrl_stream_t stream) {
  PCRL_REQUIRE(dy && dx && scale >= 1 && scale <= 4, "upsample_trilinear_bwd: scale must be 1..4 (got %d)", scale);
  const int64_t total = (int64_t)N * D * H * W;
  const size_t lds = ((size_t)H * scale * W * scale + (size_t)H * W * scale) * sizeof(float);
  if (scale > 1 && lds <= 60 * 1024 && (int64_t)N * D < ((int64_t)1 << 30)) {
    hipLaunchKernelGGL(tri_bwd_planes_kernel, dim3((unsigned)(N * D)), dim3(256), lds, as_stream(stream), dy, dx, Dims{N, D, H, W}, scale);
    return pcrl_check_launch("tri_bwd (planes)");
  }
  hipLaunchKernelGGL(tri_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, dx, Dims{N, D, H, W}, scale, total);
  return pcrl_check_launch("tri_bwd");
}

extern "C" int pcrl_sigmoid_fwd(const float* x, float* y, int64_t n, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && y, "sigmoid_fwd: null pointer");
  hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, y, n);
  return pcrl_check_launch("sigmoid_fwd");
}
extern "C" int pcrl_sigmoid_bwd(const float* dout, const float* out, float* dpre, int64_t n, pcrl_stream_t stream) {
  PCRL_REQUIRE(dout && out && dpre, "sigmoid_bwd: null pointer");
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), dout, out, dpre, n);
  return pcrl_check_launch("sigmoid_bwd");
}

extern "C" size_t pcrl_reduce_ws_bytes(int64_t n) { return (size_t)((n + RED_CHUNK - 1) / RED_CHUNK) * sizeof(double); }

extern "C" int pcrl_mse_fwd(const float* p, const float* gt, float* loss, void* ws, size_t ws_bytes, int64_t n, pcrl_stream_t stream) {
  PCRL_REQUIRE(p && gt && loss && n > 0, "mse_fwd: bad arguments");
  if (!ws || ws_bytes < pcrl_reduce_ws_bytes(n)) return pcrl_fail(PCRL_EWORKSPACE, "mse_fwd: workspace too small");
  const int blocks = (int)((n + RED_CHUNK - 1) / RED_CHUNK);
  hipLaunchKernelGGL(mse_partial_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), p, gt, (double*)ws, n);
  if (int e = pcrl_check_launch("mse_partial")) return e;
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const double*)ws, loss, blocks, 1.0 / (double)n);
  return pcrl_check_launch("mse_finish");
}
extern "C" int pcrl_mse_bwd(const float* p, const float* gt, const float* dloss, float* dp, int64_t n, pcrl_stream_t stream) {
  PCRL_REQUIRE(p && gt && dloss && dp && n > 0, "mse_bwd: bad arguments");
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), p, gt, dloss, dp, n, (float)(2.0 / (double)n));
  return pcrl_check_launch("mse_bwd");
}

extern "C" int pcrl_cosine_mean_fwd(const float* x, const float* y, float* out, float* saved, int rows, int C, float eps, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && y && out && saved && rows > 0 && C > 0, "cosine_mean_fwd: bad arguments");
  hipLaunchKernelGGL(cosine_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), x, y, out, saved, rows, C, eps);
  return pcrl_check_launch("cosine_fwd");
}
extern "C" int pcrl_cosine_mean_bwd(const float* x, const float* y, const float* saved, const float* dout, float* dx,
                                    int rows, int C, float eps, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && y && saved && dout && dx, "cosine_mean_bwd: null pointer");
  hipLaunchKernelGGL(cosine_bwd_kernel, dim3((rows * C + 255) / 256), dim3(256), 0, as_stream(stream), x, y, saved, dout, dx, rows, C, eps);
  return pcrl_check_launch("cosine_bwd");
}

static int fill_cos_terms(CosTerms& t, const void* const* x, const void* const* y, void* const* dx, const float* w, const int* C,
                          const int* group, const int* first, int nterms, int rows, int ngroups, float eps, const char* what) {
  PCRL_REQUIRE(x && y && w && C && group && nterms > 0 && nterms <= COS_MAX_TERMS && rows > 0 && ngroups > 0 && ngroups <= 8,
               "%s: bad arguments (1..%d terms, 1..8 groups)", what, COS_MAX_TERMS);
  for (int k = 0; k < nterms; ++k) {
    PCRL_REQUIRE(x[k] && y[k] && C[k] > 0 && group[k] >= 0 && group[k] < ngroups, "%s: bad term %d", what, k);
    t.x[k] = (const float*)x[k];
    t.y[k] = (const float*)y[k];
    t.dx[k] = dx ? (float*)dx[k] : nullptr;
    t.w[k] = w[k];
    t.C[k] = C[k];
    t.group[k] = group[k];
    t.first[k] = first ? first[k] : 1;
  }
  t.n = nterms;
  t.rows = rows;
  t.ngroups = ngroups;
  t.eps = eps;
  return PCRL_OK;
}
extern "C" size_t pcrl_cosine_terms_ws_bytes(int nterms) { return nterms > 0 ? (size_t)nterms * sizeof(double) : 0; }
extern "C" int pcrl_cosine_terms_fwd(const void* const* x, const void* const* y, const float* w, const int* C, const int* group, int nterms, int rows,
                                     int ngroups, float eps, float* out, void* ws, size_t ws_bytes, pcrl_stream_t stream) {
  CosTerms t;
  if (int e = fill_cos_terms(t, x, y, nullptr, w, C, group, nullptr, nterms, rows, ngroups, eps, "cosine_terms_fwd")) return e;
  PCRL_REQUIRE(out, "cosine_terms_fwd: null output");
  if (!ws || ws_bytes < pcrl_cosine_terms_ws_bytes(nterms)) return pcrl_fail(PCRL_EWORKSPACE, "cosine_terms_fwd: workspace too small");
  hipLaunchKernelGGL(cosine_terms_fwd_kernel, dim3(nterms), dim3(1024), 0, as_stream(stream), t, (double*)ws);
  hipLaunchKernelGGL(cosine_terms_sum_kernel, dim3(1), dim3(64), 0, as_stream(stream), t, (const double*)ws, out);
  return pcrl_check_launch("cosine_terms_fwd");
}
extern "C" int pcrl_cosine_terms_bwd(const void* const* x, const void* const* y, void* const* dx, const float* w, const int* C, const int* group,
                                     const int* first, int nterms, int rows, int ngroups, float eps, const float* dout, pcrl_stream_t stream) {
  CosTerms t;
  PCRL_REQUIRE(dx && first && dout, "cosine_terms_bwd: null pointer");
  if (int e = fill_cos_terms(t, x, y, dx, w, C, group, first, nterms, rows, ngroups, eps, "cosine_terms_bwd")) return e;
  for (int k = 0; k < nterms; ++k) {
    PCRL_REQUIRE(dx[k], "cosine_terms_bwd: term %d has no gradient buffer", k);
    bool seen = false;   // `first` must mark exactly the first term of every buffer: the kernel keys its work on it
    for (int j = 0; j < k; ++j) seen = seen || dx[j] == dx[k];
    PCRL_REQUIRE((first[k] != 0) == !seen, "cosine_terms_bwd: first[%d] does not mark the first term of its gradient buffer", k);
    for (int j = 0; j < k; ++j) PCRL_REQUIRE(dx[j] != dx[k] || C[j] == C[k], "cosine_terms_bwd: terms %d and %d share a buffer but not a width", j, k);
  }
  hipLaunchKernelGGL(cosine_terms_bwd_kernel, dim3((rows + 3) / 4, nterms), dim3(256), 0, as_stream(stream), t, dout);
  return pcrl_check_launch("cosine_terms_bwd");
}

// dst = the concatenation of up to 8 contiguous buffers (torch.cat of the six local views, train_3d.py:121): ONE launch instead of one
// device copy per piece.  Sizes in bytes, multiples of 16; 16-byte vectors.
struct CatParams {
  const uint4* src[8];
  int64_t vec_end[8];   // exclusive end of piece k in 16-byte vectors of dst
  int n;
};
__global__ void __launch_bounds__(256) concat_kernel(const CatParams c, uint4* __restrict__ dst) {
  const int64_t total = c.vec_end[c.n - 1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int k = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q)
      if (q + 1 < c.n && i >= c.vec_end[q]) k = q + 1;
    dst[i] = c.src[k][i - (k ? c.vec_end[k - 1] : 0)];
  }
}
extern "C" int pcrl_concat(const void* const* src, const int64_t* nbytes, int n, void* dst, pcrl_stream_t stream) {
  PCRL_REQUIRE(src && nbytes && dst && n >= 1 && n <= 8, "concat: 1..8 pieces");
  CatParams c;
  int64_t end = 0;
  for (int k = 0; k < 8; ++k) {
    if (k < n) {
      PCRL_REQUIRE(src[k] && nbytes[k] > 0 && nbytes[k] % 16 == 0 && al16(src[k]), "concat: piece %d must be a non-empty 16-byte multiple, 16-byte aligned", k);
      end += nbytes[k] / 16;
    }
    c.src[k] = k < n ? static_cast<const uint4*>(src[k]) : nullptr;
    c.vec_end[k] = end;
  }
  c.n = n;
  PCRL_REQUIRE(al16(dst), "concat: destination not 16-byte aligned");
  hipLaunchKernelGGL(concat_kernel, dim3(grid_for(end)), dim3(256), 0, as_stream(stream), c, static_cast<uint4*>(dst));
  return pcrl_check_launch("concat");
}

// Sum of the gradients the passes of a step produced for each parameter, straight into the optimizer's flat gradient arena (what autograd's
// AccumulateGrad does tensor by tensor behind train_3d.py:149): dst[offsets[t] + e] = ((src[t][0][e] + src[t][1][e]) + ...) for the tensors
// t0 .. t0 + cnt - 1, sources in the order given, null sources skipped (a tensor whose sources are all null is left alone).  The source
// pointers travel as kernel arguments (<= GS_MAX per launch), so nothing is staged through device memory and nothing waits for a copy.
constexpr int GS_MAX = 480;
struct GradSrc {
  const float* s[GS_MAX];
};
__global__ void __launch_bounds__(256) grad_sum_kernel(float* __restrict__ dst, const int64_t* __restrict__ offsets, const int64_t* __restrict__ numels,
                                                       const GradSrc src, int t0, int cnt, int nsrc) {
  const int64_t base = offsets[t0], total = offsets[t0 + cnt] - base;
  const int64_t ngroups = (total + 3) >> 2;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < ngroups; q += (int64_t)gridDim.x * 256) {
    const int64_t i = base + (q << 2);      // slots are padded to 4 floats: a group never straddles two tensors
    int lo = t0, hi = t0 + cnt;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    const int64_t e = i - offsets[lo], n = numels[lo];
    if (e >= n) continue;
    const float* const* sp = src.s + (lo - t0) * nsrc;
    bool vec = e + 4 <= n;
    for (int k = 0; k < nsrc; ++k) vec = vec && ((reinterpret_cast<uintptr_t>(sp[k]) & 15) == 0);   // e.g. a 1-element slice of a (gamma, beta) pair
    if (vec) {
      float4 a{0.f, 0.f, 0.f, 0.f};
      bool any = false;
      for (int k = 0; k < nsrc; ++k) {
        const float* g = sp[k];
        if (!g) continue;
        const float4 v = *reinterpret_cast<const float4*>(g + e);
        if (!any) { a = v; any = true; }
        else { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
      }
      if (any) *reinterpret_cast<float4*>(dst + i) = a;
    } else {
      for (int64_t r = e; r < n && r < e + 4; ++r) {
        float a = 0.f;
        bool any = false;
        for (int k = 0; k < nsrc; ++k) {
          const float* g = sp[k];
          if (!g) continue;
          if (!any) { a = g[r]; any = true; }
          else a += g[r];
        }
        if (any) dst[i + (r - e)] = a;
      }
    }
  }
}
extern "C" int pcrl_grad_sum(float* dst, const int64_t* offsets, const int64_t* numels, const int64_t* offsets_host, const void* const* srcs, int t0, int cnt,
                             int nsrc, pcrl_stream_t stream) {
  PCRL_REQUIRE(dst && offsets && numels && offsets_host && srcs && t0 >= 0 && cnt >= 1 && nsrc >= 1 && nsrc <= 8, "grad_sum: bad arguments");
  PCRL_REQUIRE(al16(dst), "grad_sum: arena not 16-byte aligned");
  const int per = GS_MAX / nsrc;
  for (int c0 = 0; c0 < cnt; c0 += per) {
    const int c = cnt - c0 < per ? cnt - c0 : per;
    GradSrc gs;
    bool any = false;
    for (int k = 0; k < c * nsrc; ++k) {
      gs.s[k] = static_cast<const float*>(srcs[(size_t)c0 * nsrc + k]);
      PCRL_REQUIRE((reinterpret_cast<uintptr_t>(gs.s[k]) & 3) == 0, "grad_sum: source %d not 4-byte aligned", c0 * nsrc + k);
      any = any || gs.s[k];
    }
    if (!any) continue;
    const int64_t elems = offsets_host[t0 + c0 + c] - offsets_host[t0 + c0];
    hipLaunchKernelGGL(grad_sum_kernel, dim3(grid_for((elems + 3) / 4)), dim3(256), 0, as_stream(stream), dst, offsets, numels, gs, t0 + c0, c, nsrc);
    if (int e = pcrl_check_launch("grad_sum")) return e;
  }
  return 0;
}

// total = l1 + l2 + beta * l4 + l5 and scaled = beta * l4 in one launch (train_3d.py:136-138): out[0] = total, out[1] = scaled
__global__ void loss_total_kernel(const float* __restrict__ l1, const float* __restrict__ l2, const float* __restrict__ l4, const float* __restrict__ l5,
                                  float beta, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float s = beta * l4[0];
    out[1] = s;
    out[0] = ((l1[0] + l2[0]) + s) + l5[0];      // the reference's order of additions: loss1 + loss2 + loss4 + local_loss
  }
}
extern "C" int pcrl_loss_total(const float* l1, const float* l2, const float* l4, const float* l5, float beta, float* out, pcrl_stream_t stream) {
  PCRL_REQUIRE(l1 && l2 && l4 && l5 && out, "loss_total: null pointer");
  hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(64), 0, as_stream(stream), l1, l2, l4, l5, beta, out);
  return pcrl_check_launch("loss_total");
}

// backward of pcrl_loss_total: out = (g, beta * g, g, g) -- the gradients of loss1, l4 and of the two cosine groups (global, local)
__global__ void loss_total_bwd_kernel(const float* __restrict__ g, float beta, float* __restrict__ out) {
  if (threadIdx.x < 4) out[threadIdx.x] = threadIdx.x == 1 ? beta * g[0] : g[0];
}
extern "C" int pcrl_loss_total_bwd(const float* g, float beta, float* out, pcrl_stream_t stream) {
  PCRL_REQUIRE(g && out, "loss_total_bwd: null pointer");
  hipLaunchKernelGGL(loss_total_bwd_kernel, dim3(1), dim3(64), 0, as_stream(stream), g, beta, out);
  return pcrl_check_launch("loss_total_bwd");
}

static int sgd_launch(float* p, const float* g, float* buf, const int64_t* offsets, const int32_t* flags, int ntensors, int64_t total, float lr,
                      float momentum, float weight_decay, float grad_scale, const float* skip, pcrl_stream_t stream) {
  PCRL_REQUIRE(p && g && buf && offsets && flags && ntensors > 0 && total > 0, "sgd_step: bad arguments");
  if (al16(p) && al16(g) && al16(buf))
    hipLaunchKernelGGL(sgd4_kernel, dim3(grid_for((total + 3) / 4)), dim3(256), 0, as_stream(stream), p, g, buf, offsets, flags, ntensors, total, lr,
                       momentum, weight_decay, grad_scale, skip);
  else
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), p, g, buf, offsets, flags, ntensors, total, lr, momentum,
                       weight_decay, grad_scale, skip);
  return pcrl_check_launch("sgd");
}

extern "C" int pcrl_sgd_step(float* p, const float* g, float* buf, const int64_t* offsets, const int32_t* flags, int ntensors,
                             int64_t total, float lr, float momentum, float weight_decay, float grad_scale, pcrl_stream_t stream) {
  return sgd_launch(p, g, buf, offsets, flags, ntensors, total, lr, momentum, weight_decay, grad_scale, nullptr, stream);
}

extern "C" int pcrl_sgd_step_guarded(float* p, const float* g, float* buf, const int64_t* offsets, const int32_t* flags, int ntensors,
                                     int64_t total, float lr, float momentum, float weight_decay, float grad_scale, const float* skip,
                                     pcrl_stream_t stream) {
  PCRL_REQUIRE(skip, "sgd_step_guarded: null skip flag");
  return sgd_launch(p, g, buf, offsets, flags, ntensors, total, lr, momentum, weight_decay, grad_scale, skip, stream);
}

// out[0] = (loss > threshold) ? 1 : 0, NaN counts as diverged exactly when the reference's `loss > 1000` would (it would not: NaN > x is false).
__global__ void guard_flag_kernel(const float* __restrict__ loss, float threshold, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (loss[0] > threshold) ? 1.f : 0.f;
}

extern "C" int pcrl_guard_flag(const float* loss, float threshold, float* out, pcrl_stream_t stream) {
  PCRL_REQUIRE(loss && out, "guard_flag: bad arguments");
  hipLaunchKernelGGL(guard_flag_kernel, dim3(1), dim3(64), 0, as_stream(stream), loss, threshold, out);
  return pcrl_check_launch("guard_flag");
}

extern "C" int pcrl_pack_conv3_weight(const float* w_ref, void* w_fwd, void* w_dgrad, int Co, int Ci, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(w_ref && Co > 0 && Ci > 0, "pack_conv3_weight: bad arguments");
  const int64_t total = (int64_t)Co * Ci * 27;
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(pack_conv3_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), w_ref, (bf16*)w_fwd, (bf16*)w_dgrad, Co, Ci);
  else if (dtype == PCRL_F32) hipLaunchKernelGGL(pack_conv3_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), w_ref, (float*)w_fwd, (float*)w_dgrad, Co, Ci);
  else return pcrl_fail(PCRL_EINVAL, "pack_conv3_weight: bad dtype %d", dtype);
  return pcrl_check_launch("pack_conv3");
}
extern "C" int pcrl_pack_convt_weight(const float* w_ref, void* w_fwd, void* w_dgrad, int Ci, int Co, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(w_ref && Co > 0 && Ci > 0, "pack_convt_weight: bad arguments");
  const int64_t total = (int64_t)Co * Ci * 8;
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(pack_convt_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), w_ref, (bf16*)w_fwd, (bf16*)w_dgrad, Ci, Co);
  else if (dtype == PCRL_F32) hipLaunchKernelGGL(pack_convt_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), w_ref, (float*)w_fwd, (float*)w_dgrad, Ci, Co);
  else return pcrl_fail(PCRL_EINVAL, "pack_convt_weight: bad dtype %d", dtype);
  return pcrl_check_launch("pack_convt");
}

// ---- NT-Xent (optional extra) ----
extern "C" size_t pcrl_ntxent_ws_bytes(int R, int C) {
  if (R <= 0 || C <= 0) return 0;
  return ((size_t)2 * R * C + (size_t)2 * R * R + (size_t)2 * R) * sizeof(float);   // zn, dzn | P, Gs | norms, per-row losses
}
extern "C" int pcrl_ntxent_fwd(const float* z, float* loss, void* ws, size_t ws_bytes, int R, int C, float tau, float eps, pcrl_stream_t stream) {
  PCRL_REQUIRE(z && loss && ws, "ntxent_fwd: null pointer");
  PCRL_REQUIRE(R >= 4 && R % 2 == 0 && C > 0 && tau > 0.f, "ntxent_fwd: need an even number >= 4 of rows, C > 0, tau > 0 (R=%d C=%d)", R, C);
  if (ws_bytes < pcrl_ntxent_ws_bytes(R, C)) return pcrl_fail(PCRL_EWORKSPACE, "ntxent_fwd: workspace too small");
  float* zn = (float*)ws;
  float* P = zn + (size_t)2 * R * C;
  float* nrm = P + (size_t)2 * R * R;
  float* li = nrm + R;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(ntx_normalize_kernel, dim3((R + 3) / 4), dim3(256), 0, st, z, zn, nrm, R, C, eps);
  // S[i][k] = sum_c zn[i][c] * (zn[k][c] / tau): fold 1/tau by scaling afterwards is a second pass -- instead scale in the row kernel input
  hipLaunchKernelGGL(sgemm_small_kernel, dim3((R + 31) / 32, (R + 31) / 32), dim3(256), 0, st, zn, (int64_t)C, (int64_t)1, zn, (int64_t)1, (int64_t)C,
                     (const float*)nullptr, P, R, R, C);
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for((int64_t)R * R)), dim3(256), 0, st, P, (int64_t)R * R, 1.f / tau);
  hipLaunchKernelGGL(ntx_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, st, P, li, R);
  hipLaunchKernelGGL(ntx_mean_kernel, dim3(1), dim3(256), 0, st, li, loss, R);
  return pcrl_check_launch("ntxent_fwd");
}
extern "C" int pcrl_ntxent_bwd(const float* z, const float* dloss, float* dz, void* ws, size_t ws_bytes, int R, int C, float tau, float eps,
                               pcrl_stream_t stream) {
  PCRL_REQUIRE(z && dloss && dz && ws, "ntxent_bwd: null pointer");
  PCRL_REQUIRE(R >= 4 && R % 2 == 0 && C > 0 && tau > 0.f, "ntxent_bwd: bad sizes (R=%d C=%d)", R, C);
  if (ws_bytes < pcrl_ntxent_ws_bytes(R, C)) return pcrl_fail(PCRL_EWORKSPACE, "ntxent_bwd: workspace too small");
  float* zn = (float*)ws;             // as left by the forward
  float* dzn = zn + (size_t)R * C;
  float* P = zn + (size_t)2 * R * C;
  float* Gs = P + (size_t)R * R;
  float* nrm = P + (size_t)2 * R * R;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(ntx_gsym_kernel, dim3((R * R + 255) / 256), dim3(256), 0, st, P, dloss, Gs, R, 1.f / tau);
  // dzn[i][c] = sum_k Gs[i][k] * zn[k][c]
  hipLaunchKernelGGL(sgemm_small_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, st, Gs, (int64_t)R, (int64_t)1, zn, (int64_t)C, (int64_t)1,
                     (const float*)nullptr, dzn, R, C, R);
  hipLaunchKernelGGL(ntx_denorm_kernel, dim3((R + 3) / 4), dim3(256), 0, st, z, zn, dzn, nrm, dz, R, C, eps);
  return pcrl_check_launch("ntxent_bwd");
}

