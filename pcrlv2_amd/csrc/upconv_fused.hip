// UpTransition's ConvTranspose3d(k=2, s=2) followed DIRECTLY by ops.0's Conv3d(3x3x3, pad 1) (models/pcrlv2_model_3d.py:52,64 then :9,33:
// `self.ops(self.up_conv(x))`, nothing between the two linear maps) as ONE linear operator on the coarse grid.
//
// The reference materialises the 2x-upsampled tensor `up` (C channels at the fine resolution: 1.07 GB at up_tr64, b = 32) and runs a
// 27-tap convolution over it; 55 % of the model's convolution FLOPs sit in these three layers.  But the 3x3x3 window of a fine
// voxel f = 2v + p (phase p in {0,1}^3) only reaches the 2x2x2 coarse voxels v + p - 1 + q (q in {0,1}^3), so
//     y0[2v + p][co] = bias_class(f)[co] + sum_q sum_ci x[v + p - 1 + q][ci] * Weff[p][q][ci][co]
//     Weff[p][q][ci][co] = sum over the (fine tap t, sub-position s) pairs that lead from q to p   of   sum_cm Wup[ci][cm][s] * W0[co][cm][t]
// -- 8 taps instead of 27 (0.30 of the multiply-adds), no `up`, no gradient of `up`.  Per axis the (t, s) pairs are
//     (p,q) = (0,0): (t=0,s=1)   (0,1): (1,0),(2,1)   (1,0): (0,0),(1,1)   (1,1): (2,0)          [t = 0,1,2 <-> offset -1,0,+1]
// and every (t, s) belongs to exactly one (p, q).  The inner convolution zero-pads `up` (not x): the transposed convolution's bias
// reaches a fine voxel through as many taps as lie inside the fine grid, hence a bias TABLE over the 27 border classes
// (first / inside / last per axis).  Zero-padding x reproduces the zero-padding of `up` for the weight part exactly.
//
// Backward (dy0 = gradient of y0):
//     dx[u][ci]            = sum over the 4x4x4 fine voxels g = 2u - 1 .. 2u + 2 of dy0[g][co] * Weff[p(g)][q(g)][ci][co]
//     dWeff[p][q][ci][co]  = sum_v x[v + p - 1 + q][ci] * dy0[2v + p][co]                       (64 small GEMMs over the coarse voxels)
//     dW0[co][cm][t]       = sum_s  sum_ci dWeff[pq(t,s)][ci][co] * Wup[ci][cm][s]  +  b_up[cm] * B_t[co]
//                            (B_t[co] = sum of dy0[.][co] over the fine voxels from which tap t stays inside the grid: `up` carries b_up)
//     dWup[ci][cm][s]      = sum_t  sum_co dWeff[pq(t,s)][ci][co] * W0[co][cm][t]
//     db_up[cm]            = sum_t sum_co W0[co][cm][t] * B_t[co]
// The three data-sized passes (forward, dx, dWeff) run in the gather implicit-GEMM kernels (conv_igemm.hip GEOM_UPC_*, conv_wgrad.hip
// WG_UPC); the weight-sized algebra runs as MFMA GEMMs with float32 plane-major results (pcrl_gemm_planes_launch) between small
// re-layout kernels in this file.  Parameters, their gradients and the state_dict stay those of the two reference layers.
#include "common.h"

// conv_igemm.hip / conv_wgrad.hip
int pcrl_upc_fwd_launch(const void* x, const void* wf, const float* bias_tab, void* y0, float* stats, int N, int D, int H, int W, int Ci, int Co,
                        int dtype, hipStream_t stream);
int pcrl_upc_dgrad_launch(const void* dy0, const void* wd, void* dx, void* ws, int64_t ws_bytes, int N, int D, int H, int W, int Ci, int Co, int dtype,
                          hipStream_t stream);
int64_t pcrl_upc_dgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Co);   // split-K workspace of the gather form (0: none)
bool pcrl_upc_fwd_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype);   // conv_igemm.hip: a brick kernel takes this shape
int pcrl_upc_fwd_impl(int N, int D, int H, int W, int Ci, int Co, int dtype);           // 0 gather, 1 wide brick (conv_brick16.hip), 2 4 x 8 x 8 brick (conv_brick.hip)
int pcrl_upc_dgrad_impl(int N, int D, int H, int W, int Ci, int Co, int dtype);
int64_t pcrl_brick16_conv_rows(int N, int D, int H, int W);
int64_t pcrl_brick_conv_rows(int N, int D, int H, int W);
int pcrl_brick8_upc_fwd_launch(const void* x, const void* w3, const float* bias_tab, void* y0, float* stats, int N, int D, int H, int W, int Ci, int Co,
                               hipStream_t stream);
int pcrl_brick8_upc_dgrad_launch(const void* dy0, const void* wd3, void* dx, int N, int D, int H, int W, int Ci, int Co, hipStream_t stream);
int pcrl_brick16_upc_fwd_launch(const void* x, const void* w3, const float* bias_tab, void* y0, float* stats, int N, int D, int H, int W, int Ci, int Co,
                                hipStream_t stream);
bool pcrl_upc_dgrad_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype);
int pcrl_brick16_upc_dgrad_launch(const void* dy0, const void* wd3, void* dx, int N, int D, int H, int W, int Ci, int Co, hipStream_t stream);
int pcrl_gemm_planes_launch(const void* a, const void* b, float* z, int64_t M, int K, int Nc, int dtype, hipStream_t stream);
size_t pcrl_upc_wgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Co);
int pcrl_upc_wgrad_launch(const void* dy0, const void* x, float* dweff, void* ws, size_t ws_bytes, int N, int D, int H, int W, int Ci, int Co,
                          int dtype, hipStream_t stream, bool accumulate);
bool pcrl_upc_wgrad_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype);
size_t pcrl_upc_wgrad3_ws_bytes(int N, int D, int H, int W, int Ci, int Co);
int pcrl_upc_wgrad3_launch(const void* dy0, const void* x, float* dweff, void* ws, size_t ws_bytes, int N, int D, int H, int W, int Ci, int Co,
                           hipStream_t stream, bool accumulate);

namespace {

// per axis: the k-th (t, s) pair of (p, q); returns the number of pairs (1 or 2)
__host__ __device__ __forceinline__ int ts_axis(int p, int q, int k, int& t, int& s) {
  if (p == 0 && q == 0) { t = 0; s = 1; return 1; }
  if (p == 0 && q == 1) { t = 1 + k; s = k; return 2; }          // (1,0), (2,1)
  if (p == 1 && q == 0) { t = k; s = k; return 2; }              // (0,0), (1,1)
  t = 2; s = 0; return 1;
}
// per axis: (t, s) -> (p, q)
__host__ __device__ __forceinline__ void pq_axis(int t, int s, int& p, int& q) {
  if (s == 1) { p = (t == 1) ? 1 : 0; q = (t == 0) ? 0 : (t == 1 ? 0 : 1); }   // (0,1)->(0,0)  (1,1)->(1,0)  (2,1)->(0,1)
  else { p = (t == 1) ? 0 : 1; q = (t == 0) ? 0 : 1; }                          // (0,0)->(1,0)  (1,0)->(0,1)  (2,0)->(1,1)
}
// per axis: fine offset e = 0..3 (voxel 2u - 1 + e) -> (p, q)
__host__ __device__ __forceinline__ void pq_of_e(int e, int& p, int& q) {
  p = (e & 1) ? 0 : 1;            // e = 0 -> (1,1), 1 -> (0,1), 2 -> (1,0), 3 -> (0,0)
  q = (e < 2) ? 1 : 0;
}
// tap t (0..2 on one axis) lies inside the fine grid for a voxel of border class c (0 first, 1 inside, 2 last)
__host__ __device__ __forceinline__ bool tap_in(int t, int c) { return !(t == 0 && c == 0) && !(t == 2 && c == 2); }

template <typename T> __device__ __forceinline__ T cvt(float v) { return from_f<T>(v); }

// a0[(t*Co+co)][cm] = w0[co][cm][t];  bu[(s*Ci+ci)][cm] = wup[ci][cm][s];  b1[cm][(s*Ci+ci)] = wup[ci][cm][s];  b2[cm][(t*Co+co)] = w0[co][cm][t]
// (any of the four outputs may be null)
template <typename T>
__global__ void __launch_bounds__(256) upc_prep_kernel(const float* __restrict__ wup, const float* __restrict__ w0, T* __restrict__ a0,
                                                       T* __restrict__ bu, T* __restrict__ b1, T* __restrict__ b2, int Ci, int Cm, int Co) {
  const int64_t n0 = (int64_t)27 * Co * Cm, nu = (int64_t)8 * Ci * Cm;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n0 + nu; i += (int64_t)gridDim.x * 256) {
    if (i < n0) {                       // i enumerates w0 in its own order [co][cm][t]: coalesced read
      const int t = (int)(i % 27), cm = (int)((i / 27) % Cm), co = (int)(i / ((int64_t)27 * Cm));
      const T v = cvt<T>(w0[i]);
      if (a0) a0[((int64_t)t * Co + co) * Cm + cm] = v;
      if (b2) b2[(int64_t)cm * (27 * Co) + t * Co + co] = v;
    } else {
      const int64_t k = i - n0;         // [ci][cm][s]
      const int s = (int)(k % 8), cm = (int)((k / 8) % Cm), ci = (int)(k / ((int64_t)8 * Cm));
      const T v = cvt<T>(wup[k]);
      if (bu) bu[((int64_t)s * Ci + ci) * Cm + cm] = v;
      if (b1) b1[(int64_t)cm * (8 * Ci) + s * Ci + ci] = v;
    }
  }
}

// P[(s*Ci+ci)][(t*Co+co)] (float32, row length M27 = 27*Co) -> Weff, stored as
//   wd[(ci*64 + e)*Co + co]            e = the fine offset triple of (p, q)        (threads: co fastest -> coalesced both ways)
//   wf[((p*Co + co)*8 + q)*Ci + ci]
//   w3[((p*Co + co)*27 + tap)*Ci + ci]  tap = (p + q) per axis: the zero-embedded 3x3x3 form the brick kernels read (optional; pre-zeroed)
template <typename T>
__global__ void __launch_bounds__(256) upc_pack_kernel(const float* __restrict__ P, T* __restrict__ wf, T* __restrict__ wd, T* __restrict__ w3, T* __restrict__ wd3,
                                                       int Ci, int Co) {
  const int64_t M27 = (int64_t)27 * Co;
  const int64_t total = (int64_t)64 * Ci * Co;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int co = (int)(i % Co), e = (int)((i / Co) % 64), ci = (int)(i / ((int64_t)64 * Co));
    int pd, qd, ph, qh, pw, qw;
    pq_of_e(e >> 4, pd, qd);
    pq_of_e((e >> 2) & 3, ph, qh);
    pq_of_e(e & 3, pw, qw);
    float acc = 0.f;
    int td, sd, th, sh, tw, sw;
    const int nd = ts_axis(pd, qd, 0, td, sd), nh = ts_axis(ph, qh, 0, th, sh), nw = ts_axis(pw, qw, 0, tw, sw);
    for (int a = 0; a < nd; ++a) {
      ts_axis(pd, qd, a, td, sd);
      for (int b = 0; b < nh; ++b) {
        ts_axis(ph, qh, b, th, sh);
        for (int c = 0; c < nw; ++c) {
          ts_axis(pw, qw, c, tw, sw);
          const int t = td * 9 + th * 3 + tw, s = sd * 4 + sh * 2 + sw;
          acc += P[((int64_t)s * Ci + ci) * M27 + (int64_t)t * Co + co];
        }
      }
    }
    const T v = cvt<T>(acc);
    wd[i] = v;
    const int p = pd * 4 + ph * 2 + pw, q = qd * 4 + qh * 2 + qw;
    wf[(((int64_t)p * Co + co) * 8 + q) * Ci + ci] = v;
    if (w3) w3[(((int64_t)p * Co + co) * 27 + (pd + qd) * 9 + (ph + qh) * 3 + (pw + qw)) * Ci + ci] = v;
    if (wd3) {   // data-gradient tap e = 2 k - 1 + par per axis: k = (e + 1) >> 1 on the coarse grid, parity (e + 1) & 1 of the fine voxel
      const int ed = e >> 4, eh = (e >> 2) & 3, ew = e & 3;
      const int k = ((ed + 1) >> 1) * 9 + ((eh + 1) >> 1) * 3 + ((ew + 1) >> 1), par = (((ed + 1) & 1) << 2) | (((eh + 1) & 1) << 1) | ((ew + 1) & 1);
      wd3[(((int64_t)ci * 27 + k) * 8 + par) * Co + co] = v;
    }
  }
}

// The same for Ci % 32 == 0 and Co % 32 == 0 (every layer of the model), tiled: block = (32 ci, one e, 32 co).  The two forms with co innermost
// (wd, wd3) are written straight from the co-fastest thread order the sums are read in; the two with ci innermost (wf, w3 -- the forward
// kernels' K-contiguous rows) go through a 32 x 32 LDS tile and leave as 64-byte runs.  (The untiled kernel above writes them as single two-byte
// stores 8 Ci / 27 Ci elements apart: 103 us for up_tr256's 8.4 M composed weights, 79 us at up_tr128 -- most of pcrl_upconv_compose.)
template <typename T>
__global__ void __launch_bounds__(256) upc_pack_tiled_kernel(const float* __restrict__ P, T* __restrict__ wf, T* __restrict__ wd, T* __restrict__ w3,
                                                             T* __restrict__ wd3, int Ci, int Co) {
  __shared__ T tile[32][33];
  const int64_t M27 = (int64_t)27 * Co;
  const int e = blockIdx.y, ci0 = blockIdx.x * 32, co0 = blockIdx.z * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  int pd, qd, ph, qh, pw, qw;
  pq_of_e(e >> 4, pd, qd);
  pq_of_e((e >> 2) & 3, ph, qh);
  pq_of_e(e & 3, pw, qw);
  int td, sd, th, sh, tw, sw;
  const int nd = ts_axis(pd, qd, 0, td, sd), nh = ts_axis(ph, qh, 0, th, sh), nw = ts_axis(pw, qw, 0, tw, sw);
  const int p = pd * 4 + ph * 2 + pw, q = qd * 4 + qh * 2 + qw;
  const int ed = e >> 4, eh = (e >> 2) & 3, ew = e & 3;
  const int k3 = ((ed + 1) >> 1) * 9 + ((eh + 1) >> 1) * 3 + ((ew + 1) >> 1), par = (((ed + 1) & 1) << 2) | (((eh + 1) & 1) << 1) | ((ew + 1) & 1);
  const int co = co0 + tx;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ci = ci0 + ty + 8 * r;
    float acc = 0.f;
    for (int a = 0; a < nd; ++a) {      // the same summation order as upc_pack_kernel: bit-identical weights
      ts_axis(pd, qd, a, td, sd);
      for (int b = 0; b < nh; ++b) {
        ts_axis(ph, qh, b, th, sh);
        for (int c = 0; c < nw; ++c) {
          ts_axis(pw, qw, c, tw, sw);
          const int t = td * 9 + th * 3 + tw, s_ = sd * 4 + sh * 2 + sw;
          acc += P[((int64_t)s_ * Ci + ci) * M27 + (int64_t)t * Co + co];
        }
      }
    }
    const T v = cvt<T>(acc);
    wd[((int64_t)ci * 64 + e) * Co + co] = v;
    if (wd3) wd3[(((int64_t)ci * 27 + k3) * 8 + par) * Co + co] = v;
    tile[ty + 8 * r][tx] = v;            // [ci][co]
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co2 = co0 + ty + 8 * r;   // thread = (co2 row, ci = tx): 32 consecutive ci per row
    const T v = tile[tx][ty + 8 * r];
    wf[(((int64_t)p * Co + co2) * 8 + q) * Ci + ci0 + tx] = v;
    if (w3) w3[(((int64_t)p * Co + co2) * 27 + (pd + qd) * 9 + (ph + qh) * 3 + (pw + qw)) * Ci + ci0 + tx] = v;
  }
}

// bias_tab[cls][co] = b0[co] + sum over the taps t inside the grid for class cls of wb[co][t],  wb[co][t] = sum_cm w0[co][cm][t] * b_up[cm].
// Block = one co: lane t < 27 of the eight 32-lane groups walks cm (the 27 taps of a (co, cm) are contiguous: coalesced), LDS combine.
__global__ void __launch_bounds__(256) upc_bias_kernel(const float* __restrict__ w0, const float* __restrict__ b_up, const float* __restrict__ b0,
                                                       float* __restrict__ tab, int Cm, int Co) {
  __shared__ double part[8][32];
  __shared__ double wb[27];
  const int co = blockIdx.x, t = threadIdx.x & 31, g = threadIdx.x >> 5;
  double a = 0.0;
  if (t < 27)
    for (int cm = g; cm < Cm; cm += 8) a += (double)w0[((int64_t)co * Cm + cm) * 27 + t] * (double)b_up[cm];
  part[g][t] = a;
  __syncthreads();
  if (threadIdx.x < 27) {
    double v = 0.0;
    for (int k = 0; k < 8; ++k) v += part[k][threadIdx.x];
    wb[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    const int cls = threadIdx.x, cd = cls / 9, ch = (cls / 3) % 3, cw = cls % 3;
    double acc = b0 ? (double)b0[co] : 0.0;
    for (int tt = 0; tt < 27; ++tt)
      if (tap_in(tt / 9, cd) && tap_in((tt / 3) % 3, ch) && tap_in(tt % 3, cw)) acc += wb[tt];
    tab[cls * Co + co] = (float)acc;
  }
}

// dweff[co][ci][pq] (float32) -> a1[(t*Co+co)][(s*Ci+ci)], value dWeff[pq(t,s)][ci][co]   (a2 = a1 transposed: transpose_kernel)
// Block = (co, 32 consecutive ci): the [32 ci][64 pq] slab of dweff (8 KB, contiguous) goes through LDS once, then every one of the 216
// (t, s) rows of a1 gets its 32 consecutive ci as one 64 / 128-byte run.  (The first version read dweff at a stride of 64 floats per
// thread -- one element per 256-byte line: 250 us for up_tr256's 28 M outputs; this one moves 33 MB in and 57 MB out in ~25 us.)
template <typename T>
__global__ void __launch_bounds__(256) upc_chain_pack_kernel(const float* __restrict__ dweff, T* __restrict__ a1, int Ci, int Co) {
  __shared__ float slab[32][65];
  const int co = blockIdx.y, ci0 = blockIdx.x * 32, tid = threadIdx.x;
  const float* src = dweff + ((int64_t)co * Ci + ci0) * 64;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = (k * 256 + tid) * 4;      // float index into the 2048-float slab
    const float4 v = *reinterpret_cast<const float4*>(src + e);
    const int r = e >> 6, c = e & 63;
    slab[r][c] = v.x; slab[r][c + 1] = v.y; slab[r][c + 2] = v.z; slab[r][c + 3] = v.w;
  }
  __syncthreads();
  const int ci = tid & 31;
  for (int ts = tid >> 5; ts < 216; ts += 8) {
    const int t = ts >> 3, sidx = ts & 7;
    int pd, qd, ph, qh, pw, qw;
    pq_axis(t / 9, sidx >> 2, pd, qd);
    pq_axis((t / 3) % 3, (sidx >> 1) & 1, ph, qh);
    pq_axis(t % 3, sidx & 1, pw, qw);
    const int pq = (pd * 4 + ph * 2 + pw) * 8 + (qd * 4 + qh * 2 + qw);
    a1[((int64_t)t * Co + co) * (8 * (int64_t)Ci) + (int64_t)sidx * Ci + ci0 + ci] = cvt<T>(slab[ci][pq]);
  }
}
// out[c][r] = in[r][c] for an R x C matrix (both multiples of 32), 32 x 32 tiles through LDS: both sides coalesced (the pack kernel writing
// the transposed copy itself strode 2-byte stores by 27*Co elements: 228 us at up_tr256)
template <typename T>
__global__ void __launch_bounds__(256) transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int R, int C) {
  __shared__ T tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) tile[ty + 8 * k][tx] = in[(int64_t)(r0 + ty + 8 * k) * C + c0 + tx];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) out[(int64_t)(c0 + ty + 8 * k) * R + r0 + tx] = tile[tx][ty + 8 * k];
}

// z1[cm][(t*Co+co)] -> dw0[co][cm][t];   z2[cm][(s*Ci+ci)] -> dwup[ci][cm][s]
__global__ void __launch_bounds__(256) upc_chain_unpack_kernel(const float* __restrict__ z1, const float* __restrict__ z2, const float* __restrict__ b_up,
                                                               const float* __restrict__ box, float* __restrict__ dw0,
                                                               float* __restrict__ dwup, int Ci, int Cm, int Co) {
  const int64_t n0 = (int64_t)27 * Co * Cm, nu = (int64_t)8 * Ci * Cm;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n0 + nu; i += (int64_t)gridDim.x * 256) {
    if (i < n0) {
      const int t = (int)(i % 27), cm = (int)((i / 27) % Cm), co = (int)(i / ((int64_t)27 * Cm));
      dw0[i] = z1[(int64_t)cm * (27 * Co) + t * Co + co] + b_up[cm] * box[t * Co + co];
    } else {
      const int64_t k = i - n0;
      const int s = (int)(k % 8), cm = (int)((k / 8) % Cm), ci = (int)(k / ((int64_t)8 * Cm));
      dwup[k] = z2[(int64_t)cm * (8 * Ci) + s * Ci + ci];
    }
  }
}

// Border-class sums of dy0 for the gradient of the transposed convolution's bias.  Block = one (n, fine d) plane of dy0 [FH][FW][C];
// thread = (channel vector, w slot).  The class of a voxel is fixed by the loop it is visited in -- rows h = 0 / inside / FH - 1, and inside
// a row the columns w = 0 / inside / FW - 1 -- so the streaming loop is one 16-byte load and VEC adds (a per-element class select over
// nine accumulators made the first version VALU-bound: 183 us for 268 MB).  part[plane][9][C], k = h class * 3 + w class.
template <typename T>
__global__ void __launch_bounds__(256) upc_class_sums_kernel(const T* __restrict__ dy, float* __restrict__ part, int FD, int FH, int FW, int C, bool zsum) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [slots][C]
  const int tid = threadIdx.x, nvec = C / VEC, cv = tid % nvec, slot = tid / nvec, nslots = 256 / nvec;
  // Plane order: the 2 N border planes (fd = 0, FD - 1: every voxel of them is read) take the FIRST block ids, the planes inside the volume (a ninth of the
  // work each under zsum) follow -- in plane order the last border plane started in the last round of blocks and alone set the kernel's duration.
  const int nsamp = gridDim.x / FD;
  int plane;
  if ((int)blockIdx.x < 2 * nsamp) plane = (blockIdx.x >> 1) * FD + ((blockIdx.x & 1) ? FD - 1 : 0);
  else {
    const int kk = blockIdx.x - 2 * nsamp;
    plane = (kk / (FD - 2)) * FD + 1 + kk % (FD - 2);
  }
  const T* base = dy + (int64_t)plane * FH * FW * C + cv * VEC;
  float acc[9][VEC];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[k][j] = 0.f;
  // a row (or a run of rows) of one h class: columns 1 .. FW - 2 over the slots, the two border columns by slot 0
  /* four rows in flight per thread (eight measured no better): the two border planes of a sample read every voxel -- 62 rows of ONE load each were a chain of 62   */
  /* memory latencies per thread, and those 2 N blocks set the kernel's duration (110-130 us at up_tr64 for 12 % of a 537 MB tensor).     */
#define ROW1(h_, K_)                                                                            \
  {                                                                                             \
    const T* row = base + (int64_t)(h_) * FW * C;                                               \
    for (int w = 1 + slot; w < FW - 1; w += nslots) {                                           \
      const Vec16<T> v = ld16(row + (int64_t)w * C);                                            \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[(K_)*3 + 1][j] += to_f(v.v[j]);       \
    }                                                                                           \
    if (slot == 0) {                                                                            \
      const Vec16<T> v0 = ld16(row), v1 = ld16(row + (int64_t)(FW - 1) * C);                    \
      _Pragma("unroll") for (int j = 0; j < VEC; ++j) {                                         \
        acc[(K_)*3 + 0][j] += to_f(v0.v[j]);                                                    \
        acc[(K_)*3 + 2][j] += to_f(v1.v[j]);                                                    \
      }                                                                                         \
    }                                                                                           \
  }
#define ROWS(h0_, h1_, K_)                                                                      \
  {                                                                                             \
    int h = (h0_);                                                                              \
    for (; h + 3 < (h1_); h += 4) {                                                             \
      const T* row = base + (int64_t)h * FW * C;                                                \
      for (int w = 1 + slot; w < FW - 1; w += nslots) {                                         \
        Vec16<T> v[4];                                                                          \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) v[u] = ld16(row + ((int64_t)u * FW + w) * C); \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                         \
          _Pragma("unroll") for (int j = 0; j < VEC; ++j) acc[(K_)*3 + 1][j] += to_f(v[u].v[j]); \
        }                                                                                       \
      }                                                                                         \
      if (slot == 0) {                                                                          \
        Vec16<T> v0[4], v1[4];                                                                  \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                         \
          v0[u] = ld16(row + (int64_t)u * FW * C);                                              \
          v1[u] = ld16(row + ((int64_t)u * FW + FW - 1) * C);                                   \
        }                                                                                       \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                         \
          _Pragma("unroll") for (int j = 0; j < VEC; ++j) {                                     \
            acc[(K_)*3 + 0][j] += to_f(v0[u].v[j]);                                             \
            acc[(K_)*3 + 2][j] += to_f(v1[u].v[j]);                                             \
          }                                                                                     \
        }                                                                                       \
      }                                                                                         \
    }                                                                                           \
    for (; h < (h1_); ++h) ROW1(h, K_)                                                          \
  }
  // zsum: dy sums to zero over all voxels per channel (it is the output of a training-mode BatchNorm backward over exactly these voxels), so the
  // fully interior class is minus the sum of the other 26 (upc_box_kernel) and a plane inside the volume only contributes its border: two
  // rows and two columns instead of FH x FW voxels (9 % at 64 x 32) -- and the result is the exact-arithmetic one, free of the rounding of dy.
  const int fd = plane % FD;
  if (zsum && fd > 0 && fd < FD - 1) {
    ROWS(0, 1, 0)
    ROWS(FH - 1, FH, 2)
    for (int h = 1 + slot; h < FH - 1; h += nslots) {
      const T* row = base + (int64_t)h * FW * C;
      const Vec16<T> v0 = ld16(row), v1 = ld16(row + (int64_t)(FW - 1) * C);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        acc[3][j] += to_f(v0.v[j]);
        acc[5][j] += to_f(v1.v[j]);
      }
    }
  } else {
    ROWS(0, 1, 0)
    ROWS(1, FH - 1, 1)
    ROWS(FH - 1, FH, 2)
  }
#undef ROWS
#undef ROW1
  if (nvec <= 64) {
    // The slots of a wave are lanes nvec apart: a fixed xor tree inside the wave, then the four waves through LDS -- one barrier instead of
    // the eighteen of the class-by-class reduction below (which was most of this kernel: a plane inside the volume reads 9 % of its voxels).
    const int wv = tid >> 6, ln = tid & 63;
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float a = acc[q][j];
        for (int o = nvec; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
        if (ln < nvec) sm[(wv * 9 + q) * C + cv * VEC + j] = a;
      }
    __syncthreads();
    for (int i = tid; i < 9 * C; i += 256) {
      const int q = i / C, c = i - q * C;
      part[((int64_t)plane * 9 + q) * C + c] = (sm[q * C + c] + sm[(9 + q) * C + c]) + (sm[(18 + q) * C + c] + sm[(27 + q) * C + c]);
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) sm[slot * C + cv * VEC + j] = acc[q][j];
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      float a = 0.f;
      for (int s2 = 0; s2 < nslots; ++s2) a += sm[s2 * C + c];
      part[((int64_t)plane * 9 + q) * C + c] = a;
    }
  }
}
// S[cls][co] = sum over the planes of class cd of part[plane][ch*3+cw][co]   (planes = N * FD, plane = n * FD + fd).
// Block = (k9, 64 channels) x cd: 16 groups of 64 lanes walk the class's planes four at a time (the plane count reaches thousands: a
// single chain of dependent loads took 320 us), combined in group order through LDS.
// blockIdx.z = chunk of the class's planes (UPC_TOTAL_CHUNKS of them; upc_box_kernel adds the chunks in order): 27 blocks walked up to
// 1 984 planes each (51 us at up_tr64).
constexpr int UPC_TOTAL_CHUNKS = 8;
__global__ void __launch_bounds__(1024) upc_class_total_kernel(const float* __restrict__ part, float* __restrict__ S, int N, int FD, int C) {
  __shared__ double red[16][64];
  const int cb = C / 64 > 0 ? (C + 63) / 64 : 1;
  const int k9 = blockIdx.x / cb, co = (blockIdx.x % cb) * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6, cd = blockIdx.y;
  const int f0 = cd == 0 ? 0 : (cd == 2 ? FD - 1 : 1), f1 = cd == 0 ? 1 : (cd == 2 ? FD : FD - 1);
  const int nf = f1 > f0 ? f1 - f0 : 0;
  const int64_t all = (int64_t)N * nf;
  const int64_t jb = all * blockIdx.z / UPC_TOTAL_CHUNKS, cnt = all * (blockIdx.z + 1) / UPC_TOTAL_CHUNKS;
  S += (int64_t)blockIdx.z * 27 * C;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (co < C) {
    auto at = [&](int64_t j) { return (double)part[((((j / nf) * FD) + f0 + (j % nf)) * 9 + k9) * C + co]; };
    int64_t j = jb + g;
    for (; j + 48 < cnt; j += 64) { a0 += at(j); a1 += at(j + 16); a2 += at(j + 32); a3 += at(j + 48); }
    for (; j < cnt; j += 16) a0 += at(j);
  }
  red[g][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (g == 0 && co < C) {
    double v = 0.0;
    for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x];
    S[(cd * 9 + k9) * C + co] = (float)v;
  }
}
// S[i] = sum over the plane chunks of Sk[k][i], in chunk order (i < 27 * C)
__global__ void __launch_bounds__(256) upc_chunk_sum_kernel(const float* __restrict__ Sk, float* __restrict__ S, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
#pragma unroll
  for (int k = 0; k < UPC_TOTAL_CHUNKS; ++k) v += (double)Sk[(int64_t)k * n + i];
  S[i] = (float)v;
}
// box[t][co] = sum over the classes for which tap t is inside the grid of S[cls][co]
__global__ void __launch_bounds__(256) upc_box_kernel(const float* __restrict__ S, float* __restrict__ box, int Co, bool accumulate, bool zsum) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 27 * Co) return;
  const int co = i % Co, t = i / Co;
  double inner = (double)S[13 * Co + co];
  if (zsum) {   // the fully interior class from the zero total (its entry of S is zero then: the planes inside the volume skipped it)
    inner = 0.0;
    for (int cls = 0; cls < 27; ++cls)
      if (cls != 13) inner -= (double)S[cls * Co + co];
  }
  double st = 0.0;
  for (int cls = 0; cls < 27; ++cls)
    if (tap_in(t / 9, cls / 9) && tap_in((t / 3) % 3, (cls / 3) % 3) && tap_in(t % 3, cls % 3)) st += cls == 13 ? inner : (double)S[cls * Co + co];
  box[i] = accumulate ? box[i] + (float)st : (float)st;
}
// db_up[cm] = sum_t sum_co w0[co][cm][t] * box[t][co];  block = cm
__global__ void __launch_bounds__(256) upc_dbup_kernel(const float* __restrict__ w0, const float* __restrict__ box, float* __restrict__ db_up, int Cm,
                                                       int Co) {
  __shared__ double red[4];
  const int cm = blockIdx.x;
  double a = 0.0;
  for (int i = threadIdx.x; i < 27 * Co; i += 256) {
    const int t = i % 27, co = i / 27;
    a += (double)w0[((int64_t)co * Cm + cm) * 27 + t] * (double)box[t * Co + co];
  }
  a = block_sum_256(a, red);
  if (threadIdx.x == 0) db_up[cm] = (float)a;
}

inline size_t al(size_t n) { return (n + 255) & ~(size_t)255; }
inline int esz(int dtype) { return dtype == PCRL_BF16 ? 2 : 4; }
inline unsigned blocks_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}
int check_upc(const char* what, int N, int D, int H, int W, int Ci, int Cm, int Co, int dtype) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return pcrl_fail(PCRL_EINVAL, "%s: bad dims %d %d %d %d", what, N, D, H, W);
  if (Ci <= 0 || Cm <= 0 || Co <= 0 || Ci % 32 || Cm % 32 || Co % 32) return pcrl_fail(PCRL_EINVAL, "%s: channels must be positive multiples of 32 (%d %d %d)", what, Ci, Cm, Co);
  if (dtype != PCRL_BF16 && dtype != PCRL_F32) return pcrl_fail(PCRL_EINVAL, "%s: bad dtype %d", what, dtype);
  if ((int64_t)N * D * H * W * 8 >= ((int64_t)1 << 31)) return pcrl_fail(PCRL_EINVAL, "%s: volume too large", what);
  return PCRL_OK;
}

}  // namespace

// ---- composed weights (once per optimizer step) ----
extern "C" size_t pcrl_upconv_compose_ws_bytes(int Ci, int Cm, int Co, int dtype) {
  if (Ci <= 0 || Cm <= 0 || Co <= 0) return 0;
  return al((size_t)27 * Co * Cm * esz(dtype)) + al((size_t)8 * Ci * Cm * esz(dtype)) + al((size_t)27 * Co * 8 * Ci * sizeof(float));
}
extern "C" int pcrl_upconv_compose(const float* w_up, const float* b_up, const float* w0, const float* b0, void* wf, void* wd, void* w3f,
                                   void* wd3, float* bias_tab, void* ws, size_t ws_bytes, int Ci, int Cm, int Co, int dtype, pcrl_stream_t stream) {
  if (int e = check_upc("upconv_compose", 1, 1, 1, 1, Ci, Cm, Co, dtype)) return e;
  PCRL_REQUIRE(w_up && b_up && w0 && wf && wd && bias_tab, "upconv_compose: null pointer");
  if (!ws || ws_bytes < pcrl_upconv_compose_ws_bytes(Ci, Cm, Co, dtype)) return pcrl_fail(PCRL_EWORKSPACE, "upconv_compose: workspace too small");
  hipStream_t st = as_stream(stream);
  char* a0 = (char*)ws;
  char* bu = a0 + al((size_t)27 * Co * Cm * esz(dtype));
  float* P = (float*)(bu + al((size_t)8 * Ci * Cm * esz(dtype)));
  const unsigned gb = blocks_for((int64_t)27 * Co * Cm + (int64_t)8 * Ci * Cm);
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(upc_prep_kernel<bf16>, dim3(gb), dim3(256), 0, st, w_up, w0, (bf16*)a0, (bf16*)bu, (bf16*)nullptr, (bf16*)nullptr, Ci, Cm, Co);
  else hipLaunchKernelGGL(upc_prep_kernel<float>, dim3(gb), dim3(256), 0, st, w_up, w0, (float*)a0, (float*)bu, (float*)nullptr, (float*)nullptr, Ci, Cm, Co);
  if (int e = pcrl_check_launch("upconv_compose (prep)")) return e;
  // P[(s,ci)][(t,co)] = sum_cm w0[co][cm][t] * wup[ci][cm][s]
  if (int e = pcrl_gemm_planes_launch(a0, bu, P, (int64_t)27 * Co, Cm, 8 * Ci, dtype, st)) return e;
  const unsigned gp = blocks_for((int64_t)64 * Ci * Co);
  if (w3f) (void)hipMemsetAsync(w3f, 0, (size_t)216 * Ci * Co * esz(dtype), st);   // 19 of a phase's 27 taps stay zero
  if (wd3) (void)hipMemsetAsync(wd3, 0, (size_t)216 * Ci * Co * esz(dtype), st);   // a parity holds 8 of the 27 taps
  static const bool tiled_on = [] { const char* e = getenv("PCRL_UPC_PACK_TILED"); return !(e && e[0] == '0'); }();   // A/B switch (bit-identical outputs)
  if (tiled_on && Ci % 32 == 0 && Co % 32 == 0 && Co / 32 <= 65535) {
    const dim3 gt((unsigned)(Ci / 32), 64, (unsigned)(Co / 32));
    if (dtype == PCRL_BF16) hipLaunchKernelGGL(upc_pack_tiled_kernel<bf16>, gt, dim3(256), 0, st, (const float*)P, (bf16*)wf, (bf16*)wd, (bf16*)w3f, (bf16*)wd3, Ci, Co);
    else hipLaunchKernelGGL(upc_pack_tiled_kernel<float>, gt, dim3(256), 0, st, (const float*)P, (float*)wf, (float*)wd, (float*)w3f, (float*)wd3, Ci, Co);
  } else if (dtype == PCRL_BF16) hipLaunchKernelGGL(upc_pack_kernel<bf16>, dim3(gp), dim3(256), 0, st, (const float*)P, (bf16*)wf, (bf16*)wd, (bf16*)w3f, (bf16*)wd3, Ci, Co);
  else hipLaunchKernelGGL(upc_pack_kernel<float>, dim3(gp), dim3(256), 0, st, (const float*)P, (float*)wf, (float*)wd, (float*)w3f, (float*)wd3, Ci, Co);
  if (int e = pcrl_check_launch("upconv_compose (pack)")) return e;
  hipLaunchKernelGGL(upc_bias_kernel, dim3(Co), dim3(256), 0, st, w0, b_up, b0, bias_tab, Cm, Co);
  return pcrl_check_launch("upconv_compose (bias)");
}

// ---- forward / data gradient ----
extern "C" int64_t pcrl_upconv_fwd_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return pcrl_upc_fwd_uses_brick(N, D, H, W, Ci, Co, dtype) ? 1 : 0;
}
extern "C" int64_t pcrl_upconv_stats_rows(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  const int impl = pcrl_upc_fwd_impl(N, D, H, W, Ci, Co, dtype);
  if (impl == 1) return 8 * pcrl_brick16_conv_rows(N, D, H, W);
  if (impl == 2) return 8 * pcrl_brick_conv_rows(N, D, H, W);
  return 8 * (((int64_t)N * D * H * W + PCRL_CONV_BM - 1) / PCRL_CONV_BM);
}
extern "C" int pcrl_upconv_fwd(const void* x, const void* wf, const void* w3f, const float* bias_tab, void* y0, float* stats_partial, int N, int D, int H,
                               int W, int Ci, int Co, int dtype, pcrl_stream_t stream) {
  if (int e = check_upc("upconv_fwd", N, D, H, W, Ci, 32, Co, dtype)) return e;
  PCRL_REQUIRE(x && wf && bias_tab && y0, "upconv_fwd: null pointer");
  if (const int impl = pcrl_upc_fwd_impl(N, D, H, W, Ci, Co, dtype)) {
    PCRL_REQUIRE(w3f, "upconv_fwd: this shape runs on a brick kernel and needs the 3x3x3 form of the composed weights (w3f)");
    if (impl == 2) return pcrl_brick8_upc_fwd_launch(x, w3f, bias_tab, y0, stats_partial, N, D, H, W, Ci, Co, as_stream(stream));
    return pcrl_brick16_upc_fwd_launch(x, w3f, bias_tab, y0, stats_partial, N, D, H, W, Ci, Co, as_stream(stream));
  }
  return pcrl_upc_fwd_launch(x, wf, bias_tab, y0, stats_partial, N, D, H, W, Ci, Co, dtype, as_stream(stream));
}
extern "C" int64_t pcrl_upconv_dgrad_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return pcrl_upc_dgrad_uses_brick(N, D, H, W, Ci, Co, dtype) ? 1 : 0;
}
static int upconv_dgrad_impl(const void* dy0, const void* wd, const void* wd3, void* dx, void* ws, int64_t ws_bytes, int N, int D, int H, int W, int Ci,
                             int Co, int dtype, pcrl_stream_t stream) {
  if (int e = check_upc("upconv_dgrad", N, D, H, W, Ci, 32, Co, dtype)) return e;
  PCRL_REQUIRE(dy0 && wd && dx, "upconv_dgrad: null pointer");
  if (const int impl = pcrl_upc_dgrad_impl(N, D, H, W, Ci, Co, dtype)) {
    PCRL_REQUIRE(wd3, "upconv_dgrad: this shape runs on a brick kernel and needs the 3x3x3 form of the composed weights (wd3)");
    if (impl == 2) return pcrl_brick8_upc_dgrad_launch(dy0, wd3, dx, N, D, H, W, Ci, Co, as_stream(stream));
    return pcrl_brick16_upc_dgrad_launch(dy0, wd3, dx, N, D, H, W, Ci, Co, as_stream(stream));
  }
  return pcrl_upc_dgrad_launch(dy0, wd, dx, ws, ws_bytes, N, D, H, W, Ci, Co, dtype, as_stream(stream));
}
extern "C" int pcrl_upconv_dgrad(const void* dy0, const void* wd, const void* wd3, void* dx, int N, int D, int H, int W, int Ci, int Co, int dtype,
                                 pcrl_stream_t stream) {
  return upconv_dgrad_impl(dy0, wd, wd3, dx, nullptr, 0, N, D, H, W, Ci, Co, dtype, stream);
}
// The same with a caller-owned workspace: small coarse grids (few row tiles) split the 64-tap reduction over blockIdx.z into float partial
// sums and a finish pass, like pcrl_conv3d_k3_fwd_ws.  pcrl_upconv_dgrad_ws_bytes() == 0: no workspace needed (ws may be null).
extern "C" int64_t pcrl_upconv_dgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 32 || Co % 32) return 0;
  if (pcrl_upc_dgrad_uses_brick(N, D, H, W, Ci, Co, dtype)) return 0;
  return pcrl_upc_dgrad_ws_bytes(N, D, H, W, Ci, Co);
}
extern "C" int pcrl_upconv_dgrad_ws(const void* dy0, const void* wd, const void* wd3, void* dx, void* ws, int64_t ws_bytes, int N, int D, int H, int W,
                                    int Ci, int Co, int dtype, pcrl_stream_t stream) {
  return upconv_dgrad_impl(dy0, wd, wd3, dx, ws, ws_bytes, N, D, H, W, Ci, Co, dtype, stream);
}

// ---- parameter gradients ----
// Two stages, because both are LINEAR in (dWeff, box sums): `accum` runs once per backward pass and adds that pass's gradient of the
// composed weights and its border-class sums of dy0 to two small persistent buffers; `finish` runs the chain rule to the reference
// parameters once, after the last pass (three passes share the weights in a pre-training step: the weight-sized GEMMs run once).
namespace {
struct FinLayout {
  size_t a1, a2, b1, b2, z1, z2, total;
};
FinLayout fin_layout(int Ci, int Cm, int Co, int dtype) {
  FinLayout L;
  size_t o = 0;
  L.a1 = o;    o += al((size_t)27 * Co * 8 * Ci * esz(dtype));
  L.a2 = o;    o += al((size_t)27 * Co * 8 * Ci * esz(dtype));
  L.b1 = o;    o += al((size_t)Cm * 8 * Ci * esz(dtype));
  L.b2 = o;    o += al((size_t)Cm * 27 * Co * esz(dtype));
  L.z1 = o;    o += al((size_t)Cm * 27 * Co * sizeof(float));
  L.z2 = o;    o += al((size_t)Cm * 8 * Ci * sizeof(float));
  L.total = o;
  return L;
}
}  // namespace
extern "C" int64_t pcrl_upconv_wgrad_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return pcrl_upc_wgrad_uses_brick(N, D, H, W, Ci, Co, dtype) ? 1 : 0;
}
static size_t acc_wg_bytes(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  const size_t a = pcrl_upc_wgrad_ws_bytes(N, D, H, W, Ci, Co);
  if (!pcrl_upc_wgrad_uses_brick(N, D, H, W, Ci, Co, dtype)) return a;
  const size_t b = pcrl_upc_wgrad3_ws_bytes(N, D, H, W, Ci, Co);
  return a > b ? a : b;
}
extern "C" size_t pcrl_upconv_wgrad_accum_ws_bytes(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return 0;
  return al(acc_wg_bytes(N, D, H, W, Ci, Co, dtype)) + al((size_t)N * 2 * D * 9 * Co * sizeof(float)) + al((size_t)9 * 27 * Co * sizeof(float));   // [27][Co] + UPC_TOTAL_CHUNKS (8) chunk copies
}
// dweff_acc: float32 [Co][Ci][64]; box_acc: float32 [27][Co]; first != 0: store, else add.  The gradient of the composed weights comes from the
// brick weight-gradient kernel where it tiles the coarse grid (pcrl_upconv_wgrad_uses_brick() != 0), else from the gather kernel: same layout.
extern "C" int pcrl_upconv_wgrad_accum(const void* x, const void* dy0, float* dweff_acc, float* box_acc, int flags, void* ws, size_t ws_bytes, int N,
                                       int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream) {
  const int first = flags & 1;        // store instead of add
  const bool zsum = (flags & 2) != 0;  // the caller states that dy0 sums to zero over all voxels per channel (see upc_class_sums_kernel)
  if (int e = check_upc("upconv_wgrad_accum", N, D, H, W, Ci, 32, Co, dtype)) return e;
  PCRL_REQUIRE(x && dy0 && dweff_acc && box_acc, "upconv_wgrad_accum: null pointer");
  if (!ws || ws_bytes < pcrl_upconv_wgrad_accum_ws_bytes(N, D, H, W, Ci, Co, dtype)) return pcrl_fail(PCRL_EWORKSPACE, "upconv_wgrad_accum: workspace too small");
  hipStream_t st = as_stream(stream);
  char* w = (char*)ws;
  const size_t wgb = al(acc_wg_bytes(N, D, H, W, Ci, Co, dtype));
  if (pcrl_upc_wgrad_uses_brick(N, D, H, W, Ci, Co, dtype)) {
    if (int e = pcrl_upc_wgrad3_launch(dy0, x, dweff_acc, w, wgb, N, D, H, W, Ci, Co, st, first == 0)) return e;
  } else {
    if (int e = pcrl_upc_wgrad_launch(dy0, x, dweff_acc, w, wgb, N, D, H, W, Ci, Co, dtype, st, first == 0)) return e;
  }
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  if (Co % vec != 0 || (Co / vec) > 256 || 256 % (Co / vec) != 0) return pcrl_fail(PCRL_EINVAL, "upconv_wgrad_accum: Co=%d not supported by the class-sum kernel", Co);
  const size_t lds_old = (size_t)(256 / (Co / vec)) * Co * sizeof(float), lds_new = (size_t)36 * Co * sizeof(float);
  const size_t lds = (Co / vec) <= 64 ? (lds_new > lds_old ? lds_new : lds_old) : lds_old;
  float* part = (float*)(w + wgb);
  float* S = (float*)(w + wgb + al((size_t)N * 2 * D * 9 * Co * sizeof(float)));
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(upc_class_sums_kernel<bf16>, dim3((unsigned)(N * 2 * D)), dim3(256), lds, st, (const bf16*)dy0, part, 2 * D, 2 * H, 2 * W, Co, zsum);
  else hipLaunchKernelGGL(upc_class_sums_kernel<float>, dim3((unsigned)(N * 2 * D)), dim3(256), lds, st, (const float*)dy0, part, 2 * D, 2 * H, 2 * W, Co, zsum);
  float* Sk = S + 27 * Co;   // [UPC_TOTAL_CHUNKS][27][Co] behind the combined [27][Co]
  hipLaunchKernelGGL(upc_class_total_kernel, dim3((unsigned)(9 * ((Co + 63) / 64)), 3, UPC_TOTAL_CHUNKS), dim3(1024), 0, st, (const float*)part, Sk, N, 2 * D, Co);
  hipLaunchKernelGGL(upc_chunk_sum_kernel, dim3((27 * Co + 255) / 256), dim3(256), 0, st, (const float*)Sk, S, 27 * Co);
  hipLaunchKernelGGL(upc_box_kernel, dim3((27 * Co + 255) / 256), dim3(256), 0, st, (const float*)S, box_acc, Co, first == 0, zsum);
  return pcrl_check_launch("upconv_wgrad_accum");
}
extern "C" size_t pcrl_upconv_wgrad_finish_ws_bytes(int Ci, int Cm, int Co, int dtype) {
  if (Ci <= 0 || Cm <= 0 || Co <= 0) return 0;
  return fin_layout(Ci, Cm, Co, dtype).total;
}
extern "C" int pcrl_upconv_wgrad_finish(const float* dweff_acc, const float* box_acc, const float* w_up, const float* b_up, const float* w0, float* dw_up,
                                        float* db_up, float* dw0, void* ws, size_t ws_bytes, int Ci, int Cm, int Co, int dtype, pcrl_stream_t stream) {
  if (int e = check_upc("upconv_wgrad_finish", 1, 1, 1, 1, Ci, Cm, Co, dtype)) return e;
  PCRL_REQUIRE(dweff_acc && box_acc && w_up && b_up && w0 && dw_up && db_up && dw0, "upconv_wgrad_finish: null pointer");
  const FinLayout L = fin_layout(Ci, Cm, Co, dtype);
  if (!ws || ws_bytes < L.total) return pcrl_fail(PCRL_EWORKSPACE, "upconv_wgrad_finish: workspace %zu < %zu", ws_bytes, L.total);
  hipStream_t st = as_stream(stream);
  char* w = (char*)ws;
  const unsigned gpre = blocks_for((int64_t)27 * Co * Cm + (int64_t)8 * Ci * Cm);
  if (dtype == PCRL_BF16) {
    hipLaunchKernelGGL(upc_prep_kernel<bf16>, dim3(gpre), dim3(256), 0, st, w_up, w0, (bf16*)nullptr, (bf16*)nullptr, (bf16*)(w + L.b1), (bf16*)(w + L.b2), Ci, Cm, Co);
    hipLaunchKernelGGL(upc_chain_pack_kernel<bf16>, dim3((unsigned)(Ci / 32), (unsigned)Co), dim3(256), 0, st, dweff_acc, (bf16*)(w + L.a1), Ci, Co);
    hipLaunchKernelGGL(transpose_kernel<bf16>, dim3((unsigned)(8 * Ci / 32), (unsigned)(27 * Co / 32)), dim3(256), 0, st, (const bf16*)(w + L.a1), (bf16*)(w + L.a2), 27 * Co, 8 * Ci);
  } else {
    hipLaunchKernelGGL(upc_prep_kernel<float>, dim3(gpre), dim3(256), 0, st, w_up, w0, (float*)nullptr, (float*)nullptr, (float*)(w + L.b1), (float*)(w + L.b2), Ci, Cm, Co);
    hipLaunchKernelGGL(upc_chain_pack_kernel<float>, dim3((unsigned)(Ci / 32), (unsigned)Co), dim3(256), 0, st, dweff_acc, (float*)(w + L.a1), Ci, Co);
    hipLaunchKernelGGL(transpose_kernel<float>, dim3((unsigned)(8 * Ci / 32), (unsigned)(27 * Co / 32)), dim3(256), 0, st, (const float*)(w + L.a1), (float*)(w + L.a2), 27 * Co, 8 * Ci);
  }
  if (int e = pcrl_check_launch("upconv_wgrad_finish (pack)")) return e;
  if (int e = pcrl_gemm_planes_launch(w + L.a1, w + L.b1, (float*)(w + L.z1), (int64_t)27 * Co, 8 * Ci, Cm, dtype, st)) return e;   // z1[cm][(t,co)]
  if (int e = pcrl_gemm_planes_launch(w + L.a2, w + L.b2, (float*)(w + L.z2), (int64_t)8 * Ci, 27 * Co, Cm, dtype, st)) return e;   // z2[cm][(s,ci)]
  hipLaunchKernelGGL(upc_dbup_kernel, dim3(Cm), dim3(256), 0, st, w0, box_acc, db_up, Cm, Co);
  hipLaunchKernelGGL(upc_chain_unpack_kernel, dim3(gpre), dim3(256), 0, st, (const float*)(w + L.z1), (const float*)(w + L.z2), b_up, box_acc, dw0, dw_up, Ci, Cm,
                     Co);
  return pcrl_check_launch("upconv_wgrad_finish");
}
// both stages for one pass (accumulators inside ws)
extern "C" size_t pcrl_upconv_wgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Cm, int Co, int dtype) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Cm <= 0 || Co <= 0) return 0;
  const size_t a = pcrl_upconv_wgrad_accum_ws_bytes(N, D, H, W, Ci, Co, dtype), f = fin_layout(Ci, Cm, Co, dtype).total;
  return al((size_t)64 * Co * Ci * sizeof(float)) + al((size_t)27 * Co * sizeof(float)) + (a > f ? a : f);
}
extern "C" int pcrl_upconv_wgrad(const void* x, const void* dy0, const float* w_up, const float* b_up, const float* w0, float* dw_up, float* db_up,
                                 float* dw0, void* ws, size_t ws_bytes, int N, int D, int H, int W, int Ci, int Cm, int Co, int dtype,
                                 pcrl_stream_t stream) {
  if (int e = check_upc("upconv_wgrad", N, D, H, W, Ci, Cm, Co, dtype)) return e;
  if (!ws || ws_bytes < pcrl_upconv_wgrad_ws_bytes(N, D, H, W, Ci, Cm, Co, dtype)) return pcrl_fail(PCRL_EWORKSPACE, "upconv_wgrad: workspace too small");
  char* w = (char*)ws;
  float* dweff = (float*)w;
  float* box = (float*)(w + al((size_t)64 * Co * Ci * sizeof(float)));
  char* rest = (char*)box + al((size_t)27 * Co * sizeof(float));
  const size_t rest_bytes = ws_bytes - (size_t)(rest - w);
  if (int e = pcrl_upconv_wgrad_accum(x, dy0, dweff, box, 1, rest, rest_bytes, N, D, H, W, Ci, Co, dtype, stream)) return e;
  return pcrl_upconv_wgrad_finish(dweff, box, w_up, b_up, w0, dw_up, db_up, dw0, rest, rest_bytes, Ci, Cm, Co, dtype, stream);
}
