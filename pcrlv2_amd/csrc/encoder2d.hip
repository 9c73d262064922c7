// Fused HBM-bound passes of the 2D ResNet-18 encoder (torchvision BasicBlock / ResNet stem as smp.Unet('resnet18') wraps them,
// models/pcrlv2_model.py:200) for gfx950, NHWC, float32 or bf16.  Each replaces a chain of separate passes over the same tensor and
// reproduces that chain's values bit for bit (the intermediate the chain stored is rounded to the storage type here as well):
//
//   out = relu(bn2(y2) + identity)         BasicBlock tail: BatchNorm2d apply + (optionally the downsample branch's BatchNorm2d apply) + add + relu
//   g = (da + db) where out > 0            its backward mask over the SUM of the two gradients that reach a block's output (the next block's
//                                          conv1 branch and its identity branch) -- autograd's aten::add + the mask
//   p, idx = MaxPool2d(3, 2, 1)(relu(bn1(y)))   the stem: BatchNorm2d apply + ReLU + max-pool from one pass over the convolution output
//                                          (the full-resolution activation has no other consumer: the decoder ignores the skips, :115-117)
//   dx = maxpool backward of (dy + dy2)    the pooled tensor feeds layer1.0's conv1 and its identity branch
#include "common.h"

namespace {

inline unsigned rc_grid(int64_t M, int nvec) {
  const int nslots = 256 / nvec;
  int64_t b = (M + nslots - 1) / nslots;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (unsigned)b;
}
inline unsigned grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// A thread owns one 16-byte channel vector for its whole life (coefficients in registers), rows strided over the grid.
template <typename T, bool NT>
__device__ __forceinline__ void bn_add_relu_body(const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                   const T* __restrict__ r, const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                   T* __restrict__ out, int64_t M, int C) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC, cv = threadIdx.x % nvec, slot = threadIdx.x / nvec, nslots = 256 / nvec;
  float sc[VEC], sh[VEC], rs[VEC], rh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int c = cv * VEC + j;
    sc[j] = scale[c]; sh[j] = shift[c];
    rs[j] = rscale ? rscale[c] : 1.f; rh[j] = rscale ? rshift[c] : 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * nslots;
#pragma unroll 4
  for (int64_t m = (int64_t)blockIdx.x * nslots + slot; m < M; m += stride) {
    const int64_t off = (m * nvec + cv) * VEC;
    const Vec16<T> v = ld16_sel<NT>(y + off), w = ld16_sel<NT>(r + off);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float t = to_f(from_f<T>(sc[j] * to_f(v.v[j]) + sh[j]));                          // what pcrl_bn_act_apply(ACT_NONE) stores
      const float i = rscale ? to_f(from_f<T>(rs[j] * to_f(w.v[j]) + rh[j])) : to_f(w.v[j]);  // the downsample branch's BatchNorm, or the identity
      const float s = t + i;
      o.v[j] = from_f<T>(s > 0.f ? s : 0.f);
    }
    st16_sel<NT>(out + off, o);
  }
}
template <typename T>
__global__ void __launch_bounds__(256) bn_add_relu_kernel(const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const T* __restrict__ r, const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                          T* __restrict__ out, int64_t M, int C, bool nt) {
  if (nt) bn_add_relu_body<T, true>(y, scale, shift, r, rscale, rshift, out, M, C);
  else bn_add_relu_body<T, false>(y, scale, shift, r, rscale, rshift, out, M, C);
}

template <typename T>
__global__ void __launch_bounds__(256) relu_mask_sum_kernel(const T* __restrict__ da, const T* __restrict__ db, const T* __restrict__ a, T* __restrict__ g,
                                                            int64_t nvec) {
  constexpr int VEC = 16 / (int)sizeof(T);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const Vec16<T> x = ld16(da + i * VEC), z = ld16(db + i * VEC), y = ld16(a + i * VEC);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = to_f(y.v[j]) > 0.f ? from_f<T>(to_f(x.v[j]) + to_f(z.v[j])) : from_f<T>(0.f);
    st16(g + i * VEC, o);
  }
}

// p[n][oh][ow][c] = max over the 3x3 window at (2oh-1, 2ow-1) of a = T(act(scale * y + shift)); idx = kh*3+kw of the FIRST maximum in scan order
// (the comparison runs on the values rounded to T: exactly the tensor the unfused chain stored and pooled).
template <typename T>
__global__ void __launch_bounds__(256) bn_relu_maxpool2d_kernel(const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                                T* __restrict__ p, uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo,
                                                                int64_t total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % nvec);
    int64_t t = i / nvec;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho), n = (int)(t / Ho);
    float sc[VEC], sh[VEC], m[VEC];
    uint8_t mi[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      sc[j] = scale[cv * VEC + j]; sh[j] = shift[cv * VEC + j];
      m[j] = -INFINITY;
      mi[j] = 255;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = 2 * oh - 1 + kh, iw = 2 * ow - 1 + kw;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          const Vec16<T> v = ld16(y + (((int64_t)n * H + ih) * W + iw) * C + cv * VEC);
          const uint8_t code = (uint8_t)(kh * 3 + kw);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float z = sc[j] * to_f(v.v[j]) + sh[j];
            const float f = to_f(from_f<T>(z > 0.f ? z : 0.f));
            if (mi[j] == 255 || f > m[j] || f != f) {     // same rule as maxpool2d_fwd_kernel (ops2d.hip): first maximum, NaN wins
              if (mi[j] == 255 && !(f > m[j] || f != f)) {
                mi[j] = code;
              } else {
                m[j] = f;
                mi[j] = code;
              }
            }
          }
        }
      }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(m[j]);
    st16(p + i * VEC, o);
#pragma unroll
    for (int j = 0; j < VEC; ++j) idx[i * VEC + j] = mi[j];
  }
}

// dx[n][ih][iw][c] = sum over the (at most 2x2) windows that contain the pixel of (dy + dy2) where the stored argmax is this pixel
template <typename T>
__global__ void __launch_bounds__(256) maxpool2d_bwd_sum_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const uint8_t* __restrict__ idx,
                                                                T* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo, int64_t total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % nvec);
    int64_t t = i / nvec;
    const int iw = (int)(t % W);
    t /= W;
    const int ih = (int)(t % H), n = (int)(t / H);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    const int oh0 = ih >> 1, oh1 = (ih + 1) >> 1;
    const int ow0 = iw >> 1, ow1 = (iw + 1) >> 1;
    for (int a = 0; a < 2; ++a) {
      const int oh = a ? oh1 : oh0;
      if (a && oh1 == oh0) continue;
      if (oh >= Ho) continue;
      const int kh = ih - (2 * oh - 1);
      if ((unsigned)kh > 2u) continue;
      for (int b = 0; b < 2; ++b) {
        const int ow = b ? ow1 : ow0;
        if (b && ow1 == ow0) continue;
        if (ow >= Wo) continue;
        const int kw = iw - (2 * ow - 1);
        if ((unsigned)kw > 2u) continue;
        const int64_t o = ((((int64_t)n * Ho + oh) * Wo + ow) * nvec + cv) * VEC;
        const Vec16<T> g = ld16(dy + o), g2 = ld16(dy2 + o);
        const uint8_t code = (uint8_t)(kh * 3 + kw);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (idx[o + j] == code) acc[j] += to_f(from_f<T>(to_f(g.v[j]) + to_f(g2.v[j])));     // the sum rounded to T first: what aten::add handed the old kernel
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(acc[j]);
    st16(dx + i * VEC, o);
  }
}

int check_c(const char* what, int C, int dtype, bool rc) {
  if (dtype != PCRL_F32 && dtype != PCRL_BF16) return pcrl_fail(PCRL_EINVAL, "%s: bad dtype %d", what, dtype);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  if (C <= 0 || C % vec != 0) return pcrl_fail(PCRL_EINVAL, "%s: C=%d must be a positive multiple of %d", what, C, vec);
  if (rc && (C / vec > 256 || 256 % (C / vec) != 0)) return pcrl_fail(PCRL_EINVAL, "%s: C=%d: channel vectors must divide 256", what, C);
  return 0;
}

}  // namespace

extern "C" int pcrl_bn_add_relu_fwd(const void* y, const float* scale, const float* shift, const void* r, const float* rscale, const float* rshift,
                                    void* out, int64_t M, int C, int dtype, pcrl_stream_t stream) {
  if (int e = check_c("bn_add_relu_fwd", C, dtype, true)) return e;
  PCRL_REQUIRE(y && scale && shift && r && out && M > 0 && (!rscale == !rshift), "bn_add_relu_fwd: bad arguments");
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const bool nt = pcrl_streaming(M * C * (int64_t)(dtype == PCRL_BF16 ? 2 : 4));
  const dim3 grid(rc_grid(M, C / vec));
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(bn_add_relu_kernel<bf16>, grid, dim3(256), 0, as_stream(stream), (const bf16*)y, scale, shift, (const bf16*)r, rscale, rshift, (bf16*)out, M, C, nt);
  else hipLaunchKernelGGL(bn_add_relu_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)y, scale, shift, (const float*)r, rscale, rshift, (float*)out, M, C, nt);
  return pcrl_check_launch("bn_add_relu_fwd");
}

extern "C" int pcrl_relu_mask_sum_bwd(const void* da, const void* db, const void* a, void* g, int64_t n, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(da && db && a && g && n > 0, "relu_mask_sum_bwd: bad arguments");
  PCRL_REQUIRE(dtype == PCRL_F32 || dtype == PCRL_BF16, "relu_mask_sum_bwd: bad dtype %d", dtype);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  PCRL_REQUIRE(n % vec == 0, "relu_mask_sum_bwd: n must be a multiple of %d", vec);
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(relu_mask_sum_kernel<bf16>, dim3(grid_for(n / vec)), dim3(256), 0, as_stream(stream), (const bf16*)da, (const bf16*)db, (const bf16*)a, (bf16*)g, n / vec);
  else hipLaunchKernelGGL(relu_mask_sum_kernel<float>, dim3(grid_for(n / vec)), dim3(256), 0, as_stream(stream), (const float*)da, (const float*)db, (const float*)a, (float*)g, n / vec);
  return pcrl_check_launch("relu_mask_sum_bwd");
}

extern "C" int pcrl_bn_relu_maxpool2d_3s2_fwd(const void* y, const float* scale, const float* shift, void* p, uint8_t* idx, int N, int H, int W, int C,
                                              int dtype, pcrl_stream_t stream) {
  if (int e = check_c("bn_relu_maxpool2d_3s2_fwd", C, dtype, false)) return e;
  PCRL_REQUIRE(y && scale && shift && p && idx && N > 0 && H > 0 && W > 0, "bn_relu_maxpool2d_3s2_fwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, vec = dtype == PCRL_BF16 ? 8 : 4;
  const int64_t total = (int64_t)N * Ho * Wo * (C / vec);
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(bn_relu_maxpool2d_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const bf16*)y, scale, shift, (bf16*)p, idx, N, H, W, C, Ho, Wo, total);
  else
    hipLaunchKernelGGL(bn_relu_maxpool2d_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)y, scale, shift, (float*)p, idx, N, H, W, C, Ho, Wo, total);
  return pcrl_check_launch("bn_relu_maxpool2d_3s2_fwd");
}

extern "C" int pcrl_maxpool2d_3s2_bwd_sum(const void* dy, const void* dy2, const uint8_t* idx, void* dx, int N, int H, int W, int C, int dtype,
                                          pcrl_stream_t stream) {
  if (int e = check_c("maxpool2d_3s2_bwd_sum", C, dtype, false)) return e;
  PCRL_REQUIRE(dy && dy2 && dx && idx && N > 0 && H > 0 && W > 0, "maxpool2d_3s2_bwd_sum: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, vec = dtype == PCRL_BF16 ? 8 : 4;
  const int64_t total = (int64_t)N * H * W * (C / vec);
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(maxpool2d_bwd_sum_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const bf16*)dy, (const bf16*)dy2, idx, (bf16*)dx, N, H, W, C, Ho, Wo, total);
  else
    hipLaunchKernelGGL(maxpool2d_bwd_sum_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)dy, (const float*)dy2, idx, (float*)dx, N, H, W, C, Ho, Wo, total);
  return pcrl_check_launch("maxpool2d_3s2_bwd_sum");
}
