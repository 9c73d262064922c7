// HBM-bound operators of the 2D PCRLv2 path for gfx950, NHWC, float32 or bf16:
//   MaxPool2d(3, stride 2, padding 1)   -- smp/torchvision ResNet stem (encoder stage 2)
//   backward of F.interpolate(scale_factor=2, mode="nearest")   -- models/pcrlv2_model.py:114 (the forward is fused into the
//                                                                   consumer convolution's gather, conv2d.hip)
//   F.interpolate(scale_factor=s, mode="bilinear") fwd/bwd on the 3-channel deep-supervision maps   -- pcrlv2_model.py:190
//   relu(t + identity) of the BasicBlock and its backward mask
// All are one coalesced pass; nothing uses atomics (gather formulations, deterministic).
#include "common.h"

namespace {

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// y[n][oh][ow][c] = max over the 3x3 window at (2oh-1, 2ow-1); idx = kh*3+kw of the FIRST maximum in scan order (torch's choice;
// NaN propagates like torch: a NaN wins).
template <typename T>
__global__ void __launch_bounds__(256) maxpool2d_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int N, int H, int W,
                                                            int C, int Ho, int Wo, int64_t total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % nvec);
    int64_t t = i / nvec;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho), n = (int)(t / Ho);
    float m[VEC];
    uint8_t mi[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      m[j] = -INFINITY;
      mi[j] = 255;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = 2 * oh - 1 + kh, iw = 2 * ow - 1 + kw;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
          const Vec16<T> v = ld16(x + (((int64_t)n * H + ih) * W + iw) * C + cv * VEC);
          const uint8_t code = (uint8_t)(kh * 3 + kw);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float f = to_f(v.v[j]);
            if (mi[j] == 255 || f > m[j] || f != f) {   // aten max_pool2d: `val > maxval || isnan(val)`, index starts at the window's first pixel
              if (mi[j] == 255 && !(f > m[j] || f != f)) {
                mi[j] = code;                            // -inf first element: keeps the index, not the value test
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
    st16(y + i * VEC, o);
#pragma unroll
    for (int j = 0; j < VEC; ++j) idx[i * VEC + j] = mi[j];
  }
}

// dx[n][ih][iw][c] = sum over the (at most 2x2) windows that contain the pixel of dy where the stored argmax is this pixel
template <typename T>
__global__ void __launch_bounds__(256) maxpool2d_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx, int N, int H,
                                                            int W, int C, int Ho, int Wo, int64_t total) {
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
    // windows oh with 2oh-1 <= ih <= 2oh+1  <=>  oh in [ceil((ih-1)/2), floor((ih+1)/2)]
    const int oh0 = ih >> 1, oh1 = (ih + 1) >> 1;   // even ih: one window (kh = 1); odd ih: two (kh = 2 and 0)
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
        const Vec16<T> g = ld16(dy + o);
        const uint8_t code = (uint8_t)(kh * 3 + kw);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (idx[o + j] == code) acc[j] += to_f(g.v[j]);
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>(acc[j]);
    st16(dx + i * VEC, o);
  }
}

// dx[n][h][w][c] = sum_{i,j in {0,1}} dy[n][2h+i][2w+j][c]
template <typename T>
__global__ void __launch_bounds__(256) nearest2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int C, int64_t total) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nvec = C / VEC;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % nvec);
    int64_t t = i / nvec;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int64_t n = t / H;
    const T* base = dy + (((n * 2 * H + 2 * h) * 2 * W + 2 * w) * (int64_t)nvec + cv) * VEC;
    const Vec16<T> a = ld16(base), b = ld16(base + (int64_t)C), c = ld16(base + (int64_t)2 * W * C), d = ld16(base + (int64_t)2 * W * C + C);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = from_f<T>((to_f(a.v[j]) + to_f(b.v[j])) + (to_f(c.v[j]) + to_f(d.v[j])));
    st16(dx + i * VEC, o);
  }
}

// torch's area_pixel_compute_source_index(align_corners=False) with the user's scale factor: src = (dst + .5) / s - .5, clamped at 0
__device__ __forceinline__ void bilinear_src(int dst, float inv_scale, int in_size, int& i0, int& i1, float& l1) {
  float src = ((float)dst + 0.5f) * inv_scale - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

// x: float32 [N][H][W][C] (C small: the 3-channel maps), y: [N][H*s][W*s][C]; thread = output pixel, all C channels (C <= 4 fast path)
__global__ void __launch_bounds__(256) bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int s, int64_t total) {
  const int Ho = H * s, Wo = W * s;
  const float inv = 1.0f / (float)s;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t t = i;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int64_t n = t / Ho;
    int h0, h1, w0, w1;
    float lh, lw;
    bilinear_src(oh, inv, H, h0, h1, lh);
    bilinear_src(ow, inv, W, w0, w1, lw);
    const float* b = x + n * H * W * C;
    const float* p00 = b + ((int64_t)h0 * W + w0) * C;
    const float* p01 = b + ((int64_t)h0 * W + w1) * C;
    const float* p10 = b + ((int64_t)h1 * W + w0) * C;
    const float* p11 = b + ((int64_t)h1 * W + w1) * C;
    float* o = y + i * C;
    for (int c = 0; c < C; ++c)
      o[c] = (1.f - lh) * ((1.f - lw) * p00[c] + lw * p01[c]) + lh * ((1.f - lw) * p10[c] + lw * p11[c]);
  }
}

// dx[n][h][w][c] = sum over the output pixels whose stencil touches (h, w): candidates oh in [(h-1)s, (h+2)s), same for w
__global__ void __launch_bounds__(256) bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C, int s, int64_t total) {
  const int Ho = H * s, Wo = W * s;
  const float inv = 1.0f / (float)s;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    int64_t t = i / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int64_t n = t / H;
    const int oh_lo = max((h - 1) * s, 0), oh_hi = min((h + 2) * s, Ho);
    const int ow_lo = max((w - 1) * s, 0), ow_hi = min((w + 2) * s, Wo);
    float acc = 0.f;
    for (int oh = oh_lo; oh < oh_hi; ++oh) {
      int h0, h1;
      float lh;
      bilinear_src(oh, inv, H, h0, h1, lh);
      float wh = 0.f;
      if (h0 == h) wh += 1.f - lh;
      if (h1 == h) wh += lh;
      if (wh == 0.f) continue;
      const float* row = dy + ((n * Ho + oh) * (int64_t)Wo) * C + c;
      float racc = 0.f;
      for (int ow = ow_lo; ow < ow_hi; ++ow) {
        int w0, w1;
        float lw;
        bilinear_src(ow, inv, W, w0, w1, lw);
        float ww = 0.f;
        if (w0 == w) ww += 1.f - lw;
        if (w1 == w) ww += lw;
        if (ww != 0.f) racc += ww * row[(int64_t)ow * C];
      }
      acc += wh * racc;
    }
    dx[i] = acc;
  }
}

// a = relu(t + r)
template <typename T>
__global__ void __launch_bounds__(256) add_relu_kernel(const T* __restrict__ t, const T* __restrict__ r, T* __restrict__ a, int64_t nvec) {
  constexpr int VEC = 16 / (int)sizeof(T);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const Vec16<T> x = ld16(t + i * VEC), y = ld16(r + i * VEC);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float v = to_f(x.v[j]) + to_f(y.v[j]);
      o.v[j] = from_f<T>(v > 0.f ? v : 0.f);
    }
    st16(a + i * VEC, o);
  }
}

// g = da where a > 0 else 0
template <typename T>
__global__ void __launch_bounds__(256) relu_mask_kernel(const T* __restrict__ da, const T* __restrict__ a, T* __restrict__ g, int64_t nvec) {
  constexpr int VEC = 16 / (int)sizeof(T);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const Vec16<T> x = ld16(da + i * VEC), y = ld16(a + i * VEC);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = to_f(y.v[j]) > 0.f ? x.v[j] : from_f<T>(0.f);
    st16(g + i * VEC, o);
  }
}

int check2d(const char* what, int N, int H, int W, int C, int dtype) {
  if (N <= 0 || H <= 0 || W <= 0) return pcrl_fail(PCRL_EINVAL, "%s: bad dims %d %d %d", what, N, H, W);
  if (dtype != PCRL_F32 && dtype != PCRL_BF16) return pcrl_fail(PCRL_EINVAL, "%s: bad dtype %d", what, dtype);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  if (C <= 0 || C % vec != 0) return pcrl_fail(PCRL_EINVAL, "%s: C=%d must be a positive multiple of %d", what, C, vec);
  return 0;
}

}  // namespace

extern "C" int pcrl_maxpool2d_3s2_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int dtype, pcrl_stream_t stream) {
  if (int e = check2d("maxpool2d_3s2_fwd", N, H, W, C, dtype)) return e;
  PCRL_REQUIRE(x && y && idx, "maxpool2d_3s2_fwd: null pointer");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, vec = dtype == PCRL_BF16 ? 8 : 4;
  const int64_t total = (int64_t)N * Ho * Wo * (C / vec);
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(maxpool2d_fwd_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const bf16*)x, (bf16*)y, idx, N, H, W, C, Ho, Wo, total);
  else
    hipLaunchKernelGGL(maxpool2d_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, idx, N, H, W, C, Ho, Wo, total);
  return pcrl_check_launch("maxpool2d_3s2_fwd");
}

extern "C" int pcrl_maxpool2d_3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int dtype, pcrl_stream_t stream) {
  if (int e = check2d("maxpool2d_3s2_bwd", N, H, W, C, dtype)) return e;
  PCRL_REQUIRE(dy && dx && idx, "maxpool2d_3s2_bwd: null pointer");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, vec = dtype == PCRL_BF16 ? 8 : 4;
  const int64_t total = (int64_t)N * H * W * (C / vec);
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(maxpool2d_bwd_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const bf16*)dy, idx, (bf16*)dx, N, H, W, C, Ho, Wo, total);
  else
    hipLaunchKernelGGL(maxpool2d_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)dy, idx, (float*)dx, N, H, W, C, Ho, Wo, total);
  return pcrl_check_launch("maxpool2d_3s2_bwd");
}

// dx: [N][H][W][C] from dy: [N][2H][2W][C]
extern "C" int pcrl_upsample2d_nearest2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, pcrl_stream_t stream) {
  if (int e = check2d("upsample2d_nearest2_bwd", N, H, W, C, dtype)) return e;
  PCRL_REQUIRE(dy && dx, "upsample2d_nearest2_bwd: null pointer");
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  const int64_t total = (int64_t)N * H * W * (C / vec);
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(nearest2_bwd_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const bf16*)dy, (bf16*)dx, H, W, C, total);
  else hipLaunchKernelGGL(nearest2_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float*)dy, (float*)dx, H, W, C, total);
  return pcrl_check_launch("upsample2d_nearest2_bwd");
}

extern "C" int pcrl_upsample2d_bilinear_fwd(const float* x, float* y, int N, int H, int W, int C, int scale, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && scale >= 1, "upsample2d_bilinear_fwd: bad arguments");
  const int64_t total = (int64_t)N * H * scale * W * scale;
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), x, y, H, W, C, scale, total);
  return pcrl_check_launch("upsample2d_bilinear_fwd");
}

extern "C" int pcrl_upsample2d_bilinear_bwd(const float* dy, float* dx, int N, int H, int W, int C, int scale, pcrl_stream_t stream) {
  PCRL_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && scale >= 1, "upsample2d_bilinear_bwd: bad arguments");
  const int64_t total = (int64_t)N * H * W * C;
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), dy, dx, H, W, C, scale, total);
  return pcrl_check_launch("upsample2d_bilinear_bwd");
}

extern "C" int pcrl_add_relu_fwd(const void* t, const void* r, void* a, int64_t n, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(t && r && a && n > 0, "add_relu_fwd: bad arguments");
  PCRL_REQUIRE(dtype == PCRL_F32 || dtype == PCRL_BF16, "add_relu_fwd: bad dtype %d", dtype);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  PCRL_REQUIRE(n % vec == 0, "add_relu_fwd: n must be a multiple of %d", vec);
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(add_relu_kernel<bf16>, dim3(grid_for(n / vec)), dim3(256), 0, as_stream(stream), (const bf16*)t, (const bf16*)r, (bf16*)a, n / vec);
  else hipLaunchKernelGGL(add_relu_kernel<float>, dim3(grid_for(n / vec)), dim3(256), 0, as_stream(stream), (const float*)t, (const float*)r, (float*)a, n / vec);
  return pcrl_check_launch("add_relu_fwd");
}

extern "C" int pcrl_relu_mask_bwd(const void* da, const void* a, void* g, int64_t n, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(da && a && g && n > 0, "relu_mask_bwd: bad arguments");
  PCRL_REQUIRE(dtype == PCRL_F32 || dtype == PCRL_BF16, "relu_mask_bwd: bad dtype %d", dtype);
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  PCRL_REQUIRE(n % vec == 0, "relu_mask_bwd: n must be a multiple of %d", vec);
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(relu_mask_kernel<bf16>, dim3(grid_for(n / vec)), dim3(256), 0, as_stream(stream), (const bf16*)da, (const bf16*)a, (bf16*)g, n / vec);
  else hipLaunchKernelGGL(relu_mask_kernel<float>, dim3(grid_for(n / vec)), dim3(256), 0, as_stream(stream), (const float*)da, (const float*)a, (float*)g, n / vec);
  return pcrl_check_launch("relu_mask_bwd");
}
