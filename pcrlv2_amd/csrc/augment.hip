// Device-side augmentation of the LUNA pre-task crops: the seven torchio transforms of the reference's input
// pipeline (data.py:73-89 -> datasets/lunaDataset.py:28-81) as hand-written gfx950 kernels on float32 volumes [B][D][H][W]
// (one random parameter set per volume; the parameters are drawn by the host side, pcrlv2_amd/data.py):
//
//   RandomFlip(axes=0) + RandomAffine(scales, degrees, linear, pad = volume minimum)   -> pcrl_aug_volume_min + pcrl_aug_affine
//   RandomBlur(std per axis)                                                           -> pcrl_aug_blur_axis  (x3, separable)
//   RandomNoise + RandomGamma                                                          -> pcrl_aug_noise_gamma
//   RandomSwap(patch, iterations)                                                      -> pcrl_aug_swap
//   ZNormalization                                                                     -> pcrl_aug_meanstd + pcrl_aug_znorm
//
// All kernels are HBM/latency-bound streaming kernels over tensors of a few MB (64 global crops of 512 KB + 192 local crops of
// 16 KB per b = 32 batch); coalesced along w, per-volume coefficients in registers / LDS.  PARITY UNPINNED against torchio (not
// installed in the image; the reference holds no vectors): the kernels follow torchio's documented behaviour and are tested
// against a float64 PyTorch restatement of the same definitions (tests/test_augment_gpu.py).
#include "common.h"

namespace {

// ---- per-volume minimum (the fill value of RandomAffine: default_pad_value = 'minimum') ----
__global__ void __launch_bounds__(1024) vol_min_kernel(const float* __restrict__ x, float* __restrict__ vmin, int64_t S) {
  const float* v = x + (int64_t)blockIdx.x * S;
  float m = INFINITY;
  for (int64_t i = threadIdx.x; i < S; i += blockDim.x) m = fminf(m, v[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
  __shared__ float red[16];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fminf(m, red[i]);
    vmin[blockIdx.x] = m;
  }
}

// ---- flip along d (optional) followed by an affine resampling about the volume centre, trilinear, outside = fill ----
// inv[b] (row-major 3x3, (d,h,w) order, isotropic voxel units): OUTPUT centred coordinate -> INPUT centred coordinate.
__global__ void __launch_bounds__(256) affine_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ inv,
                                                     const int* __restrict__ flip, const float* __restrict__ fill, int D, int H, int W) {
  const int b = blockIdx.y;
  const int64_t S = (int64_t)D * H * W;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  const int w = (int)(i % W), h = (int)((i / W) % H), d = (int)(i / ((int64_t)W * H));
  const float* m = inv + b * 9;
  const float cd = d + 0.5f - 0.5f * D, ch = h + 0.5f - 0.5f * H, cw = w + 0.5f - 0.5f * W;
  float sd = m[0] * cd + m[1] * ch + m[2] * cw + 0.5f * D - 0.5f;
  const float sh = m[3] * cd + m[4] * ch + m[5] * cw + 0.5f * H - 0.5f;
  const float sw = m[6] * cd + m[7] * ch + m[8] * cw + 0.5f * W - 0.5f;
  if (flip[b]) sd = (float)(D - 1) - sd;   // sampling flip(x) at s == sampling x at the mirrored coordinate
  const float fd = floorf(sd), fh = floorf(sh), fw = floorf(sw);
  const int d0 = (int)fd, h0 = (int)fh, w0 = (int)fw;
  const float td = sd - fd, th = sh - fh, tw = sw - fw;
  const float* v = x + b * S;
  const float pad = fill[b];
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int dd = d0 + a, hh = h0 + bb, ww = w0 + c;
        const bool in = (unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
        const float val = in ? v[((int64_t)dd * H + hh) * W + ww] : pad;
        acc += val * (a ? td : 1.f - td) * (bb ? th : 1.f - th) * (c ? tw : 1.f - tw);
      }
  y[b * S + i] = acc;
}

// ---- one axis of a separable Gaussian blur, standard deviation sigma[b] (voxels), symmetric borders, taps -r .. r ----
__global__ void __launch_bounds__(256) blur_axis_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ sigma,
                                                        int D, int H, int W, int axis, int radius) {
  __shared__ float wt[65];
  const int b = blockIdx.y;
  const int n = axis == 0 ? D : axis == 1 ? H : W;
  const int r = radius < n ? radius : n;   // as many taps as the symmetric padding of the volume provides
  if ((int)threadIdx.x <= 2 * r) {
    const float s = fmaxf(sigma[b], 1e-3f), t = (float)((int)threadIdx.x - r);
    wt[threadIdx.x] = expf(-0.5f * (t / s) * (t / s));
  }
  __syncthreads();
  float norm = 0.f;
  for (int k = 0; k <= 2 * r; ++k) norm += wt[k];
  const int64_t S = (int64_t)D * H * W;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  const int w = (int)(i % W), h = (int)((i / W) % H), d = (int)(i / ((int64_t)W * H));
  const int pos = axis == 0 ? d : axis == 1 ? h : w;
  const int64_t stride = axis == 0 ? (int64_t)H * W : axis == 1 ? W : 1;
  const float* v = x + b * S + (i - pos * stride);
  float acc = 0.f;
  for (int k = -r; k <= r; ++k) {
    int q = pos + k;
    q = q < 0 ? -1 - q : (q >= n ? 2 * n - 1 - q : q);   // symmetric (edge value repeated once): scipy 'reflect', torch flip-padding
    acc += wt[k + r] * v[q * stride];
  }
  y[b * S + i] = acc / norm;
}

// ---- counter-based normal deviates: two rounds of an integer hash -> Box-Muller ----
__device__ __forceinline__ uint32_t mix32(uint32_t v) {
  v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
  return v;
}
__device__ __forceinline__ float normal_at(uint64_t seed, uint32_t vol, uint64_t idx) {
  const uint32_t k0 = mix32((uint32_t)seed ^ mix32(vol * 0x9e3779b9u + 0x85ebca6bu));
  const uint32_t a = mix32((uint32_t)idx ^ k0), b2 = mix32((uint32_t)(idx >> 32) + 0x632be5abu + a ^ (uint32_t)(seed >> 32));
  const uint32_t c = mix32(a + 0x9e3779b9u + b2);
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
  const float u2 = (float)(c >> 8) * (1.0f / 16777216.0f);            // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// y = sign(v) |v|^gamma with v = x + std * n(0,1)   (RandomNoise, then RandomGamma; torchio keeps the sign of negative intensities)
__global__ void __launch_bounds__(256) noise_gamma_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ nstd,
                                                          const float* __restrict__ gamma, int64_t S, uint64_t seed) {
  const int b = blockIdx.y;
  const float sd = nstd[b], g = gamma[b];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[b * S + i] + sd * normal_at(seed, (uint32_t)b, (uint64_t)i);
    y[b * S + i] = copysignf(powf(fabsf(v), g), v);
  }
}

// ---- per-volume mean and 1 / (unbiased standard deviation) ----
__global__ void __launch_bounds__(1024) meanstd_kernel(const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ rstd, int64_t S) {
  const float* v = x + (int64_t)blockIdx.x * S;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = threadIdx.x; i < S; i += blockDim.x) {
    const double t = v[i];
    s1 += t;
    s2 += t * t;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  __shared__ double red[32];
  if ((threadIdx.x & 63) == 0) {
    red[(threadIdx.x >> 6) * 2] = s1;
    red[(threadIdx.x >> 6) * 2 + 1] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = 0.0;
    s2 = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
      s1 += red[2 * i];
      s2 += red[2 * i + 1];
    }
    const double m = s1 / (double)S;
    const double var = (s2 - s1 * m) / (double)(S > 1 ? S - 1 : 1);
    mean[blockIdx.x] = (float)m;
    rstd[blockIdx.x] = (float)(1.0 / fmax(sqrt(fmax(var, 0.0)), 1e-12));
  }
}

__global__ void __launch_bounds__(256) znorm_kernel(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, int64_t S) {
  const int b = blockIdx.y;
  const float m = mean[b], r = rstd[b];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (int64_t)gridDim.x * blockDim.x)
    y[b * S + i] = (x[b * S + i] - m) * r;
}

// ---- RandomSwap: `iters` exchanges of two patches per volume, in order (a later exchange may move voxels an earlier one placed).
//      One block per volume; origins[it][b][2][3]: patch corners (the host makes the two corners of an overlapping draw equal = no-op). ----
__global__ void __launch_bounds__(256) swap_kernel(float* __restrict__ x, const int* __restrict__ origins, int B, int D, int H, int W, int pd, int ph,
                                                   int pw, int iters) {
  const int b = blockIdx.x;
  float* v = x + (int64_t)b * D * H * W;
  const int P = pd * ph * pw;
  for (int it = 0; it < iters; ++it) {
    const int* o = origins + ((int64_t)it * B + b) * 6;
    const int64_t ba = ((int64_t)o[0] * H + o[1]) * W + o[2], bb = ((int64_t)o[3] * H + o[4]) * W + o[5];
    if (ba != bb) {   // block-uniform
      for (int t = threadIdx.x; t < P; t += blockDim.x) {
        const int64_t off = ((int64_t)(t / (ph * pw)) * H + (t / pw) % ph) * W + t % pw;
        const float va = v[ba + off], vb = v[bb + off];
        v[ba + off] = vb;
        v[bb + off] = va;
      }
    }
    __syncthreads();   // workgroup-scope ordering of this exchange before the next one (one CU: same L1)
  }
}

}  // namespace

extern "C" int pcrl_aug_volume_min(const float* x, float* vmin, int B, int64_t S, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && vmin && B > 0 && S > 0, "aug_volume_min: bad arguments");
  hipLaunchKernelGGL(vol_min_kernel, dim3(B), dim3(1024), 0, as_stream(stream), x, vmin, S);
  return pcrl_check_launch("aug_volume_min");
}

extern "C" int pcrl_aug_affine(const float* x, float* y, const float* inv, const int* flip, const float* fill, int B, int D, int H, int W,
                               pcrl_stream_t stream) {
  PCRL_REQUIRE(x && y && inv && flip && fill && x != y, "aug_affine: null pointer or in-place call");
  PCRL_REQUIRE(B > 0 && B <= 65535 && D > 0 && H > 0 && W > 0, "aug_affine: bad shape");
  const int64_t S = (int64_t)D * H * W;
  hipLaunchKernelGGL(affine_kernel, dim3((unsigned)((S + 255) / 256), B), dim3(256), 0, as_stream(stream), x, y, inv, flip, fill, D, H, W);
  return pcrl_check_launch("aug_affine");
}

extern "C" int pcrl_aug_blur_axis(const float* x, float* y, const float* sigma, int B, int D, int H, int W, int axis, int radius,
                                  pcrl_stream_t stream) {
  PCRL_REQUIRE(x && y && sigma && x != y, "aug_blur_axis: null pointer or in-place call");
  PCRL_REQUIRE(B > 0 && B <= 65535 && axis >= 0 && axis < 3 && radius >= 0 && radius <= 32, "aug_blur_axis: bad arguments");
  const int64_t S = (int64_t)D * H * W;
  hipLaunchKernelGGL(blur_axis_kernel, dim3((unsigned)((S + 255) / 256), B), dim3(256), 0, as_stream(stream), x, y, sigma, D, H, W, axis, radius);
  return pcrl_check_launch("aug_blur_axis");
}

extern "C" int pcrl_aug_noise_gamma(const float* x, float* y, const float* noise_std, const float* gamma, int B, int64_t S, int64_t seed,
                                    pcrl_stream_t stream) {
  PCRL_REQUIRE(x && y && noise_std && gamma && B > 0 && B <= 65535 && S > 0, "aug_noise_gamma: bad arguments");
  unsigned gx = (unsigned)((S + 255) / 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(noise_gamma_kernel, dim3(gx, B), dim3(256), 0, as_stream(stream), x, y, noise_std, gamma, S, (uint64_t)seed);
  return pcrl_check_launch("aug_noise_gamma");
}

extern "C" int pcrl_aug_meanstd(const float* x, float* mean, float* rstd, int B, int64_t S, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && mean && rstd && B > 0 && S > 0, "aug_meanstd: bad arguments");
  hipLaunchKernelGGL(meanstd_kernel, dim3(B), dim3(1024), 0, as_stream(stream), x, mean, rstd, S);
  return pcrl_check_launch("aug_meanstd");
}

extern "C" int pcrl_aug_znorm(const float* x, float* y, const float* mean, const float* rstd, int B, int64_t S, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && y && mean && rstd && B > 0 && B <= 65535 && S > 0, "aug_znorm: bad arguments");
  unsigned gx = (unsigned)((S + 255) / 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(znorm_kernel, dim3(gx, B), dim3(256), 0, as_stream(stream), x, y, mean, rstd, S);
  return pcrl_check_launch("aug_znorm");
}

extern "C" int pcrl_aug_swap(float* x, const int* origins, int B, int D, int H, int W, int pd, int ph, int pw, int iters, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && origins && B > 0 && iters >= 0, "aug_swap: bad arguments");
  PCRL_REQUIRE(pd > 0 && ph > 0 && pw > 0 && pd <= D && ph <= H && pw <= W, "aug_swap: patch larger than the volume");
  hipLaunchKernelGGL(swap_kernel, dim3(B), dim3(256), 0, as_stream(stream), x, origins, B, D, H, W, pd, ph, pw, iters);
  return pcrl_check_launch("aug_swap");
}
