// The ResNet stem of the 2D path for gfx950: conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False) of torchvision's
// ResNet-18 as smp.Unet('resnet18', in_channels=3) builds it (models/pcrlv2_model.py:200) -- forward and weight gradient, bf16 MFMA with
// float32 accumulation, reading the loader's float32 NCHW image directly (train_2d.py:139-141: `.float().cuda()` tensors).
//
// The general gather kernel (conv2d.hip) ran this layer on the image zero-padded to 8 channels: K = 49 taps x 8 = 392 (147 real), every
// input pixel re-read once per tap from L2 -- 0.73 ms forward and 0.77 ms weight gradient per 512^2 x 64-image view where HBM needs 0.14.
// Here a block owns an 8 x 32-pixel patch of the OUTPUT; the 21 x 69-pixel input window is staged in LDS once as [row][col][R G B 0] bf16
// (8 bytes per pixel, rows 560 bytes apart), and the seven taps of one kernel row kh are ONE K = 32 MFMA step: k = kw * 4 + c, and the
// K-row of output pixel (oh, ow) is the 64 contiguous bytes that start at input pixel (2 oh + kh - 3, 2 ow - 3) -- no im2col copy, consecutive
// output pixels are 16 bytes apart (k = 28..31, the eighth pixel, meets zero weights).
//   forward: D[co][px] = W[co][kh][k] * X[px][k] over the 7 kernel rows (the weight fragment is the A operand: a lane holds 4 consecutive
//            output channels of one pixel -> 8-byte stores), + BatchNorm statistics per block; blocks walk patch ranges with two LDS windows.
//   weight gradient: dW[kh][k][co] = sum_px X[px][k] * dy[px][co]: both operands fetched with ds_read_b64_tr_b16 from their natural
//            layouts (the X "matrix" has a 16-byte row pitch inside the window); four waves = 2 halves of the output channels x (4 + 3)
//            kernel rows; per-block partial slabs, fixed-order second pass.
#include "common.h"

namespace {

constexpr int PH = 8, PW = 32;                   // output patch
constexpr int WR = 2 * PH + 5, WC = 2 * PW + 6;  // input window: 21 rows x 70 columns (69 used + 1 so that every 64-byte K-row is inside)
constexpr int WPITCH = WC * 8;                   // 560 bytes: a multiple of 16
constexpr int WBYTES = WR * WPITCH;              // 11 760
constexpr int CO = 64, KH = 7;

struct StemParams {
  const float* x;     // [N][3][H][W] float32
  const bf16* w;      // packed [64][7][32] (forward) / unused (weight gradient)
  bf16* y;            // forward: [N][Ho][Wo][64]
  const bf16* dy;     // weight gradient: [N][Ho][Wo][64]
  float* stats;       // forward: [blocks][64][2] or null
  float* ws;          // weight gradient: [blocks][7][32][64]
  int N, H, W, Ho, Wo;
  int npatch, per;
};

// the input window of patch (n, h0, w0) -> registers: pixel q = tid + 256 * i of the 21 x 70 window, three plane reads each
constexpr int WPIX = WR * WC, WP = (WPIX + 255) / 256;   // 1470 pixels, 6 per thread
struct WinRegs {
  uint2 v[WP];
};
__device__ __forceinline__ void win_load(const StemParams& p, int n, int h0, int w0, int tid, WinRegs& r) {
  const int64_t plane = (int64_t)p.H * p.W;
  const float* base = p.x + (int64_t)n * 3 * plane;
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int q = tid + 256 * i;
    const int wr = q / WC, wc = q - wr * WC;
    const int h = 2 * h0 - 3 + wr, w = 2 * w0 - 3 + wc;
    const bool ok = q < WPIX && wc < WC - 1 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
    const int64_t o = ok ? (int64_t)h * p.W + w : 0;
    const float a = base[o], b = base[plane + o], c = base[2 * plane + o];
    union { bf16 h4[4]; uint2 u; } t;
    t.h4[0] = (bf16)(ok ? a : 0.f);
    t.h4[1] = (bf16)(ok ? b : 0.f);
    t.h4[2] = (bf16)(ok ? c : 0.f);
    t.h4[3] = (bf16)0.f;
    r.v[i] = t.u;
  }
}
__device__ __forceinline__ void win_store(char* win, int tid, const WinRegs& r) {
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int q = tid + 256 * i;
    if (q < WPIX) *reinterpret_cast<uint2*>(win + q * 8) = r.v[i];
  }
}
__device__ __forceinline__ void patch_of(const StemParams& p, int pb, int& n, int& h0, int& w0) {
  const int pw = p.Wo / PW, ph = p.Ho / PH;
  w0 = (pb % pw) * PW;
  pb /= pw;
  h0 = (pb % ph) * PH;
  n = pb / ph;
}

__global__ void __launch_bounds__(256, PCRL_OCC2) stem7_fwd_kernel(const StemParams p) {
  __shared__ __attribute__((aligned(16))) char win[2][WBYTES];
  __shared__ __attribute__((aligned(16))) bf16 wS[CO * KH * 32];    // 28 672 bytes
  __shared__ float red[4][CO][2];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int b_beg = blockIdx.x * p.per, b_end = min(b_beg + p.per, p.npatch);
  for (int i = tid; i < CO * KH * 32 / 8; i += 256) reinterpret_cast<uint4*>(wS)[i] = reinterpret_cast<const uint4*>(p.w)[i];
  // pixel part of the K-row address: pixel (row 2 wid + (mf >> 1), column (mf & 1) * 16 + lr) of the patch; k-group lg = 16 bytes
  int poff[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) poff[mf] = (2 * (2 * wid + (mf >> 1))) * WPITCH + ((mf & 1) * 16 + lr) * 16 + lg * 16;
  float s1[4][4], s2[4][4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[nf][r] = s2[nf][r] = 0.f;
  WinRegs wr;
  int n, h0, w0;
  if (b_beg < b_end) {
    patch_of(p, b_beg, n, h0, w0);
    win_load(p, n, h0, w0, tid, wr);
    win_store(win[0], tid, wr);
  }
  __syncthreads();
  int cur = 0;
  for (int pb = b_beg; pb < b_end; ++pb) {
    const bool more = pb + 1 < b_end;
    if (more) {
      patch_of(p, pb + 1, n, h0, w0);
      win_load(p, n, h0, w0, tid, wr);      // the next window lands while this patch is multiplied and stored
    }
    __builtin_amdgcn_sched_barrier(0);
    const char* xs = win[cur];
    f32x4 acc[4][4];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      bf16x8 fw[4], fx[4];
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) fw[nf] = *reinterpret_cast<const bf16x8*>(wS + ((nf * 16 + lr) * KH + kh) * 32 + lg * 8);
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) fx[mf] = *reinterpret_cast<const bf16x8*>(xs + poff[mf] + kh * WPITCH);
#pragma unroll
      for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nf], fx[mf], acc[mf][nf], 0, 0, 0);
    }
    patch_of(p, pb, n, h0, w0);
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
      const int64_t row = ((int64_t)n * p.Ho + h0 + 2 * wid + (mf >> 1)) * p.Wo + w0 + (mf & 1) * 16 + lr;
#pragma unroll
      for (int nf = 0; nf < 4; ++nf) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[mf][nf][r];
          s1[nf][r] += v[r];
          s2[nf][r] += v[r] * v[r];
        }
        *reinterpret_cast<bf16x4*>(p.y + row * CO + nf * 16 + 4 * lg) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
      }
    }
    if (more) win_store(win[cur ^ 1], tid, wr);
    __syncthreads();
    cur ^= 1;
  }
  if (p.stats) {
#pragma unroll
    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[nf][r], b = s2[nf][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        if (lr == 0) {
          red[wid][nf * 16 + 4 * lg + r][0] = a;
          red[wid][nf * 16 + 4 * lg + r][1] = b;
        }
      }
    __syncthreads();
    if (tid < CO) {
      float* o = p.stats + ((int64_t)blockIdx.x * CO + tid) * 2;
      o[0] = (red[0][tid][0] + red[1][tid][0]) + (red[2][tid][0] + red[3][tid][0]);
      o[1] = (red[0][tid][1] + red[1][tid][1]) + (red[2][tid][1] + red[3][tid][1]);
    }
  }
}

// 32 consecutive LDS rows (the K dimension, `pitch` bytes apart) x 16 columns starting at column cb * 16 -> the canonical MFMA operand fragment
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__device__ __forceinline__ bf16x8 tr_frag(const char* base, int row0, int cb, int pitch, int lane) {
  const int g = lane >> 4, jr = (lane & 15) >> 2, cq = lane & 3;
  const char* p0 = base + (row0 + 8 * g + jr) * pitch + (cb * 16 + 4 * cq) * 2;
  union { struct { s16x4 a, b; } s; bf16x8 f; } u;
  u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * pitch));
  return u.f;
}

// dW partial of a block: ws[block][kh][k][co] (k = kw * 4 + c; the k = 3 mod 4 and k >= 28 columns multiply zeros / the neighbouring pixel and
// are dropped by the second pass).  Wave = (half of the output channels) x (kernel rows 0..3 | 4..6).
__global__ void __launch_bounds__(256) stem7_wgrad_kernel(const StemParams p) {
  constexpr int DYB = PH * PW * CO * 2;     // 32 768 bytes
  __shared__ __attribute__((aligned(16))) char win[1][WBYTES];     // one buffer each (32 KB of dy per patch): the next patch waits in registers
  __shared__ __attribute__((aligned(16))) char dyS[1][DYB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int ch = wid & 1, kg = wid >> 1;      // output-channel half, kernel-row group
  const int kh0 = kg ? 4 : 0, nkh = kg ? 3 : 4;
  const int b_beg = blockIdx.x * p.per, b_end = min(b_beg + p.per, p.npatch);
  f32x4 acc[4][2][2];                         // [kernel row of the group][k half][co fragment of the half]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  WinRegs wr;
  u32x4 rdy[8];
  int n, h0, w0;
#define SW_LOAD(pb_)                                                                                         \
  do {                                                                                                       \
    patch_of(p, (pb_), n, h0, w0);                                                                           \
    win_load(p, n, h0, w0, tid, wr);                                                                         \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
      const int q = tid + 256 * i, px = q >> 3, pc = q & 7;                                                  \
      const int64_t row = ((int64_t)n * p.Ho + h0 + (px >> 5)) * p.Wo + w0 + (px & 31);                      \
      rdy[i] = *reinterpret_cast<const u32x4*>(p.dy + row * CO + pc * 8);                                    \
    }                                                                                                        \
  } while (0)
#define SW_STORE(buf_)                                                                                       \
  do {                                                                                                       \
    win_store(win[buf_], tid, wr);                                                                           \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(dyS[buf_] + (tid + 256 * i) * 16) = rdy[i]; \
  } while (0)
  if (b_beg < b_end) {
    SW_LOAD(b_beg);
    SW_STORE(0);
  }
  __syncthreads();
  for (int pb = b_beg; pb < b_end; ++pb) {
    const bool more = pb + 1 < b_end;
    if (more) SW_LOAD(pb + 1);
    __builtin_amdgcn_sched_barrier(0);
    const char* xs = win[0];
    const char* ds = dyS[0];
#pragma unroll
    for (int r = 0; r < PH; ++r) {            // one output row = one 32-pixel K chunk
      bf16x8 fd[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) fd[c] = tr_frag(ds, r * PW, ch * 2 + c, CO * 2, lane);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (a < nkh) {
          // K-rows of output row r at kernel row kh: 32 rows 16 bytes apart starting at window row 2 r + kh
          const char* xb = xs + (2 * r + kh0 + a) * WPITCH;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const bf16x8 fx = tr_frag(xb, 0, b, 16, lane);
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[a][b][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx, fd[c], acc[a][b][c], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();      // everybody is done with this patch's tiles
    if (more) SW_STORE(0);
    __syncthreads();
  }
#undef SW_LOAD
#undef SW_STORE
  // D[i][j]: i = k (16 b + 4 (lane >> 4) + r), j = co (32 ch + 16 c + (lane & 15))
  float* out = p.ws + (int64_t)blockIdx.x * (KH * 32 * CO);
#pragma unroll
  for (int a = 0; a < 4; ++a)
    if (a < nkh)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            out[((kh0 + a) * 32 + 16 * b + 4 * (lane >> 4) + r) * CO + 32 * ch + 16 * c + (lane & 15)] = acc[a][b][c][r];
}

// dw[co][c][kh][kw] = sum over the blocks of ws[block][kh][kw * 4 + c][co]   (fp64, fixed order; 4 slab groups per output)
__global__ void __launch_bounds__(256) stem7_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int blocks) {
  __shared__ double part[4][64];
  const int o = threadIdx.x & 63, zg = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;           // over [kh][kw][c][co] = 7 * 7 * 3 * 64 outputs
  double a0 = 0.0, a1 = 0.0;
  int co = 0, c = 0, kh = 0, kw = 0;
  const bool ok = idx < KH * 7 * 3 * CO;
  if (ok) {
    co = idx % CO;
    int t = idx / CO;
    c = t % 3; t /= 3;
    kw = t % 7; kh = t / 7;
    const float* src = ws + (int64_t)(kh * 32 + kw * 4 + c) * CO + co;
    const int64_t per = (int64_t)KH * 32 * CO;
    int z = zg;
    for (; z + 4 < blocks; z += 8) { a0 += (double)src[(int64_t)z * per]; a1 += (double)src[(int64_t)(z + 4) * per]; }
    for (; z < blocks; z += 4) a0 += (double)src[(int64_t)z * per];
  }
  part[zg][o] = a0 + a1;
  __syncthreads();
  if (zg == 0 && ok) dw[((co * 3 + c) * 7 + kh) * 7 + kw] = (float)((part[0][o] + part[1][o]) + (part[2][o] + part[3][o]));
}

__global__ void __launch_bounds__(256) stem7_pack_kernel(const float* __restrict__ w, bf16* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;     // over [co][kh][k]
  if (i >= CO * KH * 32) return;
  const int k = i & 31, kh = (i >> 5) % KH, co = i / (32 * KH);
  const int kw = k >> 2, c = k & 3;
  out[i] = (bf16)((kw < 7 && c < 3) ? w[((co * 3 + c) * 7 + kh) * 7 + kw] : 0.f);
}

struct StemPlan {
  int blocks, per;
};
StemPlan stem_plan(int64_t npatch, int per_cu) {
  const int64_t nb = npatch < 256 * per_cu ? npatch : 256 * per_cu;
  const int per = (int)((npatch + nb - 1) / nb);
  return StemPlan{(int)((npatch + per - 1) / per), per};
}
bool stem_ok(int N, int H, int W, int dtype) {
  return dtype == PCRL_BF16 && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && (H / 2) % PH == 0 && (W / 2) % PW == 0 &&
         (int64_t)N * H * W < ((int64_t)1 << 31);
}

}  // namespace

extern "C" int64_t pcrl_stem7_ok(int N, int H, int W, int dtype) { return stem_ok(N, H, W, dtype) ? 1 : 0; }
extern "C" int64_t pcrl_stem7_packed_elems(void) { return (int64_t)CO * KH * 32; }
extern "C" int64_t pcrl_stem7_stats_rows(int N, int H, int W) { return stem_plan((int64_t)N * (H / 2 / PH) * (W / 2 / PW), 4).blocks; }
extern "C" size_t pcrl_stem7_wgrad_ws_bytes(int N, int H, int W) {
  return (size_t)stem_plan((int64_t)N * (H / 2 / PH) * (W / 2 / PW), 2).blocks * KH * 32 * CO * sizeof(float);
}

extern "C" int pcrl_stem7_pack(const float* w_ref, void* out, pcrl_stream_t stream) {
  PCRL_REQUIRE(w_ref && out, "stem7_pack: null pointer");
  hipLaunchKernelGGL(stem7_pack_kernel, dim3((CO * KH * 32 + 255) / 256), dim3(256), 0, as_stream(stream), w_ref, (bf16*)out);
  return pcrl_check_launch("stem7_pack");
}

extern "C" int pcrl_stem7_fwd(const float* x, const void* wp, void* y, float* stats, int N, int H, int W, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && wp && y, "stem7_fwd: null pointer");
  PCRL_REQUIRE(stem_ok(N, H, W, dtype), "stem7_fwd: not available for N=%d H=%d W=%d dtype=%d (pcrl_stem7_ok)", N, H, W, dtype);
  const int Ho = H / 2, Wo = W / 2;
  const int64_t npatch = (int64_t)N * (Ho / PH) * (Wo / PW);
  const StemPlan pl = stem_plan(npatch, 4);
  StemParams p{x, (const bf16*)wp, (bf16*)y, nullptr, stats, nullptr, N, H, W, Ho, Wo, (int)npatch, pl.per};
  hipLaunchKernelGGL(stem7_fwd_kernel, dim3(pl.blocks), dim3(256), 0, as_stream(stream), p);
  return pcrl_check_launch("stem7_fwd");
}

extern "C" int pcrl_stem7_wgrad(const float* x, const void* dy, float* dw_ref, void* ws, size_t ws_bytes, int N, int H, int W, int dtype,
                                pcrl_stream_t stream) {
  PCRL_REQUIRE(x && dy && dw_ref, "stem7_wgrad: null pointer");
  PCRL_REQUIRE(stem_ok(N, H, W, dtype), "stem7_wgrad: not available for N=%d H=%d W=%d dtype=%d (pcrl_stem7_ok)", N, H, W, dtype);
  if (!ws || ws_bytes < pcrl_stem7_wgrad_ws_bytes(N, H, W)) return pcrl_fail(PCRL_EWORKSPACE, "stem7_wgrad: workspace too small");
  const int Ho = H / 2, Wo = W / 2;
  const int64_t npatch = (int64_t)N * (Ho / PH) * (Wo / PW);
  const StemPlan pl = stem_plan(npatch, 2);
  StemParams p{x, nullptr, nullptr, (const bf16*)dy, nullptr, (float*)ws, N, H, W, Ho, Wo, (int)npatch, pl.per};
  hipLaunchKernelGGL(stem7_wgrad_kernel, dim3(pl.blocks), dim3(256), 0, as_stream(stream), p);
  if (int e = pcrl_check_launch("stem7_wgrad")) return e;
  hipLaunchKernelGGL(stem7_wgrad_reduce_kernel, dim3((KH * 7 * 3 * CO + 63) / 64), dim3(256), 0, as_stream(stream), (const float*)ws, dw_ref, pl.blocks);
  return pcrl_check_launch("stem7_wgrad_reduce");
}
