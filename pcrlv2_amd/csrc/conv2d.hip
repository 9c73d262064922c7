// 2D convolutions of the PCRLv2 (ResNet-18 U-Net) path as implicit GEMMs on MFMA for gfx950.
//
// Replaces aten::convolution / convolution_backward(input) dispatched from the reference's 2D model: the smp/torchvision
// ResNet-18 encoder (7x7 s2 stem, 3x3 s1/s2 BasicBlock convs, 1x1 s2 downsample) and models/pcrlv2_model.py:68-128
// (DecoderBlock: nearest x2 -> Conv2dReLU x2, deep-supervision head conv3x3 + conv1x1, segmentation head conv3x3).
//
// ONE gather kernel covers every geometry.  GEMM view: rows = output pixels (M = N*Ho*Wo), columns = output channels,
// K = (kh, kw, source channel) flattened, k = tap * Cs + c with Cs a power of two >= 8 (the host zero-pads 3-channel
// tensors to 8).  Activations are NHWC, so a 16-byte slot of an A-tile row is 8 (bf16) / 4 (fp32) consecutive channels of ONE
// tap: every thread decodes the tap of ITS slot (shift/mask + a multiply-shift for tap -> kh, kw), which makes K-steps that
// straddle taps (Cs = 8, 16) as cheap as the 32-channel case; K is padded to a multiple of 32 with zero weights.
//   FWD   : source pixel (oh*s - p + kh, ow*s - p + kw); `up` = the source is read through a nearest x2 upsample
//           (F.interpolate(scale_factor=2, mode="nearest") of DecoderBlock.forward fused into the gather: ih>>1, iw>>1)
//   DGRAD : source = dy, pixel ((ih + p - kh)/s, (iw + p - kw)/s) when divisible; weights packed [ci][tap FLIPPED][co], i.e. the
//           stride-1 data gradient is literally a forward convolution with these weights (what the brick kernel runs)
// Stride-1 3x3 convolutions whose batch / extent / channel counts fit (N % 4, H % 8, W % 8, channels % 32, bf16) go to the
// LDS-halo brick kernel instead (conv_brick.hip, KD = 1: the image index plays the role of depth), forward and data gradient.
// Tile 128 x BN (BN = 32/64/128) as in conv_igemm.hip: 4 waves 2x2, LDS double-buffered and XOR-swizzled, register-staged
// (padding needs zero-fill), two staging register sets.  Epilogue: + bias, store (T or float32, columns >= Nc masked), and the
// per-tile (sum, sum^2) rows for the training-mode BatchNorm2d that follows.
#include "common.h"
#include "mma_tile.h"
#include <atomic>
#include <cstdlib>

namespace {

enum { C2_FWD = 0, C2_DGRAD = 1 };
#ifndef C2_IDX
#define C2_IDX int64_t   // element-offset type of the gather addresses (experiment: -DC2_IDX=int)
#endif

struct C2Params {
  const void* x;      // source rows [N*Hs*Ws][Cs]
  const void* w;      // packed weights [NcP][Kpad]
  const float* bias;  // [Nc] or null
  void* y;            // [M][Nc]  (T, or float when out_f32)
  float* stats;       // [gridDim.x][Nc][2] or null
  int N, Hs, Ws;      // source dims as stored
  int Ho, Wo;         // GEMM row space
  int Cs, cs_shift;
  int Nc;
  int KH, KW, kw_mul;  // kh = (tap * kw_mul) >> 16
  int stride, pad, up;
  int Ktot, Kpad;
  int out_f32;
  int placed;          // output row (n, oh, ow) is stored at pixel (oh * 2 + pa, ow * 2 + pb) of an [N][Hf][Wf] tensor (stride-2 data gradient)
  int pa, pb, Hf, Wf;
  int64_t M;
};


// NSM > 0: burst variant for layers with at most NSM K-steps (16-channel layers at full resolution, the 1x1 / 3-channel heads): all
// K-steps are loaded up front into NSM register sets, so a block waits for global memory once instead of once per step -- these
// launches are 131 072 blocks of a few hundred MFMA cycles each and were bound by exactly that latency chain.
template <typename T, int BN, int MODE, int NSM = 0>
__global__ void __launch_bounds__(256, PCRL_OCC2) conv2d_kernel(const C2Params p) {
  constexpr int BM = PCRL_CONV_BM;
  using TL = Tile<T>;
  using MM = Mma<T>;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int SLOTS = TL::SLOTS;
  constexpr int RPP = 256 / SLOTS;
  constexpr int AP = BM / RPP;
  constexpr int BP = (BN + RPP - 1) / RPP;
  constexpr int FM = 4, FN = BN / 32;
  constexpr int A_BYTES = BM * TL::ROWB, B_BYTES = BN * TL::ROWB;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  char* Bs = smem + 2 * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.w);
  const int slot = tid % SLOTS, rowp = tid / SLOTS;
  const int Cs = p.Cs;
  // logical source extent (through the nearest upsample when `up`)
  const int Hl = p.up ? 2 * p.Hs : p.Hs, Wl = p.up ? 2 * p.Ws : p.Ws;
  const int sshift = p.stride - 1;   // stride is 1 or 2

  // ---- per-thread A rows ----
  C2_IDX nbase[AP];   // row of source pixel (n, 0, 0); -1 marks a dead row
  int oy[AP], ox[AP];
#pragma unroll
  for (int ps = 0; ps < AP; ++ps) {
    const int64_t m = m0 + ps * RPP + rowp;
    nbase[ps] = -1;
    oy[ps] = ox[ps] = 0;
    if (m < p.M) {
      const int ow = (int)(m % p.Wo);
      const int64_t t = m / p.Wo;
      const int oh = (int)(t % p.Ho), n = (int)(t / p.Ho);
      nbase[ps] = (C2_IDX)n * p.Hs * p.Ws;
      if (MODE == C2_FWD) {
        oy[ps] = oh * p.stride - p.pad;
        ox[ps] = ow * p.stride - p.pad;
      } else {
        oy[ps] = oh + p.pad;
        ox[ps] = ow + p.pad;
      }
    }
  }
  // ---- per-thread B rows ----
  C2_IDX boff[BP];
  bool bok[BP];
#pragma unroll
  for (int ps = 0; ps < BP; ++ps) {
    const int brow = ps * RPP + rowp;
    bok[ps] = brow < BN;
    boff[ps] = (C2_IDX)(n0 + (bok[ps] ? brow : 0)) * p.Kpad + slot * VEC;
  }

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int S = p.Kpad / 32;
  u32x4 raA[AP], rbA[BP], raB[AP], rbB[BP];
  uint32_t aokA = 0, aokB = 0;

  // Loads are unconditional from clamped addresses (see conv_igemm.hip); validity is applied at the LDS store.
#define C2_LOAD(s_, ra, rb, aok)                                                                        \
  do {                                                                                                  \
    const int k_ = (s_)*32 + slot * VEC;                                                                \
    const int tap_ = k_ >> p.cs_shift, c_ = k_ & (Cs - 1);                                              \
    const int kh_ = (tap_ * p.kw_mul) >> 16, kw_ = tap_ - kh_ * p.KW;                                   \
    const bool kok_ = k_ < p.Ktot;                                                                      \
    aok = 0;                                                                                            \
    _Pragma("unroll") for (int ps = 0; ps < AP; ++ps) {                                                 \
      int ih_, iw_;                                                                                     \
      bool ok_ = kok_ && nbase[ps] >= 0;                                                                \
      if (MODE == C2_FWD) {                                                                             \
        ih_ = oy[ps] + kh_;                                                                             \
        iw_ = ox[ps] + kw_;                                                                             \
        ok_ = ok_ && (unsigned)ih_ < (unsigned)Hl && (unsigned)iw_ < (unsigned)Wl;                      \
        if (p.up) {                                                                                     \
          ih_ >>= 1;                                                                                    \
          iw_ >>= 1;                                                                                    \
        }                                                                                               \
      } else {                                                                                          \
        const int th_ = oy[ps] - (p.KH - 1 - kh_), tw_ = ox[ps] - (p.KW - 1 - kw_); /* taps are packed flipped */ \
        ih_ = th_ >> sshift;                                                                            \
        iw_ = tw_ >> sshift;                                                                            \
        ok_ = ok_ && th_ >= 0 && tw_ >= 0 && (ih_ << sshift) == th_ && (iw_ << sshift) == tw_ && ih_ < p.Hs && iw_ < p.Ws; \
      }                                                                                                 \
      const C2_IDX row_ = ok_ ? nbase[ps] + (C2_IDX)ih_ * p.Ws + iw_ : (C2_IDX)0;                       \
      ra[ps] = *reinterpret_cast<const u32x4*>(X + row_ * Cs + (ok_ ? c_ : 0));                         \
      aok |= (uint32_t)ok_ << ps;                                                                       \
    }                                                                                                   \
    _Pragma("unroll") for (int ps = 0; ps < BP; ++ps)                                                   \
      rb[ps] = *reinterpret_cast<const u32x4*>(Wp + boff[ps] + (C2_IDX)(s_)*32);                        \
  } while (0)

#define C2_STORE(buf_, ra, rb, aok)                                                                     \
  do {                                                                                                  \
    _Pragma("unroll") for (int ps = 0; ps < AP; ++ps)                                                   \
      *reinterpret_cast<u32x4*>(As + (buf_)*A_BYTES + TL::off(ps * RPP + rowp, slot)) =                 \
          keep_if((aok >> ps) & 1u, ra[ps]);                                                            \
    _Pragma("unroll") for (int ps = 0; ps < BP; ++ps) {                                                 \
      if (bok[ps]) *reinterpret_cast<u32x4*>(Bs + (buf_)*B_BYTES + TL::off(ps * RPP + rowp, slot)) = rb[ps]; \
    }                                                                                                   \
  } while (0)

#define C2_COMPUTE(cur_)                                                                                \
  do {                                                                                                  \
    const char* a = As + (cur_)*A_BYTES;                                                                \
    const char* b = Bs + (cur_)*B_BYTES;                                                                \
    typename MM::Frag fa[FM], fb[FN];                                                                   \
    _Pragma("unroll") for (int i = 0; i < FM; ++i) fa[i] = MM::read(a, wm * 64 + i * 16 + lr, lg);      \
    _Pragma("unroll") for (int j = 0; j < FN; ++j) fb[j] = MM::read(b, wn * (BN / 2) + j * 16 + lr, lg); \
    _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                      \
      _Pragma("unroll") for (int j = 0; j < FN; ++j) MM::mma(fa[i], fb[j], acc[i][j]);                  \
  } while (0)

#define C2_CLAMP(s_) ((s_) < S ? (s_) : S - 1)

  if (NSM > 0) {
    constexpr int NS_ = NSM > 0 ? NSM : 1;
    u32x4 ra_[NS_][AP], rb_[NS_][BP];
    uint32_t aok_[NS_];
#pragma unroll
    for (int s = 0; s < NS_; ++s) C2_LOAD(C2_CLAMP(s), ra_[s], rb_[s], aok_[s]);   // unconditional (a step past S repeats the last one, unused)
    C2_STORE(0, ra_[0], rb_[0], aok_[0]);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS_; ++s) {
      if (s < S) {   // block-uniform
        if (s + 1 < NS_ && s + 1 < S) C2_STORE((s + 1) & 1, ra_[s + 1 < NS_ ? s + 1 : 0], rb_[s + 1 < NS_ ? s + 1 : 0], aok_[s + 1 < NS_ ? s + 1 : 0]);
        if (s & 1) C2_COMPUTE(1);
        else C2_COMPUTE(0);
        __syncthreads();
      }
    }
  } else {
  C2_LOAD(0, raA, rbA, aokA);
  C2_STORE(0, raA, rbA, aokA);
  C2_LOAD(C2_CLAMP(1), raA, rbA, aokA);
  __syncthreads();
  for (int s = 0; s < S; s += 2) {
    C2_LOAD(C2_CLAMP(s + 2), raB, rbB, aokB);
    __builtin_amdgcn_sched_barrier(0);
    C2_COMPUTE(0);
    __builtin_amdgcn_sched_barrier(0);
    C2_STORE(1, raA, rbA, aokA);
    __syncthreads();
    if (s + 1 >= S) break;
    C2_LOAD(C2_CLAMP(s + 3), raA, rbA, aokA);
    __builtin_amdgcn_sched_barrier(0);
    C2_COMPUTE(1);
    __builtin_amdgcn_sched_barrier(0);
    C2_STORE(0, raB, rbB, aokB);
    __syncthreads();
  }
  }
#undef C2_LOAD
#undef C2_STORE
#undef C2_COMPUTE
#undef C2_CLAMP

  // ---- epilogue ----
  T* __restrict__ Y = reinterpret_cast<T*>(p.y);
  float* __restrict__ Yf = reinterpret_cast<float*>(p.y);
  float s1[FN], s2[FN], bv[FN];
  bool cok[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int col = n0 + wn * (BN / 2) + j * 16 + lr;
    cok[j] = col < p.Nc;
    s1[j] = 0.f;
    s2[j] = 0.f;
    bv[j] = (p.bias && cok[j]) ? p.bias[col] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + wm * 64 + i * 16 + lg * 4 + r;
      if (m < p.M) {
        int64_t orow = m;
        if (p.placed) {   // block-uniform
          const int ow = (int)(m % p.Wo);
          const int64_t t = m / p.Wo;
          const int oh = (int)(t % p.Ho);
          orow = ((t / p.Ho) * p.Hf + oh * 2 + p.pa) * p.Wf + ow * 2 + p.pb;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if (cok[j]) {
            const float v = acc[i][j][r] + bv[j];
            const int64_t o = orow * p.Nc + n0 + wn * (BN / 2) + j * 16 + lr;
            if (p.out_f32) Yf[o] = v;
            else Y[o] = from_f<T>(v);
            s1[j] += v;
            s2[j] += v * v;
          }
        }
      }
    }
  }
  if (p.stats) {
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float a = s1[j], b = s2[j];
      a += __shfl_xor(a, 16, 64);
      b += __shfl_xor(b, 16, 64);
      a += __shfl_xor(a, 32, 64);
      b += __shfl_xor(b, 32, 64);
      if (lg == 0) {
        red[((wid * FN + j) * 16 + lr) * 2 + 0] = a;
        red[((wid * FN + j) * 16 + lr) * 2 + 1] = b;
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Nc) {
      const int wn_ = tid / (BN / 2), within = tid % (BN / 2);
      const int j = within / 16, l = within % 16;
      const float* r0 = red + (((0 * 2 + wn_) * FN + j) * 16 + l) * 2;
      const float* r1 = red + (((1 * 2 + wn_) * FN + j) * 16 + l) * 2;
      float* o = p.stats + ((int64_t)blockIdx.x * p.Nc + n0 + tid) * 2;
      o[0] = r0[0] + r1[0];
      o[1] = r0[1] + r1[1];
    }
  }
}

// Packing: reference weight [Co][Ci][KH][KW] float32 -> K-contiguous rows in T, zero padded.
//   mode 0 (forward): out[co][tap * CsP + ci],  rows = round_up(Co, 32), CsP = padded source channels (Ci)
//   mode 1 (dgrad)  : out[ci][(taps-1-tap) * CsP + co],  rows = round_up(Ci, 32), CsP = padded source channels (Co); taps flipped
template <typename T>
__global__ void __launch_bounds__(256) pack_conv2d_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int taps, int CsP,
                                                          int rowsP, int Kpad, int mode) {
  const int64_t total = (int64_t)rowsP * Kpad;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int row = (int)(idx / Kpad), k = (int)(idx % Kpad);
    const int tap = k / CsP, c = k % CsP;
    float v = 0.f;
    if (tap < taps) {
      const int co = mode ? c : row, ci = mode ? row : c;
      if (co < Co && ci < Ci) v = w[((int64_t)co * Ci + ci) * taps + (mode ? taps - 1 - tap : tap)];
    }
    out[idx] = from_f<T>(v);
  }
}

// Parity-class packing for the stride-2 data gradient (see pcrl_conv2d_dgrad_s2): rows ci, k = (kh' * KWc + kw') * CsP + co with
// kh' -> original kh through khmap (3x3: class 0 = {1}, class 1 = {2, 0}; 1x1: {0}).
template <typename T>
__global__ void __launch_bounds__(256) pack_conv2d_s2_kernel(const float* __restrict__ w, T* __restrict__ out, int Co, int Ci, int KH, int KW, int CsP,
                                                             int rowsP, int Kpad, int KHc, int KWc, int kh0, int kh1, int kw0, int kw1) {
  const int64_t total = (int64_t)rowsP * Kpad;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int ci = (int)(idx / Kpad), k = (int)(idx % Kpad);
    const int tap = k / CsP, co = k % CsP;
    float v = 0.f;
    if (tap < KHc * KWc && ci < Ci && co < Co) {
      const int khc = tap / KWc, kwc = tap % KWc;
      const int kh = khc ? kh1 : kh0, kw = kwc ? kw1 : kw0;
      v = w[(((int64_t)co * Ci + ci) * KH + kh) * KW + kw];
    }
    out[idx] = from_f<T>(v);
  }
}

int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

template <typename T, int MODE> int launch_bn(const C2Params& p, int NcP, hipStream_t stream) {
  using TL = Tile<T>;
  const unsigned gx = (unsigned)((p.M + PCRL_CONV_BM - 1) / PCRL_CONV_BM);
  // 128-column tiles (two waves per SIMD) unless the grid they give is small: below 512 blocks the 64-column tile (four waves per SIMD, twice the
  // blocks) is faster -- measured on the local views' 12^2 / 6^2 / 3^2 maps (49 -> 45, 52 -> 48, 86 -> 69 us), slower on the large grids (90 -> 105 us)
  if (NcP % 128 == 0 && (int64_t)gx * (NcP / 128) >= 512) {
    hipLaunchKernelGGL((conv2d_kernel<T, 128, MODE>), dim3(gx, NcP / 128), dim3(256), 2 * (size_t)(PCRL_CONV_BM + 128) * TL::ROWB, stream, p);
  } else if (NcP % 64 == 0) {
    hipLaunchKernelGGL((conv2d_kernel<T, 64, MODE>), dim3(gx, NcP / 64), dim3(256), 2 * (size_t)(PCRL_CONV_BM + 64) * TL::ROWB, stream, p);
  } else {
    const int S = p.Kpad / 32;
    const size_t lds = 2 * (size_t)(PCRL_CONV_BM + 32) * TL::ROWB;
    if (S <= 1) hipLaunchKernelGGL((conv2d_kernel<T, 32, MODE, 1>), dim3(gx, NcP / 32), dim3(256), lds, stream, p);
    else if (S <= 3) hipLaunchKernelGGL((conv2d_kernel<T, 32, MODE, 3>), dim3(gx, NcP / 32), dim3(256), lds, stream, p);
    else if (S <= 5) hipLaunchKernelGGL((conv2d_kernel<T, 32, MODE, 5>), dim3(gx, NcP / 32), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((conv2d_kernel<T, 32, MODE>), dim3(gx, NcP / 32), dim3(256), lds, stream, p);
  }
  return pcrl_check_launch("conv2d");
}

int conv2d_common(const char* what, int mode, const void* src, const void* wp, const float* bias, void* out, float* stats, int N, int Hs, int Ws,
                  int Cs, int Ho, int Wo, int Nc, int KH, int KW, int stride, int pad, int up, int out_f32, int dtype, hipStream_t stream,
                  int placed = 0, int pa = 0, int pb = 0, int Hf = 0, int Wf = 0) {
  PCRL_REQUIRE(src && wp && out, "%s: null pointer", what);
  PCRL_REQUIRE(N > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0 && Nc > 0, "%s: bad dims", what);
  PCRL_REQUIRE(dtype == PCRL_F32 || dtype == PCRL_BF16, "%s: bad dtype %d", what, dtype);
  const int sh = ilog2_exact(Cs);
  PCRL_REQUIRE(sh >= 3, "%s: source channels must be a power of two >= 8 (got %d; zero-pad on the host)", what, Cs);
  PCRL_REQUIRE(KH >= 1 && KW >= 1 && KH * KW <= 49 && stride >= 1 && stride <= 2 && pad >= 0 && pad < KH, "%s: bad kernel geometry", what);
  PCRL_REQUIRE(!(up && (mode == C2_DGRAD || stride != 1)), "%s: the fused nearest upsample needs a stride-1 forward", what);
  C2Params p;
  p.x = src; p.w = wp; p.bias = bias; p.y = out; p.stats = stats;
  p.N = N; p.Hs = Hs; p.Ws = Ws; p.Ho = Ho; p.Wo = Wo;
  p.Cs = Cs; p.cs_shift = sh; p.Nc = Nc;
  p.KH = KH; p.KW = KW; p.kw_mul = (65536 + KW - 1) / KW;
  p.stride = stride; p.pad = pad; p.up = up;
  p.Ktot = KH * KW * Cs;
  p.Kpad = (p.Ktot + 31) / 32 * 32;
  p.out_f32 = out_f32;
  p.placed = placed; p.pa = pa; p.pb = pb; p.Hf = Hf; p.Wf = Wf;
  p.M = (int64_t)N * Ho * Wo;
  const int NcP = (Nc + 31) / 32 * 32;
  if (dtype == PCRL_BF16) return mode == C2_FWD ? launch_bn<bf16, C2_FWD>(p, NcP, stream) : launch_bn<bf16, C2_DGRAD>(p, NcP, stream);
  return mode == C2_FWD ? launch_bn<float, C2_FWD>(p, NcP, stream) : launch_bn<float, C2_DGRAD>(p, NcP, stream);
}

}  // namespace

// LDS-halo brick kernel (conv_brick.hip, KD = 1)
bool pcrl_brick_conv2d_eligible(int N, int H, int W, int Ci, int Co, int dtype);
int64_t pcrl_brick_conv2d_rows(int N, int H, int W);
int pcrl_brick_conv2d_launch(const void* x, const void* wp, const float* bias, void* y, float* stats, int N, int H, int W, int Ci, int Co, int up,
                             hipStream_t stream);
// wide-brick LDS-DMA kernel (conv_brick16.h, MODE 3): 4 images x 8 x 16 pixels per block, no upsampled source
bool pcrl_brick16_conv2d_eligible(int N, int H, int W, int Ci, int Co, int dtype);
int64_t pcrl_brick16_conv2d_rows(int N, int H, int W);
int pcrl_brick16_conv2d_launch(const void* x, const void* wp, const float* bias, void* y, float* stats, int N, int H, int W, int Ci, int Co, hipStream_t stream);
// right-sized kernel for layers with <= 32 channels on both sides (conv2d_narrow.hip)
bool pcrl_conv2d_narrow_eligible(int N, int H, int W, int Cs, int Nc, int ks, int dtype);
int64_t pcrl_conv2d_narrow_rows(int N, int H, int W);
int pcrl_conv2d_narrow_launch(const void* x, const void* wp, const float* bias, void* y, float* stats, int N, int H, int W, int Cs, int Nc, int ks,
                              int up, int out_f32, int red2, hipStream_t stream);
static std::atomic<int> g_conv2d_impl{0};   // 0 = auto (wide brick / brick / narrow kernels where eligible), 1 = always the gather kernel, 2 = auto without the wide brick (tests, A/B)
static inline bool auto_impl() { return g_conv2d_impl == 0 || g_conv2d_impl == 2; }
extern "C" void pcrl_debug_set_conv2d_impl(int impl) { g_conv2d_impl = impl; }

extern "C" int64_t pcrl_conv2d_packed_elems(int rows, int taps, int Cs) {
  return (int64_t)((rows + 31) / 32 * 32) * ((taps * Cs + 31) / 32 * 32);
}

extern "C" int pcrl_conv2d_pack(const float* w_ref, void* out, int Co, int Ci, int KH, int KW, int CsP, int mode, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(w_ref && out && Co > 0 && Ci > 0 && KH > 0 && KW > 0, "conv2d_pack: bad arguments");
  PCRL_REQUIRE(mode == 0 || mode == 1, "conv2d_pack: mode must be 0 (forward) or 1 (data gradient)");
  PCRL_REQUIRE(CsP >= (mode ? Co : Ci), "conv2d_pack: padded channel count %d too small", CsP);
  const int taps = KH * KW, rows = mode ? Ci : Co, rowsP = (rows + 31) / 32 * 32, Kpad = (taps * CsP + 31) / 32 * 32;
  const int64_t total = (int64_t)rowsP * Kpad;
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(pack_conv2d_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), w_ref, (bf16*)out, Co, Ci, taps, CsP, rowsP, Kpad, mode);
  else if (dtype == PCRL_F32)
    hipLaunchKernelGGL(pack_conv2d_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), w_ref, (float*)out, Co, Ci, taps, CsP, rowsP, Kpad, mode);
  else
    return pcrl_fail(PCRL_EINVAL, "conv2d_pack: bad dtype %d", dtype);
  return pcrl_check_launch("conv2d_pack");
}

extern "C" int64_t pcrl_conv2d_stats_rows(int N, int Ho, int Wo) { return ((int64_t)N * Ho * Wo + PCRL_CONV_BM - 1) / PCRL_CONV_BM; }

// which kernel the forward dispatcher picks: 0 gather, 1 LDS-halo brick (4 x 8 x 8), 2 right-sized narrow kernel, 3 wide brick (4 x 8 x 16, LDS-DMA)
// The 32 -> 32 channel layers (decoder block 3 at 256^2) go to the right-sized narrow kernel rather than the brick kernel: 204-219 -> 168 us per
// launch on the same box once the narrow kernel runs two waves per SIMD (PCRL_OCC2).
static bool narrow_first(int Cs, int Nc) { return Cs <= 32 && Nc <= 32; }
static int conv2d_fwd_kind(int N, int Ho, int Wo, int CiP, int Co, int KH, int KW, int stride, int pad, int out_f32, int dtype, int up) {
  if (auto_impl() && narrow_first(CiP, Co) && KH == KW && (KH == 1 || KH == 3) && stride == 1 && pad == (KH - 1) / 2 && pcrl_conv2d_narrow_eligible(N, Ho, Wo, CiP, Co, KH, dtype))
    return 2;
  if (g_conv2d_impl == 0 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !out_f32 && !up && pcrl_brick16_conv2d_eligible(N, Ho, Wo, CiP, Co, dtype)) return 3;
  if (auto_impl() && KH == 3 && KW == 3 && stride == 1 && pad == 1 && !out_f32 && pcrl_brick_conv2d_eligible(N, Ho, Wo, CiP, Co, dtype)) return 1;
  if (auto_impl() && KH == KW && (KH == 1 || KH == 3) && stride == 1 && pad == (KH - 1) / 2 && pcrl_conv2d_narrow_eligible(N, Ho, Wo, CiP, Co, KH, dtype))
    return 2;
  return 0;
}
// rows of [Co][2] statistics the forward WRITES for this geometry (the gather kernel: one per 128 output pixels; the brick kernel: one per
// 256-pixel brick; the narrow kernel: one per block) -- what the caller hands to pcrl_bn_finalize
extern "C" int64_t pcrl_conv2d_fwd_stats_rows(int N, int Hi, int Wi, int CiP, int Co, int KH, int KW, int stride, int pad, int up, int out_f32, int dtype) {
  const int Hl = up ? 2 * Hi : Hi, Wl = up ? 2 * Wi : Wi;
  const int Ho = (Hl + 2 * pad - KH) / stride + 1, Wo = (Wl + 2 * pad - KW) / stride + 1;
  const int kind = conv2d_fwd_kind(N, Ho, Wo, CiP, Co, KH, KW, stride, pad, out_f32, dtype, up);
  return kind == 3 ? pcrl_brick16_conv2d_rows(N, Ho, Wo) : kind == 1 ? pcrl_brick_conv2d_rows(N, Ho, Wo) : kind == 2 ? pcrl_conv2d_narrow_rows(N, Ho, Wo) : pcrl_conv2d_stats_rows(N, Ho, Wo);
}

// which kernel pcrl_conv2d_fwd / pcrl_conv2d_dgrad run for a geometry: 0 gather implicit GEMM, 1 LDS-halo brick kernel, 2 right-sized narrow kernel
// (bench.py's roofline classification; no launch)
extern "C" int64_t pcrl_conv2d_fwd_kind(int N, int Hi, int Wi, int CiP, int Co, int KH, int KW, int stride, int pad, int up, int out_f32, int dtype) {
  const int Hl = up ? 2 * Hi : Hi, Wl = up ? 2 * Wi : Wi;
  const int Ho = (Hl + 2 * pad - KH) / stride + 1, Wo = (Wl + 2 * pad - KW) / stride + 1;
  return conv2d_fwd_kind(N, Ho, Wo, CiP, Co, KH, KW, stride, pad, out_f32, dtype, up);
}
extern "C" int64_t pcrl_conv2d_dgrad_kind(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int CoP, int KH, int KW, int stride, int pad, int dtype) {
  if (auto_impl() && narrow_first(CoP, Ci) && KH == KW && (KH == 1 || KH == 3) && stride == 1 && pad == (KH - 1) / 2 && Hi == Ho && Wi == Wo &&
      pcrl_conv2d_narrow_eligible(N, Hi, Wi, CoP, Ci, KH, dtype))
    return 2;
  if (g_conv2d_impl == 0 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && Hi == Ho && Wi == Wo && pcrl_brick16_conv2d_eligible(N, Hi, Wi, CoP, Ci, dtype)) return 3;
  if (auto_impl() && KH == 3 && KW == 3 && stride == 1 && pad == 1 && Hi == Ho && Wi == Wo && pcrl_brick_conv2d_eligible(N, Hi, Wi, CoP, Ci, dtype)) return 1;
  if (auto_impl() && KH == KW && (KH == 1 || KH == 3) && stride == 1 && pad == (KH - 1) / 2 && Hi == Ho && Wi == Wo &&
      pcrl_conv2d_narrow_eligible(N, Hi, Wi, CoP, Ci, KH, dtype))
    return 2;
  return 0;
}

// stats_rows: rows the caller allocated for stats_partial (>= pcrl_conv2d_fwd_stats_rows(...); rows beyond the written ones are zero-filled)
extern "C" int pcrl_conv2d_fwd(const void* x, const void* wp, const float* bias, void* y, float* stats_partial, int64_t stats_rows, int N, int Hi, int Wi,
                               int CiP, int Co, int KH, int KW, int stride, int pad, int up, int out_f32, int dtype, pcrl_stream_t stream) {
  const int Hl = up ? 2 * Hi : Hi, Wl = up ? 2 * Wi : Wi;
  const int Ho = (Hl + 2 * pad - KH) / stride + 1, Wo = (Wl + 2 * pad - KW) / stride + 1;
  int kind = (x && wp) ? conv2d_fwd_kind(N, Ho, Wo, CiP, Co, KH, KW, stride, pad, out_f32, dtype, up) : 0;
  if (!y) {
    // y = NULL: the per-row statistics only -- a deep-supervision head's convolution whose map nothing reads runs for its BatchNorm's running
    // statistics alone (pcrlv2_model.py:103-106 / train_2d.py:143-168).  Served by the narrow kernel (pcrl_conv2d_fwd_stats_only_ok).
    PCRL_REQUIRE(x && wp && stats_partial && kind == 2, "conv2d_fwd: y = NULL (statistics only) is not available for this geometry (pcrl_conv2d_fwd_stats_only_ok)");
  }
  if (stats_partial) {
    const int64_t rw = kind == 3 ? pcrl_brick16_conv2d_rows(N, Ho, Wo) : kind == 1 ? pcrl_brick_conv2d_rows(N, Ho, Wo) : kind == 2 ? pcrl_conv2d_narrow_rows(N, Ho, Wo) : pcrl_conv2d_stats_rows(N, Ho, Wo);
    PCRL_REQUIRE(stats_rows >= rw, "conv2d_fwd: %lld statistics rows allocated, %lld needed (pcrl_conv2d_fwd_stats_rows)", (long long)stats_rows, (long long)rw);
    if (stats_rows > rw) (void)hipMemsetAsync(stats_partial + rw * Co * 2, 0, (size_t)(stats_rows - rw) * Co * 2 * sizeof(float), as_stream(stream));
  }
  if (kind == 3) return pcrl_brick16_conv2d_launch(x, wp, bias, y, stats_partial, N, Ho, Wo, CiP, Co, as_stream(stream));
  if (kind == 1) return pcrl_brick_conv2d_launch(x, wp, bias, y, stats_partial, N, Ho, Wo, CiP, Co, up, as_stream(stream));
  if (kind == 2) return pcrl_conv2d_narrow_launch(x, wp, bias, y, stats_partial, N, Ho, Wo, CiP, Co, KH, up, out_f32, 0, as_stream(stream));
  return conv2d_common("conv2d_fwd", C2_FWD, x, wp, bias, y, stats_partial, N, Hi, Wi, CiP, Ho, Wo, Co, KH, KW, stride, pad, up, out_f32, dtype,
                       as_stream(stream));
}

extern "C" int64_t pcrl_conv2d_fwd_stats_only_ok(int N, int Hi, int Wi, int CiP, int Co, int KH, int KW, int stride, int pad, int up, int out_f32, int dtype) {
  const int Hl = up ? 2 * Hi : Hi, Wl = up ? 2 * Wi : Wi;
  const int Ho = (Hl + 2 * pad - KH) / stride + 1, Wo = (Wl + 2 * pad - KW) / stride + 1;
  return conv2d_fwd_kind(N, Ho, Wo, CiP, Co, KH, KW, stride, pad, out_f32, dtype, up) == 2 ? 1 : 0;
}

// Data gradient of a 3x3 / stride 1 / pad 1 convolution that read its input through the nearest x2 upsample (decoder conv1,
// models/pcrlv2_model.py:114), WITH the upsample's backward: dx[N][Hc][Wc][Ci] = 2 x 2 block sums of the fine-resolution data gradient,
// which is never stored.  Available (pcrl_conv2d_dgrad_up_ok) where the right-sized narrow kernel takes the fine-resolution problem.
extern "C" int64_t pcrl_conv2d_dgrad_up_ok(int N, int Hc, int Wc, int Ci, int CoP, int dtype) {
  return (auto_impl() && pcrl_conv2d_narrow_eligible(N, 2 * Hc, 2 * Wc, CoP, Ci, 3, dtype)) ? 1 : 0;
}
extern "C" int pcrl_conv2d_dgrad_up(const void* dy, const void* wp_dgrad, void* dx, int N, int Hc, int Wc, int Ci, int CoP, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(dy && wp_dgrad && dx, "conv2d_dgrad_up: null pointer");
  PCRL_REQUIRE(pcrl_conv2d_dgrad_up_ok(N, Hc, Wc, Ci, CoP, dtype), "conv2d_dgrad_up: not available for this geometry (pcrl_conv2d_dgrad_up_ok)");
  return pcrl_conv2d_narrow_launch(dy, wp_dgrad, nullptr, dx, nullptr, N, 2 * Hc, 2 * Wc, CoP, Ci, 3, 0, 0, 1, as_stream(stream));
}

// dx[N][Hi][Wi][Ci] from dy[N][Ho][Wo][CoP]; (Ho, Wo) are the forward output dims of the (Hi, Wi) input.
extern "C" int pcrl_conv2d_dgrad(const void* dy, const void* wp_dgrad, void* dx, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int CoP, int KH,
                                 int KW, int stride, int pad, int dtype, pcrl_stream_t stream) {
  if (g_conv2d_impl == 0 && dy && wp_dgrad && dx && pcrl_conv2d_dgrad_kind(N, Hi, Wi, Ci, Ho, Wo, CoP, KH, KW, stride, pad, dtype) == 3)
    return pcrl_brick16_conv2d_launch(dy, wp_dgrad, nullptr, dx, nullptr, N, Hi, Wi, CoP, Ci, as_stream(stream));
  if (auto_impl() && KH == 3 && KW == 3 && stride == 1 && pad == 1 && dy && wp_dgrad && dx && Hi == Ho && Wi == Wo &&
      pcrl_brick_conv2d_eligible(N, Hi, Wi, CoP, Ci, dtype) && pcrl_conv2d_dgrad_kind(N, Hi, Wi, Ci, Ho, Wo, CoP, KH, KW, stride, pad, dtype) == 1)
    return pcrl_brick_conv2d_launch(dy, wp_dgrad, nullptr, dx, nullptr, N, Hi, Wi, CoP, Ci, 0, as_stream(stream));
  if (auto_impl() && KH == KW && (KH == 1 || KH == 3) && stride == 1 && pad == (KH - 1) / 2 && dy && wp_dgrad && dx && Hi == Ho && Wi == Wo &&
      pcrl_conv2d_narrow_eligible(N, Hi, Wi, CoP, Ci, KH, dtype))
    return pcrl_conv2d_narrow_launch(dy, wp_dgrad, nullptr, dx, nullptr, N, Hi, Wi, CoP, Ci, KH, 0, 0, 0, as_stream(stream));
  return conv2d_common("conv2d_dgrad", C2_DGRAD, dy, wp_dgrad, nullptr, dx, nullptr, N, Ho, Wo, CoP, Hi, Wi, Ci, KH, KW, stride, pad, 0, 0, dtype,
                       as_stream(stream));
}

// ---- 3x3 / stride 1 / pad 1 data gradient + first pass of the BatchNorm backward of the layer below (conv_brick16_bnr.hip) ----
int pcrl_brick16_dgrad2d_bnred_launch(const void* dy, const void* wp, void* dx, const void* bn_y, const float* scale, const float* shift, const float* mean,
                                      const float* rstd, float* partial, int N, int H, int W, int Ci, int Co, hipStream_t stream);
extern "C" int64_t pcrl_conv2d_dgrad_bnred_rows(int N, int H, int W, int Ci, int CoP, int act, int dtype) {
  static const bool off = [] { const char* e = getenv("PCRL_DGRAD_BNRED"); return e && e[0] == '0'; }();   // A/B switch (shared with the 3D path)
  if (off || act != PCRL_ACT_RELU || N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || CoP <= 0) return 0;
  return pcrl_conv2d_dgrad_kind(N, H, W, Ci, H, W, CoP, 3, 3, 1, 1, dtype) == 3 ? pcrl_brick16_conv2d_rows(N, H, W) : 0;
}
extern "C" int pcrl_conv2d_dgrad_bnred(const void* dy, const void* wp_dgrad, void* dx, const void* bn_y, const float* scale, const float* shift,
                                       const float* mean, const float* rstd, float* partial, int N, int H, int W, int Ci, int CoP, int act, int dtype,
                                       pcrl_stream_t stream) {
  PCRL_REQUIRE(dy && wp_dgrad && dx && bn_y && scale && shift && mean && rstd && partial, "conv2d_dgrad_bnred: null pointer");
  PCRL_REQUIRE(pcrl_conv2d_dgrad_bnred_rows(N, H, W, Ci, CoP, act, dtype) > 0,
               "conv2d_dgrad_bnred: no fused kernel for this shape / activation / dtype (pcrl_conv2d_dgrad_bnred_rows == 0)");
  return pcrl_brick16_dgrad2d_bnred_launch(dy, wp_dgrad, dx, bn_y, scale, shift, mean, rstd, partial, N, H, W, CoP, Ci, as_stream(stream));
}

// ---- stride-2 data gradient by parity classes ------------------------------------------------------------------------------------
// dx[ih][iw] of a stride-2 convolution only receives taps whose parity matches (ih + pad - kh even): run as ONE gather over all
// taps, 3 of 4 taps are idle (conv2d_dgrad above).  Here the four parity classes (a, b) = (ih & 1, iw & 1) are four small stride-1
// convolutions over dy with 1, 2, 2 and 4 taps whose outputs are stored at pixels (2q + a, 2r + b): no idle taps.
//   3x3 / pad 1: class 0 = tap kh 1 (source row q); class 1 = taps kh 2 (source q) and kh 0 (source q + 1)
//   1x1 / pad 0: only class (0, 0) is non-zero (the caller zero-fills dx)
// Hi, Wi even.  pack: rows round32(Ci), K = KHc * KWc * CoP (pcrl_conv2d_packed_elems(Ci, KHc * KWc, CoP)).
static void s2_class(int KH, int a, int& KHc, int& k0, int& k1) {
  if (KH == 3) {
    KHc = a ? 2 : 1;
    k0 = a ? 2 : 1;
    k1 = 0;
  } else {
    KHc = 1;
    k0 = k1 = 0;
  }
}

extern "C" int pcrl_conv2d_pack_s2(const float* w_ref, void* out, int Co, int Ci, int KH, int KW, int CoP, int a, int b, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(w_ref && out && Co > 0 && Ci > 0 && CoP >= Co, "conv2d_pack_s2: bad arguments");
  PCRL_REQUIRE((KH == 3 && KW == 3) || (KH == 1 && KW == 1 && a == 0 && b == 0), "conv2d_pack_s2: 3x3 (pad 1) or 1x1 (class 0,0) only");
  int KHc, KWc, kh0, kh1, kw0, kw1;
  s2_class(KH, a, KHc, kh0, kh1);
  s2_class(KW, b, KWc, kw0, kw1);
  const int rowsP = (Ci + 31) / 32 * 32, Kpad = (KHc * KWc * CoP + 31) / 32 * 32;
  const int64_t total = (int64_t)rowsP * Kpad;
  const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == PCRL_BF16)
    hipLaunchKernelGGL(pack_conv2d_s2_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), w_ref, (bf16*)out, Co, Ci, KH, KW, CoP, rowsP, Kpad, KHc, KWc, kh0, kh1, kw0, kw1);
  else if (dtype == PCRL_F32)
    hipLaunchKernelGGL(pack_conv2d_s2_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), w_ref, (float*)out, Co, Ci, KH, KW, CoP, rowsP, Kpad, KHc, KWc, kh0, kh1, kw0, kw1);
  else
    return pcrl_fail(PCRL_EINVAL, "conv2d_pack_s2: bad dtype %d", dtype);
  return pcrl_check_launch("conv2d_pack_s2");
}

extern "C" int pcrl_conv2d_dgrad_s2(const void* dy, const void* wp_class, void* dx, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int CoP, int KH, int KW,
                                    int a, int b, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE((KH == 3 && KW == 3) || (KH == 1 && KW == 1 && a == 0 && b == 0), "conv2d_dgrad_s2: 3x3 (pad 1) or 1x1 (class 0,0) only");
  PCRL_REQUIRE(Hi % 2 == 0 && Wi % 2 == 0 && Ho == Hi / 2 && Wo == Wi / 2, "conv2d_dgrad_s2: even input extents and Ho = Hi / 2 expected");
  int KHc, KWc, k0, k1;
  s2_class(KH, a, KHc, k0, k1);
  s2_class(KW, b, KWc, k0, k1);
  // a stride-1, pad-0 forward gather over dy with KHc x KWc taps at offsets {0, +1}; rows = the class's pixels (q, r)
  return conv2d_common("conv2d_dgrad_s2", C2_FWD, dy, wp_class, nullptr, dx, nullptr, N, Ho, Wo, CoP, Hi / 2, Wi / 2, Ci, KHc, KWc, 1, 0, 0, 0, dtype,
                       as_stream(stream), 1, a, b, Hi, Wi);
}
