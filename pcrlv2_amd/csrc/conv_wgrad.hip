// Weight-gradient GEMMs on MFMA for gfx950 (split-K over voxels, deterministic second pass).
//
// Replaces the weight half of aten::convolution_backward for models/pcrlv2_model_3d.py:9,33
// (3x3x3 conv) and :52,64 (ConvTranspose3d k2 s2).  Per tap t:
//     P_t[i][j] = sum_m U[m][i] * V[g(m,t)][j]
//   conv3 : U = dy (i = co), V = x  (j = ci), g(m,t) = m + delta_t (zero outside the volume)
//   convT : U = x  (i = ci), V = dy (j = co), g(m,t) = row of tap t in the 2x-upsampled volume
// and the result is written in the reference layout [i][j][t].
//
// The reduction index m is the ROW index of both NDHWC operands, while an MFMA operand wants 8
// consecutive k per lane.  bf16: tiles are staged in their natural [voxel][channel] layout and the
// fragments are fetched with ds_read_b64_tr_b16 (gfx950 transpose read), two per fragment;
// a scalar-read fallback (`tr = 0`) is kept for validation.  fp32: v_mfma_f32_16x16x4_f32 takes one
// float per lane, so the natural layout is read directly.
//
// Block = 64(i) x 64(j) result tile of ONE tap and ONE voxel split; 4 waves as 2x2, each 32x32
// (2x2 fragments); K-step = 32 voxels, double-buffered register-staged LDS tiles.
#include "common.h"
#include <atomic>
#include <mutex>

namespace {

enum { WG_CONV3 = 0, WG_UP2 = 1, WG_PLAIN = 2, WG_CONV2D = 3, WG_UPC = 4 };  // PLAIN: V is indexed by m directly (taps == 1)
// WG_UPC: the fused ConvTranspose3d(k2,s2) -> Conv3d(3x3x3) operator (conv_igemm.hip, GEOM_UPC_*): m = coarse voxel, 64 taps
// t = phase p (3 bits) x coarse tap q (3 bits); U = dy0 at the phase's fine voxel 2m + p, V = x at the coarse voxel m + p - 1 + q.

// WG_CONV2D (2D path, conv2d.hip): m = output pixel (n, oh, ow) in g = {N, 1, Ho, Wo}; V row = source pixel
// (oh*stride - pad + kh, ow*stride - pad + kw) of [N][Hs][Ws], read through a nearest x2 upsample when `up`.
// The taps are FLATTENED into the column index: j = tap * CiP + ci over Cv = ntaps * CiP columns (blockIdx.y unused, taps = 1), so
// a 64-column tile covers 64/CiP taps when the layer is narrow -- the U operand (dy) is re-read once per 64 columns, not once per
// tap (7x7 stem on 8 padded channels: 7 passes over dy instead of 49).  Every 16-byte chunk of a tile row lies inside one tap, so a
// thread decodes the tap of its chunk once.
struct Wg2d {
  int Hs, Ws, KW, stride, pad, up, CiP, ntaps;
};

struct WgradParams {
  const void* u;  // [M][Cu]
  const void* v;  // [rows][Cv]
  float* ws;      // [splits][taps][Cu][Cv]
  Dims g;         // index space of m (output voxels for conv3, input voxels for convT)
  int64_t M;
  int Cu, Cv, taps;
  int64_t chunk;  // voxels per split, multiple of 32
  Wg2d q;         // WG_CONV2D only
};

// 32-byte quad swizzle of a 64-channel bf16 row (4 quads of 16 channels): spreads the 8 rows a
// 32-lane group of a transpose-read touches over all 64 banks.
__device__ __forceinline__ int quad_sw(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

template <typename T> struct WTile;
template <> struct WTile<bf16> {
  static constexpr int ROWB = 128;  // 64 ch * 2 B
  static __device__ __forceinline__ int off(int row, int col) {  // byte offset of channel `col`
    return row * ROWB + ((((col >> 4) ^ quad_sw(row)) & 3) << 5) + ((col & 15) << 1);
  }
};
template <> struct WTile<float> {
  static constexpr int ROWB = 256;
  static __device__ __forceinline__ int off(int row, int col) { return row * ROWB + (col << 2); }
};

template <typename T, bool TR> struct WFrag;
template <> struct WFrag<float, false> {
  struct Frag { float v[8]; };
  // lane (c = lane&15, g = lane>>4) takes voxels k = 8g..8g+7 of channel cbase+c
  static __device__ __forceinline__ Frag read(const char* tile, int cbase, int lane) {
    Frag f;
    const int c = cbase + (lane & 15), g = lane >> 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = *reinterpret_cast<const float*>(tile + WTile<float>::off(8 * g + e, c));
    return f;
  }
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
#pragma unroll
    for (int e = 0; e < 8; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[e], b.v[e], c, 0, 0, 0);
  }
};
template <> struct WFrag<bf16, false> {
  using Frag = bf16x8;
  static __device__ __forceinline__ Frag read(const char* tile, int cbase, int lane) {
    Frag f;
    const int c = cbase + (lane & 15), g = lane >> 4;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = *reinterpret_cast<const bf16*>(tile + WTile<bf16>::off(8 * g + e, c));
    return f;
  }
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct WFrag<bf16, true> {
  using Frag = bf16x8;
  // ds_read_b64_tr_b16: within each 16-lane group, result lane l element j comes from source lane
  // 4j + (l&15)/4, element l&3.  Source lane s therefore points at voxel row (s&15)>>2 of the 4-row group,
  // channels 4*(s&3)..+3; lane l receives channel (l&15) of 4 consecutive voxels.  Two reads give the lane
  // voxels 8g..8g+7 of its channel -- the canonical A/B fragment of v_mfma_f32_16x16x32_bf16.
  static __device__ __forceinline__ Frag read(const char* tile, int cbase, int lane) {
    const int g = lane >> 4, jr = (lane & 15) >> 2, cq = lane & 3;
    const int row0 = 8 * g + jr;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const char* p0 = tile + WTile<bf16>::off(row0, cbase + 4 * cq);
    const char* p1 = tile + WTile<bf16>::off(row0 + 4, cbase + 4 * cq);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
    union { struct { s16x4 a, b; } s; bf16x8 f; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.f;
  }
  static __device__ __forceinline__ void mma(const Frag& a, const Frag& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

// KS = voxels per K-step (32, or 128 for the narrow full-resolution layers of the 2D path where the two barriers per step, not
// the MFMAs, set the pace).
template <typename T, int GEOM, bool TR, int KS = 32>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradParams p) {
  using WT = WTile<T>;
  using WF = WFrag<T, TR>;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int CPR = WT::ROWB / 16;  // 16-byte chunks per tile row: 8 / 16
  constexpr int RPP = 256 / CPR;      // rows per pass: 32 / 16
  constexpr int NP = KS / RPP;        // passes: 1 / 2 at KS = 32
  constexpr int TILE_BYTES = KS * WT::ROWB;

  extern __shared__ __attribute__((aligned(16))) char smem[];  // U[2], V[2]: 4 * TILE_BYTES
  char* Us = smem;
  char* Vs = smem + 2 * TILE_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wi = wid >> 1, wj = wid & 1;
  const int ntj = (p.Cv + 63) / 64;
  const int i0 = (blockIdx.x / ntj) * 64, j0 = (blockIdx.x % ntj) * 64;
  const int t = blockIdx.y;
  const int64_t mbeg = (int64_t)blockIdx.z * p.chunk;
  const int64_t mend = (mbeg + p.chunk < p.M) ? (mbeg + p.chunk) : p.M;
  const T* __restrict__ U = reinterpret_cast<const T*>(p.u);
  const T* __restrict__ V = reinterpret_cast<const T*>(p.v);
  const Dims g = p.g;

  const int chunk16 = tid % CPR, rowp = tid / CPR;
  const int ucol = chunk16 * VEC;                 // channel offset of this thread's 16 bytes in the tile
  const bool u_ok = (i0 + ucol) < p.Cu;           // Cu, Cv are multiples of 32; a 64-wide tile may hang over
  const bool v_ok = (j0 + ucol) < p.Cv;
  const int kd = t / 9, kh = (t / 3) % 3, kw = t % 3;
  const int64_t delta = (GEOM == WG_CONV3) ? tap_delta27(t, g) : 0;
  const int tap2 = (GEOM == WG_CONV2D) ? (j0 + ucol) / p.q.CiP : 0, c2 = (GEOM == WG_CONV2D) ? (j0 + ucol) % p.q.CiP : 0;
  const int kh2 = tap2 / p.q.KW, kw2 = tap2 % p.q.KW;
  const int Hl2 = p.q.up ? 2 * p.q.Hs : p.q.Hs, Wl2 = p.q.up ? 2 * p.q.Ws : p.q.Ws;
  const int vpitch = (GEOM == WG_CONV2D) ? p.q.CiP : p.Cv;
  const int vcol = (GEOM == WG_CONV2D) ? (v_ok ? c2 : 0) : j0 + (v_ok ? ucol : 0);

  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ru[NP], rv[NP];

  // Unconditional loads from clamped (always valid) addresses; validity is applied at the LDS store (a load under a
  // per-lane condition is wrapped by hipcc in a branch + `s_waitcnt vmcnt(0)`, serialising the K-step).
  const int ucol_u = u_ok ? ucol : 0;
  uint32_t uokb = 0, vokb = 0;
  // WG_CONV2D: the pixel coordinates of this thread's rows are carried from K-step to K-step (a step advances every row by KS
  // pixels = (sn2, sh2, sw2) in (image, row, column) units) instead of being decoded with three 64-bit divisions per row and step
  // -- on the narrow full-resolution layers that index arithmetic, not the MFMAs, was the kernel's instruction stream.
  int cn2[NP], ch2[NP], cw2[NP];
  int64_t cm2[NP];
  int sn2 = 0, sh2 = 0, sw2 = 0;
  if (GEOM == WG_CONV2D) {
    sw2 = KS % g.W;
    sh2 = (KS / g.W) % g.H;
    sn2 = KS / (g.W * g.H);
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      cm2[ps] = mbeg + ps * RPP + rowp;
      const int64_t mc = cm2[ps] < p.M ? cm2[ps] : 0;
      int dd;
      decode_voxel(mc, g, cn2[ps], dd, ch2[ps], cw2[ps]);
    }
  }
#define WG_LOAD2D()                                                                             \
  do {                                                                                          \
    uokb = 0;                                                                                   \
    vokb = 0;                                                                                   \
    _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                         \
      const bool live = cm2[ps] < mend;                                                         \
      const int64_t m = live ? cm2[ps] : mbeg;                                                  \
      ru[ps] = *reinterpret_cast<const u32x4*>(U + m * p.Cu + i0 + ucol_u);                     \
      int ih = ch2[ps] * p.q.stride - p.q.pad + kh2, iw = cw2[ps] * p.q.stride - p.q.pad + kw2; \
      const bool in = live && (unsigned)ih < (unsigned)Hl2 && (unsigned)iw < (unsigned)Wl2;     \
      if (p.q.up) {                                                                             \
        ih >>= 1;                                                                               \
        iw >>= 1;                                                                               \
      }                                                                                         \
      const int64_t vrow = in ? ((int64_t)cn2[ps] * p.q.Hs + ih) * p.q.Ws + iw : (int64_t)0;    \
      rv[ps] = *reinterpret_cast<const u32x4*>(V + vrow * vpitch + vcol);                       \
      uokb |= (uint32_t)(live && u_ok) << ps;                                                   \
      vokb |= (uint32_t)(in && v_ok) << ps;                                                     \
      cm2[ps] += KS;                                                                            \
      cw2[ps] += sw2;                                                                           \
      if (cw2[ps] >= g.W) {                                                                     \
        cw2[ps] -= g.W;                                                                         \
        ch2[ps] += 1;                                                                           \
      }                                                                                         \
      ch2[ps] += sh2;                                                                           \
      if (ch2[ps] >= g.H) {                                                                     \
        ch2[ps] -= g.H;                                                                         \
        cn2[ps] += 1;                                                                           \
      }                                                                                         \
      cn2[ps] += sn2;                                                                           \
    }                                                                                           \
  } while (0)
  // WG_CONV3: the same for the 3x3x3 geometry (the gather kernel serves the small volumes -- local views at 4^3 and 2^3 -- where a block
  // has few K-steps of 4 MFMAs each and the three 64-bit divisions per row and step were most of its instructions).
  int c3n[NP], c3d[NP], c3h[NP], c3w[NP];
  int64_t c3m[NP];
  int s3n = 0, s3d = 0, s3h = 0, s3w = 0;
  if (GEOM == WG_CONV3) {
    s3w = KS % g.W;
    s3h = (KS / g.W) % g.H;
    s3d = (KS / (g.W * g.H)) % g.D;
    s3n = KS / (g.W * g.H * g.D);
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      c3m[ps] = mbeg + ps * RPP + rowp;
      const int64_t mc = c3m[ps] < p.M ? c3m[ps] : 0;
      decode_voxel(mc, g, c3n[ps], c3d[ps], c3h[ps], c3w[ps]);
    }
  }
#define WG_LOAD3D()                                                                             \
  do {                                                                                          \
    uokb = 0;                                                                                   \
    vokb = 0;                                                                                   \
    _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                         \
      const bool live = c3m[ps] < mend;                                                         \
      const int64_t m = live ? c3m[ps] : mbeg;                                                  \
      ru[ps] = *reinterpret_cast<const u32x4*>(U + m * p.Cu + i0 + ucol_u);                     \
      const bool in = live && (unsigned)(c3d[ps] + kd - 1) < (unsigned)g.D && (unsigned)(c3h[ps] + kh - 1) < (unsigned)g.H && \
                      (unsigned)(c3w[ps] + kw - 1) < (unsigned)g.W;                             \
      const int64_t vrow = in ? m + delta : m;                                                  \
      rv[ps] = *reinterpret_cast<const u32x4*>(V + vrow * vpitch + vcol);                       \
      uokb |= (uint32_t)(live && u_ok) << ps;                                                   \
      vokb |= (uint32_t)(in && v_ok) << ps;                                                     \
      c3m[ps] += KS;                                                                            \
      c3w[ps] += s3w;                                                                           \
      if (c3w[ps] >= g.W) {                                                                     \
        c3w[ps] -= g.W;                                                                         \
        c3h[ps] += 1;                                                                           \
      }                                                                                         \
      c3h[ps] += s3h;                                                                           \
      if (c3h[ps] >= g.H) {                                                                     \
        c3h[ps] -= g.H;                                                                         \
        c3d[ps] += 1;                                                                           \
      }                                                                                         \
      c3d[ps] += s3d;                                                                           \
      if (c3d[ps] >= g.D) {                                                                     \
        c3d[ps] -= g.D;                                                                         \
        c3n[ps] += 1;                                                                           \
      }                                                                                         \
      c3n[ps] += s3n;                                                                           \
    }                                                                                           \
  } while (0)
  // WG_UPC: coarse coordinates carried like WG_CONV3's
  const int upd = (t >> 5) & 1, uph = (t >> 4) & 1, upw = (t >> 3) & 1;            // phase
  const int uod = upd - 1 + ((t >> 2) & 1), uoh = uph - 1 + ((t >> 1) & 1), uow = upw - 1 + (t & 1);   // coarse offset of the tap
  if (GEOM == WG_UPC) {
    s3w = KS % g.W;
    s3h = (KS / g.W) % g.H;
    s3d = (KS / (g.W * g.H)) % g.D;
    s3n = KS / (g.W * g.H * g.D);
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      c3m[ps] = mbeg + ps * RPP + rowp;
      const int64_t mc = c3m[ps] < p.M ? c3m[ps] : 0;
      decode_voxel(mc, g, c3n[ps], c3d[ps], c3h[ps], c3w[ps]);
    }
  }
#define WG_LOADUPC()                                                                            \
  do {                                                                                          \
    uokb = 0;                                                                                   \
    vokb = 0;                                                                                   \
    _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                         \
      const bool live = c3m[ps] < mend;                                                         \
      const int cn_ = live ? c3n[ps] : 0, cd_ = live ? c3d[ps] : 0, ch_ = live ? c3h[ps] : 0, cw_ = live ? c3w[ps] : 0; \
      const int64_t urow = (((int64_t)cn_ * (2 * g.D) + 2 * cd_ + upd) * (2 * g.H) + 2 * ch_ + uph) * (2 * g.W) + 2 * cw_ + upw; \
      ru[ps] = *reinterpret_cast<const u32x4*>(U + urow * p.Cu + i0 + ucol_u);                  \
      const bool in = live && (unsigned)(cd_ + uod) < (unsigned)g.D && (unsigned)(ch_ + uoh) < (unsigned)g.H && \
                      (unsigned)(cw_ + uow) < (unsigned)g.W;                                    \
      const int64_t vrow = in ? (((int64_t)cn_ * g.D + cd_ + uod) * g.H + ch_ + uoh) * g.W + cw_ + uow : (int64_t)0; \
      rv[ps] = *reinterpret_cast<const u32x4*>(V + vrow * vpitch + vcol);                       \
      uokb |= (uint32_t)(live && u_ok) << ps;                                                   \
      vokb |= (uint32_t)(in && v_ok) << ps;                                                     \
      c3m[ps] += KS;                                                                            \
      c3w[ps] += s3w;                                                                           \
      if (c3w[ps] >= g.W) {                                                                     \
        c3w[ps] -= g.W;                                                                         \
        c3h[ps] += 1;                                                                           \
      }                                                                                         \
      c3h[ps] += s3h;                                                                           \
      if (c3h[ps] >= g.H) {                                                                     \
        c3h[ps] -= g.H;                                                                         \
        c3d[ps] += 1;                                                                           \
      }                                                                                         \
      c3d[ps] += s3d;                                                                           \
      if (c3d[ps] >= g.D) {                                                                     \
        c3d[ps] -= g.D;                                                                         \
        c3n[ps] += 1;                                                                           \
      }                                                                                         \
      c3n[ps] += s3n;                                                                           \
    }                                                                                           \
  } while (0)
#define WG_LOAD(ms_)                                                                            \
  do {                                                                                          \
    uokb = 0;                                                                                   \
    vokb = 0;                                                                                   \
    _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                         \
      const int64_t m_ = (ms_) + ps * RPP + rowp;                                               \
      const bool live = m_ < mend;                                                              \
      const int64_t m = live ? m_ : mbeg;                                                       \
      ru[ps] = *reinterpret_cast<const u32x4*>(U + m * p.Cu + i0 + ucol_u);                     \
      int n, d, h, w;                                                                           \
      decode_voxel(m, g, n, d, h, w);                                                           \
      bool ok = live;                                                                           \
      int64_t vrow;                                                                             \
      if (GEOM == WG_CONV3) {                                                                   \
        const bool in = (unsigned)(d + kd - 1) < (unsigned)g.D && (unsigned)(h + kh - 1) < (unsigned)g.H && \
                        (unsigned)(w + kw - 1) < (unsigned)g.W;                                 \
        ok = ok && in;                                                                          \
        vrow = in ? m + delta : m;                                                              \
      } else if (GEOM == WG_UP2) {                                                              \
        vrow = up2_row(n, d, h, w, t, g);                                                       \
      } else if (GEOM == WG_CONV2D) {                                                           \
        int ih = h * p.q.stride - p.q.pad + kh2, iw = w * p.q.stride - p.q.pad + kw2;           \
        const bool in = (unsigned)ih < (unsigned)Hl2 && (unsigned)iw < (unsigned)Wl2;           \
        if (p.q.up) {                                                                           \
          ih >>= 1;                                                                             \
          iw >>= 1;                                                                             \
        }                                                                                       \
        ok = ok && in;                                                                          \
        vrow = in ? ((int64_t)n * p.q.Hs + ih) * p.q.Ws + iw : (int64_t)0;                      \
      } else {                                                                                  \
        vrow = m;                                                                               \
      }                                                                                         \
      rv[ps] = *reinterpret_cast<const u32x4*>(V + vrow * vpitch + vcol);                       \
      uokb |= (uint32_t)(live && u_ok) << ps;                                                   \
      vokb |= (uint32_t)(ok && v_ok) << ps;                                                     \
    }                                                                                           \
  } while (0)

#define WG_STORE(buf_)                                                                          \
  do {                                                                                          \
    _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                         \
      const int row = ps * RPP + rowp;                                                          \
      *reinterpret_cast<u32x4*>(Us + (buf_)*TILE_BYTES + WT::off(row, ucol)) = keep_if((uokb >> ps) & 1u, ru[ps]);  \
      *reinterpret_cast<u32x4*>(Vs + (buf_)*TILE_BYTES + WT::off(row, ucol)) = keep_if((vokb >> ps) & 1u, rv[ps]);  \
    }                                                                                           \
  } while (0)

  const int64_t nsteps = (mend > mbeg) ? (mend - mbeg + KS - 1) / KS : 0;
  if (nsteps > 0) {
    if (GEOM == WG_CONV2D) WG_LOAD2D();
    else if (GEOM == WG_CONV3) WG_LOAD3D();
    else if (GEOM == WG_UPC) WG_LOADUPC();
    else WG_LOAD(mbeg);
    WG_STORE(0);
  }
  __syncthreads();
  // straight-line body: the last iteration stages a duplicate of its own step into the idle buffer (see conv_igemm.hip)
  for (int64_t s = 0; s < nsteps; ++s) {
    const int cur = (int)(s & 1);
    const int64_t sn = (s + 1 < nsteps) ? s + 1 : s;
    if (GEOM == WG_CONV2D) WG_LOAD2D();      // the last iteration stages rows past `mend`: dead, zeroed at the LDS store
    else if (GEOM == WG_CONV3) WG_LOAD3D();
    else if (GEOM == WG_UPC) WG_LOADUPC();
    else WG_LOAD(mbeg + sn * KS);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < KS / 32; ++kk) {
      const char* ut = Us + cur * TILE_BYTES + kk * 32 * WT::ROWB;
      const char* vt = Vs + cur * TILE_BYTES + kk * 32 * WT::ROWB;
      typename WF::Frag fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = WF::read(ut, wi * 32 + a * 16, lane);
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = WF::read(vt, wj * 32 + b * 16, lane);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) WF::mma(fa[a], fb[b], acc[a][b]);
    }
    __builtin_amdgcn_sched_barrier(0);
    WG_STORE(cur ^ 1);
    __syncthreads();
  }
#undef WG_LOAD
#undef WG_LOAD2D
#undef WG_LOADUPC
#undef WG_LOAD3D
#undef WG_STORE

  float* __restrict__ out = p.ws + ((int64_t)blockIdx.z * p.taps + t) * (int64_t)p.Cu * p.Cv;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wi * 32 + a * 16 + (lane >> 4) * 4 + r;
        const int j = j0 + wj * 32 + b * 16 + (lane & 15);
        if (i < p.Cu && j < p.Cv) out[(int64_t)i * p.Cv + j] = acc[a][b][r];
      }
}

// Second pass: out_ref[i][j][t] = sum_z ws[z][t][i][j] for j < Cv_out (padding columns dropped), fixed order over z.
// Block = 16 consecutive (i,j) x all taps: a thread sums a few (t, ij) pairs over z (64-byte runs per tap), the [t][ij] -> [ij][t]
// transposition goes through LDS, the stores are one contiguous run.  When a block has <= 128 (t, ij) pairs (1 or 8 taps) the
// slabs are spread over 256 / pairs thread groups whose partial sums are combined in group order (the 512-slab partials of the
// 1-channel layer: 44 -> 6 us).  taps <= 64.
// [measured, r02d: a float4 / z-group form with 4-16 (i,j) per block was SLOWER (1.06 -> 1.85 ms per step): with 27 taps a wave then
//  touches 64 different 128-byte lines for 1 KB of data; the pass is bound by line requests, not by load latency.]
constexpr int RED_IJ = 16;
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                           int splits, int taps, int Cu, int Cv, int Cv_out, bool accumulate = false) {
  __shared__ float tile[RED_IJ * 64];
  __shared__ double part[256];
  const int64_t per = (int64_t)Cu * Cv;
  const int64_t n_out = (int64_t)Cu * Cv_out;
  const int64_t ij0 = (int64_t)blockIdx.x * RED_IJ;
  const int npairs = taps * RED_IJ;
  const int zg = npairs <= 128 ? 256 / npairs : 1;
  const int span = zg > 1 ? npairs * zg : npairs;
  for (int q0 = threadIdx.x; q0 < span; q0 += 256) {
    const int g = q0 / npairs, q = q0 - g * npairs;
    const int t = q / RED_IJ, l = q % RED_IJ;
    const int64_t ij = ij0 + l;
    double s = 0.0;
    if (ij < n_out) {
      const int i = (int)(ij / Cv_out), j = (int)(ij % Cv_out);
      const float* src = ws + (int64_t)t * per + (int64_t)i * Cv + j;
      const int64_t zs = (int64_t)taps * per;
      // four independent partial sums (a fixed tree, so still deterministic) keep 4+ loads in flight per thread
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int z = g;
      for (; z + 3 * zg < splits; z += 4 * zg) {
        a0 += (double)src[(int64_t)(z + 0 * zg) * zs];
        a1 += (double)src[(int64_t)(z + 1 * zg) * zs];
        a2 += (double)src[(int64_t)(z + 2 * zg) * zs];
        a3 += (double)src[(int64_t)(z + 3 * zg) * zs];
      }
      for (; z < splits; z += zg) a0 += (double)src[(int64_t)z * zs];
      s = (a0 + a1) + (a2 + a3);
    }
    if (zg > 1) part[q0] = s;
    else tile[l * taps + t] = (float)s;
  }
  if (zg > 1) {
    __syncthreads();
    if ((int)threadIdx.x < npairs) {
      double s = part[threadIdx.x];
      for (int k = 1; k < zg; ++k) s += part[k * npairs + threadIdx.x];
      tile[(threadIdx.x % RED_IJ) * taps + threadIdx.x / RED_IJ] = (float)s;
    }
  }
  __syncthreads();
  const int64_t base = ij0 * taps, lim = n_out * taps;
  for (int k = threadIdx.x; k < npairs; k += 256)
    if (base + k < lim) out[base + k] = accumulate ? out[base + k] + tile[k] : tile[k];
}

// Timing ablation (tools/double_ablation.py): the non-accumulating second passes are launched this many times (idempotent: same slabs, same output) --
// what a set of launches costs INSIDE the three-stream step is the step-time difference between 2 and 1.  Default 1.
static std::atomic<int> g_reduce_repeat{1};
extern "C" void pcrl_debug_set_reduce_repeat(int n) { g_reduce_repeat = n < 1 ? 1 : n; }

// every weight-gradient path ends here (`accumulate`: out += the sum, for gradients gathered over several passes)
int launch_wgrad_reduce(const float* ws, float* out, int splits, int taps, int Cu, int Cv, int Cv_out, hipStream_t stream, bool accumulate = false) {
  if (taps < 1 || taps > 64) return pcrl_fail(PCRL_EINVAL, "wgrad_reduce: %d taps", taps);
  const int blocks = (int)(((int64_t)Cu * Cv_out + RED_IJ - 1) / RED_IJ);
  for (int rep = accumulate ? 1 : (int)g_reduce_repeat; rep > 0; --rep)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, ws, out, splits, taps, Cu, Cv, Cv_out, accumulate);
  return pcrl_check_launch("wgrad_reduce");
}

// Second pass of the 2D weight gradient: out_ref[co][ci][t] = sum_z ws[z][co][t * CiP + ci], ci < Ci_out (padding channels dropped).
// Block = (256 / ZG) outputs x ZG split groups: thread (o, g) sums splits g, g+ZG, ... in double, the ZG partial sums are combined in
// fixed order through LDS.  ZG = 4 for the wide layers (few splits, many outputs), 16 for the narrow full-resolution layers
// (thousands of slabs, a few thousand outputs: one thread per output crawled).
template <int ZG>
__global__ void __launch_bounds__(256) wgrad2d_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int splits, int taps, int Cu,
                                                             int CiP, int Ci_out) {
  constexpr int NO = 256 / ZG;
  __shared__ double part[ZG][NO];
  const int64_t per = (int64_t)Cu * taps * CiP;
  const int64_t total = (int64_t)Cu * taps * Ci_out;
  const int o = threadIdx.x % NO, zg = threadIdx.x / NO;
  const int64_t idx = (int64_t)blockIdx.x * NO + o;
  double a = 0.0;
  int64_t dst = 0;
  if (idx < total) {
    const int ci = (int)(idx % Ci_out);
    const int64_t r = idx / Ci_out;
    const int t = (int)(r % taps), co = (int)(r / taps);
    const float* src = ws + ((int64_t)co * taps + t) * CiP + ci;
    dst = ((int64_t)co * Ci_out + ci) * taps + t;
    // four independent partial sums per thread (a fixed tree: deterministic): the walk over the slabs is a chain of dependent round trips otherwise
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int z = zg;
    for (; z + 3 * ZG < splits; z += 4 * ZG) {
      const float v0 = src[(int64_t)z * per], v1 = src[(int64_t)(z + ZG) * per], v2 = src[(int64_t)(z + 2 * ZG) * per], v3 = src[(int64_t)(z + 3 * ZG) * per];
      a += (double)v0; a1 += (double)v1; a2 += (double)v2; a3 += (double)v3;
    }
    for (; z < splits; z += ZG) a += (double)src[(int64_t)z * per];
    a = (a + a1) + (a2 + a3);
  }
  part[zg][o] = a;
  __syncthreads();
  if (zg == 0 && idx < total) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < ZG; ++q) s += part[q][o];
    out[dst] = (float)s;
  }
}
static int launch_wgrad2d_reduce(const float* ws, float* out, int splits, int taps, int Cu, int CiP, int Ci_out, hipStream_t st) {
  const int64_t total = (int64_t)Cu * taps * Ci_out;
  if (splits > 64) hipLaunchKernelGGL(wgrad2d_reduce_kernel<16>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, ws, out, splits, taps, Cu, CiP, Ci_out);
  else hipLaunchKernelGGL(wgrad2d_reduce_kernel<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, ws, out, splits, taps, Cu, CiP, Ci_out);
  return pcrl_check_launch("conv2d_wgrad_reduce");
}

// im2col of a float32 scalar field for the 1-channel convolutions: out[m][t] = s[m + delta_t] (0 outside the volume),
// t < 27 (flip: delta of tap 26-t), columns 27..31 zero.  taps == 1: out[m][0] = s[m].
template <typename T>
__global__ void __launch_bounds__(256) im2col27_kernel(const float* __restrict__ s, T* __restrict__ out, Dims g, int64_t M, int taps,
                                                       int flip) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int NV = 32 / VEC;
  const int64_t total = M * NV;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t m = idx / NV;
    const int v0 = (int)(idx % NV) * VEC;
    int n, d, h, w;
    decode_voxel(m, g, n, d, h, w);
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int col = v0 + j;
      float val = 0.f;
      if (taps == 1) {
        if (col == 0) val = s[m];
      } else if (col < 27) {
        const int t = flip ? 26 - col : col;
        const int kd = t / 9 - 1, kh = (t / 3) % 3 - 1, kw = t % 3 - 1;
        if ((unsigned)(d + kd) < (unsigned)g.D && (unsigned)(h + kh) < (unsigned)g.H && (unsigned)(w + kw) < (unsigned)g.W)
          val = s[m + ((int64_t)kd * g.H + kh) * g.W + kw];
      }
      o.v[j] = from_f<T>(val);
    }
    st16(out + m * 32 + v0, o);
  }
}

// deterministic two-stage sum of a float32 vector (bias gradient of the 1-output-channel convolutions)
__global__ void __launch_bounds__(256) vecsum_partial_kernel(const float* __restrict__ v, double* __restrict__ ws, int64_t n) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += (double)v[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) vecsum_finish_kernel(const double* __restrict__ ws, float* __restrict__ out, int blocks) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < blocks; i += 256) s += ws[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = (float)s;
}

// Weight gradients of the 1-channel convolutions over 4x8x8 bricks, bf16 (the im2col of the scalar operand never leaves the CU):
//   first layer (1 -> Co):  dw[c][t] = sum_m dy[m][c] * x[m + delta_t]          act = dy, s = x,  flip = 0
//   heads       (C -> 1):   dw[c][t] = sum_m x[m][c]  * dy[m - delta_t]         act = x,  s = dy, flip = 1
// A block walks over a range of bricks: per brick it stages act[256 voxels][64 ch] (bf16, the weight-gradient tile layout) and
// the brick's halo of the float scalar field (6x10x10, zero outside the volume) in LDS; a wave takes two of the eight 32-voxel
// K-chunks, fetches the act fragments with transpose reads and BUILDS the im2col fragments (32 voxels x 16 taps) from eight
// scalar LDS reads per lane.  Per-block partial sums -> ws[block][c][32], reduced in fixed order by wgrad_reduce_kernel.
struct ScalarWgradParams {
  const bf16* act;    // [M][C]
  const float* s;     // [M]
  float* ws;          // [gridDim.x][C][32]
  int N, D, H, W, C;
  int nbricks, per_block, flip;
};

__global__ void __launch_bounds__(256) scalar_wgrad_brick_kernel(const ScalarWgradParams p) {
  using WT = WTile<bf16>;
  using WF = WFrag<bf16, true>;
  __shared__ __attribute__((aligned(16))) char at[256 * WT::ROWB];   // 32 KiB
  __shared__ float sh[6 * 10 * 10];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i0 = blockIdx.y * 64;
  const int b_beg = blockIdx.x * p.per_block;
  const int b_end = min(b_beg + p.per_block, p.nbricks);
  const int bw = p.W / 8, bh = p.H / 8, bd = p.D / 4;

  // staging roles: act piece q = tid + 256 i -> voxel q >> 3 (brick order d,h,w), 16-byte piece q & 7
  const int pc = tid & 7;
  const bool col_ok = (i0 + pc * 8) < p.C;
  uint32_t aoff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int v = (tid >> 3) + 32 * i;
    aoff[i] = (uint32_t)((((v >> 6) * p.H + ((v >> 3) & 7)) * p.W + (v & 7)) * p.C + (col_ok ? i0 + pc * 8 : 0)) * 2u;
  }
  u32x4 ra[8];
  float rs[3];
#define SW_LOAD(b_)                                                                                          \
  do {                                                                                                       \
    int t_ = (b_);                                                                                           \
    const int w0 = (t_ % bw) * 8; t_ /= bw;                                                                  \
    const int h0 = (t_ % bh) * 8; t_ /= bh;                                                                  \
    const int d0 = (t_ % bd) * 4; t_ /= bd;                                                                  \
    const int n = t_;                                                                                        \
    const int64_t base0 = (((int64_t)n * p.D + d0) * p.H + h0) * p.W + w0;                                   \
    const char* ab = reinterpret_cast<const char*>(p.act + base0 * p.C);                                     \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const u32x4*>(ab + aoff[i]);     \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                          \
      const int q = tid + 256 * i;                                                                           \
      const int hd = q / 100, hh = (q / 10) % 10, hw = q % 10;                                               \
      const int d = d0 + hd - 1, h = h0 + hh - 1, w = w0 + hw - 1;                                           \
      const bool ok = q < 600 && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W; \
      const float val = p.s[ok ? (((int64_t)n * p.D + d) * p.H + h) * p.W + w : base0];                      \
      rs[i] = ok ? val : 0.f;                                                                                \
    }                                                                                                        \
  } while (0)
#define SW_STORE()                                                                                           \
  do {                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                            \
      *reinterpret_cast<u32x4*>(at + WT::off((tid >> 3) + 32 * i, pc * 8)) = keep_if(col_ok, ra[i]);         \
    _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                            \
      if (tid + 256 * i < 600) sh[tid + 256 * i] = rs[i];                                                    \
  } while (0)

  f32x4 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the lane's tap of im2col fragment j: t = 16 j + lr (27..31: no such tap -> zero column); flip: tap 26 - t
  int toff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int t = 16 * j + lr;
    const int tt = p.flip ? 26 - t : t;
    toff[j] = t < 27 ? ((tt / 9) * 10 + (tt / 3) % 3) * 10 + tt % 3 : -1;
  }

  if (b_beg < b_end) {
    SW_LOAD(b_beg);
    SW_STORE();
  }
  __syncthreads();
  for (int b = b_beg; b < b_end; ++b) {
    SW_LOAD(b + 1 < b_end ? b + 1 : b);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int kc = wid * 2 + q;                       // 32-voxel K-chunk: brick rows 32 kc .. 32 kc + 31
      const int line = 4 * kc + lg;                     // the lane's 8 voxels: w = 0..7 of (d,h) line `line`
      const int hbase = ((line >> 3) * 10 + (line & 7)) * 10;
      bf16x8 fb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int e = 0; e < 8; ++e) fb[j][e] = toff[j] >= 0 ? (bf16)sh[hbase + toff[j] + e] : (bf16)0.f;
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const bf16x8 fa = WF::read(at + kc * (32 * WT::ROWB), a * 16, lane);
        acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[0], acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[1], acc[a][1], 0, 0, 0);
      }
    }
    __syncthreads();
    SW_STORE();
    __syncthreads();
  }
#undef SW_LOAD
#undef SW_STORE
  // D[c][t]: lane holds c = 16 a + 4 lg + r, t = 16 j + lr.  The four waves' sums meet in LDS (fixed order): one slab per block.
  float* red = reinterpret_cast<float*>(at);   // [4 waves][64 c][32 t] floats = the 32 KiB of the act tile (the loop ended with a barrier)
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cl = a * 16 + lg * 4 + r;
      red[(wid * 64 + cl) * 32 + lr] = acc[a][0][r];
      red[(wid * 64 + cl) * 32 + 16 + lr] = acc[a][1][r];
    }
  __syncthreads();
  float* out = p.ws + (int64_t)blockIdx.x * (int64_t)p.C * 32;
  for (int q = tid; q < 64 * 32; q += 256) {
    const int cl = q >> 5, c = i0 + cl;
    if (c < p.C) out[(int64_t)c * 32 + (q & 31)] = (red[q] + red[2048 + q]) + (red[4096 + q] + red[6144 + q]);
  }
}

struct SplitPlanUp2 {
  int splits;
  int64_t chunk;
};
SplitPlanUp2 plan_up2(int64_t M, int Cu, int Cv);

// Weight gradient of the 2x2x2 / stride-2 transposed convolution with ALL EIGHT taps in one block (bf16):
//   dW[t][ci][co] = sum_m x[m][ci] * dy[up2_row(m, t)][co]
// The generic kernel above gives every tap its own block, so the x tile of a K-step is fetched eight times and a K-step
// feeds only 4 MFMAs per wave between two barriers (L2-bound, ~300 TF).  Here a K-step stages the x tile once plus the eight
// dy tiles of the same 32 input voxels (each dy row is used exactly once overall) and runs 32 MFMAs per wave on them:
// 44 % fewer staged bytes, 8 x the work per barrier.  128 accumulator registers; 72 KiB LDS (two buffers), 2 blocks per CU.
__global__ void __launch_bounds__(256, 2) wgrad_up2_alltaps_kernel(const WgradParams p) {
  using WT = WTile<bf16>;
  using WF = WFrag<bf16, true>;
  constexpr int TILE_BYTES = 32 * WT::ROWB;   // 4 KiB: 32 voxels x 64 channels
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Us = smem;                      // [2][TILE]
  char* Vs = smem + 2 * TILE_BYTES;     // [2][8 taps][TILE]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wi = wid >> 1, wj = wid & 1;
  const int ntj = p.Cv / 64;
  const int i0 = (blockIdx.x / ntj) * 64, j0 = (blockIdx.x % ntj) * 64;
  const int64_t mbeg = (int64_t)blockIdx.y * p.chunk;
  const int64_t mend = (mbeg + p.chunk < p.M) ? (mbeg + p.chunk) : p.M;
  const bf16* __restrict__ U = reinterpret_cast<const bf16*>(p.u);
  const bf16* __restrict__ V = reinterpret_cast<const bf16*>(p.v);
  const Dims g = p.g;
  const int chunk16 = tid & 7, rowp = tid >> 3;   // 16-byte piece of the 64-channel row, voxel row of the K-step
  const int ucol = chunk16 * 8;

  f32x4 acc[8][2][2];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ru, rv[8];
  bool live = false;
  // the voxel coordinates of this thread's row are carried from K-step to K-step (+32 voxels) instead of decoded with divisions
  int64_t um = mbeg + rowp;
  int un, ud, uh, uw;
  decode_voxel(um < p.M ? um : 0, g, un, ud, uh, uw);
  const int us_w = 32 % g.W, us_h = (32 / g.W) % g.H, us_d = (32 / (g.W * g.H)) % g.D, us_n = 32 / (g.W * g.H * g.D);
  int64_t vm0 = mbeg;
  int vn0, vd0, vh0, vw0;   // coordinates of row mbeg (the stand-in for dead rows)
  decode_voxel(mbeg < p.M ? mbeg : 0, g, vn0, vd0, vh0, vw0);
#define WU_LOAD(ms_)                                                                            \
  do {                                                                                          \
    live = um < mend;                                                                           \
    const int64_t m = live ? um : vm0;                                                          \
    ru = *reinterpret_cast<const u32x4*>(U + m * p.Cu + i0 + ucol);                             \
    const int n = live ? un : vn0, d = live ? ud : vd0, h = live ? uh : vh0, w = live ? uw : vw0; \
    um += 32;                                                                                   \
    uw += us_w;                                                                                 \
    if (uw >= g.W) { uw -= g.W; uh += 1; }                                                      \
    uh += us_h;                                                                                 \
    if (uh >= g.H) { uh -= g.H; ud += 1; }                                                      \
    ud += us_d;                                                                                 \
    if (ud >= g.D) { ud -= g.D; un += 1; }                                                      \
    un += us_n;                                                                                 \
    const bf16* v0 = V + up2_row(n, d, h, w, 0, g) * p.Cv + j0 + ucol;                          \
    _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                             \
      const int64_t dlt = ((int64_t)(t >> 2) * (2 * g.H) + ((t >> 1) & 1)) * (2 * g.W) + (t & 1); \
      rv[t] = *reinterpret_cast<const u32x4*>(v0 + dlt * p.Cv);                                 \
    }                                                                                           \
  } while (0)
#define WU_STORE(buf_)                                                                          \
  do {                                                                                          \
    *reinterpret_cast<u32x4*>(Us + (buf_)*TILE_BYTES + WT::off(rowp, ucol)) = keep_if(live, ru); \
    _Pragma("unroll") for (int t = 0; t < 8; ++t)                                               \
      *reinterpret_cast<u32x4*>(Vs + ((buf_)*8 + t) * TILE_BYTES + WT::off(rowp, ucol)) = keep_if(live, rv[t]); \
  } while (0)

  const int64_t nsteps = (mend > mbeg) ? (mend - mbeg + 31) / 32 : 0;
  if (nsteps > 0) {
    WU_LOAD(mbeg);
    WU_STORE(0);
  }
  __syncthreads();
  for (int64_t s = 0; s < nsteps; ++s) {
    const int cur = (int)(s & 1);
    const int64_t sn = (s + 1 < nsteps) ? s + 1 : s;
    WU_LOAD(mbeg + sn * 32);
    __builtin_amdgcn_sched_barrier(0);
    {
      const char* ut = Us + cur * TILE_BYTES;
      WF::Frag fa[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = WF::read(ut, wi * 32 + a * 16, lane);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const char* vt = Vs + (cur * 8 + t) * TILE_BYTES;
        WF::Frag fb[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = WF::read(vt, wj * 32 + b * 16, lane);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) WF::mma(fa[a], fb[b], acc[t][a][b]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    WU_STORE(cur ^ 1);
    __syncthreads();
  }
#undef WU_LOAD
#undef WU_STORE

#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float* __restrict__ out = p.ws + ((int64_t)blockIdx.y * 8 + t) * (int64_t)p.Cu * p.Cv;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + wi * 32 + a * 16 + (lane >> 4) * 4 + r;
          const int j = j0 + wj * 32 + b * 16 + (lane & 15);
          out[(int64_t)i * p.Cv + j] = acc[t][a][b][r];
        }
  }
}

// WG_UPC with the eight coarse taps of a phase in ONE block: the U tile (dy0 at the phase's fine voxels) is staged once per K-step and
// multiplied against the eight shifted x tiles (128 MFMAs per nine staged tiles; the one-tap-per-block form above stages two tiles per
// 16 MFMAs and ran at 150-190 TFLOP/s).  Block = 64 (co) x 64 (ci) tile of one phase and one voxel split; grid (tiles, 8 phases, splits);
// slabs ws[split][p * 8 + q][Cu][Cv] as wgrad_kernel writes them.
template <typename T, bool TR>
__global__ void __launch_bounds__(256, sizeof(T) == 2 ? PCRL_OCC2 : 1) wgrad_upc8_kernel(const WgradParams p) {
  using WT = WTile<T>;
  using WF = WFrag<T, TR>;
  constexpr int KS = 32;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int CPR = WT::ROWB / 16;  // 8 / 16
  constexpr int RPP = 256 / CPR;      // 32 / 16
  constexpr int NP = KS / RPP;        // 1 / 2
  constexpr int TILE_BYTES = KS * WT::ROWB;

  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][U, V0..V7]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wi = wid >> 1, wj = wid & 1;
  const int ntj = (p.Cv + 63) / 64;
  // 1-D grid: the tiles x 8 phases blocks of ONE voxel split walk the same x and dy0 rows -- they get ids congruent mod 8 (one XCD, one
  // L2) and consecutive there; the splits are dealt round-robin to the XCDs.  (With (tile, phase, split) as a 3-D grid the blocks of a
  // split were spread over all eight L2s: 750 MB fetched per launch for 130 MB of operands, rocprofv3 r02e.)
  const int ntile = ((p.Cu + 63) / 64) * ntj, per = ntile * 8, nsplit = p.q.ntaps;   // q.ntaps carries the split count here
  int member, zsplit;
  if ((nsplit & 7) == 0) {
    const int k = blockIdx.x >> 3;
    member = k % per;
    zsplit = (k / per) * 8 + (blockIdx.x & 7);
  } else {
    member = blockIdx.x % per;
    zsplit = blockIdx.x / per;
  }
  const int tix = member % ntile, ph = member / ntile;
  const int i0 = (tix / ntj) * 64, j0 = (tix % ntj) * 64;
  const int pd = (ph >> 2) & 1, phh = (ph >> 1) & 1, pw = ph & 1;
  const int64_t mbeg = (int64_t)zsplit * p.chunk;
  const int64_t mend = (mbeg + p.chunk < p.M) ? (mbeg + p.chunk) : p.M;
  const T* __restrict__ U = reinterpret_cast<const T*>(p.u);
  const T* __restrict__ V = reinterpret_cast<const T*>(p.v);
  const Dims g = p.g;
  const int chunk16 = tid % CPR, rowp = tid / CPR;
  const int ucol = chunk16 * VEC;
  const bool u_ok = (i0 + ucol) < p.Cu, v_ok = (j0 + ucol) < p.Cv;
  const int ucol_u = u_ok ? ucol : 0, vcol = j0 + (v_ok ? ucol : 0);

  f32x4 acc[8][2][2];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[q][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int cn[NP], cd[NP], ch[NP], cw[NP];
  int64_t cm[NP];
  const int sw = KS % g.W, sh = (KS / g.W) % g.H, sd = (KS / (g.W * g.H)) % g.D, sn = KS / (g.W * g.H * g.D);
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    cm[ps] = mbeg + ps * RPP + rowp;
    const int64_t mc = cm[ps] < p.M ? cm[ps] : 0;
    decode_voxel(mc, g, cn[ps], cd[ps], ch[ps], cw[ps]);
  }
  u32x4 ru[NP], rv[8][NP];
  uint32_t uokb = 0, vokb = 0;   // vokb: bit q * NP + ps
#define W8_LOAD()                                                                                 \
  do {                                                                                            \
    uokb = 0;                                                                                     \
    vokb = 0;                                                                                     \
    _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                           \
      const bool live = cm[ps] < mend;                                                            \
      const int n_ = live ? cn[ps] : 0, d_ = live ? cd[ps] : 0, h_ = live ? ch[ps] : 0, w_ = live ? cw[ps] : 0; \
      const int64_t urow = (((int64_t)n_ * (2 * g.D) + 2 * d_ + pd) * (2 * g.H) + 2 * h_ + phh) * (2 * g.W) + 2 * w_ + pw; \
      ru[ps] = *reinterpret_cast<const u32x4*>(U + urow * p.Cu + i0 + ucol_u);                    \
      uokb |= (uint32_t)(live && u_ok) << ps;                                                     \
      _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                             \
        const int dd = d_ + pd - 1 + (q >> 2), hh = h_ + phh - 1 + ((q >> 1) & 1), ww = w_ + pw - 1 + (q & 1); \
        const bool in = live && (unsigned)dd < (unsigned)g.D && (unsigned)hh < (unsigned)g.H && (unsigned)ww < (unsigned)g.W; \
        const int64_t vrow = in ? (((int64_t)n_ * g.D + dd) * g.H + hh) * g.W + ww : (int64_t)0;  \
        rv[q][ps] = *reinterpret_cast<const u32x4*>(V + vrow * p.Cv + vcol);                      \
        vokb |= (uint32_t)(in && v_ok) << (q * NP + ps);                                          \
      }                                                                                           \
      cm[ps] += KS;                                                                               \
      cw[ps] += sw;                                                                               \
      if (cw[ps] >= g.W) { cw[ps] -= g.W; ch[ps] += 1; }                                          \
      ch[ps] += sh;                                                                               \
      if (ch[ps] >= g.H) { ch[ps] -= g.H; cd[ps] += 1; }                                          \
      cd[ps] += sd;                                                                               \
      if (cd[ps] >= g.D) { cd[ps] -= g.D; cn[ps] += 1; }                                          \
      cn[ps] += sn;                                                                               \
    }                                                                                             \
  } while (0)
#define W8_STORE(buf_)                                                                            \
  do {                                                                                            \
    char* base_ = smem + (buf_) * (9 * TILE_BYTES);                                               \
    _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                           \
      const int row = ps * RPP + rowp;                                                            \
      *reinterpret_cast<u32x4*>(base_ + WT::off(row, ucol)) = keep_if((uokb >> ps) & 1u, ru[ps]); \
      _Pragma("unroll") for (int q = 0; q < 8; ++q)                                               \
        *reinterpret_cast<u32x4*>(base_ + (1 + q) * TILE_BYTES + WT::off(row, ucol)) = keep_if((vokb >> (q * NP + ps)) & 1u, rv[q][ps]); \
    }                                                                                             \
  } while (0)

  const int64_t nsteps = (mend > mbeg) ? (mend - mbeg + KS - 1) / KS : 0;
  if (nsteps > 0) {
    W8_LOAD();
    W8_STORE(0);
  }
  __syncthreads();
  for (int64_t s = 0; s < nsteps; ++s) {
    const int cur = (int)(s & 1);
    W8_LOAD();   // the last iteration stages rows past `mend`: dead, zeroed at the LDS store
    __builtin_amdgcn_sched_barrier(0);
    {
      const char* base = smem + cur * (9 * TILE_BYTES);
      typename WF::Frag fa[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = WF::read(base, wi * 32 + a * 16, lane);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        typename WF::Frag fb[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = WF::read(base + (1 + q) * TILE_BYTES, wj * 32 + b * 16, lane);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) WF::mma(fa[a], fb[b], acc[q][a][b]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    W8_STORE(cur ^ 1);
    __syncthreads();
  }
#undef W8_LOAD
#undef W8_STORE
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    float* __restrict__ out = p.ws + ((int64_t)zsplit * p.taps + ph * 8 + q) * (int64_t)p.Cu * p.Cv;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + wi * 32 + a * 16 + (lane >> 4) * 4 + r;
          const int j = j0 + wj * 32 + b * 16 + (lane & 15);
          if (i < p.Cu && j < p.Cv) out[(int64_t)i * p.Cv + j] = acc[q][a][b][r];
        }
  }
}

// 2D weight gradient of a 3 x 3 convolution (any stride, optional nearest x2 upsample in front) with ONE KERNEL ROW per block: the dy tile of a
// K-step (64 output pixels x 64 output channels) is staged once and multiplied against the THREE x tiles of taps (kh, 0), (kh, 1), (kh, 2)
// (64 pixels x 64 source channels each, gathered at (oh * s - pad + kh, ow * s - pad + kw); neighbouring pixels' kw taps overlap, so two of the three
// gathers hit L1 / L2).  The flattened-tap form of wgrad_kernel gives every (tap, 64-channel) column tile its own block, which streams dy and x from
// memory again: rocprofv3 counters on the stride-2 64 -> 128 layer at 128^2 showed 1.2 GB of L2 -> HBM requests per launch for 200 MB of operands
// (TCC miss rate 87 %) and 18 VALU instructions per MFMA.  Here the operands cross the fabric three times instead of nine, the pixel coordinates are
// carried (not decoded) from step to step, and a step is 24 MFMAs per wave between two barriers instead of 4.
// grid (co tiles x ci tiles, 3 kernel rows, splits); slabs ws[split][Cu][9 * CiP] with column (kh * 3 + kw) * CiP + ci, as wgrad_kernel<WG_CONV2D> writes them.
template <bool TR>
__global__ void __launch_bounds__(256, PCRL_OCC2) wgrad2d_row3_kernel(const WgradParams p) {
  using T = bf16;
  using WT = WTile<T>;
  using WF = WFrag<T, TR>;
  constexpr int KS = 64, VEC = 8, CPR = 8, RPP = 32, NP = KS / RPP;
  constexpr int TILE_BYTES = KS * WT::ROWB;   // 8 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][U, V0, V1, V2]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wi = wid >> 1, wj = wid & 1;
  const int CiP = p.q.CiP, ntc = (CiP + 63) / 64;
  const int i0 = (blockIdx.x / ntc) * 64, jc0 = (blockIdx.x % ntc) * 64;
  const int kh = blockIdx.y;
  const int64_t mbeg = (int64_t)blockIdx.z * p.chunk;
  const int64_t mend = (mbeg + p.chunk < p.M) ? (mbeg + p.chunk) : p.M;
  const T* __restrict__ U = reinterpret_cast<const T*>(p.u);
  const T* __restrict__ V = reinterpret_cast<const T*>(p.v);
  const Dims g = p.g;
  const int chunk16 = tid % CPR, rowp = tid / CPR;
  const int ucol = chunk16 * VEC;
  const bool u_ok = (i0 + ucol) < p.Cu, v_ok = (jc0 + ucol) < CiP;
  const int ucol_u = u_ok ? ucol : 0, vcol = jc0 + (v_ok ? ucol : 0);
  const int Hl = p.q.up ? 2 * p.q.Hs : p.q.Hs, Wl = p.q.up ? 2 * p.q.Ws : p.q.Ws;

  f32x4 acc[3][2][2];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[q][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int cn[NP], ch[NP], cw[NP];
  int64_t cm[NP];
  const int sw = KS % g.W, sh = (KS / g.W) % g.H, sn = KS / (g.W * g.H);
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    cm[ps] = mbeg + ps * RPP + rowp;
    const int64_t mc = cm[ps] < p.M ? cm[ps] : 0;
    int dd;
    decode_voxel(mc, g, cn[ps], dd, ch[ps], cw[ps]);
  }
  u32x4 ru[NP], rv[3][NP];
  uint32_t uokb = 0, vokb = 0;   // vokb: bit q * NP + ps
  // unconditional loads from clamped addresses; validity is applied at the LDS store (see wgrad_kernel)
#define R3_LOAD()                                                                                 \
  do {                                                                                            \
    uokb = 0;                                                                                     \
    vokb = 0;                                                                                     \
    _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                           \
      const bool live = cm[ps] < mend;                                                            \
      const int64_t m = live ? cm[ps] : mbeg;                                                     \
      ru[ps] = *reinterpret_cast<const u32x4*>(U + m * p.Cu + i0 + ucol_u);                       \
      uokb |= (uint32_t)(live && u_ok) << ps;                                                     \
      const int ih0 = ch[ps] * p.q.stride - p.q.pad + kh, iw0 = cw[ps] * p.q.stride - p.q.pad;    \
      const bool rok = live && (unsigned)ih0 < (unsigned)Hl;                                      \
      const int ihs = p.q.up ? (ih0 >> 1) : ih0;                                                  \
      _Pragma("unroll") for (int q = 0; q < 3; ++q) {                                             \
        const int iw = iw0 + q;                                                                   \
        const bool in = rok && (unsigned)iw < (unsigned)Wl;                                       \
        const int iws = p.q.up ? (iw >> 1) : iw;                                                  \
        const int64_t vrow = in ? ((int64_t)cn[ps] * p.q.Hs + ihs) * p.q.Ws + iws : (int64_t)0;   \
        rv[q][ps] = *reinterpret_cast<const u32x4*>(V + vrow * CiP + vcol);                       \
        vokb |= (uint32_t)(in && v_ok) << (q * NP + ps);                                          \
      }                                                                                           \
      cm[ps] += KS;                                                                               \
      cw[ps] += sw;                                                                               \
      if (cw[ps] >= g.W) { cw[ps] -= g.W; ch[ps] += 1; }                                          \
      ch[ps] += sh;                                                                               \
      if (ch[ps] >= g.H) { ch[ps] -= g.H; cn[ps] += 1; }                                          \
      cn[ps] += sn;                                                                               \
    }                                                                                             \
  } while (0)
#define R3_STORE(buf_)                                                                            \
  do {                                                                                            \
    char* base_ = smem + (buf_) * (4 * TILE_BYTES);                                               \
    _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) {                                           \
      const int row = ps * RPP + rowp;                                                            \
      *reinterpret_cast<u32x4*>(base_ + WT::off(row, ucol)) = keep_if((uokb >> ps) & 1u, ru[ps]); \
      _Pragma("unroll") for (int q = 0; q < 3; ++q)                                               \
        *reinterpret_cast<u32x4*>(base_ + (1 + q) * TILE_BYTES + WT::off(row, ucol)) = keep_if((vokb >> (q * NP + ps)) & 1u, rv[q][ps]); \
    }                                                                                             \
  } while (0)

  const int64_t nsteps = (mend > mbeg) ? (mend - mbeg + KS - 1) / KS : 0;
  if (nsteps > 0) {
    R3_LOAD();
    R3_STORE(0);
  }
  __syncthreads();
  for (int64_t s = 0; s < nsteps; ++s) {
    const int cur = (int)(s & 1);
    R3_LOAD();   // the last iteration stages rows past `mend`: dead (the clamped pixel of row m = mbeg), zeroed at the LDS store
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < KS / 32; ++kk) {
      const char* base = smem + cur * (4 * TILE_BYTES) + kk * 32 * WT::ROWB;
      typename WF::Frag fa[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = WF::read(base, wi * 32 + a * 16, lane);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        typename WF::Frag fb[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = WF::read(base + (1 + q) * TILE_BYTES, wj * 32 + b * 16, lane);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) WF::mma(fa[a], fb[b], acc[q][a][b]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    R3_STORE(cur ^ 1);
    __syncthreads();
  }
#undef R3_LOAD
#undef R3_STORE
  float* __restrict__ out = p.ws + (int64_t)blockIdx.z * (int64_t)p.Cu * p.Cv;
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + wi * 32 + a * 16 + (lane >> 4) * 4 + r;
          const int jc = jc0 + wj * 32 + b * 16 + (lane & 15);
          if (i < p.Cu && jc < CiP) out[(int64_t)i * p.Cv + (kh * 3 + q) * CiP + jc] = acc[q][a][b][r];
        }
}

// split plan of the all-taps kernel: whole rounds of 512 blocks (2 per CU), at least 16 K-steps per block
SplitPlanUp2 plan_up2(int64_t M, int Cu, int Cv) {
  const int64_t tiles = (int64_t)(Cu / 64) * (Cv / 64);
  const int64_t steps = (M + 31) / 32;
  int64_t splits = 512 / tiles;
  if (splits < 1) splits = 1;
  if (splits > (steps + 15) / 16) splits = (steps + 15) / 16;
  if (splits < 1) splits = 1;
  const int64_t per = (steps + splits - 1) / splits;
  splits = (steps + per - 1) / per;
  return SplitPlanUp2{(int)splits, per * 32};
}

struct SplitPlan {
  int splits;
  int64_t chunk;
};
SplitPlan plan_splits(int64_t M, int Cu, int Cv, int taps) {
  const int64_t tiles = (int64_t)((Cu + 63) / 64) * ((Cv + 63) / 64) * taps;
  const int64_t steps = (M + 31) / 32;
  int64_t splits = 2048 / tiles;
  if (splits < 1) splits = 1;
  const int64_t max_splits = (steps + 7) / 8;  // at least 8 K-steps per block
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int64_t steps_per = (steps + splits - 1) / splits;
  splits = (steps + steps_per - 1) / steps_per;
  return SplitPlan{(int)splits, steps_per * 32};
}

std::atomic<int> g_wgrad_tr{1};  // bf16 fragment fetch: 1 = ds_read_b64_tr_b16, 0 = scalar LDS reads

template <int GEOM>
int run_wgrad(const void* u, const void* v, float* dw_ref, void* ws, size_t ws_bytes, Dims g, int Cu, int Cv, int taps,
              int dtype, hipStream_t stream, int Cv_out = -1, Wg2d q = Wg2d{0, 0, 1, 1, 0, 0, 1, 1}) {
  if (Cv_out < 0) Cv_out = Cv;
  const int64_t M = (int64_t)g.N * g.D * g.H * g.W;
  const SplitPlan sp = plan_splits(M, Cu, Cv, taps);
  const size_t need = (size_t)sp.splits * taps * Cu * Cv * sizeof(float);
  if (ws_bytes < need || !ws) return pcrl_fail(PCRL_EWORKSPACE, "wgrad: workspace %zu < %zu", ws_bytes, need);
  WgradParams p{u, v, (float*)ws, g, M, Cu, Cv, taps, sp.chunk, q};
  dim3 grid((unsigned)(((Cu + 63) / 64) * ((Cv + 63) / 64)), (unsigned)taps, (unsigned)sp.splits);
  if (dtype == PCRL_BF16) {
    if (g_wgrad_tr) hipLaunchKernelGGL((wgrad_kernel<bf16, GEOM, true>), grid, dim3(256), 4 * 32 * 128, stream, p);
    else hipLaunchKernelGGL((wgrad_kernel<bf16, GEOM, false>), grid, dim3(256), 4 * 32 * 128, stream, p);
  } else if (dtype == PCRL_F32) {
    hipLaunchKernelGGL((wgrad_kernel<float, GEOM, false>), grid, dim3(256), 4 * 32 * 256, stream, p);
  } else {
    return pcrl_fail(PCRL_EINVAL, "wgrad: bad dtype %d", dtype);
  }
  if (int e = pcrl_check_launch("wgrad")) return e;
  return launch_wgrad_reduce((const float*)ws, dw_ref, sp.splits, taps, Cu, Cv, Cv_out, stream);
}


// ---- brick path of the 1-channel weight gradients ----
bool scalar_brick_ok(int D, int H, int W, int C, int taps, int dtype) {
  return dtype == PCRL_BF16 && g_wgrad_tr && taps == 27 && D % 4 == 0 && H % 8 == 0 && W % 8 == 0 && C % 8 == 0;
}
int scalar_brick_blocks(int64_t nbricks, int C) {
  const int ytiles = (C + 63) / 64;
  int64_t nb = 512 / ytiles;               // ~2 blocks per CU; every block leaves one partial slab for the second pass
  if (nb > nbricks) nb = nbricks;
  if (nb < 1) nb = 1;
  return (int)nb;
}
size_t scalar_brick_ws_bytes(int N, int D, int H, int W, int C) {
  const int64_t nbricks = (int64_t)N * (D / 4) * (H / 8) * (W / 8);
  return 8192 + (size_t)scalar_brick_blocks(nbricks, C) * C * 32 * sizeof(float);
}
int scalar_brick_launch(const void* act, const float* sfield, float* dw_ref, char* ws, int N, int D, int H, int W, int C, int flip,
                        hipStream_t stream) {
  const int64_t nbricks = (int64_t)N * (D / 4) * (H / 8) * (W / 8);
  const int nb = scalar_brick_blocks(nbricks, C);
  const int per = (int)((nbricks + nb - 1) / nb);
  const int blocks = (int)((nbricks + per - 1) / per);
  float* part = reinterpret_cast<float*>(ws + 8192);
  ScalarWgradParams p{(const bf16*)act, sfield, part, N, D, H, W, C, (int)nbricks, per, flip};
  hipLaunchKernelGGL(scalar_wgrad_brick_kernel, dim3((unsigned)blocks, (unsigned)((C + 63) / 64)), dim3(256), 0, stream, p);
  if (int e = pcrl_check_launch("scalar_wgrad_brick")) return e;
  return launch_wgrad_reduce((const float*)part, dw_ref, blocks, 1, C, 32, 27, stream);
}
}  // namespace

// Test hook (not part of the drop-in surface): select the bf16 fragment-fetch path.
extern "C" void pcrl_debug_set_wgrad_tr(int on) { g_wgrad_tr = on; }

// LDS-halo brick kernel (wgrad_brick.hip)
bool pcrl_wgrad_brick_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype);
int pcrl_wgrad_brick_splits(int N, int D, int H, int W, int Ci, int Co);
int pcrl_wgrad_brick_launch(const void* x, const void* dy, float* ws, int N, int D, int H, int W, int Ci, int Co, hipStream_t stream);
int pcrl_wgrad_brick_slabs(int N, int D, int H, int W, int Ci, int Co);   // partial slabs the launch writes (<= pcrl_wgrad_brick_splits)
void pcrl_wgrad_brick_set_xcd(int on, int order, int tiles);

static std::atomic<int> g_wgrad_impl{0};  // 0 = auto (brick kernel where eligible), 1 = always the gather kernel, 2 = brick kernel on its 2-D grid (no XCD co-location)
extern "C" void pcrl_debug_set_wgrad_impl(int impl) {
  // experiments: 4 = co-located launch with the old walk order, 5 = 2-D grid with the new walk order, 6 = co-located launch with 64 x 64 tiles only
  g_wgrad_impl = impl == 1 ? 1 : 0;
  pcrl_wgrad_brick_set_xcd(impl == 0 || impl == 4 || impl == 6, impl == 0 || impl == 5 || impl == 6, impl != 6);
}

// ---- fused ConvTranspose3d(k2,s2) -> Conv3d(3x3x3): gradient of the COMPOSED weights (internal; the C ABI is in upconv_fused.hip) ----
// dweff[co][ci][t], t = phase * 8 + coarse tap = sum over coarse voxels v of dy0[2v + p][co] * x[v + p - 1 + q][ci]
static SplitPlan plan_upc8(int64_t M, int Cu, int Cv) {
  const int64_t tiles = (int64_t)((Cu + 63) / 64) * ((Cv + 63) / 64) * 8;
  const int64_t steps = (M + 31) / 32;
  int64_t splits = 1024 / tiles;
  if (splits < 1) splits = 1;
  const int64_t max_splits = (steps + 7) / 8;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits >= 8) splits &= ~(int64_t)7;      // whole rounds of the eight XCDs (see the kernel's block map)
  const int64_t per = (steps + splits - 1) / splits;
  const int64_t used = (steps + per - 1) / per;
  if (!(splits >= 8 && (used & 7))) splits = used;   // keep the multiple of eight even if the last splits come out empty
  return SplitPlan{(int)splits, per * 32};
}
size_t pcrl_upc_wgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Co) {
  const SplitPlan sp = plan_upc8((int64_t)N * D * H * W, Co, Ci);
  return (size_t)sp.splits * 64 * Co * Ci * sizeof(float);
}
int pcrl_upc_wgrad_launch(const void* dy0, const void* x, float* dweff, void* ws, size_t ws_bytes, int N, int D, int H, int W, int Ci, int Co,
                          int dtype, hipStream_t stream, bool accumulate) {
  const int64_t M = (int64_t)N * D * H * W;
  const SplitPlan sp = plan_upc8(M, Co, Ci);
  const size_t need = (size_t)sp.splits * 64 * Co * Ci * sizeof(float);
  if (ws_bytes < need || !ws) return pcrl_fail(PCRL_EWORKSPACE, "upconv wgrad: workspace %zu < %zu", ws_bytes, need);
  WgradParams p{dy0, x, (float*)ws, Dims{N, D, H, W}, M, Co, Ci, 64, sp.chunk, Wg2d{0, 0, 1, 1, 0, 0, 1, sp.splits}};
  const dim3 grid((unsigned)(((Co + 63) / 64) * ((Ci + 63) / 64) * 8 * sp.splits));
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_upc8_kernel<bf16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 9 * 32 * 128);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_upc8_kernel<bf16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 9 * 32 * 128);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_upc8_kernel<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 9 * 32 * 256);
  });
  if (dtype == PCRL_BF16) {
    if (g_wgrad_tr) hipLaunchKernelGGL((wgrad_upc8_kernel<bf16, true>), grid, dim3(256), 2 * 9 * 32 * 128, stream, p);
    else hipLaunchKernelGGL((wgrad_upc8_kernel<bf16, false>), grid, dim3(256), 2 * 9 * 32 * 128, stream, p);
  } else if (dtype == PCRL_F32) {
    hipLaunchKernelGGL((wgrad_upc8_kernel<float, false>), grid, dim3(256), 2 * 9 * 32 * 256, stream, p);
  } else {
    return pcrl_fail(PCRL_EINVAL, "upconv wgrad: bad dtype %d", dtype);
  }
  if (int e = pcrl_check_launch("upconv wgrad")) return e;
  return launch_wgrad_reduce((const float*)ws, dweff, sp.splits, 64, Co, Ci, Ci, stream, accumulate);
}

// the same gradient, dweff[co][ci][p * 8 + q], from the brick kernel (wgrad_brick.hip, composed up-conv mode) where it tiles the coarse grid
bool pcrl_wgrad_brick_upc_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype);
int pcrl_wgrad_brick_upc_slabs(int N, int D, int H, int W, int Ci, int Co);
int pcrl_wgrad_brick_upc_launch(const void* x, const void* dy0, float* ws, int N, int D, int H, int W, int Ci, int Co, hipStream_t stream);
bool pcrl_upc_wgrad_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return g_wgrad_impl == 0 && g_wgrad_tr && pcrl_wgrad_brick_upc_eligible(N, D, H, W, Ci, Co, dtype);
}
size_t pcrl_upc_wgrad3_ws_bytes(int N, int D, int H, int W, int Ci, int Co) {
  return (size_t)pcrl_wgrad_brick_upc_slabs(N, D, H, W, Ci, Co) * 27 * 8 * Co * Ci * sizeof(float);
}
// Second pass of the brick form: the slabs hold the zero-embedded 3x3x3 layout ws[z][27][8 * Co][Ci] of which a phase's rows carry 8
// meaningful taps (tap (p + q) per axis; the planes a phase does not use were never written).  Sum those over z, in slab order, straight
// into the compact layout of the gather form, out[co][ci][p * 8 + q] -- 8 / 27 of the slab bytes, no intermediate [8 * Co][Ci][27] array.
// Block = one row i = p * Co + co and 64 columns ci: thread (q, lane) reads 128-byte runs, the tile is transposed through LDS so that the
// eight q of a (co, ci) leave as one 32-byte run.
__global__ void __launch_bounds__(256) wgrad_reduce_upc_kernel(const float* __restrict__ ws, float* __restrict__ out, int splits, int Co, int Ci,
                                                               bool accumulate) {
  __shared__ float tile[64][9];
  const int i = blockIdx.y, p = i / Co, co = i - p * Co, j0 = blockIdx.x * 64;
  const int q = threadIdx.x >> 5, jl = threadIdx.x & 31;
  const int tap = (((p >> 2) & 1) + ((q >> 2) & 1)) * 9 + (((p >> 1) & 1) + ((q >> 1) & 1)) * 3 + ((p & 1) + (q & 1));
  const int64_t per = (int64_t)8 * Co * Ci, zs = 27 * per;
  const float* src = ws + (int64_t)tap * per + (int64_t)i * Ci + j0 + jl;
  double a0 = 0.0, a1 = 0.0;
  const bool ok0 = j0 + jl < Ci, ok1 = j0 + jl + 32 < Ci;
  for (int z = 0; z < splits; ++z) {
    if (ok0) a0 += (double)src[(int64_t)z * zs];
    if (ok1) a1 += (double)src[(int64_t)z * zs + 32];
  }
  tile[jl][q] = (float)a0;
  tile[jl + 32][q] = (float)a1;
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int j = h * 32 + (threadIdx.x >> 3), qq = threadIdx.x & 7;
    if (j0 + j < Ci) {
      float* dst = out + ((int64_t)co * Ci + j0 + j) * 64 + p * 8 + qq;
      *dst = accumulate ? *dst + tile[j][qq] : tile[j][qq];
    }
  }
}
int pcrl_upc_wgrad3_launch(const void* dy0, const void* x, float* dweff, void* ws, size_t ws_bytes, int N, int D, int H, int W, int Ci, int Co,
                           hipStream_t stream, bool accumulate) {
  const int splits = pcrl_wgrad_brick_upc_slabs(N, D, H, W, Ci, Co);
  const size_t need = (size_t)splits * 27 * 8 * Co * Ci * sizeof(float);
  if (!ws || ws_bytes < need) return pcrl_fail(PCRL_EWORKSPACE, "upconv wgrad (brick): workspace %zu < %zu", ws_bytes, need);
  if (int e = pcrl_wgrad_brick_upc_launch(x, dy0, (float*)ws, N, D, H, W, Ci, Co, stream)) return e;
  hipLaunchKernelGGL(wgrad_reduce_upc_kernel, dim3((unsigned)((Ci + 63) / 64), (unsigned)(8 * Co)), dim3(256), 0, stream, (const float*)ws, dweff, splits, Co,
                     Ci, accumulate);
  return pcrl_check_launch("upconv wgrad (brick, reduce)");
}

extern "C" size_t pcrl_conv3d_k3_wgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Co) {
  const SplitPlan sp = plan_splits((int64_t)N * D * H * W, Co, Ci, 27);
  size_t a = (size_t)sp.splits * 27 * Co * Ci * sizeof(float);
  if (pcrl_wgrad_brick_eligible(N, D, H, W, Ci, Co, PCRL_BF16)) {
    const size_t b = (size_t)pcrl_wgrad_brick_splits(N, D, H, W, Ci, Co) * 27 * Co * Ci * sizeof(float);
    if (b > a) a = b;
  }
  return a;
}

extern "C" int pcrl_conv3d_k3_wgrad(const void* x, const void* dy, float* dw_ref, void* ws, size_t ws_bytes,
                                    int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && dy && dw_ref, "conv3d_k3_wgrad: null pointer");
  PCRL_REQUIRE(Ci > 0 && Co > 0 && Ci % 32 == 0 && Co % 32 == 0, "conv3d_k3_wgrad: channels must be multiples of 32 (Ci=%d Co=%d)", Ci, Co);
  if (g_wgrad_impl == 0 && pcrl_wgrad_brick_eligible(N, D, H, W, Ci, Co, dtype)) {
    const int splits = pcrl_wgrad_brick_slabs(N, D, H, W, Ci, Co);
    const size_t need = (size_t)splits * 27 * Co * Ci * sizeof(float);
    if (!ws || ws_bytes < need) return pcrl_fail(PCRL_EWORKSPACE, "conv3d_k3_wgrad: workspace %zu < %zu", ws_bytes, need);
    if (int e = pcrl_wgrad_brick_launch(x, dy, (float*)ws, N, D, H, W, Ci, Co, as_stream(stream))) return e;
    return launch_wgrad_reduce((const float*)ws, dw_ref, splits, 27, Co, Ci, Ci, as_stream(stream));
  }
  return run_wgrad<WG_CONV3>(dy, x, dw_ref, ws, ws_bytes, Dims{N, D, H, W}, Co, Ci, 27, dtype, as_stream(stream));
}

static bool up2_alltaps_ok(int Ci, int Co, int dtype) { return dtype == PCRL_BF16 && g_wgrad_tr && Ci % 64 == 0 && Co % 64 == 0; }

extern "C" size_t pcrl_convt3d_k2s2_wgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Co) {
  const int64_t M = (int64_t)N * D * H * W;
  const SplitPlan sp = plan_splits(M, Ci, Co, 8);
  size_t need = (size_t)sp.splits * 8 * Co * Ci * sizeof(float);
  if (Ci % 64 == 0 && Co % 64 == 0) {
    const size_t b = (size_t)plan_up2(M, Ci, Co).splits * 8 * Co * Ci * sizeof(float);
    if (b > need) need = b;
  }
  return need;
}

extern "C" int pcrl_convt3d_k2s2_wgrad(const void* x, const void* dy, float* dw_ref, void* ws, size_t ws_bytes,
                                       int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && dy && dw_ref, "convt3d_k2s2_wgrad: null pointer");
  PCRL_REQUIRE(Ci > 0 && Co > 0 && Ci % 32 == 0 && Co % 32 == 0, "convt3d_k2s2_wgrad: channels must be multiples of 32 (Ci=%d Co=%d)", Ci, Co);
  if (g_wgrad_impl == 0 && up2_alltaps_ok(Ci, Co, dtype)) {
    const int64_t M = (int64_t)N * D * H * W;
    const SplitPlanUp2 sp = plan_up2(M, Ci, Co);
    const size_t need = (size_t)sp.splits * 8 * Co * Ci * sizeof(float);
    if (ws_bytes < need || !ws) return pcrl_fail(PCRL_EWORKSPACE, "convt3d_k2s2_wgrad: workspace %zu < %zu", ws_bytes, need);
    static std::once_flag attr_once;   // hipFuncSetAttribute once per process, race-free
    constexpr int LDS = 2 * 4096 + 2 * 8 * 4096;
    std::call_once(attr_once, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_up2_alltaps_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
    WgradParams p{x, dy, (float*)ws, Dims{N, D, H, W}, M, Ci, Co, 8, sp.chunk};
    hipLaunchKernelGGL(wgrad_up2_alltaps_kernel, dim3((unsigned)((Ci / 64) * (Co / 64)), (unsigned)sp.splits), dim3(256), LDS, as_stream(stream), p);
    if (int e = pcrl_check_launch("convt3d_k2s2_wgrad")) return e;
    return launch_wgrad_reduce((const float*)ws, dw_ref, sp.splits, 8, Ci, Co, Co, as_stream(stream));
  }
  return run_wgrad<WG_UP2>(x, dy, dw_ref, ws, ws_bytes, Dims{N, D, H, W}, Ci, Co, 8, dtype, as_stream(stream));
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradients of the 1-channel convolutions as plain MFMA GEMMs over voxels, via an im2col of the SCALAR operand
// (32 columns: 27 taps + padding), which costs one extra HBM pass over a 32-channel tensor:
//   first layer  (1 -> Co): dw[c][t] = sum_m dy[m][c] * x[m+delta_t]      = (dy^T . im2col(x))[c][t]
//   heads        (C -> 1) : dw[c][t] = sum_m x[m][c]  * dy[m-delta_t]     = (x^T  . im2col_flipped(dy))[c][t];  db = sum dy
// Workspace layout: [ im2col: M*32 elements of dtype ][ 8 KiB of fp64 partials ][ split-K partials ].
// ---------------------------------------------------------------------------------------------------------------------
namespace {
size_t plain_ws_bytes(int64_t M, int Cu, int dtype_size) {
  const SplitPlan sp = plan_splits(M, Cu, 32, 1);
  return (size_t)M * 32 * dtype_size + 8192 + (size_t)sp.splits * Cu * 32 * sizeof(float);
}
int im2col_launch(const float* s, void* out, Dims g, int64_t M, int taps, int flip, int dtype, hipStream_t stream) {
  int64_t blocks = (M * (dtype == PCRL_BF16 ? 4 : 8) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (dtype == PCRL_BF16) hipLaunchKernelGGL(im2col27_kernel<bf16>, dim3((unsigned)blocks), dim3(256), 0, stream, s, (bf16*)out, g, M, taps, flip);
  else hipLaunchKernelGGL(im2col27_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, s, (float*)out, g, M, taps, flip);
  return pcrl_check_launch("im2col27");
}
}  // namespace

extern "C" size_t pcrl_conv3d_k3_c1_wgrad_ws_bytes(int N, int D, int H, int W, int Co) {
  size_t need = plain_ws_bytes((int64_t)N * D * H * W, Co, 4);
  if (D % 4 == 0 && H % 8 == 0 && W % 8 == 0 && scalar_brick_ws_bytes(N, D, H, W, Co) > need) need = scalar_brick_ws_bytes(N, D, H, W, Co);
  return need;
}

extern "C" int pcrl_conv3d_k3_c1_wgrad(const float* x, const void* dy, float* dw_ref, void* ws, size_t ws_bytes,
                                       int N, int D, int H, int W, int Co, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && dy && dw_ref, "conv3d_k3_c1_wgrad: null pointer");
  PCRL_REQUIRE(dtype == PCRL_BF16 || dtype == PCRL_F32, "conv3d_k3_c1_wgrad: bad dtype %d", dtype);
  PCRL_REQUIRE(Co > 0 && Co % 8 == 0, "conv3d_k3_c1_wgrad: Co must be a multiple of 8 (got %d)", Co);
  const int64_t M = (int64_t)N * D * H * W;
  const int esz = dtype == PCRL_BF16 ? 2 : 4;
  if (g_wgrad_impl == 0 && scalar_brick_ok(D, H, W, Co, 27, dtype)) {
    if (!ws || ws_bytes < scalar_brick_ws_bytes(N, D, H, W, Co)) return pcrl_fail(PCRL_EWORKSPACE, "conv3d_k3_c1_wgrad: workspace too small");
    return scalar_brick_launch(dy, x, dw_ref, (char*)ws, N, D, H, W, Co, 0, as_stream(stream));
  }
  if (!ws || ws_bytes < plain_ws_bytes(M, Co, esz)) return pcrl_fail(PCRL_EWORKSPACE, "conv3d_k3_c1_wgrad: workspace too small");
  const Dims g{N, D, H, W};
  char* col = (char*)ws;
  char* part = col + (size_t)M * 32 * esz + 8192;
  if (int e = im2col_launch(x, col, g, M, 27, 0, dtype, as_stream(stream))) return e;
  return run_wgrad<WG_PLAIN>(dy, col, dw_ref, part, ws_bytes - (size_t)(part - col), g, Co, 32, 1, dtype, as_stream(stream), 27);
}

// weighted column sum (norm_pool.hip): the whole weight gradient of a 1x1x1 convolution to one channel
size_t pcrl_weighted_colsum_ws_bytes(int64_t M, int C);
int pcrl_weighted_colsum(const void* v, const float* rowscale, float* out, void* ws, size_t ws_bytes, int64_t M, int C, int dtype, hipStream_t stream);
static bool to1_pointwise_ok(int C, int taps, int dtype) { return taps == 1 && C % (dtype == PCRL_BF16 ? 8 : 4) == 0 && C <= 512 && g_wgrad_impl == 0; }

extern "C" size_t pcrl_conv3d_to1_wgrad_ws_bytes(int N, int D, int H, int W, int C, int taps) {
  size_t need = plain_ws_bytes((int64_t)N * D * H * W, C, 4);
  if (taps == 1 && 8192 + pcrl_weighted_colsum_ws_bytes((int64_t)N * D * H * W, C) > need) need = 8192 + pcrl_weighted_colsum_ws_bytes((int64_t)N * D * H * W, C);
  if (taps == 27 && D % 4 == 0 && H % 8 == 0 && W % 8 == 0 && scalar_brick_ws_bytes(N, D, H, W, C) > need) need = scalar_brick_ws_bytes(N, D, H, W, C);
  return need;
}

extern "C" int pcrl_conv3d_to1_wgrad(const void* x, const float* dy, float* dw_ref, float* db, void* ws, size_t ws_bytes,
                                     int N, int D, int H, int W, int C, int taps, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && dy && dw_ref && db, "conv3d_to1_wgrad: null pointer");
  PCRL_REQUIRE(dtype == PCRL_BF16 || dtype == PCRL_F32, "conv3d_to1_wgrad: bad dtype %d", dtype);
  PCRL_REQUIRE(taps == 27 || taps == 1, "conv3d_to1_wgrad: taps must be 27 or 1 (got %d)", taps);
  PCRL_REQUIRE(C > 0 && C % 8 == 0, "conv3d_to1_wgrad: C must be a multiple of 8 (got %d)", C);
  const int64_t M = (int64_t)N * D * H * W;
  const int esz = dtype == PCRL_BF16 ? 2 : 4;
  const Dims g{N, D, H, W};
  char* col = (char*)ws;
  double* red;
  if (g_wgrad_impl == 0 && scalar_brick_ok(D, H, W, C, taps, dtype)) {
    if (!ws || ws_bytes < scalar_brick_ws_bytes(N, D, H, W, C)) return pcrl_fail(PCRL_EWORKSPACE, "conv3d_to1_wgrad: workspace too small");
    red = (double*)col;   // first 8 KiB of the workspace: partials of the bias gradient
    if (int e = scalar_brick_launch(x, dy, dw_ref, col, N, D, H, W, C, 1, as_stream(stream))) return e;
  } else if (to1_pointwise_ok(C, taps, dtype)) {
    // 1x1x1: dw[c] = sum_m x[m][c] dy[m] -- one weighted column-sum pass over x (the im2col + split-K GEMM + reduce it replaces took
    // 0.44 ms per step for 64 numbers: 2048 partial slabs reduced by four blocks)
    if (!ws || ws_bytes < 8192 + pcrl_weighted_colsum_ws_bytes(M, C)) return pcrl_fail(PCRL_EWORKSPACE, "conv3d_to1_wgrad: workspace too small");
    red = (double*)col;
    if (int e = pcrl_weighted_colsum(x, dy, dw_ref, col + 8192, ws_bytes - 8192, M, C, dtype, as_stream(stream))) return e;
  } else {
    if (!ws || ws_bytes < plain_ws_bytes(M, C, esz)) return pcrl_fail(PCRL_EWORKSPACE, "conv3d_to1_wgrad: workspace too small");
    red = (double*)(col + (size_t)M * 32 * esz);
    char* part = (char*)red + 8192;
    if (int e = im2col_launch(dy, col, g, M, taps, 1, dtype, as_stream(stream))) return e;
    if (int e = run_wgrad<WG_PLAIN>(x, col, dw_ref, part, ws_bytes - (size_t)(part - col), g, C, 32, 1, dtype, as_stream(stream), taps)) return e;
  }
  int blocks = (int)((M + 4095) / 4096);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(vecsum_partial_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dy, red, M);
  if (int e = pcrl_check_launch("vecsum_partial")) return e;
  hipLaunchKernelGGL(vecsum_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), (const double*)red, db, blocks);
  return pcrl_check_launch("vecsum_finish");
}

// ---- 2D path: weight gradient of a KHxKW / stride / pad convolution, optionally behind a fused nearest x2 upsample ----
// x: [N][Hi][Wi][CiP], dy: [N][Ho][Wo][CoP] (channel counts multiples of 8 (bf16) / 4 (fp32); zero padded by the caller),
// dw: float32 [CoP][Ci_out][KH][KW] (reference layout; padded source channels >= Ci_out dropped).
static SplitPlan plan_splits2d(int64_t M, int Cu, int Cv, int ks) {
  const int64_t tiles = (int64_t)((Cu + 63) / 64) * ((Cv + 63) / 64);
  const int64_t steps = (M + ks - 1) / ks;
  int64_t splits = 2048 / tiles;
  const int64_t max_splits = (steps + 3) / 4;   // at least 4 K-steps per block
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int64_t per = (steps + splits - 1) / splits;
  splits = (steps + per - 1) / per;
  return SplitPlan{(int)splits, per * ks};
}
// one-kernel-row-per-block form (wgrad2d_row3_kernel): K-steps of 64 pixels, whole rounds of ~1024 blocks, at least 8 steps per block.
static bool wgrad2d_row3_ok(int CiP, int CoP, int KH, int KW, int dtype) {
  return dtype == PCRL_BF16 && KH == 3 && KW == 3 && CiP >= 32 && CoP >= 32;
}
static SplitPlan plan_row3(int64_t M, int CoP, int CiP) {
  const int64_t tiles = (int64_t)((CoP + 63) / 64) * ((CiP + 63) / 64) * 3;
  const int64_t steps = (M + 63) / 64;
  int64_t splits = 1024 / tiles;
  const int64_t max_splits = (steps + 7) / 8;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int64_t per = (steps + splits - 1) / splits;
  splits = (steps + per - 1) / per;
  return SplitPlan{(int)splits, per * 64};
}
static int conv2d_wgrad_ks(int dtype) { return (dtype == PCRL_BF16 && g_wgrad_tr) ? 128 : 32; }   // 128 (template KS) measured slower: the loads, not the barriers, set the pace

// LDS-halo brick kernel with the image index as depth (wgrad_brick.hip, nkd = 1)
bool pcrl_wgrad_brick2d_eligible(int N, int H, int W, int Ci, int Co, int dtype);
int pcrl_wgrad_brick2d_splits(int N, int H, int W, int Ci, int Co);
int pcrl_wgrad_brick2d_launch(const void* x, const void* dy, float* ws, int N, int H, int W, int Ci, int Co, int up, hipStream_t stream);

// right-sized kernel for the 16/32-channel layers (wgrad2d_narrow.hip)
bool pcrl_wgrad2d_narrow_eligible(int N, int H, int W, int CiP, int CoP, int dtype);
int pcrl_wgrad2d_narrow_slabs(int N, int H, int W);
int pcrl_wgrad2d_narrow_launch(const void* x, const void* dy, float* ws, int N, int H, int W, int CiP, int CoP, int up, hipStream_t stream);

extern "C" size_t pcrl_conv2d_wgrad_ws_bytes(int N, int Ho, int Wo, int CiP, int CoP, int KH, int KW) {
  // the largest of the variants (the dtype is not known here)
  const SplitPlan a = plan_splits2d((int64_t)N * Ho * Wo, CoP, KH * KW * CiP, 32), b = plan_splits2d((int64_t)N * Ho * Wo, CoP, KH * KW * CiP, 128);
  int splits = a.splits > b.splits ? a.splits : b.splits;
  if (KH == 3 && KW == 3 && pcrl_wgrad_brick2d_eligible(N, Ho, Wo, CiP, CoP, PCRL_BF16)) {
    const int c = pcrl_wgrad_brick2d_splits(N, Ho, Wo, CiP, CoP);
    if (c > splits) splits = c;
  }
  if (KH == 3 && KW == 3 && pcrl_wgrad2d_narrow_eligible(N, Ho, Wo, CiP, CoP, PCRL_BF16)) {
    const int c = pcrl_wgrad2d_narrow_slabs(N, Ho, Wo);
    if (c > splits) splits = c;
  }
  if (wgrad2d_row3_ok(CiP, CoP, KH, KW, PCRL_BF16)) {
    const int c = plan_row3((int64_t)N * Ho * Wo, CoP, CiP).splits;
    if (c > splits) splits = c;
  }
  return (size_t)splits * KH * KW * CoP * CiP * sizeof(float);
}

extern "C" int pcrl_conv2d_wgrad(const void* x, const void* dy, float* dw_ref, void* ws, size_t ws_bytes, int N, int Hi, int Wi, int CiP, int Ci_out,
                                 int Ho, int Wo, int CoP, int KH, int KW, int stride, int pad, int up, int dtype, pcrl_stream_t stream) {
  PCRL_REQUIRE(x && dy && dw_ref, "conv2d_wgrad: null pointer");
  const int vec = dtype == PCRL_BF16 ? 8 : 4;
  PCRL_REQUIRE(CiP > 0 && CoP > 0 && CiP % vec == 0 && CoP % vec == 0 && Ci_out > 0 && Ci_out <= CiP,
               "conv2d_wgrad: channel counts must be multiples of %d (CiP=%d CoP=%d)", vec, CiP, CoP);
  PCRL_REQUIRE(KH >= 1 && KW >= 1 && KH * KW <= 49 && stride >= 1 && pad >= 0, "conv2d_wgrad: bad kernel geometry");
  PCRL_REQUIRE(dtype == PCRL_BF16 || dtype == PCRL_F32, "conv2d_wgrad: bad dtype %d", dtype);
  const int taps = KH * KW, Cv = taps * CiP;
  if (g_wgrad_impl == 0 && g_wgrad_tr && KH == 3 && KW == 3 && stride == 1 && pad == 1 && (up ? (2 * Hi == Ho && 2 * Wi == Wo) : (Hi == Ho && Wi == Wo)) &&
      pcrl_wgrad2d_narrow_eligible(N, Ho, Wo, CiP, CoP, dtype)) {
    const int slabs = pcrl_wgrad2d_narrow_slabs(N, Ho, Wo);
    const size_t need = (size_t)slabs * 9 * CoP * CiP * sizeof(float);
    if (!ws || ws_bytes < need) return pcrl_fail(PCRL_EWORKSPACE, "conv2d_wgrad: workspace %zu < %zu", ws_bytes, need);
    if (int e = pcrl_wgrad2d_narrow_launch(x, dy, (float*)ws, N, Ho, Wo, CiP, CoP, up, as_stream(stream))) return e;
    return launch_wgrad2d_reduce((const float*)ws, dw_ref, slabs, 9, CoP, CiP, Ci_out, as_stream(stream));
  }
  if (g_wgrad_impl == 0 && g_wgrad_tr && KH == 3 && KW == 3 && stride == 1 && pad == 1 && (up ? (2 * Hi == Ho && 2 * Wi == Wo) : (Hi == Ho && Wi == Wo)) &&
      pcrl_wgrad_brick2d_eligible(N, Ho, Wo, CiP, CoP, dtype)) {
    const int splits = pcrl_wgrad_brick2d_splits(N, Ho, Wo, CiP, CoP);
    const size_t need = (size_t)splits * 9 * CoP * CiP * sizeof(float);
    if (!ws || ws_bytes < need) return pcrl_fail(PCRL_EWORKSPACE, "conv2d_wgrad: workspace %zu < %zu", ws_bytes, need);
    if (int e = pcrl_wgrad_brick2d_launch(x, dy, (float*)ws, N, Ho, Wo, CiP, CoP, up, as_stream(stream))) return e;
    return launch_wgrad_reduce((const float*)ws, dw_ref, splits, 9, CoP, CiP, Ci_out, as_stream(stream));
  }
  const Dims g{N, 1, Ho, Wo};
  const int64_t M = (int64_t)N * Ho * Wo;
  if (g_wgrad_impl == 0 && g_wgrad_tr && wgrad2d_row3_ok(CiP, CoP, KH, KW, dtype)) {
    const SplitPlan sp = plan_row3(M, CoP, CiP);
    const size_t need = (size_t)sp.splits * CoP * Cv * sizeof(float);
    if (ws_bytes < need || !ws) return pcrl_fail(PCRL_EWORKSPACE, "conv2d_wgrad: workspace %zu < %zu", ws_bytes, need);
    WgradParams p{dy, x, (float*)ws, g, M, CoP, Cv, 1, sp.chunk, Wg2d{Hi, Wi, KW, stride, pad, up, CiP, taps}};
    dim3 grid((unsigned)(((CoP + 63) / 64) * ((CiP + 63) / 64)), 3u, (unsigned)sp.splits);
    static std::once_flag attr_once;
    std::call_once(attr_once, [] { hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad2d_row3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 64 * 128); });
    hipLaunchKernelGGL((wgrad2d_row3_kernel<true>), grid, dim3(256), 8 * 64 * 128, as_stream(stream), p);
    if (int e = pcrl_check_launch("conv2d_wgrad (row3)")) return e;
    return launch_wgrad2d_reduce((const float*)ws, dw_ref, sp.splits, taps, CoP, CiP, Ci_out, as_stream(stream));
  }
  const int ks = conv2d_wgrad_ks(dtype);
  const SplitPlan sp = plan_splits2d(M, CoP, Cv, ks);
  const size_t need = (size_t)sp.splits * CoP * Cv * sizeof(float);
  if (ws_bytes < need || !ws) return pcrl_fail(PCRL_EWORKSPACE, "conv2d_wgrad: workspace %zu < %zu", ws_bytes, need);
  WgradParams p{dy, x, (float*)ws, g, M, CoP, Cv, 1, sp.chunk, Wg2d{Hi, Wi, KW, stride, pad, up, CiP, taps}};
  dim3 grid((unsigned)(((CoP + 63) / 64) * ((Cv + 63) / 64)), 1u, (unsigned)sp.splits);
  hipStream_t st = as_stream(stream);
  if (dtype == PCRL_BF16) {
    if (g_wgrad_tr) hipLaunchKernelGGL((wgrad_kernel<bf16, WG_CONV2D, true, 128>), grid, dim3(256), 4 * 128 * 128, st, p);
    else hipLaunchKernelGGL((wgrad_kernel<bf16, WG_CONV2D, false>), grid, dim3(256), 4 * 32 * 128, st, p);
  } else {
    hipLaunchKernelGGL((wgrad_kernel<float, WG_CONV2D, false>), grid, dim3(256), 4 * 32 * 256, st, p);
  }
  if (int e = pcrl_check_launch("conv2d_wgrad")) return e;
  return launch_wgrad2d_reduce((const float*)ws, dw_ref, sp.splits, taps, CoP, CiP, Ci_out, st);
}
