// Implicit-GEMM convolution on MFMA for gfx950: 3x3x3 conv forward / data-gradient and the
// 2x2x2-stride-2 transposed conv (forward scatter GEMMs, data-gradient gather GEMM).
//
// Replaces aten::convolution / convolution_backward(input) dispatched from
// models/pcrlv2_model_3d.py:9,33 (LUConv.conv1) and :52,64 (UpTransition.up_conv).
//
// GEMM view: rows = voxels (M = N*D*H*W), cols = output channels, K = taps * channels.
// Activations are NDHWC so one (voxel, tap) contributes a CONTIGUOUS channel vector: the A tile
// of a K-step (one tap, 32 channels) is 128 rows x 64 B (bf16), fetched with 16-byte loads whose
// address is `row + tap_delta` -- the 27 shifted re-reads of a voxel hit L1/L2, HBM sees each
// activation once.  Weights are pre-packed K-contiguous ([co][tap][ci]) so both MFMA operands are
// plain 16-byte LDS reads.  Tile 128 x BN (BN = 32/64/128), 4 waves as 2(M) x 2(N), each wave
// 64 x BN/2 = 4 x FN fragments of v_mfma_f32_16x16x32_bf16 (bf16) or 8 x v_mfma_f32_16x16x4_f32
// (exact fp32 parity mode).  LDS tiles are double-buffered, register-staged (the halo needs
// zero-fill, which an LDS-DMA cannot do) and XOR-swizzled in 16-byte slots so a 16-lane fragment
// read covers all 64 banks.
//
// Epilogue: + bias, store, and per-tile per-channel (sum, sum of squares) from the fp32
// accumulators for the training-mode BatchNorm that follows every conv (no atomics: one partial
// row per tile, reduced in fixed order by bn_finalize).
#include "common.h"
#include "mma_tile.h"
#include <atomic>
#include <cstdlib>

namespace {

enum { GEOM_CONV3 = 0, GEOM_UP2_FWD = 1, GEOM_UP2_DGRAD = 2, GEOM_UPC_FWD = 3, GEOM_UPC_DGRAD = 4 };
// GEOM_UPC_*: ConvTranspose3d(k2,s2) followed directly by Conv3d(3x3x3, pad 1) -- UpTransition's up_conv and ops.0's conv1
// (pcrlv2_model_3d.py:52,64 then :9,33), nothing in between -- as ONE linear operator on the COARSE grid (upconv_fused.hip composes
// the weights).  A fine output voxel f = 2v + p (phase p in {0,1}^3) sees, through its 3x3x3 window on the fine grid, only the
// 2x2x2 coarse voxels v + p - 1 + q (q in {0,1}^3): 8 taps instead of 27 and no 2x-upsampled intermediate tensor.
//   UPC_FWD  : rows = coarse voxels, blockIdx.z = phase, 8 taps of K = Ci channels, output row = the phase's fine voxel; the bias is a
//              table over the 27 border classes of the fine voxel (the inner convolution zero-pads the UPSAMPLED tensor, whose bias
//              therefore enters through a position-dependent number of taps); statistics rows = 8 x tiles.
//   UPC_DGRAD: rows = coarse voxels, 64 taps = the 4x4x4 fine voxels 2u - 1 .. 2u + 2 of K = Co channels (each belongs to one (p, q)).

struct IgemmParams {
  const void* x;      // A source rows [*][K]
  const void* w;      // packed weights [z][Nc][taps][K]
  const float* bias;  // [Nc] or null
  void* y;            // output rows [*][Nc]
  float* stats;       // [gridDim.x][Nc][2] or null
  Dims g;             // index space of the GEMM rows
  int64_t M;
  int K;              // channels per tap
  int Nc;             // output channels
  int taps;           // taps accumulated inside one GEMM (27, 1 or 8)
  float* ws;          // split-K (GEOM_CONV3 and GEOM_UPC_DGRAD): [gridDim.z][M][Nc] float partial sums, or null
  int steps_per_split;
  // GEOM_UPC_* only.  1-D grid over (row tile, phase): the phases of a tile run back to back on ONE XCD and every XCD owns a contiguous
  // range of tiles (the eight phases gather the same x rows, neighbouring tiles share faces: one L2 serves them; with the phase as the
  // slow grid index every phase re-read x from HBM -- 1.6 GB per launch at up_tr64, rocprofv3 r02e).  bd * bh * bw == 128 (0: linear
  // order): the 128 rows of a tile are a compact bd x bh x bw box of voxels instead of 128 consecutive ones, so the 8 / 64 gathered
  // neighbours of a tile's rows are mostly each other's.
  int nt, zdim, bd, bh, bw;
  // GEOM_CONV3 only.  vmajor = 1: GEMM row m is (voxel v = m / N, sample n = m % N) instead of (n, v) -- a 128-row tile then holds ONE or two voxels
  // of many samples, every row of it has (nearly) the same set of taps inside the volume, and the K loop walks only the taps some row of the
  // tile uses: 8-12 of 27 on a 2^3 grid, 15.6 on average on 4^3 (the local views' deepest levels).  Skipped taps contributed
  // exact zeros, so every output value is the one the full loop produces (split-K aside: the splits cut the shorter loop).
  int vmajor;
};


// PLANES: epilogue variant for the 1-output-channel convolutions (conv_c1.hip): the GEMM columns are the 27 taps of a
// pointwise product z[t][m] = sum_c x[m][c] w[c][t]; the tile is written as float32, PLANE-major (z[col * M + m]), so the
// shifted-sum pass that follows reads every plane contiguously.  No bias, no statistics.
template <typename T, int BN, int GEOM, bool PLANES = false>
__global__ void __launch_bounds__(256, PCRL_OCC2) igemm_kernel(const IgemmParams p) {
  constexpr int BM = PCRL_CONV_BM;
  using TL = Tile<T>;
  using MM = Mma<T>;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int SLOTS = TL::SLOTS;
  constexpr int RPP = 256 / SLOTS;  // tile rows staged per pass of the 256 threads
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
  constexpr bool UPC = GEOM == GEOM_UPC_FWD || GEOM == GEOM_UPC_DGRAD;
  int tile_ = blockIdx.x, z_ = (GEOM == GEOM_CONV3) ? 0 : blockIdx.z;   // convT forward: tap; conv3: blockIdx.z = K split
  if (UPC) {
    const int id = blockIdx.x;
    if ((p.nt & 7) == 0) {
      const int k = id >> 3;
      z_ = k % p.zdim;
      tile_ = (id & 7) * (p.nt >> 3) + k / p.zdim;
    } else {
      z_ = id % p.zdim;
      tile_ = id / p.zdim;
    }
  }
  const int z = z_;
  const int64_t m0 = (int64_t)tile_ * BM;
  const int n0 = blockIdx.y * BN;
  const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ Wp = reinterpret_cast<const T*>(p.w);
  const Dims g = p.g;
  const int K = p.K;
  // row index -> voxel: linear order, or (UPC, bd > 0) tile-major boxes
  auto row_voxel = [&](int64_t m, int& n, int& d, int& h, int& w) {
    if (UPC && p.bd > 0) {
      const int r = (int)(m & (BM - 1));
      int t = (int)(m / BM);
      const int nbw = g.W / p.bw, nbh = g.H / p.bh, nbd = g.D / p.bd;
      const int bx = t % nbw; t /= nbw;
      const int by = t % nbh; t /= nbh;
      const int bz = t % nbd;
      n = t / nbd;
      w = bx * p.bw + r % p.bw;
      h = by * p.bh + (r / p.bw) % p.bh;
      d = bz * p.bd + r / (p.bw * p.bh);
    } else {
      decode_voxel(m, g, n, d, h, w);
    }
  };

  const int slot = tid % SLOTS, rowp = tid / SLOTS;

  // ---- per-thread A rows: base row in the source and tap validity ----
  int64_t abase[AP];
  uint32_t amask[AP];
#pragma unroll
  for (int ps = 0; ps < AP; ++ps) {
    const int64_t m = m0 + ps * RPP + rowp;
    abase[ps] = 0;
    amask[ps] = 0;
    if (m < p.M) {
      int n, d, h, w;
      row_voxel(m, n, d, h, w);
      if (GEOM == GEOM_CONV3) {
        abase[ps] = m;
        if (p.vmajor) {
          const int V = g.D * g.H * g.W, v = (int)(m / g.N);
          n = (int)(m - (int64_t)v * g.N);
          w = v % g.W;
          h = (v / g.W) % g.H;
          d = v / (g.W * g.H);
          abase[ps] = (int64_t)n * V + v;
        }
        amask[ps] = tap_mask27(d, h, w, g);
      } else if (GEOM == GEOM_UP2_FWD) {
        abase[ps] = m;
        amask[ps] = 1u;
      } else if (GEOM == GEOM_UPC_FWD) {
        abase[ps] = (((int64_t)n * g.D + d) * g.H + h) * g.W + w;
        const int od = (z >> 2) - 1, oh = ((z >> 1) & 1) - 1, ow = (z & 1) - 1;
        uint32_t mk = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if ((unsigned)(d + od + (t >> 2)) < (unsigned)g.D && (unsigned)(h + oh + ((t >> 1) & 1)) < (unsigned)g.H &&
              (unsigned)(w + ow + (t & 1)) < (unsigned)g.W)
            mk |= 1u << t;
        amask[ps] = mk;
      } else if (GEOM == GEOM_UPC_DGRAD) {
        abase[ps] = up2_row(n, d, h, w, 0, g);
        amask[ps] = (d == 0 ? 1u : 0u) | (d == g.D - 1 ? 2u : 0u) | (h == 0 ? 4u : 0u) | (h == g.H - 1 ? 8u : 0u) | (w == 0 ? 16u : 0u) |
                    (w == g.W - 1 ? 32u : 0u);   // border flags; a tap is dead when it needs a flagged side
      } else {
        abase[ps] = up2_row(n, d, h, w, 0, g);
        amask[ps] = 0xFFu;
      }
    } else if (GEOM == GEOM_UPC_DGRAD) {
      amask[ps] = 64u;   // row past M: every tap dead
    }
  }
  // ---- per-thread B rows ----
  int64_t boff[BP];
  bool bok[BP];
#pragma unroll
  for (int ps = 0; ps < BP; ++ps) {
    const int brow = ps * RPP + rowp;
    bok[ps] = brow < BN;
    boff[ps] = ((int64_t)(z * p.Nc + n0 + (bok[ps] ? brow : 0)) * p.taps) * K + slot * VEC;
  }

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunk = K / 32;
  constexpr bool CAN_SPLIT = GEOM == GEOM_CONV3 || GEOM == GEOM_UPC_DGRAD;
  const bool vmaj = GEOM == GEOM_CONV3 && p.vmajor != 0;
  // taps this tile walks (block-uniform): all of them, or (vmajor) the union of the tap masks of the tile's voxels
  uint64_t tmask = p.taps >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << p.taps) - 1);
  if (vmaj) {
    const int64_t mlast = (m0 + BM - 1 < p.M ? m0 + BM - 1 : p.M - 1);
    const int v0 = (int)(m0 / g.N), v1 = (int)(mlast / g.N);
    uint32_t u = 0;
    for (int v = v0; v <= v1; ++v) u |= tap_mask27(v / (g.W * g.H), (v / g.W) % g.H, v % g.W, g);
    tmask = u;
  }
  const int total_steps = __popcll(tmask) * nchunk;
  int s_off = 0, S = total_steps;
  if (CAN_SPLIT && p.ws) {   // first K-step and step count of this split
    const int per = vmaj ? (total_steps + (int)gridDim.z - 1) / (int)gridDim.z : p.steps_per_split;
    s_off = blockIdx.z * per;
    S = min(per, total_steps - s_off);
    if (S < 0) S = 0;
  }
  // Two staging register sets: the loads of K-step s+2 are issued while step s is multiplied and step s+1 waits in the
  // other set, so a load has two MFMA phases to land.
  u32x4 raA[AP], rbA[BP], raB[AP], rbB[BP];
  uint32_t aokA = 0, aokB = 0;
  // NOTE: global loads are UNCONDITIONAL (invalid taps re-read the row's own, always valid, base row and are zeroed at
  // the LDS store).  A load under a per-lane condition makes hipcc wrap it in a branch with `s_waitcnt vmcnt(0)`, which
  // serialises every load of the K-step ahead of the MFMAs (measured: 185 TF -> 655 TF, see profiles/).

#define IGEMM_LOAD(t_, c_, ra, rb, aok)                                                                 \
  do {                                                                                                  \
    int64_t delta_;                                                                                     \
    uint32_t kill_ = 0;                                                                                 \
    if (GEOM == GEOM_CONV3) delta_ = tap_delta27((t_), g);                                              \
    else if (GEOM == GEOM_UP2_DGRAD)                                                                    \
      delta_ = ((int64_t)((t_) >> 2) * (2 * g.H) + (((t_) >> 1) & 1)) * (2 * g.W) + ((t_)&1);           \
    else if (GEOM == GEOM_UPC_FWD)                                                                      \
      delta_ = ((int64_t)((z >> 2) - 1 + ((t_) >> 2)) * g.H + (((z >> 1) & 1) - 1 + (((t_) >> 1) & 1))) * g.W + ((z & 1) - 1 + ((t_)&1)); \
    else if (GEOM == GEOM_UPC_DGRAD) {                                                                  \
      const int ed_ = (t_) >> 4, eh_ = ((t_) >> 2) & 3, ew_ = (t_)&3;                                   \
      delta_ = ((int64_t)(ed_ - 1) * (2 * g.H) + (eh_ - 1)) * (2 * g.W) + (ew_ - 1);                    \
      kill_ = 64u | (ed_ == 0 ? 1u : 0u) | (ed_ == 3 ? 2u : 0u) | (eh_ == 0 ? 4u : 0u) | (eh_ == 3 ? 8u : 0u) | (ew_ == 0 ? 16u : 0u) | \
              (ew_ == 3 ? 32u : 0u);                                                                    \
    } else delta_ = 0;                                                                                  \
    aok = 0;                                                                                            \
    _Pragma("unroll") for (int ps = 0; ps < AP; ++ps) {                                                 \
      const uint32_t ok_ = (GEOM == GEOM_UPC_DGRAD) ? ((amask[ps] & kill_) == 0u ? 1u : 0u) : ((amask[ps] >> (t_)) & 1u); \
      const int64_t row_ = abase[ps] + (ok_ ? delta_ : (int64_t)0);                                     \
      ra[ps] = *reinterpret_cast<const u32x4*>(X + row_ * K + (c_)*32 + slot * VEC);                    \
      aok |= ok_ << ps;                                                                                 \
    }                                                                                                   \
    _Pragma("unroll") for (int ps = 0; ps < BP; ++ps)                                                   \
      rb[ps] = *reinterpret_cast<const u32x4*>(Wp + boff[ps] + (int64_t)(t_)*K + (c_)*32);              \
  } while (0)

#define IGEMM_STORE(buf_, ra, rb, aok)                                                                  \
  do {                                                                                                  \
    _Pragma("unroll") for (int ps = 0; ps < AP; ++ps)                                                   \
      *reinterpret_cast<u32x4*>(As + (buf_)*A_BYTES + TL::off(ps * RPP + rowp, slot)) =                 \
          keep_if((aok >> ps) & 1u, ra[ps]);                                                            \
    _Pragma("unroll") for (int ps = 0; ps < BP; ++ps) {                                                 \
      if (bok[ps]) *reinterpret_cast<u32x4*>(Bs + (buf_)*B_BYTES + TL::off(ps * RPP + rowp, slot)) = rb[ps]; \
    }                                                                                                   \
  } while (0)

#define IGEMM_COMPUTE(cur_)                                                                             \
  do {                                                                                                  \
    const char* a = As + (cur_)*A_BYTES;                                                                \
    const char* b = Bs + (cur_)*B_BYTES;                                                                \
    typename MM::Frag fa[FM], fb[FN];                                                                   \
    _Pragma("unroll") for (int i = 0; i < FM; ++i) fa[i] = MM::read(a, wm * 64 + i * 16 + lr, lg);      \
    _Pragma("unroll") for (int j = 0; j < FN; ++j) fb[j] = MM::read(b, wn * (BN / 2) + j * 16 + lr, lg); \
    _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                      \
      _Pragma("unroll") for (int j = 0; j < FN; ++j) MM::mma(fa[i], fb[j], acc[i][j]);                  \
  } while (0)

  // K-steps in order: STEP_TC is called for s_ = 0, 1, 2, ... exactly once each and hands out (tap, chunk) of step s_off + min(s_, S - 1) (the
  // tail re-loads the last step; its data is never used) from a running position: the lowest set bit of `tbits` is the current tap.
  uint64_t tbits = tmask;
  int it_c, it_n = 0;
  {
    const int so = s_off < total_steps ? s_off : 0;   // an empty split (S == 0) loads the first step and multiplies nothing
    const int skip = so / nchunk;
    it_c = so - skip * nchunk;
    for (int q = 0; q < skip; ++q) tbits &= tbits - 1;
  }
#define STEP_TC(s_, t_, c_)                                                                             \
  do {                                                                                                  \
    t_ = __builtin_ctzll(tbits);                                                                        \
    c_ = it_c;                                                                                          \
    if (++it_n < S) {                                                                                   \
      if (++it_c == nchunk) {                                                                           \
        it_c = 0;                                                                                       \
        tbits &= tbits - 1;                                                                             \
      }                                                                                                 \
    }                                                                                                   \
  } while (0)

  int t0_, c0_;
  STEP_TC(0, t0_, c0_);
  IGEMM_LOAD(t0_, c0_, raA, rbA, aokA);
  IGEMM_STORE(0, raA, rbA, aokA);
  STEP_TC(1, t0_, c0_);
  IGEMM_LOAD(t0_, c0_, raA, rbA, aokA);      // set A holds step 1
  __syncthreads();

  // Straight-line loop body, unrolled by two so that the register sets alternate statically.
  for (int s = 0; s < S; s += 2) {
    int tn, cn;
    // ---- even step s: compute buffer 0; set A (step s+1) -> buffer 1; set B <- step s+2
    STEP_TC(s + 2, tn, cn);
    IGEMM_LOAD(tn, cn, raB, rbB, aokB);
    __builtin_amdgcn_sched_barrier(0);
    IGEMM_COMPUTE(0);
    __builtin_amdgcn_sched_barrier(0);
    IGEMM_STORE(1, raA, rbA, aokA);
    __syncthreads();
    if (s + 1 >= S) break;
    // ---- odd step s+1: compute buffer 1; set B (step s+2) -> buffer 0; set A <- step s+3
    STEP_TC(s + 3, tn, cn);
    IGEMM_LOAD(tn, cn, raA, rbA, aokA);
    __builtin_amdgcn_sched_barrier(0);
    IGEMM_COMPUTE(1);
    __builtin_amdgcn_sched_barrier(0);
    IGEMM_STORE(0, raB, rbB, aokB);
    __syncthreads();
  }
#undef IGEMM_COMPUTE
#undef STEP_TC
#undef IGEMM_LOAD
#undef IGEMM_STORE

  if (PLANES) {
    float* __restrict__ Z = reinterpret_cast<float*>(p.y);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int64_t m = m0 + wm * 64 + i * 16 + lg * 4;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 16 + lr;
        float* dst = Z + (int64_t)col * p.M + m;
        if (m + 3 < p.M) {
          *reinterpret_cast<f32x4*>(dst) = acc[i][j];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (m + r < p.M) dst[r] = acc[i][j][r];
        }
      }
    }
    return;
  }
  if (CAN_SPLIT && p.ws) {   // split-K: raw float partial sums; bias, rounding and statistics happen in the finish pass
    float* __restrict__ Z = p.ws + (int64_t)blockIdx.z * p.M * p.Nc;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + wm * 64 + i * 16 + lg * 4 + r;
        if (m < p.M) {
          int64_t orow = m;
          if (GEOM == GEOM_UPC_DGRAD) {   // box-ordered rows -> the voxel's row in the output tensor
            int n, d, h, w;
            row_voxel(m, n, d, h, w);
            orow = (((int64_t)n * g.D + d) * g.H + h) * g.W + w;
          }
          if (vmaj) orow = (m % g.N) * ((int64_t)g.D * g.H * g.W) + m / g.N;
#pragma unroll
          for (int j = 0; j < FN; ++j) Z[orow * p.Nc + n0 + wn * (BN / 2) + j * 16 + lr] = acc[i][j][r];
        }
      }
    return;
  }
  // ---- epilogue: bias, store, BN statistics ----
  T* __restrict__ Y = reinterpret_cast<T*>(p.y);
  float s1[FN], s2[FN], bv[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    s1[j] = 0.f;
    s2[j] = 0.f;
    bv[j] = (p.bias && GEOM != GEOM_UPC_FWD) ? p.bias[n0 + wn * (BN / 2) + j * 16 + lr] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + wm * 64 + i * 16 + lg * 4 + r;
      if (m < p.M) {
        int64_t orow = m;
        if (GEOM == GEOM_UPC_DGRAD) {
          int n, d, h, w;
          row_voxel(m, n, d, h, w);
          orow = (((int64_t)n * g.D + d) * g.H + h) * g.W + w;
        }
        if (vmaj) orow = (m % g.N) * ((int64_t)g.D * g.H * g.W) + m / g.N;
        if (GEOM == GEOM_UP2_FWD || GEOM == GEOM_UPC_FWD) {
          int n, d, h, w;
          row_voxel(m, n, d, h, w);
          orow = up2_row(n, d, h, w, z, g);
          if (GEOM == GEOM_UPC_FWD) {   // border class of the fine voxel (0 first, 1 inside, 2 last per axis) -> row of the bias table
            const int fd = 2 * d + (z >> 2), fh = 2 * h + ((z >> 1) & 1), fw = 2 * w + (z & 1);
            const int cls = ((fd == 0 ? 0 : (fd == 2 * g.D - 1 ? 2 : 1)) * 3 + (fh == 0 ? 0 : (fh == 2 * g.H - 1 ? 2 : 1))) * 3 +
                            (fw == 0 ? 0 : (fw == 2 * g.W - 1 ? 2 : 1));
#pragma unroll
            for (int j = 0; j < FN; ++j) bv[j] = p.bias[cls * p.Nc + n0 + wn * (BN / 2) + j * 16 + lr];
          }
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const float v = acc[i][j][r] + bv[j];
          Y[orow * p.Nc + n0 + wn * (BN / 2) + j * 16 + lr] = from_f<T>(v);
          s1[j] += v;
          s2[j] += v * v;
        }
      }
    }
  }
  if (p.stats) {
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][FN][16][2]; all LDS reads finished at the last barrier
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
    if (tid < BN) {
      const int wn_ = tid / (BN / 2), within = tid % (BN / 2);
      const int j = within / 16, l = within % 16;
      const float* r0 = red + (((0 * 2 + wn_) * FN + j) * 16 + l) * 2;
      const float* r1 = red + (((1 * 2 + wn_) * FN + j) * 16 + l) * 2;
      const int64_t srow = (GEOM == GEOM_UPC_FWD) ? (int64_t)z * p.nt + tile_ : (int64_t)blockIdx.x;
      float* o = p.stats + (srow * p.Nc + n0 + tid) * 2;
      o[0] = r0[0] + r1[0];
      o[1] = r0[1] + r1[1];
    }
  }
}

template <typename T, int BN, int GEOM>
int launch_igemm(const IgemmParams& p, int zdim, hipStream_t stream) {
  using TL = Tile<T>;
  const size_t lds = 2 * (size_t)(PCRL_CONV_BM + BN) * TL::ROWB;
  dim3 grid((unsigned)((p.M + PCRL_CONV_BM - 1) / PCRL_CONV_BM), (unsigned)(p.Nc / BN), (unsigned)zdim);
  if (GEOM == GEOM_UPC_FWD || GEOM == GEOM_UPC_DGRAD) grid = dim3((unsigned)(p.nt * p.zdim), (unsigned)(p.Nc / BN), (unsigned)(GEOM == GEOM_UPC_DGRAD ? zdim : 1));
  hipLaunchKernelGGL((igemm_kernel<T, BN, GEOM>), grid, dim3(256), lds, stream, p);
  return pcrl_check_launch("igemm");
}

template <typename T, int GEOM> int dispatch_bn(const IgemmParams& p, int zdim, hipStream_t stream) {
  if (p.Nc % 128 == 0) return launch_igemm<T, 128, GEOM>(p, zdim, stream);
  if (p.Nc % 64 == 0) return launch_igemm<T, 64, GEOM>(p, zdim, stream);
  return launch_igemm<T, 32, GEOM>(p, zdim, stream);
}

template <int GEOM> int dispatch(const IgemmParams& p, int zdim, int dtype, hipStream_t stream) {
  if (dtype == PCRL_BF16) return dispatch_bn<bf16, GEOM>(p, zdim, stream);
  if (dtype == PCRL_F32) return dispatch_bn<float, GEOM>(p, zdim, stream);
  return pcrl_fail(PCRL_EINVAL, "igemm: bad dtype %d", dtype);
}

int check_dims(const char* what, int N, int D, int H, int W, int Ci, int Co) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return pcrl_fail(PCRL_EINVAL, "%s: bad dims %d %d %d %d", what, N, D, H, W);
  if (Ci % 32 != 0 || Co % 32 != 0 || Ci <= 0 || Co <= 0)
    return pcrl_fail(PCRL_EINVAL, "%s: channels must be positive multiples of 32 (Ci=%d Co=%d)", what, Ci, Co);
  return 0;
}

// Second pass of the split-K convolution: y = bias + sum over splits, rounded to T; (sum, sum^2) per 128-row tile and channel
// from the float sums, in the layout of the one-pass epilogue.  Block = (128-row tile, chunk of cw = Nc / gridDim.y <= 64
// channels): small M is exactly where this runs, so the channel chunks (16 wide when the row tiles are few) are what spreads
// it over the chip.  Thread = 4 channels x a row group; the loads of the splits are issued four at a time (the pass is
// latency bound: one dependent 16-byte load per split and row was 55 us for 1536 rows, rocprofv3 r02c).
template <typename T>
__global__ void __launch_bounds__(256) igemm_splitk_finish_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                                  T* __restrict__ y, float* __restrict__ stats, int64_t M, int Nc,
                                                                  int splits) {
  __shared__ float red[256 * 8];
  const int cw = Nc / (int)gridDim.y, c0 = blockIdx.y * cw;
  const int ncg = cw / 4, cg = threadIdx.x % ncg, rg = threadIdx.x / ncg, nrg = 256 / ncg;
  const int64_t m0 = (int64_t)blockIdx.x * PCRL_CONV_BM;
  const int col = c0 + cg * 4;
  f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *reinterpret_cast<const f32x4*>(bias + col);
  f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
  const int64_t zs = M * Nc;
  for (int r = rg; r < PCRL_CONV_BM; r += nrg) {
    const int64_t m = m0 + r;
    if (m >= M) break;
    const float* src = ws + m * Nc + col;
    f32x4 v0 = bv, v1 = f32x4{0.f, 0.f, 0.f, 0.f}, v2 = v1, v3 = v1;
    int z = 0;
    for (; z + 3 < splits; z += 4) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(src + (z + 0) * zs), t1 = *reinterpret_cast<const f32x4*>(src + (z + 1) * zs);
      const f32x4 t2 = *reinterpret_cast<const f32x4*>(src + (z + 2) * zs), t3 = *reinterpret_cast<const f32x4*>(src + (z + 3) * zs);
      v0 += t0; v1 += t1; v2 += t2; v3 += t3;
    }
    for (; z < splits; ++z) v0 += *reinterpret_cast<const f32x4*>(src + z * zs);
    const f32x4 v = (v0 + v1) + (v2 + v3);
    T* dst = y + m * Nc + col;
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = from_f<T>(v[q]);
    s1 += v;
    s2 += v * v;
  }
  if (stats) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      red[(threadIdx.x * 4 + q) * 2 + 0] = s1[q];
      red[(threadIdx.x * 4 + q) * 2 + 1] = s2[q];
    }
    __syncthreads();
    if ((int)threadIdx.x < cw) {
      const int c = threadIdx.x;
      float a = 0.f, b = 0.f;
      for (int g = 0; g < nrg; ++g) {
        const int t = g * ncg + c / 4;
        a += red[(t * 4 + (c & 3)) * 2 + 0];
        b += red[(t * 4 + (c & 3)) * 2 + 1];
      }
      stats[((int64_t)blockIdx.x * Nc + c0 + c) * 2 + 0] = a;
      stats[((int64_t)blockIdx.x * Nc + c0 + c) * 2 + 1] = b;
    }
  }
}

// Split-K plan of the gather kernel for a 3x3x3 convolution: volumes too small to fill the chip with 128-row tiles
// (8x8x4 crops at b=32: 128 blocks of 216 K-steps) are cut along K = 27 taps x Ci/32 chunks.
struct SplitPlan {
  int splits, steps_per_split;
};
// taps: K-steps per 32-channel chunk a tile walks -- 27, or (voxel-major rows, conv3_vmajor) the average number of taps inside the volume
static SplitPlan splitk_plan(int64_t M, int Ci, int Co, int taps = 27) {
  const int bn = Co % 128 == 0 ? 128 : (Co % 64 == 0 ? 64 : 32);
  const int64_t blocks = ((M + PCRL_CONV_BM - 1) / PCRL_CONV_BM) * (Co / bn);
  const int steps = taps * (Ci / 32);
  const bool shape_ok = Co == 32 || Co % 64 == 0;
  if (blocks >= 512 || !shape_ok) return SplitPlan{1, steps};
  int splits = (int)((768 + blocks - 1) / blocks);
  if (splits > steps / 8) splits = steps / 8;   // at least 8 K-steps per split
  if (splits < 2) return SplitPlan{1, steps};
  const int per = (steps + splits - 1) / splits;
  return SplitPlan{(steps + per - 1) / per, per};
}

}  // namespace

// LDS-halo brick kernel (conv_brick.hip)
bool pcrl_brick_conv_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype);
int64_t pcrl_brick_conv_rows(int N, int D, int H, int W);
int pcrl_brick_conv_launch(const void* x, const void* wp, const float* bias, void* y, float* stats,
                           int N, int D, int H, int W, int Ci, int Co, hipStream_t stream);

// wide-brick kernel (conv_brick16.hip): W % 16 == 0
bool pcrl_brick16_conv_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype);
int64_t pcrl_brick16_conv_rows(int N, int D, int H, int W);
int pcrl_brick16_conv_launch(const void* x, const void* wp, const float* bias, void* y, float* stats,
                             int N, int D, int H, int W, int Ci, int Co, hipStream_t stream);
void pcrl_brick16_set(int on);
void pcrl_brick16_set_planes(int mode);

bool pcrl_convt_up2_eligible(int Ci, int Co, int dtype);   // conv_up2.hip
int pcrl_convt_up2_launch(const void* x, const void* wp, const float* bias, void* y, int N, int D, int H, int W, int Ci, int Co,
                          hipStream_t stream);
static std::atomic<int> g_conv_impl{0};  // 0 = auto (brick kernel where eligible), 1 = always the gather kernel
void pcrl_brick_conv_set_ymap(int on);
// 0 = auto (brick kernel where eligible), 1 = always the gather kernel, 2 = gather kernel without split-K,
// 3 = brick kernel on its 2-D grid (channel tiles of a brick not co-located), 4 = 4x8x8-brick kernel also where the 4x8x16 one is eligible,
// 5 / 6 = auto with the wide-brick kernel on 4-plane bricks only / on 8-plane bricks wherever they tile (0: its own rule, conv_brick16.hip)
extern "C" void pcrl_debug_set_conv_impl(int impl) {
  g_conv_impl = (impl == 3 || impl == 4 || impl == 5 || impl == 6) ? 0 : impl;
  pcrl_brick_conv_set_ymap(impl != 3);
  pcrl_brick16_set(impl == 0 || impl == 5 || impl == 6);
  pcrl_brick16_set_planes(impl == 5 ? 0 : impl == 6 ? 2 : -1);
}
int pcrl_debug_conv_impl() { return g_conv_impl; }

extern "C" int64_t pcrl_conv3d_k3_stats_rows(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  if (g_conv_impl == 0 && pcrl_brick16_conv_eligible(N, D, H, W, Ci, Co, dtype)) return pcrl_brick16_conv_rows(N, D, H, W);
  if (g_conv_impl == 0 && pcrl_brick_conv_eligible(N, D, H, W, Ci, Co, dtype)) return pcrl_brick_conv_rows(N, D, H, W);
  return ((int64_t)N * D * H * W + PCRL_CONV_BM - 1) / PCRL_CONV_BM;
}

// informational: which kernel pcrl_conv3d_k3_fwd gives this shape -- 2: wide-brick (conv_brick16.hip), 1: brick (conv_brick.hip), 0: gather
extern "C" int64_t pcrl_conv3d_k3_fwd_kernel(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  if (g_conv_impl == 0 && pcrl_brick16_conv_eligible(N, D, H, W, Ci, Co, dtype)) return 2;
  if (g_conv_impl == 0 && pcrl_brick_conv_eligible(N, D, H, W, Ci, Co, dtype)) return 1;
  return 0;
}

// Voxel-major rows with per-tile tap skipping (IgemmParams::vmajor) for volumes of at most 8 voxels (the local views' 2^3 level: 8 of 27 taps per
// voxel).  Measured (tools/conv_probe.py --b 192, same box): 2^3, 256 -> 256 / 256 -> 512 channels 51 -> 28 / 65 -> 43 us; on the 4^3 level (15.6 of
// 27 taps) the same order is 5 % SLOWER (59 -> 64, 87 -> 92, 141 -> 146 us): a sample-major tile there gathers its 27 taps from the 64 voxels of two
// samples (16 KB, cache resident), a voxel-major one from 128 samples -- that level is bound by the gather, not by K, and keeps sample-major rows.
// PCRL_IGEMM_VMAJOR=0: off; =64: also the volumes of up to 64 voxels (A/B switch).  -> taps a tile walks on average, 0 = off.
static int conv3_vmajor_taps(int N, int D, int H, int W) {
  static const int vmax = [] { const char* e = getenv("PCRL_IGEMM_VMAJOR"); return e ? atoi(e) == 1 ? 8 : atoi(e) : 8; }();
  const int64_t V = (int64_t)D * H * W;
  if (V > vmax || (int64_t)N * V < PCRL_CONV_BM) return 0;
  const double t = (3.0 * D - 2) * (3.0 * H - 2) * (3.0 * W - 2) / (double)V;
  return t < 1 ? 1 : (int)(t + 0.5);
}

static int conv3d_k3_fwd_impl(const void* x, const void* wp, const float* bias, void* y, float* stats_partial, void* ws, int64_t ws_bytes,
                             int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream) {
  if (int e = check_dims("conv3d_k3_fwd", N, D, H, W, Ci, Co)) return e;
  PCRL_REQUIRE(x && wp && y, "conv3d_k3_fwd: null pointer");
  if (g_conv_impl == 0 && pcrl_brick16_conv_eligible(N, D, H, W, Ci, Co, dtype))
    return pcrl_brick16_conv_launch(x, wp, bias, y, stats_partial, N, D, H, W, Ci, Co, as_stream(stream));
  if (g_conv_impl == 0 && pcrl_brick_conv_eligible(N, D, H, W, Ci, Co, dtype))
    return pcrl_brick_conv_launch(x, wp, bias, y, stats_partial, N, D, H, W, Ci, Co, as_stream(stream));
  const int64_t M = (int64_t)N * D * H * W;
  IgemmParams p{x, wp, bias, y, stats_partial, Dims{N, D, H, W}, M, Ci, Co, 27, nullptr, 0};
  const int vtaps = conv3_vmajor_taps(N, D, H, W);
  p.vmajor = vtaps > 0;
  const SplitPlan sp = splitk_plan(M, Ci, Co, vtaps ? vtaps : 27);
  if (ws && sp.splits > 1 && g_conv_impl != 2) {
    PCRL_REQUIRE(ws_bytes >= (int64_t)sp.splits * M * Co * 4, "conv3d_k3_fwd: workspace too small (%lld bytes)", (long long)ws_bytes);
    p.ws = static_cast<float*>(ws);
    p.steps_per_split = sp.steps_per_split;
    if (int e = dispatch<GEOM_CONV3>(p, sp.splits, dtype, as_stream(stream))) return e;
    const int64_t row_tiles = (M + PCRL_CONV_BM - 1) / PCRL_CONV_BM;
    int cw = Co < 64 ? Co : 64;                                   // channel chunk of a block: narrower while the grid is small
    while (cw > 16 && row_tiles * (Co / cw) < 512) cw >>= 1;
    const dim3 grid((unsigned)row_tiles, (unsigned)(Co / cw));
    if (dtype == PCRL_BF16)
      hipLaunchKernelGGL(igemm_splitk_finish_kernel<bf16>, grid, dim3(256), 0, as_stream(stream), p.ws, bias, (bf16*)y, stats_partial, M, Co, sp.splits);
    else
      hipLaunchKernelGGL(igemm_splitk_finish_kernel<float>, grid, dim3(256), 0, as_stream(stream), p.ws, bias, (float*)y, stats_partial, M, Co, sp.splits);
    return pcrl_check_launch("conv3d_k3_fwd (split-K finish)");
  }
  return dispatch<GEOM_CONV3>(p, 1, dtype, as_stream(stream));
}

extern "C" int pcrl_conv3d_k3_fwd(const void* x, const void* wp, const float* bias, void* y, float* stats_partial,
                                  int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream) {
  return conv3d_k3_fwd_impl(x, wp, bias, y, stats_partial, nullptr, 0, N, D, H, W, Ci, Co, dtype, stream);
}

extern "C" int64_t pcrl_conv3d_k3_fwd_ws_bytes(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || Ci % 32 != 0 || Co % 32 != 0) return 0;
  if (g_conv_impl == 0 && (pcrl_brick_conv_eligible(N, D, H, W, Ci, Co, dtype) || pcrl_brick16_conv_eligible(N, D, H, W, Ci, Co, dtype))) return 0;
  const int64_t M = (int64_t)N * D * H * W;
  const int vtaps = conv3_vmajor_taps(N, D, H, W);
  const SplitPlan sp = splitk_plan(M, Ci, Co, vtaps ? vtaps : 27);
  return sp.splits > 1 ? (int64_t)sp.splits * M * Co * 4 : 0;
}

extern "C" int pcrl_conv3d_k3_fwd_ws(const void* x, const void* wp, const float* bias, void* y, float* stats_partial, void* ws,
                                     int64_t ws_bytes, int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream) {
  return conv3d_k3_fwd_impl(x, wp, bias, y, stats_partial, ws, ws_bytes, N, D, H, W, Ci, Co, dtype, stream);
}

// ---- data gradient + first pass of the BatchNorm backward of the layer below (conv_brick16_bnr.hip) ----
int pcrl_brick16_dgrad_bnred_launch(const void* dy, const void* wp, void* dx, const void* bn_y, const float* scale, const float* shift, const float* mean,
                                    const float* rstd, float* partial, int N, int D, int H, int W, int Ci, int Co, hipStream_t stream);
extern "C" int64_t pcrl_conv3d_k3_dgrad_bnred_rows(int N, int D, int H, int W, int Ci, int Co, int act, int dtype) {
  static const bool off = [] { const char* e = getenv("PCRL_DGRAD_BNRED"); return e && e[0] == '0'; }();   // A/B switch
  if (off || act != PCRL_ACT_RELU || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return 0;
  if (g_conv_impl == 0 && pcrl_brick16_conv_eligible(N, D, H, W, Ci, Co, dtype)) return pcrl_brick16_conv_rows(N, D, H, W);
  return 0;
}
extern "C" int pcrl_conv3d_k3_dgrad_bnred(const void* dy, const void* wp_dgrad, void* dx, const void* bn_y, const float* scale, const float* shift,
                                          const float* mean, const float* rstd, float* partial, int N, int D, int H, int W, int Ci, int Co, int act,
                                          int dtype, pcrl_stream_t stream) {
  if (int e = check_dims("conv3d_k3_dgrad_bnred", N, D, H, W, Ci, Co)) return e;
  PCRL_REQUIRE(dy && wp_dgrad && dx && bn_y && scale && shift && mean && rstd && partial, "conv3d_k3_dgrad_bnred: null pointer");
  PCRL_REQUIRE(pcrl_conv3d_k3_dgrad_bnred_rows(N, D, H, W, Ci, Co, act, dtype) > 0,
               "conv3d_k3_dgrad_bnred: no fused kernel for this shape / activation / dtype (pcrl_conv3d_k3_dgrad_bnred_rows == 0)");
  return pcrl_brick16_dgrad_bnred_launch(dy, wp_dgrad, dx, bn_y, scale, shift, mean, rstd, partial, N, D, H, W, Ci, Co, as_stream(stream));
}

extern "C" int pcrl_convt3d_k2s2_fwd(const void* x, const void* wp_fwd, const float* bias, void* y,
                                     int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream) {
  if (int e = check_dims("convt3d_k2s2_fwd", N, D, H, W, Ci, Co)) return e;
  PCRL_REQUIRE(x && wp_fwd && y, "convt3d_k2s2_fwd: null pointer");
  if (g_conv_impl == 0 && pcrl_convt_up2_eligible(Ci, Co, dtype))
    return pcrl_convt_up2_launch(x, wp_fwd, bias, y, N, D, H, W, Ci, Co, as_stream(stream));
  IgemmParams p{x, wp_fwd, bias, y, nullptr, Dims{N, D, H, W}, (int64_t)N * D * H * W, Ci, Co, 1, nullptr, 0};
  return dispatch<GEOM_UP2_FWD>(p, 8, dtype, as_stream(stream));
}

extern "C" int pcrl_convt3d_k2s2_dgrad(const void* dy, const void* wp_dgrad, void* dx,
                                       int N, int D, int H, int W, int Ci, int Co, int dtype, pcrl_stream_t stream) {
  if (int e = check_dims("convt3d_k2s2_dgrad", N, D, H, W, Ci, Co)) return e;
  PCRL_REQUIRE(dy && wp_dgrad && dx, "convt3d_k2s2_dgrad: null pointer");
  // rows = input voxels, K per tap = Co (channels of dy), output channels = Ci
  IgemmParams p{dy, wp_dgrad, nullptr, dx, nullptr, Dims{N, D, H, W}, (int64_t)N * D * H * W, Co, Ci, 8, nullptr, 0};
  return dispatch<GEOM_UP2_DGRAD>(p, 1, dtype, as_stream(stream));
}

// ---- fused ConvTranspose3d(k2,s2) -> Conv3d(3x3x3): the two MFMA passes (internal; the C ABI is in upconv_fused.hip) ----
// tile shape of the row order: a 128-voxel box that divides the volume, as cubic as the extents allow
static void upc_box(int D, int H, int W, IgemmParams& p) {
  static const int cand[][3] = {{4, 4, 8}, {4, 8, 4}, {8, 4, 4}, {2, 8, 8}, {8, 8, 2}, {8, 2, 8}, {2, 4, 16}, {4, 2, 16}, {1, 8, 16}, {8, 16, 1}, {16, 8, 1}, {1, 4, 32}};
  p.bd = p.bh = p.bw = 0;
  for (auto& c : cand)
    if (D % c[0] == 0 && H % c[1] == 0 && W % c[2] == 0) {
      p.bd = c[0]; p.bh = c[1]; p.bw = c[2];
      return;
    }
}
template <int GEOM> static int launch_upc(IgemmParams& p, int zdim, int dtype, hipStream_t stream, int splits = 1) {
  upc_box(p.g.D, p.g.H, p.g.W, p);
  p.nt = (int)((p.M + PCRL_CONV_BM - 1) / PCRL_CONV_BM);
  p.zdim = zdim;
  if ((int64_t)p.nt * zdim >= ((int64_t)1 << 31)) return pcrl_fail(PCRL_EINVAL, "upconv: grid too large");
  return dispatch<GEOM>(p, splits, dtype, stream);      // the third grid dimension: K splits (data gradient only)
}
int pcrl_upc_fwd_launch(const void* x, const void* wf, const float* bias_tab, void* y0, float* stats, int N, int D, int H, int W, int Ci, int Co,
                        int dtype, hipStream_t stream) {
  IgemmParams p{x, wf, bias_tab, y0, stats, Dims{N, D, H, W}, (int64_t)N * D * H * W, Ci, Co, 8, nullptr, 0};
  return launch_upc<GEOM_UPC_FWD>(p, 8, dtype, stream);
}
bool pcrl_brick16_upc_fwd_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype);   // conv_brick16.hip
bool pcrl_brick8_upc_fwd_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype);    // conv_brick.hip
// 0: gather kernel; 1: wide-brick kernel (conv_brick16.hip); 2: 4 x 8 x 8-brick kernel (conv_brick.hip)
int pcrl_upc_fwd_impl(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  if (g_conv_impl != 0) return 0;
  if (pcrl_brick16_upc_fwd_eligible(N, D, H, W, Ci, Co, dtype)) return 1;
  return pcrl_brick8_upc_fwd_eligible(N, D, H, W, Ci, Co, dtype) ? 2 : 0;
}
bool pcrl_upc_fwd_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype) { return pcrl_upc_fwd_impl(N, D, H, W, Ci, Co, dtype) != 0; }
bool pcrl_brick16_upc_dgrad_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype);   // conv_brick16.hip
bool pcrl_brick8_upc_dgrad_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype);    // conv_brick.hip
int pcrl_upc_dgrad_impl(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  if (g_conv_impl != 0) return 0;
  if (pcrl_brick16_upc_dgrad_eligible(N, D, H, W, Ci, Co, dtype)) return 1;
  return pcrl_brick8_upc_dgrad_eligible(N, D, H, W, Ci, Co, dtype) ? 2 : 0;
}
bool pcrl_upc_dgrad_uses_brick(int N, int D, int H, int W, int Ci, int Co, int dtype) { return pcrl_upc_dgrad_impl(N, D, H, W, Ci, Co, dtype) != 0; }
// Split-K plan of the composed data gradient on the gather kernel: K = 64 taps x Co / 32 chunks (512 steps at up_tr256) over row tiles that
// are few on the coarse grids it serves -- the 8 x 8 x 4 grid of up_tr256 (64 tiles x 4 channel tiles = one block per CU, four waves: 319 us,
// 430 TFLOP/s) and the 2^3 / 4^3 grids of the local views (12 tiles: 287 us, 90 TFLOP/s).  Splits bring the grid to ~4 blocks per CU.
static SplitPlan upc_dgrad_plan(int64_t M, int Ci, int Co) {
  const int bn = Ci % 128 == 0 ? 128 : (Ci % 64 == 0 ? 64 : 32);
  const int64_t blocks = ((M + PCRL_CONV_BM - 1) / PCRL_CONV_BM) * (Ci / bn);
  const int steps = 64 * (Co / 32);
  if (blocks >= 512) return SplitPlan{1, steps};
  int splits = (int)((1024 + blocks - 1) / blocks);
  if (splits > 16) splits = 16;
  if (splits > steps / 16) splits = steps / 16;   // at least 16 K-steps per split
  if (splits < 2) return SplitPlan{1, steps};
  const int per = (steps + splits - 1) / splits;
  return SplitPlan{(steps + per - 1) / per, per};
}
int64_t pcrl_upc_dgrad_ws_bytes(int N, int D, int H, int W, int Ci, int Co) {
  const int64_t M = (int64_t)N * D * H * W;
  const SplitPlan sp = upc_dgrad_plan(M, Ci, Co);
  return sp.splits > 1 ? (int64_t)sp.splits * M * Ci * 4 : 0;
}
int pcrl_upc_dgrad_launch(const void* dy0, const void* wd, void* dx, void* ws, int64_t ws_bytes, int N, int D, int H, int W, int Ci, int Co, int dtype,
                          hipStream_t stream) {
  // rows = coarse voxels, K per tap = Co (channels of dy0), 64 taps, output channels = Ci
  const int64_t M = (int64_t)N * D * H * W;
  IgemmParams p{dy0, wd, nullptr, dx, nullptr, Dims{N, D, H, W}, M, Co, Ci, 64, nullptr, 0};
  const SplitPlan sp = upc_dgrad_plan(M, Ci, Co);
  if (ws && sp.splits > 1 && ws_bytes >= (int64_t)sp.splits * M * Ci * 4) {
    p.ws = static_cast<float*>(ws);
    p.steps_per_split = sp.steps_per_split;
    if (int e = launch_upc<GEOM_UPC_DGRAD>(p, 1, dtype, stream, sp.splits)) return e;
    const int64_t row_tiles = (M + PCRL_CONV_BM - 1) / PCRL_CONV_BM;
    int cw = Ci < 64 ? Ci : 64;
    while (cw > 16 && row_tiles * (Ci / cw) < 512) cw >>= 1;
    const dim3 grid((unsigned)row_tiles, (unsigned)(Ci / cw));
    if (dtype == PCRL_BF16)
      hipLaunchKernelGGL(igemm_splitk_finish_kernel<bf16>, grid, dim3(256), 0, stream, p.ws, (const float*)nullptr, (bf16*)dx, (float*)nullptr, M, Ci, sp.splits);
    else
      hipLaunchKernelGGL(igemm_splitk_finish_kernel<float>, grid, dim3(256), 0, stream, p.ws, (const float*)nullptr, (float*)dx, (float*)nullptr, M, Ci, sp.splits);
    return pcrl_check_launch("upconv_dgrad (split-K finish)");
  }
  return launch_upc<GEOM_UPC_DGRAD>(p, 1, dtype, stream);
}
// Plain GEMM with a float32 plane-major result: z[col * M + m] = sum_k a[m][k] * b[col][k]   (M % 4 == 0, K % 32 == 0, Nc % 32 == 0)
template <typename T, int BN> static void gemm_planes_bn(const IgemmParams& p, hipStream_t stream) {
  const unsigned gx = (unsigned)((p.M + PCRL_CONV_BM - 1) / PCRL_CONV_BM);
  const size_t lds = 2 * (size_t)(PCRL_CONV_BM + BN) * Tile<T>::ROWB;
  hipLaunchKernelGGL((igemm_kernel<T, BN, GEOM_UP2_FWD, true>), dim3(gx, (unsigned)(p.Nc / BN), 1), dim3(256), lds, stream, p);
}
int pcrl_gemm_planes_launch(const void* a, const void* b, float* z, int64_t M, int K, int Nc, int dtype, hipStream_t stream) {
  if (M <= 0 || M % 4 != 0 || K % 32 != 0 || Nc % 32 != 0) return pcrl_fail(PCRL_EINVAL, "gemm_planes: bad sizes M=%lld K=%d Nc=%d", (long long)M, K, Nc);
  IgemmParams p{a, b, nullptr, z, nullptr, Dims{1, 1, 1, (int)(M > 0x7fffffff ? 0x7fffffff : M)}, M, K, Nc, 1, nullptr, 0};
  if (dtype == PCRL_BF16) {
    if (Nc % 128 == 0) gemm_planes_bn<bf16, 128>(p, stream);
    else if (Nc % 64 == 0) gemm_planes_bn<bf16, 64>(p, stream);
    else gemm_planes_bn<bf16, 32>(p, stream);
  } else if (dtype == PCRL_F32) {
    if (Nc % 128 == 0) gemm_planes_bn<float, 128>(p, stream);
    else if (Nc % 64 == 0) gemm_planes_bn<float, 64>(p, stream);
    else gemm_planes_bn<float, 32>(p, stream);
  } else {
    return pcrl_fail(PCRL_EINVAL, "gemm_planes: bad dtype %d", dtype);
  }
  return pcrl_check_launch("gemm_planes");
}

// Pointwise product for the C -> 1 convolutions (conv_c1.hip): z[t][m] = sum_c x[m][c] * wt[t][c], t < 32 (27 taps + zero
// padding), float32 plane-major output.  wt: [32][C] in `dtype`.  C % 32 == 0.
int pcrl_pointwise_planes_launch(const void* x, const void* wt, float* z, int64_t M, int C, int dtype, hipStream_t stream) {
  IgemmParams p{x, wt, nullptr, z, nullptr, Dims{1, 1, 1, 1}, M, C, 32, 1};
  p.g = Dims{(int)1, 1, 1, (int)(M > 0x7fffffff ? 0x7fffffff : M)};
  const unsigned gx = (unsigned)((M + PCRL_CONV_BM - 1) / PCRL_CONV_BM);
  if (dtype == PCRL_BF16) {
    const size_t lds = 2 * (size_t)(PCRL_CONV_BM + 32) * Tile<bf16>::ROWB;
    hipLaunchKernelGGL((igemm_kernel<bf16, 32, GEOM_UP2_FWD, true>), dim3(gx, 1, 1), dim3(256), lds, stream, p);
  } else {
    const size_t lds = 2 * (size_t)(PCRL_CONV_BM + 32) * Tile<float>::ROWB;
    hipLaunchKernelGGL((igemm_kernel<float, 32, GEOM_UP2_FWD, true>), dim3(gx, 1, 1), dim3(256), lds, stream, p);
  }
  return pcrl_check_launch("pointwise_planes");
}
