// LDS-halo ("brick") implicit-GEMM 3x3x3 convolution for gfx950, bf16 -- the throughput path of
// pcrl_conv3d_k3_fwd (forward and data-gradient) for volumes whose D,H,W are multiples of the brick.
//
// Replaces aten::convolution / convolution_backward(input) of LUConv.conv1 (models/pcrlv2_model_3d.py:9,33).
//
// Why: the gather kernel (conv_igemm.hip) fetches the A operand of every (tap, 32-channel) K-step from L2: 16 KB per
// 256 MFMA-cycles and CU, above what one CU gets from its XCD's L2 (~56 B/clk).  Here a block owns a spatial brick of
// TD x 8 x 8 output voxels; per 32-channel chunk it stages the brick PLUS its one-voxel halo ((TD+2) x 10 x 10 rows of
// 64 B, zero-filled outside the volume) in LDS once, and all 27 taps read their shifted rows from LDS.  Global traffic
// per chunk: the halo once (2.3x the brick) + 27 weight tiles, ~4x less than the gather kernel; MFMA operands are
// 16-byte LDS reads (A: halo row of the lane's voxel + tap offset; B: weight row), 8 x 2 fragments of
// v_mfma_f32_16x16x32_bf16 per wave and tap, three taps (one kw row) per barrier.
//
// Block = 256 threads = 4 waves; wave tile = 64 voxels x all 64 output channels (4 A + 4 B fragment reads per 16 MFMAs).
// LDS: halo 960 rows (w pitch padded to 16, bank-conflict-free for every tap offset) x 64 B = 60 KiB, single buffer --
//      the next chunk is prefetched into registers during the last tap row -- + one weight stage (3 taps x 64 x 64 B =
//      12 KiB, next stage prefetched into registers) -> 72 KiB, two blocks per CU.
#include "common.h"

// (Measured in round 4, not kept, removed: transposed MFMAs (D = W * X^T) so that a lane holds four consecutive channels of one voxel and the epilogue
// issues 16 eight-byte stores instead of 64 two-byte ones -- 44.7 ms against 43.6 on the C5 step: an eight-byte store of that layout touches 16
// different 32-byte segments, the two-byte form 4, and the per-channel statistics need four shuffle steps instead of two.)
#ifndef BRICK_ABL
#define BRICK_ABL 0   // timing ablations (wrong results): bit 0 no output stores, bit 1 no global loads in front of the first stage
#endif
#include <atomic>
#include <mutex>

namespace {

constexpr int TD = 4, TH = 8, TW = 8;
constexpr int BRICK = TD * TH * TW;                    // 256 voxels
constexpr int HH = TH + 2, HWU = TW + 2;               // halo extents in h, w (used)
constexpr int HW = 16;                                 // halo row pitch in w: padded 10 -> 16, see hoff_h()
constexpr int HLT = 240;                               // threads that stage the halo: 6 lines x 40 pieces per round
// KD = kernel extent in d: 3 (the 3x3x3 convolution) or 1 (a 3x3 convolution over a stack of independent images: the 2D path,
// conv2d.hip, runs its [N][H][W][C] batches through this kernel with the batch index as d -- no halo, no mixing, in d).
template <int KD> struct BrickGeom {
  static constexpr int HD = TD + KD - 1;               // halo extent in d
  static constexpr int LINES = HD * HH;                // (d,h) lines of 10 rows: 60 / 40
  static constexpr int HPT = (LINES + 5) / 6;          // staging rounds = pieces per thread: 10 / 7 (the last round of KD = 1 is partly idle)
  static constexpr int HALO_BYTES = HPT * 6 * HW * 64; // 60 KiB / 42 KiB
  static constexpr int TAPS = 9 * KD, NS = 3 * KD;     // taps; stages (one (kd,kh) = three kw taps) per 32-channel chunk
};
// one weight stage: 3 taps x BN rows x 64 B (12 KiB for BN = 64)

struct BrickParams {
  const bf16* x;
  const bf16* w;      // packed [Nc][taps][K]
  const float* bias;
  bf16* y;
  float* stats;       // [bricks][Nc][2] or null
  int N, D, H, W;
  int K, Nc;
  int up;             // KD = 1 only: the source is [N][D][H/2][W/2] read through a nearest x2 upsample (conv2d.hip)
  int ny;             // > 0: 1-D grid of bricks x ny ids, the ny output-channel tiles of a brick on consecutive slots of ONE XCD (they
                      // stage the same halo: fetched into that L2 once); 0: 2-D grid (bricks, tiles)
  // Axis permutation (KD = 3 only).  The kernel tiles an index space (D, H, W) = "brick axes" with 4 x 8 x 8 bricks; sd, sh, sw are the
  // strides (in voxels) of those axes in memory and td, th, tw the tap-index strides (a permutation of 9, 3, 1), so a volume whose
  // innermost extent is 4 (the 8 x 8 x 4 bottleneck level) runs with its W as the 4-deep brick axis.  Identity: (H*W, W, 1), (9, 3, 1).
  int sd, sh, sw, td, th, tw;
  // Composed modes (MODE 1 / 2, KD = 3; the operator and its weight forms are described at conv_brick16.hip's Brick16Params -- this kernel takes
  // the coarse grids the wide brick does not tile: the 8 x 8 x 4 grid of up_tr256 and the 8^3 grid of the local views' up_tr64).
  // MODE 1 (forward): x coarse, Nc = 8 * upc = 8 phases x upc channels, w the zero-embedded weights [8 * upc][27][K]; a block (one phase) walks the
  // 4 of 9 (kd, kh) stages x 2 kw taps of its phase, writes the phase's FINE voxels of y [N][2D][2H][2W][upc] and adds bias_tab[border class].
  // MODE 2 (data gradient): x = dy0 on the fine grid read as its space-to-depth view, K = 8 parities x upc (chunk c lies in parity c >> cshift),
  // w = [Nc = Ci][27][8 * upc]; a chunk walks the stages / taps of its parity; the output is an ordinary coarse tensor.
  int upc;
  const float* bias_tab;
  int cshift;
  int bd, bh, bw;        // bit of the phase / parity number (d = 4, h = 2, w = 1 of the MEMORY axes) that belongs to the brick's d, h, w axis: 2, 1, 0
  int fsd, fsh, fsw;     // strides, in fine voxels, of the brick's axes on the fine grid: 4 H W, 2 W, 1
};

// Weight tile [64 co][32 k]: a fragment read takes 16 CONSECUTIVE rows -> same swizzle as conv_igemm.hip's Tile<bf16>.
__device__ __forceinline__ int hoff_w(int row, int slot) {
  const int key = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
  return row * 64 + ((slot ^ key) << 4);
}
// Halo rows (64 B each).  An A-fragment read of tap (kd,kh,kw) takes rows R + (lr & 7) + HW * (lr >> 3) for ARBITRARY R
// (R moves with the tap), slot lr >> 4.  gfx950 serves ds_read_b128 in 16-lane groups {0-3,12-15,20-27}, ... so one
// group mixes rows R+0..3 and R+HW+4..7 at slot s with rows R+4..7 and R+HW+0..3 at slot s^1.  Exhaustive search over
// row pitches and XOR keys (period <= 8 in row>>1..3): with the natural pitch HW = 10 every key leaves >= 2-way
// conflicts on most reads (measured: SQ_LDS_BANK_CONFLICT = 55 % of SQ_LDS_IDX_ACTIVE); pitch 16 with
// key = 2 * ((row >> 2) & 1) is conflict-free for every R.
__device__ __forceinline__ int hoff_h(int row, int slot) {
  return row * 64 + ((slot ^ (((row >> 2) & 1) << 1)) << 4);
}

template <int BN, int KD, int MODE = 0>   // BN = output channels per block: 64, or 32 for the Co = 32 data gradient (wave tile 64 voxels x BN)
__global__ void __launch_bounds__(256, 2) brick_conv_kernel(const BrickParams p) {
  using BG = BrickGeom<KD>;
  static_assert(MODE == 0 || KD == 3, "the composed modes are 3-D");
  constexpr bool UPCF = MODE == 1, UPCD = MODE == 2;
  constexpr int HPT = BG::HPT, HALO_BYTES = BG::HALO_BYTES, NS = MODE ? 4 : BG::NS;
  constexpr int NKW = MODE ? 2 : 3;   // kw taps per stage
  constexpr int FN = BN / 16;
  constexpr int WT_BYTES = 3 * BN * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wbuf = smem + HALO_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;   // wave = 64 voxels x all 64 output channels
  const int lr = lane & 15, lg = lane >> 4;
  const int K = p.K, nchunk = K / 32;

  // ---- brick origin ----
  const int bw = p.W / TW, bh = p.H / TH, bd = p.D / TD;
  // consecutive workgroups go to different XCDs (8, each with its own L2): give every XCD a CONTIGUOUS range of bricks so that
  // the halo rows shared by neighbouring bricks are fetched into one L2 once
  int b = blockIdx.x, ytile = blockIdx.y;
  if (p.ny > 0) {
    const int nbr = gridDim.x / p.ny;
    if ((nbr & 7) == 0) {
      const int slot = b >> 3;
      ytile = slot % p.ny;
      b = (b & 7) * (nbr >> 3) + slot / p.ny;
    } else {
      ytile = b % p.ny;
      b = b / p.ny;
    }
  } else if ((gridDim.x & 7) == 0) {
    b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
  }
  const int n0 = ytile * BN;
  // composed modes: phase of the block (forward) / parity of a chunk (data gradient), per BRICK axis
  const int uph = UPCF ? n0 / p.upc : 0;
  const int upd = (uph >> p.bd) & 1, uphh = (uph >> p.bh) & 1, upw = (uph >> p.bw) & 1;
#define PARC(c_) ((c_) >> p.cshift)
#define PBIT(c_, b_) ((PARC(c_) >> (b_)) & 1)
  // stage number -> kd * 3 + kh of the brick axes.  Forward: the phase uses k = p, p + 1 per axis; data gradient: the parity uses k = 1 - par, 2 - par.
#define SID(c_, s_) (MODE == 0 ? (s_) : MODE == 1 ? ((upd + ((s_) >> 1)) * 3 + uphh + ((s_)&1)) \
                                                 : ((1 - PBIT(c_, p.bd) + ((s_) >> 1)) * 3 + 1 - PBIT(c_, p.bh) + ((s_)&1)))
#define KWB(c_) (MODE == 0 ? 0 : MODE == 1 ? upw : 1 - PBIT(c_, p.bw))   /* first of the kw taps in use */
  const int brick_id = b;
  const int w0 = (b % bw) * TW; b /= bw;
  const int h0 = (b % bh) * TH; b /= bh;
  const int d0 = (b % bd) * TD; b /= bd;
  const int n = b;

  // ---- halo pieces of this thread: 240 threads x 10 rounds; a round covers 6 (d,h) lines of 10 rows x 4 pieces, so the
  //      LDS offset of round i is the offset of round 0 + i * 6 lines (an immediate).  Global row clamped to a valid one.
  const int htid = tid < HLT ? tid : tid - HLT;   // the last 16 threads repeat the first 16 (same address, same data)
  const int hq = htid % 40, hl0 = htid / 40;
  const int hslot = hq & 3;
  int grow[HPT];
  uint32_t hvalid = 0;
#pragma unroll
  for (int i = 0; i < HPT; ++i) {
    const int line = hl0 + 6 * i;
    const int hd = line / HH, hh = line % HH, hw = hq >> 2;
    const int d = d0 + hd - (KD - 1) / 2, h = h0 + hh - 1, w = w0 + hw - 1;
    const bool ok = line < BG::LINES && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
    if (KD == 1 && p.up) {
      const int Hs = p.H >> 1, Ws = p.W >> 1;
      grow[i] = ok ? ((n * p.D + d) * Hs + (h >> 1)) * Ws + (w >> 1) : ((n * p.D + d0) * Hs + (h0 >> 1)) * Ws + (w0 >> 1);
    } else if (UPCD) {   // fine voxel 2 v (+ the chunk's parity) of the coarse halo voxel v
      grow[i] = n * (8 * p.D * p.H * p.W) + (ok ? 2 * d * p.fsd + 2 * h * p.fsh + 2 * w * p.fsw : 2 * d0 * p.fsd + 2 * h0 * p.fsh + 2 * w0 * p.fsw);
    } else {
      grow[i] = n * (p.D * p.H * p.W) + (ok ? d * p.sd + h * p.sh + w * p.sw : d0 * p.sd + h0 * p.sh + w0 * p.sw);
    }
    hvalid |= (uint32_t)ok << i;
  }
  const int hdst0 = hoff_h(hl0 * HW + (hq >> 2), hslot);   // + i * 6 * HW * 64

  // ---- weight staging: 3 pieces per thread (tap kw = 0,1,2 of the current (kd,kh)), row co = tid>>2, slot tid&3 ----
  const bool wthread = BN == 64 || (tid >> 2) < BN;   // BN = 32: only the first 128 threads stage weights
  const bf16* wrow = p.w + ((int64_t)(n0 + (wthread ? (tid >> 2) : 0)) * BG::TAPS) * K + (tid & 3) * 8;
  const int wdst = hoff_w(tid >> 2, tid & 3);   // within one tap tile [64][32]

  f32x4 acc[4][FN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 rh[HPT], rw[3];

#define LOAD_HALO(c_)                                                                                     \
  do {                                                                                                    \
    if (UPCD) {                                                                                           \
      const int poff = PBIT(c_, p.bd) * p.fsd + PBIT(c_, p.bh) * p.fsh + PBIT(c_, p.bw) * p.fsw;          \
      const int co0 = ((c_) - (PARC(c_) << p.cshift)) * 32 + hslot * 8;                                   \
      _Pragma("unroll") for (int i = 0; i < HPT; ++i)                                                     \
        rh[i] = *reinterpret_cast<const u32x4*>(p.x + (int64_t)(grow[i] + poff) * p.upc + co0);           \
    } else {                                                                                              \
      _Pragma("unroll") for (int i = 0; i < HPT; ++i)                                                     \
        rh[i] = *reinterpret_cast<const u32x4*>(p.x + (int64_t)grow[i] * K + (c_)*32 + hslot * 8);        \
    }                                                                                                     \
  } while (0)
#define STORE_HALO()                                                                                      \
  do {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < HPT; ++i) {                                                     \
      *reinterpret_cast<u32x4*>(halo + hdst0 + i * (6 * HW * 64)) = keep_if((hvalid >> i) & 1u, rh[i]);       \
    }                                                                                                     \
  } while (0)
#define LOAD_W(c_, s9_, kwb_) /* taps kwb_ .. kwb_ + NKW - 1 of stage (kd, kh) = s9_ */                    \
  do {                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < NKW; ++j)                                                       \
      rw[j] = *reinterpret_cast<const u32x4*>(wrow + (int64_t)(KD == 3 ? ((s9_) / 3) * p.td + ((s9_) % 3) * p.th + ((kwb_) + j) * p.tw : (s9_)*3 + j) * K + (c_)*32); \
  } while (0)
#define STORE_W()                                                                                         \
  do {                                                                                                    \
    if (wthread) {                                                                                        \
      _Pragma("unroll") for (int j = 0; j < NKW; ++j)                                                     \
        *reinterpret_cast<u32x4*>(wbuf + j * (BN * 64) + wdst) = rw[j];                                   \
    }                                                                                                     \
  } while (0)

  // Fragment addressing.  The wave's 64 voxels are one d-plane of the brick: voxel row of fragment fm = abase + fm * 32,
  // so the swizzle key of a halo read depends only on (lr & 7) + kw (tap offsets in kd, kh are multiples of the pitch):
  // three byte offsets per lane (one per kw), the (kd,kh) offset is a block-uniform add, fm is an immediate.
  int akw[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) akw[kw] = hoff_h((wid * HH + (lr >> 3)) * HW + (lr & 7) + kw, lg);
  const int bofs = hoff_w(lr, lg);   // weight row j * 16 + lr: same key for every j -> + j * 1024

  // Two fragment sets in ping-pong: the reads of tap t+1 are issued before the MFMAs of tap t.
  bf16x8 fa[2][4], fb[2][FN];
#define LOADF(B_, aoff_, wt_)                                                                             \
  do {                                                                                                    \
    fa[B_][0] = *reinterpret_cast<const bf16x8*>(halo + (aoff_));                                         \
    _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                        \
      fb[B_][j] = *reinterpret_cast<const bf16x8*>((wt_) + bofs + j * 1024);                              \
    _Pragma("unroll") for (int fm = 1; fm < 4; ++fm)                                                      \
      fa[B_][fm] = *reinterpret_cast<const bf16x8*>(halo + (aoff_) + fm * (2 * HW * 64));                 \
  } while (0)
// scheduling pipelines: n x { m MFMA, 1 LDS read }  /  n x { m MFMA, 1 LDS write }
#define PIPE_READS(n_, m_)                                                                                \
  do {                                                                                                    \
    _Pragma("unroll") for (int q = 0; q < (n_); ++q) {                                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, (m_), 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                  \
    }                                                                                                     \
  } while (0)
#define PIPE_WRITES(n_, m_)                                                                               \
  do {                                                                                                    \
    _Pragma("unroll") for (int q = 0; q < (n_); ++q) {                                                    \
      __builtin_amdgcn_sched_group_barrier(0x008, (m_), 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                                  \
    }                                                                                                     \
  } while (0)
#define MFMA_ROWS(B_, F0_, F1_)                                                                           \
  do {                                                                                                    \
    _Pragma("unroll") for (int fm = (F0_); fm < (F1_); ++fm)                                              \
      _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                      \
        acc[fm][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[B_][fm], fb[B_][j], acc[fm][j], 0, 0, 0);   \
  } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

  // One stage = the three kw taps of one (kd,kh) of one 32-channel chunk.  P = fragment set that holds tap kw = 0 on
  // entry.  Both barriers of the stage sit between halves of tap kw = 2's MFMAs, whose operands are already in
  // registers: the weight (and, once per chunk, halo) stores and the first reads of the next stage run under them.
  int c = 0, s9 = 0;
#define STAGE(P_)                                                                                         \
  do {                                                                                                    \
    int cn = c, sn = s9 + 1;                                                                              \
    if (sn == NS) { sn = 0; cn = c + 1; }                                                                  \
    const bool last = (cn == nchunk);                                                                     \
    if (last) { cn = c; sn = s9; }                                                                        \
    LOAD_W(cn, sn, 0);                                                                                    \
    const bool halo_next = (s9 == NS - 1) && !last; /* block-uniform */                                        \
    if (halo_next) LOAD_HALO(c + 1);                                                                      \
    const int tap64 = ((s9 / 3) * HH + (s9 % 3)) * (HW * 64);                                             \
    const int ntap64 = ((sn / 3) * HH + (sn % 3)) * (HW * 64);                                            \
    LOADF((P_) ^ 1, akw[1] + tap64, wbuf + 1 * (BN * 64));                                                \
    MFMA_ROWS(P_, 0, 4);                                                                                  \
    PIPE_READS(4 + FN, 16 / (4 + FN));                                                                    \
    SB();                                                                                                 \
    LOADF(P_, akw[2] + tap64, wbuf + 2 * (BN * 64));                                                      \
    MFMA_ROWS((P_) ^ 1, 0, 4);                                                                            \
    PIPE_READS(4 + FN, 16 / (4 + FN));                                                                    \
    SB();                                                                                                 \
    __syncthreads(); /* every wave holds its kw = 2 fragments: weight stage (and halo) may be replaced */ \
    STORE_W();                                                                                            \
    if (halo_next) {                                                                                      \
      STORE_HALO();                                                                                       \
      MFMA_ROWS(P_, 0, 2);                                                                                \
      PIPE_WRITES(2 * FN, 1);                                                                             \
    } else {                                                                                              \
      MFMA_ROWS(P_, 0, 2);                                                                                \
      PIPE_WRITES(3, 2);                                                                                  \
    }                                                                                                     \
    SB();                                                                                                 \
    __syncthreads();                                                                                      \
    LOADF((P_) ^ 1, akw[0] + ntap64, wbuf);                                                               \
    MFMA_ROWS(P_, 2, 4);                                                                                  \
    PIPE_READS(4 + FN, 1);                                                                                \
    SB();                                                                                                 \
    c = cn;                                                                                               \
    s9 = sn;                                                                                              \
  } while (0)

  // Composed modes: a stage = the TWO kw taps its phase / parity uses (fragment set 0 = first tap, set 1 = second); same barriers, the
  // sets no longer flip between stages.
#define STAGE2()                                                                                          \
  do {                                                                                                    \
    int cn = c, sn = s9 + 1;                                                                              \
    if (sn == NS) { sn = 0; cn = c + 1; }                                                                  \
    const bool last = (cn == nchunk);                                                                     \
    if (last) { cn = c; sn = s9; }                                                                        \
    const int kwb = KWB(c), kwbn = KWB(cn);                                                               \
    LOAD_W(cn, SID(cn, sn), kwbn);                                                                        \
    const bool halo_next = (s9 == NS - 1) && !last; /* block-uniform */                                   \
    if (halo_next) LOAD_HALO(c + 1);                                                                      \
    const int tap64 = ((SID(c, s9) / 3) * HH + (SID(c, s9) % 3)) * (HW * 64);                             \
    const int ntap64 = ((SID(cn, sn) / 3) * HH + (SID(cn, sn) % 3)) * (HW * 64);                          \
    const int a1 = kwb ? akw[2] : akw[1], an0 = kwbn ? akw[1] : akw[0];                                   \
    LOADF(1, a1 + tap64, wbuf + 1 * (BN * 64));                                                           \
    MFMA_ROWS(0, 0, 4);                                                                                   \
    PIPE_READS(4 + FN, 16 / (4 + FN));                                                                    \
    SB();                                                                                                 \
    __syncthreads(); /* every wave holds its second-tap fragments: weight stage (and halo) may be replaced */ \
    STORE_W();                                                                                            \
    if (halo_next) {                                                                                      \
      STORE_HALO();                                                                                       \
      MFMA_ROWS(1, 0, 2);                                                                                 \
      PIPE_WRITES(2 * FN, 1);                                                                             \
    } else {                                                                                              \
      MFMA_ROWS(1, 0, 2);                                                                                 \
      PIPE_WRITES(2, 2);                                                                                  \
    }                                                                                                     \
    SB();                                                                                                 \
    __syncthreads();                                                                                      \
    LOADF(0, an0 + ntap64, wbuf);                                                                         \
    MFMA_ROWS(1, 2, 4);                                                                                   \
    PIPE_READS(4 + FN, 1);                                                                                \
    SB();                                                                                                 \
    c = cn;                                                                                               \
    s9 = sn;                                                                                              \
  } while (0)

  if (BRICK_ABL & 2) {   // timing ablation: no global latency in front of the first stage
#pragma unroll
    for (int i = 0; i < HPT; ++i) rh[i] = u32x4{1u, 2u, 3u, 4u};
#pragma unroll
    for (int j = 0; j < 3; ++j) rw[j] = u32x4{1u, 2u, 3u, 4u};
  } else {
    LOAD_HALO(0);
    LOAD_W(0, SID(0, 0), KWB(0));
  }
  STORE_HALO();
  STORE_W();
  __syncthreads();
  LOADF(0, (KWB(0) ? akw[1] : akw[0]) + ((SID(0, 0) / 3) * HH + (SID(0, 0) % 3)) * (HW * 64), wbuf);

  const int nstage = NS * nchunk;
  if constexpr (MODE == 0) {
    for (int S = 0; S + 1 < nstage; S += 2) {
      STAGE(0);
      STAGE(1);
    }
    if (nstage & 1) STAGE(0);
  } else {
    for (int S = 0; S < nstage; ++S) STAGE2();
  }
  __syncthreads();   // the epilogue reuses the LDS
#undef STAGE
#undef STAGE2
#undef SID
#undef KWB
#undef PARC
#undef PBIT
#undef SB
#undef MFMA_ROWS
#undef LOADF
#undef PIPE_READS
#undef PIPE_WRITES
#undef LOAD_HALO
#undef STORE_HALO
#undef LOAD_W
#undef STORE_W

  // ---- epilogue: bias, store, BatchNorm partial statistics (one row per brick).  Voxel of (fm, r) in the wave's 64: d offset = wave, h offset =
  //      2 fm + (lg >> 1), w offset = 4 (lg & 1) + r.  Addresses (round 5, as in conv_brick16.h): a SCALAR base per (fm, r) -- the wave's first voxel +
  //      2 fm h-steps + r w-steps -- plus ONE lane offset and the store's immediate (32 j bytes): the 64-bit voxel index per (fm, r) that was rebuilt
  //      on the vector unit (34 quarter-rate multiplies + 17 64-bit multiply-adds per lane; with 9 taps per chunk in the 2D geometry the epilogue and
  //      prologue were MORE instructions than the stage loop) is gone. ----
  float s1[FN], s2[FN], bv[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    s1[j] = 0.f;
    s2[j] = 0.f;
    bv[j] = (!UPCF && p.bias) ? p.bias[n0 + j * 16 + lr] : 0.f;
  }
  const int uch0 = UPCF ? n0 - uph * p.upc : n0;   // first channel of this tile (inside its phase)
  const int ypitch = UPCF ? p.upc : p.Nc;
  // byte steps of the output along the brick's h and w axes (composed forward: a coarse step is two fine voxels)
  const int64_t HS = (int64_t)(UPCF ? 2 * p.fsh : p.sh) * ypitch * 2, WS = (int64_t)(UPCF ? 2 * p.fsw : p.sw) * ypitch * 2;
  const int64_t row0 = UPCF ? (int64_t)n * (8 * p.D * p.H * p.W) + (int64_t)(2 * (d0 + wid) + upd) * p.fsd + (int64_t)(2 * h0 + uphh) * p.fsh + (int64_t)(2 * w0 + upw) * p.fsw
                            : (int64_t)n * p.D * p.H * p.W + (int64_t)(d0 + wid) * p.sd + (int64_t)h0 * p.sh + (int64_t)w0 * p.sw;
  char* const ybase = reinterpret_cast<char*>(p.y) + (row0 * ypitch + uch0) * 2;      // wave-uniform
  const uint32_t yl = (uint32_t)((lg >> 1) * HS + (lg & 1) * 4 * WS + lr * 2);
  const int fd_ = 2 * (d0 + wid) + upd;
  const int cd = fd_ == 0 ? 0 : (fd_ == 2 * p.D - 1 ? 2 : 1);
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (UPCF) {   // bias of the fine voxel's border class (0 first, 1 inside, 2 last per axis; class number in memory order = tap strides)
        const int fh = 2 * (h0 + 2 * fm + (lg >> 1)) + uphh, fw = 2 * (w0 + 4 * (lg & 1) + r) + upw;
        const int ch = fh == 0 ? 0 : (fh == 2 * p.H - 1 ? 2 : 1), cw = fw == 0 ? 0 : (fw == 2 * p.W - 1 ? 2 : 1);
        const int cls = cd * p.td + ch * p.th + cw * p.tw;
#pragma unroll
        for (int j = 0; j < FN; ++j) bv[j] = p.bias_tab[cls * p.upc + uch0 + j * 16 + lr];
      }
      char* const yrow = ybase + (int64_t)(2 * fm) * HS + (int64_t)r * WS;   // scalar
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const float val = acc[fm][j][r] + bv[j];
        if (!(BRICK_ABL & 1) || val == 12345.678f) *reinterpret_cast<bf16*>(yrow + yl + j * 32) = (bf16)val;
        s1[j] += val;
        s2[j] += val * val;
      }
    }
  }
  if (p.stats) {
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][64 ch][2]; the loop ended with a barrier
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float a = s1[j], c2 = s2[j];
      a += __shfl_xor(a, 16, 64);
      c2 += __shfl_xor(c2, 16, 64);
      a += __shfl_xor(a, 32, 64);
      c2 += __shfl_xor(c2, 32, 64);
      if (lg == 0) {
        red[(wid * 64 + j * 16 + lr) * 2 + 0] = a;
        red[(wid * 64 + j * 16 + lr) * 2 + 1] = c2;
      }
    }
    __syncthreads();
    if (tid < BN) {
      float a = 0.f, c2 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a += red[(q * 64 + tid) * 2 + 0];
        c2 += red[(q * 64 + tid) * 2 + 1];
      }
      float* o = p.stats + ((int64_t)brick_id * p.Nc + n0 + tid) * 2;
      o[0] = a;
      o[1] = c2;
    }
  }
}

}  // namespace

// ---- internal interface used by conv_igemm.hip's dispatcher -------------------------------------------------------
// natural orientation (W % 8 == 0), or W % 4 == 0 with D % 8 == 0 and H % 8 == 0 (W becomes the 4-deep brick axis: the 8 x 8 x 4 level)
static bool brick_natural(int D, int H, int W) { return D % TD == 0 && H % TH == 0 && W % TW == 0; }
static bool brick_permuted(int D, int H, int W) { return W % TD == 0 && D % TH == 0 && H % TW == 0; }
bool pcrl_brick_conv_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return dtype == PCRL_BF16 && (brick_natural(D, H, W) || brick_permuted(D, H, W)) && Ci % 32 == 0 && Co % 32 == 0 &&
         (int64_t)N * D * H * W < (int64_t)1 << 31;
}
int64_t pcrl_brick_conv_rows(int N, int D, int H, int W) { return (int64_t)N * D * H * W / BRICK; }

static std::atomic<int> g_brick_ymap{1};
void pcrl_brick_conv_set_ymap(int on) { g_brick_ymap = on; }

int pcrl_brick_conv_launch(const void* x, const void* wp, const float* bias, void* y, float* stats,
                           int N, int D, int H, int W, int Ci, int Co, hipStream_t stream) {
  constexpr int HB = BrickGeom<3>::HALO_BYTES;
  static std::once_flag attr_once;   // hipFuncSetAttribute once per process, race-free
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(brick_conv_kernel<64, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, HB + 3 * 64 * 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(brick_conv_kernel<32, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, HB + 3 * 32 * 64);
  });
  BrickParams p{(const bf16*)x, (const bf16*)wp, bias, (bf16*)y, stats, N, D, H, W, Ci, Co, 0, 0, H * W, W, 1, 9, 3, 1};
  if (W % TW != 0) {   // W == 4: memory (D, H, W) -> brick axes (W, D, H); tap (kd', kh', kw') = (kw, kd, kh) -> index kh' * 9 + kw' * 3 + kd'
    p.D = W; p.H = D; p.W = H;
    p.sd = 1; p.sh = H * W; p.sw = W;
    p.td = 1; p.th = 9; p.tw = 3;
  }
  const unsigned bricks = (unsigned)pcrl_brick_conv_rows(N, D, H, W);
  // few bricks (the 8 x 8 x 4 level: 32 at b = 32): 32-channel tiles double the number of blocks (128 -> 256 for 256 output channels)
  const int BN = (Co % 64 == 0 && (int64_t)bricks * (Co / 64) > 192) ? 64 : 32, ny = Co / BN;
  dim3 grid(bricks, ny);
  if (g_brick_ymap && ny > 1 && (uint64_t)bricks * ny < (1u << 31)) {
    p.ny = ny;
    grid = dim3(bricks * ny);
  }
  if (BN == 64) hipLaunchKernelGGL((brick_conv_kernel<64, 3>), grid, dim3(256), HB + 3 * 64 * 64, stream, p);
  else hipLaunchKernelGGL((brick_conv_kernel<32, 3>), grid, dim3(256), HB + 3 * 32 * 64, stream, p);
  return pcrl_check_launch("brick_conv");
}

// ---- 2D path: 3x3 / stride 1 / pad 1 convolution of a stack of N images (N % 4 == 0), optionally behind a nearest x2 upsample ----
// H, W = OUTPUT dims (= logical input dims); weights packed [Co][9][Ci] (forward) or [Ci][9 flipped][Co] (data gradient).
bool pcrl_brick_conv2d_eligible(int N, int H, int W, int Ci, int Co, int dtype) {
  return dtype == PCRL_BF16 && N % TD == 0 && H % TH == 0 && W % TW == 0 && Ci % 32 == 0 && Co % 32 == 0 && (int64_t)N * H * W < (int64_t)1 << 31;
}
int64_t pcrl_brick_conv2d_rows(int N, int H, int W) { return (int64_t)(N / TD) * (H / TH) * (W / TW); }

int pcrl_brick_conv2d_launch(const void* x, const void* wp, const float* bias, void* y, float* stats, int N, int H, int W, int Ci, int Co, int up,
                             hipStream_t stream) {
  constexpr int HB = BrickGeom<1>::HALO_BYTES;
  BrickParams p{(const bf16*)x, (const bf16*)wp, bias, (bf16*)y, stats, 1, N, H, W, Ci, Co, up, 0, H * W, W, 1, 9, 3, 1};
  const unsigned bricks = (unsigned)pcrl_brick_conv2d_rows(N, H, W);
  if (Co % 64 == 0) hipLaunchKernelGGL((brick_conv_kernel<64, 1>), dim3(bricks, Co / 64), dim3(256), HB + 3 * 64 * 64, stream, p);
  else hipLaunchKernelGGL((brick_conv_kernel<32, 1>), dim3(bricks, Co / 32), dim3(256), HB + 3 * 32 * 64, stream, p);
  return pcrl_check_launch("brick_conv2d");
}

// ---- composed ConvTranspose3d -> Conv3d operator (upconv_fused.hip) on the 4 x 8 x 8 brick: forward and data gradient ----
// x: coarse [N][D][H][W][Ci]; w3: zero-embedded weights [8 * Co][27][Ci]; y0: fine [N][2D][2H][2W][Co]; stats [bricks * 8][Co][2]
static void upc_axes(BrickParams& p, int D, int H, int W) {
  p.sd = H * W; p.sh = W; p.sw = 1; p.td = 9; p.th = 3; p.tw = 1;
  p.bd = 2; p.bh = 1; p.bw = 0;
  p.fsd = 4 * H * W; p.fsh = 2 * W; p.fsw = 1;
  if (!brick_natural(D, H, W)) {   // memory (D, H, W) -> brick axes (W, D, H), as in pcrl_brick_conv_launch
    p.D = W; p.H = D; p.W = H;
    p.sd = 1; p.sh = H * W; p.sw = W;
    p.td = 1; p.th = 9; p.tw = 3;
    p.bd = 0; p.bh = 2; p.bw = 1;
    p.fsd = 1; p.fsh = 4 * H * W; p.fsw = 2 * W;
  }
}
template <int BN, int MODE> static void launch_upc8(const BrickParams& p, unsigned nblocks, hipStream_t stream) {
  constexpr int HB = BrickGeom<3>::HALO_BYTES;
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(brick_conv_kernel<BN, 3, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, HB + 3 * BN * 64);
  });
  hipLaunchKernelGGL((brick_conv_kernel<BN, 3, MODE>), dim3(nblocks), dim3(256), HB + 3 * BN * 64, stream, p);
}
bool pcrl_brick8_upc_fwd_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return pcrl_brick_conv_eligible(N, D, H, W, Ci, 8 * Co, dtype) && Co % 64 == 0 && (int64_t)N * D * H * W * 8 < ((int64_t)1 << 29);
}
int pcrl_brick8_upc_fwd_launch(const void* x, const void* w3, const float* bias_tab, void* y0, float* stats, int N, int D, int H, int W, int Ci, int Co,
                               hipStream_t stream) {
  BrickParams p{(const bf16*)x, (const bf16*)w3, nullptr, (bf16*)y0, stats, N, D, H, W, Ci, 8 * Co, 0, 8 * Co / 64, 0, 0, 0, 0, 0, 0, Co, bias_tab, 0};
  upc_axes(p, D, H, W);
  const int64_t blocks = pcrl_brick_conv_rows(N, D, H, W) * p.ny;
  if (blocks >= ((int64_t)1 << 31)) return pcrl_fail(PCRL_EINVAL, "brick (composed up-conv): grid too large");
  launch_upc8<64, 1>(p, (unsigned)blocks, stream);
  return pcrl_check_launch("brick_conv (composed up-conv forward)");
}
// dy0: fine [N][2D][2H][2W][Co]; wd3: zero-embedded weights [Ci][27][8 * Co]; dx: coarse [N][D][H][W][Ci]
static int upc8_cshift(int Co) {
  for (int k = 0; k < 8; ++k)
    if (Co == (32 << k)) return k;
  return -1;
}
bool pcrl_brick8_upc_dgrad_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return upc8_cshift(Co) >= 0 && pcrl_brick_conv_eligible(N, D, H, W, 8 * Co, Ci, dtype) && (int64_t)N * D * H * W * 8 < ((int64_t)1 << 29);
}
int pcrl_brick8_upc_dgrad_launch(const void* dy0, const void* wd3, void* dx, int N, int D, int H, int W, int Ci, int Co, hipStream_t stream) {
  const int64_t bricks = pcrl_brick_conv_rows(N, D, H, W);
  const int BN = (Ci % 64 == 0 && bricks * (Ci / 64) > 192) ? 64 : 32;
  BrickParams p{(const bf16*)dy0, (const bf16*)wd3, nullptr, (bf16*)dx, nullptr, N, D, H, W, 8 * Co, Ci, 0, Ci / BN, 0, 0, 0, 0, 0, 0, Co, nullptr, upc8_cshift(Co)};
  upc_axes(p, D, H, W);
  const int64_t blocks = bricks * p.ny;
  if (blocks >= ((int64_t)1 << 31)) return pcrl_fail(PCRL_EINVAL, "brick (composed up-conv data gradient): grid too large");
  if (BN == 64) launch_upc8<64, 2>(p, (unsigned)blocks, stream);
  else launch_upc8<32, 2>(p, (unsigned)blocks, stream);
  return pcrl_check_launch("brick_conv (composed up-conv data gradient)");
}
