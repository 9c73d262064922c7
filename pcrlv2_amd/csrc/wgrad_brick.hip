// LDS-halo ("brick") weight-gradient kernel for the 3x3x3 convolution, bf16, gfx950 -- the throughput path of
// pcrl_conv3d_k3_wgrad for volumes with D % 2 == 0, H % 8 == 0, W % 8 == 0 and Co % 64 == 0.
//
// Replaces the weight half of aten::convolution_backward for LUConv.conv1 (models/pcrlv2_model_3d.py:9,33).
//
//   dW[co][ci][t] = sum_m dy[m][co] * x[m + delta_t][ci]
//
// The gather kernel (conv_wgrad.hip) re-reads both operand tiles from L2 for each of the 27 taps (8 KB per 64 MFMA
// cycles: L2-bandwidth bound, ~240 TF).  Here a block owns a TCO(co) x TCI(ci) tile of ONE kd plane (9 taps) and walks
// over bricks of 2 x 8 x 8 = 128 voxels: per brick it stages dy[128][TCO] and the x halo for that kd
// ([2][10][12-padded] rows x TCI ch, zero outside the volume) in LDS ONCE and runs all 9 (kh,kw) taps from it.
// Operands are fetched with ds_read_b64_tr_b16 (the reduction index is the row index of both NDHWC tensors).
// Each wave owns 64 co x 16 ci of the 9 taps (36 accumulator fragments): the dy fragments are read once per
// 32-voxel K-chunk and reused by 9 taps, the x fragment of a tap feeds 4 MFMAs -> 13 KB of LDS reads per 36 MFMAs.
// Two LDS brick buffers: the next brick's pieces are requested (LDS-DMA) while the current brick is multiplied (one wait + barrier per
// brick).  Split-K over brick ranges; the partial slabs are reduced in fixed order by the same second pass as the gather kernel.
//
// Tile shapes.  Measured (timing ablation, same box): without the staging requests the kernel runs at 1 840 TFLOP/s instead of
// 1 250 -- moving 46 KB per brick and block from L2 into LDS (8.7 TB/s chip-wide) is what it spends its power on; fragment reads and
// barriers are free.  Staged bytes per MFMA go as 256 / TCI + 400 / TCO, so the eight waves of a block are used in one of four ways:
//   64 x 64   two wave groups split a brick's four K chunks (own accumulators, combined through LDS at the end)        10.3
//   128 x 64  two dy images; group g owns co half g, every wave walks all four K chunks (Co % 128 == 0)                  7.1
//   64 x 128  two x images; group g owns ci half g (Ci % 128 == 0)                                                       8.3
//   64 x 32   Ci == 32: a wave owns 32 co x 16 ci (two A fragments) -- with 64 x 64 tiles half of the waves multiplied zero columns
#include "common.h"
#include <atomic>
#include <mutex>

// x-fragment prefetch distance in steps.  Two waves per SIMD hide the LDS latency between them: 1 is enough (2 and 3 measured equal).
#ifndef WB_PF
#define WB_PF 1
#endif
#ifndef WB_ABL
#define WB_ABL 0   // timing ablations (wrong results): 1 no staging requests, 2 no x-fragment reads after the first brick, 3 no per-brick wait + barrier,
#endif             // 4 every brick's pieces from the FIRST brick's addresses (same staging traffic, all of it L2 hits)
#ifndef WB_PRIO
#define WB_PRIO 1  // 1: a wave's issue priority falls with the step inside a brick (s_setprio 3 .. 0 per quarter): of the two waves of a SIMD the one
#endif             // that is BEHIND wins the arbitration.  0: equal priorities -- the hardware then favours the older wave (waves 0-3), which reaches
                   // every end-of-brick barrier 28-37 % of a brick early (s_memtime accounting, profiles/r06_wgrad_trace.txt)
#ifndef WB_TRACE
#define WB_TRACE 0 // 1 (probe builds only, tools/wgrad_trace.py): s_memtime accounting per block, summed into g_wb_trace
#endif

namespace {

constexpr int BD = 2, BH = 8, BW = 8;
constexpr int BV = BD * BH * BW;          // 128 voxels per brick
constexpr int XH = BH + 2, XW = 12;       // halo extents (w padded 10 -> 12 for conflict-free transpose reads)
constexpr int XROWS = BD * XH * XW;       // 240
constexpr int DY_BYTES = BV * 128;        // one dy image [128 voxels][64 ch]: 16 KiB
constexpr int X_BYTES = XROWS * 128;      // one x image [240 halo rows][64 ch]: 30 KiB
constexpr int NT = 512;                   // eight waves = two groups of four: two waves per SIMD
constexpr int DYP = BV * 8 / NT;          // 16-byte pieces per thread and dy image (2)
constexpr int XP = (XROWS * 8 + NT - 1) / NT;  // pieces per thread and x image (4: 1920 pieces)

template <int TCO, int TCI> struct WCfg {
  static constexpr int IMG_DY = TCO / 64, IMG_X = TCI == 128 ? 2 : 1;
  static constexpr int KSPLIT = (TCO == 64 && TCI <= 64) ? 2 : 1;   // wave groups split the K chunks of a brick
  static constexpr bool HALFCI = TCI == 32;
  static constexpr int FA = HALFCI ? 2 : 4;                          // dy fragments (16 co each) per wave
  static constexpr int NSTEP = 36 / KSPLIT;                          // steps per wave and brick (K chunks x taps)
  static constexpr int BUF_BYTES = IMG_DY * DY_BYTES + IMG_X * X_BYTES;
  static constexpr int NPIECE = IMG_DY * DYP + IMG_X * XP;
  static_assert(NPIECE <= NSTEP, "one staging request per step");
};

struct WBrickParams {
  const bf16* dy;   // [M][Cu]
  const bf16* x;    // [M][Cv]
  float* ws;        // [splits][27][Cu][Cv]
  int N, D, H, W;
  int Cu, Cv;
  int nbricks, per_split;
  int up;           // nkd = 1 only: x is [D][H/2][W/2][Cv] read through a nearest x2 upsample (decoder conv1 of the 2D path)
  int nkd;          // 3: the 3x3x3 convolution (blockIdx.y = tile * 3 + kd); 1: a 3x3 convolution over a stack of images (2D path:
                    // the image index is d and only the centre plane kd = 1 exists; slabs hold 9 taps)
  // XCD co-located launch (nkd = 3, see plan_xcd()): a 1-D grid in chunks of 256 ids; the G = 3 * gs blocks that walk the SAME brick
  // range (three kd planes x gs tiles) get ids that are congruent mod 8, i.e. run on one XCD at the same time, so that the dy tile and
  // the overlapping x planes of a brick are fetched into that XCD's L2 once and hit there G - 1 times.  0 = the 2-D grid above.
  // Axis permutation (nkd = 3, up = 0): D, H, W above are the extents of the BRICK axes (tiled 2 x 8 x 8), sd / sh / sw their strides
  // in memory (voxels) and td / th / tw the tap-index strides (a permutation of 9, 3, 1): a volume with innermost extent 4 (the
  // 8 x 8 x 4 level) runs with its W as the 2-deep brick axis.  Identity: (H * W, W, 1), (9, 3, 1).
  int sd, sh, sw, td, th, tw;
  int xcd_map;
  int order;        // brick walk order: 0 = w, h, d, n ; 1 = w, d, h, n
  int G, Q, gpc, ngroups, ntg, pair;   // group size, groups per XCD and chunk, groups per chunk, groups, tile groups per range, 0 none / 1 pair over j / 2 pair over i
};

__device__ __forceinline__ int dy_off(int v, int col) {   // 128-byte rows, 32-byte quads XOR-swizzled (see conv_wgrad.hip)
  const int key = ((v >> 1) & 1) | (((v >> 3) & 1) << 1);
  return v * 128 + ((((col >> 4) ^ key) & 3) << 5) + ((col & 15) << 1);
}
__device__ __forceinline__ int x_off(int row, int col) {  // rows R0+{0..3} and R0+12+{0..3} of one read -> distinct quads
  return row * 128 + ((((col >> 4) ^ (row >> 1)) & 3) << 5) + ((col & 15) << 1);
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
__device__ uint4 g_wb_zero[4];   // source of halo rows outside the volume (zero-initialised device memory)
#if WB_TRACE
// [0] block cycles, [1] cycles between the last step and the end of the barrier (wave 0), [2] cycles of the steps that issue staging requests (wave 0),
// [3] bricks, [4] blocks; [8 + 2 w] / [9 + 2 w]: wave w's cycles in the vmcnt wait / in the barrier behind it
__device__ unsigned long long g_wb_trace[32];
#define WB_T_DECL() unsigned long long t_all_ = __builtin_amdgcn_s_memtime(), t_wait_ = 0, t_issue_ = 0, t_b0_ = 0, t_w0_ = 0, t_vm_ = 0; int t_n_ = 0; (void)t_b0_; (void)t_w0_
#define WB_T_BRICK() do { __builtin_amdgcn_sched_barrier(0); t_b0_ = __builtin_amdgcn_s_memtime(); ++t_n_; } while (0)
#define WB_T_ISSUED() do { __builtin_amdgcn_sched_barrier(0); t_issue_ += __builtin_amdgcn_s_memtime() - t_b0_; } while (0)
#define WB_T_WAIT0() do { __builtin_amdgcn_sched_barrier(0); t_w0_ = __builtin_amdgcn_s_memtime(); } while (0)
#define WB_T_MID() do { __builtin_amdgcn_sched_barrier(0); t_vm_ += __builtin_amdgcn_s_memtime() - t_w0_; } while (0)
#define WB_T_WAIT1() do { __builtin_amdgcn_sched_barrier(0); t_wait_ += __builtin_amdgcn_s_memtime() - t_w0_; } while (0)
#define WB_T_END() do { if ((threadIdx.x & 63) == 0) { atomicAdd(&g_wb_trace[8 + 2 * (threadIdx.x >> 6)], t_vm_); atomicAdd(&g_wb_trace[9 + 2 * (threadIdx.x >> 6)], t_wait_ - t_vm_); } \
  if (threadIdx.x == 0) { atomicAdd(&g_wb_trace[0], __builtin_amdgcn_s_memtime() - t_all_); atomicAdd(&g_wb_trace[1], t_wait_); \
    atomicAdd(&g_wb_trace[2], t_issue_); atomicAdd(&g_wb_trace[3], (unsigned long long)t_n_); atomicAdd(&g_wb_trace[4], 1ull); } } while (0)
#else
#define WB_T_DECL() do {} while (0)
#define WB_T_BRICK() do {} while (0)
#define WB_T_ISSUED() do {} while (0)
#define WB_T_WAIT0() do {} while (0)
#define WB_T_MID() do {} while (0)
#define WB_T_WAIT1() do {} while (0)
#define WB_T_END() do {} while (0)
#endif
// the quarter-of-the-brick priority (WB_PRIO): call with the compile-time step index and the steps per brick
#if WB_PRIO
#define WB_SETPRIO(st_, n_)                                          \
  do {                                                               \
    if ((st_) == 0) __builtin_amdgcn_s_setprio(3);                   \
    else if ((st_) == (n_) / 4) __builtin_amdgcn_s_setprio(2);       \
    else if ((st_) == (n_) / 2) __builtin_amdgcn_s_setprio(1);       \
    else if ((st_) == 3 * (n_) / 4) __builtin_amdgcn_s_setprio(0);   \
  } while (0)
#else
#define WB_SETPRIO(st_, n_) do {} while (0)
#endif

// One LDS-DMA request: every lane's 16 bytes at `gsrc` land at LDS byte address `lds_dst` (wave-uniform) + 16 * lane.  M0 carries the
// destination and is compiler-reserved: saved and restored inside the statement.  Not counted by hipcc: wait with s_waitcnt vmcnt.
__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ bf16x8 tr_frag(const char* p0, const char* p1) {
  union { struct { s16x4 a, b; } s; bf16x8 f; } u;
  u.s.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p0);
  u.s.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p1);
  return u.f;
}

template <int TCO, int TCI>
__global__ void __launch_bounds__(NT, 1) wgrad_brick_kernel(const WBrickParams p) {
  using C = WCfg<TCO, TCI>;
  constexpr int FA = C::FA, NSTEP = C::NSTEP, BUF_BYTES = C::BUF_BYTES, NPIECE = C::NPIECE, KSPLIT = C::KSPLIT;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // two (dy images, x halo images) brick buffers of BUF_BYTES

  const int tid = threadIdx.x, lane = tid & 63, wid = (tid >> 6) & 3, grp = tid >> 8;
  const int lg = lane >> 4, jr = (lane & 15) >> 2, cq = lane & 3;
  const int ntj = (p.Cv + TCI - 1) / TCI;
  int kd, tile, split;
  if (p.xcd_map) {
    const int id = blockIdx.x, chunk = id >> 8, r = id & 255, QG8 = 8 * p.Q * p.G;
    int rank, m;   // rank of the block's group inside its chunk (filled XCD by XCD in rounds, so a part-filled chunk stays balanced); member
    if (r < QG8) {
      const int slot = r >> 3;
      rank = (slot / p.G) * 8 + (r & 7);
      m = slot % p.G;
    } else {       // the 16 ids of a chunk that do not make whole co-located groups: plain groups of consecutive ids
      rank = 8 * p.Q + (r - QG8) / p.G;
      m = (r - QG8) % p.G;
      if (rank >= p.gpc) return;
    }
    const int g = chunk * p.gpc + rank;
    if (g >= p.ngroups) return;
    split = g / p.ntg;
    const int tg = g % p.ntg, sub = m / 3;
    kd = m % 3;
    tile = p.pair == 0 ? tg : p.pair == 1 ? 2 * tg + sub : (2 * (tg / ntj) + sub) * ntj + tg % ntj;
  } else {
    kd = p.nkd == 3 ? blockIdx.y % 3 : 1;
    tile = blockIdx.y / p.nkd;
    split = blockIdx.x;
  }
  const int i0 = (tile / ntj) * TCO, j0 = (tile % ntj) * TCI;
  // roles of this wave: dy image / x image of its group, its 16-ci block and (TCI = 32) its co half
  const int dyimg = C::IMG_DY == 2 ? grp : 0, ximg = C::IMG_X == 2 ? grp : 0;
  const int cib = C::HALFCI ? (wid & 1) : wid, cobase = C::HALFCI ? (wid >> 1) * 32 : 0;
  const int b_beg = split * p.per_split;
  const int b_end = min(b_beg + p.per_split, p.nbricks);
  const int bw = p.W / BW, bh = p.H / BH, bd = p.D / BD;
  const int Hs = p.up ? p.H >> 1 : p.H, Ws = p.up ? p.W >> 1 : p.W;   // extents of x as stored

  f32x4 acc[9][FA];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int f = 0; f < FA; ++f) acc[t][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging roles: dy piece q = tid + 256*i -> voxel q>>3, 16-byte piece q&7 ; x piece likewise over 240 halo rows.
  // Address generation is kept off the critical path: a brick's pieces sit at FIXED byte offsets from the brick's first
  // halo voxel, computed once per thread; per brick only a block-uniform 64-bit base pointer, a 6-bit boundary mask and
  // one select per piece remain.  (The first version decoded every piece per brick with 64-bit multiplies under divergent
  // branches: 2400 of the 6200 cycles a brick took, with the matrix pipe idle -- one wave per SIMD.)
  // Staging is LDS-DMA (global_load_lds_dwordx4): a wave's request fills 1 KiB = 8 rows x 128 B lane-linearly, so lane l lands on row
  // l >> 3, PHYSICAL 16-byte position l & 7 of that row; the XOR swizzle of the images is therefore applied to the SOURCE: the lane
  // fetches the logical piece that belongs at its position (the keys repeat every 64 rows, so one logical piece per thread and image).
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave index 0 .. NT / 64 - 1
  const int pp = tid & 7, srow = tid >> 3;
  const int key_dy = ((srow >> 1) & 1) | (((srow >> 3) & 1) << 1), key_x = (srow >> 1) & 3;
  const int pc_dy = ((((pp >> 1) ^ key_dy) & 3) << 1) | (pp & 1), pc = ((((pp >> 1) ^ key_x) & 3) << 1) | (pp & 1);
  // channel validity of this thread's x piece per image (Cv is a multiple of 32: a 64-wide image may hang over; TCI = 32 uses half an image)
  bool jok[C::IMG_X];
#pragma unroll
  for (int m = 0; m < C::IMG_X; ++m) jok[m] = (j0 + 64 * m + pc * 8) < p.Cv && (!C::HALFCI || pc < 4);
  const int xcol = j0 + pc * 8;
  uint32_t dyoff[DYP], xoff[XP];   // byte offsets from the brick bases
  uint32_t xedge[XP];              // which faces of the halo the piece's row lies on (bit: d-,d+,h-,h+,w-,w+); bit 6 = not a halo row
#pragma unroll
  for (int i = 0; i < DYP; ++i) {
    const int v = (tid >> 3) + (NT / 8) * i;
    dyoff[i] = (uint32_t)(((v >> 6) * p.sd + ((v >> 3) & 7) * p.sh + (v & 7) * p.sw) * p.Cu + pc_dy * 8) * 2u;
  }
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int r = (tid >> 3) + (NT / 8) * i;
    const int hd = r / (XH * XW), hh = (r / XW) % XH, hw = r % XW;
    const bool row_ok = r < XROWS && hw < BW + 2;
    // upsampled source: brick origins are even, so halo row hh maps to source row (h0/2 - 1) + ((hh + 1) >> 1): still a fixed offset
    const int shh = p.up ? (hh + 1) >> 1 : hh, shw = p.up ? (hw + 1) >> 1 : hw;
    xoff[i] = row_ok ? (uint32_t)((p.up ? (hd * Hs + shh) * Ws + shw : hd * p.sd + hh * p.sh + hw * p.sw) * p.Cv + xcol) * 2u : 0u;
    xedge[i] = (hd == 0 ? 1u : 0u) | (hd == BD - 1 ? 2u : 0u) | (hh == 0 ? 4u : 0u) | (hh == XH - 1 ? 8u : 0u) | (hw == 0 ? 16u : 0u) |
               (hw == BW + 1 ? 32u : 0u) | (row_ok ? 0u : 64u);
  }
  // Staging pipeline (two LDS brick buffers): while brick b is multiplied out of one buffer, the pieces of brick b+1 are requested
  // into the other one, one request per wave on each of the first DYP + XP steps (they have the rest of the brick to land); rows
  // outside the volume come from a 16-byte zero page.  No staging registers, no ds_write, no selects; one wait + barrier per brick.
  const char* zpage = reinterpret_cast<const char*>(g_wb_zero);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the buffers
  const char* dyb = nullptr;   // block-uniform bases of the brick being LOADED
  const char* xb = nullptr;
  uint32_t xout = 0;
  // Brick coordinates are CARRIED from brick to brick (the bricks a block originates are consecutive): decoding them with
  // three divisions and three remainders per brick sat in front of the MFMA steps, unhidden (one wave per SIMD issues in order).
  int ob = b_beg, ow0, oh0, od0, on;
  {
    // walk order: w, h, d, n -- or, with the co-located launch, w, d, h, n: the kd blocks of a group read x planes d0 + kd - 1 and
    // d0 + kd, so a plane is read again by the next brick in d; with d second-fastest that is bw bricks later (an L2 hit), not bw * bh.
    int t_ = b_beg;
    ow0 = (t_ % bw) * BW; t_ /= bw;
    if (p.order) {
      od0 = (t_ % bd) * BD; t_ /= bd;
      oh0 = (t_ % bh) * BH; t_ /= bh;
    } else {
      oh0 = (t_ % bh) * BH; t_ /= bh;
      od0 = (t_ % bd) * BD; t_ /= bd;
    }
    on = t_;
  }
  // Byte offsets of the brick being loaded are CARRIED as well (round 6): dyo = its first voxel in dy, xo = its first halo voxel in x; a step
  // along an axis and the wrap at its end are precomputed 64-bit constants.  Rebuilding them from (n, d0, h0, w0) took ~50 scalar instructions
  // with eight 64-bit multiplies per brick, executed by all eight waves at once right behind the barrier -- with the matrix pipe idle.
  const int64_t evx = p.up ? 1 : p.sw, ehx = p.up ? Ws : p.sh, edx = p.up ? (int64_t)Hs * Ws : p.sd;   // x strides (voxels) of the brick axes
  const int64_t cub = 2 * (int64_t)p.Cu, cvb = 2 * (int64_t)p.Cv;
  const int64_t dsW = BW * p.sw * cub, dsH = BH * p.sh * cub, dsD = BD * p.sd * cub;                      // dy: one brick along w / h / d
  const int64_t dwW = (int64_t)p.W * p.sw * cub, dwH = (int64_t)p.H * p.sh * cub, dwD = (int64_t)p.D * p.sd * cub;   // ... and a whole axis
  const int64_t dsN = (int64_t)p.D * p.H * p.W * cub;
  const int64_t xsW = (p.up ? BW / 2 : BW) * evx * cvb, xsH = (p.up ? BH / 2 : BH) * ehx * cvb, xsD = BD * edx * cvb;
  const int64_t xwW = (p.up ? p.W / 2 : p.W) * evx * cvb, xwH = (p.up ? p.H / 2 : p.H) * ehx * cvb, xwD = (int64_t)p.D * edx * cvb;
  const int64_t xsN = (p.up ? (int64_t)p.D * Hs * Ws : (int64_t)p.D * p.H * p.W) * cvb;
  int64_t dyo = ((int64_t)on * p.D * p.H * p.W + (int64_t)od0 * p.sd + (int64_t)oh0 * p.sh + (int64_t)ow0 * p.sw) * cub + 2 * (int64_t)i0;
  // first halo voxel (d0 + kd - 1, h0 - 1, w0 - 1): may lie outside the volume, never dereferenced then
  int64_t xo = p.up ? ((((int64_t)on * p.D + od0 + (kd - 1)) * Hs + (oh0 >> 1)) * Ws + (ow0 >> 1) - Ws - 1) * cvb
                    : ((int64_t)on * p.D * p.H * p.W + (int64_t)(od0 + kd - 1) * p.sd + (int64_t)(oh0 - 1) * p.sh + (int64_t)(ow0 - 1) * p.sw) * cvb;
#define WB_ORIGIN_NEXT()                                                                                     \
  do {                                                                                                       \
    const int w0 = ow0, h0 = oh0, d0 = od0;                                                                  \
    dyb = reinterpret_cast<const char*>(p.dy) + dyo;                                                         \
    xb = reinterpret_cast<const char*>(p.x) + xo;                                                            \
    /* faces of this brick's halo that stick out of the volume (BD = 2: the d faces are the two planes) */    \
    xout = (d0 + kd - 1 < 0 ? 1u : 0u) | (d0 + kd - 1 + BD - 1 >= p.D ? 2u : 0u) | (h0 == 0 ? 4u : 0u) |      \
           (h0 + BH == p.H ? 8u : 0u) | (w0 == 0 ? 16u : 0u) | (w0 + BW == p.W ? 32u : 0u) | 64u;            \
    if (WB_ABL != 4 && ob + 1 < b_end) { /* advance to the next brick; the last one is repeated (its loads are unused) */ \
      ++ob;                                                                                                  \
      ow0 += BW; dyo += dsW; xo += xsW;                                                                      \
      if (ow0 == p.W) {                                                                                      \
        ow0 = 0; dyo -= dwW; xo -= xwW;                                                                      \
        if (p.order) {                                                                                       \
          od0 += BD; dyo += dsD; xo += xsD;                                                                  \
          if (od0 == p.D) {                                                                                  \
            od0 = 0; dyo -= dwD; xo -= xwD;                                                                  \
            oh0 += BH; dyo += dsH; xo += xsH;                                                                \
            if (oh0 == p.H) {                                                                                \
              oh0 = 0; dyo -= dwH; xo -= xwH;                                                                \
              ++on; dyo += dsN; xo += xsN;                                                                   \
            }                                                                                                \
          }                                                                                                  \
        } else {                                                                                             \
          oh0 += BH; dyo += dsH; xo += xsH;                                                                  \
          if (oh0 == p.H) {                                                                                  \
            oh0 = 0; dyo -= dwH; xo -= xwH;                                                                  \
            od0 += BD; dyo += dsD; xo += xsD;                                                                \
            if (od0 == p.D) {                                                                                \
              od0 = 0; dyo -= dwD; xo -= xwD;                                                                \
              ++on; dyo += dsN; xo += xsN;                                                                   \
            }                                                                                                \
          }                                                                                                  \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)
  // The request is issued from inline asm: behind the builtin hipcc waits vmcnt(0) in front of the next ds_read (it cannot prove
  // that the read does not alias the DMA's LDS destination), which would expose every request's latency.  Completion is counted
  // by hand: one s_waitcnt vmcnt(0) at the end of the brick, before the barrier that hands the buffer to the readers.
#define WB_DMA_PIECE(i_, buf_)                                                                               \
  do {                                                                                                       \
    constexpr int NDY_ = C::IMG_DY * DYP;                                                                    \
    if ((i_) < NDY_) {                                                                                       \
      const int m_ = ((i_) < NDY_ ? (i_) : 0) / DYP, k_ = ((i_) < NDY_ ? (i_) : 0) % DYP;                    \
      lds_dma16(dyb + dyoff[k_] + m_ * 128, lds_base + (uint32_t)((buf_)-smem) + m_ * DY_BYTES + wv * 1024 + k_ * (NT * 16)); \
    } else {                                                                                                 \
      const int q_ = (i_) < NDY_ ? 0 : (i_) - NDY_, m_ = q_ / XP, k_ = q_ % XP;                              \
      if (8 * wv + (NT / 8) * k_ < XROWS) { /* wave-uniform: the last round covers only the first XROWS rows */ \
        const bool ok = (xedge[k_] & xout) == 0 && jok[m_];                                                  \
        lds_dma16(ok ? xb + xoff[k_] + m_ * 128 : zpage,                                                     \
                  lds_base + (uint32_t)((buf_)-smem) + C::IMG_DY * DY_BYTES + m_ * X_BYTES + wv * 1024 + k_ * (NT * 16)); \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)

  // lane parts of the fragment addresses (see the brick loop)
  int abase[FA], xbase[6];
#pragma unroll
  for (int f = 0; f < FA; ++f) abase[f] = dy_off(8 * lg + jr, cobase + f * 16 + 4 * cq);
#pragma unroll
  for (int ci = 0; ci < 6; ++ci) {
    const int c = ci < 3 ? ci : ci + 1;   // R mod 8 is one of 0,1,2,4,5,6 (R = 12 a + b, b < 3)
    const int lrow = lg * XW + jr, col = cib * 16 + 4 * cq;
    xbase[ci] = lrow * 128 + ((((col >> 4) ^ ((c + lrow) >> 1)) & 3) << 5) + ((col & 15) << 1);
  }

  WB_T_DECL();
  if (b_beg < b_end) {
    WB_ORIGIN_NEXT();                                     // brick b_beg
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) WB_DMA_PIECE(i, smem);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int b = b_beg; b < b_end; ++b) {
    // this wave's images; K chunk kc of the brick = d plane kc >> 1, h half kc & 1; KSPLIT = 2: group g owns plane g (chunks 2g, 2g + 1)
    const char* dys = smem + ((b - b_beg) & 1) * BUF_BYTES + dyimg * DY_BYTES + (KSPLIT == 2 ? grp * (2 * 4096) : 0);
    const char* xs = smem + ((b - b_beg) & 1) * BUF_BYTES + C::IMG_DY * DY_BYTES + ximg * X_BYTES + (KSPLIT == 2 ? grp * (XH * XW * 128) : 0);
    char* nxt = smem + (((b - b_beg) & 1) ^ 1) * BUF_BYTES;
    const bool more = b + 1 < b_end;                      // block-uniform
    WB_ORIGIN_NEXT();                                     // brick b + 1: its pieces are requested during this one
    __builtin_amdgcn_sched_barrier(0);
    WB_T_BRICK();

    // 36 steps = 4 K-chunks x 9 taps, fully unrolled and SOFTWARE-PIPELINED: the x fragment of step s+1 (and the dy
    // fragments of the next K-chunk) are fetched two steps before the MFMAs that use them.  With one wave per SIMD nothing else hides
    // the LDS latency: the naive order (read, wait, 4 MFMAs) ran the matrix pipe at ~40 % inside this phase.
    // lane's voxels of K-chunk kc: v = 32*kc + 8*lg + 4*q + jr (q = 0,1 = the two transpose reads);
    // halo row of voxel (vd, vh, vw) at tap (kh, kw): (vd*XH + vh + kh)*XW + vw + kw.
    // Fragment addresses without per-read arithmetic: a read's row = (block-uniform, compile-time) R + the lane's row, and
    // the XOR swizzle only looks at row bits 1-2 (x) / voxel bits 1,3 (dy).  So the lane part -- including the swizzle for
    // each residue of R mod 8 -- is precomputed (abase[4], xbase[6], before the brick loop) and R * 128 goes into the
    // instruction's immediate offset.  One wave per SIMD issues in order: every VALU saved here is an issue slot for an MFMA.
#define WB_A(kc_, f_) tr_frag(dys + abase[f_] + (kc_)*4096, dys + abase[f_] + (kc_)*4096 + 512)
#define WB_XR(kc_, t_) ((((kc_) >> 1) * XH + ((kc_)&1) * 4 + (t_) / 3) * XW + (t_) % 3)
#define WB_XADDR(r_) (xs + xbase[((r_)&7) < 3 ? ((r_)&7) : ((r_)&7) - 1] + (r_)*128)
#define WB_B(kc_, t_) tr_frag(WB_XADDR(WB_XR(kc_, t_)), WB_XADDR(WB_XR(kc_, t_) + 4))
    constexpr int PF = WB_PF;  // the x fragment is fetched PF steps (4 PF MFMAs) ahead of its use, through a (PF + 1)-deep register ring
    bf16x8 fa[2][FA], fbr[PF + 1];
#pragma unroll
    for (int f = 0; f < FA; ++f) fa[0][f] = WB_A(0, f);
#pragma unroll
    for (int q = 0; q < PF; ++q) fbr[q] = WB_B(q / 9, q % 9);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int kc = st / 9, t = st % 9;
      WB_SETPRIO(st, NSTEP);
      if (st < NPIECE) {   // piece st of brick b+1 -> the other LDS buffer
#if WB_ABL != 1
        if (more) WB_DMA_PIECE(st < NPIECE ? st : 0, nxt);
#endif
      }
#if WB_ABL == 2
      if (st + PF < NSTEP && b == b_beg) fbr[(st + PF) % (PF + 1)] = WB_B((st + PF) / 9, (st + PF) % 9);
#else
      if (st + PF < NSTEP) fbr[(st + PF) % (PF + 1)] = WB_B((st + PF) / 9, (st + PF) % 9);
#endif
      if (t == 4 && kc < 4 / KSPLIT - 1) {
#pragma unroll
        for (int f = 0; f < FA; ++f) fa[(kc + 1) & 1][f] = WB_A(kc + 1, f);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads AHEAD of this step's MFMAs (hipcc sinks them next to their use otherwise)
#pragma unroll
      for (int f = 0; f < FA; ++f) acc[t][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kc & 1][f], fbr[st % (PF + 1)], acc[t][f], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (st == NPIECE - 1) WB_T_ISSUED();
    }
#undef WB_A
#undef WB_XR
#undef WB_XADDR
#undef WB_B

    __builtin_amdgcn_sched_barrier(0);
    WB_T_WAIT0();
#if WB_ABL != 3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WB_T_MID();
    __syncthreads();   // this brick's reads and the next brick's pieces are complete
#endif
    WB_T_WAIT1();
  }
  WB_T_END();
#undef WB_ORIGIN_NEXT
#undef WB_DMA_PIECE

  // ---- KSPLIT = 2: the two wave groups hold partial sums over different K chunks: group 1 hands its accumulators to group 0 through the
  //      (now idle) LDS brick buffers in two halves (5 + 4 taps: 80 KB of the 92 KB), so the block writes ONE partial slab ----
  if (KSPLIT == 2) {
    f32x4* xch = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int t0 = half * 5, t1 = half ? 9 : 5;
      __syncthreads();   // the brick loop's last reads / the previous half's reads are done
      if (grp == 1) {
#pragma unroll
        for (int t = t0; t < t1; ++t)
#pragma unroll
          for (int f = 0; f < FA; ++f) xch[(((t - t0) * FA + f) * 4 + wid) * 64 + lane] = acc[t][f];
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int t = t0; t < t1; ++t)
#pragma unroll
          for (int f = 0; f < FA; ++f) acc[t][f] += xch[(((t - t0) * FA + f) * 4 + wid) * 64 + lane];
      }
    }
    if (grp == 1) return;
  }
  // D[i][j]: lane holds i = 16 f + 4 lg + r, j = lane & 15 (of the wave's co / ci block)
  float* out = p.ws + (int64_t)split * (9 * p.nkd) * p.Cu * p.Cv;
  const int j = j0 + ximg * 64 + cib * 16 + (lane & 15);
  if (j < p.Cv) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float* ot = out + (int64_t)(p.nkd == 3 ? kd * p.td + (t / 3) * p.th + (t % 3) * p.tw : t) * p.Cu * p.Cv;
#pragma unroll
      for (int f = 0; f < FA; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[(int64_t)(i0 + dyimg * 64 + cobase + f * 16 + lg * 4 + r) * p.Cv + j] = acc[t][f][r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Composed up-conv, weight gradient, TWO PHASES PER BLOCK (upconv_fused.hip).
// A phase needs 2 x 2 x 2 of the 27 taps: in the 3x3x3 kernel's decomposition that is two kd-plane blocks per phase, each staging its own
// dy image and x halo for 4 useful taps -- 92 KB staged per phase, brick and 64 input channels.  Here a block owns the two phases that
// differ in the brick-w bit (they use the same kd and kh taps and kw = {0,1} / {1,2}): ONE x halo of three d planes (46 KB) serves both
// kd planes of both phases, wave group g multiplies phase g's dy image against it: 39 KB per phase.  8 waves: group g = phase, wave =
// 16-ci block; a wave holds 8 taps x 4 co fragments.  K chunk, step order, staging by LDS-DMA, transpose reads and swizzles as above.
// Slabs ws[split][27][8 * Co][Ci] (the taps of a phase at (p + q) per axis; everything else unwritten) -> wgrad_reduce_upc_kernel.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int XPL2 = BD + 1;                        // halo planes
constexpr int XR2 = XPL2 * XH * XW;                 // 360 rows
constexpr int X2_BYTES = XR2 * 128;                 // 45 KiB
constexpr int XP2 = (XR2 * 8 + NT - 1) / NT;        // 6 pieces per thread
constexpr int BUF2_BYTES = 2 * DY_BYTES + X2_BYTES; // 77 KiB per brick buffer
constexpr int NPIECE2 = 2 * DYP + XP2;              // 10 requests per brick and wave
constexpr int NSTEP2 = 32;                          // 4 K chunks x 8 taps

struct WBrick2Params {
  const bf16* dy;   // fine gradient [N][2D][2H][2W][Co]
  const bf16* x;    // coarse [N][D][H][W][Cv]
  float* ws;
  int N, D, H, W;   // extents of the BRICK axes (coarse)
  int Co, Cv;
  int nbricks, per_split, nsplit;
  int sd, sh, sw, td, th, tw;       // brick-axis strides (coarse voxels) and tap-index strides, as in WBrickParams
  int dsd, dsh, dsw;                // brick-axis strides in fine voxels
  int64_t dyn;                      // fine voxels per sample
  int phoff[8];                     // fine-voxel offset of phase (memory bit order d, h, w)
  int bd, bh, bw;                   // bit of the memory phase index that belongs to the brick d / h / w axis
  int order;
};

__global__ void __launch_bounds__(NT, 1) wgrad_brick_upc2_kernel(const WBrick2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // two brick buffers: [dy image phase 0][dy image phase 1][x halo, 3 planes]
  const int tid = threadIdx.x, lane = tid & 63, wid = (tid >> 6) & 3, grp = tid >> 8;
  const int lg = lane >> 4, jr = (lane & 15) >> 2, cq = lane & 3;
  const int ntj = (p.Cv + 63) / 64, nco = p.Co / 64, ntile = 4 * nco * ntj;
  // block -> (tile, split): the tiles of one split walk the same bricks -- ids congruent mod 8 (one XCD), consecutive there
  int member, split;
  if ((p.nsplit & 7) == 0) {
    const int k = blockIdx.x >> 3;
    member = k % ntile;
    split = (k / ntile) * 8 + (blockIdx.x & 7);
  } else {
    member = blockIdx.x % ntile;
    split = blockIdx.x / ntile;
  }
  const int tj = member % ntj, cot = (member / ntj) % nco, pp = member / (ntj * nco);
  const int pbd = pp >> 1, pbh = pp & 1;                            // phase bits along the brick d and h axes; group g = the brick w bit
  const int uph0 = (pbd << p.bd) | (pbh << p.bh), uph1 = uph0 | (1 << p.bw), uphg = grp ? uph1 : uph0;
  const int j0 = tj * 64, uco = cot * 64;
  const int cib = wid;
  const int b_beg = split * p.per_split;
  const int b_end = min(b_beg + p.per_split, p.nbricks);
  const int bw = p.W / BW, bh = p.H / BH, bd = p.D / BD;

  f32x4 acc[8][4];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[t][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pcp = tid & 7, srow = tid >> 3;
  const int key_dy = ((srow >> 1) & 1) | (((srow >> 3) & 1) << 1), key_x = (srow >> 1) & 3;
  const int pc_dy = ((((pcp >> 1) ^ key_dy) & 3) << 1) | (pcp & 1), pc = ((((pcp >> 1) ^ key_x) & 3) << 1) | (pcp & 1);
  const bool jok = (j0 + pc * 8) < p.Cv;
  const int xcol = j0 + pc * 8;
  uint32_t dyoff[DYP], xoff[XP2], xedge[XP2];
#pragma unroll
  for (int i = 0; i < DYP; ++i) {
    const int v = (tid >> 3) + (NT / 8) * i;
    dyoff[i] = (uint32_t)(((v >> 6) * p.dsd + ((v >> 3) & 7) * p.dsh + (v & 7) * p.dsw) * p.Co + pc_dy * 8) * 2u;
  }
#pragma unroll
  for (int i = 0; i < XP2; ++i) {
    const int r = (tid >> 3) + (NT / 8) * i;
    const int hd = r / (XH * XW), hh = (r / XW) % XH, hw = r % XW;
    const bool row_ok = r < XR2 && hw < BW + 2;
    xoff[i] = row_ok ? (uint32_t)((hd * p.sd + hh * p.sh + hw * p.sw) * p.Cv + xcol) * 2u : 0u;
    xedge[i] = (hd == 0 ? 1u : 0u) | (hd == XPL2 - 1 ? 2u : 0u) | (hh == 0 ? 4u : 0u) | (hh == XH - 1 ? 8u : 0u) | (hw == 0 ? 16u : 0u) |
               (hw == BW + 1 ? 32u : 0u) | (row_ok ? 0u : 64u);
  }
  const int dyimg1 = (p.phoff[uph1] - p.phoff[uph0]) * p.Co * 2;   // byte distance of phase 1's image from phase 0's
  const char* zpage = reinterpret_cast<const char*>(g_wb_zero);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const char* dyb = nullptr;
  const char* xb = nullptr;
  uint32_t xout = 0;
  int ob = b_beg, ow0, oh0, od0, on;
  {
    int t_ = b_beg;
    ow0 = (t_ % bw) * BW; t_ /= bw;
    if (p.order) {
      od0 = (t_ % bd) * BD; t_ /= bd;
      oh0 = (t_ % bh) * BH; t_ /= bh;
    } else {
      oh0 = (t_ % bh) * BH; t_ /= bh;
      od0 = (t_ % bd) * BD; t_ /= bd;
    }
    on = t_;
  }
  // carried byte offsets of the brick being loaded (see wgrad_brick_kernel): phase 0's first fine voxel in dy, the first halo voxel in x
  const int64_t cob = 2 * (int64_t)p.Co, cvb = 2 * (int64_t)p.Cv;
  const int64_t dsW = BW * p.dsw * cob, dsH = BH * p.dsh * cob, dsD = BD * p.dsd * cob;
  const int64_t dwW = (int64_t)p.W * p.dsw * cob, dwH = (int64_t)p.H * p.dsh * cob, dwD = (int64_t)p.D * p.dsd * cob, dsN = p.dyn * cob;
  const int64_t xsW = BW * p.sw * cvb, xsH = BH * p.sh * cvb, xsD = BD * p.sd * cvb;
  const int64_t xwW = (int64_t)p.W * p.sw * cvb, xwH = (int64_t)p.H * p.sh * cvb, xwD = (int64_t)p.D * p.sd * cvb, xsN = (int64_t)p.D * p.H * p.W * cvb;
  int64_t dyo = ((int64_t)on * p.dyn + (int64_t)od0 * p.dsd + (int64_t)oh0 * p.dsh + (int64_t)ow0 * p.dsw + p.phoff[uph0]) * cob + 2 * (int64_t)uco;
  // first halo voxel (d0 + pbd - 1, h0 - 1, w0 - 1): may lie outside the volume, never dereferenced then
  int64_t xo = ((int64_t)on * p.D * p.H * p.W + (int64_t)(od0 + pbd - 1) * p.sd + (int64_t)(oh0 - 1) * p.sh + (int64_t)(ow0 - 1) * p.sw) * cvb;
#define W2_ORIGIN_NEXT()                                                                                     \
  do {                                                                                                       \
    const int w0 = ow0, h0 = oh0, d0 = od0;                                                                  \
    dyb = reinterpret_cast<const char*>(p.dy) + dyo;                                                         \
    xb = reinterpret_cast<const char*>(p.x) + xo;                                                            \
    xout = (d0 + pbd - 1 < 0 ? 1u : 0u) | (d0 + pbd + 1 >= p.D ? 2u : 0u) | (h0 == 0 ? 4u : 0u) |            \
           (h0 + BH == p.H ? 8u : 0u) | (w0 == 0 ? 16u : 0u) | (w0 + BW == p.W ? 32u : 0u) | 64u;            \
    if (WB_ABL != 4 && ob + 1 < b_end) {                                                                     \
      ++ob;                                                                                                  \
      ow0 += BW; dyo += dsW; xo += xsW;                                                                      \
      if (ow0 == p.W) {                                                                                      \
        ow0 = 0; dyo -= dwW; xo -= xwW;                                                                      \
        if (p.order) {                                                                                       \
          od0 += BD; dyo += dsD; xo += xsD;                                                                  \
          if (od0 == p.D) {                                                                                  \
            od0 = 0; dyo -= dwD; xo -= xwD;                                                                  \
            oh0 += BH; dyo += dsH; xo += xsH;                                                                \
            if (oh0 == p.H) { oh0 = 0; dyo -= dwH; xo -= xwH; ++on; dyo += dsN; xo += xsN; }                 \
          }                                                                                                  \
        } else {                                                                                             \
          oh0 += BH; dyo += dsH; xo += xsH;                                                                  \
          if (oh0 == p.H) {                                                                                  \
            oh0 = 0; dyo -= dwH; xo -= xwH;                                                                  \
            od0 += BD; dyo += dsD; xo += xsD;                                                                \
            if (od0 == p.D) { od0 = 0; dyo -= dwD; xo -= xwD; ++on; dyo += dsN; xo += xsN; }                 \
          }                                                                                                  \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)
#define W2_DMA_PIECE(i_, buf_)                                                                               \
  do {                                                                                                       \
    if ((i_) < 2 * DYP) {                                                                                    \
      const int m_ = ((i_) < 2 * DYP ? (i_) : 0) / DYP, k_ = ((i_) < 2 * DYP ? (i_) : 0) % DYP;              \
      lds_dma16(dyb + dyoff[k_] + m_ * dyimg1, lds_base + (uint32_t)((buf_)-smem) + m_ * DY_BYTES + wv * 1024 + k_ * (NT * 16)); \
    } else {                                                                                                 \
      const int k_ = (i_) < 2 * DYP ? 0 : (i_) - 2 * DYP;                                                    \
      if (8 * wv + (NT / 8) * k_ < XR2) {                                                                    \
        const bool ok = (xedge[k_] & xout) == 0 && jok;                                                      \
        lds_dma16(ok ? xb + xoff[k_] : zpage, lds_base + (uint32_t)((buf_)-smem) + 2 * DY_BYTES + wv * 1024 + k_ * (NT * 16)); \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)

  // lane parts of the fragment addresses; the phase's in-plane tap origin (kh0 = pbh, kw0 = group) is a row offset folded into them
  const int offrt = pbh * XW + grp;
  int abase[4], xbase[6];
#pragma unroll
  for (int f = 0; f < 4; ++f) abase[f] = dy_off(8 * lg + jr, f * 16 + 4 * cq);
#pragma unroll
  for (int ci = 0; ci < 6; ++ci) {
    const int c = ci < 3 ? ci : ci + 1;
    const int lrow = lg * XW + jr + offrt, col = cib * 16 + 4 * cq;
    xbase[ci] = lrow * 128 + ((((col >> 4) ^ ((c + lrow) >> 1)) & 3) << 5) + ((col & 15) << 1);
  }

  WB_T_DECL();
  if (b_beg < b_end) {
    W2_ORIGIN_NEXT();
#pragma unroll
    for (int i = 0; i < NPIECE2; ++i) W2_DMA_PIECE(i, smem);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int b = b_beg; b < b_end; ++b) {
    const char* dys = smem + ((b - b_beg) & 1) * BUF2_BYTES + grp * DY_BYTES;
    const char* xs = smem + ((b - b_beg) & 1) * BUF2_BYTES + 2 * DY_BYTES;
    char* nxt = smem + (((b - b_beg) & 1) ^ 1) * BUF2_BYTES;
    const bool more = b + 1 < b_end;
    W2_ORIGIN_NEXT();
    __builtin_amdgcn_sched_barrier(0);
    WB_T_BRICK();
    // K chunk kc = brick d plane kc >> 1, h half kc & 1 (32 voxels); tap t8 = (kd, kh, kw) offsets from the phase's origin
#define W2_A(kc_, f_) tr_frag(dys + abase[f_] + (kc_)*4096, dys + abase[f_] + (kc_)*4096 + 512)
#define W2_XR(kc_, t_) (((((kc_) >> 1) + ((t_) >> 2)) * XH + ((kc_)&1) * 4 + (((t_) >> 1) & 1)) * XW + ((t_)&1))
#define W2_XADDR(r_) (xs + xbase[((r_)&7) < 3 ? ((r_)&7) : ((r_)&7) - 1] + (r_)*128)
#define W2_B(kc_, t_) tr_frag(W2_XADDR(W2_XR(kc_, t_)), W2_XADDR(W2_XR(kc_, t_) + 4))
    bf16x8 fa[2][4], fbr[2];
#pragma unroll
    for (int f = 0; f < 4; ++f) fa[0][f] = W2_A(0, f);
    fbr[0] = W2_B(0, 0);
#pragma unroll
    for (int st = 0; st < NSTEP2; ++st) {
      const int kc = st / 8, t = st % 8;
      WB_SETPRIO(st, NSTEP2);
      if (st < NPIECE2) {
        if (more) W2_DMA_PIECE(st < NPIECE2 ? st : 0, nxt);
      }
      if (st + 1 < NSTEP2) fbr[(st + 1) & 1] = W2_B((st + 1) / 8, (st + 1) % 8);
      if (t == 3 && kc < 3) {
#pragma unroll
        for (int f = 0; f < 4; ++f) fa[(kc + 1) & 1][f] = W2_A(kc + 1, f);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[t][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kc & 1][f], fbr[st & 1], acc[t][f], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (st == NPIECE2 - 1) WB_T_ISSUED();
    }
#undef W2_A
#undef W2_XR
#undef W2_XADDR
#undef W2_B
    __builtin_amdgcn_sched_barrier(0);
    WB_T_WAIT0();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WB_T_MID();
    __syncthreads();
    WB_T_WAIT1();
  }
  WB_T_END();
#undef W2_ORIGIN_NEXT
#undef W2_DMA_PIECE

  // D[i][j]: lane holds i = 16 f + 4 lg + r, j = lane & 15 of the wave's ci block; rows of phase uphg in the [27][8 * Co][Cv] slab
  float* out = p.ws + (int64_t)split * 27 * (8 * p.Co) * p.Cv;
  const int j = j0 + cib * 16 + (lane & 15);
  if (j < p.Cv) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int tap = (pbd + (t >> 2)) * p.td + (pbh + ((t >> 1) & 1)) * p.th + (grp + (t & 1)) * p.tw;
      float* ot = out + (int64_t)tap * (8 * p.Co) * p.Cv;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[(int64_t)(uphg * p.Co + uco + f * 16 + lg * 4 + r) * p.Cv + j] = acc[t][f][r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// ALL 27 TAPS PER BLOCK, 64 (co) x 32 (ci) tiles -- the layers with 64 output channels (down_tr64.ops.1 32 -> 64, down_tr128.ops.0 and
// up_tr64.ops.1 64 -> 64: a third of the weight-gradient FLOPs of a step, and its least efficient third).
// The one-kd-plane kernel above stages per brick a dy image (16 KB) and an x halo of two planes for ITS kd (30 KB) and multiplies 9 taps
// out of them: 80 bytes staged per MFMA with 64 x 64 tiles, 160 with the 64 x 32 tiles of the Ci = 32 layer (half of whose x image is
// padding) -- and staging, L2 -> LDS, is what this kernel spends its time on (1 840 vs 1 250 TFLOP/s without it).  With only 64 output
// channels there is no wider tile to amortise it over (128 x 64 needs Co % 128 == 0).  Here a block multiplies ALL 27 taps from one staging:
// the x halo is the brick's FOUR d planes x 32 channels, stored in the same 240-row x 128-byte image -- halo planes 0 / 1 in the rows'
// first 64 bytes, planes 2 / 3 in their second 64 bytes (logical "channel" = 32 * (plane >> 1) + ci), so the transpose-read layout, its
// swizzle and the LDS-DMA piece map are the ones above -- 46 KB per brick for 27 x 4 x 8 = 864 MFMAs: 53 bytes per MFMA.
// Eight waves = 2 ci blocks of 16 x 4 tap groups (taps t = g, g + 4, ...: 7, 7, 7, 6); a wave holds 7 taps x 4 co fragments (112
// accumulator registers), reads the 4 dy fragments of a K chunk once and one x fragment per tap: 28 steps of 4 MFMAs per brick.
// Slabs ws[split][27][Cu][Cv] as above -> the same second pass.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int NPIECE27 = DYP + XP;   // 6 requests per brick and wave
constexpr int NSTEP27 = 28;          // 4 K chunks x 7 taps
constexpr int BUF27_BYTES = DY_BYTES + X_BYTES;

struct WBrick27Params {
  const bf16* dy;   // [M][Cu]
  const bf16* x;    // [M][Cv]
  float* ws;        // [splits][27][Cu][Cv]
  int N, D, H, W;   // extents of the brick axes
  int Cu, Cv;
  int nbricks, per_split, nsplit;
  int sd, sh, sw, td, th, tw;
  int order;
};

__global__ void __launch_bounds__(NT, 1) wgrad_brick27_kernel(const WBrick27Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // two brick buffers: [dy image][x image (four planes x 32 ch)]
  const int tid = threadIdx.x, lane = tid & 63;
  const int lg = lane >> 4, jr = (lane & 15) >> 2, cq = lane & 3;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave 0 .. 7
  const int cib = wv & 1, tg = wv >> 1;                          // 16-ci block, tap group
  const int ntj = p.Cv / 32, ntile = (p.Cu / 64) * ntj;
  int member, split;   // tiles of one split walk the same bricks: ids congruent mod 8 (one XCD), consecutive there
  if ((p.nsplit & 7) == 0) {
    const int k = blockIdx.x >> 3;
    member = k % ntile;
    split = (k / ntile) * 8 + (blockIdx.x & 7);
  } else {
    member = blockIdx.x % ntile;
    split = blockIdx.x / ntile;
  }
  const int i0 = (member / ntj) * 64, j0 = (member % ntj) * 32;
  const int b_beg = split * p.per_split;
  const int b_end = min(b_beg + p.per_split, p.nbricks);
  const int bw = p.W / BW, bh = p.H / BH, bd = p.D / BD;

  f32x4 acc[7][4];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[t][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging roles (see wgrad_brick_kernel): lane -> row tid >> 3 (+ 64 per round), physical 16-byte position tid & 7; the swizzle is applied
  // to the source.  Logical piece pc of an x row: half = pc >> 2 selects the halo plane pair, pc & 3 the 8-channel group of the 32.
  const int pp = tid & 7, srow = tid >> 3;
  const int key_dy = ((srow >> 1) & 1) | (((srow >> 3) & 1) << 1), key_x = (srow >> 1) & 3;
  const int pc_dy = ((((pp >> 1) ^ key_dy) & 3) << 1) | (pp & 1), pc = ((((pp >> 1) ^ key_x) & 3) << 1) | (pp & 1);
  const int xhalf = pc >> 2, xcol = j0 + (pc & 3) * 8;
  uint32_t dyoff[DYP], xoff[XP], xedge[XP];
#pragma unroll
  for (int i = 0; i < DYP; ++i) {
    const int v = (tid >> 3) + (NT / 8) * i;
    dyoff[i] = (uint32_t)(((v >> 6) * p.sd + ((v >> 3) & 7) * p.sh + (v & 7) * p.sw) * p.Cu + pc_dy * 8) * 2u;
  }
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int r = (tid >> 3) + (NT / 8) * i;
    const int hd2 = r / (XH * XW), hh = (r / XW) % XH, hw = r % XW;
    const int pl = hd2 + 2 * xhalf;                         // halo plane 0 .. 3 (d0 - 1 .. d0 + 2)
    const bool row_ok = r < XROWS && hw < BW + 2;
    xoff[i] = row_ok ? (uint32_t)((pl * p.sd + hh * p.sh + hw * p.sw) * p.Cv + xcol) * 2u : 0u;
    xedge[i] = (pl == 0 ? 1u : 0u) | (pl == 3 ? 2u : 0u) | (hh == 0 ? 4u : 0u) | (hh == XH - 1 ? 8u : 0u) | (hw == 0 ? 16u : 0u) |
               (hw == BW + 1 ? 32u : 0u) | (row_ok ? 0u : 64u);
  }
  const char* zpage = reinterpret_cast<const char*>(g_wb_zero);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const char* dyb = nullptr;
  const char* xb = nullptr;
  uint32_t xout = 0;
  int ob = b_beg, ow0, oh0, od0, on;
  {
    int t_ = b_beg;
    ow0 = (t_ % bw) * BW; t_ /= bw;
    if (p.order) {
      od0 = (t_ % bd) * BD; t_ /= bd;
      oh0 = (t_ % bh) * BH; t_ /= bh;
    } else {
      oh0 = (t_ % bh) * BH; t_ /= bh;
      od0 = (t_ % bd) * BD; t_ /= bd;
    }
    on = t_;
  }
  // carried byte offsets of the brick being loaded (see wgrad_brick_kernel)
  const int64_t cub = 2 * (int64_t)p.Cu, cvb = 2 * (int64_t)p.Cv;
  const int64_t vsW = BW * p.sw, vsH = BH * p.sh, vsD = BD * p.sd, vwW = (int64_t)p.W * p.sw, vwH = (int64_t)p.H * p.sh, vwD = (int64_t)p.D * p.sd;
  const int64_t vsN = (int64_t)p.D * p.H * p.W;
  int64_t vo = (int64_t)on * p.D * p.H * p.W + (int64_t)od0 * p.sd + (int64_t)oh0 * p.sh + (int64_t)ow0 * p.sw;   // first voxel of the brick
  const char* const dy0_ = reinterpret_cast<const char*>(p.dy + i0);
  // first halo voxel (d0 - 1, h0 - 1, w0 - 1): may lie outside the volume, never dereferenced then
  const char* const x0_ = reinterpret_cast<const char*>(p.x) - ((int64_t)p.sd + p.sh + p.sw) * cvb;
#define W27_ORIGIN_NEXT()                                                                                    \
  do {                                                                                                       \
    const int w0 = ow0, h0 = oh0, d0 = od0;                                                                  \
    dyb = dy0_ + vo * cub;                                                                                   \
    xb = x0_ + vo * cvb;                                                                                     \
    xout = (d0 == 0 ? 1u : 0u) | (d0 + BD == p.D ? 2u : 0u) | (h0 == 0 ? 4u : 0u) | (h0 + BH == p.H ? 8u : 0u) | \
           (w0 == 0 ? 16u : 0u) | (w0 + BW == p.W ? 32u : 0u) | 64u;                                         \
    if (WB_ABL != 4 && ob + 1 < b_end) {                                                                     \
      ++ob;                                                                                                  \
      ow0 += BW; vo += vsW;                                                                                  \
      if (ow0 == p.W) {                                                                                      \
        ow0 = 0; vo -= vwW;                                                                                  \
        if (p.order) {                                                                                       \
          od0 += BD; vo += vsD;                                                                              \
          if (od0 == p.D) { od0 = 0; vo -= vwD; oh0 += BH; vo += vsH; if (oh0 == p.H) { oh0 = 0; vo -= vwH; ++on; vo += vsN; } } \
        } else {                                                                                             \
          oh0 += BH; vo += vsH;                                                                              \
          if (oh0 == p.H) { oh0 = 0; vo -= vwH; od0 += BD; vo += vsD; if (od0 == p.D) { od0 = 0; vo -= vwD; ++on; vo += vsN; } } \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)
#define W27_DMA_PIECE(i_, buf_)                                                                              \
  do {                                                                                                       \
    if ((i_) < DYP) {                                                                                        \
      const int k_ = (i_) < DYP ? (i_) : 0;                                                                  \
      lds_dma16(dyb + dyoff[k_], lds_base + (uint32_t)((buf_)-smem) + wv * 1024 + k_ * (NT * 16));           \
    } else {                                                                                                 \
      const int k_ = (i_) < DYP ? 0 : (i_) - DYP;                                                            \
      if (8 * wv + (NT / 8) * k_ < XROWS) {                                                                  \
        const bool ok = (xedge[k_] & xout) == 0;                                                             \
        lds_dma16(ok ? xb + xoff[k_] : zpage, lds_base + (uint32_t)((buf_)-smem) + DY_BYTES + wv * 1024 + k_ * (NT * 16)); \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)

  // lane parts of the fragment addresses.  dy as above.  x: a wave's 7 taps are t = tg + 4 k; tap (kd, kh, kw) of a voxel in brick d plane
  // vd reads halo plane vd + kd -- row block (plane & 1), row half (plane >> 1: logical column 32 * half + 16 * cib + 4 * cq) -- at row
  // R = ((plane & 1) * XH + kh) * XW + kw (+ 4 * XW for the upper h half of the plane: a multiple of 8 rows, i.e. an immediate that leaves
  // the swizzle alone).  The two transpose reads of a fragment (rows R and R + 4) get their complete lane address here, once: 28 registers
  // instead of a four-way copy of the step loop (the tap group is a run-time, wave-uniform value).
  int abase[4], xa0[2][7], xa1[2][7];
#pragma unroll
  for (int f = 0; f < 4; ++f) abase[f] = dy_off(8 * lg + jr, f * 16 + 4 * cq);
#pragma unroll
  for (int vd = 0; vd < 2; ++vd)
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int t = min(tg + 4 * k, 26);                    // group 3 has six taps: its seventh slot re-reads tap 26
      const int pl = vd + t / 9, R = ((pl & 1) * XH + (t / 3) % 3) * XW + t % 3;
      const int lrow = lg * XW + jr, col = (pl >> 1) * 32 + cib * 16 + 4 * cq;
      xa0[vd][k] = (R + lrow) * 128 + ((((col >> 4) ^ ((R + lrow) >> 1)) & 3) << 5) + ((col & 15) << 1);
      xa1[vd][k] = (R + 4 + lrow) * 128 + ((((col >> 4) ^ ((R + 4 + lrow) >> 1)) & 3) << 5) + ((col & 15) << 1);
    }
  // (group 3's seventh slot multiplies tap 26 once more into an accumulator that is never written: 4 of 112 MFMAs per K chunk of two of
  // the eight waves, cheaper than a branch around them -- a conditional MFMA cost 30 spilled registers)

  WB_T_DECL();
  if (b_beg < b_end) {
    W27_ORIGIN_NEXT();
#pragma unroll
    for (int i = 0; i < NPIECE27; ++i) W27_DMA_PIECE(i, smem);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // K chunk kc = brick d plane kc >> 1, h half kc & 1 (32 voxels); tap t = (kd, kh, kw) reads halo plane (kc >> 1) + kd:
  // row block (plane & 1), row half (plane >> 1)
#define W27_A(kc_, f_) tr_frag(dys + abase[f_] + (kc_)*4096, dys + abase[f_] + (kc_)*4096 + 512)
#define W27_B(kc_, k_) tr_frag(xs + xa0[(kc_) >> 1][k_] + ((kc_)&1) * (4 * XW * 128), xs + xa1[(kc_) >> 1][k_] + ((kc_)&1) * (4 * XW * 128))

  for (int b = b_beg; b < b_end; ++b) {
    const char* dys = smem + ((b - b_beg) & 1) * BUF27_BYTES;
    const char* xs = dys + DY_BYTES;
    char* nxt = smem + (((b - b_beg) & 1) ^ 1) * BUF27_BYTES;
    const bool more = b + 1 < b_end;
    W27_ORIGIN_NEXT();
    __builtin_amdgcn_sched_barrier(0);
    WB_T_BRICK();
    bf16x8 fa[2][4], fbr[2];
#pragma unroll
    for (int f = 0; f < 4; ++f) fa[0][f] = W27_A(0, f);
    fbr[0] = W27_B(0, 0);
#pragma unroll
    for (int st = 0; st < NSTEP27; ++st) {
      const int kc = st / 7, k = st % 7;
      WB_SETPRIO(st, NSTEP27);
      if (st < NPIECE27) {
#if WB_ABL != 1
        if (more) W27_DMA_PIECE(st < NPIECE27 ? st : 0, nxt);
#endif
      }
      if (st + 1 < NSTEP27) fbr[(st + 1) & 1] = W27_B((st + 1) / 7, (st + 1) % 7);
      if (k == 3 && kc < 3) {
#pragma unroll
        for (int f = 0; f < 4; ++f) fa[(kc + 1) & 1][f] = W27_A(kc + 1, f);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[k][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kc & 1][f], fbr[st & 1], acc[k][f], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (st == NPIECE27 - 1) WB_T_ISSUED();
    }
    __builtin_amdgcn_sched_barrier(0);
    WB_T_WAIT0();
#if WB_ABL != 3
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WB_T_MID();
    __syncthreads();
#endif
    WB_T_WAIT1();
  }
  WB_T_END();
#undef W27_A
#undef W27_B
#undef W27_ORIGIN_NEXT
#undef W27_DMA_PIECE

  // D[i][j]: lane holds i = 16 f + 4 lg + r (co), j = lane & 15 of the wave's ci block
  float* out = p.ws + (int64_t)split * 27 * p.Cu * p.Cv;
  const int j = j0 + cib * 16 + (lane & 15);
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const int t = tg + 4 * k;
    if (t < 27) {
      float* ot = out + (int64_t)((t / 9) * p.td + ((t / 3) % 3) * p.th + (t % 3) * p.tw) * p.Cu * p.Cv;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) ot[(int64_t)(i0 + f * 16 + lg * 4 + r) * p.Cv + j] = acc[k][f][r];
    }
  }
}

struct BrickSplit {
  int splits, per_split;
  int xcd_map, G, Q, gpc, ngroups, ntg, pair, blocks;   // co-located launch (see WBrickParams); blocks = grid size
  int cfg;                                              // tile shape: 0 = 64 x 64, 1 = 128 x 64, 2 = 64 x 128, 3 = 64 x 32
};
constexpr int TCO_OF[4] = {64, 128, 64, 64}, TCI_OF[4] = {64, 64, 128, 32};
std::atomic<int> g_wb_xcd{1};    // 0: the 2-D grid (a (tile, kd) block range per blockIdx.y)
std::atomic<int> g_wb_order{1};
std::atomic<int> g_wb_tiles{1};  // 0: 64 x 64 tiles only

// tile shape by staged bytes per MFMA (256 / TCI + 400 / TCO, see the head of the file)
int pick_cfg(int Cu, int Cv) {
  if (!g_wb_tiles) return 0;
  if (Cv == 32) return 3;
  if (Cu % 128 == 0) return 1;
  if (Cv % 128 == 0) return 2;
  return 0;
}

BrickSplit plan(int nbricks, int Cu, int Cv, int nkd = 3) {
  const int tiles = (Cu / 64) * ((Cv + 63) / 64) * nkd;
  // One block per CU (one wave per SIMD, 256 CUs): the grid should be a whole number of rounds of 256 blocks -- 384 blocks run
  // as long as 512.  Fewer rounds = fewer partial slabs for the second pass: take the smallest k <= 3 that fills >= 90 %.
  int splits = 1;
  for (int k = 1; k <= 3; ++k) {
    splits = 256 * k / tiles;
    if (splits >= 1 && tiles * splits * 10 >= 256 * k * 9) break;
  }
  if (splits < 1) splits = 1;
  if (splits > nbricks / 16) splits = nbricks / 16;   // at least 16 bricks per block (bounds the partial slabs)
  if (splits < 1) splits = 1;
  const int per = (nbricks + splits - 1) / splits;
  splits = (nbricks + per - 1) / per;
  return BrickSplit{splits, per, 0, 0, 0, 0, 0, 0, 0, splits * tiles, 0};
}

// Co-located plan (3x3x3 only).  Groups of G = 3 (kd) x gs (tiles) blocks share a brick range; gs = 2 pairs the two tiles that
// share the larger operand stream when the tile grid allows it.  A chunk of 256 ids holds 8 * Q co-located groups (Q = 32 / G per
// XCD: 30 of its 32 CUs) plus (256 - 8 Q G) / G groups of consecutive ids on the 16 CUs left over.
BrickSplit plan_xcd(int nbricks, int Cu, int Cv, int cfg) {
  const int TCO = TCO_OF[cfg], TCI = TCI_OF[cfg];
  const int ni = Cu / TCO, ntj = (Cv + TCI - 1) / TCI, ntiles = ni * ntj;
  // fabric bytes ~ 1.25 X ni + Y ntj without pairing (X = |x|, Y = |dy|, a block reads X / ntj and Y / ni); pairing over i halves the x
  // term, pairing over j the dy term
  const double X = 1.25 * Cv * ni, Y = 1.0 * Cu * ntj;
  int pair = 0;
  if (ni % 2 == 0 && ntj % 2 == 0) pair = X >= Y ? 2 : 1;
  else if (ni % 2 == 0) pair = 2;
  else if (ntj % 2 == 0) pair = 1;
  const int gs = pair ? 2 : 1, G = 3 * gs, Q = 32 / G, gpc = 8 * Q + (256 - 8 * Q * G) / G, ntg = ntiles / gs;
  int splits = 1, k = 1;
  for (k = 1; k <= 3; ++k) {
    splits = k * gpc / ntg;
    if (splits >= 1 && splits * ntg * 10 >= k * gpc * 9) break;
  }
  if (k > 3) k = 3;
  if (splits < 1) splits = 1;
  if (splits > nbricks / 16) splits = nbricks / 16;
  if (splits < 1) splits = 1;
  const int per = (nbricks + splits - 1) / splits;
  splits = (nbricks + per - 1) / per;
  const int ngroups = splits * ntg, chunks = (ngroups + gpc - 1) / gpc;
  return BrickSplit{splits, per, 1, G, Q, gpc, ngroups, ntg, pair, chunks * 256, cfg};
}

BrickSplit plan3(int nbricks, int Cu, int Cv) {
  if (!g_wb_xcd) return plan(nbricks, Cu, Cv);
  const int cfg = pick_cfg(Cu, Cv);
  const BrickSplit a = plan_xcd(nbricks, Cu, Cv, cfg);
  if (cfg == 0 || cfg == 3) return a;
  // a larger tile halves the number of blocks per brick range: on small volumes (the 16x16x8 level: 512 bricks, at least 16 per
  // block) that leaves CUs idle -- measured 874 -> 763 TFLOP/s on 128 -> 128 channels -- so keep 64 x 64 tiles there
  const BrickSplit b = plan_xcd(nbricks, Cu, Cv, 0);
  const int used_a = a.ngroups * a.G, used_b = b.ngroups * b.G;
  return (used_a < 230 && used_b > used_a) ? b : a;
}

template <int TCO, int TCI> void launch_cfg(dim3 grid, hipStream_t stream, const WBrickParams& p) {
  static std::once_flag attr_once;   // hipFuncSetAttribute once per process and instantiation, race-free
  constexpr int lds = 2 * WCfg<TCO, TCI>::BUF_BYTES;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_brick_kernel<TCO, TCI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  hipLaunchKernelGGL((wgrad_brick_kernel<TCO, TCI>), grid, dim3(NT), lds, stream, p);
}

}  // namespace

// ---- internal interface used by conv_wgrad.hip ---------------------------------------------------------------------
// natural orientation (W % 8 == 0), or the innermost extent as the 2-deep brick axis (W % 2 == 0, D % 8 == 0, H % 8 == 0: the 8 x 8 x 4 level)
static bool wb_natural(int D, int H, int W) { return D % BD == 0 && H % BH == 0 && W % BW == 0; }
static bool wb_permuted(int D, int H, int W) { return W % BD == 0 && D % BH == 0 && H % BW == 0; }
bool pcrl_wgrad_brick_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return dtype == PCRL_BF16 && (wb_natural(D, H, W) || wb_permuted(D, H, W)) && Co % 64 == 0 && Ci % 32 == 0 &&
         (int64_t)N * D * H * W / BV < (1 << 30) && (int64_t)N * D * H * W * (Ci > Co ? Ci : Co) < ((int64_t)1 << 31);
}
// 27-taps-per-block kernel (wgrad_brick27_kernel): layers with 64 output channels and 32 or 64 input channels.
// Measured (same box, isolated, b = 32): 32 -> 64 at 64x64x32  752 -> 965 TFLOP/s (0.617 -> 0.480 ms: the 64 x 32 tile of the one-plane kernel
// staged a half-empty x image for 9 taps); 64 -> 64 at 64x64x32  1 162 -> 1 198; 64 -> 64 at 32x32x16 (4 096 bricks)  895 -> 851: with Ci = 64
// the one-plane kernel is no longer bound by its staging (both forms sit at ~0.8 transpose reads per MFMA), and on small volumes the
// larger tile count of the one-plane form fills the chip better -- so: Ci = 32 always, Ci = 64 from 8 192 bricks on.
static bool wb27_on(int Ci, int Co, int nbricks) {
  return g_wb_tiles && Co == 64 && (Ci == 32 || (Ci == 64 && nbricks >= 8192));
}
static void wb27_plan(int nbricks, int Ci, int Co, int& splits, int& per) {
  const int ntile = (Co / 64) * (Ci / 32);
  int s = 256 / ntile;                       // one block per CU: one round of 256 blocks
  if (s > nbricks / 16) s = nbricks / 16;    // at least 16 bricks per block
  if (s < 1) s = 1;
  if (s >= 8) s &= ~7;
  per = (nbricks + s - 1) / s;
  const int used = (nbricks + per - 1) / per;
  splits = (s >= 8 && (used & 7)) ? s : used;   // keep the split count a multiple of 8 (the XCD map) unless the tail would be empty
  if (splits != used && (int64_t)(splits - 1) * per >= nbricks) splits = used;
}
int pcrl_wgrad_brick_splits(int N, int D, int H, int W, int Ci, int Co) {
  // the workspace must hold the partial slabs of every launch form
  const int nb = (int)((int64_t)N * D * H * W / BV);
  int m = plan(nb, Co, Ci).splits;
  for (int cfg = 0; cfg < 4; ++cfg) {
    if ((cfg == 1 && Co % 128) || (cfg == 2 && Ci % 128) || (cfg == 3 && Ci != 32)) continue;
    const int b = plan_xcd(nb, Co, Ci, cfg).splits;
    if (b > m) m = b;
  }
  if (Co == 64 && (Ci == 32 || Ci == 64)) {
    int s27, per27;
    wb27_plan(nb, Ci, Co, s27, per27);
    if (s27 > m) m = s27;
  }
  return m;
}
int pcrl_wgrad_brick_slabs(int N, int D, int H, int W, int Ci, int Co) {
  const int nb = (int)((int64_t)N * D * H * W / BV);
  if (wb27_on(Ci, Co, nb)) {
    int s27, per27;
    wb27_plan(nb, Ci, Co, s27, per27);
    return s27;
  }
  return plan3(nb, Co, Ci).splits;
}
void pcrl_wgrad_brick_set_xcd(int on, int order, int tiles) { g_wb_xcd = on; g_wb_order = order; g_wb_tiles = tiles; }
int pcrl_wgrad_brick_launch(const void* x, const void* dy, float* ws, int N, int D, int H, int W, int Ci, int Co,
                            hipStream_t stream) {
  const int nbricks = (int)((int64_t)N * D * H * W / BV);
  if (wb27_on(Ci, Co, nbricks)) {
    int splits, per;
    wb27_plan(nbricks, Ci, Co, splits, per);
    WBrick27Params q{(const bf16*)dy, (const bf16*)x, ws, N, D, H, W, Co, Ci, nbricks, per, splits, H * W, W, 1, 9, 3, 1, g_wb_order};
    if (!wb_natural(D, H, W)) {   // memory (D, H, W) -> brick axes (W, D, H)
      q.D = W; q.H = D; q.W = H;
      q.sd = 1; q.sh = H * W; q.sw = W;
      q.td = 1; q.th = 9; q.tw = 3;
    }
    static std::once_flag attr27;
    std::call_once(attr27, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_brick27_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF27_BYTES);
    });
    const int ntile = (Co / 64) * (Ci / 32);
    hipLaunchKernelGGL(wgrad_brick27_kernel, dim3((unsigned)(ntile * splits)), dim3(NT), 2 * BUF27_BYTES, stream, q);
    return pcrl_check_launch("wgrad_brick (27 taps per block)");
  }
  const BrickSplit sp = plan3(nbricks, Co, Ci);
  WBrickParams p{(const bf16*)dy, (const bf16*)x, ws, N, D, H, W, Co, Ci, nbricks, sp.per_split, 0, 3, H * W, W, 1, 9, 3, 1,
                 sp.xcd_map, g_wb_order, sp.G, sp.Q, sp.gpc, sp.ngroups, sp.ntg, sp.pair};
  if (!wb_natural(D, H, W)) {   // memory (D, H, W) -> brick axes (W, D, H); tap (kd', kh', kw') = (kw, kd, kh) -> index kh' * 9 + kw' * 3 + kd'
    p.D = W; p.H = D; p.W = H;
    p.sd = 1; p.sh = H * W; p.sw = W;
    p.td = 1; p.th = 9; p.tw = 3;
  }
  dim3 grid((unsigned)sp.splits, (unsigned)((Co / 64) * ((Ci + 63) / 64) * 3));
  if (sp.xcd_map) grid = dim3((unsigned)sp.blocks);
  switch (sp.cfg) {
    case 1: launch_cfg<128, 64>(grid, stream, p); break;
    case 2: launch_cfg<64, 128>(grid, stream, p); break;
    case 3: launch_cfg<64, 32>(grid, stream, p); break;
    default: launch_cfg<64, 64>(grid, stream, p); break;
  }
  return pcrl_check_launch("wgrad_brick");
}


// ---- composed up-conv (upconv_fused.hip): gradient of the composed weights on the COARSE grid (wgrad_brick_upc2_kernel) ----
// x: coarse [N][D][H][W][Ci]; dy0: fine [N][2D][2H][2W][Co]; slabs ws[split][27][8 * Co][Ci] (only a phase's 8 taps written)
bool pcrl_wgrad_brick_upc_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype) {
  return pcrl_wgrad_brick_eligible(N, D, H, W, Ci, 8 * Co, dtype) && Co % 64 == 0 && (int64_t)N * D * H * W * 8 * Co < ((int64_t)1 << 31);
}
// split plan of the two-phases-per-block kernel: 4 x (Co / 64) x ceil(Ci / 64) tiles per brick range; whole rounds of the eight XCDs
static void upc2_plan(int nbricks, int Ci, int Co, int& splits, int& per) {
  const int ntile = 4 * (Co / 64) * ((Ci + 63) / 64);
  int s = 512 / ntile;
  if (s < 1) s = 1;
  if (s > nbricks / 16) s = nbricks / 16;
  if (s < 1) s = 1;
  if (s >= 8) s &= ~7;
  per = (nbricks + s - 1) / s;
  const int used = (nbricks + per - 1) / per;
  splits = (s >= 8 && (used & 7)) ? s : used;
}
int pcrl_wgrad_brick_upc_slabs(int N, int D, int H, int W, int Ci, int Co) {
  int splits, per;
  upc2_plan((int)((int64_t)N * D * H * W / BV), Ci, Co, splits, per);
  return splits;
}
int pcrl_wgrad_brick_upc_launch(const void* x, const void* dy0, float* ws, int N, int D, int H, int W, int Ci, int Co, hipStream_t stream) {
  const int nbricks = (int)((int64_t)N * D * H * W / BV);
  int splits, per;
  upc2_plan(nbricks, Ci, Co, splits, per);
  WBrick2Params p{(const bf16*)dy0, (const bf16*)x, ws, N, D, H, W, Co, Ci, nbricks, per, splits, H * W, W, 1, 9, 3, 1, 8 * H * W, 4 * W, 2,
                  (int64_t)8 * D * H * W, {0, 0, 0, 0, 0, 0, 0, 0}, 2, 1, 0, g_wb_order};
  for (int ph = 0; ph < 8; ++ph) p.phoff[ph] = (((ph >> 2) & 1) * (2 * H) + ((ph >> 1) & 1)) * (2 * W) + (ph & 1);
  if (!wb_natural(D, H, W)) {   // memory (D, H, W) -> brick axes (W, D, H)
    p.D = W; p.H = D; p.W = H;
    p.sd = 1; p.sh = H * W; p.sw = W;
    p.td = 1; p.th = 9; p.tw = 3;
    p.dsd = 2; p.dsh = 8 * H * W; p.dsw = 4 * W;
    p.bd = 0; p.bh = 2; p.bw = 1;          // brick (d, h, w) = memory (w, d, h) = phase bits (0, 2, 1)
  }
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_brick_upc2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF2_BYTES);
  });
  const int ntile = 4 * (Co / 64) * ((Ci + 63) / 64);
  hipLaunchKernelGGL(wgrad_brick_upc2_kernel, dim3((unsigned)(ntile * splits)), dim3(NT), 2 * BUF2_BYTES, stream, p);
  return pcrl_check_launch("wgrad_brick (composed up-conv, two phases per block)");
}

// ---- 2D path: weight gradient of a 3x3 / stride 1 / pad 1 convolution over N images (N % 2 == 0): the image index is the depth ----
bool pcrl_wgrad_brick2d_eligible(int N, int H, int W, int Ci, int Co, int dtype) {
  return dtype == PCRL_BF16 && N % BD == 0 && H % BH == 0 && W % BW == 0 && Co % 64 == 0 && Ci % 32 == 0 && (int64_t)N * H * W / BV < (1 << 30) &&
         (int64_t)N * H * W * (Ci > Co ? Ci : Co) < ((int64_t)1 << 31);
}
int pcrl_wgrad_brick2d_splits(int N, int H, int W, int Ci, int Co) { return plan((int)((int64_t)N * H * W / BV), Co, Ci, 1).splits; }
int pcrl_wgrad_brick2d_launch(const void* x, const void* dy, float* ws, int N, int H, int W, int Ci, int Co, int up, hipStream_t stream) {
  const int nbricks = (int)((int64_t)N * H * W / BV);
  const BrickSplit sp = plan(nbricks, Co, Ci, 1);
  WBrickParams p{(const bf16*)dy, (const bf16*)x, ws, 1, N, H, W, Co, Ci, nbricks, sp.per_split, up, 1, H * W, W, 1, 9, 3, 1, 0, 0, 0, 0, 0, 0, 0, 0};
  dim3 grid((unsigned)sp.splits, (unsigned)((Co / 64) * ((Ci + 63) / 64)));
  launch_cfg<64, 64>(grid, stream, p);
  return pcrl_check_launch("wgrad_brick2d");
}

#if WB_TRACE
// probe builds only (tools/wgrad_trace.py): the 32 accounting words, read and cleared
extern "C" int pcrl_debug_wb_trace(unsigned long long* out32) {
  unsigned long long z[32] = {0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_wb_trace), sizeof(z)) != hipSuccess) return -2;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wb_trace), z, sizeof(z)) != hipSuccess) return -3;
  return 0;
}
#endif
