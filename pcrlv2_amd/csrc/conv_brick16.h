// (shared by conv_brick16.hip -- the plain instantiations -- and conv_brick16_upc.hip -- the composed ones: two translation units because they want
// different register-allocation flags, pcrlv2_amd/build.py FILE_FLAGS)
// Wide-brick LDS-halo implicit-GEMM 3x3x3 convolution for gfx950, bf16 -- the throughput path of pcrl_conv3d_k3_fwd (forward and
// data gradient) for volumes with D % 4 == 0, H % 8 == 0, W % 16 == 0 (the 64x64x32 and 32x32x16 levels and the 16^3 local views:
// 80 % of the convolution FLOPs of a step).  conv_brick.hip (4x8x8 bricks) keeps the narrower volumes and the 2D path.
//
// Replaces aten::convolution / convolution_backward(input) of LUConv.conv1 (models/pcrlv2_model_3d.py:9,33).
//
// Why a second brick kernel: conv_brick.hip sits against the chip's power cap with the matrix pipe ~55 % busy; what moves it
// is bytes moved per FLOP.  Here a block owns a 4 x 8 x 16 brick (512 output voxels) and a wave owns one d-plane of it, 128 voxels x BN
// output channels:
//   * LDS fragment reads per MFMA: 12 x ds_read_b128 per 32 MFMAs (8 A + 4 B) instead of 8 per 16                      (x 0.75)
//   * staged bytes per MFMA: the 27 weight tiles of a 32-channel chunk (108 KiB) serve 512 voxels instead of 256, the halo is
//     6 x 10 x 18 rows for 512 voxels (2.1x) instead of 6 x 10 x 10 for 256 (2.3x)                                    (x 0.6)
//   * the halo goes global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, no ds_write_b128 (the slowest LDS
//     instruction, 13 cycles per wave), no staging registers -- which is what lets a 128-accumulator wave keep two fragment sets in
//     ping-pong at two waves per SIMD.
// A-fragment rows are 16 CONSECUTIVE w positions of one (d, h) line, so the halo keeps its natural w pitch of 18 rows; the 16-byte slot
// of a row is XOR-ed with a key that depends only on the row's w position (the d, h tap offsets are multiples of the pitch, so a lane's
// three kw addresses are computed once): key = 2 for w in {4,5,10..15}, else 0 -- found by exhaustive search over gfx950's
// ds_read_b128 lane groups ({0-3,12-15,20-27}, ...) for both parities of the line index; conflict-free for every tap.  LDS-DMA writes
// lane-linear, so the swizzle is applied to the SOURCE address (lane (row, physical slot s) loads logical slot s ^ key(row)).
// Halo rows outside the volume are loaded from a 16-byte zero page.
//
// Measured (timing ablations, same box, 128 -> 64 channels at 64x64x32): without the halo requests after the first chunk -17 %, without the
// weight staging -13 %, without both -22 % (1 560 TFLOP/s); requesting the dead planes early changes nothing.  What is left on the
// table is bytes staged per MFMA, not latency.  (Also tried: weight fragments straight from global memory into registers, no weight stage
// in LDS and no per-stage barriers -- 16 x 64-byte segments per load instruction, L1 / address-path bound: -30 %.)
// The epilogue (128 two-byte stores per lane) costs 7 % (64 -> 64 channels) to 16 % (32 -> 64: one K chunk per block) by the same kind of
// ablation; storing 4-byte channel pairs after a lane-pair exchange (64 stores, 2 DPP moves and 6 selects per fragment) measured equal
// or 3 % worse, 8-byte quads through a transposed accumulator layout 4 % worse (spills): the stores stay as they are.
// Composed modes: sourcing the halo planes a phase / parity never reads (one of T + 2 per axis, 30 % of the rows) from the zero page instead of
// the tensor measured 4 % SLOWER (960 -> 915 TFLOP/s), and not requesting them at all (exec-masked lanes; rows outside the volume zeroed once per
// block instead of read from the zero page) 1-2 % slower than the plain request, in every mode: what the halo request costs is its issue
// (address arithmetic per piece at the chunk boundary), not the bytes -- the full halo is requested, branch-free.
//
// LDS: halo 68 KiB (single buffer: 1080 rows x 64 B, rounded up to whole 1 KiB DMA pieces) + one weight stage 12 KiB = 80 KiB -> two
// blocks (eight waves) per CU.  The next chunk's halo is requested as soon as every wave holds the last fragments of the current one
// (first barrier of the chunk's last stage) and lands under that stage's remaining MFMAs and the co-resident block's work.
#pragma once
#include "common.h"
#include <atomic>
#include <mutex>

namespace {

constexpr int TD = 4, TH = 8, TW = 16;
constexpr int HH = TH + 2, HP = TW + 2;                // halo extents in h, w; w pitch = natural 18
// NW = waves per block = d-planes per brick: 4 (4 x 8 x 16 bricks, two blocks per CU) or 8 (8 x 8 x 16 bricks, one block of eight waves per CU:
// the 27 weight tiles of a chunk serve 1024 voxels and the halo is 10 / 8 instead of 6 / 4 planes per output plane -- 33 instead of 52 staged
// bytes per MFMA; see pcrl_brick16_conv_launch for where it is used)
template <int NW> struct B16Geom {
  static constexpr int HD = NW + 2;
  static constexpr int ROWS = HD * HH * HP;            // 1080 / 1800 rows of 64 B
  static constexpr int NDMA = (ROWS + 15) / 16;        // 68 / 113 wave-instructions of 16 rows (1 KiB)
  static constexpr int HALO_BYTES = NDMA * 1024;       // 69632 / 115712
  static constexpr int PPW = (NDMA + NW - 1) / NW;     // pieces per wave: 17 / 15
};
constexpr int NS = 9;                                  // stages per chunk: (kd, kh); a stage = three kw taps


struct Brick16Params {
  const bf16* x;
  const bf16* w;      // packed [Nc][27][K]
  const float* bias;
  bf16* y;
  float* stats;       // [bricks][Nc][2] or null
  int N, D, H, W;
  int K, Nc;
  int ny;             // > 0: 1-D grid of bricks x ny ids (channel tiles of a brick on one XCD, see conv_brick.hip); 0: 2-D grid
  // UPCF instantiations only (forward of the composed ConvTranspose3d -> Conv3d operator, upconv_fused.hip): x is the COARSE tensor, the
  // Nc = 8 * upc output channels are 8 phases x upc channels, w the zero-embedded 3x3x3 weights [8 * upc][27][K] in which a phase holds
  // its 2 x 2 x 2 taps at (p + q) per axis.  A block (one 64-channel tile = one phase, or part of one) walks only the 4 of 9 (kd, kh) stages
  // its phase uses, each with the two kw taps in use (STAGE2: the unused tap is neither staged nor read; 900 -> 960 TFLOP/s), writes its voxels to the phase's FINE positions of y [N][2D][2H][2W][upc]
  // and adds bias_tab[border class of the fine voxel][channel]; the statistics rows are [bricks][8 * upc][2] = [bricks * 8][upc][2].
  int upc;
  const float* bias_tab;
  // MODE 2 (data gradient of the composed operator): x is the FINE gradient dy0 [N][2D][2H][2W][upc] read as its space-to-depth view on the
  // coarse grid, K = 8 * upc channels = 8 parities x upc (chunk c of 32 channels lies in parity c >> cshift, cshift = log2(upc / 32)); w the
  // zero-embedded 3x3x3 weights [Nc = Ci][27][8 * upc] in which parity `par` holds the taps e = 2 k - 1 + par per axis (dx[v] = sum_e
  // dy0[2 v + e - 1] wd[e]: parity 0 uses k in {1, 2}, parity 1 uses k in {0, 1}).  A chunk walks the 4 of 9 (kd, kh) stages and the two kw
  // taps of ITS parity; the output is an ordinary coarse tensor [N][D][H][W][Nc].
  int cshift;
  // BNR instantiations only (data gradient with the BatchNorm backward's first pass in the epilogue, conv_brick16_bnr.hip): the output is the gradient
  // of the activation a = relu(scale * bn_y + shift) of the layer below; `stats` rows receive (sum dz, sum dz * xhat) with dz = [scale * bn_y + shift > 0]
  // * (this output, rounded to bf16 as stored), xhat = (bn_y - mean) * rstd -- what bn_bwd_reduce_kernel (norm_pool.hip) computes from a second read of
  // the stored gradient.  bn_y: [N][D][H][W][Nc] bf16, the layout of y; bn_scale / bn_shift / bn_mean / bn_rstd: Nc floats each (pcrl_bn_finalize's outputs).
  const bf16* bn_y;
  const float *bn_scale, *bn_shift, *bn_mean, *bn_rstd;
};

__device__ __forceinline__ int key_w(int hw) { return ((0xFC30 >> hw) & 1) << 1; }   // hw in [0, 18)
// weight tile [BN co][32 k]: fragment reads take 16 consecutive rows from a 16-aligned row (same swizzle as conv_brick.hip)
__device__ __forceinline__ int woff(int row, int slot) {
  const int key = (0x78 >> (((row >> 2) & 3) * 2)) & 3;
  return row * 64 + ((slot ^ key) << 4);
}

// One LDS-DMA request of the halo: the lanes whose bit 0 of `okmask` is set load 16 bytes from `base` + `voff` (SGPR base + 32-bit
// unsigned lane offset) to LDS byte address `lds_dst` (wave-uniform) + 16 * lane; the other lanes request nothing (their LDS rows were
// zeroed once per block).  EXEC is narrowed by v_cmpx and restored from `exec_all` inside the statement; M0 carries the destination.
// Issued from inline asm because hipcc waits vmcnt(0) before the next barrier / LDS access behind the builtin (it cannot prove they do
// not touch the DMA's destination), which would expose the latency of the requests; completion is counted by hand (s_waitcnt vmcnt(0)
// at the end of the chunk; the compiler's own counted waits for the weight loads can only over-wait, vmcnt retires in order).
// `okmask` and `chain` pass through as read-write operands: the address arithmetic of the NEXT piece (which starts from them) cannot be scheduled
// in front of this request, so a wave never holds more than one piece's temporaries (17 pieces' worth spilled 47 registers).
__device__ __forceinline__ void lds_dma16_masked(uint64_t base, uint32_t voff, uint32_t& okmask, uint32_t& chain, uint32_t lds_dst, uint64_t exec_all) {
  uint32_t t;
  asm volatile("v_and_b32 %0, 1, %1\n\tv_cmpx_ne_u32_e32 0, %0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\ts_mov_b64 exec, %6"
               : "=&v"(t), "+v"(okmask), "+v"(chain)
               : "v"(voff), "s"(lds_dst), "s"(base), "s"(exec_all)
               : "memory", "vcc");
}

// MODE 0: 3x3x3 convolution; 1: composed up-conv forward; 2: composed up-conv data gradient (Brick16Params); 3: 3x3 convolution of the 2D path -- the image
// index plays the role of depth (a brick = 4 images x 8 x 16 pixels), only the three kd = 1 stages are walked, the two d planes of the halo no tap reads
// are not requested, weights are packed [Nc][9][K] (round 5: the 2D step's stride-1 layers on the LDS-DMA kernel instead of conv_brick.hip's register staging).
// PERM 1: the brick's (d, h, w) axes run along the volume's (D, W, H) -- p.D, p.H, p.W are then the extents along the BRICK axes (D, W, H of the
// volume) -- for volumes whose H, not W, is a multiple of 16 (the 16 x 16 x 8 level).  A convolution commutes with a permutation of the axes
// applied to volume, taps and phases alike: only the voxel index (VOX / FVOX), the tap number of a weight row (WTAP), the phase / parity bit of
// an axis (BITH / BITW) and the border class (CLS) know the difference; rows still go through LDS one 64-byte slice per voxel.
template <int BN, int MODE = 0, int PERM = 0, int NW = 4, bool BNR = false>
__global__ void __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) brick16_conv_kernel(const Brick16Params p) {
  static_assert(!BNR || ((MODE == 0 || MODE == 3) && NW == 4), "the BatchNorm-reduce epilogue exists for the plain 4-plane data gradient only (3D, and the 2D path's 3x3 form)");
  using G = B16Geom<NW>;
  constexpr int ROWS = G::ROWS, NDMA = G::NDMA, HALO_BYTES = G::HALO_BYTES, HD = G::HD;
  constexpr int FN = BN / 16;
  constexpr bool UPCF = MODE == 1, UPCD = MODE == 2, P2D = MODE == 3;
  constexpr int NSK = P2D ? 3 : (MODE ? 4 : NS);   // stages per chunk
  constexpr int TAPS = P2D ? 9 : 27;               // taps per packed weight row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wbuf = smem + HALO_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int K = p.K, nchunk = K / 32;

  // ---- brick origin (XCD-contiguous brick ranges; the channel tiles of a brick adjacent on one XCD) ----
  const int bw = p.W / TW, bh = p.H / TH, bd = p.D / NW;
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
#define BITD(x_) (((x_) >> 2) & 1)
#define BITH(x_) (PERM ? ((x_)&1) : (((x_) >> 1) & 1))      /* bit of the volume axis the brick's h axis runs along */
#define BITW(x_) (PERM ? (((x_) >> 1) & 1) : ((x_)&1))
#define VOX(n_, d_, h_, w_) (PERM ? (((int64_t)(n_)*p.D + (d_)) * p.W + (w_)) * p.H + (h_) : (((int64_t)(n_)*p.D + (d_)) * p.H + (h_)) * p.W + (w_))
#define FVOX(n_, d_, h_, w_) (PERM ? (((int64_t)(n_) * (2 * p.D) + (d_)) * (2 * p.W) + (w_)) * (2 * p.H) + (h_) \
                                   : (((int64_t)(n_) * (2 * p.D) + (d_)) * (2 * p.H) + (h_)) * (2 * p.W) + (w_))
#define WTAP(s9_, j_) ((PERM ? ((s9_) / 3) * 9 + (j_)*3 + (s9_) % 3 : (s9_)*3 + (j_)) - (MODE == 3 ? 9 : 0))   /* (kd, kh, kw) of the brick -> tap number of the volume (2D: kd = 1, 9-tap rows) */
  const int uph = UPCF ? n0 / p.upc : 0, ukd0 = BITD(uph), ukh0 = BITH(uph);
  // stage number -> (kd * 3 + kh).  UPCF: the block's phase uses k = p, p + 1 per axis; UPCD: chunk c's parity uses k = 1 - par, 2 - par.
#define PARC(c_) ((c_) >> p.cshift)
#define SID(c_, s_) (MODE == 0 ? (s_) : MODE == 3 ? 3 + (s_) : MODE == 1 ? ((ukd0 + ((s_) >> 1)) * 3 + ukh0 + ((s_)&1)) \
                                                 : ((1 - BITD(PARC(c_)) + ((s_) >> 1)) * 3 + 1 - BITH(PARC(c_)) + ((s_)&1)))
#define KWB(c_) (MODE == 0 || MODE == 3 ? 0 : MODE == 1 ? BITW(uph) : 1 - BITW(PARC(c_)))   /* first of the kw taps in use (composed modes: two, the third tap's weights are zero) */
  const int brick_lin = b;   // NW == 4: the statistics row of this brick (kept instead of re-derived in the epilogue: seven fewer live scalars)
  const int w0 = (b % bw) * TW; b /= bw;
  const int h0 = (b % bh) * TH; b /= bh;
  const int d0 = (b % bd) * NW; b /= bd;
  const int n = b;

  // ---- halo DMA: wave `wid` issues pieces wid, wid + NW, ...; a piece = 16 rows (1 KiB), lane -> (row = piece * 16 + lane / 4, physical slot
  //      lane & 3).  Round 5 (profiles/r05_b16_isa_mix.txt): the per-piece source address used to be recomputed from the row number at every
  //      chunk boundary -- two divisions, three range checks, a 64-bit voxel index and a branch around it: ~65 instructions per piece with
  //      ten quarter-rate multiplies, 17 pieces per wave, i.e. ~1 100 instructions (~6 000 cycles) per chunk in front of the chunk's 864 (plain)
  //      or 256 (composed) MFMAs per wave.  Now the block's PLAN is made once: the source is (SGPR base of the halo's first voxel) + a 32-bit
  //      lane offset that advances from piece to piece by C1 + wrap * C2 + plane * C3 (a piece stride is 16 * NW rows = QL lines + a few rows of
  //      the 18-row line pitch: `wrap` = the lane's row crossed one more line end, `plane` = its line crossed a plane end), the two bits per
  //      piece and lane, the swizzle key of the row and "row inside the volume" live in three mask registers, rows outside the volume are
  //      zeroed ONCE and never requested (EXEC-masked request), and a chunk only adds its channel offset: ~14 instructions per piece. ----
  constexpr int PPW = G::PPW;
  constexpr int RSTEP = 16 * NW, QL = RSTEP / HP;                 // rows / whole lines between two pieces of a wave
  static_assert(PPW <= 17 && RSTEP - QL * HP < HP, "plan masks hold 16 steps; at most one line wrap per step");
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // byte strides of the source along the brick's w (row), h (line), d (plane) axes; the byte offset of chunk c_
  int RS, LS, PS;
  if (UPCD) {   // the fine gradient read as its space-to-depth view on the coarse grid: coarse steps are two fine voxels
    const int v2 = 4 * p.upc;
    RS = PERM ? v2 * (2 * p.H) : v2;
    LS = PERM ? v2 : v2 * (2 * p.W);
    PS = v2 * (2 * p.H) * (2 * p.W);
  } else {
    RS = PERM ? 2 * K * p.H : 2 * K;
    LS = PERM ? 2 * K : 2 * K * p.W;
    PS = 2 * K * p.H * p.W;
  }
  const int C1 = QL * LS + (RSTEP - QL * HP) * RS, C2 = LS - HP * RS, C3 = PS - HH * LS;   // |C2|, |C3| < 2^23 (eligibility): 24-bit multiply-adds
#define CHUNK_OFS(c_) (UPCD ? (uint32_t)(((BITD(PARC(c_)) * (2 * p.H) * (2 * p.W) + (PERM ? BITW(PARC(c_)) * (2 * p.H) + BITH(PARC(c_)) \
                                                                                       : BITH(PARC(c_)) * (2 * p.W) + BITW(PARC(c_)))) * p.upc \
                                          + ((c_) - (PARC(c_) << p.cshift)) * 32) * 2)                                               \
                            : (uint32_t)(c_) * 64u)
  // SGPR base: the halo's first voxel (d0 - 1, h0 - 1, w0 - 1) of sample n -- may lie in front of the tensor; only rows inside the volume are requested
  const uint64_t hbase = (uint64_t)(uintptr_t)p.x + (UPCD ? (uint64_t)(FVOX(n, 2 * (d0 - 1), 2 * (h0 - 1), 2 * (w0 - 1)) * (int64_t)(2 * p.upc))
                                                        : (uint64_t)(VOX(n, d0 - 1, h0 - 1, w0 - 1) * (int64_t)(2 * K)));
  // the plan: hoff0 = piece 0's offset (bit 0: piece 0 inside the volume -- offsets are multiples of 16); m_step: bit i - 1 = wrap, bit 15 + i =
  // plane of the step into piece i; m_ko: bit i - 1 = piece i inside the volume, bit 15 + i = key of piece i XOR key of piece 0
  uint32_t hoff0 = 0, m_step = 0, m_ko = 0;
  {
    int prevL = 0, prevP = 0, key0 = 0;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wid + NW * i, row = piece * 16 + (lane >> 2);
      const int L = row / HP, hw = row - L * HP, P = L / HH, hh = L - P * HH;
      const int d = d0 + P - 1, h = h0 + hh - 1, w = w0 + hw - 1;
      const bool inside = row < ROWS && (unsigned)d < (unsigned)p.D && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
      // composed forward: the block's phase reads taps k = bit, bit + 1 per axis -- one of the T + 2 planes / lines / columns of the halo is never read
      // (29 % of the rows): not requested.  (Round 4 measured this 1-2 % SLOWER -- with the old per-piece address arithmetic; with the plan it is a mask bit.)
      const bool used = P2D ? (P != 0 && P != HD - 1)      /* 2D: no tap reaches into the neighbouring images */
                            : (!UPCF || (P != (ukd0 ? 0 : HD - 1) && hh != (ukh0 ? 0 : HH - 1) && hw != (BITW(uph) ? 0 : HP - 1)));
      const bool ok = inside && used;
      const int key = key_w(hw) >> 1;
      if (i == 0) {
        hoff0 = (uint32_t)(P * PS + hh * LS + hw * RS + (((lane & 3) ^ (key << 1)) << 4)) | (ok ? 1u : 0u);
        key0 = key;
      } else {
        m_step |= (uint32_t)(L - prevL - QL) << (i - 1);
        m_step |= (uint32_t)(P - prevP) << (15 + i);
        m_ko |= (ok ? 1u : 0u) << (i - 1);
        m_ko |= (uint32_t)(key ^ key0) << (15 + i);
      }
      prevL = L;
      prevP = P;
      // rows of the halo outside the volume: zero for the block's lifetime
      // (rows >= ROWS of the last piece are never read -- and, for NW = 4, that tail of the buffer holds every wave's plan words: not touched)
      if (!inside && row < ROWS) *reinterpret_cast<u32x4*>(halo + piece * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    }
  }
  // The two mask words depend on (wave, lane / 4) only and are needed once per chunk: they wait in LDS -- NW = 4: the 512 bytes behind the halo's
  // last row (1080 rows of the 1088 its 68 pieces span: the request of rows 1080.. is masked off); NW = 8: behind the weight stage -- instead of
  // occupying two registers across the stage loop (the allocator spilled fragment addresses for them: three scratch reloads per two stages).
  uint2* plan_lds = reinterpret_cast<uint2*>(smem + (NW == 4 ? ROWS * 64 : HALO_BYTES + 3 * BN * 64)) + wid * 16 + (lane >> 2);
  *plan_lds = uint2{m_step, m_ko};
  asm volatile("" : "+v"(hoff0));   // nothing of the plan is re-derived inside the loop
  const uint64_t exec_all = __builtin_amdgcn_read_exec();
  // The pieces are a ROLLED loop (scalar counter, variable shifts): 17 unrolled copies per stage body made the stage one scheduling region of
  // ~1 000 instructions, and the allocator spilled the fragment addresses of the stage loop for it.
#define DMA_HALO(c_)                                                                                       \
  do {                                                                                                     \
    uint32_t hoff = (hoff0 & ~15u) + CHUNK_OFS(c_), m0keep;                                                \
    const uint2 plan_ = *plan_lds;                                                                         \
    uint32_t m_step = plan_.x, m_ko = plan_.y;   /* shifted right by one per piece: the bits of the next step sit at positions 0 and 16 */ \
    asm volatile("s_mov_b32 %0, m0" : "=s"(m0keep), "+v"(m_step), "+v"(m_ko));                             \
    uint32_t dst = lds_base + wid * 1024;                                                                  \
    lds_dma16_masked(hbase, hoff, hoff0, m_step, dst, exec_all);                                           \
    _Pragma("unroll 1") for (int i = 1; i < PPW; ++i) {                                                    \
      const int t1 = __builtin_amdgcn_sbfe(m_step, 0, 1), t2 = __builtin_amdgcn_sbfe(m_step, 16, 1);  /* 0 / -1 */ \
      hoff += (uint32_t)((t1 & C2) + (t2 & C3) + C1);                                                      \
      const uint32_t ha = hoff ^ ((m_ko >> 11) & 32u);                                                     \
      dst += NW * 1024;                                                                                    \
      lds_dma16_masked(hbase, ha, m_ko, m_step, dst, exec_all);                                            \
      m_step >>= 1;                                                                                        \
      m_ko >>= 1;                                                                                          \
    }                                                                                                      \
    asm volatile("s_mov_b32 m0, %0" ::"s"(m0keep));                                                        \
  } while (0)

  // ---- weight staging through registers: 3 pieces per thread (taps kw = 0,1,2 of the stage), row co = tid >> 2, slot tid & 3 ----
  const bool wthread = NW * 16 <= BN || (tid >> 2) < BN;   // compile-time true for four waves x 64 channels: no exec mask around the weight stores
  const uint32_t wlane = (uint32_t)(((n0 + (wthread ? (tid >> 2) : 0)) * TAPS) * K + (tid & 3) * 8) * 2u;   // byte offset of the thread's weight row (weights < 4 GiB: eligibility)
  const int wdst = woff(tid >> 2, tid & 3);
  u32x4 rw[3];
  constexpr int NKW = (UPCF || UPCD) ? 2 : 3;   // kw taps per stage: the composed modes stage and multiply only the two taps their phase / parity uses
#define LOAD_W(c_, s9_, kwb_) /* taps kwb_ .. kwb_ + NKW - 1 of stage (kd, kh) */                          \
  do {                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < NKW; ++j)                                                        \
      rw[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.w) + (int64_t)(WTAP(s9_, (kwb_) + j) * K + (c_)*32) * 2 + wlane); \
  } while (0)
#define STORE_W()                                                                                          \
  do {                                                                                                     \
    if (wthread) {                                                                                         \
      _Pragma("unroll") for (int j = 0; j < NKW; ++j)                                                      \
        *reinterpret_cast<u32x4*>(wbuf + j * (BN * 64) + wdst) = rw[j];                                    \
    }                                                                                                      \
  } while (0)

  f32x4 acc[8][FN];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Fragment addressing: wave = d-plane `wid`; fragment fm = h line fm, rows = 16 consecutive w.  Row of tap (kd,kh,kw):
  // ((wid + kd) * HH + fm + kh) * HP + lr + kw; the key depends on lr + kw only -> three lane offsets, (kd,kh) and fm are adds.
  int akw[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) akw[kw] = ((wid * HH) * HP + lr + kw) * 64 + ((lg ^ key_w(lr + kw)) << 4);
  const int bofs = woff(lr, lg);

  bf16x8 fa[2][4], fb[2][FN];
#define LOADA(S_, aoff_, half_)                                                                            \
  do {                                                                                                     \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                          \
      fa[S_][q] = *reinterpret_cast<const bf16x8*>(halo + (aoff_) + ((half_)*4 + q) * (HP * 64));          \
  } while (0)
#define LOADB(S_, wt_)                                                                                     \
  do {                                                                                                     \
    _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                         \
      fb[S_][j] = *reinterpret_cast<const bf16x8*>((wt_) + bofs + j * 1024);                               \
  } while (0)
#define MFMA_HALF(SA_, SB_, half_, Q0_, Q1_)                                                               \
  do {                                                                                                     \
    _Pragma("unroll") for (int q = (Q0_); q < (Q1_); ++q)                                                  \
      _Pragma("unroll") for (int j = 0; j < FN; ++j)                                                       \
        acc[(half_)*4 + q][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[SA_][q], fb[SB_][j], acc[(half_)*4 + q][j], 0, 0, 0); \
  } while (0)
#define PIPE_READS(n_, m_)                                                                                 \
  do {                                                                                                     \
    _Pragma("unroll") for (int q = 0; q < (n_); ++q) {                                                     \
      __builtin_amdgcn_sched_group_barrier(0x008, (m_), 0);                                                \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                   \
    }                                                                                                      \
  } while (0)
#define PIPE_WRITES(n_, m_)                                                                                \
  do {                                                                                                     \
    _Pragma("unroll") for (int q = 0; q < (n_); ++q) {                                                     \
      __builtin_amdgcn_sched_group_barrier(0x008, (m_), 0);                                                \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                                   \
    }                                                                                                      \
  } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

  // One stage = six half-taps (kw = 0,1,2 x voxel halves 0,1), 16 MFMAs each.  A sets alternate per half-tap (fa[0] = half 0,
  // fa[1] = half 1); B sets alternate per tap: P = the set that holds kw = 0 on entry (a stage has three taps, so P flips per stage).
  // The two barriers of a stage sit inside the LAST half-tap, whose operands are in registers: the next stage's weights are stored,
  // the next chunk's halo requested and the first fragments of the next stage read under its MFMAs.
  int c = 0, s9 = 0;
#define STAGE(P_)                                                                                          \
  do {                                                                                                     \
    int cn = c, sn = s9 + 1;                                                                               \
    if (sn == NSK) { sn = 0; cn = c + 1; }                                                                 \
    const bool last = (cn == nchunk);                                                                      \
    if (last) { cn = c; sn = s9; }                                                                         \
    LOAD_W(cn, SID(cn, sn), 0);                                                        \
    const bool halo_next = (s9 == NSK - 1) && !last; /* block-uniform */                                   \
    const int tap64 = ((SID(c, s9) / 3) * HH + (SID(c, s9) % 3)) * (HP * 64);                              \
    const int ntap64 = ((SID(cn, sn) / 3) * HH + (SID(cn, sn) % 3)) * (HP * 64);                           \
    /* ht0: kw 0, half 0 */                                                                                \
    LOADA(1, akw[0] + tap64, 1);                                                                           \
    MFMA_HALF(0, P_, 0, 0, 4);                                                                 \
    PIPE_READS(4, FN);                                                                                     \
    SB();                                                                                                  \
    /* ht1: kw 0, half 1 */                                                                                \
    LOADA(0, akw[1] + tap64, 0);                                                                           \
    LOADB((P_) ^ 1, wbuf + 1 * (BN * 64));                                                                 \
    MFMA_HALF(1, P_, 1, 0, 4);                                                                 \
    PIPE_READS(4 + FN, (4 * FN) / (4 + FN));                                                                     \
    SB();                                                                                                  \
    /* ht2: kw 1, half 0 */                                                                                \
    LOADA(1, akw[1] + tap64, 1);                                                                           \
    MFMA_HALF(0, (P_) ^ 1, 0, 0, 4);                                                                       \
    PIPE_READS(4, FN);                                                                                     \
    SB();                                                                                                  \
    /* ht3: kw 1, half 1 */                                                                                \
    LOADA(0, akw[2] + tap64, 0);                                                                           \
    LOADB(P_, wbuf + 2 * (BN * 64));                                                                       \
    MFMA_HALF(1, (P_) ^ 1, 1, 0, 4);                                                                       \
    PIPE_READS(4 + FN, (4 * FN) / (4 + FN));                                                                     \
    SB();                                                                                                  \
    /* ht4: kw 2, half 0 */                                                                                \
    LOADA(1, akw[2] + tap64, 1);                                                                           \
    MFMA_HALF(0, P_, 0, 0, 4);                                                                 \
    PIPE_READS(4, FN);                                                                                     \
    SB();                                                                                                  \
    /* ht5: kw 2, half 1 -- every LDS read of this stage (and, in a chunk's last stage, of this chunk's halo) is complete */ \
    __syncthreads();                                                                                       \
    STORE_W();                                                                         \
    if (halo_next) DMA_HALO(c + 1);                                                                        \
    MFMA_HALF(1, P_, 1, 0, 2);                                                                 \
    PIPE_WRITES(3, FN / 2);                                                                                     \
    SB();                                                                                                  \
    if (halo_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
    __syncthreads();                                                                                       \
    LOADA(0, akw[0] + ntap64, 0);                                                                          \
    LOADB((P_) ^ 1, wbuf);                                                                                 \
    MFMA_HALF(1, P_, 1, 2, 4);                                                                 \
    PIPE_READS(4 + FN, 1);                                                                                 \
    SB();                                                                                                  \
    c = cn;                                                                                                \
    s9 = sn;                                                                                               \
  } while (0)

  // Composed modes: a stage = the TWO kw taps in use (four half-taps); same pipeline, the B sets no longer flip between stages.  The stage number
  // S_ within the chunk is a compile-time constant (round 5: the four stages of a chunk are unrolled): with a run-time stage counter every stage
  // re-derived its tap, the next stage's tap and both weight offsets on the scalar unit -- ~100 scalar instructions for 64 MFMAs.
#define STAGE2(S_)                                                                                         \
  do {                                                                                                     \
    const bool last = (S_) == NSK - 1 && c + 1 == nchunk;                                                  \
    const int cn = ((S_) == NSK - 1 && !last) ? c + 1 : c, sn = last ? (S_) : (((S_) + 1) % NSK);          \
    const int kwb = KWB(c), kwbn = KWB(cn);                                                                \
    LOAD_W(cn, SID(cn, sn), kwbn);                                                                         \
    const bool halo_next = (S_) == NSK - 1 && !last; /* block-uniform */                                   \
    const int tap64 = ((SID(c, (S_)) / 3) * HH + (SID(c, (S_)) % 3)) * (HP * 64);                          \
    const int ntap64 = ((SID(cn, sn) / 3) * HH + (SID(cn, sn) % 3)) * (HP * 64);                           \
    const int a0 = kwb ? akw[1] : akw[0], a1 = kwb ? akw[2] : akw[1], an0 = kwbn ? akw[1] : akw[0];        \
    /* ht0: first tap, half 0 */                                                                           \
    LOADA(1, a0 + tap64, 1);                                                                               \
    MFMA_HALF(0, 0, 0, 0, 4);                                                                              \
    PIPE_READS(4, FN);                                                                                     \
    SB();                                                                                                  \
    /* ht1: first tap, half 1 */                                                                           \
    LOADA(0, a1 + tap64, 0);                                                                               \
    LOADB(1, wbuf + 1 * (BN * 64));                                                                        \
    MFMA_HALF(1, 0, 1, 0, 4);                                                                              \
    PIPE_READS(4 + FN, (4 * FN) / (4 + FN));                                                               \
    SB();                                                                                                  \
    /* ht2: second tap, half 0 */                                                                          \
    LOADA(1, a1 + tap64, 1);                                                                               \
    MFMA_HALF(0, 1, 0, 0, 4);                                                                              \
    PIPE_READS(4, FN);                                                                                     \
    SB();                                                                                                  \
    /* ht3: second tap, half 1 -- every LDS read of this stage (and, in a chunk's last stage, of this chunk's halo) is complete */ \
    __syncthreads();                                                                                       \
    STORE_W();                                                                                             \
    if (halo_next) DMA_HALO(c + 1);                                                                        \
    MFMA_HALF(1, 1, 1, 0, 2);                                                                              \
    PIPE_WRITES(2, FN / 2);                                                                                \
    SB();                                                                                                  \
    if (halo_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
    __syncthreads();                                                                                       \
    LOADA(0, an0 + ntap64, 0);                                                                             \
    LOADB(0, wbuf);                                                                                        \
    MFMA_HALF(1, 1, 1, 2, 4);                                                                              \
    PIPE_READS(4 + FN, 1);                                                                                 \
    SB();                                                                                                  \
  } while (0)

  DMA_HALO(0);
  LOAD_W(0, SID(0, 0), KWB(0));
  STORE_W();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  LOADA(0, (KWB(0) ? akw[1] : akw[0]) + ((SID(0, 0) / 3) * HH + (SID(0, 0) % 3)) * (HP * 64), 0);
  LOADB(0, wbuf);

  const int nstage = NSK * nchunk;
  if constexpr (MODE == 0 || MODE == 3) {
    for (int S = 0; S + 1 < nstage; S += 2) {
      STAGE(0);
      STAGE(1);
    }
    if (nstage & 1) STAGE(0);
  } else {
    for (c = 0; c < nchunk; ++c) {
      STAGE2(0);
      STAGE2(1);
      STAGE2(2);
      STAGE2(3);
    }
  }
  __syncthreads();   // the epilogue reuses the LDS
#undef STAGE
#undef STAGE2
#undef SID
#undef PARC
#undef KWB
#undef WTAP
#undef SB
#undef MFMA_HALF
#undef LOADA
#undef LOADB
#undef PIPE_READS
#undef PIPE_WRITES
#undef DMA_HALO
#undef CHUNK_OFS
#undef LOAD_W
#undef STORE_W

  // ---- epilogue: bias, store, BatchNorm partial statistics (one row per brick).  acc[fm][j][r]: voxel (d0 + wid, h0 + fm,
  //      w0 + 4 lg + r), channel n0 + 16 j + lr.  Addresses (round 5): a voxel's row is a SCALAR base (the wave's first voxel + fm h-steps + r
  //      w-steps, two scalar adds) + one lane offset (4 lg w-steps + 2 lr bytes) + 32 j as the store's immediate -- the 64-bit voxel index per
  //      (fm, r) that used to be rebuilt on the vector unit (64 quarter-rate multiplies + 64 64-bit multiply-adds per lane) is gone.  Composed
  //      forward: the bias of a fine voxel's border class is a per-(fm) scalar class plus, in bricks on the phase's w border, one lane column
  //      (r = 0 of lanes lg = 0 for the phases with an even fine w, r = 3 of lg = 3 for the odd ones): 8-16 bias loads per lane, not 128. ----
  float s1[FN], s2[FN], bv[FN], bvE[FN];
  float bsc[FN], bsh[FN], bmu[FN], brs[FN];   // BNR: the layer below's BatchNorm coefficients of this lane's FN channels
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    s1[j] = 0.f;
    s2[j] = 0.f;
    bv[j] = (!UPCF && !BNR && p.bias) ? p.bias[n0 + j * 16 + lr] : 0.f;   // BNR: a data gradient, no bias
    bvE[j] = 0.f;
  }
  const int uch0 = UPCF ? n0 - uph * p.upc : 0;                       // first channel of this tile inside its phase
  const int ypitch = UPCF ? p.upc : p.Nc;
  // byte steps of the output along the brick's w and h axes (composed forward: a coarse step is two fine voxels)
  const int fineW = PERM ? 2 * ypitch * (UPCF ? 2 * p.H : p.H) : 2 * ypitch;
  const int fineH = PERM ? 2 * ypitch : 2 * ypitch * (UPCF ? 2 * p.W : p.W);
  const int WS = UPCF ? 2 * fineW : fineW, HS = UPCF ? 2 * fineH : fineH;
  const int64_t row0 = UPCF ? FVOX(n, 2 * (d0 + wid) + BITD(uph), 2 * h0 + BITH(uph), 2 * w0 + BITW(uph)) : VOX(n, d0 + wid, h0, w0);
  char* const ybase = reinterpret_cast<char*>(p.y) + (row0 * ypitch + (UPCF ? uch0 : n0)) * 2;   // wave-uniform
  // BNR: the wave's plane of the layer below's pre-normalisation tensor (128 voxels x BN channels, the geometry of the output tile) goes to LDS by DMA
  // now -- the halo is dead -- and lands under the stores below: 16 (BN = 64) / 8 (BN = 32) one-KiB pieces per wave, a piece = 64 / SL consecutive w
  // voxels of one h line x SL 16-byte slots.  The epilogue reads it back two bytes per lane (channel 16 j + lr of voxel 4 lg + r): the four lg groups of
  // a wave sit 4 voxels = a multiple of 256 bytes apart -- the same banks -- so a voxel's slots are XOR-ed with 2 lg on the SOURCE side (BN = 64:
  // conflict-free; BN = 32, four slots per voxel: two-way).  Bytes [0, BNR_LDS0) stay free for the statistics rows' block reduction.
  constexpr int BNR_LDS0 = 4096, BNR_COEF = 2048;   // every wave loads its own copy of the coefficient piece to the same place (identical bytes)
  if (BNR) {
    constexpr int SL = BN / 8, VP = 64 / SL, PPL = 16 / VP, NPW = 8 * PPL;
    static_assert(!BNR || BNR_LDS0 + NW * 128 * BN * 2 <= HALO_BYTES, "the y tile reuses the halo buffer");
    // every vector-memory load of the main loop has long returned; saying so where the compiler sees it (the builtin, not asm) keeps its own counted
    // waits -- it protects registers of the loop's last weight loads when the epilogue reuses them -- from landing behind the requests below,
    // where vmcnt's in-order retirement would turn them into waits for the DMA
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only
    const uint64_t bnbase = (uint64_t)(uintptr_t)p.bn_y + (uint64_t)((row0 * ypitch + n0) * 2);   // wave-uniform
    const uint32_t ydst = lds_base + BNR_LDS0 + wid * (128 * BN * 2);
    const int vl = lane / SL, sl = lane % SL;
    uint32_t m0keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(m0keep));
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
      const int w = (q % PPL) * VP + vl;
      const int lsl = sl ^ (BN == 64 ? ((w >> 2) & 3) * 2 : ((w >> 2) & 1) * 2);
      const uint32_t voff = (uint32_t)((q / PPL) * HS + w * WS + lsl * 16);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(voff), "s"(ydst + q * 1024), "s"(bnbase) : "memory");
    }
    // ... and the four coefficient vectors of this channel tile as one more piece, [scale | shift | mean | rstd][BN] floats at BNR_COEF: loaded into
    // registers they would be ordinary vector-memory loads among the stores below, and the compiler's counted waits for them (it does not see the
    // asm's requests) would wait for the DMA pieces -- or, placed behind the stores, for the stores
    {
      constexpr int LPA = BN / 4;   // lanes per array (16 bytes each)
      const int arr = (lane / LPA) & 3;
      const float* src = arr == 0 ? p.bn_scale : arr == 1 ? p.bn_shift : arr == 2 ? p.bn_mean : p.bn_rstd;
      const float* lsrc = src + n0 + (lane % LPA) * 4;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(lsrc), "s"(lds_base + BNR_COEF) : "memory");
    }
    asm volatile("s_mov_b32 m0, %0" ::"s"(m0keep));
  }
  const uint32_t yl = (uint32_t)(lg * 4 * WS + lr * 2);
  // composed forward: border classes (0 first, 1 inside, 2 last per axis; class number in volume order)
  const int fd_ = 2 * (d0 + wid) + BITD(uph);
  const int cd = fd_ == 0 ? 0 : (fd_ == 2 * p.D - 1 ? 2 : 1);
  const int pwb = BITW(uph);                                          // the phase's fine w parity: 0 -> only fw = 0 can be a border, 1 -> only fw = 2 W - 1
  const bool wedge = UPCF && (pwb ? (w0 + TW == p.W) : (w0 == 0));    // wave-uniform: this brick holds the phase's w border
  const bool elane = wedge && lg == (pwb ? 3 : 0);
  int ch_cur = -1;
#pragma unroll
  for (int fm = 0; fm < 8; ++fm) {
    if (UPCF) {
      const int fh = 2 * (h0 + fm) + BITH(uph);
      const int ch = fh == 0 ? 0 : (fh == 2 * p.H - 1 ? 2 : 1);
      if (ch != ch_cur) {   // wave-uniform: at most three times per brick
        ch_cur = ch;
        const int clsI = PERM ? (cd * 3 + 1) * 3 + ch : (cd * 3 + ch) * 3 + 1;
        const int cwe = pwb ? 2 : 0, clsE = PERM ? (cd * 3 + cwe) * 3 + ch : (cd * 3 + ch) * 3 + cwe;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          bv[j] = p.bias_tab[clsI * p.upc + uch0 + j * 16 + lr];
          bvE[j] = wedge ? p.bias_tab[clsE * p.upc + uch0 + j * 16 + lr] : bv[j];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      char* const yrow = ybase + (int64_t)fm * HS + (int64_t)r * WS;   // scalar
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        float bias = bv[j];
        if (UPCF && (r == 0 || r == 3)) bias = (elane && r == (pwb ? 3 : 0)) ? bvE[j] : bv[j];
        const float val = acc[fm][j][r] + bias;
        const bf16 vb = (bf16)val;
        *reinterpret_cast<bf16*>(yrow + yl + j * 32) = vb;
        if (BNR) {
          acc[fm][j][r] = (float)vb;   // the value as stored: what the separate reduce pass would read back
        } else {
          s1[j] += val;
          s2[j] += val * val;
        }
      }
    }
    // BNR: the wave's DMA pieces (a wave reads only the plane it requested: no barrier) are the OLDEST vector-memory operations in flight; behind them
    // three lines of stores (12 FN <= 48).  vmcnt retires in order on gfx9, so "at most 12 FN outstanding" means the pieces have landed -- without waiting
    // for any store to complete (s_waitcnt vmcnt(0) here cost the 64-channel form 7 %: a store's completion is microseconds away).
    if (BNR && fm == 2) {
      if (FN == 4) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    }
  }
  if (BNR) {
    // bn_bwd_reduce_kernel's arithmetic (ReLU) on the stored values against the y tile staged above
    const char* ytile = smem + BNR_LDS0 + wid * (128 * BN * 2) + (4 * lg) * (BN * 2) + (lr >> 3) * 16 + (lr & 7) * 2;
    const float* coef = reinterpret_cast<const float*>(smem + BNR_COEF);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      bsc[j] = coef[j * 16 + lr];
      bsh[j] = coef[BN + j * 16 + lr];
      bmu[j] = coef[2 * BN + j * 16 + lr];
      brs[j] = coef[3 * BN + j * 16 + lr];
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {   // one channel fragment at a time (32 LDS reads in flight; all 128 at once spilled)
      const int jx = j ^ (BN == 64 ? lg : (lg & 1));   // the staging swizzle (source side of the DMA)
#pragma unroll
      for (int fm = 0; fm < 8; ++fm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float yv = (float)*reinterpret_cast<const bf16*>(ytile + (fm * 16 + r) * (BN * 2) + jx * 32);
          const float dz = bsc[j] * yv + bsh[j] > 0.f ? acc[fm][j][r] : 0.f;
          s1[j] += dz;
          s2[j] += dz * (yv - bmu[j]) * brs[j];
        }
      asm volatile("" : "+v"(s1[j]), "+v"(s2[j])::"memory");   // this fragment's sums are complete here, the next fragment's reads start here
    }
  }
  if (p.stats) {
    // one statistics row per 4-plane half of the brick: row numbering and summation order are those of the 4 x 8 x 16 brick for either NW
    float* red = reinterpret_cast<float*>(smem);  // [NW waves][64 ch][2]; the loop ended with a barrier
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
    const int half = tid >> 6, ch = tid & 63;
    if (half < NW / 4 && ch < BN) {
      float a = 0.f, c2 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a += red[((half * 4 + q) * 64 + ch) * 2 + 0];
        c2 += red[((half * 4 + q) * 64 + ch) * 2 + 1];
      }
      const int64_t brick_id = NW == 4 ? (int64_t)brick_lin : (((int64_t)n * (p.D / 4) + (d0 >> 2) + half) * bh + h0 / TH) * bw + w0 / TW;
      float* o = p.stats + (brick_id * p.Nc + n0 + ch) * 2;
      o[0] = a;
      o[1] = c2;
    }
  }
}

#undef BITD
#undef BITH
#undef BITW
#undef VOX
#undef FVOX

// 0: no brick tiling; 1: (4, 8, 16) bricks along (D, H, W); 2: along (D, W, H) (PERM instantiations)
inline int brick16_perm(int D, int H, int W) {
  if (D % TD == 0 && H % TH == 0 && W % TW == 0) return 1;
  if (D % TD == 0 && W % TH == 0 && H % TW == 0) return 2;
  return 0;
}


// p.D / p.H / p.W arrive as the volume's extents; perm == 2: handed to the PERM instantiation as the extents along the brick axes (D, W, H)
template <int BN, int MODE, int NW = 4, bool BNR = false>
inline int launch16(Brick16Params p, dim3 grid, hipStream_t stream, const char* what) {
  constexpr size_t lds = B16Geom<NW>::HALO_BYTES + 3 * BN * 64 + (NW == 4 ? 0 : 1024);   // NW = 8: + the halo plan's mask words
  static std::once_flag attr_once;
  std::call_once(attr_once, [&] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(brick16_conv_kernel<BN, MODE, 0, NW, BNR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(brick16_conv_kernel<BN, MODE, 1, NW, BNR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  if (brick16_perm(p.D, p.H, p.W) == 2) {
    std::swap(p.H, p.W);
    hipLaunchKernelGGL((brick16_conv_kernel<BN, MODE, 1, NW, BNR>), grid, dim3(NW * 64), lds, stream, p);
  } else {
    hipLaunchKernelGGL((brick16_conv_kernel<BN, MODE, 0, NW, BNR>), grid, dim3(NW * 64), lds, stream, p);
  }
  return pcrl_check_launch(what);
}


}  // namespace

// ---- internal interface (conv_brick16.hip) ----
bool pcrl_brick16_conv_eligible(int N, int D, int H, int W, int Ci, int Co, int dtype);
int64_t pcrl_brick16_conv_rows(int N, int D, int H, int W);
