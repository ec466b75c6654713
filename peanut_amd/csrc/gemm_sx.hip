// fp32 GEMM emulated on the bf16 matrix cores, operands pre-split in memory ("S" format).
//
// Every fp32 value is held as NP bf16 pieces, each the bf16 rounding of what the previous pieces left
// (x = hi + mid + lo exactly for NP = 3: 3 x 8 signed mantissa bits cover fp32's 24; NP = 2 keeps ~17).  A product a*b is rebuilt from
//   NP = 3, 6 MFMA terms (bf16x6): hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi   (dropped terms <= 2^-24 |ab|:
//                                   the size of fp32's own product rounding -- fp32-class results)
//   NP = 2, 3 MFMA terms (bf16x3): hi*hi + hi*lo + lo*hi                                 (~2^-16 relative)
// on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate) with fp32 accumulation.  Unlike conv_igemm_split.hip, which
// splits fp32 activations in registers inside its k-loop, the pieces here are written once by the PRODUCER of the
// tensor (conv epilogues via store_s_quad, the Winograd transforms) so that the k-loop is nothing but LDS-DMA,
// fragment reads and MFMAs (tools/micro/bf16xn_loop.hip: 235-277 TF/s fp32-equivalent for bf16x6 against
// 115-129 TF/s of the fp32 MFMA kernel on the same shapes).
//
// S layout of a [rows x C] operand: [C/16 chunks][NP planes][rows_pad][16 bf16], rows_pad a multiple of 128.
// One k-tile (16 channels) of a 128-row tile is therefore NP contiguous 4 KiB runs: four 1 KiB LDS-DMA pieces per
// plane, landing lane-linear in LDS as [plane][row][32 B].  The two 16-byte halves of a row are swapped for rows
// with bit 3 set (applied to the DMA source address and to the ds_read_b128 address alike), which makes the 16
// rows of a fragment read hit 16 distinct bank groups.  Weights use the same layout per (n-tile, k-tile).
//
// Used for the 1x1 convs and the Winograd position GEMMs in the bf16x6 / bf16x3 modes (precision = 3 / 1);
// replaces the same reference call sites as conv_pw.hip (resnet.py:267-307 conv1/conv3/downsample, the GEMM inside
// the Winograd form of conv2 and of psp_head.py:86-93).
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"

namespace peanut {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BN, int WM, int WN, int NP, int STAGES, bool PIPE = false>
__global__ __launch_bounds__(256) void gemm_sx_kernel(const ConvKParams p) {
  constexpr int BM = 128;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int A_BYTES = NP * BM * 32, B_BYTES = NP * BN * 32, STAGE = A_BYTES + B_BYTES;   // one k-tile = 16 channels
  constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024;                       // 1 KiB = 32 rows of a plane
  constexpr int PIECES = A_PIECES + B_PIECES;
  constexpr int PER_WAVE = (PIECES + 3) / 4;
  constexpr bool EVEN = PIECES % 4 == 0;   // every wave moves PER_WAVE pieces: no per-piece tests, a constant vmcnt
  constexpr int CS = BN + 4;
  constexpr int EP = (BM * CS * 4 > STAGES * STAGE) ? WM : 1;
  constexpr int ER = BM / EP;
  constexpr int SMEM_BYTES = (STAGES * STAGE > ER * CS * 4) ? STAGES * STAGE : ER * CS * 4;
  static_assert(WM * WN == 4 && TM % 32 == 0 && TN % 32 == 0, "4 waves, wave tile a multiple of 32x32");
  static_assert(NP == 2 || NP == 3, "two or three bf16 pieces");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

  const int tid = threadIdx.x;
  const Work wk = decode_work(p);
  const int mt = wk.mt, nt = wk.nt, nk = wk.kt1 - wk.kt0;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- per-lane DMA sources.  Piece i (i < A_PIECES): plane i / 4, rows (i % 4) * 32 .. + 32 of the A tile; the
  //      rest: plane / 32-row block of the weight tile.  Lane l of a piece: row l / 2, LDS half l % 2. ----
  const unsigned char* src[PER_WAVE];
  unsigned step[PER_WAVE];   // bytes to the same place in the next k-tile
  const size_t a_chunk = (size_t)NP * p.xs_rows * 32;   // bytes of one 16-channel chunk of A
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int piece = wave * PER_WAVE + j;
    const int lr = lane >> 1, lh = lane & 1;
    if (piece < A_PIECES) {
      const int q = piece / (BM / 32), r = (piece % (BM / 32)) * 32 + lr;
      const int m = m0 + r;
      const int mc = m < p.M ? m : p.M - 1;   // rows past the end compute a valid row and are dropped
      const int b = mc / p.HoWo;
      const int rem = mc - b * p.HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const size_t pix = (size_t)b * p.H * p.W + (size_t)oy * p.stride * p.W + (size_t)ox * p.stride;
      const int h = lh ^ ((r >> 3) & 1);
      src[j] = reinterpret_cast<const unsigned char*>(p.xs) + (size_t)wk.kt0 * a_chunk + ((size_t)q * p.xs_rows + pix) * 32 + h * 16;
      step[j] = (unsigned)a_chunk;
    } else {
      const int pb = piece - A_PIECES;        // may run past B_PIECES for the last wave: clamp, the copy is harmless
      const int pbc = pb < B_PIECES ? pb : B_PIECES - 1;
      const int q = pbc / (BN / 32), r = (pbc % (BN / 32)) * 32 + lr;
      const int h = lh ^ ((r >> 3) & 1);
      const unsigned char* wtile = reinterpret_cast<const unsigned char*>(p.w) +
                                   (p.mt_per_group ? (size_t)(mt / p.mt_per_group) * (size_t)p.w_group_stride : 0) +
                                   ((size_t)nt * p.nkt + wk.kt0) * B_BYTES;
      src[j] = wtile + ((size_t)q * BN + r) * 32 + h * 16;
      step[j] = B_BYTES;
    }
  }
#define SX_DMA_TILE(stage)                                                                                        \
  {                                                                                                               \
    _Pragma("unroll") for (int j = 0; j < PER_WAVE; ++j) {                                                        \
      const int piece = wave * PER_WAVE + j;                                                                      \
      if (EVEN || piece < PIECES)                                                                                 \
        __builtin_amdgcn_global_load_lds((gptr_t)src[j], (lptr_t)((stage) + piece * 1024), 16, 0, 0);            \
      src[j] += step[j];                                                                                          \
    }                                                                                                             \
  }
  // Wait until all but the newest `my_pieces` DMA pieces of this wave have landed (i.e. every tile except the one
  // issued last), then the workgroup barrier that publishes them.  Raw s_barrier: a __syncthreads would make the
  // compiler drain the DMA queue (vmcnt(0)) and with it the prefetch distance.
  const int my_pieces = (wave + 1) * PER_WAVE <= PIECES ? PER_WAVE : (PIECES - wave * PER_WAVE > 0 ? PIECES - wave * PER_WAVE : 0);
#define SX_WAIT_ALL_BUT_LAST_TILE()                                                \
  if constexpr (EVEN && PER_WAVE == 6) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }      \
  else if constexpr (EVEN && PER_WAVE == 5) { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); } \
  else if constexpr (EVEN && PER_WAVE == 4) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); } \
  else if constexpr (EVEN && PER_WAVE == 3) { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); } \
  else {                                                                           \
    switch (my_pieces) {                                                           \
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;              \
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;              \
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;              \
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;              \
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;              \
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;              \
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;             \
    }                                                                              \
  }
#define SX_BARRIER()                                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("" ::: "memory");
#define SX_DMA_LANDED_BARRIER()                    \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
  SX_BARRIER()

  // ---- MFMA fragment coordinates: lane (li, hi) reads 8 consecutive k (16 B) of row li of each 32-row tile ----
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const int frag = li * 32 + ((hi ^ ((li >> 3) & 1)) * 16);      // byte offset inside a plane, rows of 32 B
  const int a_row = wm * TM * 32 + frag;
  const int b_row = A_BYTES + wn * TN * 32 + frag;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  static_assert(PER_WAVE <= 6, "SX_WAIT_ALL_BUT_LAST_TILE covers up to 6 pieces per wave");
  ResPrefetch rp;
  rp.on = false;
  // ---- prologue: STAGES - 1 k-tiles in flight ----
  SX_DMA_TILE(smem);
  if (STAGES == 3 && nk > 1) {
    SX_DMA_TILE(smem + STAGE);
    SX_WAIT_ALL_BUT_LAST_TILE();
    SX_BARRIER();
  } else {
    SX_DMA_LANDED_BARRIER();
  }

  // byte offsets of the stage being read and of the one(s) being filled, rotated every k-tile
  int o_cur = 0, o_fill = (STAGES - 1) * STAGE, o_mid = STAGE;   // o_mid only used with three stages
#define SX_READ_FRAGS(af, bf, stage)                                                                                  \
  _Pragma("unroll") for (int q = 0; q < NP; ++q) {                                                                    \
    _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                                    \
      af[q][t] = *reinterpret_cast<const bf16x8*>((stage) + q * (BM * 32) + a_row + t * 32 * 32);                     \
    _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                                    \
      bf[q][u] = *reinterpret_cast<const bf16x8*>((stage) + q * (BN * 32) + b_row + u * 32 * 32);                     \
  }
  // the piece products of one k-tile, smallest terms first; PART 0 = first half of the list, 1 = second half, 2 = all.
  // Each accumulator receives its terms in the same order whichever way the list is cut.
#define SX_MFMA(af, bf, PART)                                                                                         \
  {                                                                                                                   \
    constexpr int NPROD = NP == 3 ? 6 : 3;                                                                            \
    constexpr int PA[6] = {NP == 3 ? 2 : 1, NP == 3 ? 0 : 0, NP == 3 ? 1 : 0, 1, 0, 0};                               \
    constexpr int PB[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};                                             \
    constexpr int LO = (PART) == 1 ? NPROD / 2 : 0, HI = (PART) == 0 ? NPROD / 2 : NPROD;                             \
    _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                                    \
      _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                                  \
        _Pragma("unroll") for (int i = LO; i < HI; ++i)                                                               \
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[i]][t], bf[PB[i]][u], acc[t][u], 0, 0, 0);        \
  }
  if constexpr (STAGES == 3 && PIPE) {
    // Fragment reads of k-tile kt+1 travel LDS -> registers under the second half of k-tile kt's MFMAs: the wait for
    // tile kt+1 and the barrier sit in the middle of the MFMA block, the reads are issued right behind it, and the
    // next iteration starts on registers that are already loaded (one barrier per k-tile as before).
    bf16x8 afA[NP][MI], bfA[NP][NI], afB[NP][MI], bfB[NP][NI];
    SX_READ_FRAGS(afA, bfA, smem + o_cur);
#define SX_PIPE_STEP(afC, bfC, afN, bfN)                                                        \
    {                                                                                             \
      const bool more = kt + 2 < nk;                                                              \
      if (more) SX_DMA_TILE(smem + o_fill);                                                       \
      SX_MFMA(afC, bfC, 0);                                                                       \
      if (more) { SX_WAIT_ALL_BUT_LAST_TILE(); SX_BARRIER(); } else { SX_DMA_LANDED_BARRIER(); }  \
      if (kt + 1 < nk) SX_READ_FRAGS(afN, bfN, smem + o_mid);                                     \
      SX_MFMA(afC, bfC, 1);                                                                       \
      { const int t_ = o_cur; o_cur = o_mid; o_mid = o_fill; o_fill = t_; }                       \
    }
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      SX_PIPE_STEP(afA, bfA, afB, bfB);
      ++kt;
      SX_PIPE_STEP(afB, bfB, afA, bfA);
      --kt;
    }
    if (kt < nk) SX_PIPE_STEP(afA, bfA, afB, bfB);
#undef SX_PIPE_STEP
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* const cur = smem + o_cur;
      const bool more = kt + STAGES - 1 < nk;
      if (more) SX_DMA_TILE(smem + o_fill);
      if (kt == nk - 1) rp = conv_res_prefetch<BM, BN, EP, 256>(p, wk, m0, n0);   // under the last k-tile's MFMAs
      bf16x8 af[NP][MI], bf[NP][NI];
      SX_READ_FRAGS(af, bf, cur);
      SX_MFMA(af, bf, 2);
      // the next k-tile must have landed; the one just issued may stay in flight across the barrier
      if (STAGES == 3 && more) { SX_WAIT_ALL_BUT_LAST_TILE(); SX_BARRIER(); }
      else { SX_DMA_LANDED_BARRIER(); }
      if (STAGES == 3) { const int t = o_cur; o_cur = o_mid; o_mid = o_fill; o_fill = t; }
      else { const int t = o_cur; o_cur = o_fill; o_fill = t; }
    }
  }
  // every wave passed a barrier after its last fragment read; the epilogue may reuse the stages
  SX_BARRIER();
#undef SX_READ_FRAGS
#undef SX_MFMA
#undef SX_DMA_TILE
#undef SX_DMA_LANDED_BARRIER
#undef SX_WAIT_ALL_BUT_LAST_TILE
#undef SX_BARRIER

  conv_epilogue<BM, BN, WM, WN, EP>(p, wk, acc, reinterpret_cast<float*>(smem), m0, n0, PIPE ? nullptr : &rp);
}

// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 tiles, 8 waves.  Measured on the 128 x 128 kernel (rocprofv3 PMC, profiles/r2c): 44 % MFMA utilisation
// with L2 at ~10 TB/s and HBM at 3.3 TB/s -- neither memory level saturated, the loop is bound by its own rhythm:
// one barrier and one DMA round trip per 16-channel k-tile, i.e. per 768 matrix-core cycles of a wave, with only two
// k-tiles of prefetch distance (1.3 us at full rate, about one loaded L2/HBM round trip).  A 256 x 256 tile quarters
// the barriers per FLOP (48 MFMAs per wave and k-tile), halves the operand bytes per FLOP ((256 + 256) / 256^2) and
// doubles the time a k-tile in flight has to land (two waves per SIMD, 3072 cycles per k-tile and SIMD; three stages
// of 48 KiB = 144 KiB of the CU's 160 KiB LDS, one workgroup per CU).  The weights keep the 128-row packing: a
// 256-wide n-tile is two adjacent packed tiles.  Wave tile 128 x 64 (4 x 2 MFMA tiles, 128 accumulator registers).
// ---------------------------------------------------------------------------------------------------------------
template <int NP, int STAGES>
__global__ __launch_bounds__(512) void gemm_sx256_kernel(const ConvKParams p) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NT = 512;
  constexpr int TM = BM / WM, TN = BN / WN;        // 128 x 64
  constexpr int MI = TM / 32, NI = TN / 32;        // 4 x 2
  constexpr int A_BYTES = NP * BM * 32, B_BYTES = NP * BN * 32, STAGE = A_BYTES + B_BYTES;
  constexpr int B128_BYTES = NP * 128 * 32;        // one packed 128-row weight tile of one k-tile
  constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024, PIECES = A_PIECES + B_PIECES;
  constexpr int PER_WAVE = PIECES / 8;
  static_assert(PIECES % 8 == 0 && (PER_WAVE == 6 || PER_WAVE == 4), "8 waves move the same number of 1 KiB pieces");
  constexpr int CS = BN + 4, EP = WM, ER = BM / EP;
  constexpr int SMEM_BYTES = (STAGES * STAGE > ER * CS * 4) ? STAGES * STAGE : ER * CS * 4;
  static_assert(STAGES == 3, "three LDS stages");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

  const int tid = threadIdx.x;
  const Work wk = decode_work(p);
  const int mt = wk.mt, nt = wk.nt, nk = wk.kt1 - wk.kt0;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const unsigned char* src[PER_WAVE];
  unsigned step[PER_WAVE];
  const size_t a_chunk = (size_t)NP * p.xs_rows * 32;
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int piece = wave * PER_WAVE + j;
    const int lr = lane >> 1, lh = lane & 1;
    if (piece < A_PIECES) {
      const int q = piece / (BM / 32), r = (piece % (BM / 32)) * 32 + lr;
      const int m = m0 + r;
      const int mc = m < p.M ? m : p.M - 1;   // rows past the end compute a valid row and are dropped
      const int b = mc / p.HoWo;
      const int rem = mc - b * p.HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const size_t pix = (size_t)b * p.H * p.W + (size_t)oy * p.stride * p.W + (size_t)ox * p.stride;
      const int h = lh ^ ((r >> 3) & 1);
      src[j] = reinterpret_cast<const unsigned char*>(p.xs) + (size_t)wk.kt0 * a_chunk + ((size_t)q * p.xs_rows + pix) * 32 + h * 16;
      step[j] = (unsigned)a_chunk;
    } else {
      const int pb = piece - A_PIECES;
      const int q = pb / (BN / 32), r = (pb % (BN / 32)) * 32 + lr;     // row of the 256-wide n-tile
      const int sub = r >> 7, r128 = r & 127;                            // which packed 128-row tile, row inside it
      const int h = lh ^ ((r >> 3) & 1);
      const unsigned char* wtile = reinterpret_cast<const unsigned char*>(p.w) +
                                   (p.mt_per_group ? (size_t)(mt / p.mt_per_group) * (size_t)p.w_group_stride : 0) +
                                   ((size_t)(2 * nt + sub) * p.nkt + wk.kt0) * B128_BYTES;
      src[j] = wtile + ((size_t)q * 128 + r128) * 32 + h * 16;
      step[j] = B128_BYTES;
    }
  }
#define SX_DMA_TILE(stage)                                                                                        \
  {                                                                                                               \
    _Pragma("unroll") for (int j = 0; j < PER_WAVE; ++j) {                                                        \
      __builtin_amdgcn_global_load_lds((gptr_t)src[j], (lptr_t)((stage) + (wave * PER_WAVE + j) * 1024), 16, 0, 0); \
      src[j] += step[j];                                                                                          \
    }                                                                                                             \
  }
#define SX_WAIT_ALL_BUT_LAST_TILE()                                                              \
  if constexpr (PER_WAVE == 6) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }              \
  else { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
#define SX_BARRIER()                                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("" ::: "memory");
#define SX_DMA_LANDED_BARRIER()                    \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
  SX_BARRIER()

  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const int frag = li * 32 + ((hi ^ ((li >> 3) & 1)) * 16);
  const int a_row = wm * TM * 32 + frag;
  const int b_row = A_BYTES + wn * TN * 32 + frag;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  SX_DMA_TILE(smem);
  if (nk > 1) {
    SX_DMA_TILE(smem + STAGE);
    SX_WAIT_ALL_BUT_LAST_TILE();
    SX_BARRIER();
  } else {
    SX_DMA_LANDED_BARRIER();
  }
  ResPrefetch rp;
  rp.on = false;
  int o_cur = 0, o_fill = 2 * STAGE, o_mid = STAGE;
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* const cur = smem + o_cur;
    const bool more = kt + 2 < nk;
    if (more) SX_DMA_TILE(smem + o_fill);
    if (kt == nk - 1) rp = conv_res_prefetch<BM, BN, EP, NT>(p, wk, m0, n0);   // under the last k-tile's MFMAs
    bf16x8 af[NP][MI], bf[NP][NI];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
      for (int t = 0; t < MI; ++t) af[q][t] = *reinterpret_cast<const bf16x8*>(cur + q * (BM * 32) + a_row + t * 32 * 32);
#pragma unroll
      for (int u = 0; u < NI; ++u) bf[q][u] = *reinterpret_cast<const bf16x8*>(cur + q * (BN * 32) + b_row + u * 32 * 32);
    }
#pragma unroll
    for (int t = 0; t < MI; ++t)
#pragma unroll
      for (int u = 0; u < NI; ++u) {
        if constexpr (NP == 3) {
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][t], bf[0][u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][t], bf[2][u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][t], bf[1][u], acc[t][u], 0, 0, 0);
        }
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][t], bf[0][u], acc[t][u], 0, 0, 0);
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][t], bf[1][u], acc[t][u], 0, 0, 0);
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][t], bf[0][u], acc[t][u], 0, 0, 0);
      }
    if (more) { SX_WAIT_ALL_BUT_LAST_TILE(); SX_BARRIER(); }
    else { SX_DMA_LANDED_BARRIER(); }
    { const int t = o_cur; o_cur = o_mid; o_mid = o_fill; o_fill = t; }
  }
#undef SX_DMA_TILE
#undef SX_DMA_LANDED_BARRIER
#undef SX_WAIT_ALL_BUT_LAST_TILE
#undef SX_BARRIER

  conv_epilogue<BM, BN, WM, WN, EP, NT>(p, wk, acc, reinterpret_cast<float*>(smem), m0, n0, &rp);
}

template <int NP>
int launch_sx256_t(ConvKParams p, float* ws, size_t ws_floats, hipStream_t stream) {
  static int slots = 0;
  p.ntiles = p.cout / 256;                       // n-tiles of THIS kernel (decode_work)
  if (p.mt_per_group) p.mt_per_group /= 2;       // 256-row tiles per Winograd position
  return launch_with_tail_split<decltype(&gemm_sx256_kernel<NP, 3>), 256, 256, 512>(&gemm_sx256_kernel<NP, 3>, p, ws, ws_floats,
                                                                                    stream, &slots);
}

template <int BN, int WM, int WN, int NP>
int launch_sx_t(const ConvKParams& p, float* ws, size_t ws_floats, hipStream_t stream) {
  static const int stages = [] { const char* e = getenv("PEANUT_SX_STAGES"); return (e && e[0] == '2') ? 2 : 3; }();
  if (stages == 2) {
    static int slots2 = 0;
    return launch_with_tail_split<decltype(&gemm_sx_kernel<BN, WM, WN, NP, 2>), 128, BN>(&gemm_sx_kernel<BN, WM, WN, NP, 2>, p,
                                                                                          ws, ws_floats, stream, &slots2);
  }
  // measured on MI355X (profiles/r2e): the pipelined fragment reads change nothing (886 vs 883 maps/s bf16x6, bf16x3
  // slower) -- the loop is not bound by LDS read latency -- so the plain loop stays the default
  static const bool pipe = [] { const char* e = getenv("PEANUT_SX_PIPE"); return e && e[0] == '1'; }();
  if (pipe) {
    static int slots3p = 0;
    return launch_with_tail_split<decltype(&gemm_sx_kernel<BN, WM, WN, NP, 3, true>), 128, BN>(&gemm_sx_kernel<BN, WM, WN, NP, 3, true>,
                                                                                                p, ws, ws_floats, stream, &slots3p);
  }
  static int slots3 = 0;
  return launch_with_tail_split<decltype(&gemm_sx_kernel<BN, WM, WN, NP, 3>), 128, BN>(&gemm_sx_kernel<BN, WM, WN, NP, 3>, p, ws,
                                                                                        ws_floats, stream, &slots3);
}

inline unsigned short bf16_piece_host(float& v) {   // same rounding as bf16_piece (conv_common.h)
  unsigned bits;
  __builtin_memcpy(&bits, &v, 4);
  bits += 0x7fffu + ((bits >> 16) & 1u);
  bits &= 0xffff0000u;
  float piece;
  __builtin_memcpy(&piece, &bits, 4);
  v -= piece;
  return (unsigned short)(bits >> 16);
}

}  // namespace

// bytes of the S-packed weights of a 1x1 layer: [n-tile][k-tile of 16][plane][bn_tile][16 bf16]
size_t sx_packed_bytes(int cin_pad, int cout, int bn_tile, int planes) {
  const size_t ntiles = (cout + bn_tile - 1) / bn_tile;
  return ntiles * (size_t)(cin_pad / 16) * planes * bn_tile * 32;
}

// w: [cout][cin_real] fp32 (a 1x1 conv's OIHW weights, or one Winograd position of U)
void pack_weights_sx(const float* w, int cout, int cin_real, int cin_pad, int bn_tile, int planes, void* out) {
  unsigned short* o = static_cast<unsigned short*>(out);
  const int ntiles = (cout + bn_tile - 1) / bn_tile, nkt = cin_pad / 16;
  for (int nt = 0; nt < ntiles; ++nt)
    for (int kt = 0; kt < nkt; ++kt) {
      unsigned short* tile = o + ((size_t)nt * nkt + kt) * planes * bn_tile * 16;
      for (int r = 0; r < bn_tile; ++r)
        for (int e = 0; e < 16; ++e) {
          const int n = nt * bn_tile + r, c = kt * 16 + e;
          float v = (n < cout && c < cin_real) ? w[(size_t)n * cin_real + c] : 0.f;
          for (int q = 0; q < planes; ++q) tile[((size_t)q * bn_tile + r) * 16 + e] = bf16_piece_host(v);
        }
    }
}

bool gemm_sx_uses_256(int cout, long long M, int mt_per_group, int bn_tile, int cin) {
  static const bool on = [] { const char* e = getenv("PEANUT_SX256"); return !(e && e[0] == '0'); }();
  static const int min_k = [] { const char* e = getenv("PEANUT_SX256_MINK"); return e ? atoi(e) : 0; }();
  return on && cin >= min_k && bn_tile == 128 && cout % 256 == 0 && mt_per_group % 2 == 0 && M * cout >= 256LL * 256 * 256;
}

// fp32 accumulate / epilogue as every other conv; p.xs (A), p.w (S-packed weights), p.s_planes set by the caller
int launch_gemm_sx(const ConvKParams& p, int bn_tile, int planes, float* ws, size_t ws_floats, hipStream_t stream) {
  if (!p.xs || p.xs_rows % 128 || p.ntaps != 1 || p.pad != 0 || p.c2 != 0 || p.c1 % 16)
    return fail(-2, "launch_gemm_sx: needs an S-format A operand of a pointwise layer");
  // 256 x 256 tiles when the shape allows: whole 256-wide n-tiles, whole 256-row tiles per Winograd position, enough
  // rows to fill the chip at one workgroup per CU (PEANUT_SX256=0 keeps the 128 x 128 kernel everywhere)
  if (gemm_sx_uses_256(p.cout, p.M, p.mt_per_group, bn_tile, p.c1)) {
    note_kernel(planes == 3 ? "gemm_sx6_256x256" : "gemm_sx3_256x256");
    return planes == 3 ? launch_sx256_t<3>(p, ws, ws_floats, stream) : launch_sx256_t<2>(p, ws, ws_floats, stream);
  }
  static const char* const names[2][3] = {{"gemm_sx3_128x128", "gemm_sx3_128x64", "gemm_sx3_128x32"},
                                          {"gemm_sx6_128x128", "gemm_sx6_128x64", "gemm_sx6_128x32"}};
  if ((planes == 2 || planes == 3) && (bn_tile == 128 || bn_tile == 64 || bn_tile == 32))
    note_kernel(names[planes - 2][bn_tile == 128 ? 0 : (bn_tile == 64 ? 1 : 2)]);
  if (planes == 3) {
    if (bn_tile == 128) return launch_sx_t<128, 2, 2, 3>(p, ws, ws_floats, stream);
    if (bn_tile == 64) return launch_sx_t<64, 2, 2, 3>(p, ws, ws_floats, stream);
    if (bn_tile == 32) return launch_sx_t<32, 4, 1, 3>(p, ws, ws_floats, stream);
  } else if (planes == 2) {
    if (bn_tile == 128) return launch_sx_t<128, 2, 2, 2>(p, ws, ws_floats, stream);
    if (bn_tile == 64) return launch_sx_t<64, 2, 2, 2>(p, ws, ws_floats, stream);
    if (bn_tile == 32) return launch_sx_t<32, 4, 1, 2>(p, ws, ws_floats, stream);
  }
  return fail(-2, "launch_gemm_sx: unsupported tile configuration");
}

}  // namespace peanut
