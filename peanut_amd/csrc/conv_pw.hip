// Pointwise (1x1, no padding, one source) convolutions and the Winograd position GEMMs: the same 128 x BN x 32 MFMA
// tiling as conv_igemm_kernel, but the k-tiles travel global memory -> LDS directly (global_load_lds_dwordx4,
// gfx950 LDS-DMA) instead of through VGPRs.
//
// Why a second kernel: measured with tools/micro/mfma_loop.hip on MI355X (2 workgroups/CU, 64 fp32 MFMAs per
// k-tile and wave): MFMAs + LDS fragment reads + one barrier sustain 150 TF/s; adding the 8 ds_write_b128 of a
// register-staged pipeline costs 8 %, its 8 global_load_dwordx4 another 4-9 % -- the VGPR traffic of the staging
// competes with the accumulate traffic of the matrix cores in the unified register file -- while 8
// global_load_lds cost 6 %.  The 1x1 convs and Winograd GEMMs are > 90 % of the forward's MFMA time, and for
// them a k-tile row is one contiguous 128-byte run, exactly what the LDS-DMA writes (wave-uniform base + lane * 16).
//
// LDS image: rows of 32 floats, unpadded (the DMA destination is lane-linear), 16-byte chunk c of row r stored at
// chunk position c ^ ((r >> 1) & 7).  The permutation is applied on the SOURCE address of the DMA (which row
// chunk a lane fetches) and again on the ds_read_b128 address -- the same involution on both sides -- and makes
// the MFMA fragment reads (16 consecutive rows, one chunk) hit 16 distinct 16-byte bank groups.
//
// Ordering (cdna_hip_programming.md, LDS-DMA rules): a stage is read only after `s_waitcnt vmcnt(0)` by the
// issuing waves followed by a workgroup barrier; it is refilled only after a barrier that follows every wave's
// last read of it.  One barrier per k-tile, double-buffered LDS, MFMA fragment reads software-pipelined in two
// halves exactly as in conv_igemm_kernel.
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"

namespace peanut {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BN, int WM, int WN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BN == 128 ? 2 : (BN == 64 ? 3 : 4))))
void conv_pw_glds_kernel(const ConvKParams p) {
  constexpr int BM = 128, BK = 32;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;
  constexpr int A_INSTR = BM / 32, B_INSTR = BN / 32;   // 1 KiB (8-row) DMA pieces per wave and k-tile
  constexpr int CS = BN + 4;
  constexpr int EP = (BM * CS > 2 * STAGE) ? WM : 1;
  constexpr int ER = BM / EP;
  constexpr int SMEM_FLOATS = (2 * STAGE > ER * CS) ? 2 * STAGE : ER * CS;
  static_assert(WM * WN == 4 && TM % 32 == 0 && TN % 32 == 0, "4 waves, wave tile a multiple of 32x32");
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const Work wk = decode_work(p);
  const int mt = wk.mt, nt = wk.nt, nk = wk.kt1 - wk.kt0;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA bases stay in SGPRs
  const int lr = lane >> 3, lp = lane & 7;   // row within an 8-row DMA piece, chunk position within the row

  // ---- per-lane DMA source pointers (advanced by one k-tile per iteration) ----
  const float* a_src[A_INSTR];
  const float* a_src2[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int r = (wave * A_INSTR + j) * 8 + lr;
    const int m = m0 + r;
    const int mc = m < p.M ? m : p.M - 1;            // rows past the end compute a valid row and are dropped
    const int b = mc / p.HoWo;
    const int rem = mc - b * p.HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const size_t pix = (size_t)b * p.H * p.W + (size_t)oy * p.stride * p.W + (size_t)ox * p.stride;
    const int c = lp ^ ((r >> 1) & 7);
    // two sources (channels [0, c1) from x, [c1, c1 + c2) from x2, same pixels): the k-tiles of x first
    const int k1 = p.c1 / BK;
    a_src[j] = wk.kt0 < k1 ? p.x + pix * p.c1 + (size_t)wk.kt0 * BK + c * 4 : p.x2 + pix * p.c2 + (size_t)(wk.kt0 - k1) * BK + c * 4;
    a_src2[j] = p.x2 + pix * p.c2 + c * 4;
  }
  int to_switch = p.c2 ? p.c1 / BK - wk.kt0 : 0x7fffffff;     // k-tiles until the A source changes (<= 0: already on x2)
  const float* b_src[B_INSTR];
  {
    // the n-tile may be a part of a wider PACKED tile (64 columns of a 128-row tile): k-tiles are then pack rows apart
    const int pack = p.pack_bn > BN ? p.pack_bn : BN;
    const int n_row = nt * BN;
    const float* wtile = p.w + (p.mt_per_group ? (size_t)(mt / p.mt_per_group) * p.w_group_stride : 0) +
                         ((size_t)(n_row / pack) * p.nkt + wk.kt0) * ((size_t)pack * BK) + (size_t)(n_row % pack) * BK;
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      const int r = (wave * B_INSTR + j) * 8 + lr;
      const int c = lp ^ ((r >> 1) & 7);
      b_src[j] = wtile + r * BK + c * 4;
    }
  }
  const int b_step = (p.pack_bn > BN ? p.pack_bn : BN) * BK;
#define PEANUT_DMA_TILE(stage)                                                                                   \
  {                                                                                                              \
    if (to_switch-- == 0) {                                                                                      \
      _Pragma("unroll") for (int j = 0; j < A_INSTR; ++j) a_src[j] = a_src2[j];                                  \
    }                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < A_INSTR; ++j) {                                                        \
      __builtin_amdgcn_global_load_lds((gptr_t)a_src[j], (lptr_t)((stage) + (wave * A_INSTR + j) * 256), 16, 0, 0); \
      a_src[j] += BK;                                                                                            \
    }                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < B_INSTR; ++j) {                                                        \
      __builtin_amdgcn_global_load_lds((gptr_t)b_src[j], (lptr_t)((stage) + A_FLOATS + (wave * B_INSTR + j) * 256), 16, 0, 0); \
      b_src[j] += b_step;                                                                                        \
    }                                                                                                            \
  }
#define PEANUT_DMA_LANDED_BARRIER()          \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
  __syncthreads();

  // ---- MFMA fragment coordinates ----
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const int swz = (li >> 1) & 7;               // (row >> 1) & 7 of every fragment row this lane reads
  int sw[BK / 8];                              // float offset of k-group ks inside a row, after the permutation
#pragma unroll
  for (int ks = 0; ks < BK / 8; ++ks) sw[ks] = ((ks * 2 + hi) ^ swz) * 4;
  const int a_row = (wm * TM + li) * BK;
  const int b_row = A_FLOATS + (wn * TN + li) * BK;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  PEANUT_FLUSH_DECL();

  constexpr int KH = BK / 16;   // 8-k groups per half
  f32x4 afA[KH][MI], bfA[KH][NI], afB[KH][MI], bfB[KH][NI];
#define PEANUT_LOAD_FRAGS(af, bf, stage, half)                                                          \
  _Pragma("unroll") for (int j = 0; j < KH; ++j) {                                                      \
    _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                      \
      af[j][t] = *reinterpret_cast<const f32x4*>((stage) + a_row + t * 32 * BK + sw[(half) * KH + j]);  \
    _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                      \
      bf[j][u] = *reinterpret_cast<const f32x4*>((stage) + b_row + u * 32 * BK + sw[(half) * KH + j]);  \
  }
#define PEANUT_MFMA_HALF(af, bf)                                                                        \
  _Pragma("unroll") for (int j = 0; j < KH; ++j)                                                        \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                    \
      _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                    \
        _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                  \
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t][kk], bf[j][u][kk], acc[t][u], 0, 0, 0);

  // ---- prologue: k-tile 0 into stage 0 ----
  PEANUT_DMA_TILE(smem);
  PEANUT_DMA_LANDED_BARRIER();
  PEANUT_LOAD_FRAGS(afA, bfA, smem, 0);

  // steady state, per k-tile kt: DMA k-tile kt+1 into the other stage; read fragment set B (second half of kt);
  // MFMAs on set A; wait for the DMA + barrier; read set A of kt+1; MFMAs on set B.
  int kt = 0;
  for (; kt + 1 < nk; ++kt) {
    float* const nxt = smem + ((kt + 1) & 1) * STAGE;
    const float* const cur = smem + (kt & 1) * STAGE;
    PEANUT_DMA_TILE(nxt);
    PEANUT_LOAD_FRAGS(afB, bfB, cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    PEANUT_MFMA_HALF(afA, bfA);
    __builtin_amdgcn_sched_barrier(0);
    PEANUT_DMA_LANDED_BARRIER();
    PEANUT_LOAD_FRAGS(afA, bfA, nxt, 0);
    __builtin_amdgcn_sched_barrier(0);
    PEANUT_MFMA_HALF(afB, bfB);
    __builtin_amdgcn_sched_barrier(0);
    PEANUT_FLUSH_STEP();
  }
  const ResPrefetch rp = conv_res_prefetch<BM, BN, EP, 256>(p, wk, m0, n0);   // under the last k-tile's MFMAs
  {   // last k-tile
    const float* const cur = smem + (kt & 1) * STAGE;
    PEANUT_LOAD_FRAGS(afB, bfB, cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    PEANUT_MFMA_HALF(afA, bfA);
    PEANUT_MFMA_HALF(afB, bfB);
  }
  PEANUT_FLUSH_FINISH();
  __syncthreads();
#undef PEANUT_LOAD_FRAGS
#undef PEANUT_MFMA_HALF
#undef PEANUT_DMA_TILE
#undef PEANUT_DMA_LANDED_BARRIER

  conv_epilogue<BM, BN, WM, WN, EP>(p, wk, acc, smem, m0, n0, &rp);
}

// ---------------------------------------------------------------------------------------------------------------
// 256 x 128 tiles, 8 waves (wave tile 64 x 64 as above), THREE stages of 48 KiB = 144 of the CU's 160 KiB, one
// workgroup per CU.  Same fragment reads, same swizzle, same MFMA order per wave as the 128 x 128 kernel; what changes:
// a k-tile is requested two iterations before it is read (the two-stage kernel has half an iteration of cover: 0.85 us
// at full matrix-core rate, less than a loaded HBM round trip), the operand bytes per FLOP drop by a quarter (6 instead
// of 8 LDS-DMA instructions per wave and k-tile) and one barrier serves eight waves.  For the large pointwise layers
// (see conv_pw_uses_256).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void conv_pw_glds256_kernel(const ConvKParams p) {
  constexpr int BM = 256, BN = 128, BK = 32, WM = 4, WN = 2, NT = 512, STAGES = 3;
  constexpr int TM = BM / WM, TN = BN / WN;      // 64 x 64
  constexpr int MI = TM / 32, NI = TN / 32;      // 2 x 2
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;   // 48 KiB
  constexpr int A_INSTR = BM / 64, B_INSTR = BN / 64;   // 1 KiB (8-row) DMA pieces per wave and k-tile: 4 + 2
  constexpr int CS = BN + 4;
  constexpr int SMEM_FLOATS = (STAGES * STAGE > BM * CS) ? STAGES * STAGE : BM * CS;
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const Work wk = decode_work(p);
  const int mt = wk.mt, nt = wk.nt, nk = wk.kt1 - wk.kt0;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 3, lp = lane & 7;

  const float* a_src[A_INSTR];
  const float* a_src2[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int r = (wave * A_INSTR + j) * 8 + lr;
    const int m = m0 + r;
    const int mc = m < p.M ? m : p.M - 1;
    const int b = mc / p.HoWo;
    const int rem = mc - b * p.HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const size_t pix = (size_t)b * p.H * p.W + (size_t)oy * p.stride * p.W + (size_t)ox * p.stride;
    const int c = lp ^ ((r >> 1) & 7);
    const int k1 = p.c1 / BK;
    a_src[j] = wk.kt0 < k1 ? p.x + pix * p.c1 + (size_t)wk.kt0 * BK + c * 4 : p.x2 + pix * p.c2 + (size_t)(wk.kt0 - k1) * BK + c * 4;
    a_src2[j] = p.x2 + pix * p.c2 + c * 4;
  }
  int to_switch = p.c2 ? p.c1 / BK - wk.kt0 : 0x7fffffff;
  const float* b_src[B_INSTR];
  {
    const float* wtile = p.w + (p.mt_per_group ? (size_t)(mt / p.mt_per_group) * p.w_group_stride : 0) +
                         ((size_t)nt * p.nkt + wk.kt0) * (BN * BK);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      const int r = (wave * B_INSTR + j) * 8 + lr;
      const int c = lp ^ ((r >> 1) & 7);
      b_src[j] = wtile + r * BK + c * 4;
    }
  }
#define PW256_DMA_TILE(stage)                                                                                    \
  {                                                                                                              \
    if (to_switch-- == 0) {                                                                                      \
      _Pragma("unroll") for (int j = 0; j < A_INSTR; ++j) a_src[j] = a_src2[j];                                  \
    }                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < A_INSTR; ++j) {                                                        \
      __builtin_amdgcn_global_load_lds((gptr_t)a_src[j], (lptr_t)((stage) + (wave * A_INSTR + j) * 256), 16, 0, 0); \
      a_src[j] += BK;                                                                                            \
    }                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < B_INSTR; ++j) {                                                        \
      __builtin_amdgcn_global_load_lds((gptr_t)b_src[j], (lptr_t)((stage) + A_FLOATS + (wave * B_INSTR + j) * 256), 16, 0, 0); \
      b_src[j] += BN * BK;                                                                                       \
    }                                                                                                            \
  }
#define PW256_BARRIER()                                   \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("" ::: "memory");

  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const int swz = (li >> 1) & 7;
  int sw[BK / 8];
#pragma unroll
  for (int ks = 0; ks < BK / 8; ++ks) sw[ks] = ((ks * 2 + hi) ^ swz) * 4;
  const int a_row = (wm * TM + li) * BK;
  const int b_row = A_FLOATS + (wn * TN + li) * BK;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  PEANUT_FLUSH_DECL();

  // prologue: two k-tiles in flight
  PW256_DMA_TILE(smem);
  if (nk > 1) {
    PW256_DMA_TILE(smem + STAGE);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // all but the newest tile (6 DMA instructions per wave and tile)
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  PW256_BARRIER();
  // one barrier per k-tile; the fragment reads of a k-tile are not carried across it (measured: splitting them in
  // halves around a mid-iteration barrier as the two-stage kernel does is 2 % SLOWER here -- with two waves per SIMD
  // the other wave's MFMAs already cover them, and the extra scheduling fences cost more)
  int o_cur = 0, o_mid = STAGE, o_fill = 2 * STAGE;
  // The two waves of a SIMD (waves w and w + 4) request their share of the next k-tile at DIFFERENT points of the
  // iteration: waves 0-3 at the top, waves 4-7 after their first 32 MFMAs.  Issuing the six LDS-DMA instructions costs a
  // wave several hundred cycles; with both waves of a SIMD doing it right after the barrier the matrix pipe starves at the
  // head of every k-tile.  Calibrated on the bare loop (tools/micro/pw256_loop.hip, profiles/r5a): 132-136 -> 143 TF/s
  // (no requests at all: 149); spreading the requests between the MFMAs instead: 120.
  const bool late = p.phase_shift && wave >= 4;
  // Grouped GEMM whose groups end in zero padding (Winograd positions: 3 200 tiles in 13 x 256 rows): a wave whose 64 rows are
  // all padding reads no fragments and issues no MFMAs -- its SIMD then carries one wave instead of two and the tile takes
  // about half the time.  The wave keeps requesting its share of the k-tiles and meets every barrier; its accumulators stay zero.
  const bool idle = p.group_valid > 0 && p.mt_per_group > 0 && (mt % p.mt_per_group) * BM + wm * TM >= p.group_valid;
#define PW256_MFMA_GROUP(j)                                                                               \
  _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                        \
    _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                        \
      _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                      \
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t][kk], bf[j][u][kk], acc[t][u], 0, 0, 0);
  if (idle) {
    // no fragments, no MFMAs: this wave only keeps the ring going (its share of every k-tile's requests, every barrier)
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 2 < nk;
      if (more) PW256_DMA_TILE(smem + o_fill);
      if (more) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
      else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      PW256_BARRIER();
      { const int t = o_cur; o_cur = o_mid; o_mid = o_fill; o_fill = t; }
    }
  } else {
  for (int kt = 0; kt < nk; ++kt) {
    const float* const cur = smem + o_cur;
    const bool more = kt + 2 < nk;
    if (more && !late) PW256_DMA_TILE(smem + o_fill);
    f32x4 af[BK / 8][MI], bf[BK / 8][NI];
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
#pragma unroll
      for (int t = 0; t < MI; ++t) af[j][t] = *reinterpret_cast<const f32x4*>(cur + a_row + t * 32 * BK + sw[j]);
#pragma unroll
      for (int u = 0; u < NI; ++u) bf[j][u] = *reinterpret_cast<const f32x4*>(cur + b_row + u * 32 * BK + sw[j]);
    }
    PW256_MFMA_GROUP(0);
    PW256_MFMA_GROUP(1);
    __builtin_amdgcn_sched_barrier(0);
    if (more && late) PW256_DMA_TILE(smem + o_fill);
    __builtin_amdgcn_sched_barrier(0);
    PW256_MFMA_GROUP(2);
    PW256_MFMA_GROUP(3);
    PEANUT_FLUSH_STEP();
    // the next k-tile must have landed; the one just requested may stay in flight across the barrier
    if (more) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    PW256_BARRIER();
    { const int t = o_cur; o_cur = o_mid; o_mid = o_fill; o_fill = t; }
  }
  }
#undef PW256_DMA_TILE
#undef PW256_BARRIER
#undef PW256_MFMA_GROUP
  PEANUT_FLUSH_FINISH();
  conv_epilogue<BM, BN, WM, WN, 1, NT>(p, wk, acc, smem, m0, n0);
}

// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 tiles, 8 waves as 4 x 2, wave tile 64 x 128 (128 accumulator registers), TWO stages of 64 KiB, one workgroup
// per CU.  Per wave and k-tile: 8 LDS-DMA pieces and 24 ds_read_b128 for 128 MFMAs -- two thirds of the 256 x 128 kernel's
// operand traffic per FLOP (LDS-DMA, L2 and LDS reads alike), and an iteration is twice as long, so one iteration of cover
// (6.8 us at full rate) is enough for a loaded HBM round trip.  Calibrated on the bare loop (tools/micro/pw256_loop.hip,
// profiles/r5a): 144.5 TF/s against 132-136 for the 256 x 128 loop (no requests at all: 150.8).  No room for the second
// accumulator set of the two-level accumulation: the Winograd position GEMMs stay on the 256 x 128 kernel.  The weights
// keep their 128-wide packing: a 256-wide n-tile is two consecutive packed tiles.  The A and B halves of a k-tile are
// requested half an iteration apart by the two waves of a SIMD (see conv_pw_glds256_kernel).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void conv_pw_glds256w_kernel(const ConvKParams p) {
  constexpr int BM = 256, BN = 256, BK = 32, WM = 4, WN = 2, NT = 512, STAGES = 2;
  constexpr int TM = BM / WM, TN = BN / WN;      // 64 x 128
  constexpr int MI = TM / 32, NI = TN / 32;      // 2 x 4
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;   // 64 KiB
  constexpr int A_INSTR = BM / 64, B_INSTR = BN / 64;   // 4 + 4
  constexpr int CS = BN + 4, ER = BM / WM;
  constexpr int SMEM_FLOATS = (STAGES * STAGE > ER * CS) ? STAGES * STAGE : ER * CS;
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const Work wk = decode_work(p);
  const int mt = wk.mt, nt = wk.nt, nk = wk.kt1 - wk.kt0;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 3, lp = lane & 7;

  const float* a_src[A_INSTR];
  const float* a_src2[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int r = (wave * A_INSTR + j) * 8 + lr;
    const int m = m0 + r;
    const int mc = m < p.M ? m : p.M - 1;
    const int b = mc / p.HoWo;
    const int rem = mc - b * p.HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const size_t pix = (size_t)b * p.H * p.W + (size_t)oy * p.stride * p.W + (size_t)ox * p.stride;
    const int c = lp ^ ((r >> 1) & 7);
    const int k1 = p.c1 / BK;
    a_src[j] = wk.kt0 < k1 ? p.x + pix * p.c1 + (size_t)wk.kt0 * BK + c * 4 : p.x2 + pix * p.c2 + (size_t)(wk.kt0 - k1) * BK + c * 4;
    a_src2[j] = p.x2 + pix * p.c2 + c * 4;
  }
  int to_switch = p.c2 ? p.c1 / BK - wk.kt0 : 0x7fffffff;
  const float* b_src[B_INSTR];
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int r = (wave * B_INSTR + j) * 8 + lr;      // row of the 256-wide n-tile: packed 128-tile 2 nt + (r >> 7), its row r & 127
    const int c = lp ^ ((r >> 1) & 7);
    b_src[j] = p.w + ((size_t)(2 * nt + (r >> 7)) * p.nkt + wk.kt0) * (128 * BK) + (r & 127) * BK + c * 4;
  }
#define PW256W_DMA_TILE(stage)                                                                                   \
  {                                                                                                              \
    if (to_switch-- == 0) {                                                                                      \
      _Pragma("unroll") for (int j = 0; j < A_INSTR; ++j) a_src[j] = a_src2[j];                                  \
    }                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < A_INSTR; ++j) {                                                        \
      __builtin_amdgcn_global_load_lds((gptr_t)a_src[j], (lptr_t)((stage) + (wave * A_INSTR + j) * 256), 16, 0, 0); \
      a_src[j] += BK;                                                                                            \
    }                                                                                                            \
    _Pragma("unroll") for (int j = 0; j < B_INSTR; ++j) {                                                        \
      __builtin_amdgcn_global_load_lds((gptr_t)b_src[j], (lptr_t)((stage) + A_FLOATS + (wave * B_INSTR + j) * 256), 16, 0, 0); \
      b_src[j] += 128 * BK;                                                                                      \
    }                                                                                                            \
  }
#define PW256W_BARRIER()                                  \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("" ::: "memory");

  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const int swz = (li >> 1) & 7;
  int sw[BK / 8];
#pragma unroll
  for (int ks = 0; ks < BK / 8; ++ks) sw[ks] = ((ks * 2 + hi) ^ swz) * 4;
  const int a_row = (wm * TM + li) * BK;
  const int b_row = A_FLOATS + (wn * TN + li) * BK;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  PW256W_DMA_TILE(smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PW256W_BARRIER();
  const bool late = p.phase_shift && wave >= 4;
#define PW256W_READ_PAIR(g)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                    \
    _Pragma("unroll") for (int t = 0; t < MI; ++t) af[j][t] = *reinterpret_cast<const f32x4*>(cur + a_row + t * 32 * BK + sw[(g) + j]); \
    _Pragma("unroll") for (int u = 0; u < NI; ++u) bf[j][u] = *reinterpret_cast<const f32x4*>(cur + b_row + u * 32 * BK + sw[(g) + j]); \
  }
#define PW256W_MFMA_PAIR()                                                                                \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                      \
      _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                      \
        _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                    \
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t][kk], bf[j][u][kk], acc[t][u], 0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const float* const cur = smem + (kt & 1) * STAGE;
    float* const fill = smem + ((kt + 1) & 1) * STAGE;
    const bool more = kt + 1 < nk;
    if (more && !late) PW256W_DMA_TILE(fill);
    f32x4 af[2][MI], bf[2][NI];
    PW256W_READ_PAIR(0);
    PW256W_MFMA_PAIR();
    __builtin_amdgcn_sched_barrier(0);
    if (more && late) PW256W_DMA_TILE(fill);
    __builtin_amdgcn_sched_barrier(0);
    PW256W_READ_PAIR(2);
    PW256W_MFMA_PAIR();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PW256W_BARRIER();
  }
#undef PW256W_DMA_TILE
#undef PW256W_BARRIER
#undef PW256W_READ_PAIR
#undef PW256W_MFMA_PAIR
  conv_epilogue<BM, BN, WM, WN, WM, NT>(p, wk, acc, smem, m0, n0);
}

template <int BN, int WM, int WN>
int launch_pw_t(const ConvKParams& p, float* ws, size_t ws_floats, hipStream_t stream) {
  static SlotCache slots;
  return launch_with_tail_split<decltype(&conv_pw_glds_kernel<BN, WM, WN>), 128, BN>(&conv_pw_glds_kernel<BN, WM, WN>, p, ws,
                                                                                      ws_floats, stream, &slots);
}

}  // namespace

bool conv_pw_enabled() { return opt(OPT_PW_GLDS) != 0; }

// whether the 256 x 128 kernel takes a [M x cout] pointwise layer (mt_per_group: 128-row tiles per weight group, 0 = plain)
// Measured per layer (profiles/r2t): +2-3 % on the K >= 1024 layers (layer3/4 conv1, layer4 downsample: 129 -> 133 TF/s),
// -3 % on the K = 512 ones (short k-loops: with one workgroup per CU nobody computes under a tile's epilogue), level on
// the K = 2048 Winograd GEMM of the bottleneck.  Grouped GEMMs need whole 256-row tiles per weight group.
int conv_pw_256_min_k() { return (int)opt(OPT_PW256_MINK); }
bool conv_pw_uses_256(int cout, long long M, int mt_per_group, int bn_tile, int cin) {
  const long long min_tiles = opt(OPT_PW256_MINTILES);
  return bn_tile == 128 && cin >= conv_pw_256_min_k() && mt_per_group % 2 == 0 && M * cout >= min_tiles * 256 * 128;
}

// whether the 256 x 256 kernel takes a pointwise layer: whole 256-wide n-tiles, no weight groups, one running sum (the
// Winograd position GEMMs keep the 256 x 128 kernel and its two-level accumulation).  Measured per layer at the headline
// shape (profiles/r5c): K = 768 -> N = 1024 (layer3.0 conv3 + downsample) 1.441 -> 1.386 ms, K = 1536 -> N = 2048 (layer4.0)
// 5.350 -> 5.148 ms (140.8 TF/s); the N = 512 layers LOSE 5-6 % (900 tiles over 256 CUs: 3.5 rounds) and so do the
// K = 512 -> N = 2048 ones (-3 %: 16 k-tiles per tile, and with one workgroup per CU nothing runs under the epilogue).
// Hence: at least pw256w_mink (768; 0 = off) input channels and pw256w_mintiles (1536: six rounds) tiles (options.h).
bool conv_pw_uses_256w(int cout, long long M, int mt_per_group, int bn_tile, int cin, int flush) {
  const int min_k = (int)opt(OPT_PW256W_MINK);
  const long long min_tiles = opt(OPT_PW256W_MINTILES);
  return min_k > 0 && bn_tile == 128 && cout % 256 == 0 && mt_per_group == 0 && flush == 0 && cin >= min_k &&
         ((M + 255) / 256) * (cout / 256) >= min_tiles;
}

// A 128-wide-packed layer with few input channels (<= bn64_maxk: its time is its epilogue's traffic) runs 128 x 64 tiles -- three
// workgroups per CU -- while the 128 x 128 tiling would not give every CU its two workgroups (one 240 x 240 map: layer3 conv3 is 64
// tiles of 128 x 128); from pw64_maxtiles tiles on the wider tile wins (one 720 x 720 map: 512 tiles, 65 -> 50 us; profiles/r5z).
bool conv_pw_narrow_tiles(int cin, int cout, long long M, int bn_tile, int mt_per_group) {
  (void)mt_per_group;
  if (bn_tile != 128 || cin > opt(OPT_BN64_MAXK) || cout % 128 != 0) return false;
  return ((M + 127) / 128) * (cout / 128) < opt(OPT_PW64_MAXTILES);
}

// fp32, BK = 32, 1x1, pad 0, one source (checked by the caller)
int launch_conv_pw(const ConvKParams& p, int bn_tile, float* ws, size_t ws_floats, hipStream_t stream) {
  const int phase_shift_w = opt(OPT_PW256_PHASE) != 0;
  // a handful of data rows per weight group (the PSP pyramid at batch 1): weight streaming, no LDS (gemm_skinny.hip)
  if (gemm_skinny_takes(p, bn_tile, ws ? ws_floats : 0)) return launch_gemm_skinny(p, ws, ws_floats, stream);
  // (the persistent 256 x 256 kernel first: its gate starts at 512 input channels by default, above the A-resident kernel's K = 128 / 256
  // layers; lowering pw256wp_mink hands those to it)
  if (conv_pw_uses_256wp(p.cout, p.M, p.stride, p.mt_per_group, bn_tile, p.c1, p.c2, p.flush)) {
    const int rc = launch_conv_pw256wp(p, ws, ws_floats, stream);
    if (rc != 1) return rc;
  }
  if (conv_pw_uses_ares(p.c1, p.cout, p.M, p.stride, p.c2 != 0, p.flush, bn_tile)) return launch_conv_pw_ares(p, bn_tile, stream);
  if (conv_pw_uses_256w(p.cout, p.M, p.mt_per_group, bn_tile, p.c1 + p.c2, p.flush)) {
    static SlotCache slots256w;
    ConvKParams q = p;
    q.ntiles = p.cout / 256;
    q.phase_shift = phase_shift_w;
    note_kernel("conv_pw_glds_256x256");
    return launch_with_tail_split<decltype(&conv_pw_glds256w_kernel), 256, 256, 512>(&conv_pw_glds256w_kernel, q, ws, ws_floats, stream,
                                                                                      &slots256w);
  }
  if (conv_pw_uses_256p(p.cout, p.M, p.mt_per_group, bn_tile, p.c1 + p.c2, p.flush, (long long)(p.M / p.HoWo) * p.H * p.W) && !(p.flush && p.res)) {
    const int rc = launch_conv_pw256p(p, ws, ws_floats, stream);
    if (rc != 1) return rc;            // 1: more items per workgroup than its plan table holds -> the kernels below
  }
  if (conv_pw_uses_256(p.cout, p.M, p.mt_per_group, bn_tile, p.c1 + p.c2)) {
    static SlotCache slots256;
    ConvKParams q = p;
    if (q.mt_per_group) q.mt_per_group /= 2;       // 256-row tiles per weight group
    q.phase_shift = phase_shift_w;
    if (opt(OPT_PW256_SKIP_PAD) == 0) q.group_valid = 0;
    note_kernel("conv_pw_glds_256x128");
    return launch_with_tail_split<decltype(&conv_pw_glds256_kernel), 256, 128, 512>(&conv_pw_glds256_kernel, q, ws, ws_floats, stream,
                                                                                     &slots256);
  }
  if (conv_pw_narrow_tiles(p.c1 + p.c2, p.cout, p.M, bn_tile, p.mt_per_group)) {      // 64-wide tiles over 128-wide packing
    ConvKParams q = p;
    q.pack_bn = 128;
    q.ntiles = p.ntiles * 2;
    note_kernel("conv_pw_glds_128x64");
    return launch_pw_t<64, 2, 2>(q, ws, ws_floats, stream);
  }
  note_kernel(bn_tile == 128 ? "conv_pw_glds_128x128" : (bn_tile == 64 ? "conv_pw_glds_128x64" : "conv_pw_glds_128x32"));
  if (bn_tile == 128) return launch_pw_t<128, 2, 2>(p, ws, ws_floats, stream);
  if (bn_tile == 64) return launch_pw_t<64, 2, 2>(p, ws, ws_floats, stream);
  if (bn_tile == 32) return launch_pw_t<32, 4, 1>(p, ws, ws_floats, stream);
  return fail(-2, "launch_conv_pw: unsupported tile configuration");
}

}  // namespace peanut
