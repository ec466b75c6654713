// fp32 GEMM emulated on the bf16 matrix cores with fp32 activations: the A operand stays fp32 in HBM and in LDS and is
// split into bf16 pieces IN REGISTERS, after the fragment read ("RS" = register split); only the weights are pre-split.
//
// Why (round 3): round 2 kept activations as pre-split bf16 pieces ("S" format, 6 bytes per value, written by the
// producer next to the fp32 tensor).  That k-loop was nothing but LDS-DMA + fragment reads + MFMAs, but the forward then
// moved 81.6 GB per batch-32 step against 58.9 GB in fp32 mode, and its expanding 1x1 convs (fp32 output + S copy +
// residual: 14-15.5 bytes per output element) were HBM-bound as a pipeline.  Here the data flow is EXACTLY the fp32
// mode's (same tensors, same Winograd transforms, same fused conv3 + downsample layers, same epilogue); what changes
// is the inner product: a k-tile of 16 channels of A travels global -> LDS as fp32 by LDS-DMA (64-byte rows, like
// conv_pw.hip), a lane reads its 8 consecutive k of a row (two ds_read_b128) and peels off the bf16 pieces
//     p0 = bf16(x), r = x - p0 (exact), p1 = bf16(r), r' = r - p1 (exact), p2 = bf16(r')      x = p0 + p1 + p2 exactly
// with v_cvt_pk_bf16_f32 (round to nearest even, two values per instruction), a shift / mask to widen a piece again and
// v_sub_f32: 5.5 VALU instructions per A element.  A wave tile of 64 (M) x 128 (N) splits 16 elements per lane and
// k-tile (88 VALU) for 48 MFMAs (v_mfma_f32_32x32x16_bf16, 32 cycles each on its SIMD): ~2 VALU per MFMA, inside the
// <= 5 issue slots a bf16 MFMA hides (MI355X_MICROARCH.md, cycle constants).  The split is repeated by the WN waves that
// share an A row block -- which is why the wave tile is wide in N, not in M.
//
// Product terms: NP = 3 -> hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi (bf16x6, dropped terms
// <= 2^-24 |ab|: fp32-class), NP = 2 -> hi*hi + hi*lo + lo*hi (bf16x3), smallest terms first, fp32 accumulation.
// KIND = RS_FP16X3 (rs_common.h): the NP = 2 loop on FP16 pieces and v_mfma_f32_32x32x16_f16 -- 2 x 11 significand bits,
// fp32-class results from three products as long as the activations stay inside fp16's exponent range; the weights are
// scaled per layer before the split and ConvKParams::alpha undoes it in the epilogue.  Same cost as bf16x3 (1 520 vs
// 1 103 maps/s for bf16x6, profiles/r3r); on real data 7 % slower than the bf16 three-product kernel and equal on zeros:
// wider significands toggle more, and the launch is power-bound like the six-product one.
//
// LDS image of a stage: A [BM rows][16 floats], 16-byte chunk c of row r stored at chunk position c ^ ((r >> 2) & 3)
// (applied to the DMA source address and to the ds_read_b128 address alike): the 16 lanes of a ds_read_b128 lane group
// then hit 16 distinct 16-byte bank groups (measured: SQ_LDS_BANK_CONFLICT = 0).  B: the weights' pieces, packed once at
// load time as [n-tile][k-tile][plane][rows][16 bf16] (pack_weights_sx; a 256-wide n-tile is two adjacent packed 128-row
// tiles), the two 16-byte halves of a row swapped for rows with bit 3 set.  Three stages; a k-tile is requested two iterations before it is read (counted s_waitcnt vmcnt(N) + raw
// s_barrier: a __syncthreads would drain the DMA queue).
//
// Measured (MI355X, profiles/r3a-r3d): headline forward 946 (round 2: activations pre-split by their producers, "S"
// format) -> 1075-1090 maps/s; the 256 x 256 kernel sustains 195-230 TF/s fp32-equivalent on the K >= 1024 layers with
// the matrix pipes busy 58 % of the cycles at an effective 2.0 GHz.  On all-zero operands the same launch runs 27 %
// faster (242-251 TF/s): it is bound by the power / clock governor, not by its schedule -- a rotated schedule for the
// two waves of a SIMD, a four-stage software-pipelined loop (split of k-tile kt + 1 under the MFMAs of kt: 3.6 % fewer
// cycles at a 1.3 % lower clock) and spreading the LDS-DMA requests between the MFMAs all measured within 1 %, and were
// not kept.  Removing the split arithmetic altogether (wrong results) gains 11 %, removing the LDS-DMA 13 %.
//
// Replaces the same reference call sites as conv_pw.hip: resnet.py:267-307 (conv1 / conv3 / downsample), the position
// GEMMs of the Winograd form of conv2 and of psp_head.py:86-93.
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"
#include "rs_common.h"

namespace peanut {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// PACK = rows of one packed weight tile (the layer's bn_tile: 128 or 64); the n-tile of the kernel may be wider (256 = two
// packed tiles) or narrower (64 rows of a 128-row packed tile)
// (Round 4, profiles/r5q / r5y: the waves of the 256 x 256 kernel sit 41 % of their time at s_waitcnt / s_barrier; a four-stage
// variant that read TWO k-tiles per barrier and requested the next two at the top of the iteration -- half the barriers, the same
// lead time -- was slower, 1 642 -> 1 588 maps/s in fp16x3: the waits are for operands arriving, not for the barrier itself.)
template <int BM, int BN, int WM, int WN, int KIND, int PACK>
__global__ __launch_bounds__(64 * WM * WN) void gemm_rs_kernel(const ConvKParams p) {
  constexpr int NP = rs_pieces(KIND);
  constexpr int NT = 64 * WM * WN, NW = WM * WN, STAGES = 3;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int A_BYTES = BM * 64, B_BYTES = NP * BN * 32, STAGE = A_BYTES + B_BYTES;   // one k-tile = 16 channels
  constexpr int BSUB = PACK;
  constexpr int BSUB_BYTES = NP * BSUB * 32;
  constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024, PIECES = A_PIECES + B_PIECES;   // 1 KiB DMA pieces
  constexpr int PER_WAVE = (PIECES + NW - 1) / NW;
  constexpr bool EVEN = PIECES % NW == 0;
  constexpr int CS = BN + 4;
  constexpr int EP = (BM * CS * 4 > STAGES * STAGE) ? WM : 1;
  constexpr int ER = BM / EP;
  constexpr int SMEM_BYTES = (STAGES * STAGE > ER * CS * 4) ? STAGES * STAGE : ER * CS * 4;
  static_assert(TM % 32 == 0 && TN % 32 == 0 && (NP == 2 || NP == 3) && PER_WAVE <= 6, "tile configuration");
  static_assert(KIND == RS_BF16X3 || KIND == RS_BF16X6 || KIND == RS_FP16X3, "emulation kind");
  static_assert(BN % PACK == 0 || PACK % BN == 0, "the n-tile is whole packed weight tiles, or a whole fraction of one");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

  const int tid = threadIdx.x;
  const Work wk = decode_work(p);
  const int mt = wk.mt, nt = wk.nt, nk = wk.kt1 - wk.kt0;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- per-lane DMA sources.  Piece i < A_PIECES: rows 16 i .. 16 i + 15 of the A tile (lane l: row l / 4, chunk
  //      position l % 4); the rest: 32 rows of a plane of the weight tile (lane l: row l / 2, half l % 2). ----
  const unsigned char* src[PER_WAVE];
  const unsigned char* src2[PER_WAVE];     // A pieces: the same place in the second source (k-tile 0 of x2)
  unsigned step[PER_WAVE];
  const int k1 = p.c1 / 16;                // k-tiles of the first source
#pragma unroll
  for (int j = 0; j < PER_WAVE; ++j) {
    const int piece = wave * PER_WAVE + j;
    if (piece < A_PIECES) {
      const int r = piece * 16 + (lane >> 2);
      const int m = m0 + r;
      const int mc = m < p.M ? m : p.M - 1;   // rows past the end compute a valid row and are dropped
      const int b = mc / p.HoWo;
      const int rem = mc - b * p.HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const size_t pix = (size_t)b * p.H * p.W + (size_t)oy * p.stride * p.W + (size_t)ox * p.stride;
      const int c = (lane & 3) ^ ((r >> 2) & 3);
      const unsigned char* s1 = reinterpret_cast<const unsigned char*>(p.x + pix * p.c1) + (size_t)wk.kt0 * 64 + c * 16;
      const unsigned char* s2 = reinterpret_cast<const unsigned char*>(p.x2 + pix * p.c2) + c * 16;
      src[j] = wk.kt0 < k1 ? s1 : s2 + (size_t)(wk.kt0 - k1) * 64;
      src2[j] = s2;
      step[j] = 64;
    } else {
      const int pb = piece - A_PIECES;
      const int pbc = pb < B_PIECES ? pb : B_PIECES - 1;      // past the end (uneven split): clamp, never issued
      const int q = pbc / (BN / 32), r = (pbc % (BN / 32)) * 32 + (lane >> 1);
      const int nrow = nt * BN + r;                            // row of the layer's weight matrix
      const int sub = nrow / BSUB, rs = nrow % BSUB;           // packed tile, row inside it
      const int h = (lane & 1) ^ ((r >> 3) & 1);
      const unsigned char* wtile = reinterpret_cast<const unsigned char*>(p.w) +
                                   (p.mt_per_group ? (size_t)(mt / p.mt_per_group) * (size_t)p.w_group_stride : 0) +
                                   ((size_t)sub * p.nkt + wk.kt0) * BSUB_BYTES;
      src[j] = wtile + ((size_t)q * BSUB + rs) * 32 + h * 16;
      src2[j] = src[j];
      step[j] = BSUB_BYTES;
    }
  }
  int to_switch = p.c2 ? k1 - wk.kt0 : 0x7fffffff;     // k-tiles until the A source changes (<= 0: already on x2)
  // the source switch (fused conv3 + downsample layers: [t2 | x]) only concerns the A pieces
#define RS_DMA_TILE(stage)                                                                                        \
  {                                                                                                               \
    if (to_switch-- == 0) {                                                                                       \
      _Pragma("unroll") for (int j = 0; j < PER_WAVE; ++j)                                                        \
        if (wave * PER_WAVE + j < A_PIECES) src[j] = src2[j];                                                     \
    }                                                                                                             \
    _Pragma("unroll") for (int j = 0; j < PER_WAVE; ++j) {                                                        \
      const int piece = wave * PER_WAVE + j;                                                                      \
      if (EVEN || piece < PIECES)                                                                                 \
        __builtin_amdgcn_global_load_lds((gptr_t)src[j], (lptr_t)((stage) + piece * 1024), 16, 0, 0);            \
      src[j] += step[j];                                                                                          \
    }                                                                                                             \
  }
  const int my_pieces = (wave + 1) * PER_WAVE <= PIECES ? PER_WAVE : (PIECES - wave * PER_WAVE > 0 ? PIECES - wave * PER_WAVE : 0);
#define RS_WAIT_ALL_BUT_LAST_TILE()                                                \
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
#define RS_BARRIER()                                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("" ::: "memory");
#define RS_DMA_LANDED_BARRIER()                    \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
  RS_BARRIER()

  // ---- MFMA fragment coordinates: lane (li, hi) takes k = 8 hi .. 8 hi + 7 of row li of each 32-row block ----
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const int af = (li >> 2) & 3;                                    // chunk permutation of this lane's A rows
  const int a_off0 = (wm * TM + li) * 64 + (((2 * hi) ^ af) * 16);
  const int a_off1 = (wm * TM + li) * 64 + (((2 * hi + 1) ^ af) * 16);
  const int b_row = A_BYTES + (wn * TN + li) * 32 + ((hi ^ ((li >> 3) & 1)) * 16);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

#define RS_HEAD(cur)                                                                                              \
  {                                                                                                               \
    _Pragma("unroll") for (int t = 0; t < MI; ++t) {                                                              \
      const f32x4 v0 = *reinterpret_cast<const f32x4*>((cur) + a_off0 + t * 32 * 64);                             \
      const f32x4 v1 = *reinterpret_cast<const f32x4*>((cur) + a_off1 + t * 32 * 64);                             \
      split_frag<KIND>(v0, v1, ap[t]);                                                                              \
    }                                                                                                             \
    _Pragma("unroll") for (int q = 0; q < NP; ++q)                                                                \
      _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                              \
        bf[q][u] = *reinterpret_cast<const u32x4*>((cur) + q * (BN * 32) + b_row + u * 32 * 32);                  \
  }
#define RS_MFMA_ROWS(T0, T1)                                                                                      \
  _Pragma("unroll") for (int t = (T0); t < (T1); ++t)                                                             \
    _Pragma("unroll") for (int u = 0; u < NI; ++u) {                                                              \
      if constexpr (NP == 3) {                                                                                    \
        acc[t][u] = mfma_pieces<KIND>(ap[t][2], bf[0][u], acc[t][u]);                                        \
        acc[t][u] = mfma_pieces<KIND>(ap[t][0], bf[2][u], acc[t][u]);                                        \
        acc[t][u] = mfma_pieces<KIND>(ap[t][1], bf[1][u], acc[t][u]);                                        \
      }                                                                                                           \
      acc[t][u] = mfma_pieces<KIND>(ap[t][1], bf[0][u], acc[t][u]);                                          \
      acc[t][u] = mfma_pieces<KIND>(ap[t][0], bf[1][u], acc[t][u]);                                          \
      acc[t][u] = mfma_pieces<KIND>(ap[t][0], bf[0][u], acc[t][u]);                                          \
    }
  // the next k-tile must have landed; the one just requested may stay in flight across the barrier
#define RS_NEXT_TILE_BARRIER(more)                                  \
  if (more) { RS_WAIT_ALL_BUT_LAST_TILE(); RS_BARRIER(); }          \
  else { RS_DMA_LANDED_BARRIER(); }

  ResPrefetch rp;
  rp.on = false;
  u32x4 ap[MI][NP], bf[NP][NI];
  RS_DMA_TILE(smem);
  if (nk > 1) {
    RS_DMA_TILE(smem + STAGE);
    RS_WAIT_ALL_BUT_LAST_TILE();
    RS_BARRIER();
  } else {
    RS_DMA_LANDED_BARRIER();
  }
  int o_cur = 0, o_fill = 2 * STAGE, o_mid = STAGE;
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* const cur = smem + o_cur;
    const bool more = kt + 2 < nk;
    if (more) RS_DMA_TILE(smem + o_fill);
    // residual rows requested under the last k-tile's MFMAs -- not in the 232-register eight-wave kernel, where the 32
    // registers of the prefetch spill; its epilogue requests them before the accumulators go through LDS
    if (NW < 8 && kt == nk - 1) rp = conv_res_prefetch<BM, BN, EP, NT>(p, wk, m0, n0);
    RS_HEAD(cur);
    RS_MFMA_ROWS(0, MI);
    RS_NEXT_TILE_BARRIER(more);
    { const int t = o_cur; o_cur = o_mid; o_mid = o_fill; o_fill = t; }
  }
#undef RS_HEAD
#undef RS_MFMA_ROWS
#undef RS_NEXT_TILE_BARRIER
#undef RS_DMA_TILE
#undef RS_DMA_LANDED_BARRIER
#undef RS_WAIT_ALL_BUT_LAST_TILE
#undef RS_BARRIER

  conv_epilogue<BM, BN, WM, WN, EP, NT>(p, wk, acc, reinterpret_cast<float*>(smem), m0, n0, NW < 8 ? &rp : nullptr);
}

template <int BM, int BN, int WM, int WN, int KIND, int PACK>
int launch_rs_t(ConvKParams p, float* ws, size_t ws_floats, hipStream_t stream) {
  static SlotCache slots;
  p.ntiles = (p.cout + BN - 1) / BN;                // n-tiles of THIS kernel (decode_work)
  if (p.mt_per_group) p.mt_per_group = p.mt_per_group * 128 / BM;   // BM-row tiles per weight group
  return launch_with_tail_split<decltype(&gemm_rs_kernel<BM, BN, WM, WN, KIND, PACK>), BM, BN, 64 * WM * WN>(
      &gemm_rs_kernel<BM, BN, WM, WN, KIND, PACK>, p, ws, ws_floats, stream, &slots);
}


}  // namespace

// bytes of the pre-split weights of a 1x1 layer: [n-tile][k-tile of 16][plane][bn_tile][16 pieces of 16 bits];
// planes = the emulation kind (rs_common.h: RS_BF16X3 / RS_BF16X6 / RS_FP16X3)
size_t sx_packed_bytes(int cin_pad, int cout, int bn_tile, int planes) {
  const size_t ntiles = (cout + bn_tile - 1) / bn_tile;
  return ntiles * (size_t)(cin_pad / 16) * rs_pieces(planes) * bn_tile * 32;
}

float sx_pack_scale(const float* w, size_t n, int planes) { return rs_pack_scale(w, n, planes); }

// w: [cout][cin_real] fp32 (a 1x1 conv's OIHW weights, or one Winograd position of U); pieces of w * wscale
void pack_weights_sx(const float* w, int cout, int cin_real, int cin_pad, int bn_tile, int planes, float wscale, void* out) {
  unsigned short* o = static_cast<unsigned short*>(out);
  const int ntiles = (cout + bn_tile - 1) / bn_tile, nkt = cin_pad / 16, np = rs_pieces(planes);
  for (int nt = 0; nt < ntiles; ++nt)
    for (int kt = 0; kt < nkt; ++kt) {
      unsigned short* tile = o + ((size_t)nt * nkt + kt) * np * bn_tile * 16;
      for (int r = 0; r < bn_tile; ++r)
        for (int e = 0; e < 16; ++e) {
          const int n = nt * bn_tile + r, c = kt * 16 + e;
          float v = (n < cout && c < cin_real) ? w[(size_t)n * cin_real + c] * wscale : 0.f;
          for (int q = 0; q < np; ++q) tile[((size_t)q * bn_tile + r) * 16 + e] = rs_piece_host(v, planes);
        }
    }
}

// 256 x 256 tiles (one workgroup per CU) where the shape fills the chip with them and K is long enough to amortise
// a tile's epilogue, which nobody computes under with one workgroup per CU; else 128 x 128 (two per CU) / 128 x 64.
bool gemm_rs_uses_256(int cout, long long M, int mt_per_group, int bn_tile, int cin) {
  const int min_k = (int)opt(OPT_RS256_MINK);
  const long long min_tiles = opt(OPT_RS256_MINTILES);
  return cin >= min_k && bn_tile == 128 && cout % 256 == 0 && mt_per_group % 2 == 0 && M * cout >= min_tiles * 256 * 256;
}

// 64 x 64 tiles when the 128-row tiling would leave more than half of the CUs without a tile AND the k-loop is too short
// for split-K to fill them (batch-1 shapes: layer1-3 of a 240 x 240 map).  Measured (profiles/r3i): 240 x 240 map forward
// 1.46 -> 1.31 ms; with long k-loops the opposite holds -- the detector's res4 conv1 at 3 350 pixels (K = 1024: 54 tiles
// cut 4 ways along K, 16 k-tiles per workgroup) beats 212 workgroups of 64 k-tiles each: batch-1 detector 4.96 vs 5.51 ms
// -- hence the K bound.
bool gemm_rs_uses_64(int cout, long long M, int bn_tile, int cin) {
  const int max_tiles = (int)opt(OPT_RS64_MAXTILES);
  const int max_k = (int)opt(OPT_RS64_MAXK);
  const long long t128 = ((M + 127) / 128) * ((cout + bn_tile - 1) / bn_tile);
  return cout % 64 == 0 && cin <= max_k && t128 < max_tiles;
}

// family names: gemm_rs6_* (bf16, six products), gemm_rs3_* (bf16, three), gemm_rs3h_* (fp16, three)
const char* gemm_rs_kernel_name(int cout, long long M, int mt_per_group, int bn_tile, int cin, int planes) {
  static const char* const names[3][4] = {{"gemm_rs3_64x64", "gemm_rs3_256x256", "gemm_rs3_128x128", "gemm_rs3_128x64"},
                                          {"gemm_rs6_64x64", "gemm_rs6_256x256", "gemm_rs6_128x128", "gemm_rs6_128x64"},
                                          {"gemm_rs3h_64x64", "gemm_rs3h_256x256", "gemm_rs3h_128x128", "gemm_rs3h_128x64"}};
  const int k = planes - 2;
  if (k < 0 || k > 2) return "gemm_rs?";
  if (gemm_rs_uses_64(cout, M, bn_tile, cin)) return names[k][0];
  if (gemm_rs_uses_256(cout, M, mt_per_group, bn_tile, cin)) return names[k][1];
  return names[k][bn_tile == 128 ? 2 : 3];
}

namespace {
template <int KIND>
int launch_gemm_rs_kind(const ConvKParams& p, int bn_tile, float* ws, size_t ws_floats, hipStream_t stream) {
  if (gemm_rs_uses_64(p.cout, p.M, bn_tile, p.c1 + p.c2))
    return bn_tile == 128 ? launch_rs_t<64, 64, 2, 2, KIND, 128>(p, ws, ws_floats, stream) : launch_rs_t<64, 64, 2, 2, KIND, 64>(p, ws, ws_floats, stream);
  if (gemm_rs_uses_256(p.cout, p.M, p.mt_per_group, bn_tile, p.c1 + p.c2)) return launch_rs_t<256, 256, 4, 2, KIND, 128>(p, ws, ws_floats, stream);
  if (bn_tile == 128) return launch_rs_t<128, 128, 2, 2, KIND, 128>(p, ws, ws_floats, stream);
  return launch_rs_t<128, 64, 2, 2, KIND, 64>(p, ws, ws_floats, stream);
}
}  // namespace

// p.x / p.x2: fp32 A (two sources allowed), p.w: S-packed weights (bn_tile rows per packed tile), p.nkt = cin / 16
int launch_gemm_rs(const ConvKParams& p, int bn_tile, int planes, float* ws, size_t ws_floats, hipStream_t stream) {
  if (p.ntaps != 1 || p.pad != 0 || p.c1 % 16 || p.c2 % 16 || (p.c2 && p.stride != 1) || (bn_tile != 128 && bn_tile != 64) ||
      planes < RS_BF16X3 || planes > RS_FP16X3)
    return fail(-2, "launch_gemm_rs: needs a pointwise layer with 16-channel granularity and 64- or 128-row weight tiles");
  note_kernel(gemm_rs_kernel_name(p.cout, p.M, p.mt_per_group, bn_tile, p.c1 + p.c2, planes));
  if (planes == RS_BF16X6) return launch_gemm_rs_kind<RS_BF16X6>(p, bn_tile, ws, ws_floats, stream);
  if (planes == RS_FP16X3) return launch_gemm_rs_kind<RS_FP16X3>(p, bn_tile, ws, ws_floats, stream);
  return launch_gemm_rs_kind<RS_BF16X3>(p, bn_tile, ws, ws_floats, stream);
}

}  // namespace peanut
