// Fused convolution (1x1 / 3x3, any stride / dilation) + BatchNorm(eval) + residual + ReLU as an
// im2col-free implicit GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// one rounding per product, bit-identical to an fmaf chain -- the arithmetic class of the
// reference's fp32 PyTorch path).
//
// Replaces the reference's conv -> BN -> ReLU module chains:
//   prediction/mmseg/models/backbones/resnet.py:267-307 (Bottleneck), :591-624 (deep stem),
//   prediction/mmseg/models/utils/res_layer.py:55-64 (downsample),
//   prediction/mmseg/models/decode_heads/psp_head.py:39-46,86-93 (PPM 1x1, 3x3 bottleneck),
//   prediction/mmseg/models/decode_heads/decode_head.py:225-230 (conv_seg).
//
// GEMM view:  Y[m][n] = sum_k A[m][k] * Wt[n][k],  m = (b, oy, ox) output pixel, n = out channel,
// k = (channel chunk, filter tap, channel-in-chunk).  Activations are NHWC so a k-tile of A is, per
// output pixel, one contiguous BK*4-byte segment of the (shifted) input pixel; out-of-image taps
// are zero-filled while staging, so nothing like an im2col buffer ever exists in HBM.
//
// Tiling (wave64, 4 waves / workgroup):
//   block tile BM x BN (128 x {128,64,32}), k-tile BK (32; 16 for the 14->16-channel stem conv).
//   LDS holds A[BM][BK+4] and W[BN][BK+4] (row pad = one 16-B slot -> conflict-free ds_read_b128),
//   double-buffered; the next k-tile travels HBM/L2 -> VGPR while the current one feeds the MFMAs,
//   and is written to the other LDS buffer after one barrier per k-tile.
//   Each lane fetches 4 consecutive k of its row with ONE ds_read_b128 (lanes 0-31 take k 0..3 of
//   an 8-k group, lanes 32-63 take k 4..7), which feeds 4 MFMA k-steps: the k order inside the
//   sum is permuted, A and B consistently, which fp32 addition order tolerance covers.
//   Accumulator map (guide sec. 3): col = lane&31 -> out channel (contiguous in NHWC -> 128-B store
//   segments), row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> pixel.
//   Workgroup -> tile map is XCD-aware: each XCD (private 4 MiB L2) owns a contiguous run of
//   tiles, n-tile fastest, so the co-resident workgroups of an XCD share A panels.
#include <stdlib.h>

#include "conv_common.h"

namespace peanut {

// Software-pipelined staging, split in two so that address arithmetic never sits between a load and its
// issue slot:
//   prep_addr   -- addresses of the NEXT k-tile to be loaded (A gathered from the shifted input pixels, W
//                  linear) + iterator advance.  Pure ALU, scheduled anywhere in the MFMA shadow.
//   issue_loads -- the global loads themselves, from addresses computed one iteration earlier, so they go
//                  out at the top of the iteration and have a whole MFMA phase to land.
// Branch-free on purpose: out-of-image taps / rows read a 16-byte zero page instead of being predicated
// (arithmetic select of the address -- a ?: on pointers is lowered to exec-masked code) and the iterator
// advances with selects, so the steady-state k-loop body is ONE basic block.
template <int BN, int BK, int A_PER>
__device__ __forceinline__ void prep_addr(const ConvKParams& p, KIter& it, const int (&a_iy0)[A_PER],
                                          const int (&a_ix0)[A_PER], const int (&a_pix)[A_PER], int a_c4,
                                          unsigned long long (&a_addr)[A_PER], const float*& b_tile) {
  const bool second = it.cbase >= p.c1;
  const float* src = second ? p.x2 : p.x;
  const int C = second ? p.c2 : p.c1, cb = second ? it.cbase - p.c1 : it.cbase;
  const int dy = it.ky * p.dil, dx = it.kx * p.dil;
  static_for<A_PER>([&](auto J) __attribute__((always_inline)) {
    constexpr int j = decltype(J)::value;
    const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
    const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const float* ptr = src + (size_t)(a_pix[j] + iy * p.W + ix) * C + cb + a_c4;
    const unsigned long long m = ok ? ~0ull : 0ull;
    a_addr[j] = ((unsigned long long)ptr & m) | ((unsigned long long)p.zeros & ~m);
  });
  b_tile = it.wtile;
  it.wtile += BN * BK;
  const int tap1 = it.tap + 1, kx1 = it.kx + 1;
  const bool wrap = tap1 == p.ntaps, kxw = kx1 == p.kw;
  it.tap = wrap ? 0 : tap1;
  it.ky = wrap ? 0 : (kxw ? it.ky + 1 : it.ky);
  it.kx = (wrap || kxw) ? 0 : kx1;
  it.cbase += wrap ? BK : 0;
}

template <int BN, int BK, int A_PER, int B_PER>
__device__ __forceinline__ void issue_loads(const ConvKParams& p, const unsigned long long (&a_addr)[A_PER],
                                            const float* b_tile, int tid, f32x4 (&ra)[A_PER], f32x4 (&rb)[B_PER]) {
  constexpr int B_F4 = BN * (BK / 4);
  static_for<A_PER>([&](auto J) __attribute__((always_inline)) {
    constexpr int j = decltype(J)::value;
    // address_space(1): a global_load, not a flat_load (flat loads also count on lgkmcnt and would force the
    // LDS fragment reads to be waited for in order with them)
    ra[j] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(a_addr[j]);
  });
  static_for<B_PER>([&](auto J) __attribute__((always_inline)) {
    constexpr int j = decltype(J)::value;
    const int idx = tid + 256 * j;
    if constexpr (B_F4 % 256 == 0) {
      rb[j] = *reinterpret_cast<const f32x4*>(b_tile + idx * 4);
    } else {
      rb[j] = *reinterpret_cast<const f32x4*>(idx < B_F4 ? b_tile + idx * 4 : p.zeros);
    }
  });
}

template <int BM, int BN, int BK, int A_PER, int B_PER>
__device__ __forceinline__ void store_tiles(float* stage, int tid, const f32x4 (&ra)[A_PER],
                                            const f32x4 (&rb)[B_PER]) {
  constexpr int KV = BK / 4, LS = BK + 4, A_F4 = BM * KV, B_F4 = BN * KV;
  static_for<A_PER>([&](auto J) __attribute__((always_inline)) {
    constexpr int j = decltype(J)::value;
    const int idx = tid + 256 * j;
    if (A_F4 % 256 == 0 || idx < A_F4)
      *reinterpret_cast<f32x4*>(stage + (idx / KV) * LS + (idx % KV) * 4) = ra[j];
  });
  static_for<B_PER>([&](auto J) __attribute__((always_inline)) {
    constexpr int j = decltype(J)::value;
    const int idx = tid + 256 * j;
    if (B_F4 % 256 == 0 || idx < B_F4)
      *reinterpret_cast<f32x4*>(stage + BM * LS + (idx / KV) * LS + (idx % KV) * 4) = rb[j];
  });
}

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvKParams p) {
  constexpr int KV = BK / 4;            // f32x4 per tile row
  constexpr int LS = BK + 4;            // LDS row stride (floats)
  constexpr int A_F4 = BM * KV, B_F4 = BN * KV;
  constexpr int A_PER = (A_F4 + 255) / 256, B_PER = (B_F4 + 255) / 256;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int STAGE = (BM + BN) * LS;  // floats per LDS stage
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile is a multiple of the 32x32 MFMA");

  constexpr int CS = BN + 4;             // epilogue staging row stride (floats)
  // epilogue passes: when the pipeline buffers are smaller than a whole staged tile (BK = 16), the tile is
  // staged in WM row-slabs (one per wave row) so that LDS -- hence workgroups per CU -- is set by the pipeline
  constexpr int EP = (BM * CS > 2 * STAGE) ? WM : 1;
  constexpr int ER = BM / EP;
  constexpr int SMEM_FLOATS = (2 * STAGE > ER * CS) ? 2 * STAGE : ER * CS;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const Work wk = decode_work(p);
  const int mt = wk.mt, nt = wk.nt, nk = wk.kt1 - wk.kt0;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread staging coordinates ----
  int a_iy0[A_PER], a_ix0[A_PER], a_pix[A_PER];
  const int a_c4 = (tid % KV) * 4;
#pragma unroll
  for (int j = 0; j < A_PER; ++j) {
    const int idx = tid + 256 * j;
    const int row = idx / KV;
    const int m = m0 + row;
    if (row < BM && m < p.M) {
      const int b = m / p.HoWo;
      const int rem = m - b * p.HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[j] = oy * p.stride - p.pad;
      a_ix0[j] = ox * p.stride - p.pad;
      a_pix[j] = b * p.H * p.W;
    } else {
      a_iy0[j] = -(1 << 28);  // fails every bounds test -> zero row
      a_ix0[j] = 0;
      a_pix[j] = 0;
    }
  }

  f32x4 ra[A_PER], rb[B_PER];
  KIter it;
  it.tap = wk.kt0 % p.ntaps;                       // channel chunk outer, filter tap inner
  it.cbase = (wk.kt0 / p.ntaps) * BK;
  it.ky = it.tap / p.kw;
  it.kx = it.tap - it.ky * p.kw;
  it.wtile = p.w + (p.mt_per_group ? (size_t)(mt / p.mt_per_group) * p.w_group_stride : 0) + ((size_t)nt * p.nkt + wk.kt0) * (BN * BK);

  unsigned long long a_addr[A_PER];
  const float* b_tile;
#define PEANUT_PREP_ADDR() prep_addr<BN, BK, A_PER>(p, it, a_iy0, a_ix0, a_pix, a_c4, a_addr, b_tile)
#define PEANUT_ISSUE_LOADS() issue_loads<BN, BK, A_PER, B_PER>(p, a_addr, b_tile, tid, ra, rb)
#define PEANUT_STORE_TILES(stage) store_tiles<BM, BN, BK, A_PER, B_PER>(stage, tid, ra, rb)

  // ---- MFMA fragment coordinates ----
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const int a_off = (wm * TM + li) * LS + hi * 4;
  const int b_off = BM * LS + (wn * TN + li) * LS + hi * 4;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  // ---- pipeline prologue ----
  PEANUT_PREP_ADDR();
  PEANUT_ISSUE_LOADS();              // k-tile 0
  PEANUT_STORE_TILES(smem);
  PEANUT_PREP_ADDR();
  if (nk > 1) PEANUT_ISSUE_LOADS();  // k-tile 1
  PEANUT_PREP_ADDR();                // addresses of k-tile 2
  __syncthreads();

  // MFMA fragments of one k-tile are read in two halves (KH groups of 8 k each) into two register sets so that
  // the ds_reads of one half are in flight while the matrix cores work on the other:
  //   set A = first half of the current k-tile (read during the previous k-tile's second half),
  //   set B = second half (read while set A is being consumed).
  constexpr int KH = BK / 16;   // 8-k groups per half
  f32x4 afA[KH][MI], bfA[KH][NI], afB[KH][MI], bfB[KH][NI];
#define PEANUT_LOAD_FRAGS(af, bf, stage, half)                                                           \
  _Pragma("unroll") for (int j = 0; j < KH; ++j) {                                                       \
    _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                       \
      af[j][t] = *reinterpret_cast<const f32x4*>((stage) + a_off + t * 32 * LS + ((half) * KH + j) * 8); \
    _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                       \
      bf[j][u] = *reinterpret_cast<const f32x4*>((stage) + b_off + u * 32 * LS + ((half) * KH + j) * 8); \
  }
#define PEANUT_MFMA_HALF(af, bf)                                                                         \
  _Pragma("unroll") for (int j = 0; j < KH; ++j)                                                         \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                     \
      _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                     \
        _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                   \
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t][kk], bf[j][u][kk], acc[t][u], 0, 0, 0);

  PEANUT_LOAD_FRAGS(afA, bfA, smem, 0);
  // steady state (no conditionals inside).  Per k-tile kt:
  //   registers -> LDS[nxt] (k-tile kt+1), issue the global loads of k-tile kt+2, issue the reads of fragment set B;
  //   first-half MFMAs on set A with the address arithmetic of k-tile kt+3 in their shadow;
  //   barrier (LDS[nxt] complete, LDS[cur] no longer needed); issue the reads of set A for k-tile kt+1;
  //   second-half MFMAs on set B.
  int kt = 0;
  for (; kt + 2 < nk; ++kt) {
    float* const nxt = smem + ((kt + 1) & 1) * STAGE;
    const float* const cur = smem + (kt & 1) * STAGE;
    PEANUT_STORE_TILES(nxt);
    PEANUT_ISSUE_LOADS();
    PEANUT_LOAD_FRAGS(afB, bfB, cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    PEANUT_PREP_ADDR();
    PEANUT_MFMA_HALF(afA, bfA);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    PEANUT_LOAD_FRAGS(afA, bfA, nxt, 0);
    __builtin_amdgcn_sched_barrier(0);
    PEANUT_MFMA_HALF(afB, bfB);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kt + 1 < nk) {   // second-to-last k-tile: nothing left to load from global memory
    float* const nxt = smem + ((kt + 1) & 1) * STAGE;
    const float* const cur = smem + (kt & 1) * STAGE;
    PEANUT_STORE_TILES(nxt);
    PEANUT_LOAD_FRAGS(afB, bfB, cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    PEANUT_MFMA_HALF(afA, bfA);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    PEANUT_LOAD_FRAGS(afA, bfA, nxt, 0);
    __builtin_amdgcn_sched_barrier(0);
    PEANUT_MFMA_HALF(afB, bfB);
    __builtin_amdgcn_sched_barrier(0);
    ++kt;
  }
  {                    // last k-tile
    const float* const cur = smem + (kt & 1) * STAGE;
    PEANUT_LOAD_FRAGS(afB, bfB, cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    PEANUT_MFMA_HALF(afA, bfA);
    PEANUT_MFMA_HALF(afB, bfB);
  }
  __syncthreads();
#undef PEANUT_LOAD_FRAGS
#undef PEANUT_MFMA_HALF

  conv_epilogue<BM, BN, WM, WN, EP>(p, wk, acc, smem, m0, n0);
}

// ------------------------------------------------------------------------------------------------
const float* zero_page() {
  static float* pages[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!pages[dev]) {
    float* ptr = nullptr;
    if (hipMalloc(&ptr, 4096) != hipSuccess) return nullptr;
    if (hipMemset(ptr, 0, 4096) != hipSuccess) return nullptr;
    pages[dev] = ptr;
  }
  return pages[dev];
}

void conv_pick_tiles(int cin_pad, int cout, int* bn_tile, int* bk, bool pointwise) {
  const int forced_bk = (int)opt(OPT_FP32_BK);
  *bk = (cin_pad % 32 == 0) ? 32 : 16;
  if (forced_bk == 16) *bk = 16;   // experiment knob: 3 workgroups/CU instead of 2 in the fp32 kernel
  *bn_tile = cout >= 128 ? 128 : (cout > 32 ? 64 : 32);
  // 128 x 64 tiles (48 KB LDS, three workgroups per CU instead of two) for the layers with <= 256 input channels: their
  // time goes to the epilogue's memory phases (output + residual per few k-tiles), which more resident workgroups
  // overlap better.  Measured (profiles/r2h): layer1 conv3 0.37 -> 0.30 ms, layer2 conv3 0.24 -> 0.20, layer3 conv3
  // 0.645 -> 0.573; larger K loses (the matrix-core share grows).  Option bn64_maxk overrides the threshold.
  // Pointwise layers (round 4): only up to pw_bn64_maxk (128) channels are PACKED 64 wide; a 129..256-channel layer is packed 128
  // wide and launch_conv_pw picks the 64-wide tile per shape (conv_pw_glds_kernel<64> reads halves of the 128-row packed tiles).
  const int bn64_maxk = (int)opt(pointwise ? OPT_PW_BN64_MAXK : OPT_BN64_MAXK);
  if (*bn_tile == 128 && cin_pad <= bn64_maxk) *bn_tile = 64;
}

size_t conv_packed_floats(int cin_pad, int cout, int kh, int kw, int bn_tile) {
  const int cout_pad = (cout + bn_tile - 1) / bn_tile * bn_tile;
  return (size_t)cout_pad * kh * kw * cin_pad;
}

// w_oihw [cout][cin_real][kh][kw]  ->  [ntile][ktile][BN][BK], ktile = chunk * ntaps + tap
void pack_conv_weights(const float* w, int cout, int cin_real, int cin_pad, int kh, int kw, int bn_tile,
                       int bk, float* out) {
  const int ntaps = kh * kw;
  const int cout_pad = (cout + bn_tile - 1) / bn_tile * bn_tile;
  const int nchunks = cin_pad / bk;
  const int nkt = nchunks * ntaps;
  for (int n = 0; n < cout_pad; ++n) {
    const int nt = n / bn_tile, nn = n % bn_tile;
    for (int ch = 0; ch < nchunks; ++ch)
      for (int tap = 0; tap < ntaps; ++tap) {
        float* dst = out + (((size_t)nt * nkt + (size_t)ch * ntaps + tap) * bn_tile + nn) * bk;
        for (int c = 0; c < bk; ++c) {
          const int ci = ch * bk + c;
          dst[c] = (n < cout && ci < cin_real) ? w[((size_t)n * cin_real + ci) * ntaps + tap] : 0.f;
        }
      }
  }
}

template <int BM, int BN, int BK, int WM, int WN>
static int launch_t(const ConvKParams& p, float* ws, size_t ws_floats, hipStream_t stream) {
  static SlotCache slots;
  return launch_with_tail_split<decltype(&conv_igemm_kernel<BM, BN, BK, WM, WN>), BM, BN>(
      &conv_igemm_kernel<BM, BN, BK, WM, WN>, p, ws, ws_floats, stream, &slots);
}

int launch_conv(const ConvDesc& d, const ConvArgs& a, hipStream_t stream) {
  if (a.c1 + a.c2 != d.cin) return fail(-2, "launch_conv: c1 + c2 != cin");
  // gemm_rs.hip reads two sources at one pixel stride only for stride 1; a strided two-source pointwise layer of an emulated
  // mode runs on the fp32 MFMA kernels instead (its fp32-packed weights are always uploaded; exact fp32: nothing is lost)
  const bool rs_fallback = d.rs == 1 && a.c2 != 0 && d.stride != 1;
  const int kgran = (d.rs && !rs_fallback) ? 16 : d.bk;     // k-tile of the kernel that will run
  if (a.c1 % kgran != 0 || (a.c2 % kgran) != 0) return fail(-2, "launch_conv: channel split not a multiple of the k-tile");
  ConvKParams p{};
  p.x = a.x; p.x2 = a.x2 ? a.x2 : a.x; p.w = d.w_packed; p.scale = d.scale; p.shift = d.shift;
  p.res = a.res; p.y = a.y;
  p.zeros = zero_page();
  if (!p.zeros) return fail(-3, "launch_conv: zero page allocation failed");
  p.H = a.H; p.W = a.W; p.c1 = a.c1; p.c2 = a.c2; p.Ho = a.Ho; p.Wo = a.Wo; p.cout = d.cout;
  p.kw = d.kw; p.ntaps = d.kh * d.kw; p.stride = d.stride; p.pad = d.pad; p.dil = d.dil; p.relu = d.relu;
  p.HoWo = a.Ho * a.Wo;
  const long long M = (long long)a.B * p.HoWo;
  if (M <= 0 || M > 0x7fffffffLL || (long long)a.B * a.H * a.W > 0x7fffffffLL)
    return fail(-2, "launch_conv: problem size out of range");
  p.M = (int)M;
  p.nkt = (d.cin / d.bk) * p.ntaps;
  p.ntiles = d.cout_pad / d.bn_tile;
  p.n_full = 0; p.n_sp = 0; p.split_p = 1; p.partial = nullptr;
  p.alpha = 1.f;
  p.flush = (d.rs && !rs_fallback) ? 0 : d.flush_ch / d.bk;   // k-tiles per partial sum (conv_pw.hip; every other kernel keeps one running sum)
  p.mt_per_group = a.mt_per_group; p.w_group_stride = (long long)a.w_group_stride; p.ss_group_stride = a.ss_group_stride;
  p.group_valid = a.mt_per_group ? a.group_valid_rows : 0;
  if (a.defer) a.defer->valid = false;      // set by launch_with_tail_split alone, when it left its partial tiles unsummed
  p.defer = a.defer;
  p.group_rows = a.mt_per_group ? a.group_rows : nullptr;
  if (d.rs && !rs_fallback) {   // emulated-fp32 GEMM on the bf16 matrix cores, fp32 activations split in registers
    if (!d.w_s) return fail(-2, "launch_conv: register-split layer without pre-split weights");
    p.w = static_cast<const float*>(d.w_s);
    p.nkt = (d.cin / 16) * p.ntaps;
    p.alpha = d.s_alpha;
    if (d.rs == 2) return launch_conv_rs(p, d.bn_tile, d.s_planes, a.ws, a.ws_floats, stream);
    return launch_gemm_rs(p, d.bn_tile, d.s_planes, a.ws, a.ws_floats, stream);
  }
  if (d.bk == 32 && p.ntaps == 1 && p.pad == 0 && p.c1 % 32 == 0 && p.c2 % 32 == 0 && (p.c2 == 0 || p.stride == 1) && conv_pw_enabled())
    return launch_conv_pw(p, d.bn_tile, a.ws, a.ws_floats, stream);
  if (conv_patch_eligible(d, a)) return launch_conv_patch(p, d, a.B, stream);
  {
    static const char* const names[2][3] = {{"conv_igemm_128x128x32", "conv_igemm_128x64x32", "conv_igemm_128x32x32"},
                                            {"conv_igemm_128x128x16", "conv_igemm_128x64x16", "conv_igemm_128x32x16"}};
    note_kernel(names[d.bk == 32 ? 0 : 1][d.bn_tile == 128 ? 0 : (d.bn_tile == 64 ? 1 : 2)]);
  }
  if (d.bk == 32) {
    if (d.bn_tile == 128) return launch_t<128, 128, 32, 2, 2>(p, a.ws, a.ws_floats, stream);
    if (d.bn_tile == 64) return launch_t<128, 64, 32, 2, 2>(p, a.ws, a.ws_floats, stream);
    if (d.bn_tile == 32) return launch_t<128, 32, 32, 4, 1>(p, a.ws, a.ws_floats, stream);
  } else if (d.bk == 16) {
    if (d.bn_tile == 128) return launch_t<128, 128, 16, 2, 2>(p, a.ws, a.ws_floats, stream);
    if (d.bn_tile == 64) return launch_t<128, 64, 16, 2, 2>(p, a.ws, a.ws_floats, stream);
    if (d.bn_tile == 32) return launch_t<128, 32, 16, 4, 1>(p, a.ws, a.ws_floats, stream);
  }
  return fail(-2, "launch_conv: unsupported tile configuration");
}

}  // namespace peanut
