// Short-K pointwise convs and Winograd position GEMMs (K = 128 or 256 input channels) as a PERSISTENT, A-RESIDENT kernel.
//
// Why: the expanding 1x1 convs of a Bottleneck (resnet.py:267-307 conv3: K = C/4 -> N = C, + identity, ReLU) and the
// position GEMMs of the narrow Winograd layers are 4-8 k-tiles long.  On the tile-per-workgroup kernels of conv_pw.hip
// every 128 x 64 tile pays a pipeline fill, an epilogue with nothing under it and a workgroup launch per 6.8 us of
// matrix-core work: the family runs at 96 TF/s with the matrix pipes busy 0.635 of the cycles (profiles/r4o), and a
// layer's time is its MFMA time PLUS its HBM time.
//
// Here one workgroup per CU (8 waves, wave tile 64 x 32) walks a contiguous range of (m-tile, n-tile) units, n fastest:
//   * the 128 x K A tile of the current m-tile stays in LDS (K / 32 slices of 16 KiB) while the n-tiles pass by; when
//     the last n-tile of an m-tile has read a slice for the last time, that slice is refilled IN PLACE with the next
//     m-tile's, so the A stream never stops either;
//   * only the weights stream: one 128 x 32 k-tile (16 KiB) per iteration into a two-stage ring, requested a full
//     iteration ahead, across unit boundaries -- no per-tile prologue;
//   * the epilogue of unit u runs INSIDE the k-loop of unit u + 1, from registers: the finished accumulators are copied
//     to a second register set, and every iteration finishes 32 / (K/32) of them per lane -- residual loads at the top
//     of the iteration (hidden from hipcc's wait-count bookkeeping: inline asm, cdna_hip_programming.md 5.7 form (ii)),
//     scale / shift / add / ReLU and the stores after the MFMAs.  A lane's 32 values are rows x one column, a wave
//     instruction covers two rows x 32 consecutive channels = two whole 128-byte lines.
//   * the two waves of a SIMD request their LDS-DMA pieces half an iteration apart (see conv_pw_glds256_kernel).
// Arithmetic: the same MFMA fragment layout and k order as the other fp32 kernels, so results are bit-identical to theirs
// (two-level accumulation every 64 channels included: FLUSH).
//
// LDS: (K/32 + 2) x 16 KiB = all 160 KiB at K = 256.  vmcnt protocol per iteration and wave: residual loads (asm) and
// weight pieces (LDS-DMA) are requested at the top / in the middle, `s_waitcnt vmcnt(0)` after the MFMAs retires them
// together with the previous iteration's stores and A refill pieces; this iteration's stores and A refill pieces are
// issued AFTER that wait and travel under the next iteration.
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"

namespace peanut {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// A global load hipcc does not count (cdna_hip_programming.md 5.7, form (iii)): the destination is valid only after the
// caller's wait_uncounted(); every such load is unconditional and its result is read only below that wait, so that no
// compiler copy of a destination can sit between the load and the wait (tests/test_abi.py audits the generated code).
__device__ __forceinline__ float load_uncounted(const float* ptr) {
  float v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
  return v;
}
// The wait names every destination as an INPUT: that keeps the registers allocated to the loads until the data has landed
// even on paths where the values are not used afterwards (first unit, no previous epilogue) -- hipcc would otherwise hand a
// dead destination to something else while its load is still in flight.
template <int CH>
__device__ __forceinline__ void wait_uncounted(const float (&r)[CH], float a, float b) {
  static_assert(CH == 4 || CH == 8, "chunk of 4 or 8 values");
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (CH == 4)
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(a), "v"(b) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(a), "v"(b)
                 : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int NK, bool FLUSH>
__global__ __launch_bounds__(512) void conv_pw_ares_kernel(const ConvKParams p) {
  constexpr int BK = 32, K = NK * BK, TILE = 128 * BK;   // TILE: floats of one 128-row k-tile (16 KiB)
  constexpr int CH = 32 / NK;                            // accumulator registers of the previous unit finished per iteration
  static_assert(NK == 4 || NK == 8, "K = 128 or 256");
  __shared__ __attribute__((aligned(1024))) float smem[(NK + 2) * TILE];
  float* const bst = smem + NK * TILE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 3, lp = lane & 7;
  const int wm = wave >> 2, wn = wave & 3;               // 2 x 4 waves, wave tile 64 x 32
  const int li = lane & 31, hi = lane >> 5;
  const bool late = p.phase_shift && wave >= 4;

  // ---- this workgroup's units: a contiguous range of (m-tile, n-tile), n fastest; neighbouring ranges on one XCD ----
  const int G = gridDim.x, bid = blockIdx.x;
  const int lw = (bid & 7) * (G >> 3) + (bid >> 3);      // the launcher keeps G a multiple of 8
  const int NT = p.ntiles;
  const long long U = (long long)p.mtiles * NT;
  const int u0 = (int)(U * lw / G), u1 = (int)(U * (lw + 1) / G);
  if (u0 >= u1) return;

  // ---- lane constants of the LDS-DMA pieces: two 8-row pieces of A and of B per wave and k-tile ----
  const int pbn = p.ares_pbn;                            // n-tile the weights were packed with (64 or 128)
  unsigned a_off[2], b_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 8 + lr;
    const int c = lp ^ ((r >> 1) & 7);
    a_off[j] = r * K + c * 4;
    b_off[j] = (r / pbn) * (NK * pbn * BK) + (r % pbn) * BK + c * 4;
  }
  auto dma_a = [&](int mt, int s) {     // slice s of m-tile mt -> smem[s]
    const float* base = p.x + (size_t)mt * (128 * K) + s * BK;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + a_off[j]), (lptr_t)(smem + s * TILE + (wave * 2 + j) * 256), 16, 0, 0);
  };
  auto dma_b = [&](int mt, int nt, int kt, int stage) {
    const float* base = p.w + (p.mt_per_group ? (size_t)(mt / p.mt_per_group) * p.w_group_stride : 0) + (size_t)nt * (128 * K) + kt * (pbn * BK);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + b_off[j]), (lptr_t)(bst + stage * TILE + (wave * 2 + j) * 256), 16, 0, 0);
  };

  // ---- MFMA fragment coordinates (conv_pw.hip's swizzle) ----
  const int swz = (li >> 1) & 7;
  int sw[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) sw[ks] = ((ks * 2 + hi) ^ swz) * 4;
  const int a_row = (wm * 64 + li) * BK;
  const int b_row = (wn * 32 + li) * BK;

  f32x16 acc[2], acc2[2], prev[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; acc2[t][r] = 0.f; prev[t][r] = 0.f; }

  // ---- epilogue addressing.  Lane (wm, wn, li, hi) holds column wn*32 + li of rows wm*64 + 4*hi + rowoff(idx), idx = 0..31 the
  // linear index into its two f32x16, rowoff(idx) = 8 * (idx >> 2) + (idx & 3): four consecutive rows, then a jump of eight.
  // Residual and output are walked by running per-lane pointers.  Without a residual the loads read the zero page (stride 0).
  const size_t lane_el = (size_t)(wm * 64 + 4 * hi) * p.cout + wn * 32 + li;
  const bool has_res = p.res != nullptr;
  const size_t rstride = has_res ? (size_t)p.cout : 0;
  const size_t ostride = (size_t)p.cout;
  const float* res_run = has_res ? p.res + lane_el : p.zeros;    // previous unit's residual, advanced chunk by chunk
  float* out_run = p.y + lane_el;
  const float* prev_sc = p.scale + wn * 32 + li;
  const float* prev_sh = p.shift + wn * 32 + li;
  bool prev_valid = false;
  float scv = 1.f, shv = 0.f;
  const float alpha = p.alpha;
  const bool relu = p.relu != 0;

  // one chunk of the previous unit's epilogue, split around the MFMAs: loads at the top, the rest after wait_uncounted()
#define ARES_CHUNK_LOADS(kt)                                                              \
  float resv[CH];                                                                         \
  if ((kt) == 0) {                                                                        \
    scv = load_uncounted(prev_sc);                                                        \
    shv = load_uncounted(prev_sh);                                                        \
  }                                                                                       \
  _Pragma("unroll") for (int i = 0; i < CH; ++i)                                          \
    resv[i] = load_uncounted(res_run + (size_t)(8 * (i >> 2) + (i & 3)) * rstride);       \
  res_run += (size_t)(2 * CH) * rstride;
#define ARES_CHUNK_FINISH(kt)                                                             \
  if (prev_valid) {                                                   \
    _Pragma("unroll") for (int i = 0; i < CH; ++i) {                                      \
      const int idx = (kt) * CH + i;                                                      \
      float v = prev[idx >> 4][idx & 15] * (scv * alpha) + shv;                           \
      v += resv[i];                                                                       \
      if (relu) v = relu_keep_nan(v);                                                     \
      out_run[(size_t)(8 * (i >> 2) + (i & 3)) * ostride] = v;                            \
    }                                                                                     \
  }                                                                                       \
  out_run += (size_t)(2 * CH) * ostride;

  // ---- prologue: the whole A tile of the first m-tile and the first weight k-tile ----
  {
    const int mt = u0 / NT, nt = u0 - mt * NT;
#pragma unroll
    for (int s = 0; s < NK; ++s) dma_a(mt, s);
    dma_b(mt, nt, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  int stage = 0;
  for (int u = u0; u < u1; ++u) {
    const int mt = u / NT, nt = u - mt * NT;
    const bool nxt = u + 1 < u1;
    const int mt_n = nxt ? (u + 1) / NT : mt;
    const int nt_n = nxt ? (u + 1) - mt_n * NT : 0;
    const bool refill_next = nxt && mt_n != mt;     // this unit is the last reader of its m-tile's slices: refill them for the next

    static_for<NK>([&](auto kc) {
      constexpr int kt = decltype(kc)::value;
      const float* const cur_b = bst + stage * TILE;
      const float* const cur_a = smem + kt * TILE;
      const bool more = (kt + 1 < NK) || nxt;
      // -- top: residual loads of this iteration's chunk of the previous unit; the next weight k-tile --
      ARES_CHUNK_LOADS(kt)
      auto request = [&]() {
        if (more) {
          if (kt + 1 < NK) dma_b(mt, nt, kt + 1, stage ^ 1);
          else dma_b(mt_n, nt_n, 0, stage ^ 1);
        }
      };
      if (!late) request();
      __builtin_amdgcn_sched_barrier(0);
      // -- 32 MFMAs: four 8-k groups of (2 A fragments, 1 B fragment) --
      f32x4 af[4][2], bf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        af[j][0] = *reinterpret_cast<const f32x4*>(cur_a + a_row + sw[j]);
        af[j][1] = *reinterpret_cast<const f32x4*>(cur_a + a_row + 32 * BK + sw[j]);
        bf[j] = *reinterpret_cast<const f32x4*>(cur_b + b_row + sw[j]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t][kk], bf[j][kk], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (late) request();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 2; j < 4; ++j)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t][kk], bf[j][kk], acc[t], 0, 0, 0);
      if (FLUSH && (kt & 1)) {      // partial sums of 64 channels (conv_common.h: PEANUT_FLUSH_*)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc2[t] += acc[t];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        }
      }
      // -- bottom: everything this wave has requested has landed; finish the chunk; refill the slice the PREVIOUS iteration
      // read last (it is free since the barrier that ended that iteration; the refill lands under the next iteration and is
      // retired by its wait, two iterations before the slice is read again) --
      wait_uncounted<CH>(resv, scv, shv);
      ARES_CHUNK_FINISH(kt)
      if (kt >= 1) { if (refill_next) dma_a(mt_n, kt - 1); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stage ^= 1;
    });

    // unit finished: its accumulators become the "previous" set, the epilogue pointers move to its tile
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      prev[t] = FLUSH ? acc2[t] : acc[t];
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; acc2[t][r] = 0.f; }
    }
    {
      const size_t unit_el = (size_t)mt * 128 * p.cout + (size_t)nt * 128 + lane_el;
      res_run = has_res ? p.res + unit_el : p.zeros;
      out_run = p.y + unit_el;
      const int ss_off = (p.mt_per_group && p.ss_group_stride) ? (mt / p.mt_per_group) * p.ss_group_stride : 0;
      prev_sc = p.scale + ss_off + nt * 128 + wn * 32 + li;
      prev_sh = p.shift + ss_off + nt * 128 + wn * 32 + li;
      prev_valid = true;
    }
    // the last slice of a drained m-tile is free only now (the barrier above ended its last read)
    if (refill_next) dma_a(mt_n, NK - 1);
  }

  // ---- drain: the last unit's epilogue ----
  static_for<NK>([&](auto kc) {
    constexpr int kt = decltype(kc)::value;
    ARES_CHUNK_LOADS(kt)
    wait_uncounted<CH>(resv, scv, shv);
    ARES_CHUNK_FINISH(kt)
  });
#undef ARES_CHUNK_LOADS
#undef ARES_CHUNK_FINISH
}

template <int NK, bool FLUSH>
int launch_ares_t(ConvKParams p, hipStream_t stream) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
    return fail(-3, "conv_pw_ares: no current device");
  const long long U = (long long)p.mtiles * p.ntiles;
  long long G = U < cus ? U : cus;
  G -= G % 8;
  if (G < 8) return fail(-2, "conv_pw_ares: launch too small");
  hipLaunchKernelGGL((conv_pw_ares_kernel<NK, FLUSH>), dim3((unsigned)G), dim3(512), 0, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-3, std::string("conv_pw_ares launch: ") + hipGetErrorString(e));
  return 0;
}

}  // namespace

// Which pointwise layers / grouped GEMMs take the A-resident kernel: K = 128 or 256 from one source, stride 1, whole
// 128-row and 128-column tiles (every Bottleneck conv3 and Winograd position GEMM of the prediction net at batch sizes
// that fill the chip), one running sum or partial sums of 64 channels, and at least pw_ares_minunits units (two
// per CU).  Option pw_ares = 0 switches the kernel off (options.h).
bool conv_pw_uses_ares(int cin, int cout, long long M, int stride, bool two_source, int flush_ktiles, int bn_tile) {
  const bool on = opt(OPT_PW_ARES) != 0;
  const long long min_units = opt(OPT_PW_ARES_MINUNITS);
  if (!on || two_source || (cin != 128 && cin != 256) || stride != 1 || M % 128 != 0 || cout % 128 != 0) return false;
  if (bn_tile != 64 && bn_tile != 128) return false;
  if (flush_ktiles != 0 && flush_ktiles != 2) return false;
  return (M / 128) * (cout / 128) >= min_units;
}

int launch_conv_pw_ares(const ConvKParams& p0, int bn_tile, hipStream_t stream) {
  ConvKParams p = p0;
  p.phase_shift = opt(OPT_PW256_PHASE) != 0;
  p.ares_pbn = bn_tile;
  p.mtiles = p.M / 128;
  p.ntiles = p.cout / 128;
  note_kernel("conv_pw_ares_128x128");
  const bool flush = p.flush == 2;
  if (p.c1 == 256) return flush ? launch_ares_t<8, true>(p, stream) : launch_ares_t<8, false>(p, stream);
  return flush ? launch_ares_t<4, true>(p, stream) : launch_ares_t<4, false>(p, stream);
}

}  // namespace peanut
