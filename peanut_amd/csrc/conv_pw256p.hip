// conv_pw_glds256p_kernel: the 256 x 128 three-stage LDS-DMA GEMM of conv_pw.hip as a PERSISTENT kernel.
//
// conv_pw_glds256_kernel runs one tile per workgroup, one workgroup per CU: every tile pays a pipeline fill (two k-tiles
// requested, nothing to compute until the first has landed) and an epilogue under which nothing computes -- 2.4 % of the
// family's time at K >= 1024 (profiles/r4d), ~15 % at K = 512, which is why the K = 512 layers had to stay on the two-stage
// 128 x 128 kernel at 121 TF/s.  Here one workgroup per CU walks a list of work items (whole tiles, then its share of the
// tail's split-K parts -- the same plan as launch_with_tail_split):
//   * the three-stage ring runs ACROSS item boundaries: the first two k-tiles of the next item are requested during the last
//     two iterations of the current one -- no fill;
//   * the epilogue of item i runs inside the first eight iterations of item i + 1, from registers: the finished accumulators
//     are copied to a second set, each of those iterations loads the residual of eight of them at its top (asm loads hipcc
//     does not count, waited by the iteration's own `s_waitcnt vmcnt(6)`: they are older than the LDS-DMA requests), and
//     after the MFMAs applies scale / shift / residual / ReLU and stores them.  A lane's values are rows x one column; a wave
//     instruction covers two rows x 32 consecutive channels = two whole 128-byte lines;
//   * the two waves of a SIMD request their LDS-DMA pieces half an iteration apart (conv_pw_glds256_kernel).
// Same MFMA fragment layout and k order as the other fp32 kernels: bit-identical results (one running sum: the Winograd
// position GEMMs with their two-level accumulation keep conv_pw_glds256_kernel).
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"

namespace peanut {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// A global load hipcc does not count (see conv_pw_ares.hip / tools/audit_uncounted_loads.py): wave-uniform base in SGPRs +
// a 32-bit lane offset.  The leading s_nop covers the case that hipcc materialised the base through v_readfirstlane right
// before the statement (VALU-written SGPR read by a VMEM instruction: 5 wait states hipcc does not pad inside an asm).
__device__ __forceinline__ float p_load_uncounted(const float* base, unsigned byte_off) {
  float v;
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(base) : "memory");
  return v;
}
// the iteration's wait: all but the `N` youngest vector-memory operations of this wave have completed (N = 6: the LDS-DMA
// pieces requested this iteration stay in flight; N = 0 at the end of the stream).  Names every uncounted destination as an
// input so that their registers stay allocated until the data has landed.
template <int N>
__device__ __forceinline__ void p_wait(const float (&r)[8], float a, float b, float c, float d) {
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (N == 6)
    asm volatile("s_waitcnt vmcnt(6)" ::"v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(a), "v"(b),
                 "v"(c), "v"(d)
                 : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]), "v"(a), "v"(b),
                 "v"(c), "v"(d)
                 : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

struct PItem { int mt, nt, kt0, kt1, part; };   // part >= 0: split-K part -> partial tile #part

// RAGGED: the last m-tile may hang over the end of the rows (M % 256 != 0): its A rows are clamped, its stores predicated.
// FLUSH: two-level fp32 accumulation (partial sums of 64 channels: conv_common.h PEANUT_FLUSH_*) for the grouped Winograd
// position GEMMs.  A third accumulator set does not fit: the finished totals stay in the second-level set and are stored during
// the next item's first TWO iterations (32 values each; such layers have no residual), before its first flush needs the set.
template <bool RAGGED, bool FLUSH>
__global__ __launch_bounds__(512) void conv_pw_glds256p_kernel(const ConvKParams p) {
  constexpr int BM = 256, BN = 128, BK = 32, WN = 2;
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;   // 48 KiB
  constexpr int A_INSTR = 4, B_INSTR = 2;
  // ONE LDS object: the three stages and, behind them, the workgroup's plan table (a second __shared__ variable would make hipcc
  // attach alias scopes to the LDS accesses and wait `vmcnt(0)` between every LDS-DMA request and the fragment reads: conv_pw256wp.hip)
  constexpr int kMaxItems = 120;
  __shared__ __attribute__((aligned(1024))) float smem[3 * STAGE + kMaxItems * 8];
  int* const plan = reinterpret_cast<int*>(smem + 3 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 3, lp = lane & 7;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const bool late = p.phase_shift && wave >= 4;

  // ---- this workgroup's items: nf whole tiles (a contiguous run of the tile order), then its share of the split parts ----
  const int G = gridDim.x, bid = blockIdx.x;
  const int lw = (bid & 7) * (G >> 3) + (bid >> 3);                    // neighbouring runs on one XCD (G is a multiple of 8)
  const int nf = p.n_full / G;                                         // the launcher makes n_full a multiple of G
  const int sp0 = (int)((long long)p.n_sp * lw / G), sp1 = (int)((long long)p.n_sp * (lw + 1) / G);
  // stream-K tail: this workgroup's run of the tail's k-tiles, as fragments of consecutive tail tiles
  const long long skU = p.sk_units;
  const int u0 = (skU > 0 && lw < p.sk_g) ? (int)(skU * lw / p.sk_g) : 0, u1 = (skU > 0 && lw < p.sk_g) ? (int)(skU * (lw + 1) / p.sk_g) : 0;
  const int upt = skU > 0 ? p.nkt / p.sk_q : 1;           // stream units (sk_q k-tiles each) per tile
  const int n_frag = u1 > u0 ? (u1 - 1) / upt - u0 / upt + 1 : 0;
  const int n_items = nf + (skU > 0 ? n_frag : sp1 - sp0);
  if (n_items == 0) return;
  if (n_items > kMaxItems) __builtin_trap();      // the launcher bounds the plan; never write past the LDS table
  // The items are worked out ONCE, one per thread, into the LDS plan table (round 5, as in conv_pw256wp.hip): the 64-bit divisions of
  // the stream-K bookkeeping were inlined at every use of item_at -- four places -- and cost the kernel 100-170 spilled SGPRs.
  if (tid < n_items) {
    const int i = tid;
    PItem it;
    int tile;
    if (i >= nf && skU > 0) {
      const int j = u0 / upt + (i - nf);                    // tail tile of this fragment
      const int t0 = j * upt;
      it.kt0 = ((u0 > t0 ? u0 : t0) - t0) * p.sk_q;
      it.kt1 = ((u1 < t0 + upt ? u1 : t0 + upt) - t0) * p.sk_q;
      it.part = j * p.sk_maxp + (lw - sk_owner(skU, p.sk_g, (long long)t0));   // fragments of a tile in workgroup order
      tile = p.n_full + j;
    } else if (i < nf) {
      // p_order: the G/8 workgroups of an XCD take consecutive tiles of the XCD's run at every step -- co-resident workgroups
      // then share A panels (n fastest) and walk the weights together, as a tile-per-workgroup launch does; else one
      // contiguous run per workgroup (each A panel fetched once per n-tile: 2.7 x the algorithmic traffic at N = 512)
      tile = p.p_order ? (lw / (G >> 3)) * (nf * (G >> 3)) + i * (G >> 3) + lw % (G >> 3) : lw * nf + i;
      it.kt0 = 0; it.kt1 = p.nkt; it.part = -1;
    } else {
      const int s = sp0 + (i - nf);
      const int j = s / p.split_p, part = s - j * p.split_p;
      tile = p.n_full + j;
      it.part = s;
      it.kt0 = (int)((long long)part * p.nkt / p.split_p);
      it.kt1 = (int)((long long)(part + 1) * p.nkt / p.split_p);
    }
    tile_to_mn(p, tile, &it.mt, &it.nt);
    int* e = plan + i * 8;
    e[0] = it.mt; e[1] = it.nt; e[2] = it.kt0; e[3] = it.kt1; e[4] = it.part;
  }
  __syncthreads();
  auto item_at = [&](int i) {
    const int* e = plan + i * 8;
    PItem it;
    it.mt = __builtin_amdgcn_readfirstlane(e[0]);
    it.nt = __builtin_amdgcn_readfirstlane(e[1]);
    it.kt0 = __builtin_amdgcn_readfirstlane(e[2]);
    it.kt1 = __builtin_amdgcn_readfirstlane(e[3]);
    it.part = __builtin_amdgcn_readfirstlane(e[4]);
    return it;
  };

  // ---- the request cursor: runs two k-tiles ahead of the compute cursor, across item boundaries.  Per lane it holds only
  // the pixel index of its four A rows; everything else is a lane constant or wave-uniform.  A piece's address is
  //   source base (uniform) + pixel * channels * 4 + k-tile offset (uniform) + chunk constant.
  // conv_pw_uses_256p guarantees that the input (B*H*W rows of the wider source) and the output stay below 4 GiB (32-bit byte offsets).
  unsigned a_pix[A_INSTR], a_chunk[A_INSTR], b_off[B_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int r = (wave * A_INSTR + j) * 8 + lr;
    a_chunk[j] = (unsigned)((lp ^ ((r >> 1) & 7)) * 16);
    a_pix[j] = 0;
  }
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int r = (wave * B_INSTR + j) * 8 + lr;
    b_off[j] = (unsigned)(r * BK * 4 + (lp ^ ((r >> 1) & 7)) * 16);
  }
  int d_item = 0, d_left = 0, d_kt = 0;      // item of the cursor, its k-tiles not requested yet, the next k-tile (absolute)
  const float* d_wtile = p.w;                // weights of (n-tile of the cursor's item, k-tile 0)
  auto cursor_open = [&](int i) {
    const PItem it = item_at(i);
    const int m0 = it.mt * BM;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      int m = m0 + (wave * A_INSTR + j) * 8 + lr;
      if (RAGGED) m = m < p.M ? m : p.M - 1;                     // rows past the end fetch a valid row and are dropped
      const int b = m / p.HoWo;
      const int rem = m - b * p.HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_pix[j] = (unsigned)(b * p.H * p.W + oy * p.stride * p.W + ox * p.stride);
    }
    d_wtile = p.w + (p.mt_per_group ? (size_t)(it.mt / p.mt_per_group) * p.w_group_stride : 0) + (size_t)it.nt * p.nkt * (BN * BK);
    d_kt = it.kt0;
    d_left = it.kt1 - it.kt0;
  };
  const int k1 = p.c1 / BK;
  // The cursor's next k-tile into `stage`.  Once the stream has ended (d_left == 0: the last two iterations of the
  // workgroup) the last k-tile is requested AGAIN, into a stage nobody reads any more: every iteration then issues exactly
  // six LDS-DMA pieces, and its wait is one unconditional `s_waitcnt vmcnt(6)` -- no second wait arm for hipcc to merge.
  auto request = [&](float* stage) {
    const bool live = d_left > 0;
    if (!live) --d_kt;
    const bool second = d_kt >= k1;                               // two sources: k-tiles [0, k1) from x, the rest from x2
    const char* abase = reinterpret_cast<const char*>(second ? p.x2 : p.x) + (size_t)(second ? d_kt - k1 : d_kt) * (BK * 4);
    const unsigned cbytes = (unsigned)(second ? p.c2 : p.c1) * 4u;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(abase + (a_pix[j] * cbytes + a_chunk[j])), (lptr_t)(stage + (wave * A_INSTR + j) * 256), 16, 0, 0);
    const char* bbase = reinterpret_cast<const char*>(d_wtile + (size_t)d_kt * (BN * BK));
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(bbase + b_off[j]), (lptr_t)(stage + A_FLOATS + (wave * B_INSTR + j) * 256), 16, 0, 0);
    ++d_kt;
    if (live) --d_left;
  };
  // after an iteration's wait: if the cursor's item is exhausted, open the next one (pixel indices: four integer divisions
  // per lane -- kept OUT of the stretch between the uncounted loads and their wait)
  auto cursor_advance = [&]() {
    if (d_left == 0 && d_item + 1 < n_items) {
      ++d_item;
      cursor_open(d_item);
    }
  };

  // ---- MFMA fragment coordinates ----
  const int swz = (li >> 1) & 7;
  int sw[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) sw[ks] = ((ks * 2 + hi) ^ swz) * 4;
  const int a_row = (wm * 64 + li) * BK;
  const int b_row = A_FLOATS + (wn * 64 + li) * BK;

  // accumulators; the previous item's finished values: `prev` (a copy), or -- FLUSH -- the second-level set `acc2` itself
  f32x16 acc[2][2], prev[2][2], acc2[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[t][u][r] = 0.f; prev[t][u][r] = 0.f; acc2[t][u][r] = 0.f; }
#define P256_PREV(t, u, r) (FLUSH ? acc2[t][u][r] : prev[t][u][r])
#define P256_FLUSH_STEP()                                                                                 \
  if (FLUSH) {                                                                                            \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                         \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                     \
        acc2[t][u] += acc[t][u];                                                                          \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;                                \
      }                                                                                                   \
  }

  // ---- epilogue state of the PREVIOUS item.  Chunk c (0..7) = accumulator block (t, u) = (c >> 2, (c >> 1) & 1), registers
  // r = (c & 1) * 8 .. + 7: rows t*32 + (r & 3) + 8 * (r >> 2) (+ wm*64 + 4*hi), column u*32 + li (+ wn*64).  Bases are
  // wave-uniform, the lane contributes a constant byte offset; whole tiles only (the gate), so no bounds are checked.
  const int cout = p.cout;
  bool prev_valid = false, prev_raw = false;
  const float* prev_res = p.zeros;        // uniform: residual of the previous item's tile (or the zero page, stride 0)
  float* prev_out = p.y;                  // uniform: its output tile (or its raw partial tile)
  const float* prev_ss = p.scale;         // uniform: scale of its first column
  unsigned prev_rs = 0;                   // bytes between rows of the residual (0: zero page)
  unsigned prev_os = (unsigned)cout * 4;  // bytes between rows of the output
  int prev_rows = BM;                     // RAGGED: rows of the previous item's tile that exist (raw partial tiles: all)
  const unsigned lane_row = (unsigned)(wm * 64 + 4 * hi), lane_col = (unsigned)(wn * 64 + li);
  const unsigned ss_lane = lane_col * 4;
  const unsigned shift_delta = (unsigned)((p.shift - p.scale) * 4);
  float sc0 = 1.f, sc1 = 1.f, sh0 = 0.f, sh1 = 0.f;
  const bool has_res = p.res != nullptr;
  const float alpha = p.alpha;
  const bool relu = p.relu != 0;

#define P256_EPI_SS_LOADS()                                                                                       \
  {                                                                                                               \
    sc0 = p_load_uncounted(prev_ss, ss_lane);                                                                     \
    sc1 = p_load_uncounted(prev_ss, ss_lane + 128);                                                               \
    sh0 = p_load_uncounted(prev_ss, ss_lane + shift_delta);                                                       \
    sh1 = p_load_uncounted(prev_ss, ss_lane + shift_delta + 128);                                                 \
  }
#define P256_EPI_LOADS(c)                                                                                         \
  float resv[8];                                                                                                  \
  {                                                                                                               \
    if ((c) == 0) {                                                                                               \
      sc0 = p_load_uncounted(prev_ss, ss_lane);                                                                   \
      sc1 = p_load_uncounted(prev_ss, ss_lane + 128);                                                             \
      sh0 = p_load_uncounted(prev_ss, ss_lane + shift_delta);                                                     \
      sh1 = p_load_uncounted(prev_ss, ss_lane + shift_delta + 128);                                               \
    }                                                                                                             \
    const unsigned lane_off = (lane_row * prev_rs) + (prev_rs ? lane_col * 4 : 0);                                \
    /* RAGGED: the clamped row offsets are worked out afresh in every group (the limit goes through an opaque copy): groups c  \
       and c + 2 read the same rows, and hipcc otherwise keeps the eight products alive between them -- five of them in      \
       scratch (the <RAGGED, !FLUSH> variant sits at 256 VGPRs) */                                                  \
    int row_limit = prev_rows - 1;                                                                                \
    if (RAGGED) asm volatile("" : "+s"(row_limit));                                                               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                               \
      const int r = ((c) & 1) * 8 + i, row = ((c) >> 2) * 32 + (r & 3) + 8 * (r >> 2);                            \
      unsigned roff = lane_off + (unsigned)row * prev_rs;                                                         \
      if (RAGGED) {                                                                                               \
        const int rr = (int)lane_row + row;                                                                       \
        /* (24-bit multiply: rows and the row stride in bytes are far below 2^24; the 32-bit product + add otherwise becomes a     \
           v_mad_u64_u32 whose unused upper addend half lands on a register with a load in flight -- harmless, but the audit     \
           of uncounted loads, tests/test_abi.py, rightly has no notion of "unused half") */                                   \
        roff = __umul24((unsigned)(rr < row_limit ? rr : row_limit), prev_rs) + (prev_rs ? lane_col * 4 : 0);     \
      }                                                                                                           \
      resv[i] = p_load_uncounted(prev_res, roff + (prev_rs ? (((c) >> 1) & 1) * 128u : 0u));                      \
    }                                                                                                             \
  }
#define P256_EPI_FINISH(c) P256_EPI_FINISH_RV(c, resv)
#define P256_EPI_FINISH_RV(c, RV)                                                                                 \
  if (prev_valid) {                                                                                               \
    const float scv = (((c) >> 1) & 1) ? sc1 : sc0, shv = (((c) >> 1) & 1) ? sh1 : sh0;                           \
    char* const obase = reinterpret_cast<char*>(prev_out) + (lane_row * prev_os + lane_col * 4 + (((c) >> 1) & 1) * 128u); \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                               \
      const int r = ((c) & 1) * 8 + i, row = ((c) >> 2) * 32 + (r & 3) + 8 * (r >> 2);                            \
      float v = P256_PREV((c) >> 2, ((c) >> 1) & 1, r);                                                           \
      if (!prev_raw) {                                                                                            \
        v = v * (scv * alpha) + shv;                                                                              \
        v += RV[i];                                                                                               \
        if (relu) v = relu_keep_nan(v);                                                                           \
      }                                                                                                           \
      if (!RAGGED || (int)lane_row + row < prev_rows) *reinterpret_cast<float*>(obase + (unsigned)row * prev_os) = v; \
    }                                                                                                             \
  }

  // ---- prologue: two k-tiles in flight ----
  cursor_open(0);
  request(smem);
  cursor_advance();
  request(smem + STAGE);
  cursor_advance();
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  int o_cur = 0, o_mid = STAGE, o_fill = 2 * STAGE;

#define P256_MFMA_PAIR()                                                                                  \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                      \
      _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                       \
        _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                     \
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t][kk], bf[j][u][kk], acc[t][u], 0, 0, 0);
#define P256_READ_PAIR(g)                                                                                            \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                    \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) af[j][t] = *reinterpret_cast<const f32x4*>(cur + a_row + t * 32 * BK + sw[(g) + j]); \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) bf[j][u] = *reinterpret_cast<const f32x4*>(cur + b_row + u * 32 * BK + sw[(g) + j]); \
  }
  // one k-tile; the caller defines P256_EPI_LOADS_IF / P256_EPI_FINISH_IF / P256_RESV for the epilogue chunk it carries
#define P256_ITERATION()                                                                                  \
  {                                                                                                       \
    const float* const cur = smem + o_cur;                                                                \
    float* const fill = smem + o_fill;                                                                    \
    float resv_none[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                                        \
    (void)resv_none;                                                                                      \
    P256_EPI_LOADS_IF                                                                                     \
    if (!late) request(fill);                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    f32x4 af[2][2], bf[2][2];                                                                             \
    P256_READ_PAIR(0);                                                                                    \
    P256_MFMA_PAIR();                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    if (late) request(fill);                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    P256_READ_PAIR(2);                                                                                    \
    P256_MFMA_PAIR();                                                                                     \
    p_wait<6>(P256_RESV, sc0, sc1, sh0, sh1);                                                             \
    P256_EPI_FINISH_IF                                                                                    \
    P256_FLUSH_IF                                                                                         \
    cursor_advance();                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
    __builtin_amdgcn_s_barrier();                                                                         \
    asm volatile("" ::: "memory");                                                                        \
    { const int tmp = o_cur; o_cur = o_mid; o_mid = o_fill; o_fill = tmp; }                               \
  }

  for (int ci = 0; ci < n_items; ++ci) {
    const PItem it = item_at(ci);
    const int nk = it.kt1 - it.kt0;
    // the first eight iterations carry the previous item's epilogue (chunk = iteration)
    static_for<8>([&](auto kc) {
      constexpr int c = decltype(kc)::value;
      if (c < nk) {
        if constexpr (!FLUSH) {
#define P256_EPI_LOADS_IF P256_EPI_LOADS(c)
#define P256_EPI_FINISH_IF P256_EPI_FINISH(c)
#define P256_FLUSH_IF
#define P256_RESV resv
          P256_ITERATION()
#undef P256_EPI_LOADS_IF
#undef P256_EPI_FINISH_IF
#undef P256_FLUSH_IF
#undef P256_RESV
        } else {
          // iteration 0 stores chunks 0-3 of the previous totals, iteration 1 chunks 4-7 and clears the set; odd iterations flush
#define P256_EPI_LOADS_IF if (c == 0) P256_EPI_SS_LOADS()
#define P256_EPI_FINISH_IF                                                                   \
  if (c == 0) { P256_EPI_FINISH_RV(0, resv_none) P256_EPI_FINISH_RV(1, resv_none) P256_EPI_FINISH_RV(2, resv_none) P256_EPI_FINISH_RV(3, resv_none) } \
  if (c == 1) {                                                                              \
    P256_EPI_FINISH_RV(4, resv_none) P256_EPI_FINISH_RV(5, resv_none) P256_EPI_FINISH_RV(6, resv_none) P256_EPI_FINISH_RV(7, resv_none) \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                            \
      _Pragma("unroll") for (int u = 0; u < 2; ++u)                                          \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc2[t][u][r] = 0.f;                  \
  }
#define P256_FLUSH_IF if (c & 1) { P256_FLUSH_STEP() }
#define P256_RESV resv_none
          P256_ITERATION()
#undef P256_EPI_LOADS_IF
#undef P256_EPI_FINISH_IF
#undef P256_FLUSH_IF
#undef P256_RESV
        }
      }
    });
    for (int kt = 8; kt < nk; ++kt) {
#define P256_EPI_LOADS_IF
#define P256_EPI_FINISH_IF
#define P256_FLUSH_IF if (kt & 1) { P256_FLUSH_STEP() }
#define P256_RESV resv_none
      P256_ITERATION()
#undef P256_EPI_LOADS_IF
#undef P256_EPI_FINISH_IF
#undef P256_FLUSH_IF
#undef P256_RESV
    }
    if (FLUSH && (nk & 1)) { P256_FLUSH_STEP() }       // a split part with an odd number of k-tiles: its last one
    // an item shorter than eight k-tiles (a split part): the rest of the previous epilogue, with nothing under it
    if (!FLUSH && nk < 8 && prev_valid) {
      static_for<8>([&](auto kc) {
        constexpr int c = decltype(kc)::value;
        if (c >= nk) {
          P256_EPI_LOADS(c)
          p_wait<0>(resv, sc0, sc1, sh0, sh1);
          P256_EPI_FINISH(c)
        }
      });
    }
    // the item's accumulators become the "previous" set (FLUSH: they already sit in acc2, and acc is zero)
    if (!FLUSH) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          prev[t][u] = acc[t][u];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
        }
    }
    {
      const size_t tile_el = (size_t)it.mt * BM * cout + (size_t)it.nt * BN;
      prev_valid = true;
      prev_raw = it.part >= 0;
      prev_rows = BM;
      if (prev_raw) {
        prev_out = p.partial + (size_t)it.part * (BM * BN);
        prev_os = BN * 4;
        prev_res = p.zeros;
        prev_rs = 0;
      } else {
        if (RAGGED && p.M - it.mt * BM < BM) prev_rows = p.M - it.mt * BM;
        prev_out = p.y + tile_el;
        prev_os = (unsigned)cout * 4;
        prev_res = has_res ? p.res + tile_el : p.zeros;
        prev_rs = has_res ? (unsigned)cout * 4 : 0;
      }
      prev_ss = p.scale + ((p.mt_per_group && p.ss_group_stride) ? (it.mt / p.mt_per_group) * p.ss_group_stride : 0) + it.nt * BN;
    }
  }
  // ---- drain: the last item's epilogue ----
  static_for<8>([&](auto kc) {
    constexpr int c = decltype(kc)::value;
    P256_EPI_LOADS(c)
    p_wait<0>(resv, sc0, sc1, sh0, sh1);
    P256_EPI_FINISH(c)
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the repeated requests of the stream's end
#undef P256_PREV
#undef P256_FLUSH_STEP
#undef P256_EPI_SS_LOADS
#undef P256_EPI_FINISH_RV
#undef P256_ITERATION
#undef P256_READ_PAIR
#undef P256_MFMA_PAIR
#undef P256_EPI_LOADS
#undef P256_EPI_FINISH
}

}  // namespace

// Which pointwise layers take the persistent 256 x 128 kernel: one running sum (no two-level accumulation), no weight
// groups, whole 128-wide packed n-tiles, at least pw256p_mink input channels (8 k-tiles: the previous item's epilogue is spread
// over the first eight iterations of the next) and at least pw256p_mintiles tiles.
bool conv_pw_uses_256p(int cout, long long M, int mt_per_group, int bn_tile, int cin, int flush_ktiles, long long in_pixels) {
  const int min_k = (int)opt(OPT_PW256P_MINK);
  if (min_k <= 0 || bn_tile != 128 || cin < min_k || cin < 256) return false;
  // Winograd position GEMMs (partial sums of 64 channels): measured per layer (profiles/r5o) K = 512 +2.5 %, K = 2048 -4 % (the
  // totals leave in two bursts of 32 stores) -> up to pw256p_flush input channels only
  if (flush_ktiles != 0 && (flush_ktiles != 2 || cin > opt(OPT_PW256P_FLUSH))) return false;
  if (mt_per_group % 2 != 0) return false;                                 // grouped GEMM: whole 256-row tiles per weight group
  if (cout % 128 != 0) return false;                                       // whole 128-wide n-tiles: the register epilogue checks no column bounds
  // 32-bit byte offsets: into the output (M rows) and into the INPUT, whose pixel indices b*H*W + oy*stride*W + ox*stride run over
  // in_pixels = B*H*W rows -- four times M for a stride-2 layer (0: the caller has no geometry; a stride-1 layer has in_pixels = M)
  const long long in_rows = in_pixels > M ? in_pixels : M;
  if (in_rows * (long long)cin * 4 >= (1LL << 32) || M * (long long)cout * 4 >= (1LL << 32)) return false;
  return ((M + 255) / 256) * (cout / 128) >= opt(OPT_PW256P_MINTILES);
}

int launch_conv_pw256p(const ConvKParams& p0, float* ws, size_t ws_floats, hipStream_t stream) {
  ConvKParams p = p0;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
    return fail(-3, "conv_pw256p: no current device");
  const int mtiles = (p.M + 255) / 256;
  const int T = mtiles * p.ntiles;
  const int G = cus - cus % 8;          // one workgroup per CU whatever the tile count: a launch with fewer tiles than CUs is all split parts
  // Every workgroup gets the same number of whole tiles (n_full / G); the T mod G tiles left over are cut into split_p
  // k-ranges each (raw partial tiles + the ordered reduce, as in launch_with_tail_split) and dealt out evenly.  split_p
  // minimises the busiest workgroup's extra work: ceil(t * p / G) parts of nkt / p k-tiles plus ~1.5 k-tile times per part (raw
  // store, cursor switch); parts keep at least four k-tiles, preferably eight (the previous epilogue rides on eight iterations).
  const int t = T % G;
  int sp = 1;
  if (t > 0) {
    double best = 1e30;
    for (int cand = 1; cand <= 16 && p.nkt / cand >= 4; ++cand) {
      if (ws == nullptr || (size_t)t * cand * 256 * 128 > ws_floats) break;
      const double parts = (double)(((long long)t * cand + G - 1) / G);
      const double cost = parts * ((double)p.nkt / cand + 1.5 + (p.nkt / cand < 8 ? 2.0 : 0.0));
      if (cost < best - 1e-9) { best = cost; sp = cand; }
    }
  }
  p.split_p = sp;
  p.n_sp = t * sp;
  p.n_full = T - t;
  // Stream-K for the tail (round 4): its t * nkt k-tiles as one stream, cut into equal runs for the first sk_g workgroups (runs of
  // at least four k-tiles).  Against the uniform split above -- whose parts come in whole multiples per workgroup (132 tail tiles
  // cut three ways = 396 parts over 256 workgroups: two parts for most, 21 k-tiles where 16.5 would do) -- every workgroup ends
  // within one k-tile of the others.  A tile then has up to sk_maxp fragments; the reduce sums them in workgroup order.
  p.sk_units = 0; p.sk_maxp = 0; p.sk_g = 0; p.sk_q = 1;
  // (only for tails of at least a quarter of a round: measured per layer, profiles/r6b -- 132 tail tiles of 32 k-tiles 0.502 -> 0.489 ms,
  // 8 or 32 tail tiles level or 1 % slower: their runs are a few k-tiles long and all fragments)
  // The two-level accumulation variant streams in units of TWO k-tiles (one partial sum of 64 channels): its register epilogue
  // hands the previous item's totals out during the next item's first two iterations, so no fragment may be shorter, and partial
  // sums then cover the same channel groups as in an uncut tile.
  const int q = p.flush ? 2 : 1;
  if (t * 4 >= G && opt(OPT_PW256P_STREAMK) != 0 && ws != nullptr && p.nkt % q == 0) {
    const int upt = p.nkt / q;                                      // stream units per tile
    const long long U = (long long)t * upt;
    const int Gs = (int)std::min<long long>(G, std::max<long long>(1, U * q / 4));
    const int run = (int)(U / Gs);                                  // shortest run, in units
    const int maxp = run > 0 ? (upt + run - 1) / run + 1 : 0;
    const double parts_now = (double)(((long long)t * sp + G - 1) / G);
    const double cost_now = parts_now * ((double)p.nkt / sp + 1.5 + (p.nkt / sp < 8 ? 2.0 : 0.0));
    const double cost_stream = (double)((U + Gs - 1) / Gs) * q + 3.0;  // a run is two fragments on average: two raw stores / cursor switches
    if (run * q >= 4 && (size_t)t * maxp * 256 * 128 <= ws_floats && cost_stream < cost_now - 0.5) {
      p.sk_units = (int)U; p.sk_maxp = maxp; p.sk_g = Gs; p.sk_q = q;
      p.n_sp = 0; p.split_p = 1;
    }
  }
  if (p.n_sp > 0 && (!ws || (size_t)p.n_sp * 256 * 128 > ws_floats)) return fail(-2, "conv_pw256p: split-K scratch too small");
  // more items per workgroup than the LDS plan table holds (a layer near the 4 GiB output bound, or a part with few CUs): nothing is
  // launched and the caller takes the tile-per-workgroup kernel, as launch_conv_pw256wp's callers do
  if (p.n_full / G + (p.n_sp + G - 1) / G + 4 > 120) return 1;
  p.partial = ws;
  p.mtiles = mtiles;
  p.nchunk = (int)opt(OPT_NCHUNK);
  p.phase_shift = opt(OPT_PW256_PHASE) != 0;
  p.p_order = opt(OPT_PW256P_ORDER) != 0;
  note_kernel("conv_pw_glds_256x128p");
  if (p.mt_per_group) p.mt_per_group /= 2;       // 256-row tiles per weight group
  const bool ragged = p.M % 256 != 0;
  if (p.flush) {
    if (p.res) return fail(-2, "conv_pw256p: the two-level accumulation variant takes no residual");
    if (ragged) hipLaunchKernelGGL((conv_pw_glds256p_kernel<true, true>), dim3((unsigned)G), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((conv_pw_glds256p_kernel<false, true>), dim3((unsigned)G), dim3(512), 0, stream, p);
  } else {
    if (ragged) hipLaunchKernelGGL((conv_pw_glds256p_kernel<true, false>), dim3((unsigned)G), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((conv_pw_glds256p_kernel<false, false>), dim3((unsigned)G), dim3(512), 0, stream, p);
  }
  if (p.n_sp > 0 || p.sk_units > 0)
    hipLaunchKernelGGL((conv_splitk_reduce_kernel<256, 128>), dim3((unsigned)t, 256 / 16), dim3(256), 0, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-3, std::string("conv_pw256p launch: ") + hipGetErrorString(e));
  return 0;
}

}  // namespace peanut
