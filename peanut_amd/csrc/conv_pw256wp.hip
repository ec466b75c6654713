// conv_pw_glds256wp_kernel: the 256 x 256 two-stage LDS-DMA GEMM of conv_pw.hip (8 waves as 4 x 2, wave tile 64 x 128) as a
// PERSISTENT kernel (round 5).
//
// conv_pw_glds256w_kernel has the lowest operand traffic per FLOP of the fp32 kernels (two thirds of the 256 x 128 tile's) and the
// busiest matrix pipes (0.90 against 0.83 for the persistent 256 x 128 kernel, profiles/r6f), but as one tile per workgroup and one
// workgroup per CU it pays a pipeline fill and an exposed 512 KiB epilogue per tile (13 % of a K = 512 tile) and quantises badly at
// N = 512 (900 tiles over 256 CUs) -- it was gated to K >= 768 and >= 1 536 tiles.  Here one workgroup per CU walks a list of
// items (whole tiles in the XCD-interleaved order of conv_pw256p.hip, then its run of the stream-K tail):
//   * the two-stage ring runs ACROSS item boundaries: the first k-tile of the next item is requested during the last iteration of
//     the current one;
//   * there is no room for a second accumulator set (128 of the wave's 256 registers are accumulators), so the epilogue of item i
//     runs IN PLACE inside the first iteration of item i + 1: that iteration walks the accumulator blocks in 8 / NPRE groups of
//     NPRE blocks; a group's blocks are finished (scale / shift / residual / ReLU) and stored, then restarted by an MFMA whose C
//     operand is the constant 0, then run through the whole k-tile while the residual of the NEXT group travels
//     (NPRE x 16 registers; the first group's residual and the scale / shift values are requested during the item's own last
//     iteration).  Every block still sees its k-steps in the same order: results are bit-identical to the other fp32 kernels.
//   * waits are `__builtin_amdgcn_s_waitcnt` (not inline asm): hipcc's own wait insertion then knows what has landed and adds
//     no `vmcnt` of its own in front of the residual's first use (which would also wait for the LDS-DMA requests in flight).
// One running sum only (the Winograd position GEMMs with their two-level accumulation keep the 256 x 128 kernels), no weight
// groups, whole tiles (M % 256 == 0, cout % 256 == 0).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "conv_common.h"

namespace peanut {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// s_waitcnt immediates (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8])
constexpr int kWaitVm0 = 0x0F70;                      // vmcnt(0)
constexpr int wait_vm(int n) {                        // vmcnt(n), n <= 63 (a larger count saturates: the hardware counter has six bits)
  return 0x0F70 | ((n > 63 ? 63 : n) & 15) | ((((n > 63 ? 63 : n) >> 4) & 3) << 14);
}
constexpr int kWaitLgkm0 = 0xC07F;                    // lgkmcnt(0)

constexpr int kMaxItems = 120;                       // items (whole tiles + tail fragments) of one workgroup: the plan table in LDS

// lane value (< 2^24) x uniform (< 2^24): one v_mul_u32_u24
__device__ __forceinline__ unsigned lane_mul24(unsigned a, unsigned b) { return __umul24(a, b); }

struct WItem { int m0, nt, kt0, kt1, part; };   // first row, 256-wide n-tile, k-tile range; part >= 0: raw partial tile #part

// RES: the layer has a residual.  NPRE: accumulator blocks per epilogue group (4: two groups = the two 32-row halves of the wave
// tile; 2: four groups).  Stride 1 only (pixel index = output row: no divisions anywhere).
template <bool RES, int NPRE>
__global__ __launch_bounds__(512) void conv_pw_glds256wp_kernel(const ConvKParams p) {
  constexpr int BM = 256, BN = 256, BK = 32, WN = 2;
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;   // 64 KiB
  constexpr int NG = 8 / NPRE, GPT = 4 / NPRE;          // epilogue groups per tile / per 32-row half
  static_assert(NPRE == 4 || NPRE == 2, "blocks per epilogue group");
  // ONE LDS object (the plan table behind the two stages): with a second __shared__ variable hipcc attaches alias scopes to the
  // LDS accesses and then puts an `s_waitcnt vmcnt(0)` between every LDS-DMA request and the fragment reads that follow it
  __shared__ __attribute__((aligned(1024))) float smem[2 * STAGE + kMaxItems * 8];
  int* const plan = reinterpret_cast<int*>(smem + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 3, lp = lane & 7;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const bool late = p.phase_shift && wave >= 4;

  // ---- this workgroup's items (the plan of conv_pw256p.hip): nf whole tiles, then its share of the tail.  Worked out once, one
  // item per thread, into LDS: the 64-bit divisions of the stream-K bookkeeping stay out of the loop. ----
  int n_items;
  {
    const int G = gridDim.x, bid = blockIdx.x;
    const int lw = (bid & 7) * (G >> 3) + (bid >> 3);                    // neighbouring runs on one XCD (G is a multiple of 8)
    const int nf = p.n_full / G;
    const int sp0 = (int)((long long)p.n_sp * lw / G), sp1 = (int)((long long)p.n_sp * (lw + 1) / G);
    const long long skU = p.sk_units;
    const int u0 = (skU > 0 && lw < p.sk_g) ? (int)(skU * lw / p.sk_g) : 0, u1 = (skU > 0 && lw < p.sk_g) ? (int)(skU * (lw + 1) / p.sk_g) : 0;
    const int upt = p.nkt / p.sk_q;                         // stream units (sk_q = 2 k-tiles: no fragment is shorter than two iterations) per tile
    const int n_frag = u1 > u0 ? (u1 - 1) / upt - u0 / upt + 1 : 0;
    n_items = nf + (skU > 0 ? n_frag : sp1 - sp0);
    if (n_items == 0) return;
    if (n_items > kMaxItems) __builtin_trap();      // the launcher bounds the plan; never write past the LDS table
    if (tid < n_items) {
      const int i = tid;
      int tile, kt0, kt1, part;
      if (i >= nf && skU > 0) {
        const int j = u0 / upt + (i - nf);                    // tail tile of this fragment
        const int t0 = j * upt;
        kt0 = ((u0 > t0 ? u0 : t0) - t0) * p.sk_q;
        kt1 = ((u1 < t0 + upt ? u1 : t0 + upt) - t0) * p.sk_q;
        part = j * p.sk_maxp + (lw - sk_owner(skU, p.sk_g, (long long)t0));   // fragments of a tile in workgroup order
        tile = p.n_full + j;
      } else if (i < nf) {
        tile = p.p_order ? (lw / (G >> 3)) * (nf * (G >> 3)) + i * (G >> 3) + lw % (G >> 3) : lw * nf + i;
        kt0 = 0; kt1 = p.nkt; part = -1;
      } else {
        const int s = sp0 + (i - nf);
        const int j = s / p.split_p, pt = s - j * p.split_p;
        tile = p.n_full + j;
        part = s;
        kt0 = (int)((long long)pt * p.nkt / p.split_p);
        kt1 = (int)((long long)(pt + 1) * p.nkt / p.split_p);
      }
      int mt, nt;
      tile_to_mn(p, tile, &mt, &nt);
      int* e = plan + i * 8;
      e[0] = mt * BM; e[1] = nt; e[2] = kt0; e[3] = kt1; e[4] = part;
    }
    __syncthreads();
  }
  auto item_at = [&](int i) {
    const int* e = plan + i * 8;
    WItem it;
    it.m0 = __builtin_amdgcn_readfirstlane(e[0]);
    it.nt = __builtin_amdgcn_readfirstlane(e[1]);
    it.kt0 = __builtin_amdgcn_readfirstlane(e[2]);
    it.kt1 = __builtin_amdgcn_readfirstlane(e[3]);
    it.part = __builtin_amdgcn_readfirstlane(e[4]);
    return it;
  };

  // ---- the request cursor: one k-tile ahead of the compute cursor, across item boundaries.  All of its state is wave-uniform;
  // a lane contributes three constants: its row inside the wave's 32-row share (8 rows per LDS-DMA piece, four pieces per operand)
  // and the two chunk positions of the LDS image's swizzle (even / odd piece).  A piece's address is
  //   source base + first row of the piece * row bytes + k-tile offset  (uniform)  +  lane row * row bytes + chunk  (32-bit lane part)
  const unsigned row0 = (unsigned)(wave * 32 + lr);
  const unsigned c_e = (unsigned)((lp ^ ((row0 >> 1) & 7)) * 16), c_o = (unsigned)((lp ^ (((row0 + 8) >> 1) & 7)) * 16);
  const unsigned b_lane_e = (unsigned)lr * (BK * 4) + c_e, b_lane_o = (unsigned)lr * (BK * 4) + c_o;
  const size_t b_wave = ((size_t)(wave >> 2) * p.nkt * 128 + (size_t)(wave & 3) * 32) * BK;   // floats: packed tile 2 nt + (wave >> 2), its rows (wave & 3) * 32 ...
  int d_item = 0, d_left = 0, d_kt = 0, d_m0 = 0;
  const float* d_wtile = p.w;
  auto cursor_open = [&](int i) {
    const WItem it = item_at(i);
    d_m0 = it.m0;
    d_wtile = p.w + (size_t)(2 * it.nt) * p.nkt * (128 * BK) + b_wave;
    d_kt = it.kt0;
    d_left = it.kt1 - it.kt0;
  };
  const int k1 = p.c1 / BK;
  auto request_now = [&](float* stage) {
    const bool second = d_kt >= k1;                               // two sources: k-tiles [0, k1) from x, the rest from x2
    const unsigned cbytes = (unsigned)(second ? p.c2 : p.c1) * 4u;
    const char* abase = reinterpret_cast<const char*>(second ? p.x2 : p.x) + (size_t)(second ? d_kt - k1 : d_kt) * (BK * 4) +
                        (size_t)d_m0 * cbytes;
    const unsigned a_lane = lane_mul24(lr, cbytes);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(abase + (size_t)(wave * 32 + 8 * j) * cbytes + (a_lane + ((j & 1) ? c_o : c_e))),
                                       (lptr_t)(stage + (wave * 4 + j) * 256), 16, 0, 0);
    const char* bbase = reinterpret_cast<const char*>(d_wtile + (size_t)d_kt * (128 * BK));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(bbase + (size_t)(8 * j) * (BK * 4) + ((j & 1) ? b_lane_o : b_lane_e)),
                                       (lptr_t)(stage + A_FLOATS + (wave * 4 + j) * 256), 16, 0, 0);
    ++d_kt;
    --d_left;
  };
  auto request = [&](float* stage) {
    if (d_left > 0) request_now(stage);
  };
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
  const int b_row = A_FLOATS + (wn * 128 + li) * BK;

  f32x16 acc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // ---- epilogue state.  Block (t, u) of a lane: rows t*32 + (i & 3) + 8 * (i >> 2) (+ wm*64 + 4*hi), column u*32 + li (+ wn*128); a
  // wave instruction covers two rows x 32 consecutive channels = two whole 128-byte lines.  Bases are wave-uniform.  Branch-free:
  // a raw partial tile is "scale 1, shift 0, residual from the zero page, no ReLU".
  const unsigned lane_row = (unsigned)(wm * 64 + 4 * hi), lane_colb = (unsigned)(wn * 128 + li) * 4u;
  struct Epi {
    const char* res;     // residual of the tile (or the zero page, stride 0)
    char* out;           // the output tile, or the raw partial tile
    const char* ss;      // scale of the tile's first column
    unsigned rs, os;     // bytes between rows of the residual / the output
    bool raw;
  };
  // before the first item: a "previous item" whose tile is the dump tile in the scratch
  Epi prev{reinterpret_cast<const char*>(p.zeros), reinterpret_cast<char*>(p.dump), reinterpret_cast<const char*>(p.scale), 0u,
           (unsigned)BN * 4u, true};
  Epi cur = prev;
  float prev_lo = 0.f;                // lower clamp of the previous item's values: 0 (ReLU) or -inf (none / raw)
  const unsigned shift_delta = (unsigned)((p.shift - p.scale) * 4);
  float R[NPRE][16];                 // residual of one epilogue group
  float sc[4], sh[4];                // scale * alpha and shift of the lane's four columns
#pragma unroll
  for (int u = 0; u < NPRE; ++u)
#pragma unroll
    for (int i = 0; i < 16; ++i) R[u][i] = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) { sc[u] = 1.f; sh[u] = 0.f; }

  // residual of block (t, u) of tile `e` into rv
  auto load_res = [&](const Epi& e, auto tc, auto uc, float (&rv)[16]) {
    constexpr int t = decltype(tc)::value, u = decltype(uc)::value;
    if constexpr (RES) {
      const unsigned lane_off = lane_mul24(lane_row, e.rs) + (e.rs ? lane_colb : 0u);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = t * 32 + (i & 3) + 8 * (i >> 2);
        rv[i] = *reinterpret_cast<const float*>(e.res + ((size_t)row * e.rs + (e.rs ? u * 128u : 0u)) + lane_off);
      }
    }
  };
  // scale / shift of the tile's columns: requested in the item's last iteration (no arithmetic on them there: a use would make
  // hipcc wait for the loads, and with them for the LDS-DMA requests just issued) ...
  auto load_ss = [&](const Epi& e) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      sc[u] = *reinterpret_cast<const float*>(e.ss + (size_t)(u * 128) + lane_colb);
      sh[u] = *reinterpret_cast<const float*>(e.ss + (size_t)(shift_delta + u * 128) + lane_colb);
    }
  };
  // ... and brought into their final form where the epilogue starts (a raw partial tile: scale 1, shift 0)
  auto finish_ss = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      sc[u] = prev.raw ? 1.f : sc[u] * p.alpha;
      sh[u] = prev.raw ? 0.f : sh[u];
    }
  };
  // The epilogue of block (t, u) of the PREVIOUS item, in two steps: the values are finished IN PLACE in the accumulator registers
  // (which consumes the block's residual registers: the next group's residual can then be requested BEFORE this group's stores are
  // issued -- vmcnt retires in order, so a wait for those loads then leaves the stores in flight) ...
  auto finish_block = [&](auto tc, auto uc, const float (&rv)[16]) {
    constexpr int t = decltype(tc)::value, u = decltype(uc)::value;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = acc[t][u][i] * sc[u] + sh[u];
      if constexpr (RES) v += rv[i];
      acc[t][u][i] = v < prev_lo ? prev_lo : v;   // NaN stays NaN (relu_keep_nan)
    }
  };
  // ... and stored
  auto store_block = [&](auto tc, auto uc) {
    constexpr int t = decltype(tc)::value, u = decltype(uc)::value;
    const unsigned lane_off = lane_mul24(lane_row, prev.os) + lane_colb;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = t * 32 + (i & 3) + 8 * (i >> 2);
      *reinterpret_cast<float*>(prev.out + ((size_t)row * prev.os + u * 128u) + lane_off) = acc[t][u][i];
    }
  };

#ifdef PEANUT_WP_TRACE      // tools/micro/wp_probe.hip: s_memtime stamp of every iteration's end, by kind, for the first 16 workgroups
  long long* const trace = reinterpret_cast<long long*>(p.partial + ((size_t)48 << 20)) + blockIdx.x * 1024;
  int trace_n = 0;
#define WP_TRACE(kind)                                                                                        \
  if (blockIdx.x < 16 && tid == 0) {                                                                          \
    ++trace_n;                                                                                                \
    if (trace_n < 1024) trace[trace_n] = (long long)(__builtin_amdgcn_s_memtime() << 2) | (kind);             \
    trace[0] = trace_n;                                                                                       \
  }
#else
#define WP_TRACE(kind)
#endif
#define WP_SCHED() __builtin_amdgcn_sched_barrier(0)
#define WP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
  int ph = 0;                                          // stage that holds the current k-tile
  auto end_iteration = [&](int kind) {
    WP_SCHED();
    __builtin_amdgcn_s_waitcnt(kWaitVm0);              // the next k-tile has landed (and this iteration's stores / loads are done)
    cursor_advance();
    __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WP_TRACE(kind)
    ph ^= 1;
  };
  // k-groups [g0, g0 + ng) of the current k-tile on all eight blocks
  auto mfma_groups = [&](const float* cur_stage, auto g0c, auto ngc) {
    constexpr int g0 = decltype(g0c)::value, ng = decltype(ngc)::value;
    f32x4 af[ng][2], bf[ng][4];
#pragma unroll
    for (int j = 0; j < ng; ++j) {
#pragma unroll
      for (int t = 0; t < 2; ++t) af[j][t] = *reinterpret_cast<const f32x4*>(cur_stage + a_row + t * 32 * BK + sw[g0 + j]);
#pragma unroll
      for (int u = 0; u < 4; ++u) bf[j][u] = *reinterpret_cast<const f32x4*>(cur_stage + b_row + u * 32 * BK + sw[g0 + j]);
    }
#pragma unroll
    for (int j = 0; j < ng; ++j)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[t][u] = WP_MFMA(af[j][t][kk], bf[j][u][kk], acc[t][u]);
  };

  // ---- the iteration kinds ----
  // steady state (the 256 x 256 loop of conv_pw.hip)
  auto iter_steady = [&]() {
    const float* const cs = smem + ph * STAGE;
    float* const fill = smem + (ph ^ 1) * STAGE;
    if (!late) request(fill);
    WP_SCHED();
    mfma_groups(cs, IC<0>{}, IC<2>{});
    WP_SCHED();
    if (late) request(fill);
    WP_SCHED();
    mfma_groups(cs, IC<2>{}, IC<2>{});
    end_iteration(1);
  };
  // the last iteration of an item: also requests what the item's epilogue will need first (scale / shift, the first group's
  // residual); its second half reads its fragments one k-group at a time (the registers of the other group hold the residual)
  auto iter_last = [&]() {
    const float* const cs = smem + ph * STAGE;
    float* const fill = smem + (ph ^ 1) * STAGE;
    if (!late) request(fill);
    WP_SCHED();
    mfma_groups(cs, IC<0>{}, IC<2>{});
    WP_SCHED();
    if (late) request(fill);
    load_ss(cur);
    static_for<NPRE>([&](auto uc) { load_res(cur, IC<0>{}, uc, R[decltype(uc)::value]); });
    WP_SCHED();
    mfma_groups(cs, IC<2>{}, IC<1>{});
    WP_SCHED();
    mfma_groups(cs, IC<3>{}, IC<1>{});
    end_iteration(2);
  };
  // the first iteration of an item: group after group, the previous item's blocks are finished, stored, restarted from zero and
  // taken through the whole k-tile, while the next group's residual travels.  Order inside a group: finish all its blocks, request
  // the next group's residual, (in the first group: the LDS-DMA request -- all eight waves at the same point here, so that the
  // number of operations in flight is the same in every wave,) store the blocks -- so the wait at the group's end covers the residual
  // and leaves the request and the stores in flight (vmcnt retires in order); the stores get two groups to complete, the last
  // group's even the following iteration.  ONE straight path: with branches around the stores hipcc's own wait insertion no longer
  // knows what is in flight and falls back to `vmcnt(0)` in front of every residual use.  The workgroup's very first iteration
  // has no previous item: its "epilogue" goes to a dump tile in the scratch (p.dump).
  auto iter_first = [&]() {
    const float* const cs = smem + ph * STAGE;
    float* const fill = smem + (ph ^ 1) * STAGE;
    finish_ss();
    WP_SCHED();
    static_for<NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value, t = g / GPT, ub = (g % GPT) * NPRE;
      constexpr int tn = (g + 1) / GPT, ubn = ((g + 1) % GPT) * NPRE;     // the next group
      static_for<NPRE>([&](auto uc) { finish_block(IC<t>{}, IC<ub + decltype(uc)::value>{}, R[decltype(uc)::value]); });
      WP_SCHED();
      if constexpr (RES && g + 1 < NG) {
        static_for<NPRE>([&](auto uc) { load_res(prev, IC<tn>{}, IC<ubn + decltype(uc)::value>{}, R[decltype(uc)::value]); });
        WP_SCHED();
      }
      if constexpr (g == 0) {
        request_now(fill);      // unconditional (every item has at least two k-tiles): the waits below count on its eight operations
        WP_SCHED();
      }
      static_for<NPRE>([&](auto uc) { store_block(IC<t>{}, IC<ub + decltype(uc)::value>{}); });
      WP_SCHED();
#pragma unroll
      for (int gp = 0; gp < 4; ++gp) {
        f32x4 af, bf[NPRE];
        af = *reinterpret_cast<const f32x4*>(cs + a_row + t * 32 * BK + sw[gp]);
#pragma unroll
        for (int u = 0; u < NPRE; ++u) bf[u] = *reinterpret_cast<const f32x4*>(cs + b_row + (ub + u) * 32 * BK + sw[gp]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int u = 0; u < NPRE; ++u)
            acc[t][ub + u] = (gp == 0 && kk == 0) ? WP_MFMA(af[kk], bf[u][kk], zero16) : WP_MFMA(af[kk], bf[u][kk], acc[t][ub + u]);
      }
      WP_SCHED();
      if constexpr (RES && g + 1 < NG) {
        __builtin_amdgcn_s_waitcnt(wait_vm(16 * NPRE + (g == 0 ? 8 : 0)));     // the next group's residual has landed
        WP_SCHED();
      }
    });
    // end of the iteration: the request has landed; the last group's stores may stay in flight (they are younger than the request)
    WP_SCHED();
    __builtin_amdgcn_s_waitcnt(wait_vm(16 * NPRE));
    cursor_advance();
    __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WP_TRACE(0)
    ph ^= 1;
  };
  // the epilogue of the last item, with nothing under it
  auto drain = [&]() {
    finish_ss();
    static_for<NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value, t = g / GPT, ub = (g % GPT) * NPRE;
      constexpr int tn = (g + 1) / GPT, ubn = ((g + 1) % GPT) * NPRE;
      static_for<NPRE>([&](auto uc) { finish_block(IC<t>{}, IC<ub + decltype(uc)::value>{}, R[decltype(uc)::value]); });
      WP_SCHED();
      if constexpr (RES && g + 1 < NG) {
        static_for<NPRE>([&](auto uc) { load_res(prev, IC<tn>{}, IC<ubn + decltype(uc)::value>{}, R[decltype(uc)::value]); });
        WP_SCHED();
      }
      static_for<NPRE>([&](auto uc) { store_block(IC<t>{}, IC<ub + decltype(uc)::value>{}); });
      WP_SCHED();
      if constexpr (RES && g + 1 < NG) __builtin_amdgcn_s_waitcnt(wait_vm(16 * NPRE));
    });
  };

  // ---- staggered start (p.stagger: the launch's spread in units of 64 x 127 cycles ~ 3.4 us).  All workgroups walk tiles of the
  // same length, so without it every CU reaches its tile boundary in the same few microseconds and the whole chip writes 64 MiB of
  // output (and reads as much residual) in one burst that HBM cannot absorb inside one iteration; spread over a few iterations the
  // same traffic hides under the MFMAs.  Workgroups that carry a tail part start first (their lists are longer).
  if (p.stagger > 0) {
    const int G = gridDim.x, bid = blockIdx.x;
    const int lw = (bid & 7) * (G >> 3) + (bid >> 3);
    const int slot = (int)(((long long)((lw * 97) % G) * p.stagger) / G);
    for (int i = 0; i < slot; ++i) __builtin_amdgcn_s_sleep(127);
  }

  // ---- prologue: the first k-tile ----
  cursor_open(0);
  request(smem);
  cursor_advance();
  __builtin_amdgcn_s_waitcnt(kWaitVm0);
  __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- the item loop: every item has at least two k-tiles (the launcher's plan), so an item is
  //   [its first iteration, carrying the previous item's epilogue]  [steady iterations]  [its last iteration, prefetching its own epilogue's first needs]
  // (ONE copy of each iteration kind: a second instance of the first iteration -- peeled for the workgroup's first item, or for
  // one-k-tile items -- sent hipcc's register allocation into hundreds of spills.) ----
  const bool has_res = RES && p.res != nullptr;
  auto epi_of = [&](const WItem& it) {
    Epi e;
    const size_t tile_el = (size_t)it.m0 * p.cout + (size_t)it.nt * BN;
    e.raw = it.part >= 0;
    if (e.raw) {
      e.out = reinterpret_cast<char*>(p.partial + (size_t)it.part * (BM * BN));
      e.os = BN * 4;
      e.res = reinterpret_cast<const char*>(p.zeros);
      e.rs = 0;
    } else {
      e.out = reinterpret_cast<char*>(p.y + tile_el);
      e.os = (unsigned)p.cout * 4;
      e.res = reinterpret_cast<const char*>(has_res ? p.res + tile_el : p.zeros);
      e.rs = has_res ? (unsigned)p.cout * 4 : 0;
    }
    e.ss = reinterpret_cast<const char*>(p.scale + it.nt * BN);
    return e;
  };
  for (int ci = 0; ci < n_items; ++ci) {
    const WItem it = item_at(ci);
    cur = epi_of(it);
    const int nk = it.kt1 - it.kt0;
    iter_first();
    for (int kt = 1; kt + 1 < nk; ++kt) iter_steady();
    iter_last();
    prev = cur;
    prev_lo = (p.relu != 0 && !cur.raw) ? 0.f : -__builtin_huge_valf();
  }
  drain();
#undef WP_SCHED
#undef WP_MFMA
#undef WP_TRACE
}

}  // namespace

// Which pointwise layers take the persistent 256 x 256 kernel: stride 1, one running sum, no weight groups, whole 256-wide n-tiles of
// 128-wide packed weights, whole 256-row m-tiles, at least pw256wp_mink input channels and pw256wp_mintiles tiles.  (Operand and
// output bases are 64-bit wave-uniform values, lane offsets stay inside one tile: no 4 GiB limit on any tensor.)
bool conv_pw_uses_256wp(int cout, long long M, int stride, int mt_per_group, int bn_tile, int c1, int c2, int flush_ktiles) {
  const int min_k = (int)opt(OPT_PW256WP_MINK);
  const int cin = c1 + c2;
  if (min_k <= 0 || bn_tile != 128 || cin < min_k || cin < 64 || stride != 1 || mt_per_group != 0 || flush_ktiles != 0) return false;
  if (cout % 256 != 0 || M % 256 != 0 || c1 % 32 != 0 || c2 % 32 != 0) return false;
  return (M / 256) * (cout / 256) >= opt(OPT_PW256WP_MINTILES);
}

// returns 1 (nothing launched) when the tail's partial tiles do not fit the scratch
int launch_conv_pw256wp(const ConvKParams& p0, float* ws, size_t ws_floats, hipStream_t stream) {
  ConvKParams p = p0;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8)
    return fail(-3, "conv_pw256wp: no current device");
  constexpr size_t kTile = (size_t)256 * 256;
  p.ntiles = p.cout / 256;
  const int mtiles = p.M / 256;
  const int T = mtiles * p.ntiles;
  const int G = cus - cus % 8;
  // The plan of launch_conv_pw256p: every workgroup the same number of whole tiles; the T mod G tiles left over as a uniform
  // split (small tails) or as one stream of k-tiles in equal runs (tails of at least a quarter of a round).
  const int t = T % G;
  int sp = 1;
  if (t > 0) {
    double best = 1e30;
    for (int cand = 1; cand <= 16 && p.nkt / cand >= 2; ++cand) {
      if (ws == nullptr || (size_t)t * cand * kTile > ws_floats) break;
      const double parts = (double)(((long long)t * cand + G - 1) / G);
      const double cost = parts * ((double)p.nkt / cand + 1.0);
      if (cost < best - 1e-9) { best = cost; sp = cand; }
    }
  }
  p.split_p = sp;
  p.n_sp = t * sp;
  p.n_full = T - t;
  // stream-K tail in units of TWO k-tiles: every fragment then has at least the two iterations the kernel's item loop needs
  p.sk_units = 0; p.sk_maxp = 0; p.sk_g = 0; p.sk_q = 2;
  if (t * 4 >= G && opt(OPT_PW256P_STREAMK) != 0 && ws != nullptr && p.nkt % 2 == 0) {
    const int upt = p.nkt / 2;
    const long long U = (long long)t * upt;
    const int Gs = (int)std::min<long long>(G, std::max<long long>(1, U / 2));
    const int run = (int)(U / Gs);                                  // shortest run, in units
    const int maxp = run > 0 ? (upt + run - 1) / run + 1 : 0;
    const double parts_now = (double)(((long long)t * sp + G - 1) / G);
    const double cost_now = parts_now * ((double)p.nkt / sp + 1.0);
    const double cost_stream = (double)((U + Gs - 1) / Gs) * 2 + 2.0;
    if (run >= 2 && (size_t)t * maxp * kTile <= ws_floats && cost_stream < cost_now - 0.5) {
      p.sk_units = (int)U; p.sk_maxp = maxp; p.sk_g = Gs;
      p.n_sp = 0; p.split_p = 1;
    }
  }
  // scratch: the tail's raw partial tiles, then one dump tile (the target of every workgroup's first, empty epilogue)
  const size_t part_tiles = p.sk_units > 0 ? (size_t)t * p.sk_maxp : (size_t)p.n_sp;
  if (!ws || (part_tiles + 1) * kTile > ws_floats) return 1;                    // no scratch: the caller takes another kernel
  p.dump = ws + part_tiles * kTile;
  if (p.n_full / G + (p.n_sp + G - 1) / G + 4 > kMaxItems) return 1;            // the workgroup's plan table (whole tiles + its tail parts / <= 3 fragments)
  p.partial = ws;
  p.mtiles = mtiles;
  p.nchunk = (int)opt(OPT_NCHUNK);
  p.phase_shift = opt(OPT_PW256_PHASE) != 0;
  p.p_order = opt(OPT_PW256P_ORDER) != 0;
  p.mt_per_group = 0;
  p.stagger = (int)opt(OPT_PW256WP_STAGGER);
  note_kernel("conv_pw_glds_256x256p");
  // blocks per epilogue group: with a residual two (32 residual registers; four would need 64 and the kernel then sits at the 256-register
  // limit with spills), without one four (fewer repeated fragment reads in an item's first iteration); pw256wp_npre = 2 forces two
  // the residual variant exists with two blocks per group only: <true, 4> compiled to 255 VGPRs + 4 spilled (20 B of scratch) and was
  // removed in round 6 -- pw256wp_npre = 4 applies to the layers without a residual
  const long long npre_opt = opt(OPT_PW256WP_NPRE);
  const bool npre2 = npre_opt == 2 || p.res != nullptr;
  if (p.res) {
    hipLaunchKernelGGL((conv_pw_glds256wp_kernel<true, 2>), dim3((unsigned)G), dim3(512), 0, stream, p);
  } else {
    if (npre2) hipLaunchKernelGGL((conv_pw_glds256wp_kernel<false, 2>), dim3((unsigned)G), dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((conv_pw_glds256wp_kernel<false, 4>), dim3((unsigned)G), dim3(512), 0, stream, p);
  }
  if (p.n_sp > 0 || p.sk_units > 0)
    hipLaunchKernelGGL((conv_splitk_reduce_kernel<256, 256>), dim3((unsigned)t, 256 / 16), dim3(256), 0, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-3, std::string("conv_pw256wp launch: ") + hipGetErrorString(e));
  return 0;
}

}  // namespace peanut
