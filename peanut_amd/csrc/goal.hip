// Long-term goal selection on the device: the reference's
// Agent_State.update_global_goal (nav/agent/agent_state.py:376-415) and the geodesic distance transform of
// FMMPlanner.set_goal / set_multi_goal (nav/agent/utils/fmm_planner.py:55-75).
//
// The reference computes the geodesic field with scikit-fmm (`skfmm.distance`, heap-ordered fast marching, second
// order) on the host: ~0.2 s per call on a 960x960 map, the largest CPU cost per step once stages 1-3 run on the
// GPU.  A heap is inherently serial.  Here the same discretisation (second-order upwind differences with the
// first-order fallback, masked cells excluded from every stencil; oracle/fmm_ref.c states the update rule that is
// mirrored operation by operation, in double) is solved by relaxation, arranged so that every stage provably ends:
//
//   * tiles: the map is cut into 32x32 tiles; a workgroup stages a tile plus a 2-cell halo in LDS and relaxes it until nothing
//     in the tile changes -- since round 5 block-wise (fmm_round_blocked_kernel): each of the 16 waves owns an 8 x 8 block, keeps
//     a private copy of it with its two-cell ring and iterates it to convergence between two workgroup barriers (Jacobi inside
//     the wave, which needs no barrier), then publishes it; the older form (fmm_round_kernel, option fmm_blocked = 0) sweeps
//     the whole tile once per barrier pair.  A tile that changed wakes its four neighbours for the next round (the stencil is
//     axis-aligned: no diagonal dependency); rounds are plain launches over the tile grid in which sleeping tiles exit at
//     once; the host reads one counter every few rounds.  Only a ring of tiles is awake in a round, so a round costs the
//     LATENCY of its slowest tile: the number of iterations the front needs to cross it times the dependent instruction
//     chain of one update (profiles/r8/README.md), which is what the blocked form and update_cell_local_fast shorten;
//   * stage A, first order: u <- min(u, update(u)) from +inf.  Monotone, hence convergent, to the unique first-order
//     field u1 (the classic fast iterative method);
//   * stage B, second order, on a FIXED dependency graph: a neighbour may feed a cell only if it precedes the cell in
//     a given ordering field `ord` (stage A's u1 at first).  The graph is acyclic, so a plain fixed-point sweep
//     u <- update(u) settles cell after cell in that order -- no "keep the smaller" is needed (the second-order term
//     is not monotone in its second neighbour: keeping minima would freeze transients) and none of the feedback that
//     makes an unrestricted second-order relaxation oscillate (two cells each taking the other for its upwind
//     neighbour) can occur;
//   * stage B is repeated with `ord` = its own result, warm-started (later passes only touch the cells whose order
//     moved).  Iterated until nothing changes (5-6 passes) the result is the self-consistent second-order field, 0.14
//     cell from the heap-ordered march on maze maps (reached within six passes, the cap),
//     which bounds the cost: every pass is a front propagation across the map, ~1.5 ms at 960 x 960.
//
// Difference to the heap-ordered march: when the two-axis root falls BELOW the value of the later of its two
// neighbours (possible with second-order terms next to walls), scikit-fmm's march keeps it -- the neighbour was
// already frozen -- while the relaxation keeps the causal one-axis root.  Measured on maze maps: <= 0.14 cell
// max-abs, far from walls identical to rounding; tests hold the field to <= 0.5 cell of the oracle and the selected
// goal cell to equality.
//
// The rest of update_global_goal is fused around it: obstacle dilation by the collision disk + collision / visited
// overrides -> traversible map (:382-386), exp(-d / temperature) weights with the "stuck: keep the last weights"
// rule (:395-399), value = target_pred * weights and its first-occurrence argmax (:401-413).
//
// The field needs the map, not the prediction: peanut_goal_select_begin puts the traversible map, the initialisation and the first
// batch of relaxation rounds on a stream of the handle at once; the forward the caller enqueues next (Agent_State.update_state:
// update_prediction then update_global_goal) runs beside them, peanut_goal_select continues the solve on that stream, and the
// caller's stream joins before the weights are formed.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "net_common.h"

#pragma clang fp contract(off)

namespace peanut {
namespace {

constexpr int TILE = 32, HALO = 2, LT = TILE + 2 * HALO;   // 36
constexpr int MAX_SWEEPS = 96;                             // per round and tile (a front crosses a tile in <= 64)
constexpr int ROUNDS_PER_CHECK = 8, MAX_ROUNDS_PER_CHECK = 96;
constexpr int MAX_ORDER_PASSES = 24;    // default of option fmm_max_passes: a ceiling, the passes end at their fixed point (round 5; a cap of six left the deployed 960 x 960 map unconverged); measured distance to the heap-ordered march: 3 passes 0.39 / 0.70 cell (maze / cluttered map), 4 passes 0.14 / 0.47, fixed point (5-6) 0.14 / 0.03

enum : unsigned char { ST_MASKED = 0, ST_FREE = 1, ST_SEED = 2 };

// ---- traversible map: ~dilate(rint(obstacle), disk(rad)), collision -> 0, visited -> 1  (agent_state.py:382-386) ----
// A workgroup covers 32 x 8 cells.  The obstacle bits of its rows (8 + 2 rad of them, 32 + 2 rad columns wide: one 64-bit word per
// row while rad <= 16) are formed once with a ballot per row; a cell is blocked when any row of its disk has an obstacle bit under
// that row's chord -- 2 rad + 1 AND-tests instead of (2 rad + 1)^2 loads (85 -> ~12 us on the 960 x 960 map).
constexpr int TRAV_MAX_RAD = 16;
__global__ __launch_bounds__(256) void goal_trav_kernel(const float* __restrict__ obst, const unsigned char* __restrict__ collision,
                                                        const unsigned char* __restrict__ visited, int H, int W, int rad,
                                                        unsigned char* __restrict__ trav) {
  __shared__ unsigned long long rowbits[8 + 2 * TRAV_MAX_RAD];
  __shared__ int chord[2 * TRAV_MAX_RAD + 1];      // half-width of the disk at row offset dy
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 8;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int span = 32 + 2 * rad;                    // <= 64
  for (int rr = wave; rr < 8 + 2 * rad; rr += 4) {
    const int r = r0 - rad + rr, c = c0 - rad + lane;
    const bool ob = lane < span && r >= 0 && r < H && c >= 0 && c < W && rintf(obst[(size_t)r * W + c]) != 0.f;
    const unsigned long long m = __ballot(ob);
    if (lane == 0) rowbits[rr] = m;
  }
  if (threadIdx.x <= 2 * rad) {
    const int dy = (int)threadIdx.x - rad;
    int w = 0;
    while ((w + 1) * (w + 1) + dy * dy <= rad * rad) ++w;
    chord[threadIdx.x] = w;
  }
  __syncthreads();
  const int lc = threadIdx.x & 31, lr = threadIdx.x >> 5;
  const int c = c0 + lc, r = r0 + lr;
  if (r >= H || c >= W) return;
  bool blocked = false;
  for (int dy = -rad; dy <= rad; ++dy) {
    const int w = chord[dy + rad];
    // bits lc + rad - w .. lc + rad + w of the row (bit b = column c0 - rad + b)
    const unsigned long long mask = ((w + w + 1 >= 64) ? ~0ull : ((1ull << (w + w + 1)) - 1ull)) << (lc + rad - w);
    blocked = blocked || (rowbits[lr + rad + dy] & mask) != 0ull;
  }
  unsigned char t = blocked ? 0 : 1;
  const size_t i = (size_t)r * W + c;
  if (collision && collision[i] == 1) t = 0;
  if (visited && visited[i] == 1) t = 1;
  trav[i] = t;
}
// any radius (the footprint test cell by cell): used beyond TRAV_MAX_RAD
__global__ __launch_bounds__(256) void goal_trav_wide_kernel(const float* __restrict__ obst, const unsigned char* __restrict__ collision,
                                                             const unsigned char* __restrict__ visited, int H, int W, int rad,
                                                             unsigned char* __restrict__ trav) {
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), r = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (r >= H || c >= W) return;
  bool blocked = false;
  for (int dy = -rad; dy <= rad && !blocked; ++dy) {
    const int rr = r + dy;
    if (rr < 0 || rr >= H) continue;
    for (int dx = -rad; dx <= rad; ++dx) {
      if (dx * dx + dy * dy > rad * rad) continue;
      const int cc = c + dx;
      if (cc < 0 || cc >= W) continue;
      if (rintf(obst[(size_t)rr * W + cc]) != 0.f) { blocked = true; break; }
    }
  }
  unsigned char t = blocked ? 0 : 1;
  const size_t i = (size_t)r * W + c;
  if (collision && collision[i] == 1) t = 0;
  if (visited && visited[i] == 1) t = 1;
  trav[i] = t;
}

// state / distance initialisation: masked where not traversible, seeds (one cell and/or a mask) at distance 0
__global__ __launch_bounds__(256) void fmm_init_kernel(const unsigned char* __restrict__ trav, const unsigned char* __restrict__ seed_mask,
                                                       int seed_r, int seed_c, int H, int W, unsigned char* __restrict__ state,
                                                       double* __restrict__ dist, unsigned char* __restrict__ active, int tiles_x) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int r = i / W, c = i - r * W;
  const bool seed = (r == seed_r && c == seed_c) || (seed_mask && seed_mask[i] == 1);
  state[i] = seed ? ST_SEED : (trav[i] ? ST_FREE : ST_MASKED);
  dist[i] = seed ? 0.0 : INFINITY;
  if (seed) active[(r / TILE) * tiles_x + (c / TILE)] = 1;
}

// one axis of distanceMarcher::updatePointOrderTwo.  m1/m2 = the values one and two steps towards smaller indices,
// p1/p2 towards larger ones; INFINITY = masked / outside / not reached / not allowed to feed this cell.
struct AxisTerm { double a, b, c, v1; };
__device__ __forceinline__ AxisTerm axis_term(double m1, double m2, double p1, double p2) {
  AxisTerm t{0.0, 0.0, 0.0, INFINITY};
  double v1 = INFINITY, v2 = INFINITY;
  // j = -1 first; j = +1 only when strictly closer; the second neighbour when it is not farther than the first (<=).
  // The library's loop leaves a v2 found in the j = -1 direction in place when j = +1 then supplies the closer v1 without
  // a qualifying second neighbour (oracle/fmm_ref.c keeps that); here v2 always belongs to v1's own side: the mixed pair
  // (v1 from one side, v2 from the other) only arises where two fronts meet, is not an upwind difference, and with it the
  // ordering passes of this solver stop reaching a fixed point (measured: profiles/r3f)
  if (m1 < v1) { v1 = m1; v2 = (m2 <= v1) ? m2 : INFINITY; }
  if (p1 < v1) { v1 = p1; v2 = (p2 <= v1) ? p2 : INFINITY; }
  t.v1 = v1;
  if (v2 < INFINITY) {
    const double aa = 9.0 / 4.0, tp = (1.0 / 3.0) * (4.0 * v1 - v2);
    t.a = aa; t.b = -2.0 * aa * tp; t.c = aa * tp * tp;
  } else if (v1 < INFINITY) {
    t.a = 1.0; t.b = -2.0 * v1; t.c = v1 * v1;
  }
  return t;
}
__device__ __forceinline__ double solve_root(double a, double b, double c) {
  c -= 1.0;
  const double det = b * b - 4.0 * a * c;
  return det >= 0.0 ? (-b + sqrt(det)) / 2.0 / a : -1.0;
}
// Upwind selection: the axis with the smaller neighbour alone first; the other axis joins only when its neighbour
// lies strictly below that one-axis value (it is then upwind of the cell) and the joint root stays above it.
__device__ __forceinline__ double update_cell(const AxisTerm& y, const AxisTerm& x) {
  const bool hy = y.v1 < INFINITY, hx = x.v1 < INFINITY;
  if (!hy && !hx) return INFINITY;
  const bool y_first = hy && (!hx || y.v1 <= x.v1);
  const AxisTerm& s = y_first ? y : x;
  const AxisTerm& o = y_first ? x : y;
  const double u1 = solve_root(s.a, s.b, s.c);
  if (o.v1 < u1) {
    const double u2 = solve_root(s.a + o.a, s.b + o.b, s.c + o.c);
    if (u2 > o.v1) return u2;
  }
  return u1;
}

// The same update with the quadratic solved RELATIVE to the smaller upwind value, in single precision: the unknown
// is u - base with base = min(v1_y, v1_x), every input a difference of neighbouring values (O(1) cells), so float
// carries ~1e-7 cell per update where the double form needs its 15 digits only to cancel the common offset (~10^2..10^3
// cells) out of b^2 - 4ac.  A fraction of the double instruction count (v_sqrt_f32 instead of a double square-root
// sequence); the result is added back onto `base` in double.
struct AxisPick { double v1, v2; };      // nearest upwind value and the one behind it (INFINITY: none)
__device__ __forceinline__ AxisPick axis_pick(double m1, double m2, double p1, double p2) {
  AxisPick t{INFINITY, INFINITY};
  if (m1 < t.v1) { t.v1 = m1; t.v2 = (m2 <= t.v1) ? m2 : INFINITY; }      // as axis_term
  if (p1 < t.v1) { t.v1 = p1; t.v2 = (p2 <= t.v1) ? p2 : INFINITY; }
  return t;
}
__device__ __forceinline__ double update_cell_local(const AxisPick& y, const AxisPick& x) {
  const bool hy = y.v1 < INFINITY, hx = x.v1 < INFINITY;
  if (!hy && !hx) return INFINITY;
  const bool y_first = hy && (!hx || y.v1 <= x.v1);
  const AxisPick& s = y_first ? y : x;
  const AxisPick& o = y_first ? x : y;
  const double base = s.v1;
  const bool s2 = s.v2 < INFINITY;
  const float as = s2 ? 2.25f : 1.0f;
  const float ts = s2 ? (float)(s.v1 - s.v2) * (1.0f / 3.0f) : 0.0f;      // (4 v1 - v2) / 3 - base
  float r = ts + (s2 ? (2.0f / 3.0f) : 1.0f);                             // one-axis root: t + 1 / sqrt(a)
  if (o.v1 < INFINITY) {
    const float ov = (float)(o.v1 - base);
    if (ov < r) {
      const bool o2 = o.v2 < INFINITY;
      const float ao = o2 ? 2.25f : 1.0f;
      const float to = o2 ? ov + (float)(o.v1 - o.v2) * (1.0f / 3.0f) : ov;
      const float A = as + ao, B = as * ts + ao * to, C = as * ts * ts + ao * to * to - 1.0f;
      const float det = B * B - A * C;
      if (det >= 0.0f) {
        const float u2 = (B + sqrtf(det)) / A;
        if (u2 > ov) r = u2;
      }
    }
  }
  return base + (double)r;
}

// update_cell_local(axis_pick(ym1, ym2, yp1, yp2), axis_pick(xm1, xm2, xp1, xp2)) written for a short dependent instruction
// chain: the blocked round kernel below is bound by the latency of one evaluation (one wave advances the front one cell per
// iteration, and the waves of a tile share four SIMDs), and the form above compiles to ~245 instructions.  Same formulas;
// the picks through min / max / compare-select instead of nested updates, no structures passed by reference, the hardware's
// 1-ulp square root and a reciprocal constant (a + a' is 2, 3.25 or 4.5) instead of the correctly rounded sequences: the
// result differs from the form above by <= 2 ulp of a float around 1 (~2e-7 cell), below the 1e-6 cell at which stage B
// already treats a change as noise.  Values that must not feed the cell arrive as INFINITY (the caller reads them from a
// slot that holds it); SECOND = false: no second neighbours (stage A).
template <bool SECOND>
__device__ __forceinline__ double update_cell_local_fast(double ym1, double ym2, double yp1, double yp2, double xm1, double xm2,
                                                          double xp1, double xp2) {
  double v1y, v1x, v2y = INFINITY, v2x = INFINITY;
  if (SECOND) {
    // axis_pick: the j = -1 side unless the j = +1 side is strictly closer; the second neighbour of that side when it is not farther
    const bool ty = yp1 < ym1, tx = xp1 < xm1;
    v1y = ty ? yp1 : ym1;
    v1x = tx ? xp1 : xm1;
    const double c2y = ty ? yp2 : ym2, c2x = tx ? xp2 : xm2;
    v2y = (c2y <= v1y) ? c2y : INFINITY;
    v2x = (c2x <= v1x) ? c2x : INFINITY;
  } else {
    v1y = fmin(ym1, yp1);
    v1x = fmin(xm1, xp1);
  }
  // the axis with the smaller neighbour first (equal values: either)
  const double s1 = fmin(v1y, v1x), o1 = fmax(v1y, v1x);
  float as = 1.0f, ao = 1.0f, ts = 0.0f, r = 1.0f, rA = 0.5f;
  const float ov = (float)(o1 - s1);                                       // +inf when the other axis has no upwind neighbour
  float to = ov;
  if (SECOND) {
    const bool y_first = v1y <= v1x;
    const double s2v = y_first ? v2y : v2x, o2v = y_first ? v2x : v2y;
    const bool s2 = s2v < INFINITY, o2 = o2v < INFINITY;
    as = s2 ? 2.25f : 1.0f;
    ts = s2 ? (float)(s1 - s2v) * (1.0f / 3.0f) : 0.0f;
    r = ts + (s2 ? (2.0f / 3.0f) : 1.0f);
    ao = o2 ? 2.25f : 1.0f;
    to = o2 ? ov + (float)(o1 - o2v) * (1.0f / 3.0f) : ov;
    rA = (s2 == o2) ? (s2 ? (1.0f / 4.5f) : 0.5f) : (1.0f / 3.25f);
  }
  const float A = as + ao, B = as * ts + ao * to, C = as * ts * ts + ao * to * to - 1.0f;
  // the two products pinned to registers of their own: hipcc otherwise forms (B, A) * (B, C) and takes the difference with a packed add
  // that reads the high register into its low half (op_sel) -- the form that is not safe beside fp16 / bf16 MFMA waves on gfx950
  // (common.h, PEANUT_NO_PK_F32; this kernel runs beside the prediction forward).  Same arithmetic: two products, one subtraction.
  float bb = B * B, ac = A * C;
  asm volatile("" : "+v"(bb), "+v"(ac));
  const float det = bb - ac;
  const float u2 = (B + __builtin_amdgcn_sqrtf(det)) * rA;
  if (ov < r && det >= 0.0f && u2 > ov) r = u2;                            // (ov = +inf: no joint root)
  return s1 < INFINITY ? s1 + (double)r : INFINITY;
}

// SECOND = false: stage A (first order, monotone).  SECOND = true: stage B (second order on the graph given by ord).
// Jacobi sweeps of the tile in LDS, one cell per lane.  A cell is only re-evaluated when one of the cells its stencil
// reads changed in the previous sweep (the update is a pure function of those and of its own value, so skipping it
// changes nothing): once the tile is loaded, the work follows the front -- a band a few cells wide -- instead of all
// 1024 cells, and with one 8 x 8 block per wave most waves skip a sweep altogether.  Only a ring of tiles is awake in
// a round (<= 64 of 900 on the agent's map), so a round costs the LATENCY of one tile: sweeps x time per sweep; hence
// one evaluation per lane (1024 lanes) rather than four, the feed predicates hoisted out of the sweeps, and the
// quadratic solved in single precision relative to the smaller upwind value (update_cell_local).  `active_clear`
// (the flags of the round after next) is wiped here, one byte per tile, which saves a fill launch per round.
template <bool SECOND, bool LOCAL32>
__global__ __launch_bounds__(1024) void fmm_round_kernel(double* __restrict__ dist, const double* __restrict__ ord,
                                                         const unsigned char* __restrict__ state, int H, int W, int tiles_x, int tiles_y,
                                                         const unsigned char* __restrict__ active_in, unsigned char* __restrict__ active_out,
                                                         unsigned char* __restrict__ active_clear, unsigned int* __restrict__ changed_tiles) {
  const int tile = blockIdx.x;
  if (threadIdx.x == 0) active_clear[tile] = 0;
  if (!active_in[tile]) return;
  __shared__ double d[LT][LT + 1];
  __shared__ double o[SECOND ? LT : 1][LT + 1];
  __shared__ unsigned char st[LT][LT + 4];
  __shared__ unsigned char chg[2][LT][LT + 4];      // cells that changed in the previous / in this sweep (halo: never)
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int r0 = ty * TILE - HALO, c0 = tx * TILE - HALO;
  for (int i = threadIdx.x; i < LT * LT; i += 1024) {
    const int ly = i / LT, lx = i - ly * LT;
    const int r = r0 + ly, c = c0 + lx;
    const bool in = r >= 0 && r < H && c >= 0 && c < W;
    d[ly][lx] = in ? dist[(size_t)r * W + c] : INFINITY;
    if (SECOND) o[ly][lx] = in ? ord[(size_t)r * W + c] : INFINITY;
    st[ly][lx] = in ? state[(size_t)r * W + c] : ST_MASKED;
    chg[0][ly][lx] = 0;
    chg[1][ly][lx] = 0;
  }
  __syncthreads();
  // one cell per lane, one 8 x 8 block per wave (16 waves): a wave none of whose cells saw a change around it skips
  // the sweep as a whole, so the sweep costs one evaluation on the waves the front passes through and nothing elsewhere
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ly = HALO + 8 * (wave >> 2) + (lane >> 3), lx = HALO + 8 * (wave & 3) + (lane & 7);
  // which neighbours may feed the cell: fixed for the round (state and `ord` do not change), so worked out once --
  // bit 0..3: y-1, y+1, x-1, x+1 precede the cell; bit 4..7: y-2, y+2, x-2, x+2 precede those; bit 8: the cell is
  // free.  Stage A: every neighbour may feed, second neighbours never.
  unsigned feed = st[ly][lx] == ST_FREE ? 0x100u : 0u;
  if (SECOND) {
    const double oi = o[ly][lx];
    const double oym = o[ly - 1][lx], oyp = o[ly + 1][lx], oxm = o[ly][lx - 1], oxp = o[ly][lx + 1];
    feed |= (oym < oi ? 1u : 0u) | (oyp < oi ? 2u : 0u) | (oxm < oi ? 4u : 0u) | (oxp < oi ? 8u : 0u);
    // (<=: two equal seeds in line feed the cell next to them; the graph stays acyclic, the first step is strict)
    feed |= (o[ly - 2][lx] <= oym ? 16u : 0u) | (o[ly + 2][lx] <= oyp ? 32u : 0u) | (o[ly][lx - 2] <= oxm ? 64u : 0u) |
            (o[ly][lx + 2] <= oxp ? 128u : 0u);
  } else {
    feed |= 0xfu;
  }
  bool ever = false;
  int sweeps = 0;
  for (; sweeps < MAX_SWEEPS; ++sweeps) {
    const int cur = sweeps & 1;
    const double old = d[ly][lx];
    double nv = old;
    bool need = sweeps == 0;        // a tile wakes because a neighbour changed its halo: everything once
    if (!need) {
      need = chg[cur][ly - 1][lx] | chg[cur][ly + 1][lx] | chg[cur][ly][lx - 1] | chg[cur][ly][lx + 1];
      if (SECOND) need |= chg[cur][ly - 2][lx] | chg[cur][ly + 2][lx] | chg[cur][ly][lx - 2] | chg[cur][ly][lx + 2];
    }
    if (need && (feed & 0x100u)) {
      // a neighbour feeds the cell only if it precedes it in `ord`; the second one only if it precedes the first
      const double ym1 = (feed & 1u) ? d[ly - 1][lx] : INFINITY, yp1 = (feed & 2u) ? d[ly + 1][lx] : INFINITY;
      const double xm1 = (feed & 4u) ? d[ly][lx - 1] : INFINITY, xp1 = (feed & 8u) ? d[ly][lx + 1] : INFINITY;
      double ym2 = INFINITY, yp2 = INFINITY, xm2 = INFINITY, xp2 = INFINITY;
      if (SECOND) {
        ym2 = (feed & 16u) ? d[ly - 2][lx] : INFINITY; yp2 = (feed & 32u) ? d[ly + 2][lx] : INFINITY;
        xm2 = (feed & 64u) ? d[ly][lx - 2] : INFINITY; xp2 = (feed & 128u) ? d[ly][lx + 2] : INFINITY;
      }
      double u;
      if (LOCAL32) u = update_cell_local(axis_pick(ym1, ym2, yp1, yp2), axis_pick(xm1, xm2, xp1, xp2));
      else u = update_cell(axis_term(ym1, ym2, yp1, yp2), axis_term(xm1, xm2, xp1, xp2));
      if (SECOND) {
        // changes at the rounding level of the update (1e-12 relative in double; 1e-6 cell with the single-precision
        // local solve) are noise, not information
        if (old < INFINITY && fabs(u - old) <= (LOCAL32 ? 1e-6 : 1e-12 * fmax(1.0, old))) u = old;
      } else {
        u = fmin(u, old);
      }
      nv = u;
    }
    const bool chd = nv != old;
    const int any = __syncthreads_or(chd);
    if (!any) break;
    ever = true;
    d[ly][lx] = nv;
    chg[cur ^ 1][ly][lx] = chd;
    __syncthreads();
  }
  if (!ever) return;      // (uniform: `ever` derives from __syncthreads_or)
  {
    const int r = r0 + ly, c = c0 + lx;
    if (r < H && c < W) dist[(size_t)r * W + c] = d[ly][lx];
  }
  if (threadIdx.x == 0) {
    atomicAdd(changed_tiles, 1u);
    if (sweeps == MAX_SWEEPS) active_out[tile] = 1;       // not settled inside the budget: go on next round
    if (ty > 0) active_out[tile - tiles_x] = 1;
    if (ty + 1 < tiles_y) active_out[tile + tiles_x] = 1;
    if (tx > 0) active_out[tile - 1] = 1;
    if (tx + 1 < tiles_x) active_out[tile + 1] = 1;
  }
}

// The same round, blocked (round 5).  In the kernel above the front moves one cell per sweep and every sweep costs two
// workgroup barriers over 16 waves plus the change-tracking reads: ~1.1 us per sweep, 70-90 us for a front to cross a tile,
// and a round costs exactly that (only a ring of tiles is awake, so rounds are latency, not throughput).  Here a wave owns
// one 8 x 8 block of the tile and keeps a PRIVATE copy of it with its two-cell ring (12 x 12) in LDS: between two workgroup
// barriers it relaxes the block to convergence against the frozen ring -- Jacobi iterations inside the wave, which runs in
// lockstep, so they need no barrier at all (LDS operations of one wave execute in order) -- and then publishes the block.
// A front crosses a tile in 4-8 barrier pairs instead of 32-64, and an iteration costs the update alone.  A block is
// relaxed again only when one of its four neighbour blocks (the stencil is axis-aligned) published a change.  The
// schedule is deterministic (every block reads the tile as of the last barrier); stage A's monotone field and stage B's
// fixed dependency graph make the result independent of the schedule up to the update's own noise floor.
constexpr int BLK = 8, PT = BLK + 2 * HALO, MAX_OUTER = 64;      // private tile 12 x 12
template <bool SECOND, bool LOCAL32>
__global__ __launch_bounds__(1024) void fmm_round_blocked_kernel(double* __restrict__ dist, const double* __restrict__ ord,
                                                                 const unsigned char* __restrict__ state, int H, int W, int tiles_x, int tiles_y,
                                                                 const unsigned char* __restrict__ active_in, unsigned char* __restrict__ active_out,
                                                                 unsigned char* __restrict__ active_clear, unsigned int* __restrict__ changed_tiles,
                                                                 int max_inner) {
  const int tile = blockIdx.x;
  if (threadIdx.x == 0) active_clear[tile] = 0;
  if (!active_in[tile]) return;
  __shared__ double d[LT][LT + 1];
  __shared__ double o[SECOND ? LT : 1][LT + 1];
  __shared__ unsigned char st[LT][LT + 4];
  __shared__ double priv[16][PT * (PT + 1) + 2];      // per wave: 12 rows of 13 (one of padding) + a slot that holds +inf
  __shared__ unsigned char blk[2][6][8];      // block (by, bx) at [by + 1][bx + 1]; the ring around the 4 x 4 blocks stays 0
  __shared__ int anyflag[2];
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int r0 = ty * TILE - HALO, c0 = tx * TILE - HALO;
  for (int i = threadIdx.x; i < LT * LT; i += 1024) {
    const int ly = i / LT, lx = i - ly * LT;
    const int r = r0 + ly, c = c0 + lx;
    const bool in = r >= 0 && r < H && c >= 0 && c < W;
    d[ly][lx] = in ? dist[(size_t)r * W + c] : INFINITY;
    if (SECOND) o[ly][lx] = in ? ord[(size_t)r * W + c] : INFINITY;
    st[ly][lx] = in ? state[(size_t)r * W + c] : ST_MASKED;
  }
  if (threadIdx.x < 2 * 6 * 8) (&blk[0][0][0])[threadIdx.x] = 0;
  if (threadIdx.x < 2) anyflag[threadIdx.x] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // wave -> block through a Latin square (waves w, w + 4, w + 8, w + 12 share a SIMD): the blocks of a row AND of a column
  // of the tile sit on four different SIMDs, so a front that runs along either axis does not queue its blocks on one of them.
  // The square is SIMD = bx ^ m(by) with m = multiplication by x in GF(4) (0, 2, 3, 1: round 6), which also puts the two main
  // diagonals on four different SIMDs -- a front that crosses the tile obliquely; m = the Gray code (0, 1, 3, 2) of round 5 paired
  // them (solver on its own 1.92 / 1.84 -> 1.81 / 1.83 ms per call, beside the forward 4.52 / 4.54 -> 4.45 / 4.48: profiles/r9o).
  // Which wave relaxes a block does not enter the arithmetic.
  const int by = wave >> 2, bx = (wave & 3) ^ ((0x1320 >> (4 * by)) & 3);
  const int ly = HALO + BLK * by + (lane >> 3), lx = HALO + BLK * bx + (lane & 7);
  const int py = HALO + (lane >> 3), px = HALO + (lane & 7);
  unsigned feed = st[ly][lx] == ST_FREE ? 0x100u : 0u;       // as in fmm_round_kernel
  if (SECOND) {
    const double oi = o[ly][lx];
    const double oym = o[ly - 1][lx], oyp = o[ly + 1][lx], oxm = o[ly][lx - 1], oxp = o[ly][lx + 1];
    feed |= (oym < oi ? 1u : 0u) | (oyp < oi ? 2u : 0u) | (oxm < oi ? 4u : 0u) | (oxp < oi ? 8u : 0u);
    feed |= (o[ly - 2][lx] <= oym ? 16u : 0u) | (o[ly + 2][lx] <= oyp ? 32u : 0u) | (o[ly][lx - 2] <= oxm ? 64u : 0u) |
            (o[ly][lx + 2] <= oxp ? 128u : 0u);
  } else {
    feed |= 0xfu;
  }
  double* pv = priv[wave];
  constexpr int PS = PT + 1, INF_SLOT = PT * PS;
  if (lane == 0) pv[INF_SLOT] = INFINITY;
  // where the eight stencil values are read: the neighbour's cell of the private tile if it may feed this cell, the +inf slot if not
  const int pc = py * PS + px;
  const int a_ym1 = (feed & 1u) ? pc - PS : INF_SLOT, a_yp1 = (feed & 2u) ? pc + PS : INF_SLOT;
  const int a_xm1 = (feed & 4u) ? pc - 1 : INF_SLOT, a_xp1 = (feed & 8u) ? pc + 1 : INF_SLOT;
  const int a_ym2 = (feed & 16u) ? pc - 2 * PS : INF_SLOT, a_yp2 = (feed & 32u) ? pc + 2 * PS : INF_SLOT;
  const int a_xm2 = (feed & 64u) ? pc - 2 : INF_SLOT, a_xp2 = (feed & 128u) ? pc + 2 : INF_SLOT;
  const bool is_free = (feed & 0x100u) != 0;
  bool ever = false, again = false;      // again: the block's own iterations hit their cap
  int outer = 0;
  for (; outer < MAX_OUTER; ++outer) {
    const int cur = outer & 1, nxt = cur ^ 1;
    // the tile wakes because a neighbour changed its halo: every block once; afterwards only next to a published change
    const bool need = outer == 0 || again || blk[cur][by][bx + 1] || blk[cur][by + 2][bx + 1] || blk[cur][by + 1][bx] || blk[cur][by + 1][bx + 2];
    if (need) {      // (wave-uniform) private copy of the block and its ring, as of the last barrier
      for (int i = lane; i < PT * PT; i += 64) {
        const int qy = i / PT, qx = i - qy * PT;
        pv[qy * PS + qx] = d[BLK * by + qy][BLK * bx + qx];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) anyflag[cur] = 0;
    if (lane == 0) blk[cur][by + 1][bx + 1] = 0;      // (read before the barrier above; written again in the next outer step)
    if (need) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const double own0 = pv[pc];
      double val = own0;
      again = true;
      for (int inner = 0; inner < max_inner; ++inner) {
        const double ym1 = pv[a_ym1], yp1 = pv[a_yp1], xm1 = pv[a_xm1], xp1 = pv[a_xp1];
        double ym2 = INFINITY, yp2 = INFINITY, xm2 = INFINITY, xp2 = INFINITY;
        if (SECOND) { ym2 = pv[a_ym2]; yp2 = pv[a_yp2]; xm2 = pv[a_xm2]; xp2 = pv[a_xp2]; }
        double u;
        if (LOCAL32) u = update_cell_local_fast<SECOND>(ym1, ym2, yp1, yp2, xm1, xm2, xp1, xp2);
        else u = update_cell(axis_term(ym1, ym2, yp1, yp2), axis_term(xm1, xm2, xp1, xp2));
        if (SECOND) {
          if (val < INFINITY && fabs(u - val) <= (LOCAL32 ? 1e-6 : 1e-12 * fmax(1.0, val))) u = val;
        } else {
          u = fmin(u, val);
        }
        const double nv = is_free ? u : val;
        const bool chd = nv != val;
        if (!__any(chd)) { again = false; break; }
        // every lane has read before any lane writes (one instruction stream, LDS operations of a wave execute in order);
        // the next iteration reads after the write
        val = nv;
        asm volatile("" ::: "memory");
        pv[pc] = nv;
        asm volatile("" ::: "memory");
      }
      if (__any(val != own0)) {
        d[ly][lx] = val;
        if (lane == 0) { blk[nxt][by + 1][bx + 1] = 1; anyflag[nxt] = 1; }
      } else if (again && lane == 0) {
        anyflag[nxt] = 1;      // (cannot happen without a change; kept so that a capped block is never dropped)
      }
    }
    __syncthreads();
    if (!anyflag[nxt]) break;
    ever = true;
  }
  if (!ever) return;
  {
    const int r = r0 + ly, c = c0 + lx;
    if (r < H && c < W) dist[(size_t)r * W + c] = d[ly][lx];
  }
  if (threadIdx.x == 0) {
    atomicAdd(changed_tiles, 1u);
    if (outer == MAX_OUTER) active_out[tile] = 1;
    if (ty > 0) active_out[tile - tiles_x] = 1;
    if (ty + 1 < tiles_y) active_out[tile + tiles_x] = 1;
    if (tx > 0) active_out[tile - 1] = 1;
    if (tx + 1 < tiles_x) active_out[tile + 1] = 1;
  }
}

// stage B starts from the seeds again: distances back to +inf (seeds 0)
__global__ __launch_bounds__(256) void fmm_restart_kernel(const unsigned char* __restrict__ state, int n, double* __restrict__ dist) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dist[i] = state[i] == ST_SEED ? 0.0 : INFINITY;
}

// max over the reached cells (for `ma.filled(dd, np.max(dd) + 1)`); bits of a non-negative double order like an integer
__global__ __launch_bounds__(256) void fmm_max_kernel(const double* __restrict__ dist, int n, unsigned long long* __restrict__ max_bits) {
  double m = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double v = dist[i];
    if (v < INFINITY && v > m) m = v;
  }
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(max_bits, (unsigned long long)__double_as_longlong(m));
}
__global__ __launch_bounds__(256) void fmm_fill_kernel(const double* __restrict__ dist, int n, const unsigned long long* __restrict__ max_bits,
                                                       double* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double v = dist[i];
  out[i] = v < INFINITY ? v : __longlong_as_double((long long)*max_bits) + 1.0;
}

// ---- weights over the local window: exp(-dd / temperature) (:395-396), their sum (:398) ----
__global__ __launch_bounds__(256) void goal_weight_kernel(const double* __restrict__ dist, int W, int gx1, int gy1, int lw, int lh,
                                                          double temperature, int frontier, double* __restrict__ wt, double* __restrict__ sum) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double w = 0.0;
  if (i < lw * lh) {
    const int r = i / lh, c = i - r * lh;
    double dd = dist[(size_t)(gx1 + r) * W + (gy1 + c)];
    if (frontier) {            // dist_weight_temperature == 0: frontier-based exploration (:404-406)
      if (dd < 60.0) dd = INFINITY;
      w = exp(-dd / 100.0);
    } else {
      w = exp(-dd / temperature);
    }
    wt[i] = w;
  }
  for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sum, (part[0] + part[1]) + (part[2] + part[3]));
}

// value = target_pred * dd_wt (or one of them alone) and its first-occurrence argmax (:401-413); two stages
struct ArgMax { double v; int i; };
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__global__ __launch_bounds__(256) void goal_argmax_kernel(const float* __restrict__ target_pred, const double* __restrict__ wt_new,
                                                          const double* __restrict__ wt_last, const double* __restrict__ sum, int have_last,
                                                          int mode, int n, double* __restrict__ value_out, ArgMax* __restrict__ partial) {
  // mode 0: target_pred * wt, 1: target_pred alone (temperature -1), 2: wt alone (temperature 0)
  const bool keep_last = mode != 2 && have_last && *sum < 10.0;      // (:398) "stuck inside obstacle, use last dd_wt"
  const double* wt = keep_last ? wt_last : wt_new;
  ArgMax best{-INFINITY, 0x7fffffff};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double tp = target_pred ? (double)target_pred[i] : 1.0;
    const double v = mode == 1 ? tp : (mode == 2 ? wt[i] : tp * wt[i]);
    if (value_out) value_out[i] = v;
    if (v > best.v) { best.v = v; best.i = i; }       // ascending i per lane: ties keep the first
  }
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax other{__shfl_xor(best.v, o), __shfl_xor(best.i, o)};
    best = better(best, other);
  }
  __shared__ ArgMax part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = better(better(part[0], part[1]), better(part[2], part[3]));
}
__global__ __launch_bounds__(64) void goal_argmax_final_kernel(const ArgMax* __restrict__ partial, int nparts, const double* __restrict__ sum,
                                                               int have_last, int mode, const double* __restrict__ wt_new,
                                                               double* __restrict__ wt_last, int n, int* __restrict__ out_idx,
                                                               double* __restrict__ out_val) {
  ArgMax best{-INFINITY, 0x7fffffff};
  for (int i = threadIdx.x; i < nparts; i += 64) best = better(best, partial[i]);
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax other{__shfl_xor(best.v, o), __shfl_xor(best.i, o)};
    best = better(best, other);
  }
  if (threadIdx.x == 0) {
    out_idx[0] = best.i;
    out_idx[1] = (mode != 2 && have_last && *sum < 10.0) ? 1 : 0;      // 1: the last weights were kept
    out_val[0] = best.v;
    out_val[1] = *sum;
  }
}

}  // namespace
}  // namespace peanut

using namespace peanut;

struct peanut_goal {
  peanut::Options opts = peanut::default_options();   // tuning options of this handle (options.h): snapshot at creation
  int H = 0, W = 0, rad = 0, tiles_x = 0, tiles_y = 0;
  DevBuf trav, state, dist, order, active, counters, maxbits, wt_new, wt_last, sum, partial, out_idx, out_val, value;
  bool have_last = false;
  int last_lw = 0, last_lh = 0;
  int last_rounds = 0, last_passes = 0;
  bool last_converged = true;      // false: the ordering passes hit MAX_ORDER_PASSES with the last one still changing tiles
  int round_hint[1 + 64] = {0};    // rounds each stage needed in the previous solve (run_stage); stage 0 + up to 64 ordering passes (fmm_max_passes is clamped to 64)
  // peanut_goal_select_begin: the field of the next select is solved on `side`, behind `ev_inputs` only, so that it runs next to
  // whatever the caller enqueued on its stream after the call (the map-prediction forward that produces target_pred)
  hipStream_t side = nullptr;
  hipEvent_t ev_inputs = nullptr, ev_field = nullptr;
  unsigned int* host_counters = nullptr;   // pinned: the per-round counters of a batch land here without blocking the enqueuing thread
  // peanut_goal_select_begin: traversible map, initialisation and the first batch of stage-A rounds are already on `side`
  struct Begun {
    bool on = false;
    const float* obst = nullptr;
    const unsigned char *col = nullptr, *vis = nullptr;
    int lmb[4] = {0, 0, 0, 0}, loc_r = 0, loc_c = 0, batch = 0, cur = 0;
  } begun;
  ~peanut_goal() {
    if (ev_inputs) (void)hipEventDestroy(ev_inputs);
    if (ev_field) (void)hipEventDestroy(ev_field);
    if (side) (void)hipStreamDestroy(side);
    if (host_counters) (void)hipHostFree(host_counters);
  }
};

namespace {

// `batch` relaxation rounds of a stage and the read-back of their "tiles changed" counters, enqueued (no synchronisation)
template <bool SECOND>
int enqueue_rounds(peanut_goal* g, int* cur, int batch, hipStream_t s) {
  const int H = g->H, W = g->W, nt = g->tiles_x * g->tiles_y;
  unsigned char* act = (unsigned char*)g->active.p;
  unsigned int* counters = (unsigned int*)g->counters.p;
  if (!g->host_counters) PEANUT_HIP_CHECK(hipHostMalloc((void**)&g->host_counters, MAX_ROUNDS_PER_CHECK * sizeof(unsigned int), hipHostMallocDefault));
  PEANUT_HIP_CHECK(hipMemsetAsync(counters, 0, batch * sizeof(unsigned int), s));
  for (int k = 0; k < batch; ++k) {
    // three flag arrays in rotation: read, written (clean since the round before last), wiped for the next round
    unsigned char* in = act + (size_t)(*cur) * nt;
    unsigned char* out = act + (size_t)((*cur + 1) % 3) * nt;
    unsigned char* clr = act + (size_t)((*cur + 2) % 3) * nt;
    const bool local32 = opt(OPT_FMM_LOCAL32) != 0, blocked = opt(OPT_FMM_BLOCKED) != 0;
#define PEANUT_FMM_LAUNCH(KERNEL, ...)                                                                                                 \
  hipLaunchKernelGGL(KERNEL, dim3(nt), dim3(1024), 0, s, (double*)g->dist.p, (const double*)g->order.p, (const unsigned char*)g->state.p, \
                     H, W, g->tiles_x, g->tiles_y, in, out, clr, counters + k, ##__VA_ARGS__)
    if (blocked) {
      const int max_inner = (int)std::min<long long>(std::max<long long>(opt(OPT_FMM_INNER), 1), 1024);
      if (local32) PEANUT_FMM_LAUNCH((fmm_round_blocked_kernel<SECOND, true>), max_inner);
      else PEANUT_FMM_LAUNCH((fmm_round_blocked_kernel<SECOND, false>), max_inner);
    } else {
      if (local32) PEANUT_FMM_LAUNCH((fmm_round_kernel<SECOND, true>));
      else PEANUT_FMM_LAUNCH((fmm_round_kernel<SECOND, false>));
    }
#undef PEANUT_FMM_LAUNCH
    *cur = (*cur + 1) % 3;
  }
  PEANUT_HIP_CHECK(hipMemcpyAsync(g->host_counters, counters, batch * sizeof(unsigned int), hipMemcpyDeviceToHost, s));
  return 0;
}

// The host reads the per-round "tiles changed" counters once per batch of rounds.  The first batch of a stage is as long as that
// stage needed in the previous solve on this handle (+2: consecutive solves of an episode see almost the same map), so that a
// stage normally costs ONE host synchronisation; rounds after the field has settled wake no tile and cost a few microseconds each.
inline int first_batch(const peanut_goal* g, int stage) { return std::min(std::max(g->round_hint[stage] + 2, 4), MAX_ROUNDS_PER_CHECK); }

// rounds of one stage until a round changes nothing; *total = tiles changed over the whole stage.  enqueued > 0: the first batch
// (that many rounds) is already on the stream (peanut_goal_select_begin)
template <bool SECOND>
int run_stage(peanut_goal* g, int stage, int* cur, int* rounds_used, unsigned long long* total, hipStream_t s, int enqueued = 0) {
  const int max_rounds = 32 * (g->tiles_x + g->tiles_y) + 64;    // generous bound on the front's path, in tiles
  *total = 0;
  int batch = enqueued > 0 ? enqueued : first_batch(g, stage);
  int needed = 0;
  for (int round = 0; round < max_rounds; round += batch, batch = ROUNDS_PER_CHECK) {
    if (!(round == 0 && enqueued > 0))
      if (int rc = enqueue_rounds<SECOND>(g, cur, batch, s)) return rc;
    PEANUT_HIP_CHECK(hipStreamSynchronize(s));
    const unsigned int* host = g->host_counters;
    *rounds_used += batch;
    for (int k = 0; k < batch; ++k) {
      *total += host[k];
      if (host[k]) needed = round + k + 1;
    }
    if (host[batch - 1] == 0) {
      g->round_hint[stage] = needed;
      return 0;
    }
  }
  return fail(PEANUT_EHIP, "fmm: a relaxation stage did not settle within its round budget");
}

// state, distances and the seeds' tiles; leaves the seed tiles active in flag array 0
int field_preamble(peanut_goal* g, const unsigned char* trav, const unsigned char* seed_mask, int seed_r, int seed_c, hipStream_t s) {
  const int H = g->H, W = g->W, n = H * W, nt = g->tiles_x * g->tiles_y;
  unsigned char* act = (unsigned char*)g->active.p;     // three flag arrays in rotation (run_stage)
  unsigned char* seed_tiles = act + 3 * (size_t)nt;     // fourth array: the tiles that hold seeds
  PEANUT_HIP_CHECK(hipMemsetAsync(act, 0, 4 * (size_t)nt, s));
  hipLaunchKernelGGL(fmm_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, trav, seed_mask, seed_r, seed_c, H, W,
                     (unsigned char*)g->state.p, (double*)g->dist.p, seed_tiles, g->tiles_x);
  PEANUT_HIP_CHECK(hipMemcpyAsync(act, seed_tiles, nt, hipMemcpyDeviceToDevice, s));
  return 0;
}

// begun_batch > 0: the preamble and that many stage-A rounds are already on `s` (begun_cur = the flag rotation after them)
int solve_field(peanut_goal* g, const unsigned char* trav, const unsigned char* seed_mask, int seed_r, int seed_c, hipStream_t s,
                int begun_batch = 0, int begun_cur = 0) {
  const int H = g->H, W = g->W, n = H * W, nt = g->tiles_x * g->tiles_y;
  unsigned char* act = (unsigned char*)g->active.p;     // three flag arrays in rotation (run_stage)
  unsigned char* seed_tiles = act + 3 * (size_t)nt;     // fourth array: the tiles that hold seeds
  int cur = 0, rounds = 0;
  unsigned long long changed = 0;
  // stage A: first-order field
  if (begun_batch > 0) {
    cur = begun_cur;
  } else if (int rc = field_preamble(g, trav, seed_mask, seed_r, seed_c, s)) {
    return rc;
  }
  if (int rc = run_stage<false>(g, 0, &cur, &rounds, &changed, s, begun_batch)) return rc;
  // stage B: second order on the graph ordered by the previous field, until a pass changes nothing
  g->last_passes = 0;
  g->last_converged = false;
  const int max_passes = (int)std::min<long long>(std::max<long long>(opt(OPT_FMM_MAX_PASSES), 2), 64);
  for (int pass = 0; pass < max_passes; ++pass) {
    PEANUT_HIP_CHECK(hipMemcpyAsync(g->order.p, g->dist.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (pass == 0) {      // from the seeds again
      hipLaunchKernelGGL(fmm_restart_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned char*)g->state.p, n, (double*)g->dist.p);
      PEANUT_HIP_CHECK(hipMemsetAsync(act, 0, 3 * (size_t)nt, s));
      cur = 0;
      PEANUT_HIP_CHECK(hipMemcpyAsync(act, seed_tiles, nt, hipMemcpyDeviceToDevice, s));
    } else {              // warm start: every tile re-examines its cells under the new ordering
      PEANUT_HIP_CHECK(hipMemsetAsync(act + (size_t)cur * nt, 1, nt, s));
    }
    if (int rc = run_stage<true>(g, 1 + pass, &cur, &rounds, &changed, s)) return rc;
    g->last_passes = pass + 1;
    if (pass > 0 && changed == 0) { g->last_converged = true; break; }
  }
  g->last_rounds = rounds;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PEANUT_EHIP, std::string("fmm: ") + hipGetErrorString(e));
  return 0;
}

}  // namespace

extern "C" {

int peanut_goal_create(peanut_goal_t** out, int full_h, int full_w, int col_rad) {
  if (!out || full_h < 1 || full_w < 1 || col_rad < 0) return fail(PEANUT_EINVAL, "peanut_goal_create: bad arguments");
  auto g = std::make_unique<peanut_goal>();
  g->H = full_h; g->W = full_w; g->rad = col_rad;
  g->tiles_x = (full_w + TILE - 1) / TILE;
  g->tiles_y = (full_h + TILE - 1) / TILE;
  const size_t n = (size_t)full_h * full_w;
  int rc;
  if ((rc = g->trav.ensure(n)) || (rc = g->state.ensure(n)) || (rc = g->dist.ensure(n * sizeof(double))) ||
      (rc = g->order.ensure(n * sizeof(double))) || (rc = g->active.ensure(4 * (size_t)g->tiles_x * g->tiles_y)) || (rc = g->counters.ensure(MAX_ROUNDS_PER_CHECK * sizeof(unsigned int))) ||
      (rc = g->maxbits.ensure(sizeof(unsigned long long))) || (rc = g->wt_new.ensure(n * sizeof(double))) ||
      (rc = g->wt_last.ensure(n * sizeof(double))) || (rc = g->value.ensure(n * sizeof(double))) || (rc = g->sum.ensure(sizeof(double))) ||
      (rc = g->partial.ensure(1024 * sizeof(ArgMax))) || (rc = g->out_idx.ensure(2 * sizeof(int))) || (rc = g->out_val.ensure(2 * sizeof(double))))
    return rc;
  *out = g.release();
  return 0;
}

void peanut_goal_destroy(peanut_goal_t* g) { delete g; }

int peanut_goal_reset(peanut_goal_t* g) {
  if (!g) return fail(PEANUT_EINVAL, "peanut_goal_reset: null handle");
  g->have_last = false;
  if (g->begun.on) { (void)hipStreamSynchronize(g->side); g->begun.on = false; }
  return 0;
}

namespace {
int ensure_side(peanut_goal* g) {
  if (g->side) return 0;
  // highest priority: the field is a chain of short launches (a ring of tiles per round) whose latency is the cost; its
  // workgroups should take the next free CU slots ahead of the forward's large grids
  int prio_least = 0, prio_greatest = 0;
  PEANUT_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  PEANUT_HIP_CHECK(hipStreamCreateWithPriority(&g->side, hipStreamNonBlocking, prio_greatest));
  PEANUT_HIP_CHECK(hipEventCreateWithFlags(&g->ev_inputs, hipEventDisableTiming));
  PEANUT_HIP_CHECK(hipEventCreateWithFlags(&g->ev_field, hipEventDisableTiming));
  return 0;
}
// np.clip(loc + lmb, 0, full - 1)  (:389-390)
void seed_cell(const peanut_goal* g, const int lmb[4], int loc_r, int loc_c, int* sr, int* sc) {
  int r = loc_r + lmb[0], c = loc_c + lmb[2];
  *sr = r < 0 ? 0 : (r > g->H - 1 ? g->H - 1 : r);
  *sc = c < 0 ? 0 : (c > g->W - 1 ? g->W - 1 : c);
}
}  // namespace

int peanut_goal_select_begin(peanut_goal_t* g, const float* full_obstacle, const uint8_t* collision_map, const uint8_t* visited_vis,
                             const int lmb[4], int loc_r, int loc_c, void* stream) {
  peanut::OptionScope option_scope(g ? &g->opts : nullptr);
  if (!g || !full_obstacle || !lmb) return fail(PEANUT_EINVAL, "peanut_goal_select_begin: null argument");
  if (lmb[0] < 0 || lmb[2] < 0 || lmb[1] > g->H || lmb[3] > g->W || lmb[1] - lmb[0] < 1 || lmb[3] - lmb[2] < 1)
    return fail(PEANUT_EINVAL, "peanut_goal_select_begin: bad local map boundaries");
  if (int rc = ensure_side(g)) return rc;
  if (g->begun.on) PEANUT_HIP_CHECK(hipStreamSynchronize(g->side));      // a begin nobody finished: its rounds still own the scratch
  PEANUT_HIP_CHECK(hipEventRecord(g->ev_inputs, (hipStream_t)stream));
  PEANUT_HIP_CHECK(hipStreamWaitEvent(g->side, g->ev_inputs, 0));
  // from here on work may be in flight on g->side that writes the handle's scratch: the handle is marked busy BEFORE the first
  // enqueue (for no inputs yet: obst = nullptr matches no select, so whoever comes next synchronises the side stream first), and an
  // enqueue that fails part-way lets the side stream run out before the error is returned
  g->begun.on = true;
  g->begun.obst = nullptr; g->begun.col = nullptr; g->begun.vis = nullptr;
  int cur = 0, batch = 0, rc = 0;
  int sr, sc;
  seed_cell(g, lmb, loc_r, loc_c, &sr, &sc);
  rc = peanut_goal_traversible(g, full_obstacle, collision_map, visited_vis, nullptr, g->side);
  if (!rc) rc = field_preamble(g, (const unsigned char*)g->trav.p, nullptr, sr, sc, g->side);
  if (!rc) { batch = first_batch(g, 0); rc = enqueue_rounds<false>(g, &cur, batch, g->side); }
  if (!rc && hipGetLastError() != hipSuccess) rc = fail(PEANUT_EHIP, "peanut_goal_select_begin: launch failed");
  if (rc) {
    (void)hipStreamSynchronize(g->side);
    g->begun.on = false;
    return rc;
  }
  g->begun.obst = full_obstacle; g->begun.col = collision_map; g->begun.vis = visited_vis;
  for (int i = 0; i < 4; ++i) g->begun.lmb[i] = lmb[i];
  g->begun.loc_r = loc_r; g->begun.loc_c = loc_c; g->begun.batch = batch; g->begun.cur = cur;
  return 0;
}

int peanut_goal_rounds(peanut_goal_t* g) { return g ? g->last_rounds : PEANUT_EINVAL; }
int peanut_goal_passes(peanut_goal_t* g) { return g ? g->last_passes : PEANUT_EINVAL; }
int peanut_goal_converged(peanut_goal_t* g) { return g ? (g->last_converged ? 1 : 0) : PEANUT_EINVAL; }

int peanut_fmm_distance(peanut_goal_t* g, const uint8_t* traversible, const uint8_t* goal_mask, int goal_r, int goal_c, int fill_mode,
                        double* dist_out, void* stream) {
  peanut::OptionScope option_scope(g ? &g->opts : nullptr);
  if (!g || !traversible || !dist_out) return fail(PEANUT_EINVAL, "peanut_fmm_distance: null argument");
  if (!goal_mask && (goal_r < 0 || goal_r >= g->H || goal_c < 0 || goal_c >= g->W))
    return fail(PEANUT_EINVAL, "peanut_fmm_distance: goal cell outside the map");
  hipStream_t s = (hipStream_t)stream;
  if (g->begun.on) { PEANUT_HIP_CHECK(hipStreamSynchronize(g->side)); g->begun.on = false; }      // (an unfinished select_begin owns the scratch)
  if (int rc = solve_field(g, traversible, goal_mask, goal_mask && goal_r < 0 ? -1 : goal_r, goal_c, s)) return rc;
  const int n = g->H * g->W;
  if (fill_mode == 0) {
    PEANUT_HIP_CHECK(hipMemcpyAsync(dist_out, g->dist.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
  } else {
    PEANUT_HIP_CHECK(hipMemsetAsync(g->maxbits.p, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(fmm_max_kernel, dim3(256), dim3(256), 0, s, (const double*)g->dist.p, n, (unsigned long long*)g->maxbits.p);
    hipLaunchKernelGGL(fmm_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const double*)g->dist.p, n,
                       (const unsigned long long*)g->maxbits.p, dist_out);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("peanut_fmm_distance: ") + hipGetErrorString(e));
}

int peanut_goal_traversible(peanut_goal_t* g, const float* full_obstacle, const uint8_t* collision_map, const uint8_t* visited_vis,
                            uint8_t* trav_out, void* stream) {
  if (!g || !full_obstacle) return fail(PEANUT_EINVAL, "peanut_goal_traversible: null argument");
  if (g->rad <= TRAV_MAX_RAD)
    hipLaunchKernelGGL(goal_trav_kernel, dim3((g->W + 31) / 32, (g->H + 7) / 8), dim3(256), 0, (hipStream_t)stream, full_obstacle,
                       collision_map, visited_vis, g->H, g->W, g->rad, trav_out ? trav_out : (unsigned char*)g->trav.p);
  else
    hipLaunchKernelGGL(goal_trav_wide_kernel, dim3((g->W + 31) / 32, (g->H + 7) / 8), dim3(256), 0, (hipStream_t)stream, full_obstacle,
                       collision_map, visited_vis, g->H, g->W, g->rad, trav_out ? trav_out : (unsigned char*)g->trav.p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("peanut_goal_traversible: ") + hipGetErrorString(e));
}

int peanut_goal_select(peanut_goal_t* g, const float* full_obstacle, const uint8_t* collision_map, const uint8_t* visited_vis,
                       const int lmb[4], int loc_r, int loc_c, const float* target_pred, double dist_weight_temperature,
                       int map_resolution, int goal_rc_out[2], double stats_out[4], double* dist_out, double* value_out, void* stream) {
  peanut::OptionScope option_scope(g ? &g->opts : nullptr);
  if (!g || !full_obstacle || !lmb || !goal_rc_out) return fail(PEANUT_EINVAL, "peanut_goal_select: null argument");
  const int gx1 = lmb[0], gx2 = lmb[1], gy1 = lmb[2], gy2 = lmb[3];
  const int lw = gx2 - gx1, lh = gy2 - gy1;
  if (gx1 < 0 || gy1 < 0 || gx2 > g->H || gy2 > g->W || lw < 1 || lh < 1) return fail(PEANUT_EINVAL, "peanut_goal_select: bad local map boundaries");
  const int mode = dist_weight_temperature == -1 ? 1 : (dist_weight_temperature == 0 ? 2 : 0);
  if (mode != 2 && !target_pred) return fail(PEANUT_EINVAL, "peanut_goal_select: target_pred is needed unless dist_weight_temperature == 0");
  hipStream_t s = (hipStream_t)stream;
  int sr, sc;
  seed_cell(g, lmb, loc_r, loc_c, &sr, &sc);
  // the field needs the map inputs only: after peanut_goal_select_begin with these very inputs its traversible map, its
  // initialisation and the first batch of stage-A rounds are already running on the handle's own stream, next to what the caller
  // enqueued on `s` since (the forward that produces target_pred); the rest follows there and `s` joins before the weights
  bool aside = false;
  if (g->begun.on) {
    const peanut_goal::Begun& b = g->begun;
    aside = b.obst == full_obstacle && b.col == collision_map && b.vis == visited_vis && b.lmb[0] == gx1 && b.lmb[1] == gx2 &&
            b.lmb[2] == gy1 && b.lmb[3] == gy2 && b.loc_r == loc_r && b.loc_c == loc_c;
    g->begun.on = false;
    if (!aside) PEANUT_HIP_CHECK(hipStreamSynchronize(g->side));      // begun for other inputs: let it run out, solve these
  }
  if (aside) {
    if (int rc = solve_field(g, (const unsigned char*)g->trav.p, nullptr, sr, sc, g->side, g->begun.batch, g->begun.cur)) {
      (void)hipStreamSynchronize(g->side);      // `s` never joined the side stream: let it run out before the scratch is reused
      return rc;
    }
    PEANUT_HIP_CHECK(hipEventRecord(g->ev_field, g->side));
    PEANUT_HIP_CHECK(hipStreamWaitEvent(s, g->ev_field, 0));
  } else {
    if (int rc = peanut_goal_traversible(g, full_obstacle, collision_map, visited_vis, nullptr, stream)) return rc;
    if (int rc = solve_field(g, (const unsigned char*)g->trav.p, nullptr, sr, sc, s)) return rc;
  }
  if (g->have_last && (g->last_lw != lw || g->last_lh != lh)) g->have_last = false;
  const int n = lw * lh;
  const double temperature = dist_weight_temperature / (double)map_resolution;      // (:395)
  PEANUT_HIP_CHECK(hipMemsetAsync(g->sum.p, 0, sizeof(double), s));
  hipLaunchKernelGGL(goal_weight_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const double*)g->dist.p, g->W, gx1, gy1, lw, lh,
                     temperature, mode == 2 ? 1 : 0, (double*)g->wt_new.p, (double*)g->sum.p);
  const int nparts = std::min(1024, (n + 255) / 256);
  hipLaunchKernelGGL(goal_argmax_kernel, dim3(nparts), dim3(256), 0, s, target_pred, (const double*)g->wt_new.p, (const double*)g->wt_last.p,
                     (const double*)g->sum.p, g->have_last ? 1 : 0, mode, n, value_out ? value_out : (double*)g->value.p, (ArgMax*)g->partial.p);
  hipLaunchKernelGGL(goal_argmax_final_kernel, dim3(1), dim3(64), 0, s, (const ArgMax*)g->partial.p, nparts, (const double*)g->sum.p,
                     g->have_last ? 1 : 0, mode, (const double*)g->wt_new.p, (double*)g->wt_last.p, n, (int*)g->out_idx.p, (double*)g->out_val.p);
  int idx[2];
  double val[2];
  PEANUT_HIP_CHECK(hipMemcpyAsync(idx, g->out_idx.p, sizeof(idx), hipMemcpyDeviceToHost, s));
  PEANUT_HIP_CHECK(hipMemcpyAsync(val, g->out_val.p, sizeof(val), hipMemcpyDeviceToHost, s));
  if (dist_out) PEANUT_HIP_CHECK(hipMemcpyAsync(dist_out, g->dist.p, (size_t)g->H * g->W * sizeof(double), hipMemcpyDeviceToDevice, s));
  PEANUT_HIP_CHECK(hipStreamSynchronize(s));
  if (!idx[1] && mode != 2) {    // self.dd_wt = dd_wt (:410): the fresh weights become the last ones unless the old ones were kept
    PEANUT_HIP_CHECK(hipMemcpyAsync(g->wt_last.p, g->wt_new.p, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
    g->last_lw = lw; g->last_lh = lh;
  }
  g->have_last = true;
  goal_rc_out[0] = idx[0] / lh;
  goal_rc_out[1] = idx[0] - goal_rc_out[0] * lh;
  if (stats_out) { stats_out[0] = val[0]; stats_out[1] = val[1]; stats_out[2] = idx[1]; stats_out[3] = g->last_rounds; }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("peanut_goal_select: ") + hipGetErrorString(e));
}

}  // extern "C"
