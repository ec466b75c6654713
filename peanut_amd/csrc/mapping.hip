// Stage 2 of the PEANUT hot path: egocentric -> allocentric semantic-map projection, the HIP
// counterpart of Semantic_Mapping.forward (nav/agent/mapping.py:52-179) with its helpers
// (nav/agent/utils/depth_utils.py:129-252, nav/agent/utils/model.py:7-43).
//
// The reference is ~60 tiny tensor ops plus 8 rounds over a 35 MB voxel grid.  Here it is a short
// chain of small kernels on one stream, with no host synchronisation and no dense voxel grid:
//
//   map_points   depth pixel -> normalised (x,y,z), SAME fp32 operation order as the reference
//                (+ the five statistics that decide the low-stairs branch, mapping.py:90-97, exactly)
//   map_keys     mask, splat position, base-cell key of every point
//   map_alloc / map_fill / map_place   group the points by base cell WITHOUT a sort: every cell gets a contiguous
//                segment (its size counted by map_keys, its start taken from one atomic cursor by the cell's
//                first point), the points drop into their cell's segment in arrival order, and every point then
//                finds its final slot as segment start + (number of points of the segment with a smaller index)
//                -- the order inside a cell is point order by construction, which is all the replay needs; where a
//                cell's segment lies is irrelevant.  map_place also writes, per placed point, the 1-D splat
//                weights + feature values (contiguous); the segment starts are the dense cell lookup table
//   map_voxels   one thread per (touched voxel, feature): replays the reference's 8 corner passes for
//                that voxel only -- contributions added IN POINT ORDER, rintf after every pass -- and
//                adds the (integer-valued, hence order-free) result into the two height projections
//   map_finish   thresholds/clamps -> the 100x100 egocentric window (+ fp_map_pred), clears scratch; pose integration
//                + the two affine_grid theta rows; re-arms the per-cell tables (seven launches per step in all)
//   map_warp     rotation resample -> translation resample -> max with the previous map, fused
//
// Why the voxel trick is exact: splat_feat_nd rounds the WHOLE grid after each corner pass
// (depth_utils.py:249-250), so a voxel's final value depends only on the contributions it receives
// per pass, in order.  In pass c a voxel v receives exactly the points whose floor cell is v - c, and
// scatter_add_ on the CPU adds them in point order; grouping points by floor cell with a STABLE sort
// keeps that order, so each voxel can be evaluated independently and bit-identically.  Unsafe corners
// have weight 0 and only add +0.0.  The projections sum integer-valued floats (< 2^24), so atomics do
// not introduce order dependence.
//
// fp32 contraction is OFF in this file: where the reference's CPU kernels use fused multiply-adds
// (affine_grid's bmm, grid_sample's blend, lerp) fmaf is written explicitly.
#include <algorithm>
#include <memory>
#include <vector>

#include "../../include/peanut_hip.h"
#include "common.h"
#include "net_common.h"

#pragma clang fp contract(off)

namespace peanut {
namespace {

constexpr unsigned INVALID_KEY = 1u << 20;   // > 100*100*80; keys are sorted on 21 bits

struct MapP {
  int h, w, N, ncat, F, C;       // point grid (frame / du_scale), points, semantic channels, feature rows (1+ncat), map channels (4+ncat)
  int du, fh, fw;                // du_scale and the full frame size (mapping.py:23,49,60,80)
  int vr, zb, M;                 // vision range (cells), z bins, local map size (cells)
  float xc, zc, f;               // camera matrix (depth_utils.py:27-34)
  float agent_h, shift_x;        // 88 cm, 250 cm
  float res, vr_half, vr_f, z_mid, z_span;
  int min_z, max_z;
  float thr_map, thr_exp, thr_cat;
  unsigned allh_mask;            // feature rows that use the all-height projection (mapping.py:107-113)
  int x1, y1;                    // paste window origin (mapping.py:130-133)
  float half_cells;              // map_size_cm // (resolution*2) = 240
  int toilet_ch;                 // obs channel of feat[0, 1+4]
};

// order-preserving float <-> uint maps (for atomicMax / atomicMin on floats)
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

// Statistics of the low-stairs heuristic (mapping.py:90-97), accumulated by map_points:
//   [0] n      = #points with -1 < z < 1                      (len(my_zs))
//   [1] mid    = #those with 0.2 < my_z < 0.7
//   [2] le     = #those with my_z <= 0.2
//   [3] max_le = max{my_z : my_z <= 0.2}  (ordered-uint)       [4] min_gt = min{my_z : my_z > 0.2}
// torch.quantile(my_zs, 0.03) > 0.2 only needs the two order statistics around rank 0.03*(n-1):
// both lie above 0.2 iff le <= k_lo, both at or below iff le >= k_hi + 1, and otherwise they are
// exactly max_le and min_gt -- so no sort/selection is needed to reproduce the branch bit for bit.
struct StairStats { unsigned n, mid, le, max_le, min_gt; };

__device__ __forceinline__ bool stairs_branch(const StairStats& s) {
  if (s.n == 0) return false;
  const float ranks = 0.03f * (float)(s.n - 1);        // q (fp32 tensor) * last_index
  const int k_lo = (int)ranks, k_hi = (int)ceilf(ranks);
  bool q_gt;
  if ((int)s.le <= k_lo) q_gt = true;
  else if ((int)s.le >= k_hi + 1) q_gt = false;
  else {
    const float a = ord2f(s.max_le), b = ord2f(s.min_gt), wgt = ranks - (float)k_lo, diff = b - a;
    const float q = (fabsf(wgt) < 0.5f) ? fmaf(wgt, diff, a) : fmaf(wgt - 1.0f, diff, b);   // ATen lerp
    q_gt = q > 0.2f;
  }
  return q_gt && ((float)s.mid > (float)(0.2 * (double)s.n));
}

// ---- 1. depth -> normalised coordinates (mapping.py:59-88) + stairs statistics ----
// ---- 0. du_scale > 1 (mapping.py:60,80-82): depth sub-sampled [::s, ::s], semantic channels AvgPool2d(s) -> an observation
//         of (h / s) x (w / s) points in the layout the other kernels read (RGB channels are not used by the mapping) ----
__global__ __launch_bounds__(256) void map_decimate_kernel(const float* __restrict__ obs, float* __restrict__ dec, MapP P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int ch = 3 + blockIdx.y;                       // depth, then the semantic channels
  if (p >= P.N) return;
  const int r = p / P.w, c = p - r * P.w;
  const float* src = obs + (size_t)ch * P.fh * P.fw;
  float v;
  if (ch == 3) {
    v = src[(size_t)(r * P.du) * P.fw + c * P.du];
  } else {                                             // ATen's avg_pool2d: window summed row by row, then divided
    float sum = 0.0f;
    for (int i = 0; i < P.du; ++i)
      for (int j = 0; j < P.du; ++j) sum += src[(size_t)(r * P.du + i) * P.fw + c * P.du + j];
    v = sum / (float)(P.du * P.du);
  }
  dec[(size_t)ch * P.N + p] = v;
}

__global__ __launch_bounds__(256) void map_points_kernel(const float* __restrict__ obs, float* __restrict__ coords,
                                                         StairStats* __restrict__ stats, MapP P) {
  __shared__ unsigned s_n, s_mid, s_le, s_max, s_min;
  if (threadIdx.x == 0) { s_n = 0; s_mid = 0; s_le = 0; s_max = 0u; s_min = 0xffffffffu; }
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P.N) {
    const int r = p / P.w, c = p - r * P.w;
    const float d = obs[3 * P.N + p];
    const float gx = (float)(c * P.du), gz = (float)(P.fh - 1 - r * P.du);     // grid_x / grid_z [::scale, ::scale]
    float X = ((gx - P.xc) * d) / P.f;          // get_point_cloud_from_z_t
    float Z = ((gz - P.zc) * d) / P.f;
    float Y = d;
    Z = Z + P.agent_h;                          // transform_camera_view_t (R = I)
    X = X + P.shift_x;                          // transform_pose_t (R = I, shift (250, 0))
    Y = Y + 0.0f;
    const float xs = (((X / P.res) - P.vr_half) / P.vr_f) * 2.0f;
    const float ys = (((Y / P.res) - P.vr_half) / P.vr_f) * 2.0f;
    const float zs = (((Z / P.res) - P.z_mid) / P.z_span) * 2.0f;
    coords[p] = xs;
    coords[P.N + p] = ys;
    coords[2 * P.N + p] = zs;
    if (zs > -1.0f && zs < 1.0f) {
      const float m = zs * 2.0f + 1.6f;
      atomicAdd(&s_n, 1u);
      if (m > 0.2f && m < 0.7f) atomicAdd(&s_mid, 1u);
      if (m <= 0.2f) { atomicAdd(&s_le, 1u); atomicMax(&s_max, f2ord(m)); }
      else atomicMin(&s_min, f2ord(m));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_n) {
    atomicAdd(&stats->n, s_n);
    if (s_mid) atomicAdd(&stats->mid, s_mid);
    if (s_le) { atomicAdd(&stats->le, s_le); atomicMax(&stats->max_le, s_max); }
    if (s_min != 0xffffffffu) atomicMin(&stats->min_gt, s_min);
  }
}

// ---- 3. splat position + base-cell key (depth_utils.py:217-236) ----
__global__ __launch_bounds__(256) void map_keys_kernel(const float* __restrict__ obs, const float* __restrict__ coords,
                                                       const StairStats* __restrict__ stats, float* __restrict__ pos,
                                                       unsigned* __restrict__ keys, int* __restrict__ cell_cnt,
                                                       int* __restrict__ cell_first, MapP P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  float xs = coords[p], ys = coords[P.N + p], zs = coords[2 * P.N + p];
  if (stairs_branch(*stats)) {
    const bool below = (zs * 2.0f + 1.6f) < 0.7f;
    const bool no_toilet = obs[P.toilet_ch * P.N + p] == 0.0f;
    if (below && no_toilet) { xs = 99999.0f; ys = 99999.0f; zs = 99999.0f; }
  }
  const float px = (xs * (float)P.vr) / 2.0f + (float)P.vr / 2.0f;
  const float py = (ys * (float)P.vr) / 2.0f + (float)P.vr / 2.0f;
  const float pz = (zs * (float)P.zb) / 2.0f + (float)P.zb / 2.0f;
  pos[p] = px;
  pos[P.N + p] = py;
  pos[2 * P.N + p] = pz;
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  unsigned key = INVALID_KEY;
  // a point has at least one safe corner per dimension iff 0 <= floor <= dim-1
  if (fx >= 0.0f && fx <= (float)(P.vr - 1) && fy >= 0.0f && fy <= (float)(P.vr - 1) && fz >= 0.0f &&
      fz <= (float)(P.zb - 1))
    key = ((unsigned)fx * (unsigned)P.vr + (unsigned)fy) * (unsigned)P.zb + (unsigned)fz;
  keys[p] = key;
  if (key != INVALID_KEY) {
    atomicAdd(&cell_cnt[key], 1);        // points of the cell (order-free)
    atomicMin(&cell_first[key], p);      // its first point: the one that will claim the segment
  }
}

// ---- 3b. grouping by base cell, point order inside a cell (what a stable sort by cell would give) ----
// The voxel replay needs every cell's points contiguous and in point order (CPU scatter_add_ adds them in that order,
// and fp32 sums depend on it).  A sort would also order the CELLS, which nobody needs; so: count, claim, fill, rank.
__global__ __launch_bounds__(256) void map_alloc_kernel(const unsigned* __restrict__ keys, const int* __restrict__ cell_cnt,
                                                        const int* __restrict__ cell_first, int* __restrict__ cell_head,
                                                        int* __restrict__ cursor, int N) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const unsigned k = keys[p];
  if (k != INVALID_KEY && cell_first[k] == p) cell_head[k] = atomicAdd(cursor, cell_cnt[k]);
}

__global__ __launch_bounds__(256) void map_fill_kernel(const unsigned* __restrict__ keys, const int* __restrict__ cell_head,
                                                       int* __restrict__ cell_fill, unsigned* __restrict__ seg, int N) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const unsigned k = keys[p];
  if (k != INVALID_KEY) seg[cell_head[k] + atomicAdd(&cell_fill[k], 1)] = (unsigned)p;    // arrival order
}

__global__ __launch_bounds__(256) void map_place_kernel(const unsigned* __restrict__ keys, const int* __restrict__ cell_head,
                                                        const int* __restrict__ cell_cnt, const unsigned* __restrict__ seg,
                                                        const int* __restrict__ cursor, const float* __restrict__ obs,
                                                        const float* __restrict__ pos, unsigned* __restrict__ skey,
                                                        unsigned* __restrict__ sidx, float* __restrict__ wts6,
                                                        float* __restrict__ feat_s, MapP P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.N) return;
  if (p >= *cursor) skey[p] = INVALID_KEY;           // the tail behind the last segment holds no point
  const unsigned k = keys[p];
  if (k == INVALID_KEY) return;
  const int s0 = cell_head[k], m = cell_cnt[k];
  int rank = 0;
  for (int i = 0; i < m; ++i) rank += seg[s0 + i] < (unsigned)p;     // points of my cell that precede me
  const int j = s0 + rank;
  skey[j] = k;
  sidx[j] = (unsigned)p;
  const float g2 = (float)(k % (unsigned)P.zb);
  const float g1 = (float)((k / (unsigned)P.zb) % (unsigned)P.vr);
  const float g0 = (float)(k / ((unsigned)P.zb * (unsigned)P.vr));
  const float p0 = pos[p], p1 = pos[P.N + p], p2 = pos[2 * P.N + p];
  // wts_ix = 1 - |pos - (floor + ix)|  (depth_utils.py:229)
  wts6[0 * P.N + j] = 1.0f - fabsf(p0 - g0);
  wts6[1 * P.N + j] = 1.0f - fabsf(p0 - (g0 + 1.0f));
  wts6[2 * P.N + j] = 1.0f - fabsf(p1 - g1);
  wts6[3 * P.N + j] = 1.0f - fabsf(p1 - (g1 + 1.0f));
  wts6[4 * P.N + j] = 1.0f - fabsf(p2 - g2);
  wts6[5 * P.N + j] = 1.0f - fabsf(p2 - (g2 + 1.0f));
  for (int f = 1; f < P.F; ++f) feat_s[(f - 1) * P.N + j] = obs[(3 + f) * P.N + p];
}

// ---- 4. re-arm the per-cell tables for the next frame (only the touched entries) ----
__device__ __forceinline__ void map_rearm(int p, const unsigned* __restrict__ keys, int* __restrict__ cell_head,
                                          int* __restrict__ cell_cnt, int* __restrict__ cell_first, int* __restrict__ cell_fill,
                                          int* __restrict__ cursor, int N) {
  if (p == 0) *cursor = 0;
  if (p >= N) return;
  const unsigned k = keys[p];
  if (k == INVALID_KEY) return;
  cell_head[k] = -1; cell_cnt[k] = 0; cell_first[k] = 0x7fffffff; cell_fill[k] = 0;
}

// ---- 5. per-voxel replay of the 8 corner passes ----
__global__ __launch_bounds__(256) void map_voxels_kernel(const float* __restrict__ wts6, const float* __restrict__ feat_s,
                                                         const unsigned* __restrict__ skey,
                                                         const int* __restrict__ cell_head, const int* __restrict__ cell_cnt,
                                                         float* __restrict__ proj, MapP P) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)P.N * 8 * P.F;
  if (t >= total) return;
  const int f = (int)(t % P.F);
  const int c = (int)((t / P.F) & 7);
  const int i = (int)(t / (8 * P.F));
  const unsigned k = skey[i];
  if (k == INVALID_KEY || (i > 0 && skey[i - 1] == k)) return;   // not a group head
  const int g2 = (int)(k % (unsigned)P.zb);
  const int g1 = (int)((k / (unsigned)P.zb) % (unsigned)P.vr);
  const int g0 = (int)(k / ((unsigned)P.zb * (unsigned)P.vr));
  const int v0 = g0 + ((c >> 2) & 1), v1 = g1 + ((c >> 1) & 1), v2 = g2 + (c & 1);
  // safe_ix: 0 < pos_ix < dim (strict), depth_utils.py:225
  if (!(v0 > 0 && v0 < P.vr && v1 > 0 && v1 < P.vr && v2 > 0 && v2 < P.zb)) return;
  // canonical owner of voxel v = the smallest corner c' whose floor cell v - c' is occupied
  for (int cc = 0; cc < c; ++cc) {
    const int h0 = v0 - ((cc >> 2) & 1), h1 = v1 - ((cc >> 1) & 1), h2 = v2 - (cc & 1);
    if (h0 < 0 || h1 < 0 || h2 < 0) continue;   // (h <= dim-1 always holds since v < dim)
    if (cell_head[(h0 * P.vr + h1) * P.zb + h2] >= 0) return;
  }
  float val = 0.0f;
  const float* fs = feat_s + (size_t)(f > 0 ? f - 1 : 0) * P.N;
  for (int cc = 0; cc < 8; ++cc) {
    const int c0 = (cc >> 2) & 1, c1 = (cc >> 1) & 1, c2 = cc & 1;
    const int h0 = v0 - c0, h1 = v1 - c1, h2 = v2 - c2;
    if (h0 < 0 || h1 < 0 || h2 < 0) continue;
    const unsigned hk = (unsigned)((h0 * P.vr + h1) * P.zb + h2);
    const int hd = cell_head[hk];
    if (hd < 0) continue;
    const float* wa = wts6 + (size_t)c0 * P.N;
    const float* wb = wts6 + (size_t)(2 + c1) * P.N;
    const float* wc = wts6 + (size_t)(4 + c2) * P.N;
    // the cell's points are the cell_cnt[hk] entries from hd on: the loads of four of them go out together (the loop used to end on
    // `skey[j] != hk`, one dependent load per point: this kernel was 45 % of the step), the adds stay in point order
    const int end = hd + cell_cnt[hk];
    int j = hd;
    for (; j + 4 <= end; j += 4) {
      const float a0 = wa[j], a1 = wa[j + 1], a2 = wa[j + 2], a3 = wa[j + 3];
      const float b0 = wb[j], b1 = wb[j + 1], b2 = wb[j + 2], b3 = wb[j + 3];
      const float d0 = wc[j], d1 = wc[j + 1], d2 = wc[j + 2], d3 = wc[j + 3];
      const float f0 = (f == 0) ? 1.0f : fs[j], f1 = (f == 0) ? 1.0f : fs[j + 1], f2 = (f == 0) ? 1.0f : fs[j + 2], f3 = (f == 0) ? 1.0f : fs[j + 3];
      val = val + f0 * ((a0 * b0) * d0);              // ((1*w0)*w1)*w2; scatter_add_ in point order
      val = val + f1 * ((a1 * b1) * d1);
      val = val + f2 * ((a2 * b2) * d2);
      val = val + f3 * ((a3 * b3) * d3);
    }
    for (; j < end; ++j) {
      const float wts = (wa[j] * wb[j]) * wc[j];
      const float ft = (f == 0) ? 1.0f : fs[j];
      val = val + ft * wts;
    }
    val = rintf(val);                                 // torch.round of the whole grid after this pass
  }
  if (val != 0.0f) {
    // voxels.transpose(2,3): projections are indexed [f][dim1][dim0]
    const int o = (f * P.vr + v1) * P.vr + v0;
    atomicAdd(&proj[o], val);                                                   // all_height_proj
    if (v2 >= P.min_z && v2 < P.max_z) atomicAdd(&proj[P.F * P.vr * P.vr + o], val);   // agent_height_proj
  }
}

// ---- 6. thresholds -> egocentric window, fp_map_pred; clears the projections for the next frame ----
__device__ __forceinline__ void map_view(int t, float* __restrict__ proj, float* __restrict__ view,
                                         float* __restrict__ fp_map_pred, StairStats* __restrict__ stats, const MapP& P) {
  if (t == 0) { stats->n = 0; stats->mid = 0; stats->le = 0; stats->max_le = 0u; stats->min_gt = 0xffffffffu; }
  const int cells = P.vr * P.vr;
  if (t >= cells) return;
  float* all_h = proj;
  float* agent_h = proj + P.F * cells;
  const float a0 = agent_h[t], e0 = all_h[t];
  const float m = fminf(fmaxf(a0 / P.thr_map, 0.0f), 1.0f);
  const float e = fminf(fmaxf(e0 / P.thr_exp, 0.0f), 1.0f);
  view[t] = m;
  view[cells + t] = e;
  view[2 * cells + t] = 0.0f;
  view[3 * cells + t] = 0.0f;
  fp_map_pred[t] = m;
  for (int f = 1; f < P.F; ++f) {
    const float v = ((P.allh_mask >> f) & 1u) ? all_h[f * cells + t] : agent_h[f * cells + t];
    view[(3 + f) * cells + t] = fminf(fmaxf(v / P.thr_cat, 0.0f), 1.0f);
  }
  for (int f = 0; f < P.F; ++f) { all_h[f * cells + t] = 0.0f; agent_h[f * cells + t] = 0.0f; }
}

// ---- 7. pose integration (mapping.py:143-167) + affine thetas (model.py:19-38) ----
struct WarpT { float c, s, tx, ty; };

__device__ __forceinline__ void map_pose(const float* __restrict__ rel, float* __restrict__ pose, WarpT* __restrict__ wt,
                                         const MapP& P) {
  const float k = 57.29577951308232f;
  float p0 = pose[0], p1 = pose[1], p2 = pose[2];
  const float r0 = rel[0], r1 = rel[1], r2 = rel[2];
  // sin / cos of a float32 argument, evaluated in double and rounded once: the correctly rounded float32 value, which is
  // what ATen's CPU kernels (SLEEF, <= 1 ulp) return in all but rare cases; the device's sinf / cosf are 1-2 ulp off more
  // often, and the warp's grid_sample amplifies a last-bit difference of the rotation into ~3e-5 on map cells
  const float ang = p2 / k;
  const float sa = (float)sin((double)ang), ca = (float)cos((double)ang);
  p1 = p1 + (r0 * sa + r1 * ca);
  p0 = p0 + (r0 * ca - r1 * sa);
  p2 = p2 + r2 * k;
  p2 = fmodf(p2 - 180.0f, 360.0f) + 180.0f;
  p2 = fmodf(p2 + 180.0f, 360.0f) - 180.0f;
  pose[0] = p0; pose[1] = p1; pose[2] = p2;
  const float sx = (-(((p0 * 100.0f) / P.res) - P.half_cells)) / P.half_cells;
  const float sy = (-(((p1 * 100.0f) / P.res) - P.half_cells)) / P.half_cells;
  const float st2 = 90.0f - p2;
  const float tr = (st2 * 3.14159265358979323846f) / 180.0f;
  wt->c = (float)cos((double)tr);
  wt->s = (float)sin((double)tr);
  wt->tx = sx;
  wt->ty = sy;
}

// ---- 6+7+4 in one launch: egocentric view, pose integration, re-arming of the per-cell tables ----
__global__ __launch_bounds__(256) PEANUT_NO_PK_F32 void map_finish_kernel(float* __restrict__ proj, float* __restrict__ view,
                                                         float* __restrict__ fp_map_pred, StairStats* __restrict__ stats,
                                                         const float* __restrict__ rel, float* __restrict__ pose, WarpT* __restrict__ wt,
                                                         const unsigned* __restrict__ keys, int* __restrict__ cell_head,
                                                         int* __restrict__ cell_cnt, int* __restrict__ cell_first,
                                                         int* __restrict__ cell_fill, int* __restrict__ cursor, MapP P) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) map_pose(rel, pose, wt, P);
  map_view(t, proj, view, fp_map_pred, stats, P);
  map_rearm(t, keys, cell_head, cell_cnt, cell_first, cell_fill, cursor, P.N);
}

// F.affine_grid base coordinate (align_corners=False): linspace(-1,1,n)[i] * (n-1) / n with ATen's
// symmetric linspace (each half one fma) -- verified bit-exact against torch on the build host.
__device__ __forceinline__ float base_coord(int i, int n) {
  const float step = 2.0f / (float)(n - 1);
  const float l = (i < n / 2) ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(n - i - 1), 1.0f);
  return (l * (float)(n - 1)) / (float)n;
}

// agent_view[c] at integer pixel (y, x): the 100x100 window pasted at (y1, x1), zero elsewhere
__device__ __forceinline__ float view_at(const float* __restrict__ view, int c, int y, int x, const MapP& P) {
  const int wy = y - P.y1, wx = x - P.x1;
  if ((unsigned)wy >= (unsigned)P.vr || (unsigned)wx >= (unsigned)P.vr) return 0.0f;
  return view[(c * P.vr + wy) * P.vr + wx];
}

struct Tap { int x0, y0; float nw, ne, sw, se; bool any; };

// grid_sample(bilinear, zeros, align_corners=True) tap for normalised coordinate (gx, gy)
__device__ __forceinline__ Tap make_tap(float gx, float gy, int n) {
  const float sf = (float)(n - 1) / 2.0f;
  const float ix = (gx + 1.0f) * sf, iy = (gy + 1.0f) * sf;
  const float xw = floorf(ix), yn = floorf(iy);
  const float w = ix - xw, e = 1.0f - w, nn = iy - yn, s = 1.0f - nn;
  Tap t;
  t.nw = s * e; t.ne = s * w; t.sw = nn * e; t.se = nn * w;
  // clamp before the int conversion so far-out-of-range coordinates stay out of range
  t.x0 = (int)fminf(fmaxf(xw, -2.0f), (float)n + 1.0f);
  t.y0 = (int)fminf(fmaxf(yn, -2.0f), (float)n + 1.0f);
  t.any = true;
  return t;
}

// rotated[c][yy][xx] = grid_sample(agent_view, rot_grid)[c][yy][xx]; zero outside the image
__device__ __forceinline__ float rotated_at(const float* __restrict__ view, int c, int yy, int xx, const Tap& rt,
                                            bool in_img, const MapP& P) {
  if (!in_img) return 0.0f;
  const float nwv = view_at(view, c, rt.y0, rt.x0, P), nev = view_at(view, c, rt.y0, rt.x0 + 1, P);
  const float swv = view_at(view, c, rt.y0 + 1, rt.x0, P), sev = view_at(view, c, rt.y0 + 1, rt.x0 + 1, P);
  float acc = nwv * rt.nw;
  acc = fmaf(nev, rt.ne, acc);
  acc = fmaf(swv, rt.sw, acc);
  acc = fmaf(sev, rt.se, acc);
  return acc;
}
// the same value with the four window reads unconditional (clamped address, the result replaced by 0 outside the window): the
// sixteen reads of an output pixel and channel go out together instead of one wait per branch
struct TapAddr { int o[4]; bool ok[4]; };
__device__ __forceinline__ TapAddr tap_addr(const Tap& rt, bool in_img, const MapP& P) {
  TapAddr a;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int wy = rt.y0 + (k >> 1) - P.y1, wx = rt.x0 + (k & 1) - P.x1;
    a.ok[k] = in_img && (unsigned)wy < (unsigned)P.vr && (unsigned)wx < (unsigned)P.vr;
    a.o[k] = a.ok[k] ? wy * P.vr + wx : 0;
  }
  return a;
}

__global__ __launch_bounds__(256) void map_warp_kernel(const float* __restrict__ view, const float* __restrict__ maps_last,
                                                       float* __restrict__ map_pred, const WarpT* __restrict__ wtp,
                                                       MapP P) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int M = P.M;
  if (t >= M * M) return;
  const int y = t / M, x = t - y * M;
  const WarpT wt = *wtp;
  // translation grid (theta2 = [[1,-0,tx],[0,1,ty]]): bmm as an fma chain over k = 0,1,2
  const float bx = base_coord(x, M), by = base_coord(y, M);
  float gxt = bx * 1.0f; gxt = fmaf(by, -0.0f, gxt); gxt = fmaf(1.0f, wt.tx, gxt);
  float gyt = bx * 0.0f; gyt = fmaf(by, 1.0f, gyt); gyt = fmaf(1.0f, wt.ty, gyt);
  const Tap tt = make_tap(gxt, gyt, M);
  // the four `rotated` pixels this output blends, each with its own rotation tap
  Tap rt[4] = {};
  bool in_img[4], touch = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int yy = tt.y0 + (q >> 1), xx = tt.x0 + (q & 1);
    in_img[q] = (unsigned)yy < (unsigned)M && (unsigned)xx < (unsigned)M;
    if (in_img[q]) {
      const float rbx = base_coord(xx, M), rby = base_coord(yy, M);
      float gxr = rbx * wt.c; gxr = fmaf(rby, -wt.s, gxr); gxr = fmaf(1.0f, 0.0f, gxr);
      float gyr = rbx * wt.s; gyr = fmaf(rby, wt.c, gyr); gyr = fmaf(1.0f, 0.0f, gyr);
      rt[q] = make_tap(gxr, gyr, M);
      // does this rotation tap reach the pasted window at all?
      const bool hit = rt[q].x0 + 1 >= P.x1 && rt[q].x0 < P.x1 + P.vr && rt[q].y0 + 1 >= P.y1 && rt[q].y0 < P.y1 + P.vr;
      in_img[q] = hit;      // a miss contributes exactly 0 (all four view taps are outside the window)
      touch |= hit;
    }
  }
  const size_t plane = (size_t)M * M;
  if (!__any(touch)) {
    // (most of the map: outside the warped window the output is max(maps_last, 0).)  All channels' loads first, then the stores:
    // the loop `load, wait, store` per channel cost one memory latency per channel, 14 in a row -- most of this kernel's 22 us
    for (int c0 = 0; c0 < P.C; c0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (c0 + u < P.C) ? maps_last[(size_t)(c0 + u) * plane + t] : 0.0f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c0 + u < P.C) map_pred[(size_t)(c0 + u) * plane + t] = fmaxf(v[u], 0.0f);
    }
    return;
  }
  TapAddr ta[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ta[q] = tap_addr(rt[q], touch && in_img[q], P);
  for (int c = 0; c < P.C; ++c) {
    const float last = maps_last[(size_t)c * plane + t];
    float tr = 0.0f;
    if (c != 2 && c != 3) {           // (a pixel of this wave that touches nothing reads element 0 sixteen times and gets 0)
      const float* vc = view + (size_t)c * P.vr * P.vr;
      float w[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) w[q][k] = vc[ta[q].o[k]];
      float vq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float nwv = ta[q].ok[0] ? w[q][0] : 0.0f, nev = ta[q].ok[1] ? w[q][1] : 0.0f;
        const float swv = ta[q].ok[2] ? w[q][2] : 0.0f, sev = ta[q].ok[3] ? w[q][3] : 0.0f;
        float acc = nwv * rt[q].nw;
        acc = fmaf(nev, rt[q].ne, acc);
        acc = fmaf(swv, rt[q].sw, acc);
        acc = fmaf(sev, rt[q].se, acc);
        vq[q] = (touch && in_img[q]) ? acc : 0.0f;
      }
      tr = vq[0] * tt.nw;
      tr = fmaf(vq[1], tt.ne, tr);
      tr = fmaf(vq[2], tt.sw, tr);
      tr = fmaf(vq[3], tt.se, tr);
      if (!touch) tr = 0.0f;
    }
    map_pred[(size_t)c * plane + t] = fmaxf(last, tr);     // torch.max over the stacked pair (mapping.py:175-177)
  }
}


// ---- Agent_State.update_local_map's bookkeeping after the projection (nav/agent/agent_state.py:281-296), one launch ----
// channel 2 (current location) cleared, the trajectory square set in channels 2 and 3, the explored-area footprint stamped
// into channel 1 around up to two centres.  One thread per map cell.
struct MarkP {
  int m, r0, r1, c0, c1, rad, n_centres, cr[2], cc[2];
};
__device__ __forceinline__ bool stamp_hits(int r, int c, int cr, int cc, int m, int rad, const unsigned char* __restrict__ selem) {
  // a footprint cell (dr, dc) lands on row cr + dr, or on cr + dr + m when that is negative (torch's negative indices wrap)
  int dr = r - cr, dc = c - cc;
  if (dr > rad) dr -= m;
  if (dc > rad) dc -= m;
  if (dr < -rad || dr > rad || dc < -rad || dc > rad) return false;
  if (cr + dr >= 0 ? (cr + dr != r) : (cr + dr + m != r)) return false;
  if (cc + dc >= 0 ? (cc + dc != c) : (cc + dc + m != c)) return false;
  return selem[(dr + rad) * (2 * rad + 1) + (dc + rad)] != 0;
}
__global__ __launch_bounds__(256) void map_mark_agent_kernel(float* __restrict__ local_map, const unsigned char* __restrict__ selem, MarkP p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.m * p.m) return;
  const int r = i / p.m, c = i - r * p.m;
  const size_t plane = (size_t)p.m * p.m;
  const bool traj = r >= p.r0 && r < p.r1 && c >= p.c0 && c < p.c1;
  local_map[2 * plane + i] = traj ? 1.0f : 0.0f;
  if (traj) local_map[3 * plane + i] = 1.0f;
  bool hit = false;
  for (int k = 0; k < p.n_centres; ++k) hit = hit || stamp_hits(r, c, p.cr[k], p.cc[k], p.m, p.rad, selem);
  if (hit) local_map[plane + i] = 1.0f;
}
}  // namespace
}  // namespace peanut

using namespace peanut;

struct peanut_map {
  peanut_map_cfg cfg{};
  MapP P{};
  float* coords = nullptr;   // [3][N]
  float* pos = nullptr;      // [3][N]
  unsigned *keys = nullptr, *seg = nullptr, *skeys = nullptr, *sidx = nullptr;
  int* cell_head = nullptr;  // [vr*vr*zb], -1 = empty: start of the cell's segment
  int *cell_cnt = nullptr, *cell_first = nullptr, *cell_fill = nullptr, *cursor = nullptr;
  StairStats* stats = nullptr;
  float* wts6 = nullptr;     // [6][N]
  float* feat_s = nullptr;   // [ncat][N]
  float* proj = nullptr;     // [2][F][vr][vr]
  float* view = nullptr;     // [C][vr][vr]
  float* obs_dec = nullptr;  // [C][N]: the observation at du_scale > 1 (map_decimate_kernel)
  WarpT* wt = nullptr;
  bool use_graph = false;    // peanut_map_use_graph: the launches of a step replayed as one hipGraph
  GraphCache graphs;
  ~peanut_map() {
    graphs.clear();
    for (void* p : {(void*)coords, (void*)pos, (void*)keys, (void*)seg, (void*)skeys, (void*)sidx, (void*)cell_head,
                    (void*)cell_cnt, (void*)cell_first, (void*)cell_fill, (void*)cursor,
                    (void*)stats, (void*)wts6, (void*)feat_s, (void*)proj, (void*)view, (void*)obs_dec, (void*)wt})
      if (p) (void)hipFree(p);
  }
};

extern "C" {

int peanut_map_create(peanut_map_t** out, const peanut_map_cfg* c) {
  if (!out || !c) return fail(PEANUT_EINVAL, "peanut_map_create: null argument");
  if (c->du_scale < 1 || c->du_scale > 8 || c->frame_height % c->du_scale || c->frame_width % c->du_scale)
    return fail(PEANUT_EINVAL, "peanut_map_create: du_scale must be 1..8 and divide the frame size");
  if (c->frame_height < 2 || c->frame_width < 2 || c->num_sem_categories < 5 || c->num_sem_categories > 28 ||
      c->map_resolution < 1 || c->vision_range < 2 || c->global_downscaling < 1)
    return fail(PEANUT_EINVAL, "peanut_map_create: unsupported configuration");
  auto h = std::make_unique<peanut_map>();
  h->cfg = *c;
  MapP& P = h->P;
  P.du = c->du_scale; P.fh = c->frame_height; P.fw = c->frame_width;
  P.h = P.fh / P.du; P.w = P.fw / P.du; P.N = P.h * P.w;
  P.ncat = c->num_sem_categories; P.F = 1 + P.ncat; P.C = 4 + P.ncat;
  P.vr = c->vision_range;
  const int res = c->map_resolution;
  const int max_h = (int)(360 / res), min_h = (int)(-40 / res);      // mapping.py:31-32 (int() truncation)
  P.zb = max_h - min_h;
  const int local_cm = c->map_size_cm / c->global_downscaling;
  P.M = local_cm / res;
  if ((long long)P.vr * P.vr * P.zb >= (long long)INVALID_KEY)
    return fail(PEANUT_EINVAL, "peanut_map_create: voxel grid too large for 20-bit keys");
  // camera matrix in double like numpy, then to fp32 as torch does for python scalars
  P.xc = (float)((P.fw - 1.0) / 2.0);
  P.zc = (float)((P.fh - 1.0) / 2.0);
  P.f = (float)((P.fw / 2.0) / tan(c->hfov / 2.0 * 3.14159265358979323846 / 180.0));   // np.deg2rad(x) = x * pi / 180
  P.agent_h = (float)(c->camera_height * 100.0);
  P.shift_x = (float)(P.vr * res / 2);
  P.res = (float)res;
  P.vr_half = (float)(P.vr / 2);                 // vision_range // 2.
  P.vr_f = (float)P.vr;
  P.z_mid = floorf((float)(max_h + min_h) / 2.0f);   // (max_h + min_h) // 2.
  P.z_span = (float)(max_h - min_h);
  P.min_z = (int)(25.0 / res - min_h);
  P.max_z = (int)((c->camera_height * 100.0 + 1) / res - min_h);
  P.thr_map = (float)c->map_pred_threshold; P.thr_exp = (float)c->exp_pred_threshold; P.thr_cat = (float)c->cat_pred_threshold;
  P.allh_mask = (P.ncat <= 16) ? ((1u << (1 + 5)) | (1u << (1 + 2))) : ((1u << (1 + 3)) | (1u << (1 + 9)) | (1u << (1 + 14)));
  P.x1 = local_cm / (res * 2) - P.vr / 2;
  P.y1 = local_cm / (res * 2);
  P.half_cells = (float)(local_cm / (res * 2));
  P.toilet_ch = 4 + 4;
  if (P.y1 + P.vr > P.M || P.x1 < 0 || P.x1 + P.vr > P.M) return fail(PEANUT_EINVAL, "peanut_map_create: window outside map");
  const size_t N = P.N, cells = (size_t)P.vr * P.vr;
  PEANUT_HIP_CHECK(hipMalloc(&h->coords, 3 * N * sizeof(float)));
  PEANUT_HIP_CHECK(hipMalloc(&h->pos, 3 * N * sizeof(float)));
  PEANUT_HIP_CHECK(hipMalloc(&h->keys, N * sizeof(unsigned)));
  PEANUT_HIP_CHECK(hipMalloc(&h->seg, N * sizeof(unsigned)));
  PEANUT_HIP_CHECK(hipMalloc(&h->skeys, N * sizeof(unsigned)));
  PEANUT_HIP_CHECK(hipMalloc(&h->sidx, N * sizeof(unsigned)));
  PEANUT_HIP_CHECK(hipMalloc(&h->cell_head, cells * P.zb * sizeof(int)));
  PEANUT_HIP_CHECK(hipMemset(h->cell_head, 0xff, cells * P.zb * sizeof(int)));
  PEANUT_HIP_CHECK(hipMalloc(&h->cell_cnt, cells * P.zb * sizeof(int)));
  PEANUT_HIP_CHECK(hipMemset(h->cell_cnt, 0, cells * P.zb * sizeof(int)));
  PEANUT_HIP_CHECK(hipMalloc(&h->cell_fill, cells * P.zb * sizeof(int)));
  PEANUT_HIP_CHECK(hipMemset(h->cell_fill, 0, cells * P.zb * sizeof(int)));
  PEANUT_HIP_CHECK(hipMalloc(&h->cell_first, cells * P.zb * sizeof(int)));
  {
    std::vector<int> big(cells * P.zb, 0x7fffffff);
    PEANUT_HIP_CHECK(hipMemcpy(h->cell_first, big.data(), big.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  PEANUT_HIP_CHECK(hipMalloc(&h->cursor, sizeof(int)));
  PEANUT_HIP_CHECK(hipMemset(h->cursor, 0, sizeof(int)));
  PEANUT_HIP_CHECK(hipMalloc(&h->stats, sizeof(StairStats)));
  {
    const StairStats init = {0u, 0u, 0u, 0u, 0xffffffffu};   // map_view re-arms it after every frame
    PEANUT_HIP_CHECK(hipMemcpy(h->stats, &init, sizeof(init), hipMemcpyHostToDevice));
  }
  PEANUT_HIP_CHECK(hipMalloc(&h->wts6, 6 * N * sizeof(float)));
  PEANUT_HIP_CHECK(hipMalloc(&h->feat_s, (size_t)P.ncat * N * sizeof(float)));
  PEANUT_HIP_CHECK(hipMalloc(&h->proj, 2 * P.F * cells * sizeof(float)));
  PEANUT_HIP_CHECK(hipMemset(h->proj, 0, 2 * P.F * cells * sizeof(float)));
  PEANUT_HIP_CHECK(hipMalloc(&h->view, P.C * cells * sizeof(float)));
  PEANUT_HIP_CHECK(hipMalloc(&h->wt, sizeof(WarpT)));
  if (P.du > 1) PEANUT_HIP_CHECK(hipMalloc(&h->obs_dec, (size_t)P.C * N * sizeof(float)));
  PEANUT_HIP_CHECK(hipDeviceSynchronize());
  *out = h.release();
  return 0;
}

void peanut_map_destroy(peanut_map_t* h) { delete h; }

int peanut_map_dims(peanut_map_t* h, int dims[4]) {
  if (!h || !dims) return fail(PEANUT_EINVAL, "null argument");
  dims[0] = h->P.C; dims[1] = h->P.M; dims[2] = h->P.vr; dims[3] = h->P.N;
  return 0;
}

int peanut_map_forward(peanut_map_t* h, const float* obs, const float* pose_obs, const float* maps_last,
                       float* poses_inout, float* fp_map_pred, float* map_pred, void* stream) {
  if (!h || !obs || !pose_obs || !maps_last || !poses_inout || !fp_map_pred || !map_pred)
    return fail(PEANUT_EINVAL, "peanut_map_forward: null argument");
  if (maps_last == map_pred) return fail(PEANUT_EINVAL, "peanut_map_forward: map_pred must not alias maps_last");
  hipStream_t s = (hipStream_t)stream;
  const float* obs_full = obs;
  auto enqueue = [&]() -> int {
    const MapP& P = h->P;
    const int nb = (P.N + 255) / 256;
    const float* obs = obs_full;
    if (P.du > 1) {
      hipLaunchKernelGGL(map_decimate_kernel, dim3(nb, 1 + P.ncat), dim3(256), 0, s, obs_full, h->obs_dec, P);
      obs = h->obs_dec;
    }
    hipLaunchKernelGGL(map_points_kernel, dim3(nb), dim3(256), 0, s, obs, h->coords, h->stats, P);
    hipLaunchKernelGGL(map_keys_kernel, dim3(nb), dim3(256), 0, s, obs, h->coords, h->stats, h->pos, h->keys, h->cell_cnt,
                       h->cell_first, P);
    hipLaunchKernelGGL(map_alloc_kernel, dim3(nb), dim3(256), 0, s, h->keys, h->cell_cnt, h->cell_first, h->cell_head, h->cursor, P.N);
    hipLaunchKernelGGL(map_fill_kernel, dim3(nb), dim3(256), 0, s, h->keys, h->cell_head, h->cell_fill, h->seg, P.N);
    hipLaunchKernelGGL(map_place_kernel, dim3(nb), dim3(256), 0, s, h->keys, h->cell_head, h->cell_cnt, h->seg, h->cursor, obs, h->pos,
                       h->skeys, h->sidx, h->wts6, h->feat_s, P);
    const long long vt = (long long)P.N * 8 * P.F;
    hipLaunchKernelGGL(map_voxels_kernel, dim3((unsigned)((vt + 255) / 256)), dim3(256), 0, s, h->wts6, h->feat_s, h->skeys,
                       h->cell_head, h->cell_cnt, h->proj, P);
    const int nfin = (std::max(P.vr * P.vr, P.N) + 255) / 256;
    hipLaunchKernelGGL(map_finish_kernel, dim3(nfin), dim3(256), 0, s, h->proj, h->view, fp_map_pred, h->stats, pose_obs, poses_inout,
                       h->wt, h->keys, h->cell_head, h->cell_cnt, h->cell_first, h->cell_fill, h->cursor, P);
    hipLaunchKernelGGL(map_warp_kernel, dim3((P.M * P.M + 255) / 256), dim3(256), 0, s, h->view, maps_last, map_pred,
                       h->wt, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PEANUT_EHIP, std::string("peanut_map_forward: ") + hipGetErrorString(e));
    return 0;
  };
  if (!h->use_graph) return enqueue();
  return h->graphs.run({(uintptr_t)obs, (uintptr_t)pose_obs, (uintptr_t)maps_last, (uintptr_t)poses_inout,
                        (uintptr_t)fp_map_pred, (uintptr_t)map_pred, (uintptr_t)s}, s, enqueue);
}

int peanut_map_mark_agent(float* local_map, int channels, int m, int r0, int r1, int c0, int c1, const uint8_t* selem, int selem_radius,
                          int n_centres, const int* centres_rc, void* stream) {
  if (!local_map || !selem || (n_centres > 0 && !centres_rc)) return fail(PEANUT_EINVAL, "peanut_map_mark_agent: null argument");
  if (channels < 4 || m < 1 || selem_radius < 0 || 2 * selem_radius + 1 > m || n_centres < 0 || n_centres > 2)
    return fail(PEANUT_EINVAL, "peanut_map_mark_agent: bad dimensions");
  if (r0 < 0 || c0 < 0 || r1 > m || c1 > m) return fail(PEANUT_EINVAL, "peanut_map_mark_agent: the trajectory square is not a normalised slice");
  MarkP p{m, r0, r1, c0, c1, selem_radius, n_centres, {0, 0}, {0, 0}};
  for (int k = 0; k < n_centres; ++k) {
    const int cr = centres_rc[2 * k], cc = centres_rc[2 * k + 1];
    // the footprint's index range must be one torch accepts: [-m, m)
    if (cr - selem_radius < -m || cr + selem_radius >= m || cc - selem_radius < -m || cc + selem_radius >= m)
      return fail(PEANUT_EINVAL, "peanut_map_mark_agent: footprint index out of range");
    p.cr[k] = cr; p.cc[k] = cc;
  }
  hipLaunchKernelGGL(map_mark_agent_kernel, dim3((m * m + 255) / 256), dim3(256), 0, (hipStream_t)stream, local_map, selem, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("peanut_map_mark_agent: ") + hipGetErrorString(e));
}

int peanut_map_use_graph(peanut_map_t* h, int enable) {
  if (!h) return fail(PEANUT_EINVAL, "null handle");
  h->use_graph = enable != 0;
  if (!h->use_graph) h->graphs.clear();
  return 0;
}

}  // extern "C"
