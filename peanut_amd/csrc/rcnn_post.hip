// peanut_rcnn_inference: the whole of detectron2's GeneralizedRCNN.inference + detector_postprocess as configured by
// nav/agent/utils/COCO-InstSeg/mask_rcnn_R_101_cat9.yaml -- what `DefaultPredictor(img)["instances"]` yields at
// nav/agent/utils/segmentation.py:45 -- behind ONE C entry point, with no torch in it:
//
//   front end (rcnn_api.hip)  ->  RPN proposal selection  ->  box head  ->  detection selection  ->  mask head  ->  paste
//
// detectron2 is third party and absent from the reference checkout: the stages restate its
// published v0.6 definitions (find_top_rpn_proposals, Box2BoxTransform.apply_deltas, ROIPooler level assignment,
// fast_rcnn_inference_single_image, mask_rcnn_inference, detector_postprocess / paste_masks_in_image); parity is
// pinned against oracle/rcnn_ref.py only (PARITY UNPINNED).
//
// Everything data-dependent stays on the device with fixed capacities (1000 proposals and 9000 class candidates per
// image) and device-side counts; the only host read is the number of detections per image, which the caller needs
// anyway and which sizes the mask head.  The selection stages that detectron2 writes as torch glue are kernels here:
//   rpn_topk_kernel      per (level, image) top-k of the objectness logits: three-pass radix select on the
//                        order-preserving key bits (LDS histogram), then a 1024-wide bitonic sort -- one workgroup
//   rpn_decode_kernel    anchors (closed form) + apply_deltas + clip + validity, 64-bit sort keys (score | position)
//   sort_keys_kernel     per image bitonic sort of up to 16384 keys in LDS (ties: first position first, = a stable sort)
//   nms kernels          the block-bit-matrix NMS of rcnn_ops.hip with device-side segment lengths
//   compact_*_kernel     ordered compaction of the kept boxes (+ FPN level assignment / output scaling)
//   box_post_kernel      softmax + per-class box decode + score threshold -> class candidates
//   mask_prob_kernel     class channel of the mask logits + sigmoid, un-shuffling the 2x2 transposed-conv outputs
#include <math.h>
#include <string.h>

#include <algorithm>

#include "rcnn_internal.h"

#pragma clang fp contract(off)

namespace peanut {
namespace {

constexpr int kLevels = 5;
constexpr int kMaxAnchors = 8;
constexpr float kScaleClamp = 4.135166556742356f;   // log(1000 / 16), Box2BoxTransform

__device__ __forceinline__ unsigned ord_key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_key_inv(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct RpnLevels {
  const float* obj[kLevels];      // [B, h, w, A]
  const float* delta[kLevels];    // [B, h, w, 4A]
  int h[kLevels], w[kLevels], n[kLevels], k[kLevels], koff[kLevels + 1];
  float stride[kLevels];
  float cell[kLevels][kMaxAnchors][4];   // cell anchors (x0, y0, x1, y1) around (0, 0)
  int A;
};

// The per-level top-k in slices (round 5): a level's objectness logits are cut into up to 8 ranges, every range selects its own
// k largest (one workgroup each: the finest level of an 800 x 1067 frame is 160 000 logits, and ONE workgroup making its four
// passes over them was 115 us, the longest kernel of the back half at batch 1), and rpn_rank_select_kernel takes the k largest of
// a level's <= 8 k candidates by counting -- which also leaves them in score order.
constexpr int kMaxSlices = 32, kMaxSlicesPerLevel = 8;
struct RpnSlices {
  int n_items, ctot;                                     // workgroups per image; candidates per image
  int lvl[kMaxSlices], start[kMaxSlices], cnt[kMaxSlices], ks[kMaxSlices], coff[kMaxSlices];
  int lvl_coff[kLevels], lvl_cnt[kLevels];               // a level's candidates: offset and number
};

// bitonic sort (descending) of n = power of two 64-bit keys in LDS, any thread count that divides n / 2 evenly
__device__ __forceinline__ void bitonic_desc(unsigned long long* s, int n) {
  for (int size = 2; size <= n; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1));        // index of the lower element of pair t
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const unsigned long long a = s[lo], b = s[hi];
        if ((a < b) == desc) { s[lo] = b; s[hi] = a; }
      }
    }
  __syncthreads();
}

// the bin (counting from the top) in which the `need`-th largest element falls, and how many elements lie above it
__device__ __forceinline__ void find_bin(const int* hist, int nbins, int need, int* s_part, int* s_out) {
  const int W = nbins / 64;
  if (threadIdx.x < 64) {
    int a = 0;
    for (int i = 0; i < W; ++i) a += hist[threadIdx.x * W + i];
    s_part[threadIdx.x] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0, g = 63;
    for (; g > 0; --g) {
      if (acc + s_part[g] >= need) break;
      acc += s_part[g];
    }
    int b = (g + 1) * W - 1;
    for (; b > g * W; --b) {
      if (acc + hist[b] >= need) break;
      acc += hist[b];
    }
    s_out[0] = b;
    s_out[1] = acc;
  }
  __syncthreads();
}

// ---- per (level, image): the k largest objectness logits, sorted descending (ties: lower index first) ----
// SLICED: workgroup blockIdx.x takes range sl.start/cnt of level sl.lvl and writes its sl.ks largest as 64-bit keys (score | index
// within the level), unordered, to cand[b][sl.coff ...]; else: one workgroup per level, the k largest sorted into sel_idx / sel_score
template <bool SLICED>
__global__ __launch_bounds__(1024) void rpn_topk_kernel(const RpnLevels lv, const RpnSlices sl, int Ktot, int* __restrict__ sel_idx,
                                                        float* __restrict__ sel_score, unsigned long long* __restrict__ cand) {
  const int b = blockIdx.y;
  const int l = SLICED ? sl.lvl[blockIdx.x] : (int)blockIdx.x;
  const int first = SLICED ? sl.start[blockIdx.x] : 0;
  const int n = SLICED ? sl.cnt[blockIdx.x] : lv.n[l], k = SLICED ? sl.ks[blockIdx.x] : lv.k[l];
  const float* x = lv.obj[l] + (size_t)b * lv.n[l] + first;
  __shared__ int hist[2048];
  __shared__ int part[64];
  __shared__ int res[2];
  __shared__ int cnt, taken_eq;
  __shared__ unsigned long long sbuf[1024];
  __shared__ int wave_cnt[16];
  const int tid = threadIdx.x;
  int need = k;
  unsigned prefix = 0;   // bits of the threshold key fixed so far
  // three radix passes: bits 31..21, 20..10, 9..0
  const int shift[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  int eq_count = 0;
  for (int pass = 0; pass < 3; ++pass) {
    const int nb = 1 << bits[pass];
    for (int i = tid; i < 2048; i += 1024) hist[i] = 0;
    __syncthreads();
    // sixteen loads in flight per thread before the first LDS atomic: at batch 1 this kernel is five workgroups, and a
    // load-use-load chain over the 163 200 logits of the finest level cost 155 us (profiles/r3h; combining equal bins inside
    // the wave before the atomic was measured at four times the cost: the lanes of a wave hit too many distinct bins, r8h)
    for (int i0 = tid; i0 < n; i0 += 1024 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = (i0 + u * 1024 < n) ? x[i0 + u * 1024] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (i0 + u * 1024 >= n) break;
        const unsigned key = ord_key(v[u]);
        const bool match = pass == 0 || (key >> (shift[pass] + bits[pass])) == prefix;
        if (match) atomicAdd(&hist[(key >> shift[pass]) & (nb - 1)], 1);
      }
    }
    __syncthreads();
    find_bin(hist, nb < 64 ? 64 : nb, need, part, res);
    prefix = (prefix << bits[pass]) | (unsigned)res[0];
    need -= res[1];
    if (pass == 2) eq_count = hist[res[0]];
    __syncthreads();
  }
  const unsigned thr = prefix;      // exact key of the k-th largest element; `need` (>= 1) of the elements equal to it are taken
  if (tid == 0) { cnt = 0; taken_eq = 0; }
  sbuf[tid] = 0ull;
  __syncthreads();
  if (eq_count == need) {           // every element equal to the threshold is in: one unordered pass
    for (int i0 = tid; i0 < n; i0 += 1024 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (i0 + u * 1024 < n) ? x[i0 + u * 1024] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 1024;
        if (i >= n) break;
        const unsigned key = ord_key(v[u]);
        if (key >= thr) sbuf[atomicAdd(&cnt, 1)] = ((unsigned long long)key << 32) | (0xffffffffu - (unsigned)(first + i));
      }
    }
  } else {                          // ties across the cut: the first `need` of them by index (ordered chunks)
    for (int i0 = 0; i0 < n; i0 += 1024) {
      const int i = i0 + tid;
      const unsigned key = i < n ? ord_key(x[i]) : 0u;
      if (i < n && key > thr) sbuf[atomicAdd(&cnt, 1)] = ((unsigned long long)key << 32) | (0xffffffffu - (unsigned)(first + i));
      const bool eq = i < n && key == thr;
      const unsigned long long bal = __ballot(eq);
      const int lane = tid & 63, wv = tid >> 6;
      if (lane == 0) wave_cnt[wv] = __popcll(bal);
      __syncthreads();
      int before = taken_eq;
      for (int q = 0; q < wv; ++q) before += wave_cnt[q];
      before += __popcll(bal & ((1ull << lane) - 1ull));
      if (eq && before < need) sbuf[atomicAdd(&cnt, 1)] = ((unsigned long long)key << 32) | (0xffffffffu - (unsigned)(first + i));
      __syncthreads();
      if (tid == 0) { int t = 0; for (int q = 0; q < 16; ++q) t += wave_cnt[q]; taken_eq += t; }
      __syncthreads();
    }
  }
  if (SLICED) {
    __syncthreads();
    if (tid < k) cand[(size_t)b * sl.ctot + sl.coff[blockIdx.x] + tid] = sbuf[tid];
    return;
  }
  bitonic_desc(sbuf, 1024);
  if (tid < k) {
    const unsigned long long e = sbuf[tid];
    const size_t o = (size_t)b * Ktot + lv.koff[l] + tid;
    sel_idx[o] = (int)(0xffffffffu - (unsigned)(e & 0xffffffffull));
    sel_score[o] = ord_key_inv((unsigned)(e >> 32));
  }
}

// the k largest of a level's candidates, in descending order, by counting (keys are distinct: the index is part of the key);
// grid (ceil(max candidates / 64), levels, B), 512 threads as in rank_sort_keys_kernel below
__global__ __launch_bounds__(512) void rpn_rank_select_kernel(const RpnLevels lv, const RpnSlices sl, int Ktot,
                                                              const unsigned long long* __restrict__ cand, int* __restrict__ sel_idx,
                                                              float* __restrict__ sel_score) {
  constexpr int NW = 8;
  __shared__ __attribute__((aligned(16))) unsigned long long sk[kMaxSlicesPerLevel * 1024];
  __shared__ int part[NW][64];
  const int l = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = sl.lvl_cnt[l];
  if ((int)blockIdx.x * 64 >= C) return;
  const unsigned long long* g = cand + (size_t)b * sl.ctot + sl.lvl_coff[l];
  for (int i = tid; i < C; i += 64 * NW) sk[i] = g[i];
  __syncthreads();
  const int mine_i = blockIdx.x * 64 + lane;
  const unsigned long long mine = mine_i < C ? sk[mine_i] : 0ull;
  const int per = ((C + NW - 1) / NW + 1) & ~1, j0 = min(C, wave * per), j1 = min(C, j0 + per);
  int larger = 0, j = j0;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  for (; j + 16 <= j1; j += 16) {
    u64x2 kk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) kk[u] = *reinterpret_cast<const u64x2*>(&sk[j + 2 * u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) larger += (kk[u][0] > mine) + (kk[u][1] > mine);
  }
  for (; j < j1; ++j) larger += sk[j] > mine;
  part[wave][lane] = larger;
  __syncthreads();
  if (wave == 0 && mine_i < C) {
    int r = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) r += part[w][lane];
    if (r < lv.k[l]) {
      const size_t o = (size_t)b * Ktot + lv.koff[l] + r;
      sel_idx[o] = (int)(0xffffffffu - (unsigned)(mine & 0xffffffffull));
      sel_score[o] = ord_key_inv((unsigned)(mine >> 32));
    }
  }
}

// Box2BoxTransform.apply_deltas for one box (same operation order as the torch expression)
__device__ __forceinline__ void apply_deltas1(const float* d, float ax0, float ay0, float ax1, float ay1, float wx, float wy,
                                              float ww, float wh, float* o) {
  const float widths = ax1 - ax0, heights = ay1 - ay0;
  const float ctr_x = ax0 + 0.5f * widths, ctr_y = ay0 + 0.5f * heights;
  const float dx = d[0] / wx, dy = d[1] / wy;
  const float dw = fminf(d[2] / ww, kScaleClamp), dh = fminf(d[3] / wh, kScaleClamp);
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
  const float pw = expf(dw) * widths, ph = expf(dh) * heights;
  o[0] = pcx - 0.5f * pw; o[1] = pcy - 0.5f * ph; o[2] = pcx + 0.5f * pw; o[3] = pcy + 0.5f * ph;
}
__device__ __forceinline__ bool finite4(const float* b) { return isfinite(b[0]) && isfinite(b[1]) && isfinite(b[2]) && isfinite(b[3]); }
__device__ __forceinline__ void clip4(float* b, float h, float w) {
  b[0] = fminf(fmaxf(b[0], 0.f), w); b[1] = fminf(fmaxf(b[1], 0.f), h);
  b[2] = fminf(fmaxf(b[2], 0.f), w); b[3] = fminf(fmaxf(b[3], 0.f), h);
}

// ---- decode the selected anchors: RPN.predict_proposals + the validity tests of find_top_rpn_proposals ----
__global__ __launch_bounds__(256) void rpn_decode_kernel(const RpnLevels lv, int B, int Ktot, int Kpad, const int* __restrict__ sel_idx,
                                                         const float* __restrict__ sel_score, float img_h, float img_w, float wx,
                                                         float wy, float ww, float wh, float* __restrict__ cbox,
                                                         unsigned long long* __restrict__ ckey, int* __restrict__ ccat) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * Kpad) return;
  const int b = i / Kpad, j = i - b * Kpad;
  if (j >= Ktot) { ckey[i] = 0ull; return; }
  int l = 0;
  while (l + 1 < kLevels && j >= lv.koff[l + 1]) ++l;
  const size_t o = (size_t)b * Ktot + j;
  const int idx = sel_idx[o];
  const float score = sel_score[o];
  const int A = lv.A, a = idx % A, cellpos = idx / A;
  const int yy = cellpos / lv.w[l], xx = cellpos - yy * lv.w[l];
  const float sx = (float)xx * lv.stride[l], sy = (float)yy * lv.stride[l];
  const float* c = lv.cell[l][a];
  const float* d = lv.delta[l] + ((size_t)b * lv.n[l] / A + cellpos) * (4 * A) + a * 4;
  float box[4];
  apply_deltas1(d, sx + c[0], sy + c[1], sx + c[2], sy + c[3], wx, wy, ww, wh, box);
  bool valid = finite4(box) && isfinite(score);
  clip4(box, img_h, img_w);
  valid = valid && (box[2] - box[0]) > 0.f && (box[3] - box[1]) > 0.f;
  float* ob = cbox + o * 4;
  ob[0] = box[0]; ob[1] = box[1]; ob[2] = box[2]; ob[3] = box[3];
  ccat[o] = l;
  ckey[i] = valid ? (((unsigned long long)ord_key(score) << 32) | (0xffffffffu - (unsigned)j)) : 0ull;
}

// ---- per image: sort the keys descending (key 0 = invalid, ends up last); nvalid[b] = number of valid keys ----
template <int Kpad>
__global__ __launch_bounds__(1024) void sort_keys_kernel(unsigned long long* __restrict__ keys, int* __restrict__ nvalid) {
  __shared__ unsigned long long sk[Kpad];
  unsigned long long* g = keys + (size_t)blockIdx.x * Kpad;
  for (int i = threadIdx.x; i < Kpad; i += 1024) sk[i] = g[i];
  bitonic_desc(sk, Kpad);
  for (int i = threadIdx.x; i < Kpad; i += 1024) {
    const unsigned long long v = sk[i];
    g[i] = v;
    if (v != 0ull && (i + 1 == Kpad || sk[i + 1] == 0ull)) nvalid[blockIdx.x] = i + 1;
  }
  if (threadIdx.x == 0 && sk[0] == 0ull) nvalid[blockIdx.x] = 0;
}
// The same ordering by counting (round 5): the keys are distinct (a candidate's position is part of its key) except for
// 0 = invalid, so a key's place in the descending order is the number of larger keys.  A bitonic sort of 8 192 keys is 91
// barrier-separated passes of ONE workgroup (65 us, whatever the batch); here every key counts the larger ones among the n_live
// real entries on its own: a workgroup takes 64 keys, its eight waves an eighth of the comparisons each (a wave reads one LDS
// word per step, the same for all lanes), n_live / 64 workgroups per image.  Out of place (every workgroup reads all keys).
__global__ __launch_bounds__(512) void rank_sort_keys_kernel(const unsigned long long* __restrict__ keys, int Kpad, int n_live,
                                                             unsigned long long* __restrict__ sorted, int* __restrict__ nvalid) {
  constexpr int NW = 8;                      // waves = slices of the comparison range
  __shared__ __attribute__((aligned(16))) unsigned long long sk[8192];
  __shared__ int part[NW][64];
  __shared__ int nz[NW];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long* g = keys + (size_t)b * Kpad;
  int nonzero = 0;
  for (int i = tid; i < n_live; i += 64 * NW) {
    const unsigned long long v = g[i];
    sk[i] = v;
    nonzero += v != 0ull;
  }
  for (int o = 32; o > 0; o >>= 1) nonzero += __shfl_xor(nonzero, o);
  if (lane == 0) nz[wave] = nonzero;
  __syncthreads();
  const int mine_i = blockIdx.x * 64 + lane;
  const unsigned long long mine = mine_i < n_live ? sk[mine_i] : 0ull;
  const int per = ((n_live + NW - 1) / NW + 1) & ~1, j0 = min(n_live, wave * per), j1 = min(n_live, j0 + per);     // even starts: 16-byte reads
  int larger = 0;
  int j = j0;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  for (; j + 16 <= j1; j += 16) {     // sixteen keys in flight (one read and its wait per step left the loop at ~120 cycles a key)
    u64x2 k[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) k[u] = *reinterpret_cast<const u64x2*>(&sk[j + 2 * u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) larger += (k[u][0] > mine) + (k[u][1] > mine);
  }
  for (; j < j1; ++j) larger += sk[j] > mine;
  part[wave][lane] = larger;
  __syncthreads();
  if (wave == 0 && mine != 0ull) {
    int r = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) r += part[w][lane];
    sorted[(size_t)b * Kpad + r] = mine;
  }
  if (blockIdx.x == 0 && tid == 0) {
    int t = 0;
    for (int w = 0; w < NW; ++w) t += nz[w];
    nvalid[b] = t;
  }
}

// the same for the first count[b] keys of each segment (all non-zero): sorts the next power of two, at least 64
template <int KMAX>
__global__ __launch_bounds__(1024) void sort_keys_counted_kernel(unsigned long long* __restrict__ keys, const int* __restrict__ count,
                                                                 int* __restrict__ nvalid) {
  __shared__ unsigned long long sk[KMAX];
  unsigned long long* g = keys + (size_t)blockIdx.x * KMAX;
  const int n = min(count[blockIdx.x], KMAX);
  int n2 = 64;
  while (n2 < n) n2 <<= 1;
  for (int i = threadIdx.x; i < n2; i += 1024) sk[i] = i < n ? g[i] : 0ull;
  bitonic_desc(sk, n2);
  for (int i = threadIdx.x; i < n; i += 1024) g[i] = sk[i];
  if (threadIdx.x == 0) nvalid[blockIdx.x] = n;
}

// ---- boxes / categories / scores in sorted order (row r of image b <- candidate position encoded in its key) ----
__global__ __launch_bounds__(256) void gather_sorted_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ nvalid,
                                                            int Kpad, int Kcap, int B, const float* __restrict__ cbox, const int* __restrict__ ccat,
                                                            float* __restrict__ sbox, int* __restrict__ scat, float* __restrict__ sscore) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * Kcap) return;
  const int b = i / Kcap, r = i - b * Kcap;
  const unsigned long long key = r < nvalid[b] ? keys[(size_t)b * Kpad + r] : 0ull;   // past the count: whatever an earlier call left
  float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
  int cat = -1;
  float sc = 0.f;
  if (key != 0ull) {
    const int j = (int)(0xffffffffu - (unsigned)(key & 0xffffffffull));
    box = *reinterpret_cast<const float4*>(cbox + ((size_t)b * Kcap + j) * 4);
    cat = ccat[(size_t)b * Kcap + j];
    sc = ord_key_inv((unsigned)(key >> 32));
  }
  *reinterpret_cast<float4*>(sbox + (size_t)i * 4) = box;
  scat[i] = cat;
  sscore[i] = sc;
}

// ---- NMS of score-sorted boxes, equal categories only (rcnn_ops.hip), fixed stride n_cap per image, live length
//      count[b] read on the device: the bit matrix of image b has n_cap rows of `words` 64-bit words ----
__global__ __launch_bounds__(64) void nms_mask_dev_kernel(const float* __restrict__ boxes_all, const int* __restrict__ cat_all, int n_cap,
                                                          int words, const int* __restrict__ count, float thr,
                                                          unsigned long long* __restrict__ ws) {
  const int seg = blockIdx.z;
  const int n = min(count[seg], n_cap);
  const int nwords = (n + 63) >> 6;
  const int row_blk = blockIdx.y, col_blk = blockIdx.x;
  if (row_blk >= nwords || col_blk >= nwords || col_blk < row_blk) return;
  const float* boxes = boxes_all + (size_t)seg * n_cap * 4;
  const int* cat = cat_all + (size_t)seg * n_cap;
  unsigned long long* mask = ws + (size_t)seg * n_cap * words;
  const int i = row_blk * 64 + threadIdx.x;
  __shared__ float sb[64][4];
  __shared__ int sc[64];
  const int j0 = col_blk * 64;
  if (j0 + (int)threadIdx.x < n) {
    const float* bx = boxes + (size_t)(j0 + threadIdx.x) * 4;
    sb[threadIdx.x][0] = bx[0]; sb[threadIdx.x][1] = bx[1]; sb[threadIdx.x][2] = bx[2]; sb[threadIdx.x][3] = bx[3];
    sc[threadIdx.x] = cat[j0 + threadIdx.x];
  }
  __syncthreads();
  if (i >= n) return;
  const float* a = boxes + (size_t)i * 4;
  const float ax0 = a[0], ay0 = a[1], ax1 = a[2], ay1 = a[3];
  const float area_a = (ax1 - ax0) * (ay1 - ay0);
  const int ca = cat[i];
  unsigned long long bitsv = 0;
  const int lim = min(64, n - j0);
  for (int k = (row_blk == col_blk ? (int)threadIdx.x + 1 : 0); k < lim; ++k) {
    if (sc[k] != ca) continue;
    const float ix0 = fmaxf(ax0, sb[k][0]), iy0 = fmaxf(ay0, sb[k][1]);
    const float ix1 = fminf(ax1, sb[k][2]), iy1 = fminf(ay1, sb[k][3]);
    const float iw = fmaxf(ix1 - ix0, 0.f), ih = fmaxf(iy1 - iy0, 0.f);
    const float inter = iw * ih;
    const float area_b = (sb[k][2] - sb[k][0]) * (sb[k][3] - sb[k][1]);
    if (inter / (area_a + area_b - inter) > thr) bitsv |= 1ull << k;
  }
  mask[(size_t)i * words + col_blk] = bitsv;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, lane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

// Greedy scan over the suppression matrix of one image, 64 boxes (one matrix word) per step, 16 waves.
//   wave 0: the step's 64 x 64 diagonal block decides which of its still-alive boxes survive (64 readlane rounds);
//   all waves: wave g ORs rows 4g .. 4g+3 of the step (if they survived) into the `removed` words right of the diagonal.
// The rows of step b+1 are loaded -- unconditionally, survivors are only known later -- while step b is decided
// (WPL words per lane in registers, words <= 64 * WPL; WPL = 0: any width, loaded when needed).  The scan stops once
// max_keep boxes are kept: both consumers take the first `cap` survivors in score order, later flags are zero.
template <int WPL>
__global__ __launch_bounds__(1024) void nms_scan_dev_kernel(const unsigned long long* __restrict__ ws, int n_cap, int words,
                                                            const int* __restrict__ count, unsigned char* __restrict__ keep_all,
                                                            int max_keep) {
  extern __shared__ unsigned long long removed[];   // [words]
  __shared__ unsigned long long s_alive;
  __shared__ int s_done;
  const int seg = blockIdx.x;
  const int n = min(count[seg], n_cap);
  const int nwords = (n + 63) >> 6;
  const unsigned long long* mask = ws + (size_t)seg * n_cap * words;
  unsigned char* keep = keep_all + (size_t)seg * n_cap;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int w = tid; w < nwords; w += 1024) removed[w] = 0;
  constexpr int R = WPL > 0 ? WPL : 1;
  unsigned long long cur[4][R], nxt[4][R], diag_cur = 0, diag_nxt = 0;
  auto load_step = [&](int b, unsigned long long (&dst)[4][R], unsigned long long& dg) {
    if constexpr (WPL > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = b * 64 + wave * 4 + r;
#pragma unroll
        for (int q = 0; q < WPL; ++q) {
          const int w = lane + 64 * q;
          dst[r][q] = (row < n && w > b && w < nwords) ? mask[(size_t)row * words + w] : 0ull;
        }
      }
    }
    if (wave == 0) {
      const int i = b * 64 + lane;
      dg = i < n ? mask[(size_t)i * words + b] : 0ull;
    }
  };
  if (nwords > 0) load_step(0, cur, diag_cur);
  __syncthreads();
  int kept = 0;
  for (int b = 0; b < nwords; ++b) {
    if (b + 1 < nwords) load_step(b + 1, nxt, diag_nxt);
    if (wave == 0) {
      unsigned long long alive = ~removed[b];
      if (b == nwords - 1 && (n & 63)) alive &= (1ull << (n & 63)) - 1;
      // walk the boxes that are still alive AND suppress something inside the block (a box with an empty row changes nothing; the
      // serial walk over all 64 lanes was ~2.5 of the 3.4 us a step took): the next one survives and strikes what it suppresses
      const unsigned long long rows = __ballot(diag_cur != 0ull);
      for (unsigned long long todo = alive & rows; todo != 0ull;) {
        const int l = __builtin_ctzll(todo);
        alive &= ~readlane64(diag_cur, l);
        todo = alive & rows & ~((2ull << l) - 1ull);
      }
      int done = 0;
      if (kept + __builtin_popcountll(alive) >= max_keep) {   // keep the first max_keep - kept of them
        unsigned long long a = alive, t = 0;
        for (int k = kept; k < max_keep; ++k) { const unsigned long long low = a & (0ull - a); t |= low; a ^= low; }
        alive = t;
        done = 1;
      }
      kept += __builtin_popcountll(alive);
      const int i = b * 64 + lane;
      if (i < n) keep[i] = (alive >> lane) & 1ull;
      if (lane == 0) { s_alive = alive; s_done = done; }
    }
    __syncthreads();
    const unsigned long long alive = s_alive;
    const int done = s_done;
    if (done) {
      for (int i = (b + 1) * 64 + tid; i < n; i += 1024) keep[i] = 0;
      break;
    }
    if constexpr (WPL > 0) {
#pragma unroll
      for (int q = 0; q < WPL; ++q) {
        unsigned long long acc = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((alive >> (wave * 4 + r)) & 1ull) acc |= cur[r][q];
        if (acc) atomicOr(&removed[lane + 64 * q], acc);
      }
    } else {
      for (int w = b + 1 + lane; w < nwords; w += 64) {
        unsigned long long acc = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = b * 64 + wave * 4 + r;
          if ((alive >> (wave * 4 + r)) & 1ull) acc |= mask[(size_t)row * words + w];
        }
        if (acc) atomicOr(&removed[w], acc);
      }
    }
    __syncthreads();
    if constexpr (WPL > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < WPL; ++q) cur[r][q] = nxt[r][q];
    }
    diag_cur = diag_nxt;
  }
}

void launch_nms_scan(const unsigned long long* ws, int n_cap, int words, const int* count, unsigned char* keep, int max_keep, int B,
                     hipStream_t s) {
  const size_t lds = (size_t)words * 8;
  if (words <= 128) hipLaunchKernelGGL(nms_scan_dev_kernel<2>, dim3(B), dim3(1024), lds, s, ws, n_cap, words, count, keep, max_keep);
  else if (words <= 256) hipLaunchKernelGGL(nms_scan_dev_kernel<4>, dim3(B), dim3(1024), lds, s, ws, n_cap, words, count, keep, max_keep);
  else hipLaunchKernelGGL(nms_scan_dev_kernel<0>, dim3(B), dim3(1024), lds, s, ws, n_cap, words, count, keep, max_keep);
}

// ordered prefix of a predicate over [0, n) with 1024 threads: calls emit(r, position) for the first `limit` hits
template <class Pred, class Emit>
__device__ __forceinline__ int ordered_compact(int n, int limit, Pred pred, Emit emit, int* wave_cnt, int* running) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) *running = 0;
  __syncthreads();
  for (int r0 = 0; r0 < n; r0 += 1024) {
    const int r = r0 + tid;
    const bool hit = r < n && pred(r);
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int pos = *running;
    for (int q = 0; q < wv; ++q) pos += wave_cnt[q];
    pos += __popcll(bal & ((1ull << lane) - 1ull));
    if (hit && pos < limit) emit(r, pos);
    __syncthreads();
    if (tid == 0) { int t = 0; for (int q = 0; q < 16; ++q) t += wave_cnt[q]; *running += t; }
    __syncthreads();
    if (*running >= limit) break;
  }
  return min(*running, limit);
}

// ROIPooler.assign_boxes_to_levels (canonical size 224 at level 4, levels 2..5) -> 0..3
__device__ __forceinline__ int assign_level(const float* b) {
  const float size = sqrtf((b[2] - b[0]) * (b[3] - b[1]));
  float lv = floorf(4.f + log2f(size / 224.f + 1e-8f));
  lv = fminf(fmaxf(lv, 2.f), 5.f);
  return (int)lv - 2;
}

// ---- post-NMS top-k proposals of every image -> rois [B * cap, 5], FPN level, objectness logit, count ----
__global__ __launch_bounds__(1024) void compact_proposals_kernel(const float* __restrict__ sbox, const float* __restrict__ sscore,
                                                                 const unsigned char* __restrict__ keep, const int* __restrict__ nvalid,
                                                                 int Kcap, int cap, float* __restrict__ rois, int* __restrict__ levels,
                                                                 float* __restrict__ logits, int* __restrict__ count,
                                                                 int* __restrict__ n_keys) {
  __shared__ int wave_cnt[16], running;
  const int b = blockIdx.x;
  const int n = min(nvalid[b], Kcap);
  const float* bx = sbox + (size_t)b * Kcap * 4;
  const unsigned char* kp = keep + (size_t)b * Kcap;
  float* ro = rois + (size_t)b * cap * 5;
  int* lo = levels + (size_t)b * cap;
  for (int p = threadIdx.x; p < cap; p += 1024) {      // rows past the count: an empty roi on image 0
    ro[p * 5 + 0] = 0.f; ro[p * 5 + 1] = 0.f; ro[p * 5 + 2] = 0.f; ro[p * 5 + 3] = 0.f; ro[p * 5 + 4] = 0.f;
    lo[p] = 0;
    logits[(size_t)b * cap + p] = 0.f;
  }
  __syncthreads();
  const int c = ordered_compact(
      n, cap, [&](int r) { return kp[r] != 0; },
      [&](int r, int pos) {
        const float* s = bx + (size_t)r * 4;
        float* d = ro + (size_t)pos * 5;
        d[0] = (float)b; d[1] = s[0]; d[2] = s[1]; d[3] = s[2]; d[4] = s[3];
        lo[pos] = assign_level(s);
        logits[(size_t)b * cap + pos] = sscore[(size_t)b * Kcap + r];
      },
      wave_cnt, &running);
  if (threadIdx.x == 0) { count[b] = c; n_keys[b] = 0; }   // n_keys: the detection candidates box_post_kernel appends
}

// ---- NMS level by level (round 5).  batched_nms only ever compares boxes of one pyramid level, and the per-level top-k leaves
//      every level's candidates in score order already: so the suppression matrix is five small blocks per image (16 x 16 words
//      each instead of 75 x 75 with four fifths of the pairs skipped), the greedy scans of the five levels run side by side
//      (16 steps each instead of up to 75 in a row: the scan was the longest kernel of the back half at batch 1), and no global
//      sort comes first.  The post-NMS top-k -- the kept boxes of all levels in score order -- is then taken by counting.
//      Invalid candidates (non-finite or empty boxes: key 0) neither suppress nor survive.
constexpr int kLevelCap = 1024;      // candidates per level and image (rpn_pre_nms_topk <= 1024)
__global__ __launch_bounds__(64) void nms_mask_levels_kernel(const RpnLevels lv, const float* __restrict__ cbox,
                                                             const unsigned long long* __restrict__ ckey, int Ktot, int Kpad, float thr,
                                                             unsigned long long* __restrict__ ws) {
  constexpr int WORDS = kLevelCap / 64;
  const int seg = blockIdx.z, b = seg / kLevels, l = seg - b * kLevels;
  const int n = lv.k[l];
  const int nwords = (n + 63) >> 6;
  const int row_blk = blockIdx.y, col_blk = blockIdx.x;
  if (row_blk >= nwords || col_blk >= nwords || col_blk < row_blk) return;
  const float* boxes = cbox + ((size_t)b * Ktot + lv.koff[l]) * 4;
  const unsigned long long* keys = ckey + (size_t)b * Kpad + lv.koff[l];
  unsigned long long* mask = ws + (size_t)seg * kLevelCap * WORDS;
  const int i = row_blk * 64 + threadIdx.x;
  __shared__ float sb[64][4];
  __shared__ unsigned char sv[64];
  const int j0 = col_blk * 64;
  if (j0 + (int)threadIdx.x < n) {
    const float* bx = boxes + (size_t)(j0 + threadIdx.x) * 4;
    sb[threadIdx.x][0] = bx[0]; sb[threadIdx.x][1] = bx[1]; sb[threadIdx.x][2] = bx[2]; sb[threadIdx.x][3] = bx[3];
    sv[threadIdx.x] = keys[j0 + threadIdx.x] != 0ull;
  }
  __syncthreads();
  if (i >= n) return;
  const float* a = boxes + (size_t)i * 4;
  const float ax0 = a[0], ay0 = a[1], ax1 = a[2], ay1 = a[3];
  const float area_a = (ax1 - ax0) * (ay1 - ay0);
  unsigned long long bitsv = 0;
  if (keys[i] != 0ull) {
    const int lim = min(64, n - j0);
    for (int k = (row_blk == col_blk ? (int)threadIdx.x + 1 : 0); k < lim; ++k) {
      if (!sv[k]) continue;
      const float ix0 = fmaxf(ax0, sb[k][0]), iy0 = fmaxf(ay0, sb[k][1]);
      const float ix1 = fminf(ax1, sb[k][2]), iy1 = fminf(ay1, sb[k][3]);
      const float iw = fmaxf(ix1 - ix0, 0.f), ih = fmaxf(iy1 - iy0, 0.f);
      const float inter = iw * ih;
      const float area_b = (sb[k][2] - sb[k][0]) * (sb[k][3] - sb[k][1]);
      if (inter / (area_a + area_b - inter) > thr) bitsv |= 1ull << k;
    }
  }
  mask[(size_t)i * WORDS + col_blk] = bitsv;
}

// the kept, valid candidates of an image in score order by counting (their keys are distinct); the first `cap` of them become the
// rois.  grid (ceil(Ktot / 64), B), 512 threads; keepl: [B * levels][kLevelCap] flags of the per-level scans
__global__ __launch_bounds__(512) void rank_compact_proposals_kernel(const RpnLevels lv, const float* __restrict__ cbox,
                                                                     const unsigned long long* __restrict__ ckey,
                                                                     const unsigned char* __restrict__ keepl, int Ktot, int Kpad, int cap,
                                                                     float* __restrict__ rois, int* __restrict__ levels,
                                                                     float* __restrict__ logits, int* __restrict__ count,
                                                                     int* __restrict__ n_keys) {
  constexpr int NW = 8;
  __shared__ __attribute__((aligned(16))) unsigned long long sk[8192];
  __shared__ int part[NW][64];
  __shared__ int nz[NW];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int nonzero = 0;
  for (int j = tid; j < Ktot; j += 64 * NW) {
    int l = 0;
    while (l + 1 < kLevels && j >= lv.koff[l + 1]) ++l;
    const bool kept = keepl[((size_t)b * kLevels + l) * kLevelCap + (j - lv.koff[l])] != 0;
    const unsigned long long v = kept ? ckey[(size_t)b * Kpad + j] : 0ull;
    sk[j] = v;
    nonzero += v != 0ull;
  }
  for (int o = 32; o > 0; o >>= 1) nonzero += __shfl_xor(nonzero, o);
  if (lane == 0) nz[wave] = nonzero;
  __syncthreads();
  int total = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) total += nz[w];
  const int c = min(total, cap);
  const int mine_i = blockIdx.x * 64 + lane;
  const unsigned long long mine = mine_i < Ktot ? sk[mine_i] : 0ull;
  const int per = ((Ktot + NW - 1) / NW + 1) & ~1, j0 = min(Ktot, wave * per), j1 = min(Ktot, j0 + per);
  int larger = 0, j = j0;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  for (; j + 16 <= j1; j += 16) {
    u64x2 kk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) kk[u] = *reinterpret_cast<const u64x2*>(&sk[j + 2 * u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) larger += (kk[u][0] > mine) + (kk[u][1] > mine);
  }
  for (; j < j1; ++j) larger += sk[j] > mine;
  part[wave][lane] = larger;
  __syncthreads();
  float* ro = rois + (size_t)b * cap * 5;
  int* lo = levels + (size_t)b * cap;
  if (wave == 0 && mine != 0ull) {
    int r = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) r += part[w][lane];
    if (r < cap) {
      const float* sbx = cbox + ((size_t)b * Ktot + mine_i) * 4;
      float* d = ro + (size_t)r * 5;
      d[0] = (float)b; d[1] = sbx[0]; d[2] = sbx[1]; d[3] = sbx[2]; d[4] = sbx[3];
      lo[r] = assign_level(sbx);
      logits[(size_t)b * cap + r] = ord_key_inv((unsigned)(mine >> 32));
    }
  }
  if (blockIdx.x == 0) {      // rows past the count: an empty roi on image 0
    for (int p = c + tid; p < cap; p += 64 * NW) {
      ro[p * 5 + 0] = 0.f; ro[p * 5 + 1] = 0.f; ro[p * 5 + 2] = 0.f; ro[p * 5 + 3] = 0.f; ro[p * 5 + 4] = 0.f;
      lo[p] = 0;
      logits[(size_t)b * cap + p] = 0.f;
    }
    if (tid == 0) { count[b] = c; n_keys[b] = 0; }
  }
}

// ---- FastRCNNOutputLayers.inference for one roi: softmax, per-class decode, clip, score threshold ----
__global__ __launch_bounds__(256) PEANUT_NO_PK_F32 void box_post_kernel(const float* __restrict__ cls_logits, const float* __restrict__ deltas,
                                                       const float* __restrict__ rois, const int* __restrict__ count, int B, int cap,
                                                       int K, int Kpad, float img_h, float img_w, float wx, float wy, float ww, float wh,
                                                       float score_thresh, float* __restrict__ dbox, unsigned long long* __restrict__ dkey,
                                                       int* __restrict__ dcat, int* __restrict__ n_keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;      // roi index
  if (i >= B * cap) return;
  const int b = i / cap, p = i - b * cap;
  unsigned long long* keys = dkey + (size_t)b * Kpad;
  if (p >= count[b]) return;
  const float* lg = cls_logits + (size_t)i * (K + 1);
  float m = lg[0];
  for (int k = 1; k <= K; ++k) m = fmaxf(m, lg[k]);
  float e[32], s = 0.f;
  for (int k = 0; k <= K; ++k) { e[k] = expf(lg[k] - m); s += e[k]; }
  bool ok = true;
  for (int k = 0; k <= K; ++k) { e[k] = e[k] / s; ok = ok && isfinite(e[k]); }
  const float* r = rois + (size_t)i * 5;
  float boxes[32][4];
  for (int k = 0; k < K; ++k) {
    apply_deltas1(deltas + (size_t)i * 4 * K + 4 * k, r[1], r[2], r[3], r[4], wx, wy, ww, wh, boxes[k]);
    ok = ok && finite4(boxes[k]);
  }
  for (int k = 0; k < K; ++k) {
    const int c = p * K + k;
    clip4(boxes[k], img_h, img_w);
    float* ob = dbox + ((size_t)b * cap * K + c) * 4;
    ob[0] = boxes[k][0]; ob[1] = boxes[k][1]; ob[2] = boxes[k][2]; ob[3] = boxes[k][3];
    dcat[(size_t)b * cap * K + c] = k;
    // candidates above the threshold are appended (n_keys[b] zeroed by compact_proposals_kernel); the order of arrival
    // does not matter: the keys carry their candidate index and are unique, the sort that follows is total
    if (ok && e[k] > score_thresh)
      // (the builtin, not HIP's atomicAdd wrapper: this kernel is compiled without packed fp32, common.h, and the wrapper would stay a call)
      keys[__hip_atomic_fetch_add(n_keys + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = ((unsigned long long)ord_key(e[k]) << 32) | (0xffffffffu - (unsigned)c);
  }
}


// ---- the first `cap` kept detections of every image; detector_postprocess scaling / clipping / non-empty filter ----
__global__ __launch_bounds__(1024) void compact_dets_kernel(const float* __restrict__ sbox, const int* __restrict__ scat,
                                                            const float* __restrict__ sscore, const unsigned char* __restrict__ keep,
                                                            const int* __restrict__ nvalid, int Kcap, int cap, float sx, float sy,
                                                            float out_h, float out_w, float* __restrict__ box_in,
                                                            float* __restrict__ box_out, float* __restrict__ score, int* __restrict__ cls,
                                                            int* __restrict__ count) {
  __shared__ int wave_cnt[16], running;
  __shared__ int sel[1024];
  const int b = blockIdx.x;
  const int n = min(nvalid[b], Kcap);
  const unsigned char* kp = keep + (size_t)b * Kcap;
  // 1) the first `cap` survivors of the NMS, in score order
  const int c1 = ordered_compact(n, min(cap, 1024), [&](int r) { return kp[r] != 0; }, [&](int r, int pos) { sel[pos] = r; }, wave_cnt,
                                 &running);
  __syncthreads();
  // 2) scale to the output resolution, clip, drop empty boxes (keeps the order)
  const int c2 = ordered_compact(
      c1, cap,
      [&](int q) {
        const float* s = sbox + ((size_t)b * Kcap + sel[q]) * 4;
        float o[4] = {s[0] * sx, s[1] * sy, s[2] * sx, s[3] * sy};
        clip4(o, out_h, out_w);
        return (o[2] - o[0]) > 0.f && (o[3] - o[1]) > 0.f;
      },
      [&](int q, int pos) {
        const size_t src = (size_t)b * Kcap + sel[q], dst = (size_t)b * cap + pos;
        const float* s = sbox + src * 4;
        float o[4] = {s[0] * sx, s[1] * sy, s[2] * sx, s[3] * sy};
        clip4(o, out_h, out_w);
        for (int t = 0; t < 4; ++t) { box_in[dst * 4 + t] = s[t]; box_out[dst * 4 + t] = o[t]; }
        score[dst] = sscore[src];
        cls[dst] = scat[src];
      },
      wave_cnt, &running);
  if (threadIdx.x == 0) count[b] = c2;
}

struct Offsets64 { int off[65]; };

// ---- compact per-image detection lists -> one list (image-major): mask rois, levels, caller outputs ----
__global__ __launch_bounds__(256) void pack_dets_kernel(const float* __restrict__ box_in, const float* __restrict__ box_out,
                                                        const float* __restrict__ score, const int* __restrict__ cls, int cap, int B,
                                                        Offsets64 of, float* __restrict__ mrois, int* __restrict__ mlevels,
                                                        float* __restrict__ o_boxes, float* __restrict__ o_scores, int* __restrict__ o_cls) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= of.off[B]) return;
  int b = 0;
  while (i >= of.off[b + 1]) ++b;
  const size_t src = (size_t)b * cap + (i - of.off[b]);
  const float* s = box_in + src * 4;
  float* r = mrois + (size_t)i * 5;
  r[0] = (float)b; r[1] = s[0]; r[2] = s[1]; r[3] = s[2]; r[4] = s[3];
  mlevels[i] = assign_level(s);
  for (int t = 0; t < 4; ++t) o_boxes[(size_t)i * 4 + t] = box_out[src * 4 + t];
  o_scores[i] = score[src];
  o_cls[i] = cls[src];
}

// ---- mask_rcnn_inference: logits [n, P, 4P, K] (2x2 sub-pixels of the transposed conv along the width) ->
//      probabilities [n, 2P, 2P] of each instance's class ----
__global__ __launch_bounds__(256) void mask_prob_kernel(const float* __restrict__ logits, const int* __restrict__ cls, int n, int P, int K,
                                                        float* __restrict__ probs) {
  const int M = 2 * P;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * M * M) return;
  const int X = i % M, Y = (i / M) % M, inst = i / (M * M);
  const float v = logits[(((size_t)inst * P + (Y >> 1)) * (4 * P) + (X >> 1) * 4 + (Y & 1) * 2 + (X & 1)) * K + cls[inst]];
  probs[i] = 1.f / (1.f + expf(-v));
}

inline int launch_sort_keys(unsigned long long* keys, int Kpad, int B, int* nvalid, hipStream_t s) {
  switch (Kpad) {
    case 16384: hipLaunchKernelGGL(sort_keys_kernel<16384>, dim3(B), dim3(1024), 0, s, keys, nvalid); break;
    case 8192: hipLaunchKernelGGL(sort_keys_kernel<8192>, dim3(B), dim3(1024), 0, s, keys, nvalid); break;
    case 4096: hipLaunchKernelGGL(sort_keys_kernel<4096>, dim3(B), dim3(1024), 0, s, keys, nvalid); break;
    case 2048: hipLaunchKernelGGL(sort_keys_kernel<2048>, dim3(B), dim3(1024), 0, s, keys, nvalid); break;
    default: return fail(PEANUT_EINVAL, "rcnn: unsupported sort size");
  }
  return 0;
}

// Mask pasting (paste_masks_in_image, restated in rcnn_ops.hip: paste_masks_kernel) and the per-category accumulation
// of SemanticPredMaskRCNN.get_prediction (seg_accum.hip) in one pass over the output pixels: a pixel walks its image's
// detections in score order, evaluates the pasted mask bit of those that pass the score gates with the arithmetic of
// paste_masks_kernel -- skipping, as that arithmetic would, instances whose four bilinear taps all fall outside the
// M x M mask -- and adds it to the class channel.  The [n, H, W] instance masks (30 MB per frame) are never written.
struct SemGoal { int cat[64]; };
__global__ __launch_bounds__(256) void paste_accumulate_kernel(const float* __restrict__ mprobs, const float* __restrict__ boxes,
                                                               const float* __restrict__ scores, const int* __restrict__ classes,
                                                               const Offsets64 of, int M, int H, int W, float mask_thr, int n_cats,
                                                               float thr, float goal_thr, const SemGoal goal, float* __restrict__ out,
                                                               long long total) {
  const int ch = n_cats + 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const long long t = i / W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
    const int goal_cat = goal.cat[b];
    for (int j = of.off[b]; j < of.off[b + 1]; ++j) {
      const int cls = classes[j];
      if (cls < 0 || cls >= n_cats) continue;
      const float sc = scores[j];
      if (sc < thr) continue;
      if (cls == goal_cat && sc < goal_thr) continue;
      const float* bx = boxes + (size_t)j * 4;
      const float v = paste_value(mprobs + (size_t)j * M * M, M, bx, x, y);       // (0 when all four taps fall outside the mask)
      const float bit = v >= mask_thr ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c)
        if (c == cls) acc[c] += bit;
    }
    for (int c = 0; c < ch; ++c) out[(size_t)i * ch + c] = c < 32 ? acc[c] : 0.f;
  }
}

inline int launch_sort_keys_counted(unsigned long long* keys, int Kpad, int B, const int* count, int* nvalid, hipStream_t s) {
  switch (Kpad) {
    case 16384: hipLaunchKernelGGL(sort_keys_counted_kernel<16384>, dim3(B), dim3(1024), 0, s, keys, count, nvalid); break;
    case 8192: hipLaunchKernelGGL(sort_keys_counted_kernel<8192>, dim3(B), dim3(1024), 0, s, keys, count, nvalid); break;
    case 4096: hipLaunchKernelGGL(sort_keys_counted_kernel<4096>, dim3(B), dim3(1024), 0, s, keys, count, nvalid); break;
    case 2048: hipLaunchKernelGGL(sort_keys_counted_kernel<2048>, dim3(B), dim3(1024), 0, s, keys, count, nvalid); break;
    default: return fail(PEANUT_EINVAL, "rcnn: unsupported sort size");
  }
  return 0;
}

inline unsigned blocks_for(long long n, int per = 256) { return (unsigned)std::max<long long>(1, (n + per - 1) / per); }
inline int pow2_at_least(int n) { int p = 2048; while (p < n) p <<= 1; return p; }

}  // namespace
}  // namespace peanut

using namespace peanut;

struct peanut_rcnn::PostBufs {
  DevBuf pyr[5], obj_all, dl_all;   // objectness / deltas of the five levels in ONE buffer each, level after level (the fused RPN chain needs that)
  DevBuf keepl, lvl_count;          // level-wise NMS: keep flags [B * levels][kLevelCap], candidates per (image, level)
  int lvl_count_B = 0, lvl_count_k[kLevels] = {0};
  DevBuf sel_idx, sel_score, cand, cbox, ckey, ckey_sorted, ccat, sbox, scat, sscore, keep, nms_ws, nvalid;
  DevBuf rois, roi_level, roi_logit, prop_count;
  DevBuf x7, f1, f2, cls, bbox;
  DevBuf dbox, dkey, dcat, dsbox, dscat, dsscore, dkeep, dnvalid, dkeys;
  DevBuf det_in, det_out, det_score, det_cls, det_count;
  DevBuf mrois, mlevel, mx0, mx1, mdeconv, mlogits, mprobs, splitk, wino_v, wino_m;
  DevBuf range_flag;   // fp16x3: set when a checked stage output is not finite (see flag_nonfinite)
  // stage timing (peanut_rcnn_set_stage_timing): one event per stage boundary of the last call, and what each stage had to do
  static constexpr int kStages = 9;
  bool timing = false;
  hipEvent_t ev[kStages + 1] = {};
  bool ev_made = false;
  int ev_count = 0;                    // boundaries recorded by the last call (a call without detections ends early)
  double work[kStages] = {};           // FLOPs (mfma stages) or algorithmic bytes (hbm stages) of the last call
  ~PostBufs() {
    if (ev_made)
      for (auto& e : ev) (void)hipEventDestroy(e);
  }
};

namespace {

// fp16x3 detector: an activation outside fp16's range (Winograd-domain values included: B^T d B reaches ~100 x the
// activations) makes the emulated layer's output inf - inf = NaN, and the selection kernels drop non-finite boxes and scores --
// an overflow would come out as an image WITHOUT detections.  The stage outputs every later result depends on (RPN
// objectness of all levels, class scores / box deltas, mask logits) are therefore scanned, and the call fails with
// PEANUT_ERANGE instead (the prediction model's check_range does the same on its logits).
__global__ __launch_bounds__(256) void flag_nonfinite_kernel(const float* __restrict__ p, size_t n, int* flag) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = p[i];
    bad |= !(fabsf(v) <= 3.402823466e38f);      // false for inf and NaN
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}
void flag_nonfinite(const float* p, size_t n, int* flag, hipStream_t s) {
  if (n == 0) return;
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(flag_nonfinite_kernel, dim3(blocks), dim3(256), 0, s, p, n, flag);
}

int conv_on(const ConvLayer* L, const float* x, float* y, int B, int H, int W, float* splitk, float* wv, float* wm, hipStream_t s) {
  ConvArgs a{};
  a.x = x; a.y = y; a.B = B; a.H = H; a.W = W; a.c1 = L->d.cin; a.c2 = 0;
  a.Ho = conv_out_dim(H, L->d.kh, L->d.stride, L->d.pad, L->d.dil);
  a.Wo = conv_out_dim(W, L->d.kw, L->d.stride, L->d.pad, L->d.dil);
  a.ws = splitk; a.ws_floats = kSplitKScratchFloats;
  return launch_conv_layer(*L, a, wv, wm, s);
}

int add_head_conv(peanut_rcnn* h, const std::string& name, const float* w_oihw, const float* bias, int cout, int cin, int k, int pad,
                  int relu, bool allow_wino, ConvLayer** out) {
  auto L = std::make_unique<ConvLayer>();
  L->name = name;
  std::vector<float> shift(bias, bias + cout);
  int rc = upload_conv(*L, w_oihw, nullptr, shift.data(), cout, cin, cin, k, k, 1, pad, 1, relu, h->cfg.precision);
  if (rc) return rc;
  if (allow_wino && h->cfg.conv_algo == PEANUT_ALGO_AUTO && wino_eligible(cin, cout, k, k, 1, pad, 1, h->cfg.precision) &&
      (rc = upload_wino(*L, w_oihw, cout, cin, cin, h->cfg.precision)))
    return rc;
  *out = L.get();
  h->convs.push_back(std::move(L));
  return 0;
}

}  // namespace

// called by peanut_rcnn_create when the state dict carries roi_heads.* (rcnn_api.hip)
int peanut_rcnn_build_heads(peanut_rcnn* h, const TensorMap& tm) {
  const peanut_rcnn_cfg& c = h->cfg;
  const int P = c.box_pooler_resolution, F = c.fpn_out, K = c.num_classes, fc = c.fc_dim, mc = c.mask_conv_dim;
  if (K < 1 || K > 31 || P < 1 || c.mask_pooler_resolution < 1 || fc % 32 || mc % 32 || c.num_anchors > kMaxAnchors ||
      c.rpn_pre_nms_topk < 1 || c.rpn_pre_nms_topk > 1024 || c.rpn_post_nms_topk < 1 || c.detections_per_image < 1 ||
      c.detections_per_image > 1024 || (long long)c.rpn_post_nms_topk * K > 16384)
    return fail(PEANUT_EINVAL, "rcnn: unsupported ROI-head configuration");
  int rc = 0;
  auto get = [&](const std::string& k, std::initializer_list<int64_t> shp) -> const peanut_tensor* {
    int64_t sh[4] = {0, 0, 0, 0};
    int nd = 0;
    for (int64_t v : shp) sh[nd++] = v;
    return tm.get(k, nd, sh, &rc);
  };
  // fc1 consumes the ROIAlign output flattened NHWC ((y*P+x)*C + c); detectron2 flattens NCHW (c*P*P + y*P + x)
  const peanut_tensor *w1 = get("roi_heads.box_head.fc1.weight", {fc, (int64_t)F * P * P}), *b1 = get("roi_heads.box_head.fc1.bias", {fc});
  const peanut_tensor *w2 = get("roi_heads.box_head.fc2.weight", {fc, fc}), *b2 = get("roi_heads.box_head.fc2.bias", {fc});
  const peanut_tensor *wc = get("roi_heads.box_predictor.cls_score.weight", {K + 1, fc}), *bc = get("roi_heads.box_predictor.cls_score.bias", {K + 1});
  const peanut_tensor *wb = get("roi_heads.box_predictor.bbox_pred.weight", {4 * K, fc}), *bb = get("roi_heads.box_predictor.bbox_pred.bias", {4 * K});
  if (!w1 || !b1 || !w2 || !b2 || !wc || !bc || !wb || !bb) return rc;
  {
    std::vector<float> perm((size_t)fc * F * P * P);
    for (int f = 0; f < fc; ++f)
      for (int ch = 0; ch < F; ++ch)
        for (int y = 0; y < P; ++y)
          for (int x = 0; x < P; ++x)
            perm[(size_t)f * F * P * P + (size_t)(y * P + x) * F + ch] = w1->data[(size_t)f * F * P * P + (size_t)ch * P * P + y * P + x];
    if ((rc = add_head_conv(h, "roi_heads.box_head.fc1", perm.data(), b1->data, fc, F * P * P, 1, 0, 1, false, &h->fc1))) return rc;
  }
  if ((rc = add_head_conv(h, "roi_heads.box_head.fc2", w2->data, b2->data, fc, fc, 1, 0, 1, false, &h->fc2))) return rc;
  if ((rc = add_head_conv(h, "roi_heads.box_predictor.cls_score", wc->data, bc->data, K + 1, fc, 1, 0, 0, false, &h->cls_score))) return rc;
  if ((rc = add_head_conv(h, "roi_heads.box_predictor.bbox_pred", wb->data, bb->data, 4 * K, fc, 1, 0, 0, false, &h->bbox_pred))) return rc;
  int cin = F;
  for (int i = 0; i < c.num_mask_convs; ++i) {
    const std::string n = "roi_heads.mask_head.mask_fcn" + std::to_string(i + 1);
    const peanut_tensor *w = get(n + ".weight", {mc, cin, 3, 3}), *b = get(n + ".bias", {mc});
    if (!w || !b) return rc;
    ConvLayer* L = nullptr;
    if ((rc = add_head_conv(h, n, w->data, b->data, mc, cin, 3, 1, 1, true, &L))) return rc;
    h->mask_fcn.push_back(L);
    cin = mc;
  }
  const peanut_tensor *wd = get("roi_heads.mask_head.deconv.weight", {mc, mc, 2, 2}), *bd = get("roi_heads.mask_head.deconv.bias", {mc});
  const peanut_tensor *wp = get("roi_heads.mask_head.predictor.weight", {K, mc, 1, 1}), *bp = get("roi_heads.mask_head.predictor.bias", {K});
  if (!wd || !bd || !wp || !bp) return rc;
  {
    // ConvTranspose2d(k = 2, s = 2) = four 1x1 convs, one per output sub-pixel (dy, dx): rows (dy*2+dx)*C + n; the
    // ConvTranspose2d weight is [in c][out n][dy][dx]
    std::vector<float> w4((size_t)4 * mc * mc), b4((size_t)4 * mc);
    for (int dy = 0; dy < 2; ++dy)
      for (int dx = 0; dx < 2; ++dx)
        for (int n = 0; n < mc; ++n) {
          b4[(size_t)(dy * 2 + dx) * mc + n] = bd->data[n];
          for (int ch = 0; ch < mc; ++ch)
            w4[((size_t)(dy * 2 + dx) * mc + n) * mc + ch] = wd->data[(((size_t)ch * mc + n) * 2 + dy) * 2 + dx];
        }
    if ((rc = add_head_conv(h, "roi_heads.mask_head.deconv", w4.data(), b4.data(), 4 * mc, mc, 1, 0, 1, false, &h->deconv))) return rc;
  }
  if ((rc = add_head_conv(h, "roi_heads.mask_head.predictor", wp->data, bp->data, K, mc, 1, 0, 0, false, &h->mask_pred))) return rc;
  h->has_heads = true;
  return 0;
}

namespace {
// what SemanticPredMaskRCNN.get_prediction makes of the instances (segmentation.py:47-60), or nothing (out == null)
struct SemanticOut {
  float* out = nullptr;          // [B,H,W,n_cats+1]
  int n_cats = 0;
  float thr = 0.f, goal_thr = 0.f;
  const int32_t* goal_cat = nullptr;   // host [B], -1 = none; null = none
};
}  // namespace

static int rcnn_inference_impl(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, int* n_det_host, float* boxes,
                               float* scores, int32_t* classes, uint8_t* masks, const SemanticOut& sem, void* stream) {
  if (!h || !img_bgr || !n_det_host || !boxes || !scores || !classes) return fail(PEANUT_EINVAL, "peanut_rcnn_inference: null argument");
  OptionScope option_scope(&h->opts);
  if (!h->has_heads) return fail(PEANUT_EINVAL, "peanut_rcnn_inference: the handle was created without roi_heads.* tensors");
  if (B < 1 || B > 64) return fail(PEANUT_EINVAL, "peanut_rcnn_inference: 1 <= B <= 64");
  const peanut_rcnn_cfg& c = h->cfg;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  int rs[2], pd[2], lhw[10];
  if ((rc = peanut_rcnn_plan(h, B, H, W, rs, pd, lhw, nullptr, nullptr))) return rc;
  const int nh = rs[0], nw = rs[1];
  if (!h->post) h->post = std::make_shared<peanut_rcnn::PostBufs>();
  peanut_rcnn::PostBufs& pb = *h->post;
  const int A = c.num_anchors, F = c.fpn_out, K = c.num_classes;

  // ---- geometry of the selection stages ----
  RpnLevels lv{};
  lv.A = A;
  int Ktot = 0;
  for (int l = 0; l < kLevels; ++l) {
    lv.h[l] = lhw[2 * l]; lv.w[l] = lhw[2 * l + 1];
    lv.n[l] = lv.h[l] * lv.w[l] * A;
    lv.k[l] = std::min(lv.n[l], c.rpn_pre_nms_topk);
    lv.koff[l] = Ktot;
    Ktot += lv.k[l];
    lv.stride[l] = (float)(4 << l);
    for (int a = 0; a < A; ++a) {   // DefaultAnchorGenerator.generate_cell_anchors (double arithmetic, then float32)
      const double size = c.anchor_sizes[l], r = c.aspect_ratios[a];
      const double w = sqrt(size * size / r), hh = r * w;
      lv.cell[l][a][0] = (float)(-w / 2.0); lv.cell[l][a][1] = (float)(-hh / 2.0);
      lv.cell[l][a][2] = (float)(w / 2.0); lv.cell[l][a][3] = (float)(hh / 2.0);
    }
  }
  lv.koff[kLevels] = Ktot;
  const int Kpad = pow2_at_least(Ktot);
  if (Kpad > 16384) return fail(PEANUT_EINVAL, "peanut_rcnn_inference: too many proposal candidates per image");
  const int cap = c.rpn_post_nms_topk, N = B * cap;
  const int D = c.detections_per_image;
  const int Kc = cap * K, Kcpad = pow2_at_least(Kc);

  // ---- buffers ----
  for (int l = 0; l < kLevels; ++l) {
    if ((rc = pb.pyr[l].ensure((size_t)B * lv.h[l] * lv.w[l] * F * 4)))
      return rc;
  }
  const int words = (Ktot + 63) / 64, dwords = (Kc + 63) / 64;
  if ((rc = pb.sel_idx.ensure((size_t)B * Ktot * 4)) || (rc = pb.sel_score.ensure((size_t)B * Ktot * 4)) ||
      (rc = pb.cbox.ensure((size_t)B * Ktot * 16)) || (rc = pb.ckey.ensure((size_t)B * Kpad * 8)) || (rc = pb.ckey_sorted.ensure((size_t)B * Kpad * 8)) || (rc = pb.ccat.ensure((size_t)B * Ktot * 4)) ||
      (rc = pb.sbox.ensure((size_t)B * Ktot * 16)) || (rc = pb.scat.ensure((size_t)B * Ktot * 4)) || (rc = pb.sscore.ensure((size_t)B * Ktot * 4)) ||
      (rc = pb.keep.ensure((size_t)B * Ktot)) || (rc = pb.nvalid.ensure((size_t)B * 4)) ||
      (rc = pb.nms_ws.ensure(std::max((size_t)B * Ktot * words, (size_t)B * Kc * dwords) * 8)) ||
      (rc = pb.rois.ensure((size_t)N * 20)) || (rc = pb.roi_level.ensure((size_t)N * 4)) || (rc = pb.roi_logit.ensure((size_t)N * 4)) ||
      (rc = pb.prop_count.ensure((size_t)B * 4)) ||
      (rc = pb.x7.ensure((size_t)N * c.box_pooler_resolution * c.box_pooler_resolution * F * 4)) || (rc = pb.f1.ensure((size_t)N * c.fc_dim * 4)) ||
      (rc = pb.f2.ensure((size_t)N * c.fc_dim * 4)) || (rc = pb.cls.ensure((size_t)N * (K + 1) * 4)) || (rc = pb.bbox.ensure((size_t)N * 4 * K * 4)) ||
      (rc = pb.dbox.ensure((size_t)B * Kc * 16)) || (rc = pb.dkey.ensure((size_t)B * Kcpad * 8)) || (rc = pb.dcat.ensure((size_t)B * Kc * 4)) ||
      (rc = pb.dsbox.ensure((size_t)B * Kc * 16)) || (rc = pb.dscat.ensure((size_t)B * Kc * 4)) || (rc = pb.dsscore.ensure((size_t)B * Kc * 4)) ||
      (rc = pb.dkeep.ensure((size_t)B * Kc)) || (rc = pb.dnvalid.ensure((size_t)B * 4)) || (rc = pb.dkeys.ensure((size_t)B * 4)) ||
      (rc = pb.det_in.ensure((size_t)B * D * 16)) || (rc = pb.det_out.ensure((size_t)B * D * 16)) || (rc = pb.det_score.ensure((size_t)B * D * 4)) ||
      (rc = pb.det_cls.ensure((size_t)B * D * 4)) || (rc = pb.det_count.ensure((size_t)B * 4)) ||
      (rc = pb.splitk.ensure(kSplitKScratchFloats * sizeof(float))))
    return rc;

  // stage timing: an event on the call's stream at every stage boundary (bench.py's roofline of stage 1's back half)
  if (pb.timing && !pb.ev_made) {
    for (auto& e : pb.ev) PEANUT_HIP_CHECK(hipEventCreate(&e));
    pb.ev_made = true;
  }
  pb.ev_count = 0;
  auto mark = [&]() {
    if (pb.timing && pb.ev_count <= peanut_rcnn::PostBufs::kStages) (void)hipEventRecord(pb.ev[pb.ev_count++], s);
  };
  for (double& w : pb.work) w = 0.0;
  mark();
  // ---- front end: pyramid p2..p6, objectness, anchor deltas ----
  float *pyr[5], *obj[5], *dl[5];
  {
    size_t anchors = 0;
    for (int l = 0; l < kLevels; ++l) anchors += (size_t)B * lv.n[l];
    if ((rc = pb.obj_all.ensure(anchors * 4)) || (rc = pb.dl_all.ensure(anchors * 16))) return rc;
    size_t off = 0;
    for (int l = 0; l < kLevels; ++l) {
      pyr[l] = (float*)pb.pyr[l].p;
      obj[l] = (float*)pb.obj_all.p + off;
      dl[l] = (float*)pb.dl_all.p + off * 4;
      off += (size_t)B * lv.n[l];
    }
  }
  if ((rc = peanut_rcnn_forward_front(h, img_bgr, B, H, W, pyr, obj, dl, stream))) return rc;
  for (int l = 0; l < kLevels; ++l) { lv.obj[l] = obj[l]; lv.delta[l] = dl[l]; }
  const bool range_check = c.precision == PEANUT_PREC_FP16X3;
  if (range_check) {
    if ((rc = pb.range_flag.ensure(4))) return rc;
    PEANUT_HIP_CHECK(hipMemsetAsync(pb.range_flag.p, 0, 4, s));
    for (int l = 0; l < kLevels; ++l) flag_nonfinite(obj[l], (size_t)B * lv.n[l], (int*)pb.range_flag.p, s);
  }

  {
    double fl = 0.0;
    (void)peanut_rcnn_plan(h, B, H, W, nullptr, nullptr, nullptr, nullptr, &fl);
    pb.work[0] = fl * B;                                                          // front end: direct-form (NOMINAL) conv FLOPs
    long long anchors = 0;
    for (int l = 0; l < kLevels; ++l) anchors += lv.n[l];
    pb.work[1] = (double)B * anchors * 20.0;                                      // RPN selection: objectness + deltas read once
  }
  mark();
  // ---- RPN: per-level top-k, decode, per-image sort, NMS (per level), post-NMS top-k ----
  {
    RpnSlices sl{};
    const int target = (int)std::max<long long>(opt(OPT_RCNN_TOPK_SLICE), 0);      // logits per workgroup; 0: one workgroup per level
    int max_c = 0;
    if (target > 0 && c.rpn_pre_nms_topk <= 1024) {
      for (int l = 0; l < kLevels; ++l) {
        const int room = kMaxSlices - sl.n_items - (kLevels - 1 - l);      // every later level needs at least one workgroup
        const int ns = std::min(std::min(kMaxSlicesPerLevel, room), std::max(1, (lv.n[l] + target - 1) / target));
        const int per = (lv.n[l] + ns - 1) / ns;
        sl.lvl_coff[l] = sl.ctot;
        for (int q = 0; q < ns; ++q) {
          const int st = q * per, cn = std::min(per, lv.n[l] - st);
          if (cn <= 0) break;
          const int it = sl.n_items++;
          sl.lvl[it] = l; sl.start[it] = st; sl.cnt[it] = cn; sl.ks[it] = std::min(cn, lv.k[l]); sl.coff[it] = sl.ctot;
          sl.ctot += sl.ks[it];
        }
        sl.lvl_cnt[l] = sl.ctot - sl.lvl_coff[l];
        max_c = std::max(max_c, sl.lvl_cnt[l]);
      }
    }
    if (sl.n_items > 0) {
      if ((rc = pb.cand.ensure((size_t)B * sl.ctot * 8))) return rc;
      hipLaunchKernelGGL(rpn_topk_kernel<true>, dim3(sl.n_items, B), dim3(1024), 0, s, lv, sl, Ktot, (int*)pb.sel_idx.p, (float*)pb.sel_score.p,
                         (unsigned long long*)pb.cand.p);
      hipLaunchKernelGGL(rpn_rank_select_kernel, dim3((max_c + 63) / 64, kLevels, B), dim3(512), 0, s, lv, sl, Ktot,
                         (const unsigned long long*)pb.cand.p, (int*)pb.sel_idx.p, (float*)pb.sel_score.p);
    } else {
      hipLaunchKernelGGL(rpn_topk_kernel<false>, dim3(kLevels, B), dim3(1024), 0, s, lv, sl, Ktot, (int*)pb.sel_idx.p, (float*)pb.sel_score.p,
                         (unsigned long long*)nullptr);
    }
  }
  hipLaunchKernelGGL(rpn_decode_kernel, dim3(blocks_for((long long)B * Kpad)), dim3(256), 0, s, lv, B, Ktot, Kpad, (const int*)pb.sel_idx.p,
                     (const float*)pb.sel_score.p, (float)nh, (float)nw, c.rpn_bbox_weights[0], c.rpn_bbox_weights[1], c.rpn_bbox_weights[2],
                     c.rpn_bbox_weights[3], (float*)pb.cbox.p, (unsigned long long*)pb.ckey.p, (int*)pb.ccat.p);
  if (opt(OPT_RCNN_NMS_LEVELS) != 0 && Ktot <= 8192 && c.rpn_pre_nms_topk <= kLevelCap) {
    // level by level: five suppression blocks and five scans per image side by side, the post-NMS top-k by counting
    constexpr int LW = kLevelCap / 64;
    const int segs = B * kLevels;
    if ((rc = pb.nms_ws.ensure((size_t)segs * kLevelCap * LW * 8)) || (rc = pb.keepl.ensure((size_t)segs * kLevelCap)) ||
        (rc = pb.lvl_count.ensure((size_t)segs * 4)))
      return rc;
    if (pb.lvl_count_B != B || memcmp(pb.lvl_count_k, lv.k, sizeof(lv.k)) != 0) {      // (per plan shape: a synchronous upload, once)
      std::vector<int> hc((size_t)segs);
      for (int q = 0; q < segs; ++q) hc[q] = lv.k[q % kLevels];
      PEANUT_HIP_CHECK(hipMemcpy(pb.lvl_count.p, hc.data(), hc.size() * sizeof(int), hipMemcpyHostToDevice));
      pb.lvl_count_B = B;
      memcpy(pb.lvl_count_k, lv.k, sizeof(lv.k));
    }
    hipLaunchKernelGGL(nms_mask_levels_kernel, dim3(LW, LW, segs), dim3(64), 0, s, lv, (const float*)pb.cbox.p,
                       (const unsigned long long*)pb.ckey.p, Ktot, Kpad, c.rpn_nms_thresh, (unsigned long long*)pb.nms_ws.p);
    launch_nms_scan((const unsigned long long*)pb.nms_ws.p, kLevelCap, LW, (const int*)pb.lvl_count.p, (unsigned char*)pb.keepl.p,
                    kLevelCap + 1, segs, s);
    hipLaunchKernelGGL(rank_compact_proposals_kernel, dim3((Ktot + 63) / 64, B), dim3(512), 0, s, lv, (const float*)pb.cbox.p,
                       (const unsigned long long*)pb.ckey.p, (const unsigned char*)pb.keepl.p, Ktot, Kpad, cap, (float*)pb.rois.p,
                       (int*)pb.roi_level.p, (float*)pb.roi_logit.p, (int*)pb.prop_count.p, (int*)pb.dkeys.p);
  } else {
  const unsigned long long* sorted_keys = (const unsigned long long*)pb.ckey.p;
  if (opt(OPT_RCNN_RANK_SORT) != 0 && Ktot <= 8192) {
    hipLaunchKernelGGL(rank_sort_keys_kernel, dim3((Ktot + 63) / 64, B), dim3(512), 0, s, (const unsigned long long*)pb.ckey.p, Kpad, Ktot,
                       (unsigned long long*)pb.ckey_sorted.p, (int*)pb.nvalid.p);
    sorted_keys = (const unsigned long long*)pb.ckey_sorted.p;
  } else if ((rc = launch_sort_keys((unsigned long long*)pb.ckey.p, Kpad, B, (int*)pb.nvalid.p, s))) {
    return rc;
  }
  hipLaunchKernelGGL(gather_sorted_kernel, dim3(blocks_for((long long)B * Ktot)), dim3(256), 0, s, sorted_keys,
                     (const int*)pb.nvalid.p, Kpad, Ktot, B, (const float*)pb.cbox.p, (const int*)pb.ccat.p, (float*)pb.sbox.p, (int*)pb.scat.p, (float*)pb.sscore.p);
  hipLaunchKernelGGL(nms_mask_dev_kernel, dim3(words, words, B), dim3(64), 0, s, (const float*)pb.sbox.p, (const int*)pb.scat.p, Ktot, words,
                     (const int*)pb.nvalid.p, c.rpn_nms_thresh, (unsigned long long*)pb.nms_ws.p);
  launch_nms_scan((const unsigned long long*)pb.nms_ws.p, Ktot, words, (const int*)pb.nvalid.p, (unsigned char*)pb.keep.p, cap, B, s);
  hipLaunchKernelGGL(compact_proposals_kernel, dim3(B), dim3(1024), 0, s, (const float*)pb.sbox.p, (const float*)pb.sscore.p,
                     (const unsigned char*)pb.keep.p, (const int*)pb.nvalid.p, Ktot, cap, (float*)pb.rois.p, (int*)pb.roi_level.p,
                     (float*)pb.roi_logit.p, (int*)pb.prop_count.p, (int*)pb.dkeys.p);

  }

  mark();
  // ---- box head: ROIAlignV2 7x7 over p2..p5, two FC layers, class scores and box deltas ----
  const float* feats[4] = {pyr[0], pyr[1], pyr[2], pyr[3]};
  const int fhw[8] = {lv.h[0], lv.w[0], lv.h[1], lv.w[1], lv.h[2], lv.w[2], lv.h[3], lv.w[3]};
  const float fsc[4] = {1.f / 4, 1.f / 8, 1.f / 16, 1.f / 32};
  const int P = c.box_pooler_resolution;
  if ((rc = peanut_roi_align(feats, fhw, fsc, 4, F, (const float*)pb.rois.p, (const int*)pb.roi_level.p, N, P, 0, 1, (float*)pb.x7.p, stream))) return rc;
  {
    double pyr_bytes = 0.0;
    for (int l = 0; l < 4; ++l) pyr_bytes += (double)B * lv.h[l] * lv.w[l] * F * 4.0;
    pb.work[2] = pyr_bytes + (double)N * P * P * F * 4.0;                         // ROIAlign 7x7: the pyramid once + its output
    const double fc = c.fc_dim;
    pb.work[3] = 2.0 * N * ((double)P * P * F * fc + fc * fc + fc * (K + 1) + fc * 4.0 * K);   // box head: the four FC GEMMs
    pb.work[4] = (double)N * ((K + 1) + 4.0 * K) * 4.0 + (double)B * Kc * 24.0;   // class candidates, sort keys, boxes
  }
  mark();
  float* sk = (float*)pb.splitk.p;
  if ((rc = conv_on(h->fc1, (const float*)pb.x7.p, (float*)pb.f1.p, N, 1, 1, sk, nullptr, nullptr, s))) return rc;
  if ((rc = conv_on(h->fc2, (const float*)pb.f1.p, (float*)pb.f2.p, N, 1, 1, sk, nullptr, nullptr, s))) return rc;
  if ((rc = conv_on(h->cls_score, (const float*)pb.f2.p, (float*)pb.cls.p, N, 1, 1, sk, nullptr, nullptr, s))) return rc;
  if ((rc = conv_on(h->bbox_pred, (const float*)pb.f2.p, (float*)pb.bbox.p, N, 1, 1, sk, nullptr, nullptr, s))) return rc;
  if (range_check) {
    flag_nonfinite((const float*)pb.cls.p, (size_t)N * (K + 1), (int*)pb.range_flag.p, s);
    flag_nonfinite((const float*)pb.bbox.p, (size_t)N * 4 * K, (int*)pb.range_flag.p, s);
  }

  mark();
  // ---- fast_rcnn_inference: class candidates, per-image sort, class-wise NMS, top detections ----
  hipLaunchKernelGGL(box_post_kernel, dim3(blocks_for(N)), dim3(256), 0, s, (const float*)pb.cls.p, (const float*)pb.bbox.p, (const float*)pb.rois.p,
                     (const int*)pb.prop_count.p, B, cap, K, Kcpad, (float)nh, (float)nw, c.roi_bbox_weights[0], c.roi_bbox_weights[1],
                     c.roi_bbox_weights[2], c.roi_bbox_weights[3], c.score_thresh_test, (float*)pb.dbox.p, (unsigned long long*)pb.dkey.p,
                     (int*)pb.dcat.p, (int*)pb.dkeys.p);
  if ((rc = launch_sort_keys_counted((unsigned long long*)pb.dkey.p, Kcpad, B, (const int*)pb.dkeys.p, (int*)pb.dnvalid.p, s))) return rc;
  hipLaunchKernelGGL(gather_sorted_kernel, dim3(blocks_for((long long)B * Kc)), dim3(256), 0, s, (const unsigned long long*)pb.dkey.p,
                     (const int*)pb.dnvalid.p, Kcpad, Kc, B,
                     (const float*)pb.dbox.p, (const int*)pb.dcat.p, (float*)pb.dsbox.p, (int*)pb.dscat.p, (float*)pb.dsscore.p);
  hipLaunchKernelGGL(nms_mask_dev_kernel, dim3(dwords, dwords, B), dim3(64), 0, s, (const float*)pb.dsbox.p, (const int*)pb.dscat.p, Kc, dwords,
                     (const int*)pb.dnvalid.p, c.nms_thresh_test, (unsigned long long*)pb.nms_ws.p);
  launch_nms_scan((const unsigned long long*)pb.nms_ws.p, Kc, dwords, (const int*)pb.dnvalid.p, (unsigned char*)pb.dkeep.p, std::min(D, 1024), B, s);
  // detector_postprocess scales boxes by (W / nw, H / nh): python doubles narrowed to float32 by the tensor product
  const float sx = (float)((double)W / (double)nw), sy = (float)((double)H / (double)nh);
  hipLaunchKernelGGL(compact_dets_kernel, dim3(B), dim3(1024), 0, s, (const float*)pb.dsbox.p, (const int*)pb.dscat.p, (const float*)pb.dsscore.p,
                     (const unsigned char*)pb.dkeep.p, (const int*)pb.dnvalid.p, Kc, D, sx, sy, (float)H, (float)W, (float*)pb.det_in.p,
                     (float*)pb.det_out.p, (float*)pb.det_score.p, (int*)pb.det_cls.p, (int*)pb.det_count.p);

  mark();
  // ---- the one host read: detections per image ----
  PEANUT_HIP_CHECK(hipMemcpyAsync(n_det_host, pb.det_count.p, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  int range_bad = 0;
  if (range_check) PEANUT_HIP_CHECK(hipMemcpyAsync(&range_bad, pb.range_flag.p, 4, hipMemcpyDeviceToHost, s));
  PEANUT_HIP_CHECK(hipStreamSynchronize(s));
  static const char* const kRangeMsg =
      "peanut_rcnn_inference: precision fp16x3 -- a value left fp16's range (|x| >= 65520 in an emulated layer; for the Winograd "
      "layers that is the transformed input B^T d B, up to ~100 x the activations); rerun the detector with bf16x6 or fp32";
  if (range_bad) {
    for (int b = 0; b < B; ++b) n_det_host[b] = 0;
    return fail(PEANUT_ERANGE, kRangeMsg);
  }
  Offsets64 of{};
  for (int b = 0; b < B; ++b) of.off[b + 1] = of.off[b] + n_det_host[b];
  const int n = of.off[B];
  if (n == 0) {
    if (sem.out) PEANUT_HIP_CHECK(hipMemsetAsync(sem.out, 0, (size_t)B * H * W * (sem.n_cats + 1) * sizeof(float), s));
    return 0;
  }
  const int Pm = c.mask_pooler_resolution, mc = c.mask_conv_dim;
  size_t vf = 0, mf = 0;
  for (const ConvLayer* L : h->mask_fcn)
    if (L->has_wino) { size_t v, m; wino_scratch_floats(*L, n, Pm, Pm, &v, &m); vf = std::max(vf, v); mf = std::max(mf, m); }
  const size_t act = (size_t)n * Pm * Pm * std::max(F, mc) * 4;
  if ((rc = pb.mrois.ensure((size_t)n * 20)) || (rc = pb.mlevel.ensure((size_t)n * 4)) || (rc = pb.mx0.ensure(act)) || (rc = pb.mx1.ensure(act)) ||
      (rc = pb.mdeconv.ensure((size_t)n * Pm * Pm * 4 * mc * 4)) || (rc = pb.mlogits.ensure((size_t)n * Pm * Pm * 4 * K * 4)) ||
      (rc = pb.mprobs.ensure((size_t)n * 4 * Pm * Pm * 4)) || (vf && (rc = pb.wino_v.ensure(vf * 4))) || (mf && (rc = pb.wino_m.ensure(mf * 4))))
    return rc;
  mark();
  {
    double pyr_bytes = 0.0;
    for (int l = 0; l < 4; ++l) pyr_bytes += (double)B * lv.h[l] * lv.w[l] * F * 4.0;
    pb.work[6] = pyr_bytes + (double)n * Pm * Pm * F * 4.0;                       // ROIAlign 14x14
    double fl = 0.0;
    int cin = F;
    for (const ConvLayer* L : h->mask_fcn) {
      // EXECUTED FLOPs: a layer that runs as Winograd F(m x m, 3 x 3) multiplies (m + 2)^2 positions per m x m output tile
      // instead of 9 taps per output pixel
      const ConvLayer* R = L->has_wino ? wino_pick_form(L, n, Pm, Pm) : L;
      if (R->has_wino) {
        const double tiles = (double)n * ((Pm + R->wino_m - 1) / R->wino_m) * ((Pm + R->wino_m - 1) / R->wino_m);
        fl += 2.0 * tiles * R->wino_np() * (double)cin * L->d.cout;
      } else {
        fl += 2.0 * n * Pm * Pm * (double)cin * L->d.cout * 9.0;
      }
      cin = L->d.cout;
    }
    fl += 2.0 * n * Pm * Pm * (double)cin * mc * 4.0;                             // 2x2 stride-2 transposed conv
    fl += 2.0 * n * 4.0 * Pm * Pm * (double)mc * K;                               // class logits
    pb.work[7] = fl;
    pb.work[8] = (double)n * 4.0 * Pm * Pm * (K + 1) * 4.0 + (sem.out ? (double)B * H * W * (sem.n_cats + 1) * 4.0 : 0.0) +
                 (masks ? (double)n * H * W : 0.0);                               // logits -> probabilities, paste / accumulate
  }
  hipLaunchKernelGGL(pack_dets_kernel, dim3(blocks_for(n)), dim3(256), 0, s, (const float*)pb.det_in.p, (const float*)pb.det_out.p,
                     (const float*)pb.det_score.p, (const int*)pb.det_cls.p, D, B, of, (float*)pb.mrois.p, (int*)pb.mlevel.p, boxes, scores, classes);
  if (!masks && !sem.out) {
    hipError_t e0 = hipGetLastError();
    return e0 == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("peanut_rcnn_inference: ") + hipGetErrorString(e0));
  }
  // ---- mask head: ROIAlignV2 14x14, conv stack, 2x2 transposed conv, class logits, sigmoid; paste at 0.5 ----
  if ((rc = peanut_roi_align(feats, fhw, fsc, 4, F, (const float*)pb.mrois.p, (const int*)pb.mlevel.p, n, Pm, 0, 1, (float*)pb.mx0.p, stream))) return rc;
  mark();
  float *cur = (float*)pb.mx0.p, *nxt = (float*)pb.mx1.p;
  for (const ConvLayer* L : h->mask_fcn) {
    if ((rc = conv_on(L, cur, nxt, n, Pm, Pm, sk, L->has_wino ? (float*)pb.wino_v.p : nullptr, L->has_wino ? (float*)pb.wino_m.p : nullptr, s))) return rc;
    std::swap(cur, nxt);
  }
  if ((rc = conv_on(h->deconv, cur, (float*)pb.mdeconv.p, n, Pm, Pm, sk, nullptr, nullptr, s))) return rc;
  if ((rc = conv_on(h->mask_pred, (const float*)pb.mdeconv.p, (float*)pb.mlogits.p, n, Pm, Pm * 4, sk, nullptr, nullptr, s))) return rc;
  mark();
  hipLaunchKernelGGL(mask_prob_kernel, dim3(blocks_for((long long)n * 4 * Pm * Pm)), dim3(256), 0, s, (const float*)pb.mlogits.p, (const int*)classes, n,
                     Pm, K, (float*)pb.mprobs.p);
  if (range_check) flag_nonfinite((const float*)pb.mlogits.p, (size_t)n * Pm * Pm * 4 * K, (int*)pb.range_flag.p, s);
  if (masks && (rc = peanut_paste_masks((const float*)pb.mprobs.p, boxes, n, 2 * Pm, H, W, c.mask_threshold, masks, stream))) return rc;
  if (sem.out) {
    SemGoal goal{};
    for (int b = 0; b < B; ++b) goal.cat[b] = sem.goal_cat ? sem.goal_cat[b] : -1;
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(paste_accumulate_kernel, dim3(blocks_for(total)), dim3(256), 0, s, (const float*)pb.mprobs.p, (const float*)boxes,
                       (const float*)scores, (const int*)classes, of, 2 * Pm, H, W, c.mask_threshold, sem.n_cats, sem.thr, sem.goal_thr, goal,
                       sem.out, total);
  }
  mark();
  if (range_check) {      // the mask head's own overflow: one more 4-byte read (fp16x3 only)
    PEANUT_HIP_CHECK(hipMemcpyAsync(&range_bad, pb.range_flag.p, 4, hipMemcpyDeviceToHost, s));
    PEANUT_HIP_CHECK(hipStreamSynchronize(s));
    if (range_bad) return fail(PEANUT_ERANGE, kRangeMsg);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("peanut_rcnn_inference: ") + hipGetErrorString(e));
}

extern "C" int peanut_rcnn_inference(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, int* n_det_host, float* boxes,
                                     float* scores, int32_t* classes, uint8_t* masks, void* stream) {
  return rcnn_inference_impl(h, img_bgr, B, H, W, n_det_host, boxes, scores, classes, masks, SemanticOut{}, stream);
}

extern "C" int peanut_rcnn_semantic(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, int n_cats, float sem_pred_prob_thr,
                                    float goal_thr, const int32_t* goal_cat_host, float* semantic, int* n_det_host, float* boxes,
                                    float* scores, int32_t* classes, uint8_t* masks, void* stream) {
  if (!semantic) return fail(PEANUT_EINVAL, "peanut_rcnn_semantic: null output");
  if (n_cats < 1 || n_cats > 31) return fail(PEANUT_EINVAL, "peanut_rcnn_semantic: 1 <= n_cats <= 31 required");
  SemanticOut so;
  so.out = semantic; so.n_cats = n_cats; so.thr = sem_pred_prob_thr; so.goal_thr = goal_thr; so.goal_cat = goal_cat_host;
  return rcnn_inference_impl(h, img_bgr, B, H, W, n_det_host, boxes, scores, classes, masks, so, stream);
}

extern "C" int peanut_rcnn_set_stage_timing(peanut_rcnn_t* h, int on) {
  if (!h) return fail(PEANUT_EINVAL, "peanut_rcnn_set_stage_timing: null handle");
  if (!h->post) h->post = std::make_shared<peanut_rcnn::PostBufs>();
  h->post->timing = on != 0;
  h->post->ev_count = 0;
  return 0;
}

extern "C" int peanut_rcnn_stage_times(peanut_rcnn_t* h, int max_stages, const char** names, const char** bounds, double* ms, double* work) {
  if (!h || !h->post || !h->post->timing) return fail(PEANUT_EINVAL, "peanut_rcnn_stage_times: stage timing is not enabled on this handle");
  peanut_rcnn::PostBufs& pb = *h->post;
  static const char* const kNames[peanut_rcnn::PostBufs::kStages] = {
      "front_end", "rpn_selection", "roi_align_7x7", "box_head_fc", "box_postprocess_nms", "host_read_detection_counts",
      "roi_align_14x14", "mask_head_convs", "mask_probabilities_paste"};
  static const char* const kBounds[peanut_rcnn::PostBufs::kStages] = {"mfma", "hbm", "hbm", "mfma", "hbm", "host", "hbm", "mfma", "hbm"};
  const int n = pb.ev_count - 1;
  if (n < 1) return fail(PEANUT_EINVAL, "peanut_rcnn_stage_times: no timed call has run");
  if (hipEventSynchronize(pb.ev[n]) != hipSuccess) return fail(PEANUT_EHIP, "peanut_rcnn_stage_times: event synchronise failed");
  for (int i = 0; i < n && i < max_stages; ++i) {
    float t = 0.f;
    (void)hipEventElapsedTime(&t, pb.ev[i], pb.ev[i + 1]);
    if (names) names[i] = kNames[i];
    if (bounds) bounds[i] = kBounds[i];
    if (ms) ms[i] = t;
    if (work) work[i] = pb.work[i];
  }
  return n;
}

// test / bisect hook: device pointers of the stage outputs of the last peanut_rcnn_inference call
extern "C" int peanut_rcnn_debug_stage(peanut_rcnn_t* h, const char* name, const void** dev, size_t* bytes) {
  if (!h || !h->post || !name || !dev) return fail(PEANUT_EINVAL, "peanut_rcnn_debug_stage: no inference has run");
  peanut_rcnn::PostBufs& pb = *h->post;
  const struct { const char* n; DevBuf* b; } tab[] = {
      {"rois", &pb.rois}, {"roi_level", &pb.roi_level}, {"roi_logit", &pb.roi_logit}, {"prop_count", &pb.prop_count}, {"cls", &pb.cls},
      {"bbox", &pb.bbox}, {"det_in", &pb.det_in}, {"det_out", &pb.det_out}, {"det_score", &pb.det_score}, {"det_cls", &pb.det_cls},
      {"det_count", &pb.det_count}, {"mprobs", &pb.mprobs}, {"sel_idx", &pb.sel_idx}, {"sel_score", &pb.sel_score}, {"nvalid", &pb.nvalid}};
  for (const auto& t : tab)
    if (!strcmp(t.n, name)) { *dev = t.b->p; if (bytes) *bytes = t.b->bytes; return 0; }
  return fail(PEANUT_EINVAL, std::string("peanut_rcnn_debug_stage: unknown stage ") + name);
}
