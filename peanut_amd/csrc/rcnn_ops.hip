// The three custom operators of detectron2's Mask R-CNN inference that are not convolutions -- the ones
// detectron2 itself implements natively in `detectron2._C` / a fused torch path:
//   roi_align    ROIAlign (aligned=True, adaptive sampling)      ROI_BOX_HEAD / ROI_MASK_HEAD.POOLER_TYPE ROIAlignV2
//   nms          greedy IoU suppression of score-sorted boxes    RPN.NMS_THRESH 0.7, ROI_HEADS.NMS_THRESH_TEST 0.5
//   paste_masks  paste_masks_in_image + threshold                (mask -> image bilinear resample, >= 0.5)
// as used by SemanticPredMaskRCNN.get_prediction through DefaultPredictor (nav/agent/utils/segmentation.py:45,
// yaml nav/agent/utils/COCO-InstSeg/mask_rcnn_R_101_cat9.yaml).  Algorithms restated from detectron2 v0.6 /
// torchvision's published operator definitions; parity is pinned against oracle/rcnn_ref.py only.
#include "../../include/peanut_hip.h"
#include "common.h"

namespace peanut {

// ---------------------------------------------------------------------------------------------------------
// ROIAlign over an FPN pyramid, NHWC features, one launch for all levels.
// rois [N,5] = (batch index, x0, y0, x1, y1) in input-image pixels; level[n] selects the feature map.
// ---------------------------------------------------------------------------------------------------------
struct Pyramid { const float* feat[4]; int h[4], w[4]; float scale[4]; };

__device__ __forceinline__ float4 bilinear4(const float* __restrict__ base, int H, int W, int C, float y, float x) {
  float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return z;
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
  const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  const float4 v1 = *reinterpret_cast<const float4*>(base + ((size_t)y_low * W + x_low) * C);
  const float4 v2 = *reinterpret_cast<const float4*>(base + ((size_t)y_low * W + x_high) * C);
  const float4 v3 = *reinterpret_cast<const float4*>(base + ((size_t)y_high * W + x_low) * C);
  const float4 v4 = *reinterpret_cast<const float4*>(base + ((size_t)y_high * W + x_high) * C);
  z.x = w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
  z.y = w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
  z.z = w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
  z.w = w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
  return z;
}

__global__ __launch_bounds__(256) void roi_align_kernel(Pyramid pyr, const float* __restrict__ rois,
                                                        const int* __restrict__ level, int C, int P, int sampling_ratio,
                                                        int aligned, float* __restrict__ out, long long total) {
  const int groups = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    long long t = i / groups;
    const int pw = (int)(t % P);
    t /= P;
    const int ph = (int)(t % P);
    const int n = (int)(t / P);
    const float* r = rois + (size_t)n * 5;
    const int lv = level[n];
    const int H = pyr.h[lv], W = pyr.w[lv];
    const float sc = pyr.scale[lv];
    const float off = aligned ? 0.5f : 0.f;
    const float sw = r[1] * sc - off, sh = r[2] * sc - off, ew = r[3] * sc - off, eh = r[4] * sc - off;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    const float bh = rh / (float)P, bw = rw / (float)P;
    const int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
    const int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
    const float count = fmaxf((float)(gh * gw), 1.f);
    const float* base = pyr.feat[lv] + (size_t)((int)r[0]) * H * W * C + g * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int iy = 0; iy < gh; ++iy) {
      const float y = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
        const float4 v = bilinear4(base, H, W, C, y, x);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    acc.x /= count; acc.y /= count; acc.z /= count; acc.w /= count;
    *reinterpret_cast<float4*>(out + (size_t)i * 4) = acc;
  }
}

// ---------------------------------------------------------------------------------------------------------
// NMS of boxes already sorted by descending score, restricted to equal category ids (batched_nms), for up to
// kMaxSegs independent segments (images) per launch.
// Pass 1: suppression bit matrix per segment (box i suppresses j > i when IoU > thr).
// Pass 2: one wave per segment walks the matrix in 64-box blocks: the decisions inside a block only need the
// block's 64x64 diagonal words (resolved in registers with readlane, no memory traffic), then the rows of the
// kept boxes are OR-ed into the running "removed" set with all loads independent -- n/64 dependent steps
// instead of n.
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxSegs = 64;
struct NmsSegs {
  int off[kMaxSegs + 1];            // box offsets of the segments
  long long ws_off[kMaxSegs];       // word offsets of their bit matrices
};

__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes_all, const int* __restrict__ cat_all,
                                                      const NmsSegs segs, float thr, unsigned long long* __restrict__ ws) {
  const int seg = blockIdx.z;
  const int n = segs.off[seg + 1] - segs.off[seg];
  const int words = (n + 63) >> 6;
  const int row_blk = blockIdx.y, col_blk = blockIdx.x;
  if (row_blk >= words || col_blk >= words || col_blk < row_blk) return;
  const float* boxes = boxes_all + (size_t)segs.off[seg] * 4;
  const int* cat = cat_all ? cat_all + segs.off[seg] : nullptr;
  unsigned long long* mask = ws + segs.ws_off[seg];
  const int i = row_blk * 64 + threadIdx.x;
  __shared__ float sb[64][4];
  __shared__ int sc[64];
  const int j0 = col_blk * 64;
  if (j0 + (int)threadIdx.x < n) {
    const float* b = boxes + (size_t)(j0 + threadIdx.x) * 4;
    sb[threadIdx.x][0] = b[0]; sb[threadIdx.x][1] = b[1]; sb[threadIdx.x][2] = b[2]; sb[threadIdx.x][3] = b[3];
    sc[threadIdx.x] = cat ? cat[j0 + threadIdx.x] : 0;
  }
  __syncthreads();
  if (i >= n) return;
  const float* a = boxes + (size_t)i * 4;
  const float ax0 = a[0], ay0 = a[1], ax1 = a[2], ay1 = a[3];
  const float area_a = (ax1 - ax0) * (ay1 - ay0);
  const int ca = cat ? cat[i] : 0;
  unsigned long long bits = 0;
  const int lim = min(64, n - j0);
  for (int k = (row_blk == col_blk ? (int)threadIdx.x + 1 : 0); k < lim; ++k) {
    if (sc[k] != ca) continue;
    const float ix0 = fmaxf(ax0, sb[k][0]), iy0 = fmaxf(ay0, sb[k][1]);
    const float ix1 = fminf(ax1, sb[k][2]), iy1 = fminf(ay1, sb[k][3]);
    const float iw = fmaxf(ix1 - ix0, 0.f), ih = fmaxf(iy1 - iy0, 0.f);
    const float inter = iw * ih;
    const float area_b = (sb[k][2] - sb[k][0]) * (sb[k][3] - sb[k][1]);
    if (inter / (area_a + area_b - inter) > thr) bits |= 1ull << k;
  }
  mask[(size_t)i * words + col_blk] = bits;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, lane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ ws, const NmsSegs segs,
                                                      unsigned char* __restrict__ keep_all) {
  extern __shared__ unsigned long long removed[];   // [words]
  const int seg = blockIdx.x;
  const int n = segs.off[seg + 1] - segs.off[seg];
  const int words = (n + 63) >> 6;
  const unsigned long long* mask = ws + segs.ws_off[seg];
  unsigned char* keep = keep_all + segs.off[seg];
  const int lane = threadIdx.x;
  for (int w = lane; w < words; w += 64) removed[w] = 0;
  __syncthreads();
  for (int b = 0; b < words; ++b) {
    const int i = b * 64 + lane;
    // rows only carry words for columns >= their own block (earlier ones were never written: zero-filled)
    const unsigned long long diag = i < n ? mask[(size_t)i * words + b] : 0ull;
    unsigned long long alive = ~removed[b];                      // uniform
    if (b == words - 1 && (n & 63)) alive &= (1ull << (n & 63)) - 1;
    for (int l = 0; l < 64; ++l) {
      const unsigned long long d = readlane64(diag, l);          // uniform
      if ((alive >> l) & 1ull) alive &= ~d;
    }
    if (i < n) keep[i] = (alive >> lane) & 1ull;
    // fold the kept rows of this block into the later words
    for (int w = b + 1 + lane; w < words; w += 64) {
      unsigned long long acc = 0;
      unsigned long long rest = alive;
      while (rest) {
        const int r = __builtin_ctzll(rest);
        rest &= rest - 1;
        acc |= mask[(size_t)(b * 64 + r) * words + w];
      }
      removed[w] |= acc;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// paste_masks_in_image: for every instance, resample its MxM probability map into its box on the HxW image
// (grid_sample bilinear, align_corners=False, zeros outside) and threshold.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void paste_masks_kernel(const float* __restrict__ masks, const float* __restrict__ boxes,
                                                          int M, int H, int W, float thr, unsigned char* __restrict__ out,
                                                          long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    const long long t = i / W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const float* b = boxes + (size_t)n * 4;
    const float v = paste_value(masks + (size_t)n * M * M, M, b, x, y);
    out[i] = v >= thr ? 1 : 0;
  }
}

static inline unsigned grid_for(long long items) {
  long long g = (items + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace peanut

using namespace peanut;

extern "C" {

int peanut_roi_align(const float* const* feats, const int* feat_hw, const float* scales, int n_levels, int C,
                     const float* rois, const int* levels, int n_rois, int pooled, int sampling_ratio, int aligned,
                     float* out, void* stream) {
  if (!feats || !feat_hw || !scales || !out || n_levels < 1 || n_levels > 4 || C % 4 || pooled < 1)
    return fail(PEANUT_EINVAL, "peanut_roi_align: bad argument");
  if (n_rois == 0) return 0;
  if (!rois || !levels) return fail(PEANUT_EINVAL, "peanut_roi_align: null rois");
  Pyramid p{};
  for (int l = 0; l < n_levels; ++l) { p.feat[l] = feats[l]; p.h[l] = feat_hw[2 * l]; p.w[l] = feat_hw[2 * l + 1]; p.scale[l] = scales[l]; }
  const long long total = (long long)n_rois * pooled * pooled * (C / 4);
  hipLaunchKernelGGL(roi_align_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p, rois, levels, C, pooled,
                     sampling_ratio, aligned, out, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("roi_align: ") + hipGetErrorString(e));
}

size_t peanut_nms_workspace_bytes(int n) { return (size_t)n * ((n + 63) / 64) * sizeof(unsigned long long); }

int peanut_nms_segments(const float* boxes_sorted, const int* categories, const int* seg_offsets_host, int n_segments,
                        float iou_threshold, void* workspace, unsigned char* keep, void* stream) {
  if (n_segments == 0) return 0;
  if (!boxes_sorted || !seg_offsets_host || !workspace || !keep || n_segments < 0)
    return fail(PEANUT_EINVAL, "peanut_nms_segments: bad argument");
  hipStream_t s = (hipStream_t)stream;
  long long ws_words = 0;
  for (int s0 = 0; s0 < n_segments; s0 += kMaxSegs) {
    const int cnt = n_segments - s0 < kMaxSegs ? n_segments - s0 : kMaxSegs;
    NmsSegs segs{};
    int max_words = 0;
    const long long first_word = ws_words;
    for (int k = 0; k < cnt; ++k) {
      const int lo = seg_offsets_host[s0 + k], hi = seg_offsets_host[s0 + k + 1];
      if (hi < lo) return fail(PEANUT_EINVAL, "peanut_nms_segments: offsets must be non-decreasing");
      const int n = hi - lo, words = (n + 63) / 64;
      if ((size_t)words * sizeof(unsigned long long) > 60 * 1024) return fail(PEANUT_EINVAL, "peanut_nms: more than 491520 boxes");
      segs.off[k] = lo; segs.off[k + 1] = hi;
      segs.ws_off[k] = ws_words;
      ws_words += (long long)n * words;
      if (words > max_words) max_words = words;
    }
    if (max_words == 0) continue;
    unsigned long long* ws = (unsigned long long*)workspace;
    PEANUT_HIP_CHECK(hipMemsetAsync(ws + first_word, 0, (size_t)(ws_words - first_word) * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(nms_mask_kernel, dim3(max_words, max_words, cnt), dim3(64), 0, s, boxes_sorted, categories, segs,
                       iou_threshold, ws);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(cnt), dim3(64), (size_t)max_words * sizeof(unsigned long long), s,
                       (const unsigned long long*)ws, segs, keep);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("nms: ") + hipGetErrorString(e));
}

int peanut_nms(const float* boxes_sorted, const int* categories, int n, float iou_threshold, void* workspace,
               unsigned char* keep, void* stream) {
  if (n == 0) return 0;
  if (n < 0) return fail(PEANUT_EINVAL, "peanut_nms: bad argument");
  const int off[2] = {0, n};
  return peanut_nms_segments(boxes_sorted, categories, off, 1, iou_threshold, workspace, keep, stream);
}

int peanut_paste_masks(const float* masks, const float* boxes, int n, int M, int H, int W, float threshold,
                       unsigned char* out, void* stream) {
  if (n == 0) return 0;
  if (!masks || !boxes || !out || M < 1 || H < 1 || W < 1) return fail(PEANUT_EINVAL, "peanut_paste_masks: bad argument");
  const long long total = (long long)n * H * W;
  hipLaunchKernelGGL(paste_masks_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, masks, boxes, M, H, W,
                     threshold, out, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("paste_masks: ") + hipGetErrorString(e));
}

}  // extern "C"
