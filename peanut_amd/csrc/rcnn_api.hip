// Stage 1 front end: the part of detectron2's Mask R-CNN (R-101-FPN) inference that is dense convolution --
// test-time preprocessing, ResNet-101 bottom-up path, FPN top-down path, RPN head -- as one static launch
// sequence on the fused implicit-GEMM conv kernels, plus three small streaming kernels.
//
// Reference boundary: SemanticPredMaskRCNN.__init__/get_prediction build and call detectron2's
// DefaultPredictor (nav/agent/utils/segmentation.py:31-38,45) with
// nav/agent/utils/COCO-InstSeg/mask_rcnn_R_101_cat9.yaml.  detectron2 itself is not in the reference
// checkout, so the module graph below follows detectron2 v0.6's published definitions under that yaml
// (BasicStem, BottleneckBlock with STRIDE_IN_1X1, FrozenBatchNorm2d folded to scale/shift, FPN with sum
// fusion and LastLevelMaxPool, StandardRPNHead).  Proposal selection / NMS / ROIAlign / box + mask heads are
// rcnn_ops.hip and rcnn_post.hip (peanut_rcnn_inference).  Parity is pinned only against the torch
// restatement oracle/rcnn_ref.py ("parity unpinned" w.r.t. the reference).
#include <string.h>

#include <math.h>
#include <stdlib.h>
#include "rcnn_internal.h"

namespace peanut {
namespace {

// ---- uint8 BGR [B,H,W,3] -> bilinear resize -> round to 8-bit -> (x - mean) / std -> NHWC [B,Hp,Wp,16] ----
// (zero in the padded border and in channels 3..15).  One thread per output pixel.
struct Norm3 { float mean[3], inv_std[3]; };

// one pixel of the network input: the frame resized as detectron2's ResizeShortestEdge does it for uint8 images --
// PIL.Image.resize(BILINEAR), i.e. Pillow's ImagingResample (src/libImaging/Resample.c, 8 bits per channel):
// a horizontal pass, then a vertical pass over its uint8 result, each a convolution with fixed-point coefficients
// (22 fractional bits, rounded half up, clipped to 0..255).  Both passes are evaluated here for one output pixel:
// the <= ksy rows of the intermediate image it needs are recomputed on the fly (integer arithmetic: exact).  Then
// (x - mean) / std; zeros in the padding up to the size-divisible canvas.  Tables: see resize_tables.
struct ResizeTab { const int* x; const int* y; int ksx, ksy; };
__device__ __forceinline__ int clip8_fixed(int acc) {
  const int v = acc >> 22;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}
__device__ __forceinline__ float4 rcnn_input_pixel(const uint8_t* __restrict__ img, long long b, int y, int x, int H, int W, int nh,
                                                   int nw, const Norm3& nm, const ResizeTab& rz) {
  if (y >= nh || x >= nw) return make_float4(0.f, 0.f, 0.f, 0.f);
  const int* tx = rz.x + (size_t)x * (2 + rz.ksx);
  const int* ty = rz.y + (size_t)y * (2 + rz.ksy);
  const int x0 = tx[0], nx = tx[1], y0 = ty[0], ny = ty[1];
  int acc[3] = {1 << 21, 1 << 21, 1 << 21};
  for (int j = 0; j < ny; ++j) {
    const uint8_t* row = img + ((b * H + (y0 + j)) * W + x0) * 3;
    int h[3] = {1 << 21, 1 << 21, 1 << 21};
    for (int i = 0; i < nx; ++i) {
      const int k = tx[2 + i];
      h[0] += (int)row[i * 3 + 0] * k; h[1] += (int)row[i * 3 + 1] * k; h[2] += (int)row[i * 3 + 2] * k;
    }
    const int k = ty[2 + j];
    acc[0] += clip8_fixed(h[0]) * k; acc[1] += clip8_fixed(h[1]) * k; acc[2] += clip8_fixed(h[2]) * k;
  }
  float c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) c[k] = ((float)clip8_fixed(acc[k]) - nm.mean[k]) * nm.inv_std[k];
  return make_float4(c[0], c[1], c[2], 0.f);
}

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1, stretched by the scale when
// shrinking), in double like the C original: row i of the table = {first source index, taps, ks weights}.
static std::vector<int> resize_tables(int in_size, int out_size, int* ks_out) {
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const int ks = (int)ceil(support) * 2 + 1;
  std::vector<int> tab((size_t)out_size * (2 + ks), 0);
  std::vector<double> kk(ks);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      const double w = a < 1.0 ? 1.0 - a : 0.0;
      kk[x] = w;
      ww += w;
    }
    int* row = tab.data() + (size_t)xx * (2 + ks);
    row[0] = xmin; row[1] = xmax;
    for (int x = 0; x < xmax; ++x) {
      const double v = ww != 0.0 ? kk[x] / ww : kk[x];
      row[2 + x] = v < 0 ? (int)(-0.5 + v * (double)(1 << 22)) : (int)(0.5 + v * (double)(1 << 22));
    }
  }
  *ks_out = ks;
  return tab;
}

__global__ __launch_bounds__(256) void rcnn_preprocess_kernel(const uint8_t* __restrict__ img, float* __restrict__ out,
                                                              int H, int W, int nh, int nw, int Hp, int Wp, Norm3 nm,
                                                              ResizeTab rz, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const long long t = i / Wp;
    const int y = (int)(t % Hp);
    const long long b = t / Hp;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* o = reinterpret_cast<float4*>(out + (size_t)i * 16);
    o[0] = rcnn_input_pixel(img, b, y, x, H, W, nh, nw, nm, rz); o[1] = z; o[2] = z; o[3] = z;
  }
}

// operator-level export (peanut_rcnn_preprocess): the same pixels as plain NCHW [B,3,Hp,Wp], for bisecting / parity
__global__ __launch_bounds__(256) void rcnn_preprocess_nchw_kernel(const uint8_t* __restrict__ img, float* __restrict__ out,
                                                                   int H, int W, int nh, int nw, int Hp, int Wp, Norm3 nm,
                                                                   ResizeTab rz, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const long long t = i / Wp;
    const int y = (int)(t % Hp);
    const long long b = t / Hp;
    const float4 v = rcnn_input_pixel(img, b, y, x, H, W, nh, nw, nm, rz);
    float* o = out + ((size_t)b * 3 * Hp + y) * Wp + x;
    o[0] = v.x; o[(size_t)Hp * Wp] = v.y; o[(size_t)2 * Hp * Wp] = v.z;
  }
}

// The same image in 2x2 space-to-depth form: out [B, Hp/2, Wp/2, 16], channel (dy*2 + dx)*4 + c = channel c of pixel
// (2Y + dy, 2X + dx) (c = 3: zero).  The 7x7 stride-2 stem conv over 3 channels becomes a 4x4 stride-1 conv over these
// 16 (add_stem_s2d): 16 taps x 16 channels instead of 49 taps x 16 zero-padded channels in the implicit GEMM.
__global__ __launch_bounds__(256) void rcnn_preprocess_s2d_kernel(const uint8_t* __restrict__ img, float* __restrict__ out,
                                                                  int H, int W, int nh, int nw, int Hp, int Wp, Norm3 nm,
                                                                  ResizeTab rz, long long total) {
  const int H2 = Hp >> 1, W2 = Wp >> 1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i & 3);                   // (dy, dx) of this thread: four threads per output pixel
    const long long pix = i >> 2;
    const int X = (int)(pix % W2);
    const long long t = pix / W2;
    const int Y = (int)(t % H2);
    const long long b = t / H2;
    reinterpret_cast<float4*>(out + (size_t)pix * 16)[q] = rcnn_input_pixel(img, b, 2 * Y + (q >> 1), 2 * X + (q & 1), H, W, nh, nw, nm, rz);
  }
}

// ---- FPN top-down: y[b,h,w,c] += top[b,h/2,w/2,c]  (F.interpolate(scale_factor=2, nearest) + add) ----
__global__ __launch_bounds__(256) void add_upsampled2x_kernel(float* __restrict__ y, const float* __restrict__ top, int H,
                                                              int W, int C, long long total) {
  const int groups = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    long long pix = i / groups;
    const int x = (int)(pix % W);
    pix /= W;
    const int yy = (int)(pix % H);
    const long long b = pix / H;
    const float4 t = *reinterpret_cast<const float4*>(top + (((b * (H / 2) + yy / 2) * (W / 2)) + x / 2) * C + g * 4);
    float4* d = reinterpret_cast<float4*>(y + i * 4);
    float4 v = *d;
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    *d = v;
  }
}

// ---- LastLevelMaxPool: max_pool2d(kernel 1, stride 2) = take every second pixel ----
__global__ __launch_bounds__(256) void subsample2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                         int C, int Ho, int Wo, long long total) {
  const int groups = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    long long pix = i / groups;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const long long b = pix / Ho;
    *reinterpret_cast<float4*>(y + i * 4) =
        *reinterpret_cast<const float4*>(x + ((b * H + oy * 2) * W + ox * 2) * C + g * 4);
  }
}

inline unsigned grid_for(long long items) {
  long long g = (items + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace
}  // namespace peanut

using namespace peanut;

namespace {

// Winograd form request of the detector's front-end convs: 0 = per-shape policy; PEANUT_RCNN_WINO_M = 4 pins F(4x4) (A/B)
int rcnn_wino_request() {
  const int m = (int)opt(OPT_RCNN_WINO_M);
  return (m >= 4 && m <= 6) ? m : 0;
}

int add_rconv(peanut_rcnn* h, const TensorMap& tm, const std::string& name, int cin, int cin_pad, int cout, int k,
              int stride, int pad, bool norm, int relu, ConvLayer** out) {
  int rc = 0;
  const int64_t wshape[4] = {cout, cin, k, k};
  const peanut_tensor* w = tm.get(name + ".weight", 4, wshape, &rc);
  if (!w) return rc;
  std::vector<float> scale(cout, 1.f), shift(cout, 0.f);
  if (norm) {
    if ((rc = bn_fold_eps(tm, name + ".norm", cout, h->cfg.bn_eps, scale.data(), shift.data()))) return rc;
  } else {
    const int64_t cshape[1] = {cout};
    const peanut_tensor* b = tm.get(name + ".bias", 1, cshape, &rc);
    if (!b) return rc;
    for (int n = 0; n < cout; ++n) shift[n] = b->data[n];
  }
  auto L = std::make_unique<ConvLayer>();
  L->name = name;
  if ((rc = upload_conv(*L, w->data, scale.data(), shift.data(), cout, cin, cin_pad, k, k, stride, pad, 1, relu,
                        h->cfg.precision)))
    return rc;
  // fp32 kernels: the prediction planner's forms (F(6x6) with an F(4x4) twin, chosen per shape in push_rconv; the position
  // GEMMs accumulate in two levels, net_common.h); emulated modes: F(4x4)
  if (h->cfg.conv_algo == PEANUT_ALGO_AUTO && wino_eligible(cin_pad, cout, k, k, stride, pad, 1, h->cfg.precision) &&
      (rc = upload_wino_forms(*L, w->data, scale.data(), shift.data(), cout, cin, cin_pad, pad, 1, relu, h->cfg.precision,
                              rs_planes_of(h->cfg.precision) ? 4 : rcnn_wino_request())))
    return rc;
  *out = L.get();
  h->convs.push_back(std::move(L));
  return 0;
}

// BasicStem's 7x7 stride-2 pad-3 conv over 3 channels restated on the 2x2 space-to-depth image (rcnn_preprocess_s2d_kernel):
// output (oy, ox) reads input rows 2 oy + ky - 3; in blocks of two rows that is block oy + by - 2, row dy inside it, with
// ky = 2 by + dy - 1 (by = 0..3) -- a 4x4 stride-1 pad-2 conv whose taps outside 0 <= ky <= 6 are zero.  Same products,
// summed in another order.
int add_stem_s2d(peanut_rcnn* h, const TensorMap& tm, const std::string& name, int cout, ConvLayer** out) {
  int rc = 0;
  const int64_t wshape[4] = {cout, 3, 7, 7};
  const peanut_tensor* w = tm.get(name + ".weight", 4, wshape, &rc);
  if (!w) return rc;
  std::vector<float> scale(cout, 1.f), shift(cout, 0.f);
  if ((rc = bn_fold_eps(tm, name + ".norm", cout, h->cfg.bn_eps, scale.data(), shift.data()))) return rc;
  std::vector<float> w2((size_t)cout * 16 * 4 * 4, 0.f);
  for (int o = 0; o < cout; ++o)
    for (int c = 0; c < 3; ++c)
      for (int by = 0; by < 4; ++by)
        for (int dy = 0; dy < 2; ++dy)
          for (int bx = 0; bx < 4; ++bx)
            for (int dx = 0; dx < 2; ++dx) {
              const int ky = 2 * by + dy - 1, kx = 2 * bx + dx - 1;
              if (ky < 0 || ky > 6 || kx < 0 || kx > 6) continue;
              w2[(((size_t)o * 16 + (dy * 2 + dx) * 4 + c) * 4 + by) * 4 + bx] = w->data[(((size_t)o * 3 + c) * 7 + ky) * 7 + kx];
            }
  auto L = std::make_unique<ConvLayer>();
  L->name = name;
  if ((rc = upload_conv(*L, w2.data(), scale.data(), shift.data(), cout, 16, 16, 4, 4, 1, 2, 1, 1, h->cfg.precision))) return rc;
  *out = L.get();
  h->convs.push_back(std::move(L));
  return 0;
}

// conv3 + FrozenBN and the block's stride-1 shortcut conv + FrozenBN as one pointwise layer over [conv2 output | block
// input] (detectron2 BottleneckBlock.forward: out = conv3(out) + shortcut(x), then ReLU): scales folded into the weights,
// shifts added -- the same restatement as pred_api.hip: add_fused_c3d.
int add_fused_c3s(peanut_rcnn* h, const TensorMap& tm, const std::string& blk, int bott, int cin, int cout, ConvLayer** out) {
  int rc = 0;
  const int64_t w3s[4] = {cout, bott, 1, 1}, wss[4] = {cout, cin, 1, 1};
  const peanut_tensor* w3 = tm.get(blk + ".conv3.weight", 4, w3s, &rc);
  if (!w3) return rc;
  const peanut_tensor* ws = tm.get(blk + ".shortcut.weight", 4, wss, &rc);
  if (!ws) return rc;
  std::vector<float> s3(cout), b3(cout), ss(cout), bs(cout);
  if ((rc = bn_fold_eps(tm, blk + ".conv3.norm", cout, h->cfg.bn_eps, s3.data(), b3.data()))) return rc;
  if ((rc = bn_fold_eps(tm, blk + ".shortcut.norm", cout, h->cfg.bn_eps, ss.data(), bs.data()))) return rc;
  const int k = bott + cin;
  std::vector<float> w((size_t)cout * k), shift(cout);
  for (int n = 0; n < cout; ++n) {
    for (int c = 0; c < bott; ++c) w[(size_t)n * k + c] = s3[n] * w3->data[(size_t)n * bott + c];
    for (int c = 0; c < cin; ++c) w[(size_t)n * k + bott + c] = ss[n] * ws->data[(size_t)n * cin + c];
    shift[n] = b3[n] + bs[n];
  }
  auto L = std::make_unique<ConvLayer>();
  L->name = blk + ".conv3+shortcut";
  if ((rc = upload_conv(*L, w.data(), nullptr, shift.data(), cout, k, k, 1, 1, 1, 0, 1, 1,
                        rs_planes_of(h->cfg.precision) ? h->cfg.precision : PEANUT_PREC_FP32))) return rc;
  *out = L.get();
  h->convs.push_back(std::move(L));
  return 0;
}

void resized_hw(const peanut_rcnn_cfg& c, int h, int w, int* nh, int* nw) {
  // detectron2 ResizeShortestEdge.get_output_shape
  const double size = (double)c.min_size;
  const double scale = size / (h < w ? h : w);
  double newh = h < w ? size : scale * h, neww = h < w ? scale * w : size;
  const double mx = newh > neww ? newh : neww;
  if (mx > c.max_size) {
    const double s = (double)c.max_size / mx;
    newh *= s;
    neww *= s;
  }
  *nh = (int)(newh + 0.5);
  *nw = (int)(neww + 0.5);
}

void push_rconv(RPlan& pl, Arena& ar, const ConvLayer* L, const Act& in, const Act* res, const Act& out, int ext_slot = -1,
                std::vector<Act>* keep = nullptr) {
  if (L->has_wino) L = wino_pick_form(L, in.B, in.H, in.W);
  ROp op;
  op.kind = R_CONV; op.name = L->name; op.conv = L; op.in = in; op.out = out;
  op.kernel = L->d.rs ? std::string(L->d.rs == 2 ? "conv_rs" : "gemm_rs") + (L->d.s_planes == 3 ? "6" : (L->d.s_planes == 4 ? "3h" : "3"))
                      : "conv_igemm_128x" + std::to_string(L->d.bn_tile) + "x" + std::to_string(L->d.bk);
  if (res) { op.res = *res; op.has_res = true; }
  op.flops = conv_flops(L, out);
  op.ext_slot = ext_slot;
  if (L->has_wino) {
    size_t vf, mf;
    wino_scratch_floats(*L, in.B, in.H, in.W, &vf, &mf);
    op.wino_v.bytes = vf * sizeof(float); op.wino_v.off = ar.alloc(op.wino_v.bytes);
    op.wino_m.bytes = mf * sizeof(float); op.wino_m.off = ar.alloc(op.wino_m.bytes);
    op.has_wino = true;
    if (keep) {                // (an op that runs next to later ones keeps its scratch until they have joined)
      keep->push_back(op.wino_v);
      keep->push_back(op.wino_m);
    } else {
      ar.release(op.wino_v.off, op.wino_v.bytes);
      ar.release(op.wino_m.off, op.wino_m.bytes);
    }
  }
  pl.ops.push_back(op);
}

Act conv_out_act(Arena& ar, const ConvLayer* L, const Act& in) {
  const ConvDesc& d = L->d;
  return make_act(ar, in.B, conv_out_dim(in.H, d.kh, d.stride, d.pad, d.dil), conv_out_dim(in.W, d.kw, d.stride, d.pad, d.dil),
                  d.cout);
}

std::unique_ptr<RPlan> build_rplan(const peanut_rcnn* h, int B, int H, int W) {
  auto pl = std::make_unique<RPlan>();
  pl->B = B; pl->H = H; pl->W = W;
  resized_hw(h->cfg, H, W, &pl->nh, &pl->nw);
  const int d = h->cfg.size_divisibility;
  pl->Hp = (pl->nh + d - 1) / d * d;
  pl->Wp = (pl->nw + d - 1) / d * d;
  Arena ar;
  auto rel = [&](const Act& t) { ar.release(t.off, t.bytes); };
  pl->splitk.bytes = kSplitKScratchFloats * sizeof(float);
  pl->splitk.off = ar.alloc(pl->splitk.bytes);
  // BasicStem, on the space-to-depth image when there is such a form of it (Hp, Wp are multiples of 32)
  const ConvLayer* stem = h->stem_s2d ? h->stem_s2d : h->stem;
  Act x = h->stem_s2d ? make_act(ar, B, pl->Hp / 2, pl->Wp / 2, 16) : make_act(ar, B, pl->Hp, pl->Wp, 16);
  { ROp op; op.kind = R_PREPROCESS; op.name = "preprocess"; op.kernel = "rcnn_preprocess"; op.out = x; pl->ops.push_back(op); }
  Act s = make_act(ar, B, conv_out_dim(pl->Hp, 7, 2, 3, 1), conv_out_dim(pl->Wp, 7, 2, 3, 1), h->stem->d.cout);
  push_rconv(*pl, ar, stem, x, nullptr, s);
  pl->ops.back().flops = conv_flops(h->stem, s);   // nominal: the 7x7 form
  rel(x);
  Act cur = make_act(ar, B, conv_out_dim(s.H, 3, 2, 1, 1), conv_out_dim(s.W, 3, 2, 1, 1), s.C);
  { ROp op; op.kind = R_MAXPOOL; op.name = "stem.maxpool"; op.kernel = "maxpool"; op.in = s; op.out = cur; pl->ops.push_back(op); }
  rel(s);
  // res2..res5
  Act feats[4];
  for (size_t si = 0; si < h->stages.size(); ++si) {
    for (const auto& blk : h->stages[si]) {
      Act idn = cur;
      bool own = false;
      if (blk.shortcut && !blk.c3s) {
        idn = conv_out_act(ar, blk.shortcut, cur);
        push_rconv(*pl, ar, blk.shortcut, cur, nullptr, idn);
        own = true;
      }
      Act t1 = conv_out_act(ar, blk.c1, cur);
      push_rconv(*pl, ar, blk.c1, cur, nullptr, t1);
      Act t2 = conv_out_act(ar, blk.c2, t1);
      push_rconv(*pl, ar, blk.c2, t1, nullptr, t2);
      {   // conv1's output is read by conv2 alone: conv2's input transform may sum conv1's split-K partial tiles (common.h: DeferredSplit)
        ROp& o2 = pl->ops[pl->ops.size() - 1];
        ROp& o1 = pl->ops[pl->ops.size() - 2];
        if (opt(OPT_DEFER_SPLITK) != 0 && o2.has_wino && o1.conv == blk.c1 && !o1.has_res && o1.ext_slot < 0 &&
            conv_layer_accepts_deferred(*o2.conv, t1.B, t1.H, t1.W))
          o1.defer_ok = true;
      }
      rel(t1);
      Act y = conv_out_act(ar, blk.c3, t2);
      if (blk.c3s) {                                    // [t2 | x] x [W3'; Ws'] + shifts, ReLU: conv3 and the shortcut in one GEMM
        push_rconv(*pl, ar, blk.c3s, t2, nullptr, y);
        pl->ops.back().in2 = cur;
        pl->ops.back().has_in2 = true;
      } else {
        push_rconv(*pl, ar, blk.c3, t2, &idn, y);      // + shortcut, ReLU fused
      }
      rel(t2);
      if (own) rel(idn);
      // the block input dies here unless it is a stage output that the FPN laterals still need
      bool is_feat = false;
      for (size_t q = 0; q < si; ++q) is_feat |= (cur.off == feats[q].off && cur.bytes == feats[q].bytes);
      if (!is_feat) rel(cur);
      cur = y;
    }
    feats[si] = cur;
  }
  // FPN top-down (levels 5 -> 2).  The 3x3 output convs of p5, p4, p3 depend on their own lateral sum only: with rcnn_fpn_overlap
  // they are marked for the side stream and run next to the chain lateral -> top-down add -> ... -> p2's output conv (a frame at
  // batch 1 leaves most CUs idle in every one of these launches).  What they read and their scratch stays allocated until the join.
  const bool fpn_side = opt(OPT_RCNN_FPN_OVERLAP) != 0;
  std::vector<Act> keep;
  if (fpn_side) {
    pl->splitk_side.bytes = kSplitKSideFloats * sizeof(float);
    pl->splitk_side.off = ar.alloc(pl->splitk_side.bytes);
  }
  Act prev{};
  Act p[5];
  for (int lvl = 3; lvl >= 0; --lvl) {
    Act lat = conv_out_act(ar, h->lateral[lvl], feats[lvl]);
    push_rconv(*pl, ar, h->lateral[lvl], feats[lvl], nullptr, lat);
    rel(feats[lvl]);
    if (lvl < 3) {
      ROp op; op.kind = R_ADD_UP; op.name = "fpn_topdown" + std::to_string(lvl + 2); op.kernel = "add_upsampled2x";
      op.in = prev; op.out = lat; pl->ops.push_back(op);
      if (fpn_side) keep.push_back(prev); else rel(prev);
    }
    prev = lat;
    p[lvl] = conv_out_act(ar, h->output[lvl], prev);
    const bool side = fpn_side && lvl > 0;
    push_rconv(*pl, ar, h->output[lvl], prev, nullptr, p[lvl], lvl, side ? &keep : nullptr);
    pl->ops.back().side = side;
  }
  rel(prev);
  p[4] = make_act(ar, B, (p[3].H - 1) / 2 + 1, (p[3].W - 1) / 2 + 1, p[3].C);
  { ROp op; op.kind = R_SUBSAMPLE; op.name = "p6"; op.kernel = "subsample2"; op.in = p[3]; op.out = p[4]; op.ext_slot = 4; op.in_ext_slot = 3;
    op.join_side = fpn_side; pl->ops.push_back(op); }
  for (const Act& t : keep) rel(t);
  // RPN head on p2..p6: the fused chain over all levels (RpnFused; taken at run time when the caller's objectness / delta buffers
  // are contiguous in level order or absent) ...
  if (h->rpn_conv->has_wino && opt(OPT_RCNN_RPN_FUSED) != 0) {
    RpnFused& f = pl->rpn;
    f.form = wino_pick_form(h->rpn_conv, B, p[0].H, p[0].W);
    const int cin = f.form->d.cin, cout = f.form->d.cout, A = h->rpn_obj->d.cout;
    long long tiles = 0, rows = 0;
    double fl = 0.0;
    for (int lvl = 0; lvl < 5; ++lvl) {
      int th, tw;
      long long nt, mp;
      wino_geometry(B, p[lvl].H, p[lvl].W, 1, &th, &tw, &nt, &mp, 128, f.form->wino_m);
      f.in[lvl] = p[lvl];
      f.tile_off[lvl] = tiles; tiles += nt;
      f.row_off[lvl] = rows; rows += (long long)B * p[lvl].H * p[lvl].W;
      Act out_l = p[lvl]; out_l.C = cout;
      fl += conv_flops(h->rpn_conv, out_l);
      out_l.C = A; fl += conv_flops(h->rpn_obj, out_l);
      out_l.C = 4 * A; fl += conv_flops(h->rpn_delta, out_l);
    }
    f.gran = tiles >= 2048 ? 256 : 128;
    f.m_pad_total = (tiles + f.gran - 1) / f.gran * f.gran;
    f.rows = rows;
    if ((long long)f.form->wino_np() * f.m_pad_total <= 0x7fffffffLL && rows <= 0x7fffffffLL) {
      f.v = make_act(ar, 1, 1, (int)((long long)f.form->wino_np() * f.m_pad_total), cin);
      f.m = make_act(ar, 1, 1, (int)((long long)f.form->wino_np() * f.m_pad_total), cout);
      f.t_all = make_act(ar, 1, 1, (int)rows, cout);
      f.obj_all = make_act(ar, 1, 1, (int)rows, A);
      f.dl_all = make_act(ar, 1, 1, (int)rows, 4 * A);
      f.on = true;
      ROp op;
      op.kind = R_RPN_FUSED; op.name = "proposal_generator.rpn_head[p2..p6 fused]"; op.kernel = "rpn_fused"; op.flops = fl;
      pl->ops.push_back(op);
      rel(f.v); rel(f.m); rel(f.t_all); rel(f.obj_all); rel(f.dl_all);
    }
  }
  // ... and level by level (fifteen convs; the form that runs when the outputs are scattered buffers, or option rcnn_rpn_fused = 0)
  for (int lvl = 0; lvl < 5; ++lvl) {
    pl->lvl_h[lvl] = p[lvl].H; pl->lvl_w[lvl] = p[lvl].W;
    Act t = conv_out_act(ar, h->rpn_conv, p[lvl]);
    push_rconv(*pl, ar, h->rpn_conv, p[lvl], nullptr, t);
    pl->ops.back().in_ext_slot = lvl;
    pl->ops.back().rpn_level = true;
    Act o = conv_out_act(ar, h->rpn_obj, t);
    push_rconv(*pl, ar, h->rpn_obj, t, nullptr, o, 5 + lvl);
    pl->ops.back().rpn_level = true;
    Act dl = conv_out_act(ar, h->rpn_delta, t);
    push_rconv(*pl, ar, h->rpn_delta, t, nullptr, dl, 10 + lvl);
    pl->ops.back().rpn_level = true;
    rel(t); rel(o); rel(dl);
  }
  size_t hw = pl->splitk.off + Arena::round_up(pl->splitk.bytes);
  if (pl->rpn.on)
    for (const Act* a : {&pl->rpn.v, &pl->rpn.m, &pl->rpn.t_all, &pl->rpn.obj_all, &pl->rpn.dl_all})
      if (a->off + Arena::round_up(a->bytes) > hw) hw = a->off + Arena::round_up(a->bytes);
  for (const auto& op : pl->ops)
    for (const Act* a : {&op.in, &op.in2, &op.res, &op.out, &op.wino_v, &op.wino_m})
      if (a->bytes && a->off + Arena::round_up(a->bytes) > hw) hw = a->off + Arena::round_up(a->bytes);
  pl->bytes = hw;
  return pl;
}

RPlan* get_rplan(peanut_rcnn* h, int B, int H, int W) {
  if (B <= 0 || H < 32 || W < 32) { set_error("rcnn: need B >= 1 and H, W >= 32"); return nullptr; }
  const std::string key = std::to_string(B) + "x" + std::to_string(H) + "x" + std::to_string(W);
  auto it = h->plans.find(key);
  if (it == h->plans.end()) {
    auto pl = build_rplan(h, B, H, W);
    const std::vector<int> tx = resize_tables(W, pl->nw, &pl->ksx), ty = resize_tables(H, pl->nh, &pl->ksy);
    if (pl->rz_x.ensure(tx.size() * sizeof(int)) || pl->rz_y.ensure(ty.size() * sizeof(int)) ||
        hipMemcpy(pl->rz_x.p, tx.data(), tx.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(pl->rz_y.p, ty.data(), ty.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      set_error("rcnn: resize table upload failed");
      return nullptr;
    }
    it = h->plans.emplace(key, std::move(pl)).first;
  }
  return it->second.get();
}

}  // namespace

extern "C" {

int peanut_rcnn_create(peanut_rcnn_t** out, const peanut_rcnn_cfg* cfg, const peanut_tensor* tensors, int n) {
  if (!out || !cfg || (!tensors && n > 0)) return fail(PEANUT_EINVAL, "peanut_rcnn_create: null argument");
  if (cfg->depth != 50 && cfg->depth != 101 && cfg->depth != 152) return fail(PEANUT_EINVAL, "rcnn: depth must be 50/101/152");
  if (cfg->fpn_out % 32 || cfg->num_anchors < 1 || cfg->min_size < 32 || cfg->size_divisibility != 32)
    return fail(PEANUT_EINVAL, "rcnn: unsupported configuration");
  if (!precision_known(cfg->precision)) return fail(PEANUT_EINVAL, "rcnn: precision must be PEANUT_PREC_{FP32,BF16X3,FP16X3,BF16X6}");
  if (cfg->conv_algo != PEANUT_ALGO_AUTO && cfg->conv_algo != PEANUT_ALGO_DIRECT) return fail(PEANUT_EINVAL, "rcnn: bad conv_algo");
  auto h = std::make_unique<peanut_rcnn>();
  OptionScope option_scope(&h->opts);
  h->cfg = *cfg;
  TensorMap tm;
  for (int i = 0; i < n; ++i)
    if (tensors[i].name) tm.m[tensors[i].name] = &tensors[i];
  int rc;
  if ((rc = add_rconv(h.get(), tm, "backbone.bottom_up.stem.conv1", 3, 16, cfg->stem_out, 7, 2, 3, true, 1, &h->stem))) return rc;
  const bool stem_s2d = opt(OPT_RCNN_STEM_S2D) != 0;
  if (stem_s2d && (rc = add_stem_s2d(h.get(), tm, "backbone.bottom_up.stem.conv1", cfg->stem_out, &h->stem_s2d))) return rc;
  const int nblocks[3][4] = {{3, 4, 6, 3}, {3, 4, 23, 3}, {3, 8, 36, 3}};
  const int* nb = nblocks[cfg->depth == 50 ? 0 : (cfg->depth == 101 ? 1 : 2)];
  int cin = cfg->stem_out, bott = cfg->res2_out / 4, cout = cfg->res2_out;
  for (int si = 0; si < 4; ++si) {
    std::vector<peanut_rcnn::Block> blocks;
    for (int bi = 0; bi < nb[si]; ++bi) {
      const std::string p = "backbone.bottom_up.res" + std::to_string(si + 2) + "." + std::to_string(bi);
      const int s = (bi == 0 && si > 0) ? 2 : 1;
      const int s1 = cfg->stride_in_1x1 ? s : 1, s3 = cfg->stride_in_1x1 ? 1 : s;
      peanut_rcnn::Block b{nullptr, nullptr, nullptr, nullptr, nullptr};
      if (cin != cout && (rc = add_rconv(h.get(), tm, p + ".shortcut", cin, cin, cout, 1, s, 0, true, 0, &b.shortcut))) return rc;
      if ((rc = add_rconv(h.get(), tm, p + ".conv1", cin, cin, bott, 1, s1, 0, true, 1, &b.c1))) return rc;
      if ((rc = add_rconv(h.get(), tm, p + ".conv2", bott, bott, bott, 3, s3, 1, true, 1, &b.c2))) return rc;
      if ((rc = add_rconv(h.get(), tm, p + ".conv3", bott, bott, cout, 1, 1, 0, true, 1, &b.c3))) return rc;   // ReLU after the add
      // conv_algo AUTO only: DIRECT keeps detectron2's op-for-op form (conv3 -> FrozenBN, shortcut -> FrozenBN, add, ReLU)
      if (cfg->conv_algo == PEANUT_ALGO_AUTO && b.shortcut && s == 1 && bott % 32 == 0 && cin % 32 == 0 && conv_pw_enabled() &&
          (rc = add_fused_c3s(h.get(), tm, p, bott, cin, cout, &b.c3s)))
        return rc;
      blocks.push_back(b);
      cin = cout;
    }
    h->stages.push_back(blocks);
    bott *= 2;
    cout *= 2;
  }
  int c = cfg->res2_out;
  for (int lvl = 0; lvl < 4; ++lvl, c *= 2) {
    const std::string l = std::to_string(lvl + 2);
    if ((rc = add_rconv(h.get(), tm, "backbone.fpn_lateral" + l, c, c, cfg->fpn_out, 1, 1, 0, false, 0, &h->lateral[lvl]))) return rc;
    if ((rc = add_rconv(h.get(), tm, "backbone.fpn_output" + l, cfg->fpn_out, cfg->fpn_out, cfg->fpn_out, 3, 1, 1, false, 0, &h->output[lvl]))) return rc;
  }
  const int f = cfg->fpn_out;
  if ((rc = add_rconv(h.get(), tm, "proposal_generator.rpn_head.conv", f, f, f, 3, 1, 1, false, 1, &h->rpn_conv))) return rc;
  if ((rc = add_rconv(h.get(), tm, "proposal_generator.rpn_head.objectness_logits", f, f, cfg->num_anchors, 1, 1, 0, false, 0, &h->rpn_obj))) return rc;
  if ((rc = add_rconv(h.get(), tm, "proposal_generator.rpn_head.anchor_deltas", f, f, cfg->num_anchors * 4, 1, 1, 0, false, 0, &h->rpn_delta))) return rc;
  if (tm.m.count("roi_heads.box_head.fc1.weight") && (rc = peanut_rcnn_build_heads(h.get(), tm))) return rc;
  PEANUT_HIP_CHECK(hipDeviceSynchronize());
  *out = h.release();
  return 0;
}

void peanut_rcnn_destroy(peanut_rcnn_t* h) { delete h; }

int peanut_rcnn_set_option(peanut_rcnn_t* h, const char* key, long long value) {
  if (!h) return fail(PEANUT_EINVAL, "peanut_rcnn_set_option: null handle");
  const int i = option_index(key);
  if (i < 0) return fail(PEANUT_EINVAL, std::string("peanut_rcnn_set_option: unknown option '") + (key ? key : "(null)") + "'");
  if (option_table()[i].upload_time)
    return fail(PEANUT_EINVAL, std::string("peanut_rcnn_set_option: option '") + option_table()[i].key +
                                   "' shapes the uploaded weights; set it with peanut_set_default_option before creating the handle");
  h->opts.v[i] = value;
  h->plans.clear();
  return 0;
}

int peanut_rcnn_plan(peanut_rcnn_t* h, int B, int H, int W, int resized_hw_out[2], int padded_hw_out[2], int level_hw[10],
                     size_t* workspace_bytes, double* flops_per_image) {
  if (!h) return fail(PEANUT_EINVAL, "null handle");
  OptionScope option_scope(&h->opts);
  RPlan* pl = get_rplan(h, B, H, W);
  if (!pl) return PEANUT_EINVAL;
  if (resized_hw_out) { resized_hw_out[0] = pl->nh; resized_hw_out[1] = pl->nw; }
  if (padded_hw_out) { padded_hw_out[0] = pl->Hp; padded_hw_out[1] = pl->Wp; }
  if (level_hw) for (int l = 0; l < 5; ++l) { level_hw[2 * l] = pl->lvl_h[l]; level_hw[2 * l + 1] = pl->lvl_w[l]; }
  if (workspace_bytes) *workspace_bytes = pl->bytes;
  if (flops_per_image) {
    double f = 0;
    for (const auto& op : pl->ops) f += op.flops;
    *flops_per_image = f / B;
  }
  return 0;
}

// the front end's launch sequence; with `events` (ops + 1 entries) a HIP event is recorded before the first and after every
// op, and `families` (ops entries) receives the kernel family each conv launch really picked (peanut_last_conv_kernel)
static int rcnn_front_impl(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, float* const* pyramid,
                           float* const* objectness, float* const* deltas, void* stream, hipEvent_t* events,
                           std::string* families) {
  RPlan* pl = get_rplan(h, B, H, W);
  if (!pl) return PEANUT_EINVAL;
  int rc;
  if ((rc = h->ws.ensure(pl->bytes))) return rc;
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)h->ws.p;
  auto P = [&](const Act& a) { return (float*)(base + a.off); };
  auto ext = [&](int slot) -> float* {
    if (slot < 0) return nullptr;
    if (slot < 5) return pyramid ? pyramid[slot] : nullptr;
    if (slot < 10) return objectness ? objectness[slot - 5] : nullptr;
    return deltas ? deltas[slot - 10] : nullptr;
  };
  // outputs the caller asked for are written in place (and read from there by the ops that consume them)
  auto OUT = [&](const ROp& op) -> float* { float* e = ext(op.ext_slot); return e ? e : P(op.out); };
  auto IN = [&](const ROp& op) -> float* { float* e = ext(op.in_ext_slot); return e ? e : P(op.in); };
  // the fused RPN chain needs every level's objectness (and deltas) in ONE buffer, level after level: true for the library's own
  // scratch and for peanut_rcnn_inference's buffers; a caller with scattered buffers gets the level-by-level launches
  bool rpn_fused = pl->rpn.on;
  if (rpn_fused) {
    const int A = h->rpn_obj->d.cout;
    for (int pass = 0; pass < 2 && rpn_fused; ++pass) {
      float* const* arr = pass == 0 ? objectness : deltas;
      const int w = pass == 0 ? A : 4 * A;
      if (!arr) continue;
      int given = 0;
      for (int l = 0; l < 5; ++l) given += arr[l] != nullptr;
      if (given == 0) continue;
      if (given != 5) { rpn_fused = false; break; }
      for (int l = 1; l < 5; ++l)
        if (arr[l] != arr[0] + (size_t)pl->rpn.row_off[l] * w) rpn_fused = false;
    }
  }
  if (events) PEANUT_HIP_CHECK(hipEventRecord(events[0], s));
  // side stream of the FPN output convs (not while probing: the per-op events time one stream)
  const bool use_side = !events && pl->splitk_side.bytes != 0;
  if (use_side && !h->side) {
    PEANUT_HIP_CHECK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    PEANUT_HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    PEANUT_HIP_CHECK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  }
  bool side_pending = false;
  size_t op_index = 0;
  DeferredSplit deferred{};        // handed from a conv1 that skipped its split-K reduce to the conv2 right behind it
  for (const auto& op : pl->ops) {
    const bool skip = (op.kind == R_RPN_FUSED && !rpn_fused) || (op.rpn_level && rpn_fused);
    if (op.join_side && side_pending) {
      PEANUT_HIP_CHECK(hipEventRecord(h->ev_join, h->side));
      PEANUT_HIP_CHECK(hipStreamWaitEvent(s, h->ev_join, 0));
      side_pending = false;
    }
    if (skip) {
      if (families) families[op_index] = "skipped";
      ++op_index;
      if (events) PEANUT_HIP_CHECK(hipEventRecord(events[op_index], s));
      continue;
    }
    switch (op.kind) {
      case R_RPN_FUSED: {
        const RpnFused& f = pl->rpn;
        const ConvLayer& L = *f.form;
        const int cin = L.d.cin, cout = L.d.cout, A = h->rpn_obj->d.cout;
        float* V = P(f.v);
        float* M = P(f.m);
        float* T = P(f.t_all);
        for (int l = 0; l < 5; ++l) {
          const float* xin = ext(l) ? ext(l) : P(f.in[l]);
          if ((rc = launch_wino_input(xin, V + (size_t)f.tile_off[l] * cin, B, f.in[l].H, f.in[l].W, cin, 1, s, f.gran, L.wino_m, f.m_pad_total)))
            return rc;
        }
        ConvArgs g{};
        g.x = V; g.y = M;
        g.B = 1; g.H = 1; g.W = (int)((long long)L.wino_np() * f.m_pad_total); g.c1 = cin; g.c2 = 0; g.Ho = 1; g.Wo = g.W;
        g.ws = P(pl->splitk); g.ws_floats = kSplitKScratchFloats;
        g.mt_per_group = (int)(f.m_pad_total / 128); g.w_group_stride = L.wino.rs ? L.wino_group_bytes : L.wino_group_floats;
        if ((rc = launch_conv(L.wino, g, s))) return rc;
        for (int l = 0; l < 5; ++l)
          if ((rc = launch_wino_output(M + (size_t)f.tile_off[l] * cout, L.d.scale, L.d.shift, nullptr, T + (size_t)f.row_off[l] * cout, B,
                                       f.in[l].H, f.in[l].W, cout, 1, L.d.relu, s, f.gran, L.wino_m, f.m_pad_total)))
            return rc;
        ConvArgs a{};
        a.x = T; a.B = 1; a.H = 1; a.W = (int)f.rows; a.c1 = cout; a.c2 = 0; a.Ho = 1; a.Wo = a.W;
        a.ws = P(pl->splitk); a.ws_floats = kSplitKScratchFloats;
        a.y = (objectness && objectness[0]) ? objectness[0] : P(f.obj_all);
        if ((rc = launch_conv(h->rpn_obj->d, a, s))) return rc;
        a.y = (deltas && deltas[0]) ? deltas[0] : P(f.dl_all);
        if ((rc = launch_conv(h->rpn_delta->d, a, s))) return rc;
        (void)A;
        break;
      }
      case R_PREPROCESS: {
        Norm3 nm;
        for (int k = 0; k < 3; ++k) { nm.mean[k] = h->cfg.pixel_mean[k]; nm.inv_std[k] = 1.0f / h->cfg.pixel_std[k]; }
        const long long total = (long long)B * pl->Hp * pl->Wp;   // one thread per input pixel in either layout
        const ResizeTab rz{(const int*)pl->rz_x.p, (const int*)pl->rz_y.p, pl->ksx, pl->ksy};
        if (h->stem_s2d)
          hipLaunchKernelGGL(rcnn_preprocess_s2d_kernel, dim3(grid_for(total)), dim3(256), 0, s, img_bgr, P(op.out), H, W, pl->nh,
                             pl->nw, pl->Hp, pl->Wp, nm, rz, total);
        else
          hipLaunchKernelGGL(rcnn_preprocess_kernel, dim3(grid_for(total)), dim3(256), 0, s, img_bgr, P(op.out), H, W, pl->nh,
                             pl->nw, pl->Hp, pl->Wp, nm, rz, total);
        break;
      }
      case R_CONV: {
        ConvArgs a{};
        a.x = IN(op);
        a.res = op.has_res ? P(op.res) : nullptr;
        a.y = OUT(op);
        a.B = op.in.B; a.H = op.in.H; a.W = op.in.W; a.c1 = op.in.C; a.c2 = 0; a.Ho = op.out.H; a.Wo = op.out.W;
        if (op.has_in2) { a.x2 = P(op.in2); a.c2 = op.in2.C; }
        a.ws = P(pl->splitk); a.ws_floats = kSplitKScratchFloats;
        hipStream_t os = s;
        if (op.side && use_side) {      // behind everything enqueued so far, next to what follows
          PEANUT_HIP_CHECK(hipEventRecord(h->ev_fork, s));
          PEANUT_HIP_CHECK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
          os = h->side;
          a.ws = P(pl->splitk_side); a.ws_floats = kSplitKSideFloats;
          side_pending = true;
        }
        DeferredSplit* produced = deferred.valid ? &deferred : nullptr;     // (the previous op's, consumed by this one)
        if (op.defer_ok && !produced) a.defer = &deferred;
        if ((rc = launch_conv_layer(*op.conv, a, op.has_wino ? P(op.wino_v) : nullptr, op.has_wino ? P(op.wino_m) : nullptr, os, produced))) return rc;
        break;
      }
      case R_MAXPOOL:
        if ((rc = launch_maxpool3x3s2(P(op.in), P(op.out), op.in.B, op.in.H, op.in.W, op.in.C, op.out.H, op.out.W, s))) return rc;
        break;
      case R_ADD_UP: {
        if (op.out.H != op.in.H * 2 || op.out.W != op.in.W * 2) return fail(PEANUT_EINVAL, "rcnn: FPN levels are not exact 2x multiples");
        const long long total = (long long)op.out.B * op.out.H * op.out.W * (op.out.C / 4);
        hipLaunchKernelGGL(add_upsampled2x_kernel, dim3(grid_for(total)), dim3(256), 0, s, P(op.out), P(op.in), op.out.H,
                           op.out.W, op.out.C, total);
        break;
      }
      case R_SUBSAMPLE: {
        const long long total = (long long)op.out.B * op.out.H * op.out.W * (op.out.C / 4);
        hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for(total)), dim3(256), 0, s, IN(op), OUT(op), op.in.H, op.in.W,
                           op.in.C, op.out.H, op.out.W, total);
        break;
      }
    }
    if (families) {
      static const char* const kind_names[] = {"rcnn_preprocess", "conv", "maxpool", "fpn_add_upsampled", "subsample2", "wino+rpn_fused"};
      families[op_index] = op.kind == R_CONV ? std::string(op.has_wino ? "wino+" : "") + noted_kernel() : kind_names[op.kind];
    }
    ++op_index;
    if (events) PEANUT_HIP_CHECK(hipEventRecord(events[op_index], s));
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PEANUT_EHIP, std::string("peanut_rcnn_forward_front: ") + hipGetErrorString(e));
  return 0;
}

int peanut_rcnn_forward_front(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, float* const* pyramid,
                              float* const* objectness, float* const* deltas, void* stream) {
  if (!h || !img_bgr) return fail(PEANUT_EINVAL, "peanut_rcnn_forward_front: null argument");
  OptionScope option_scope(&h->opts);
  return rcnn_front_impl(h, img_bgr, B, H, W, pyramid, objectness, deltas, stream, nullptr, nullptr);
}

int peanut_rcnn_probe_front(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, int reps, int max_ops,
                            const char** names, const char** kernels, double* ms, double* flops, void* stream) {
  if (!h || !img_bgr || reps < 1) return fail(PEANUT_EINVAL, "peanut_rcnn_probe_front: bad argument");
  OptionScope option_scope(&h->opts);
  RPlan* pl = get_rplan(h, B, H, W);
  if (!pl) return PEANUT_EINVAL;
  const size_t n = pl->ops.size();
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev) PEANUT_HIP_CHECK(hipEventCreate(&e));
  h->probe_families.assign(n, std::string());
  std::vector<double> sum(n, 0.0);
  int rc = 0;
  for (int r = 0; r < reps && !rc; ++r) {
    if ((rc = rcnn_front_impl(h, img_bgr, B, H, W, nullptr, nullptr, nullptr, stream, ev.data(), h->probe_families.data()))) break;
    if (hipEventSynchronize(ev[n]) != hipSuccess) { rc = fail(PEANUT_EHIP, "peanut_rcnn_probe_front: event synchronise failed"); break; }
    for (size_t i = 0; i < n; ++i) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, ev[i], ev[i + 1]);
      sum[i] += t;
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (rc) return rc;
  for (size_t i = 0; i < n && (int)i < max_ops; ++i) {
    if (names) names[i] = pl->ops[i].name.c_str();
    if (kernels) kernels[i] = h->probe_families[i].c_str();
    if (ms) ms[i] = sum[i] / reps;
    if (flops) flops[i] = pl->ops[i].flops;
  }
  return (int)n;
}

int peanut_rcnn_preprocess(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, float* out_nchw, void* stream) {
  if (!h || !img_bgr || !out_nchw) return fail(PEANUT_EINVAL, "peanut_rcnn_preprocess: null argument");
  OptionScope option_scope(&h->opts);
  RPlan* pl = get_rplan(h, B, H, W);
  if (!pl) return PEANUT_EINVAL;
  Norm3 nm;
  for (int k = 0; k < 3; ++k) { nm.mean[k] = h->cfg.pixel_mean[k]; nm.inv_std[k] = 1.0f / h->cfg.pixel_std[k]; }
  const long long total = (long long)B * pl->Hp * pl->Wp;
  const ResizeTab rz{(const int*)pl->rz_x.p, (const int*)pl->rz_y.p, pl->ksx, pl->ksy};
  hipLaunchKernelGGL(rcnn_preprocess_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img_bgr, out_nchw, H, W, pl->nh,
                     pl->nw, pl->Hp, pl->Wp, nm, rz, total);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("peanut_rcnn_preprocess: ") + hipGetErrorString(e));
}

}  // extern "C"
