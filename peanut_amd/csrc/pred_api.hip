// C-ABI entry points of the map-prediction forward (see include/peanut_hip.h) and the host-side
// planner that turns the reference's module graph into one static launch sequence per (B,H,W).
//
// Graph followed (reference files under prediction/mmseg/models):
//   backbones/resnet.py:659-674   stem -> maxpool -> layer1..4
//   backbones/resnet.py:267-307   Bottleneck: 1x1 -> 3x3(stride,dilation) -> 1x1, (+downsample), add, ReLU
//   utils/res_layer.py:43-95      first block of a stage: stride + 1x1 downsample, contracted dilation
//   decode_heads/psp_head.py:48-59,95-117   PPM, concat, 3x3 bottleneck, cls_seg
//   segmentors/encoder_decoder.py:70-80     final bilinear resize to the input size
#include <math.h>
#include <string.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "net_common.h"

namespace peanut {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
static thread_local const char* g_kernel = "";
void note_kernel(const char* family) { g_kernel = family; }
const char* noted_kernel() { return g_kernel; }

namespace {

enum OpKind { OP_TO_NHWC, OP_CONV, OP_CONV_NCHW, OP_MAXPOOL, OP_PPM_POOL, OP_PPM_UP, OP_PPM_TERM, OP_UPSAMPLE, OP_WINO_IN, OP_WINO_GEMM, OP_WINO_OUT };

struct Op {
  OpKind kind;
  std::string name;
  std::string kernel;   // kernel symbol family the op launches (groups ops in profiles)
  const ConvLayer* conv = nullptr;
  Act in, in2, res, out;
  bool has_in2 = false, has_res = false;
  double flops = 0;
  double bytes = 0;            // algorithmic HBM bytes: every operand read once, the result written once
  int wino_mt_per_group = 0;   // OP_WINO_GEMM: 128-row tiles per Winograd position
  int wino_valid_rows = 0;     // OP_WINO_GEMM: tiles per position before the padding to whole GEMM tiles
  int wino_gran = 128;         // row padding of the Winograd position GEMMs
  // two-stream schedule of the PSP head (build_plan): branch 1 ops run on the handle's side stream; the first of them
  // waits for everything enqueued so far on the caller's stream (fork), join_before makes the caller's stream wait for
  // the side stream before this op
  int branch = 0;
  bool fork_before = false, join_before = false;
  // grouped pointwise conv (the per-scale PSP convs in one launch): m-tiles per weight group; the PSP kernels then
  // address scale s at row s * ppm_scale_rows
  int group_mt = 0;
  int ppm_scale_rows = 0;
  bool defer_ok = false;       // OP_CONV: the next op is the Winograd input transform that alone reads this output and can sum split-K partial
                               // tiles itself (common.h: DeferredSplit): this conv may skip its split-K reduce (round 6, option defer_splitk)
};

struct Plan {
  int B = 0, H = 0, W = 0;
  Act splitk;   // scratch for tail split-K partial tiles, alive for the whole forward
  Act splitk2;  // the same for convs on the side stream (valid when the plan has branch 1 ops)
  bool keep_all = false;
  size_t bytes = 0;
  std::vector<Op> ops;
  std::map<std::string, Act> named;
};

}  // namespace
}  // namespace peanut

using namespace peanut;

struct peanut_conv {
  Options opts = default_options();   // this handle's tuning options (options.h): snapshot of the process defaults at creation
  ConvLayer L;
  DevBuf ws;   // tail split-K scratch, allocated on first forward
  DevBuf wino_v, wino_m;   // Winograd scratch, grown on demand
};

struct peanut_pred {
  Options opts = default_options();   // this handle's tuning options (options.h): snapshot of the process defaults at creation
  DeferredSplit deferred{};           // run time: handed from a conv1 that skipped its split-K reduce to the Winograd input transform behind it
  peanut_pred_cfg cfg{};
  int cin_pad = 0;
  std::vector<std::unique_ptr<ConvLayer>> convs;
  // structure
  ConvLayer* stem[3] = {nullptr, nullptr, nullptr};
  struct Block { ConvLayer *c1, *c2, *c3, *down, *c3d; };   // c3d: conv3 and the stride-1 downsample as one two-source GEMM (add_fused_c3d)
  std::vector<std::vector<Block>> layers;
  std::vector<ConvLayer*> ppm;
  ConvLayer* bottleneck = nullptr;     // unfolded form: 3x3 over cat([x, ppm...])
  ConvLayer* bottleneck_x = nullptr;   // folded form: 3x3 over x only ...
  std::vector<ConvLayer*> ppm_q;       // ... plus the per-scale tables Q_s = W_s (x) p_s (1x1 convs, cout = 9*hc)
  // the per-scale 1x1 convs as ONE grouped GEMM each (weights / scale / shift of the scales back to back), used at small
  // batch where four tiny launches + four reduces cost more than the padded rows (build_grouped)
  std::unique_ptr<ConvLayer> ppm_grouped, ppmq_grouped;
  ConvLayer* conv_seg = nullptr;
  int feat_channels = 0;
  // runtime
  bool keep_all = false;
  std::map<std::string, std::unique_ptr<Plan>> plans;
  Plan* last_plan = nullptr;
  DevBuf ws;
  bool use_graph = false;   // peanut_pred_use_graph: replay each (shape, buffers) launch sequence as a hipGraph
  GraphCache graphs;
  // event probe (bench.py roofline): per-op HIP events recorded inside forward
  bool probe = false;
  Plan* probe_plan = nullptr;
  std::vector<std::vector<hipEvent_t>> probe_events;  // one vector (n_ops + 1 events) per forward
  std::vector<hipEvent_t> event_pool;
  // side stream of the two-stream PSP-head schedule (created on first use) and its fork / join events
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  ~peanut_pred() {
    for (auto& v : probe_events) for (auto e : v) (void)hipEventDestroy(e);
    for (auto e : event_pool) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (side) (void)hipStreamDestroy(side);
  }
};

namespace {

int bn_fold(peanut_pred* h, const TensorMap& tm, const std::string& bn, int cout, float* scale, float* shift) {
  return bn_fold_eps(tm, bn, cout, h->cfg.bn_eps, scale, shift);
}

// conv weight + (BatchNorm | bias) -> ConvLayer with folded scale/shift.
// BN(eval): y = x*alpha + beta, alpha = weight/sqrt(var+eps), beta = bias - mean*alpha  (fp32, the
// form ATen's CPU batch_norm inference uses).
int add_conv(peanut_pred* h, const TensorMap& tm, const std::string& conv, const std::string& bn, int cin,
             int cin_pad, int cout, int k, int stride, int pad, int dil, int relu, ConvLayer** out) {
  int rc = 0;
  const int64_t wshape[4] = {cout, cin, k, k};
  const peanut_tensor* w = tm.get(conv + ".weight", 4, wshape, &rc);
  if (!w) return rc;
  std::vector<float> scale(cout, 1.f), shift(cout, 0.f);
  const int64_t cshape[1] = {cout};
  if (!bn.empty()) {
    if ((rc = bn_fold(h, tm, bn, cout, scale.data(), shift.data()))) return rc;
  } else {
    const peanut_tensor* b = tm.get(conv + ".bias", 1, cshape, &rc); if (!b) return rc;
    for (int n = 0; n < cout; ++n) shift[n] = b->data[n];
  }
  auto L = std::make_unique<ConvLayer>();
  L->name = conv;
  rc = upload_conv(*L, w->data, scale.data(), shift.data(), cout, cin, cin_pad, k, k, stride, pad, dil, relu,
                   h->cfg.precision);
  if (rc) return rc;
  if (h->cfg.conv_algo == PEANUT_ALGO_AUTO && wino_eligible(cin_pad, cout, k, k, stride, pad, dil, h->cfg.precision, 64) &&
      (rc = upload_wino_forms(*L, w->data, scale.data(), shift.data(), cout, cin, cin_pad, pad, dil, relu, h->cfg.precision, 0)))
    return rc;
  *out = L.get();
  h->convs.push_back(std::move(L));
  return 0;
}

// conv3 + BN3 and the block's stride-1 downsample conv + BN as ONE pointwise layer over the channel concatenation
// [conv2 output | block input] (resnet.py:289-305: out = bn3(conv3(t2)) + bn_d(conv_d(x)), then ReLU): the BN scales are
// folded into the weights (W' = alpha[n] * W[n][c]), the shifts add up.  One launch instead of two, and the identity
// tensor (as large as the block output) is neither written nor read back.
int add_fused_c3d(peanut_pred* h, const TensorMap& tm, const std::string& blk, int planes, int inplanes, int cout, ConvLayer** out) {
  int rc = 0;
  const int64_t w3s[4] = {cout, planes, 1, 1}, wds[4] = {cout, inplanes, 1, 1};
  const peanut_tensor* w3 = tm.get(blk + ".conv3.weight", 4, w3s, &rc);
  if (!w3) return rc;
  const peanut_tensor* wd = tm.get(blk + ".downsample.0.weight", 4, wds, &rc);
  if (!wd) return rc;
  std::vector<float> s3(cout), b3(cout), sd(cout), bd(cout);
  if ((rc = bn_fold(h, tm, blk + ".bn3", cout, s3.data(), b3.data()))) return rc;
  if ((rc = bn_fold(h, tm, blk + ".downsample.1", cout, sd.data(), bd.data()))) return rc;
  const int cin = planes + inplanes;
  std::vector<float> w((size_t)cout * cin), shift(cout);
  for (int n = 0; n < cout; ++n) {
    for (int c = 0; c < planes; ++c) w[(size_t)n * cin + c] = s3[n] * w3->data[(size_t)n * planes + c];
    for (int c = 0; c < inplanes; ++c) w[(size_t)n * cin + planes + c] = sd[n] * wd->data[(size_t)n * inplanes + c];
    shift[n] = b3[n] + bd[n];
  }
  auto L = std::make_unique<ConvLayer>();
  L->name = blk + ".conv3+downsample";
  if ((rc = upload_conv(*L, w.data(), nullptr, shift.data(), cout, cin, cin, 1, 1, 1, 0, 1, 1,
                        rs_planes_of(h->cfg.precision) ? h->cfg.precision : PEANUT_PREC_FP32))) return rc;
  *out = L.get();
  h->convs.push_back(std::move(L));
  return 0;
}

// One layer object whose weight / scale / shift buffers hold those of `parts` (identical shapes and tilings, fp32
// path) back to back: a grouped GEMM then takes m-tile group g against block g (ConvArgs::mt_per_group,
// w_group_stride, ss_group_stride).  Null when the parts do not qualify.
std::unique_ptr<ConvLayer> build_grouped(const std::vector<ConvLayer*>& parts, const std::string& name, int* rc) {
  *rc = 0;
  if (parts.size() < 2) return nullptr;
  const ConvDesc& d0 = parts[0]->d;
  for (const ConvLayer* L : parts)
    if (L->d.kh != 1 || L->d.kw != 1 || L->d.cin != d0.cin || L->d.cout != d0.cout || L->d.bn_tile != d0.bn_tile ||
        L->d.bk != d0.bk || L->d.relu != d0.relu || L->w.bytes != parts[0]->w.bytes || L->ss.bytes != parts[0]->ss.bytes)
      return nullptr;
  auto G = std::make_unique<ConvLayer>();
  G->name = name;
  G->d = d0;
  G->cin_real = parts[0]->cin_real;
  const size_t wb = parts[0]->w.bytes, sb = (size_t)d0.cout_pad * sizeof(float), n = parts.size();
  if ((*rc = G->w.ensure(wb * n)) || (*rc = G->ss.ensure(2 * sb * n))) return nullptr;
  for (size_t i = 0; i < n; ++i) {
    if (hipMemcpy((char*)G->w.p + i * wb, parts[i]->w.p, wb, hipMemcpyDeviceToDevice) != hipSuccess ||
        hipMemcpy((char*)G->ss.p + i * sb, parts[i]->d.scale, sb, hipMemcpyDeviceToDevice) != hipSuccess ||
        hipMemcpy((char*)G->ss.p + (n + i) * sb, parts[i]->d.shift, sb, hipMemcpyDeviceToDevice) != hipSuccess) {
      *rc = fail(PEANUT_EHIP, name + ": grouping the per-scale layers failed");
      return nullptr;
    }
  }
  G->d.w_packed = (const float*)G->w.p;
  G->d.scale = (const float*)G->ss.p;
  G->d.shift = (const float*)G->ss.p + n * d0.cout_pad;
  G->d.w_s = nullptr;
  G->d.rs = 0;      // the grouped form runs the fp32 kernel on the parts' fp32-packed weights
  return G;
}

// the kernel symbol family launch_conv will pick (conv_igemm.hip: launch_conv)
std::string conv_kernel_name(const ConvDesc& d, bool two_source = false, long long M = 0, int mt_per_group = 0, const Act* out = nullptr,
                             bool has_res = false) {
  if (out && !two_source && !has_res) {      // conv_patch.hip: the stem's 3x3 convs
    ConvArgs a{};
    a.B = out->B; a.Ho = out->H; a.Wo = out->W;
    if (conv_patch_eligible(d, a)) return conv_patch_kernel_name(d, false);
  }
  if (d.rs == 2) return std::string(d.s_planes == 3 ? "conv_rs6_128x" : (d.s_planes == 4 ? "conv_rs3h_128x" : "conv_rs3_128x")) + std::to_string(d.bn_tile);
  if (d.rs) return gemm_rs_kernel_name(d.cout, M, mt_per_group, d.bn_tile, d.cin, d.s_planes);
  if (d.bk == 32 && d.kh == 1 && d.kw == 1 && d.pad == 0 && (!two_source || d.stride == 1) && d.cin % 32 == 0 && conv_pw_enabled())
  {
    const int flush = d.flush_ch / d.bk;
    if (conv_pw_uses_256wp(d.cout, M, d.stride, mt_per_group, d.bn_tile, d.cin, 0, flush))
      return "conv_pw_glds_256x256p";
    if (conv_pw_uses_ares(d.cin, d.cout, M, d.stride, two_source, flush, d.bn_tile)) return "conv_pw_ares_128x128";
    if (conv_pw_uses_256w(d.cout, M, mt_per_group, d.bn_tile, d.cin, flush)) return "conv_pw_glds_256x256";
    if (conv_pw_uses_256p(d.cout, M, mt_per_group, d.bn_tile, d.cin, flush, M * d.stride * d.stride)) return "conv_pw_glds_256x128p";
    if (conv_pw_uses_256(d.cout, M, mt_per_group, d.bn_tile, d.cin)) return "conv_pw_glds_256x128";
    if (conv_pw_narrow_tiles(d.cin, d.cout, M, d.bn_tile, mt_per_group)) return "conv_pw_glds_128x64";
    return "conv_pw_glds_128x" + std::to_string(d.bn_tile);
  }
  return "conv_igemm_128x" + std::to_string(d.bn_tile) + "x" + std::to_string(d.bk);
}

// One conv layer into the plan.  `ar` is only needed for layers that carry a Winograd form (scratch for the transformed
// tensors).
void push_conv(Plan& pl, const ConvLayer* L, const Act& in, const Act* in2, const Act* res, const Act& out, Arena* ar = nullptr) {
  const bool wino = L->has_wino && !in2 && ar && (long long)in.B * in.H * in.W >= wino_min_pixels(L->d.cin);
  if (wino) L = wino_pick_form(L, in.B, in.H, in.W);
  if (wino) {
    // V = B^T d B  ->  36 grouped GEMMs  ->  A^T M A + BN/residual/ReLU   (winograd.hip)
    int th, tw;
    long long n_tiles, m_pad;
    const bool rs_gemm = L->wino.rs != 0;
    const int gran = wino_gran_for(*L, in.B, in.H, in.W);
    wino_geometry(in.B, in.H, in.W, L->d.dil, &th, &tw, &n_tiles, &m_pad, gran, L->wino_m);
    const double np = (double)L->wino_np();
    const long long npl = L->wino_np();
    Act v = make_act(*ar, 1, 1, (int)(npl * m_pad), in.C);
    Act m = make_act(*ar, 1, 1, (int)(npl * m_pad), L->d.cout);
    const std::string fam = L->wino_m == 6 ? "wino6_" : (L->wino_m == 5 ? "wino5_" : "wino_");   // op names / kernel families tell the forms apart
    const std::string tag = "[" + fam;
    Op a; a.kind = OP_WINO_IN; a.name = L->name + tag + "in]"; a.kernel = fam + "input"; a.conv = L; a.in = in; a.out = v;
    a.wino_gran = gran;
    a.bytes = (double)in.bytes + np * (double)n_tiles * in.C * 4.0;
    pl.ops.push_back(a);
    Op g; g.kind = OP_WINO_GEMM; g.name = L->name + tag + "gemm]"; g.kernel = conv_kernel_name(L->wino, false, npl * m_pad, (int)(m_pad / 128)); g.conv = L;
    g.in = v; g.in.W = (int)(npl * m_pad); g.in.C = in.C; g.out = m; g.wino_mt_per_group = (int)(m_pad / 128); g.wino_gran = gran; g.wino_valid_rows = (int)n_tiles;
    g.flops = 2.0 * np * (double)m_pad * L->d.cout * L->cin_real;     // executed, not the direct-form count
    g.bytes = np * (double)m_pad * (in.C * 4.0 + L->d.cout * 4.0) +
              np * (rs_gemm ? (double)L->wino_group_bytes : (double)L->wino_group_floats * 4);
    pl.ops.push_back(g);
    Op o; o.kind = OP_WINO_OUT; o.name = L->name + tag + "out]"; o.kernel = fam + "output"; o.conv = L; o.in = m; o.out = out;
    if (res) { o.res = *res; o.has_res = true; }
    o.wino_gran = gran;
    o.bytes = np * (double)n_tiles * L->d.cout * 4 + (double)out.bytes + (res ? (double)out.bytes : 0.0);
    pl.ops.push_back(o);
    ar->release(v.off, v.bytes);
    ar->release(m.off, m.bytes);
    return;
  }
  Op op;
  op.kind = OP_CONV; op.name = L->name; op.conv = L; op.in = in; op.out = out;
  op.kernel = conv_kernel_name(L->d, in2 != nullptr, (long long)out.B * out.H * out.W, 0, &out, res != nullptr);
  if (in2) { op.in2 = *in2; op.has_in2 = true; }
  if (res) { op.res = *res; op.has_res = true; }
  op.flops = conv_flops(L, out);
  op.bytes = (double)in.bytes + (in2 ? (double)in2->bytes : 0.0) + (double)out.bytes + (res ? (double)out.bytes : 0.0) +
             (L->d.rs == 2 ? (double)sx_conv_packed_bytes(L->d.cin, L->d.cout, L->d.kh, L->d.kw, L->d.bn_tile, L->d.s_planes)
              : L->d.rs ? (double)sx_packed_bytes(L->d.cin, L->d.cout, L->d.bn_tile, L->d.s_planes)
                      : (double)conv_packed_floats(L->d.cin, L->d.cout, L->d.kh, L->d.kw, L->d.bn_tile) * 4);
  pl.ops.push_back(op);
}

// Build the static launch sequence + workspace layout for one input shape.
std::unique_ptr<Plan> build_plan(const peanut_pred* h, int B, int H, int W) {
  auto pl = std::make_unique<Plan>();
  pl->B = B; pl->H = H; pl->W = W; pl->keep_all = h->keep_all;
  Arena ar;
  ar.keep_all = h->keep_all;
  auto rel = [&](const Act& t) { ar.release(t.off, t.bytes); };

  pl->splitk.bytes = kSplitKScratchFloats * sizeof(float);
  pl->splitk.off = ar.alloc(pl->splitk.bytes);
  pl->splitk.B = pl->splitk.H = pl->splitk.W = 1; pl->splitk.C = 0;
  // input layout change: NCHW -> NHWC, channels zero-padded to a multiple of 16 -- unless the first stem conv reads the NCHW
  // input itself (conv_patch.hip, option stem_nchw): one pass over the input instead of two and a half
  const ConvDesc& d0 = h->stem[0]->d;
  const int ho0 = conv_out_dim(H, d0.kh, d0.stride, d0.pad, d0.dil), wo0 = conv_out_dim(W, d0.kw, d0.stride, d0.pad, d0.dil);
  const bool nchw_stem = conv_patch_nchw_eligible(d0, B, ho0, wo0, h->cfg.in_channels);
  Act x{};
  x.B = B; x.H = H; x.W = W; x.C = h->cin_pad;
  if (!nchw_stem) {
    x = make_act(ar, B, H, W, h->cin_pad);
    Op op; op.kind = OP_TO_NHWC; op.name = "nchw_to_nhwc"; op.kernel = "nchw_to_nhwc"; op.out = x;
    op.bytes = (double)B * H * W * h->cfg.in_channels * 4.0 + (double)x.bytes;      // HBM-bound: read NCHW, write padded NHWC
    pl->ops.push_back(op);
  }

  // deep stem (resnet.py:591-624) + maxpool (:638)
  for (int i = 0; i < 3; ++i) {
    const ConvDesc& d = h->stem[i]->d;
    Act y = make_act(ar, B, conv_out_dim(x.H, d.kh, d.stride, d.pad, d.dil),
                     conv_out_dim(x.W, d.kw, d.stride, d.pad, d.dil), d.cout);
    if (i == 0 && nchw_stem) {
      Op op; op.kind = OP_CONV_NCHW; op.name = h->stem[0]->name; op.kernel = conv_patch_kernel_name(d, true); op.conv = h->stem[0];
      op.in = x; op.out = y;
      op.flops = conv_flops(h->stem[0], y);
      op.bytes = (double)B * H * W * h->cfg.in_channels * 4.0 + (double)y.bytes + (double)conv_packed_floats(d.cin, d.cout, d.kh, d.kw, d.bn_tile) * 4;
      pl->ops.push_back(op);
    } else {
      push_conv(*pl, h->stem[i], x, nullptr, nullptr, y);
    }
    pl->named["stem" + std::to_string(i)] = y;
    if (!(i == 0 && nchw_stem)) rel(x);
    x = y;
  }
  {
    Act y = make_act(ar, B, conv_out_dim(x.H, 3, 2, 1, 1), conv_out_dim(x.W, 3, 2, 1, 1), x.C);
    Op op; op.kind = OP_MAXPOOL; op.name = "maxpool"; op.kernel = "maxpool"; op.in = x; op.out = y;
    op.bytes = (double)x.bytes + (double)y.bytes;
    pl->ops.push_back(op);
    pl->named["pool"] = y;
    rel(x);
    x = y;
  }
  // residual stages (the same plan in every precision mode: activations are fp32 throughout)
  for (size_t li = 0; li < h->layers.size(); ++li) {
    for (size_t bi = 0; bi < h->layers[li].size(); ++bi) {
      const auto& blk = h->layers[li][bi];
      const ConvDesc& d2 = blk.c2->d;
      Act t1 = make_act(ar, B, x.H, x.W, blk.c1->d.cout);
      push_conv(*pl, blk.c1, x, nullptr, nullptr, t1);
      const int h2 = conv_out_dim(x.H, 3, d2.stride, d2.pad, d2.dil), w2 = conv_out_dim(x.W, 3, d2.stride, d2.pad, d2.dil);
      Act t2 = make_act(ar, B, h2, w2, d2.cout);
      const size_t c1_idx = pl->ops.size() - 1;
      push_conv(*pl, blk.c2, t1, nullptr, nullptr, t2, &ar);
      {   // conv1's output is read by conv2 alone: its Winograd input transform may sum conv1's split-K partial tiles (common.h: DeferredSplit)
        const Op& win = pl->ops[c1_idx + 1];
        if (opt(OPT_DEFER_SPLITK) != 0 && !h->keep_all && win.kind == OP_WINO_IN && pl->ops[c1_idx].kind == OP_CONV && !win.conv->d.rs &&
            wino_input_accepts_deferred(t1.B, t1.H, t1.W, t1.C, win.conv->d.dil, win.wino_gran, win.conv->wino_m))
          pl->ops[c1_idx].defer_ok = true;
      }
      rel(t1);
      Act idn = x;
      bool own_idn = false;
      const bool fused_c3d = blk.c3d != nullptr && !h->keep_all;   // debug plans keep the two separate layers
      if (blk.down && !fused_c3d) {
        idn = make_act(ar, B, h2, w2, blk.down->d.cout);
        push_conv(*pl, blk.down, x, nullptr, nullptr, idn);
        own_idn = true;
      }
      Act y = make_act(ar, B, h2, w2, blk.c3->d.cout);
      if (fused_c3d) push_conv(*pl, blk.c3d, t2, &x, nullptr, y);     // [t2 | x] x [W3'; Wd'] + shifts, ReLU: conv3 and the downsample in one GEMM
      else push_conv(*pl, blk.c3, t2, nullptr, &idn, y);              // BN3 + identity + ReLU fused (resnet.py:289-305)
      rel(t2);
      if (own_idn) rel(idn);
      rel(x);
      x = y;
    }
    pl->named["layer" + std::to_string(li + 1)] = x;
  }
  // PSP head.  In the folded form the pyramid branch (pooling, the per-scale 1x1 convs, the Q tables and the 9-tap
  // term R: small, latency- or HBM-bound kernels) only meets the 3x3 bottleneck conv over x at that conv's residual
  // input, so it runs on a side stream underneath the bottleneck's Winograd input transform and GEMM.  Its buffers
  // then stay allocated until the bottleneck is planned (nothing of the main branch may reuse them) and its convs
  // get their own split-K scratch.  Measured (profiles/r2m): +0.6 % at batch 32, +0.8 % on one 720 x 720 map, -3 % on
  // one 240 x 240 map (the GEMM is then shorter than the branch and the two event waits cost more than they hide), so
  // the branch only forks when the GEMM has >= 4096 rows.  PEANUT_PPM_OVERLAP=0 / 1 forces it off / on.
  const int overlap_env = (int)opt(OPT_PPM_OVERLAP);
  const bool overlap = h->bottleneck_x != nullptr && (overlap_env >= 0 ? overlap_env == 1 : (long long)B * x.H * x.W >= 4096);
  const size_t side_first = pl->ops.size();
  if (overlap) {
    pl->splitk2.bytes = kSplitKScratchFloats * sizeof(float);
    pl->splitk2.off = ar.alloc(pl->splitk2.bytes);
    pl->splitk2.B = pl->splitk2.H = pl->splitk2.W = 1; pl->splitk2.C = 0;
  }
  std::vector<Act> side_bufs;   // released after the bottleneck when the branch overlaps it
  auto rel_side = [&](const Act& t) { if (overlap) side_bufs.push_back(t); else rel(t); };
  int nbins = 0, kmax = 0;
  for (int i = 0; i < h->cfg.n_pool_scales; ++i) {
    nbins += h->cfg.pool_scales[i] * h->cfg.pool_scales[i];
    kmax = std::max(kmax, h->cfg.pool_scales[i]);
  }
  // Small batches: the per-scale 1x1 convs (and the per-scale Q tables) as ONE grouped GEMM each -- every scale padded
  // to the same `srows` rows (whole 128-row tiles), block s of the grouped weights for scale s.  Four launches + four
  // split-K reduces of 15 + 6 us become one of each (batch 1: 170 -> 45 us for the eight convs); taken while the padded
  // rows stay within 2x of what the separate launches would compute.  Not in debug (keep_all) plans, whose named
  // tensors keep the packed layout.
  int srows = 0;
  if (h->ppm_grouped && h->ppmq_grouped && !h->keep_all) {
    const int cand = (B * kmax * kmax + 127) / 128 * 128;
    long long separate = 0;
    for (int i = 0; i < h->cfg.n_pool_scales; ++i) separate += ((long long)B * h->cfg.pool_scales[i] * h->cfg.pool_scales[i] + 127) / 128 * 128;
    const int env = (int)opt(OPT_PPM_GROUPED);
    if (env != 0 && ((long long)h->cfg.n_pool_scales * cand <= 2 * separate || env == 1)) srows = cand;
  }
  const int prow = srows ? h->cfg.n_pool_scales * srows : nbins * B;       // rows of pooled / table / q
  Act pooled = make_act(ar, 1, 1, prow, x.C);
  {
    Op op; op.kind = OP_PPM_POOL; op.name = "ppm_pool"; op.kernel = "ppm_pool"; op.in = x; op.out = pooled; op.ppm_scale_rows = srows;
    op.bytes = (double)x.bytes + (double)pooled.bytes;
    const size_t sf = ppm_pool_scratch_floats(B, x.H, x.C, h->cfg.pool_scales, h->cfg.n_pool_scales);
    if (sf) {   // per-row partial sums of the two-pass pooling
      Act scr; scr.B = 1; scr.H = 1; scr.W = 1; scr.C = 0; scr.bytes = sf * sizeof(float); scr.off = ar.alloc(scr.bytes);
      op.in2 = scr; op.has_in2 = true;
      pl->ops.push_back(op);
      rel_side(scr);
    } else {
      pl->ops.push_back(op);
    }
  }
  Act table = make_act(ar, 1, 1, prow, h->cfg.head_channels);
  // pooled/table are SCALE-MAJOR [scale][B][k*k][C] (pspnet_aux.hip: ppm_pool_kernel), so the 1x1 conv
  // of each scale (psp_head.py:39-46) runs on one contiguous [B*k*k, C] matrix.
  auto push_grouped = [&](const ConvLayer* G, const Act& in, const Act& out) {
    push_conv(*pl, G, in, nullptr, nullptr, out);
    Op& op = pl->ops.back();
    op.group_mt = srows / 128;
    op.flops = 2.0 * (double)prow * G->d.cout * G->cin_real;      // executed (padded rows included)
  };
  if (srows) {
    push_grouped(h->ppm_grouped.get(), pooled, table);
  } else {
    size_t row0 = 0;
    for (int i = 0; i < h->cfg.n_pool_scales; ++i) {
      const int k = h->cfg.pool_scales[i];
      Act in = pooled, out = table;
      in.off = pooled.off + row0 * (size_t)x.C * sizeof(float);
      in.H = 1; in.W = k * k; in.C = x.C; in.B = B;
      out.off = table.off + row0 * (size_t)h->cfg.head_channels * sizeof(float);
      out.H = 1; out.W = k * k; out.C = h->cfg.head_channels; out.B = B;
      push_conv(*pl, h->ppm[i], in, nullptr, nullptr, out);
      row0 += (size_t)B * k * k;
    }
  }
  pl->named["ppm_table"] = table;
  rel_side(pooled);
  Act bt;
  int term_scale_rows = 0;
  if (h->bottleneck_x) {
    // folded pyramid half (pspnet_aux.hip: ppm_conv_term_kernel): Q_s = table_s x W_s, then the 9-tap
    // bilinear evaluation R, then the 3x3 conv over x alone with R as its residual term
    const int hc = h->cfg.head_channels;
    Act q = make_act(ar, 1, 1, prow, 9 * hc);
    if (srows) {
      push_grouped(h->ppmq_grouped.get(), table, q);
      term_scale_rows = srows;
    } else {
      size_t row0 = 0;
      for (int i = 0; i < h->cfg.n_pool_scales; ++i) {
        const int k = h->cfg.pool_scales[i];
        Act in = table, out = q;
        in.off = table.off + row0 * (size_t)hc * sizeof(float);
        in.H = 1; in.W = k * k; in.C = hc; in.B = B;
        out.off = q.off + row0 * (size_t)(9 * hc) * sizeof(float);
        out.H = 1; out.W = k * k; out.C = 9 * hc; out.B = B;
        push_conv(*pl, h->ppm_q[i], in, nullptr, nullptr, out);
        row0 += (size_t)B * k * k;
      }
    }
    rel_side(table);
    Act r = make_act(ar, B, x.H, x.W, hc);
    {
      Op op; op.kind = OP_PPM_TERM; op.name = "ppm_conv_term"; op.kernel = "ppm_conv_term"; op.in = q; op.out = r; op.ppm_scale_rows = term_scale_rows;
      op.bytes = (double)q.bytes + (double)r.bytes;
      pl->ops.push_back(op);
    }
    rel_side(q);
    const size_t side_end = pl->ops.size();
    bt = make_act(ar, B, x.H, x.W, h->bottleneck_x->d.cout);
    push_conv(*pl, h->bottleneck_x, x, nullptr, &r, bt, &ar);
    if (overlap) {
      for (size_t i = side_first; i < side_end; ++i) pl->ops[i].branch = 1;
      pl->ops[side_first].fork_before = true;
      for (size_t i = side_end; i < pl->ops.size(); ++i)
        if (pl->ops[i].has_res) { pl->ops[i].join_before = true; break; }   // the one consumer of R
      for (const auto& t : side_bufs) rel(t);
    }
    // the grouped per-scale GEMMs of the pyramid branch run on the skinny kernel at batch 1 (run_op, gemm_skinny.hip)
    for (size_t i = side_first; i < side_end; ++i) {
      Op& op = pl->ops[i];
      if (op.kind == OP_CONV && op.group_mt == 1 && opt(OPT_PW_SKINNY) != 0 && B * kmax * kmax <= 64 &&
          op.conv->d.bn_tile == 128 && op.conv->d.bk == 32 && op.conv->d.cout % 128 == 0 && (op.conv->d.cin / 32) % 4 == 0)
        op.kernel = "gemm_skinny";
    }
    rel(r);
  } else {
    Act up = make_act(ar, B, x.H, x.W, h->cfg.n_pool_scales * h->cfg.head_channels);
    { Op op; op.kind = OP_PPM_UP; op.name = "ppm_upsample_concat"; op.kernel = "ppm_upsample_concat"; op.in = table; op.out = up; pl->ops.push_back(op); }
    rel(table);
    bt = make_act(ar, B, x.H, x.W, h->bottleneck->d.cout);
    push_conv(*pl, h->bottleneck, x, &up, nullptr, bt);  // cat([x, ppm...]) is never materialised for x
    rel(up);
  }
  pl->named["bottleneck"] = bt;
  rel(x);
  Act lo = make_act(ar, B, bt.H, bt.W, h->conv_seg->d.cout);
  push_conv(*pl, h->conv_seg, bt, nullptr, nullptr, lo);
  pl->named["logits_lowres"] = lo;
  rel(bt);
  {
    Op op; op.kind = OP_UPSAMPLE; op.name = "upsample_logits"; op.kernel = "upsample_logits"; op.in = lo;
    op.bytes = (double)lo.bytes + (double)B * H * W * h->conv_seg->d.cout * 4.0;
    pl->ops.push_back(op);
  }
  rel(lo);
  return pl;   // pl->bytes (high-water mark) is filled in by get_plan
}

}  // namespace

// The arena's `top` can shrink on release; the planner needs the high-water mark.  Recompute it
// from the ops (every Act the plan references must fit).
static size_t plan_high_water(const Plan& pl) {
  size_t hw = 0;
  auto upd = [&](const Act& a) { if (a.bytes && a.off + Arena::round_up(a.bytes) > hw) hw = a.off + Arena::round_up(a.bytes); };
  for (const auto& op : pl.ops) {
    upd(op.in); upd(op.in2); upd(op.res); upd(op.out);
  }
  for (const auto& kv : pl.named) upd(kv.second);
  upd(pl.splitk);
  upd(pl.splitk2);
  return hw;
}

static Plan* get_plan(peanut_pred* h, int B, int H, int W) {
  if (B <= 0 || H < 16 || W < 16) { set_error("forward: need B >= 1 and H, W >= 16"); return nullptr; }
  const std::string key = std::to_string(B) + "x" + std::to_string(H) + "x" + std::to_string(W) + (h->keep_all ? "k" : "");
  auto it = h->plans.find(key);
  if (it == h->plans.end()) {
    auto pl = build_plan(h, B, H, W);
    pl->bytes = plan_high_water(*pl);
    it = h->plans.emplace(key, std::move(pl)).first;
  }
  return it->second.get();
}

static int run_op(peanut_pred* h, const Plan& pl, const Op& op, const float* in_dev, float* out_dev, int sigmoid,
                  hipStream_t s) {
  char* base = (char*)h->ws.p;
  auto P = [&](const Act& a) { return (float*)(base + a.off); };
  switch (op.kind) {
    case OP_TO_NHWC:
      return launch_nchw_to_nhwc_pad(in_dev, P(op.out), pl.B, h->cfg.in_channels, pl.H, pl.W, h->cin_pad, s);
    case OP_CONV_NCHW:
      return launch_conv_patch_nchw(op.conv->d, in_dev, h->cfg.in_channels, P(op.out), pl.B, pl.H, pl.W, op.out.H, op.out.W, s);
    case OP_CONV: {
      ConvArgs a{};
      a.x = P(op.in);
      a.x2 = op.has_in2 ? P(op.in2) : nullptr;
      a.res = op.has_res ? P(op.res) : nullptr;
      a.y = P(op.out);
      a.B = op.in.B; a.H = op.in.H; a.W = op.in.W;
      a.c1 = op.in.C; a.c2 = op.has_in2 ? op.in2.C : 0;
      a.Ho = op.out.H; a.Wo = op.out.W;
      a.ws = P(op.branch ? pl.splitk2 : pl.splitk); a.ws_floats = kSplitKScratchFloats;
      int scale_rows[8] = {0};
      if (op.group_mt) {     // per-scale PSP convs as one grouped GEMM over [scales * srows, C]
        // rows of each scale that hold data (B * k^2): the skinny kernel (gemm_skinny.hip) computes only those
        int most = 0;
        for (int i = 0; i < h->cfg.n_pool_scales && i < 8; ++i) {
          scale_rows[i] = pl.B * h->cfg.pool_scales[i] * h->cfg.pool_scales[i];
          most = std::max(most, scale_rows[i]);
        }
        a.group_valid_rows = most;
        a.group_rows = scale_rows;
        a.mt_per_group = op.group_mt;
        a.w_group_stride = op.conv->w.bytes / sizeof(float) / (size_t)h->cfg.n_pool_scales;
        a.ss_group_stride = op.conv->d.cout_pad;
      }
      if (op.defer_ok) a.defer = &h->deferred;
      return launch_conv(op.conv->d, a, s);
    }
    case OP_WINO_IN: {
      const int rc = launch_wino_input(P(op.in), P(op.out), op.in.B, op.in.H, op.in.W, op.in.C, op.conv->d.dil, s, op.wino_gran, op.conv->wino_m, 0,
                                       h->deferred.valid ? &h->deferred : nullptr);
      h->deferred.valid = false;
      return rc;
    }
    case OP_WINO_GEMM: {
      ConvArgs a{};
      a.x = P(op.in); a.y = P(op.out);
      a.B = 1; a.H = 1; a.W = op.in.W; a.c1 = op.in.C; a.c2 = 0; a.Ho = 1; a.Wo = op.in.W;
      a.ws = P(op.branch ? pl.splitk2 : pl.splitk); a.ws_floats = kSplitKScratchFloats;
      a.mt_per_group = op.wino_mt_per_group;
      a.group_valid_rows = op.wino_valid_rows;
      a.w_group_stride = op.conv->wino.rs ? op.conv->wino_group_bytes : op.conv->wino_group_floats;
      return launch_conv(op.conv->wino, a, s);
    }
    case OP_WINO_OUT:
      return launch_wino_output(P(op.in), op.conv->d.scale, op.conv->d.shift, op.has_res ? P(op.res) : nullptr, P(op.out),
                                op.out.B, op.out.H, op.out.W, op.out.C, op.conv->d.dil, op.conv->d.relu, s, op.wino_gran, op.conv->wino_m);
    case OP_MAXPOOL:
      return launch_maxpool3x3s2(P(op.in), P(op.out), op.in.B, op.in.H, op.in.W, op.in.C, op.out.H, op.out.W, s);
    case OP_PPM_POOL:
      return launch_ppm_pool2(P(op.in), op.has_in2 ? P(op.in2) : nullptr, P(op.out), op.in.B, op.in.H, op.in.W, op.in.C,
                              h->cfg.pool_scales, h->cfg.n_pool_scales, s, op.ppm_scale_rows);
    case OP_PPM_UP:
      return launch_ppm_upsample_concat(P(op.in), P(op.out), op.out.B, op.out.H, op.out.W, h->cfg.head_channels,
                                        h->cfg.pool_scales, h->cfg.n_pool_scales, h->cfg.align_corners, s);
    case OP_PPM_TERM:
      return launch_ppm_conv_term(P(op.in), P(op.out), op.out.B, op.out.H, op.out.W, h->cfg.head_channels,
                                  h->cfg.pool_scales, h->cfg.n_pool_scales, h->cfg.align_corners, s, op.ppm_scale_rows);
    case OP_UPSAMPLE:
      return launch_upsample_logits(P(op.in), out_dev, op.in.B, op.in.H, op.in.W, op.in.C, pl.H, pl.W,
                                    h->cfg.align_corners, sigmoid, s);
  }
  return fail(PEANUT_EINVAL, "unknown op");
}

extern "C" {

const char* peanut_last_error(void) { return g_err.c_str(); }
const char* peanut_last_conv_kernel(void) { return noted_kernel(); }
int peanut_abi_version(void) { return 15; }
const char* peanut_build_arch(void) { return "gfx950"; }
#ifndef PEANUT_SOURCE_HASH
#define PEANUT_SOURCE_HASH ""
#endif
const char* peanut_source_hash(void) { return PEANUT_SOURCE_HASH; }

// ---- tuning options (options.h) ----
namespace {
int option_lookup(const char* key, const char* who) {
  const int i = option_index(key);
  if (i < 0) return fail(PEANUT_EINVAL, std::string(who) + ": unknown option '" + (key ? key : "(null)") + "'");
  return i;
}
int handle_set_option(Options& o, const char* key, long long value, const char* who) {
  const int i = option_lookup(key, who);
  if (i < 0) return i;
  if (option_table()[i].upload_time)
    return fail(PEANUT_EINVAL, std::string(who) + ": option '" + option_table()[i].key +
                                   "' shapes the uploaded weights; set it with peanut_set_default_option before creating the handle");
  o.v[i] = value;
  return 0;
}
}  // namespace

int peanut_set_default_option(const char* key, long long value) {
  const int i = option_lookup(key, "peanut_set_default_option");
  if (i < 0) return i;
  default_options().v[i] = value;
  return 0;
}
int peanut_get_default_option(const char* key, long long* value) {
  const int i = option_lookup(key, "peanut_get_default_option");
  if (i < 0) return i;
  if (value) *value = default_options().v[i];
  return 0;
}
const char* peanut_option_list(void) {
  static const std::string text = [] {
    std::string t;
    const OptionInfo* tab = option_table();
    for (int i = 0; i < OPT_COUNT; ++i)
      t += std::string(tab[i].key) + "=" + std::to_string(tab[i].def) + (tab[i].upload_time ? " [create-time] " : " ") + tab[i].help + "\n";
    return t;
  }();
  return text.c_str();
}
int peanut_pred_set_option(peanut_pred_t* h, const char* key, long long value) {
  if (!h) return fail(PEANUT_EINVAL, "peanut_pred_set_option: null handle");
  if (h->probe) return fail(PEANUT_EINVAL, "peanut_pred_set_option: collect and disable the probe first");
  if (int rc = handle_set_option(h->opts, key, value, "peanut_pred_set_option")) return rc;
  h->graphs.clear();        // launch plans depend on the options: rebuilt on the next forward
  h->plans.clear();
  h->last_plan = nullptr;
  h->probe_plan = nullptr;
  return 0;
}
int peanut_pred_get_option(peanut_pred_t* h, const char* key, long long* value) {
  if (!h) return fail(PEANUT_EINVAL, "peanut_pred_get_option: null handle");
  const int i = option_lookup(key, "peanut_pred_get_option");
  if (i < 0) return i;
  if (value) *value = h->opts.v[i];
  return 0;
}
int peanut_conv_set_option(peanut_conv_t* c, const char* key, long long value) {
  if (!c) return fail(PEANUT_EINVAL, "peanut_conv_set_option: null handle");
  return handle_set_option(c->opts, key, value, "peanut_conv_set_option");
}

int peanut_pred_create(peanut_pred_t** out, const peanut_pred_cfg* cfg, const peanut_tensor* tensors, int n) {
  if (!out || !cfg || (!tensors && n > 0)) return fail(PEANUT_EINVAL, "peanut_pred_create: null argument");
  if (cfg->n_pool_scales < 1 || cfg->n_pool_scales > 8) return fail(PEANUT_EINVAL, "n_pool_scales must be 1..8");
  if (cfg->in_channels < 1 || cfg->num_classes < 1 || cfg->num_classes > 32)
    return fail(PEANUT_EINVAL, "in_channels >= 1 and 1 <= num_classes <= 32 required");
  if (cfg->head_channels % 32) return fail(PEANUT_EINVAL, "head_channels must be a multiple of 32");
  if (!precision_known(cfg->precision)) return fail(PEANUT_EINVAL, "precision must be PEANUT_PREC_{FP32,BF16X3,FP16X3,BF16X6}");
  if (cfg->conv_algo != PEANUT_ALGO_AUTO && cfg->conv_algo != PEANUT_ALGO_DIRECT)
    return fail(PEANUT_EINVAL, "conv_algo must be PEANUT_ALGO_{AUTO,DIRECT}");
  auto h = std::make_unique<peanut_pred>();
  OptionScope option_scope(&h->opts);
  h->cfg = *cfg;
  h->cin_pad = (cfg->in_channels + 15) / 16 * 16;
  TensorMap tm;
  for (int i = 0; i < n; ++i)
    if (tensors[i].name) tm.m[tensors[i].name] = &tensors[i];
  int rc;
  // stem (resnet.py:591-624): 3x3 s2 -> 3x3 -> 3x3, stem_channels = 64
  const int sc = 64;
  if ((rc = add_conv(h.get(), tm, "backbone.stem.0", "backbone.stem.1", cfg->in_channels, h->cin_pad, sc / 2, 3, 2, 1, 1, 1, &h->stem[0]))) return rc;
  if ((rc = add_conv(h.get(), tm, "backbone.stem.3", "backbone.stem.4", sc / 2, sc / 2, sc / 2, 3, 1, 1, 1, 1, &h->stem[1]))) return rc;
  if ((rc = add_conv(h.get(), tm, "backbone.stem.6", "backbone.stem.7", sc / 2, sc / 2, sc, 3, 1, 1, 1, 1, &h->stem[2]))) return rc;
  const int stage_blocks[4] = {3, 4, 6, 3};  // ResNet.arch_settings[50] (resnet.py:386-392)
  int inplanes = sc;
  for (int li = 0; li < 4; ++li) {
    const int planes = 64 << li, stride = cfg->strides[li], dilation = cfg->dilations[li];
    if (stride < 1 || dilation < 1) return fail(PEANUT_EINVAL, "strides/dilations must be >= 1");
    const int first_dil = (dilation > 1 && cfg->contract_dilation) ? dilation / 2 : dilation;  // res_layer.py:67-74
    std::vector<peanut_pred::Block> blocks;
    for (int bi = 0; bi < stage_blocks[li]; ++bi) {
      const std::string p = "backbone.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
      const int s = bi == 0 ? stride : 1, d = bi == 0 ? first_dil : dilation;
      peanut_pred::Block b{nullptr, nullptr, nullptr, nullptr, nullptr};
      if ((rc = add_conv(h.get(), tm, p + ".conv1", p + ".bn1", inplanes, inplanes, planes, 1, 1, 0, 1, 1, &b.c1))) return rc;
      if ((rc = add_conv(h.get(), tm, p + ".conv2", p + ".bn2", planes, planes, planes, 3, s, d, d, 1, &b.c2))) return rc;
      // conv3: BN only; the block's final ReLU follows the residual add and is fused here
      if ((rc = add_conv(h.get(), tm, p + ".conv3", p + ".bn3", planes, planes, planes * 4, 1, 1, 0, 1, 1, &b.c3))) return rc;
      if (bi == 0 && (stride != 1 || inplanes != planes * 4)) {
        if ((rc = add_conv(h.get(), tm, p + ".downsample.0", p + ".downsample.1", inplanes, inplanes, planes * 4, 1, stride, 0, 1, 0, &b.down))) return rc;
        // conv_algo AUTO only: DIRECT keeps the reference's op-for-op form (conv3 -> BN, downsample -> BN, add, ReLU)
        if (cfg->conv_algo == PEANUT_ALGO_AUTO && s == 1 && planes % 32 == 0 && inplanes % 32 == 0 && conv_pw_enabled() &&
            (rc = add_fused_c3d(h.get(), tm, p, planes, inplanes, planes * 4, &b.c3d)))
          return rc;
      }
      blocks.push_back(b);
      inplanes = planes * 4;
    }
    h->layers.push_back(blocks);
  }
  h->feat_channels = inplanes;
  const int hc = cfg->head_channels;
  for (int i = 0; i < cfg->n_pool_scales; ++i) {
    if (cfg->pool_scales[i] < 1) return fail(PEANUT_EINVAL, "pool_scales must be >= 1");
    const std::string p = "decode_head.psp_modules." + std::to_string(i) + ".1";
    ConvLayer* L;
    if ((rc = add_conv(h.get(), tm, p + ".conv", p + ".bn", inplanes, inplanes, hc, 1, 1, 0, 1, 1, &L))) return rc;
    h->ppm.push_back(L);
  }
  const int cat = inplanes + cfg->n_pool_scales * hc;
  if (!cfg->fold_ppm) {
    if ((rc = add_conv(h.get(), tm, "decode_head.bottleneck.conv", "decode_head.bottleneck.bn", cat, cat, hc, 3, 1, 1, 1, 1, &h->bottleneck))) return rc;
  } else {
    // Split W[hc][cat][3][3] into the x part (conv proper) and one folded 1x1 table-builder per scale:
    //   Wq_s[(tap*hc + n)][c] = alpha[n] * W[n][feat + s*hc + c][tap]   (alpha = BN scale, so that the
    //   evaluated term enters the conv epilogue after the scale: relu(alpha*conv_x + beta + alpha*T)).
    const int64_t wshape[4] = {hc, cat, 3, 3};
    const peanut_tensor* w = tm.get("decode_head.bottleneck.conv.weight", 4, wshape, &rc);
    if (!w) return rc;
    std::vector<float> scale(hc), shift(hc);
    if ((rc = bn_fold(h.get(), tm, "decode_head.bottleneck.bn", hc, scale.data(), shift.data()))) return rc;
    std::vector<float> wx((size_t)hc * inplanes * 9);
    for (int n = 0; n < hc; ++n)
      memcpy(&wx[(size_t)n * inplanes * 9], &w->data[(size_t)n * cat * 9], (size_t)inplanes * 9 * sizeof(float));
    {
      auto L = std::make_unique<ConvLayer>();
      L->name = "decode_head.bottleneck.conv[x]";
      if ((rc = upload_conv(*L, wx.data(), scale.data(), shift.data(), hc, inplanes, inplanes, 3, 3, 1, 1, 1, 1, cfg->precision))) return rc;
      if (cfg->conv_algo == PEANUT_ALGO_AUTO && wino_eligible(inplanes, hc, 3, 3, 1, 1, 1, cfg->precision) &&
          (rc = upload_wino_forms(*L, wx.data(), scale.data(), shift.data(), hc, inplanes, inplanes, 1, 1, 1, cfg->precision,
                                  wino_head_tile(cfg->precision))))
        return rc;
      h->bottleneck_x = L.get();
      h->convs.push_back(std::move(L));
    }
    std::vector<float> wq((size_t)9 * hc * hc);
    for (int s = 0; s < cfg->n_pool_scales; ++s) {
      for (int tap = 0; tap < 9; ++tap)
        for (int n = 0; n < hc; ++n)
          for (int c = 0; c < hc; ++c)
            wq[((size_t)tap * hc + n) * hc + c] = scale[n] * w->data[((size_t)n * cat + inplanes + (size_t)s * hc + c) * 9 + tap];
      auto L = std::make_unique<ConvLayer>();
      L->name = "decode_head.bottleneck.conv[ppm" + std::to_string(s) + "]";
      // exact fp32 for the tiny table GEMMs regardless of the handle's precision
      if ((rc = upload_conv(*L, wq.data(), nullptr, nullptr, 9 * hc, hc, hc, 1, 1, 1, 0, 1, 0, PEANUT_PREC_FP32))) return rc;
      h->ppm_q.push_back(L.get());
      h->convs.push_back(std::move(L));
    }
  }
  if ((rc = add_conv(h.get(), tm, "decode_head.conv_seg", "", hc, hc, cfg->num_classes, 1, 1, 0, 1, 0, &h->conv_seg))) return rc;
  if (cfg->fold_ppm) {
    h->ppm_grouped = build_grouped(h->ppm, "decode_head.psp_modules.*.1.conv", &rc);
    if (rc) return rc;
    h->ppmq_grouped = build_grouped(h->ppm_q, "decode_head.bottleneck.conv[ppm*]", &rc);
    if (rc) return rc;
  }
  PEANUT_HIP_CHECK(hipDeviceSynchronize());
  *out = h.release();
  return 0;
}

void peanut_pred_destroy(peanut_pred_t* h) { delete h; }

size_t peanut_pred_workspace_bytes(peanut_pred_t* h, int B, int H, int W) {
  if (!h) return 0;
  OptionScope option_scope(&h->opts);
  Plan* pl = get_plan(h, B, H, W);
  return pl ? pl->bytes : 0;
}

double peanut_pred_flops_per_map(peanut_pred_t* h, int H, int W) {
  if (!h) return 0;
  OptionScope option_scope(&h->opts);
  Plan* pl = get_plan(h, 1, H, W);
  if (!pl) return 0;
  double f = 0;
  for (const auto& op : pl->ops) f += op.flops;
  return f;
}

int peanut_pred_debug_keep(peanut_pred_t* h, int keep) {
  if (!h) return fail(PEANUT_EINVAL, "null handle");
  h->keep_all = keep != 0;
  return 0;
}

int peanut_pred_debug_tensor(peanut_pred_t* h, const char* name, const float** dev_out, int dims[4]) {
  if (!h || !name || !dev_out || !dims) return fail(PEANUT_EINVAL, "null argument");
  if (!h->last_plan || !h->last_plan->keep_all) return fail(PEANUT_EINVAL, "debug_tensor: run a forward after peanut_pred_debug_keep(h, 1)");
  auto it = h->last_plan->named.find(name);
  if (it == h->last_plan->named.end()) return fail(PEANUT_EINVAL, std::string("unknown tensor '") + name + "'");
  const Act& a = it->second;
  *dev_out = (const float*)((char*)h->ws.p + a.off);
  dims[0] = a.B; dims[1] = a.H; dims[2] = a.W; dims[3] = a.C;
  return 0;
}

int peanut_pred_debug_read(peanut_pred_t* h, const char* name, float* dst_dev, size_t max_floats, int dims[4],
                           void* stream) {
  const float* src = nullptr;
  int rc = peanut_pred_debug_tensor(h, name, &src, dims);
  if (rc || !dst_dev) return rc;
  const size_t n = (size_t)dims[0] * dims[1] * dims[2] * dims[3];
  if (n > max_floats) return fail(PEANUT_EINVAL, "debug_read: destination too small");
  PEANUT_HIP_CHECK(hipMemcpyAsync(dst_dev, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int peanut_pred_forward(peanut_pred_t* h, const float* in_dev, float* out_dev, int B, int H, int W, int apply_sigmoid,
                        void* stream) {
  if (!h || !in_dev || !out_dev) return fail(PEANUT_EINVAL, "peanut_pred_forward: null argument");
  OptionScope option_scope(&h->opts);
  Plan* pl = get_plan(h, B, H, W);
  if (!pl) return PEANUT_EINVAL;
  int rc;
  if (pl->bytes > h->ws.bytes) h->graphs.clear();   // captured launches point into the old workspace
  if ((rc = h->ws.ensure(pl->bytes))) return rc;
  h->last_plan = pl;
  h->deferred.valid = false;       // a forward that failed between a deferring conv1 and its consumer must not leave its note behind
  hipStream_t s = (hipStream_t)stream;
  if (!h->probe) {
    bool two_streams = false;
    for (const auto& op : pl->ops) two_streams |= op.branch != 0;
    if (two_streams && !h->side) {
      PEANUT_HIP_CHECK(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
      PEANUT_HIP_CHECK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
      PEANUT_HIP_CHECK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    auto enqueue = [&]() -> int {
      for (const auto& op : pl->ops) {
        if (op.fork_before) {
          PEANUT_HIP_CHECK(hipEventRecord(h->ev_fork, s));
          PEANUT_HIP_CHECK(hipStreamWaitEvent(h->side, h->ev_fork, 0));
        }
        if (op.join_before) {
          PEANUT_HIP_CHECK(hipEventRecord(h->ev_join, h->side));
          PEANUT_HIP_CHECK(hipStreamWaitEvent(s, h->ev_join, 0));
        }
        if (int r = run_op(h, *pl, op, in_dev, out_dev, apply_sigmoid, op.branch ? h->side : s)) return r;
      }
      return 0;
    };
    if (!h->use_graph || h->keep_all) return enqueue();
    return h->graphs.run({(uintptr_t)pl, (uintptr_t)in_dev, (uintptr_t)out_dev, (uintptr_t)apply_sigmoid, (uintptr_t)s}, s, enqueue);
  }
  if (h->probe_plan && h->probe_plan != pl) return fail(PEANUT_EINVAL, "probe: shape changed while probing; collect first");
  h->probe_plan = pl;
  std::vector<hipEvent_t> ev(pl->ops.size() + 1);
  for (auto& e : ev) {
    if (!h->event_pool.empty()) { e = h->event_pool.back(); h->event_pool.pop_back(); }
    else PEANUT_HIP_CHECK(hipEventCreate(&e));
  }
  PEANUT_HIP_CHECK(hipEventRecord(ev[0], s));
  for (size_t i = 0; i < pl->ops.size(); ++i) {
    if ((rc = run_op(h, *pl, pl->ops[i], in_dev, out_dev, apply_sigmoid, s))) return rc;
    PEANUT_HIP_CHECK(hipEventRecord(ev[i + 1], s));
  }
  h->probe_events.push_back(std::move(ev));
  return 0;
}

int peanut_pred_use_graph(peanut_pred_t* h, int enable) {
  if (!h) return fail(PEANUT_EINVAL, "null handle");
  h->use_graph = enable != 0;
  if (!h->use_graph) h->graphs.clear();
  return 0;
}

int peanut_pred_probe_enable(peanut_pred_t* h, int enable) {
  if (!h) return fail(PEANUT_EINVAL, "null handle");
  h->probe = enable != 0;
  return 0;
}

int peanut_pred_probe_collect(peanut_pred_t* h, int max_ops, const char** names, const char** kernels, double* ms_sum,
                              double* flops, double* bytes, int* n_forwards) {
  if (!h) return fail(PEANUT_EINVAL, "null handle");
  Plan* pl = h->probe_plan;
  if (!pl || h->probe_events.empty()) { if (n_forwards) *n_forwards = 0; return 0; }
  const int n = (int)pl->ops.size();
  std::vector<double> sum(n, 0.0);
  for (auto& ev : h->probe_events) {
    PEANUT_HIP_CHECK(hipEventSynchronize(ev.back()));
    for (int i = 0; i < n; ++i) {
      float t = 0;
      PEANUT_HIP_CHECK(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
      sum[i] += t;
    }
    for (auto e : ev) h->event_pool.push_back(e);
  }
  if (n_forwards) *n_forwards = (int)h->probe_events.size();
  h->probe_events.clear();
  h->probe_plan = nullptr;
  for (int i = 0; i < n && i < max_ops; ++i) {
    if (names) names[i] = pl->ops[i].name.c_str();
    if (kernels) kernels[i] = pl->ops[i].kernel.c_str();
    if (ms_sum) ms_sum[i] = sum[i];
    if (flops) flops[i] = pl->ops[i].flops;
    if (bytes) bytes[i] = pl->ops[i].bytes;
  }
  return n;
}

// ---- operator-level conv ----
int peanut_conv_create(peanut_conv_t** out, const float* w, const float* scale, const float* shift, int cout, int cin,
                       int cin_pad, int kh, int kw, int stride, int pad, int dil, int relu, int precision, int conv_algo) {
  if (!out || !w) return fail(PEANUT_EINVAL, "peanut_conv_create: null argument");
  if (cout < 1 || cin < 1 || kh < 1 || kw < 1 || stride < 1 || dil < 1 || pad < 0)
    return fail(PEANUT_EINVAL, "peanut_conv_create: bad geometry");
  if (!precision_known(precision)) return fail(PEANUT_EINVAL, "peanut_conv_create: bad precision");
  auto c = std::make_unique<peanut_conv>();
  OptionScope option_scope(&c->opts);
  c->L.name = "conv";
  if (conv_algo != PEANUT_ALGO_AUTO && conv_algo != PEANUT_ALGO_DIRECT) return fail(PEANUT_EINVAL, "peanut_conv_create: bad conv_algo");
  int rc = upload_conv(c->L, w, scale, shift, cout, cin, cin_pad, kh, kw, stride, pad, dil, relu, precision);
  if (rc) return rc;
  if (conv_algo == PEANUT_ALGO_AUTO && wino_eligible(cin_pad, cout, kh, kw, stride, pad, dil, precision) &&
      (rc = upload_wino(c->L, w, cout, cin, cin_pad, precision)))
    return rc;
  PEANUT_HIP_CHECK(hipDeviceSynchronize());
  *out = c.release();
  return 0;
}

void peanut_conv_destroy(peanut_conv_t* c) { delete c; }

int peanut_conv_precision(peanut_conv_t* c) {
  if (!c) return PEANUT_EINVAL;
  const ConvDesc& d = (c->L.has_wino && c->L.wino.rs) ? c->L.wino : c->L.d;
  if (d.rs) return precision_of_planes(d.s_planes);
  return PEANUT_PREC_FP32;
}

int peanut_conv_forward(peanut_conv_t* c, const float* x, const float* x2, int c1, const float* res, float* y, int B,
                        int H, int W, void* stream) {
  if (!c || !x || !y) return fail(PEANUT_EINVAL, "peanut_conv_forward: null argument");
  OptionScope option_scope(&c->opts);
  const ConvDesc& d = c->L.d;
  ConvArgs a{};
  a.x = x; a.x2 = x2; a.res = res; a.y = y; a.B = B; a.H = H; a.W = W;
  a.c1 = x2 ? c1 : d.cin;
  a.c2 = x2 ? d.cin - c1 : 0;
  a.Ho = conv_out_dim(H, d.kh, d.stride, d.pad, d.dil);
  a.Wo = conv_out_dim(W, d.kw, d.stride, d.pad, d.dil);
  if (a.Ho < 1 || a.Wo < 1) return fail(PEANUT_EINVAL, "peanut_conv_forward: empty output");
  if (int rc = c->ws.ensure(kSplitKScratchFloats * sizeof(float))) return rc;
  a.ws = (float*)c->ws.p; a.ws_floats = kSplitKScratchFloats;
  if (c->L.has_wino && !x2) {
    size_t vf, mf;
    wino_scratch_floats(c->L, B, H, W, &vf, &mf);
    // growing a scratch buffer frees the old one: make sure no earlier launch still uses it
    if (vf * sizeof(float) > c->wino_v.bytes || mf * sizeof(float) > c->wino_m.bytes) PEANUT_HIP_CHECK(hipDeviceSynchronize());
    if (int rc = c->wino_v.ensure(vf * sizeof(float))) return rc;
    if (int rc = c->wino_m.ensure(mf * sizeof(float))) return rc;
  }
  return launch_conv_layer(c->L, a, (float*)c->wino_v.p, (float*)c->wino_m.p, (hipStream_t)stream);
}

int peanut_debug_weight_pieces(const float* values, int n, int precision, unsigned short* pieces, float* pack_scale) {
  const int planes = rs_planes_of(precision);
  if (!values || !pieces || !pack_scale || n < 1 || !planes) return fail(PEANUT_EINVAL, "peanut_debug_weight_pieces: bad arguments");
  // one 1 x n "layer" through the real packer (tile 0, k-tiles of 16 values), then gathered back per plane
  const int cin_pad = (n + 15) / 16 * 16, bn = 64, np = planes == 4 ? 2 : planes;
  *pack_scale = sx_pack_scale(values, (size_t)n, planes);
  std::vector<unsigned char> packed(sx_packed_bytes(cin_pad, 1, bn, planes));
  pack_weights_sx(values, 1, n, cin_pad, bn, planes, *pack_scale, packed.data());
  const unsigned short* o = reinterpret_cast<const unsigned short*>(packed.data());
  for (int q = 0; q < 3; ++q)
    for (int c = 0; c < n; ++c)
      pieces[(size_t)q * n + c] = q < np ? o[((size_t)(c / 16) * np + q) * bn * 16 + (c % 16)] : (unsigned short)0;
  return 0;
}

long long peanut_debug_deferred_splitk_count(void) { return wino_deferred_count(); }

int peanut_debug_wino_weights(const float* w_oihw, int cout, int cin, int tile, float* out) {
  if (!w_oihw || !out || cout < 1 || cin < 1 || tile < 4 || tile > 6) return fail(PEANUT_EINVAL, "peanut_debug_wino_weights: bad arguments");
  wino_transform_weights(w_oihw, cout, cin, out, tile);
  return 0;
}

}  // extern "C"
