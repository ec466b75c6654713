// Pieces shared by the network planners (pred_api.hip: PSPNet; rcnn_api.hip: Mask R-CNN front end):
// device buffers, conv layers resident on the device, the workspace arena, state-dict lookup.
#pragma once
#include <stdlib.h>
#include <math.h>

#include <stdint.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/peanut_hip.h"
#include "common.h"
#include "options.h"

namespace peanut {

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int ensure(size_t n) {
    if (n <= bytes) return 0;
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) return fail(PEANUT_EHIP, std::string("hipMalloc(") + std::to_string(n) + "): " + hipGetErrorString(e));
    bytes = n;
    return 0;
  }
};

// Replays a fixed launch sequence as a hipGraph.  Keyed by the caller on everything baked into the captured
// launches (plan, device pointers, flags).  The first call with a new key runs the sequence directly (so that
// one-time initialisation -- occupancy queries, the zero page, function attributes -- happens outside a capture),
// the second captures + instantiates, later ones are a single hipGraphLaunch.
struct GraphCache {
  struct Entry { std::vector<uintptr_t> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; bool warmed = false; };
  std::vector<Entry> entries;
  static constexpr size_t kMaxEntries = 16;
  void clear() {
    for (auto& e : entries) {
      if (e.exec) (void)hipGraphExecDestroy(e.exec);
      if (e.graph) (void)hipGraphDestroy(e.graph);
    }
    entries.clear();
  }
  ~GraphCache() { clear(); }
  template <class F>
  int run(const std::vector<uintptr_t>& key, hipStream_t s, F&& enqueue) {
    if (s == nullptr) return enqueue();   // the legacy default stream cannot be captured: plain launches
    Entry* hit = nullptr;
    for (auto& e : entries) if (e.key == key) { hit = &e; break; }
    if (!hit) {
      if (entries.size() >= kMaxEntries) {   // drop the oldest
        if (entries.front().exec) (void)hipGraphExecDestroy(entries.front().exec);
        if (entries.front().graph) (void)hipGraphDestroy(entries.front().graph);
        entries.erase(entries.begin());
      }
      entries.emplace_back();
      entries.back().key = key;
      entries.back().warmed = true;
      return enqueue();
    }
    if (!hit->exec) {
      PEANUT_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      const int rc = enqueue();
      hipGraph_t g = nullptr;
      const hipError_t e = hipStreamEndCapture(s, &g);
      if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
      if (e != hipSuccess) return fail(PEANUT_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
      hipGraphExec_t x = nullptr;
      const hipError_t e2 = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
      if (e2 != hipSuccess) { (void)hipGraphDestroy(g); return fail(PEANUT_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e2)); }
      hit->graph = g;
      hit->exec = x;
    }
    PEANUT_HIP_CHECK(hipGraphLaunch(hit->exec, s));
    return 0;
  }
};

// A conv layer resident on the device.
struct ConvLayer {
  std::string name;
  ConvDesc d{};
  int cin_real = 0;
  DevBuf w, ss;  // packed weights; scale||shift
  // Winograd F(4x4,3x3) form of a stride-1 3x3 layer (winograd.hip): `wino` is the descriptor of the 36
  // grouped GEMMs (a 1x1 conv with unit scale / zero shift); d.scale/d.shift/d.relu apply in the output transform
  bool has_wino = false;
  ConvDesc wino{};
  size_t wino_group_floats = 0;
  DevBuf wino_w, wino_ss;
  // pre-split bf16 pieces of the weights (gemm_rs.hip) for the emulated-fp32 modes: of the layer itself when it is
  // pointwise (d.w_s), of the 36 Winograd position matrices (wino.w_s, wino_group_bytes apart)
  DevBuf w_s, wino_w_s;
  size_t wino_group_bytes = 0;
  // output tile of the Winograd form: 4 (F(4x4,3x3), 36 positions), 5 (F(5x5,3x3), 49) or 6 (F(6x6,3x3), 64)
  int wino_m = 4;
  int wino_np() const { return (wino_m + 2) * (wino_m + 2); }
  // the same layer with another Winograd form (a chain: an F(6x6) layer keeps an F(4x4) twin and, where wino5_wanted says
  // so, an F(5x5) one behind it): the planner picks per shape (wino_pick_form)
  std::unique_ptr<ConvLayer> alt;
};

// emulation kind of a precision mode (rs_common.h: 2 = bf16x3, 3 = bf16x6, 4 = fp16x3; 0: fp32 MFMA mode)
inline int rs_planes_of(int precision) {
  return precision == PEANUT_PREC_BF16X6 ? 3 : (precision == PEANUT_PREC_BF16X3 ? 2 : (precision == PEANUT_PREC_FP16X3 ? 4 : 0));
}
inline bool precision_known(int precision) { return precision == PEANUT_PREC_FP32 || rs_planes_of(precision) != 0; }
inline int precision_of_planes(int planes) {
  return planes == 3 ? PEANUT_PREC_BF16X6 : (planes == 2 ? PEANUT_PREC_BF16X3 : (planes == 4 ? PEANUT_PREC_FP16X3 : PEANUT_PREC_FP32));
}
// PEANUT_RS_CONV=0 keeps the non-pointwise layers of the emulated modes on the fp32 MFMA kernel (A/B measurements)
inline bool rs_conv_enabled() {
  return opt(OPT_RS_CONV) != 0;
}
// n-tile of a pointwise layer's pre-split weights in the emulated-fp32 modes (0: the layer stays on the fp32 kernel)
inline int rs_bn_tile(int cin_pad, int cout, int kh, int kw, int pad) {
  if (kh != 1 || kw != 1 || pad != 0 || cin_pad % 16 != 0 || cout < 64) return 0;
  // 128 x 64 tiles (three workgroups per CU) for the short-K layers whose time is their epilogue's HBM traffic
  const int bn64_maxk = (int)opt(OPT_RS_BN64_MAXK);
  return (cout >= 128 && cin_pad > bn64_maxk) ? 128 : 64;
}

inline int upload_conv(ConvLayer& L, const float* w_oihw, const float* scale, const float* shift, int cout, int cin,
                int cin_pad, int kh, int kw, int stride, int pad, int dil, int relu, int precision) {
  if (cin_pad % 16 != 0 || cin_pad < cin) return fail(PEANUT_EINVAL, L.name + ": cin_pad must be a multiple of 16 and >= cin");
  ConvDesc& d = L.d;
  d.cin = cin_pad; d.cout = cout; d.kh = kh; d.kw = kw; d.stride = stride; d.pad = pad; d.dil = dil; d.relu = relu;
  conv_pick_tiles(cin_pad, cout, &d.bn_tile, &d.bk, kh == 1 && kw == 1 && pad == 0);
  // emulated-fp32 modes: the pointwise layers run on gemm_rs.hip (every other layer stays on the fp32 MFMA kernels)
  d.rs = 0;
  d.s_planes = 0;
  d.s_alpha = 1.f;
  d.flush_ch = 0;
  if (rs_planes_of(precision) && rs_bn_tile(cin_pad, cout, kh, kw, pad)) {
    d.rs = 1;
    d.s_planes = rs_planes_of(precision);
    d.bn_tile = rs_bn_tile(cin_pad, cout, kh, kw, pad);
  } else if (rs_planes_of(precision) && rs_conv_enabled() && (kh > 1 || kw > 1) && cin_pad % 16 == 0) {
    d.rs = 2;                                                   // conv_rs.hip
    d.s_planes = rs_planes_of(precision);
    d.bn_tile = cout >= 128 ? 128 : (cout > 32 ? 64 : 32);
  }
  d.cout_pad = (cout + d.bn_tile - 1) / d.bn_tile * d.bn_tile;
  L.cin_real = cin;
  const size_t nw = conv_packed_floats(cin_pad, cout, kh, kw, d.bn_tile);
  std::vector<float> packed(nw);
  pack_conv_weights(w_oihw, cout, cin, cin_pad, kh, kw, d.bn_tile, d.bk, packed.data());
  std::vector<float> ss(2 * (size_t)d.cout_pad, 0.f);
  for (int n = 0; n < cout; ++n) {
    ss[n] = scale ? scale[n] : 1.f;
    ss[d.cout_pad + n] = shift ? shift[n] : 0.f;
  }
  int rc;
  if ((rc = L.w.ensure(nw * sizeof(float)))) return rc;
  if ((rc = L.ss.ensure(ss.size() * sizeof(float)))) return rc;
  PEANUT_HIP_CHECK(hipMemcpy(L.w.p, packed.data(), nw * sizeof(float), hipMemcpyHostToDevice));
  PEANUT_HIP_CHECK(hipMemcpy(L.ss.p, ss.data(), ss.size() * sizeof(float), hipMemcpyHostToDevice));
  d.w_packed = (const float*)L.w.p;
  d.scale = (const float*)L.ss.p;
  d.shift = (const float*)L.ss.p + d.cout_pad;
  d.w_s = nullptr;
  if (d.rs) {
    std::vector<unsigned char> ps(d.rs == 2 ? sx_conv_packed_bytes(cin_pad, cout, kh, kw, d.bn_tile, d.s_planes)
                                            : sx_packed_bytes(cin_pad, cout, d.bn_tile, d.s_planes));
    const float wscale = sx_pack_scale(w_oihw, (size_t)cout * cin * kh * kw, d.s_planes);
    d.s_alpha = 1.f / wscale;                                                       // a power of two: exact
    if (d.rs == 2) pack_weights_sx_conv(w_oihw, cout, cin, cin_pad, kh, kw, d.bn_tile, d.s_planes, wscale, ps.data());
    else pack_weights_sx(w_oihw, cout, cin, cin_pad, d.bn_tile, d.s_planes, wscale, ps.data());
    if ((rc = L.w_s.ensure(ps.size()))) return rc;
    PEANUT_HIP_CHECK(hipMemcpy(L.w_s.p, ps.data(), ps.size(), hipMemcpyHostToDevice));
    d.w_s = L.w_s.p;
  }
  return 0;
}

// Adds the Winograd form to an uploaded stride-1 3x3 layer (keeps the direct form for the two-source path).
inline bool wino_eligible(int cin_pad, int cout, int kh, int kw, int stride, int pad, int dil, int precision, int min_cin_default = 128) {
  // measured (profiles/r2s): from 128 input channels on (+1.2 % on the headline, +1.8 % on the detector; 64 adds 0.1 % with
  // the F(4x4) form).  The prediction planner passes 64: with F(6x6) (V / M 1.78 x the activations instead of 2.25 x) layer1's
  // conv2 goes from 0.331 to 0.236 ms per launch at batch 32 (profiles/r4h: 829.4 -> 835.9 maps/s), and push_conv keeps
  // such layers on the direct kernel where the launch is too small to pay for three (wino_min_pixels).  The transforms
  // are fp32 in every precision mode.
  // In the three-product modes the direct register-split kernel (conv_rs.hip) is the faster one for those narrow layers
  // (fp16x3, layer1 conv2: 0.175 ms direct, 0.21 ms as Winograd; bf16x6: 0.248 -> 0.22 ms; profiles/r4i): they keep 128.
  const int env_min = (int)opt(OPT_WINO_MIN_CIN);
  const bool three_products = precision == PEANUT_PREC_FP16X3 || precision == PEANUT_PREC_BF16X3;
  const int min_cin = env_min ? env_min : (three_products && min_cin_default < 128 ? 128 : min_cin_default);
  return kh == 3 && kw == 3 && stride == 1 && pad == dil && cin_pad >= min_cin && cin_pad % 32 == 0 && cout % 4 == 0 && cout >= 64;
}

// Layers with fewer than 128 input channels take their Winograd form only from this many input pixels on: below it the
// direct kernel's one launch beats transform + GEMM + transform (one 240 x 240 map: 3 600 pixels in layer1)
inline long long wino_min_pixels(int cin_pad) {
  const long long env = opt(OPT_WINO_NARROW_MINPIX);
  return cin_pad < 128 ? env : 0;
}

// Tile size of a layer's Winograd form (requested = 0: this policy; 4 / 6: the caller's choice; PEANUT_WINO_M = 4 / 6
// overrides both, read at every upload).  F(6x6,3x3) executes 1.78 multiplies per output instead of 2.25 and its V / M
// tensors are that much smaller, at about 3 x the rounding error of the F(4x4) form (winograd.hip).  Measured on the
// headline forward (profiles/r3y): every eligible layer as F(6x6) +5.3 % throughput but 7.9e-6 -> 3.5e-5 on the golden
// logits -- nearly all of it from the PSP bottleneck, whose K = 2048 accumulation error goes through A^T's factors of up to
// 32 (1024 in 2-D) straight into conv_seg; the BACKBONE layers alone: +3.5 %, 7.9e-6 -> 7.2e-6 .. 1.26e-5 (distance to
// the float64 run 5.3e-6 -> 7.6e-6).  So: the prediction planner asks for F(6x6) in the backbone and F(4x4) in the head.
// Which of the two forms a backbone layer runs is decided per shape (wino_pick_form): with dilation 4 a 60 x 60 map's
// sub-grids are 15 x 15 -- 3 x 3 tiles of 6 (18 rows) or 4 x 4 tiles of 4 (16): 576 position-tiles either way, F(4x4) stays
// -- while a 720 x 720 map's 23 x 23 sub-grids take 4 x 4 tiles of 6 or 6 x 6 tiles of 4, and F(6x6) executes 0.71 of it.
inline int wino_tile_for(int dil, int requested) {
  const int forced = (int)opt(OPT_WINO_M);
  if (forced >= 4 && forced <= 6) return forced;
  if (requested >= 4 && requested <= 6) return requested;
  return dil <= (int)opt(OPT_WINO6_MAXDIL) ? 6 : 4;            // A/B knob: largest dilation that gets an F(6x6) form at all
}

// Winograd form of the PSP bottleneck over x: 0 = the backbone's policy (F(6x6) with an F(4x4) twin, chosen per shape) on
// the fp32 MFMA kernels, whose position GEMMs accumulate in two levels (wino_flush_channels); F(4x4) in the emulated
// modes, where gemm_rs.hip keeps one running sum (its 256 x 256 kernel has no registers for a second accumulator set, and
// measured with the second level on the 128 x 128 kernel -- profiles/r4f -- F(6x6) bought no throughput there and cost
// accuracy: 8.9e-6 -> 1.15e-5 / 7.0e-6 -> 9.3e-6 from the float64 run at 480 x 480 in bf16x6 / fp16x3).
// PEANUT_WINO_HEAD_M = 4 / 5 / 6 pins one form.
inline int wino_head_tile(int precision) {
  const int m = (int)opt(OPT_WINO_HEAD_M);
  if (m >= 4 && m <= 6) return m;
  return rs_planes_of(precision) ? 4 : 0;
}

// Channels per partial sum of the position GEMMs' two-level fp32 accumulation (conv_common.h: PEANUT_FLUSH_*; 0 = one
// running sum).  The rounding error of a Winograd layer is the accumulation error of its position GEMMs times the
// amplification of A^T (up to 32 per dimension for F(6x6)): with partial sums of 64 channels that error drops about
// three-fold at K = 2048 (fp32 simulation: F(4x4) 4.1e-6 -> 0.95e-6 rms, F(6x6) 1.2e-5 -> 3.0e-6), and F(6x6) in the PSP
// bottleneck lands at 7.6e-6 .. 1.0e-5 on the golden logits -- next to F(4x4) with one running sum (6.7-7.9e-6);
// F(6x6) with one running sum: 3.0-4.0e-5 (profiles/r4e, r4f).  Every Winograd GEMM of the fp32 MFMA kernels does it
// (measured cost: within noise); PEANUT_WINO_FLUSH_CH overrides the 64 (a multiple of 32; 0 = off).
inline int wino_flush_channels(bool rs) {
  if (rs) return 0;
  const int ch = (int)opt(OPT_WINO_FLUSH_CH);
  return (ch > 0 && ch % 32 == 0) ? ch : 0;
}

// Does a backbone layer also carry the F(5x5,3x3) form (winograd.hip)?  Its point is divisibility: the dilation-4 layers
// of a 480 x 480 map work on 15 x 15 sub-grids, which 5 x 5 tiles cover exactly (441 position-tiles against 576 with
// either other form: the position GEMMs and both transforms of layer4.1 / layer4.2 conv2 shrink to 0.77).  At dilation 1 / 2
// F(6x6) executes less on every map size the agent uses, so those layers do not pay the upload for a third form
// (PEANUT_WINO5_MINDIL: smallest dilation that gets one; 0 = none, 1 = every backbone layer).
inline bool wino5_wanted(int dil) {
  const int mindil = (int)opt(OPT_WINO5_MINDIL);
  return mindil > 0 && dil >= mindil;
}

// wino_m: 4, 5 or 6, or 0 = wino_tile_for's choice
inline int upload_wino(ConvLayer& L, const float* w_oihw, int cout, int cin, int cin_pad, int precision, int wino_m = 4) {
  ConvDesc& g = L.wino;
  L.wino_m = wino_tile_for(L.d.dil, wino_m);
  const int np = L.wino_np();
  g.cin = cin_pad; g.cout = cout; g.kh = g.kw = 1; g.stride = 1; g.pad = 0; g.dil = 1; g.relu = 0;
  conv_pick_tiles(cin_pad, cout, &g.bn_tile, &g.bk, true);
  g.rs = 0;
  g.s_planes = 0;
  g.s_alpha = 1.f;
  if (rs_planes_of(precision) && rs_bn_tile(cin_pad, cout, 1, 1, 0)) {   // see upload_conv
    g.rs = 1;
    g.s_planes = rs_planes_of(precision);
    g.bn_tile = rs_bn_tile(cin_pad, cout, 1, 1, 0);
  }
  g.bk = 32;
  g.cout_pad = (cout + g.bn_tile - 1) / g.bn_tile * g.bn_tile;
  g.w_s = nullptr;
  g.flush_ch = wino_flush_channels(g.rs != 0);
  const size_t gf = conv_packed_floats(cin_pad, cout, 1, 1, g.bn_tile);
  std::vector<float> U((size_t)np * cout * cin);
  wino_transform_weights(w_oihw, cout, cin, U.data(), L.wino_m);
  std::vector<float> packed((size_t)np * gf);
  for (int pos = 0; pos < np; ++pos) {
    const float* u = U.data() + (size_t)pos * cout * cin;
    pack_conv_weights(u, cout, cin, cin_pad, 1, 1, g.bn_tile, g.bk, packed.data() + pos * gf);
  }
  std::vector<float> ss(2 * (size_t)g.cout_pad, 0.f);
  for (int n = 0; n < g.cout_pad; ++n) ss[n] = 1.f;
  int rc;
  if ((rc = L.wino_w.ensure(packed.size() * sizeof(float)))) return rc;
  if ((rc = L.wino_ss.ensure(ss.size() * sizeof(float)))) return rc;
  PEANUT_HIP_CHECK(hipMemcpy(L.wino_w.p, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
  PEANUT_HIP_CHECK(hipMemcpy(L.wino_ss.p, ss.data(), ss.size() * sizeof(float), hipMemcpyHostToDevice));
  g.w_packed = (const float*)L.wino_w.p;
  g.scale = (const float*)L.wino_ss.p;
  g.shift = (const float*)L.wino_ss.p + g.cout_pad;
  L.wino_group_floats = gf;
  if (g.rs) {
    const size_t gb = sx_packed_bytes(cin_pad, cout, g.bn_tile, g.s_planes);
    std::vector<unsigned char> ps((size_t)np * gb);
    const float wscale = sx_pack_scale(U.data(), U.size(), g.s_planes);     // one scale for all positions (one launch)
    g.s_alpha = 1.f / wscale;
    for (int pos = 0; pos < np; ++pos)
      pack_weights_sx(U.data() + (size_t)pos * cout * cin, cout, cin, cin_pad, g.bn_tile, g.s_planes, wscale, ps.data() + pos * gb);
    if ((rc = L.wino_w_s.ensure(ps.size()))) return rc;
    PEANUT_HIP_CHECK(hipMemcpy(L.wino_w_s.p, ps.data(), ps.size(), hipMemcpyHostToDevice));
    g.w_s = L.wino_w_s.p;
    L.wino_group_bytes = gb;
  }
  L.has_wino = true;
  return 0;
}

// The Winograd forms of one stride-1 3x3 layer as the prediction planner wants them: the policy's form (wino_tile_for) and,
// when that is F(6x6), an F(4x4) twin plus -- where wino5_wanted says so -- an F(5x5) one, each a full ConvLayer of its own
// behind L.alt (wino_pick_form chooses per shape).  `requested`: 0 = policy, 4 / 5 / 6 = that form only.
inline int upload_wino_forms(ConvLayer& L, const float* w_oihw, const float* scale, const float* shift, int cout, int cin,
                             int cin_pad, int pad, int dil, int relu, int precision, int requested) {
  int rc;
  if ((rc = upload_wino(L, w_oihw, cout, cin, cin_pad, precision, requested))) return rc;
  if (requested != 0 || L.wino_m != 6) return 0;
  L.alt = std::make_unique<ConvLayer>();
  L.alt->name = L.name;
  if ((rc = upload_conv(*L.alt, w_oihw, scale, shift, cout, cin, cin_pad, 3, 3, 1, pad, dil, relu, precision)) ||
      (rc = upload_wino(*L.alt, w_oihw, cout, cin, cin_pad, precision, 4)))
    return rc;
  if (L.alt->wino_m != 4) { L.alt.reset(); return 0; }           // PEANUT_WINO_M forces one form
  if (wino5_wanted(dil)) {
    auto& A5 = L.alt->alt;
    A5 = std::make_unique<ConvLayer>();
    A5->name = L.name;
    if ((rc = upload_conv(*A5, w_oihw, scale, shift, cout, cin, cin_pad, 3, 3, 1, pad, dil, relu, precision)) ||
        (rc = upload_wino(*A5, w_oihw, cout, cin, cin_pad, precision, 5)))
      return rc;
  }
  return 0;
}

// Row padding per Winograd position: whole 256-row tiles where the position GEMMs run on a 256-row kernel (the
// three-stage fp32 one of conv_pw.hip for K >= 1024; the 256 x 256 one of gemm_rs.hip), else 128
inline int wino_gran_for(const ConvLayer& L, int B, int H, int W) {
  int th, tw;
  long long n_tiles, m_pad;
  wino_geometry(B, H, W, L.d.dil, &th, &tw, &n_tiles, &m_pad, 256, L.wino_m);
  // ... unless whole 256-row tiles pad the positions by more than an eighth over 128-row ones (four 480 x 480 maps, F(5x5) at
  // dilation 4: 576 tiles per position -> 768 rows against 640): the 128-row kernels then execute less.  (Sixteen maps' PSP
  // bottleneck, 1 600 tiles -> 1 792 against 1 664, stays on the 256-row kernel: faster per row, and its waves skip padding rows.)
  {
    long long n128, m128;
    wino_geometry(B, H, W, L.d.dil, &th, &tw, &n128, &m128, 128, L.wino_m);
    if (m_pad * 8 > m128 * 9) return 128;
  }
  const ConvDesc& g = L.wino;
  const long long np = L.wino_np();
  if (g.rs) return gemm_rs_uses_256(g.cout, np * m_pad, (int)(m_pad / 128), g.bn_tile, g.cin) ? 256 : 128;
  return (conv_pw_enabled() && (conv_pw_uses_256(g.cout, np * m_pad, (int)(m_pad / 128), g.bn_tile, g.cin) ||
                                conv_pw_uses_256p(g.cout, np * m_pad, (int)(m_pad / 128), g.bn_tile, g.cin, g.flush_ch / 32)))
             ? 256 : 128;
}

// position-rows the Winograd GEMMs of this layer execute on an input [B,H,W,*]: positions x padded tiles
inline long long wino_padded_rows(const ConvLayer& L, int B, int H, int W) {
  int th, tw;
  long long n_tiles, m_pad;
  wino_geometry(B, H, W, L.d.dil, &th, &tw, &n_tiles, &m_pad, wino_gran_for(L, B, H, W), L.wino_m);
  return (long long)L.wino_np() * m_pad;
}

// F(6x6) layer with an F(4x4) twin: on small maps the 64 positions each pad their few tiles to a whole GEMM tile and the
// 36-position form executes less (one 240 x 240 map: 25 tiles per position -> 64 x 128 rows against 36 x 128); with
// dilation 4 it depends on how the sub-grids divide into tiles
inline const ConvLayer* wino_pick_form(const ConvLayer* L, int B, int H, int W) {
  if (!L->has_wino || !L->alt || !L->alt->has_wino) return L;
  // the F(4x4) twin, and the cheapest of the larger-tile forms (a tie goes to the smaller tile: less rounding error)
  const ConvLayer* f4 = nullptr;
  const ConvLayer* big = nullptr;
  long long big_rows = 0;
  for (const ConvLayer* c = L; c && c->has_wino; c = c->alt.get()) {
    if (c->wino_m == 4) { f4 = c; continue; }
    const long long r = wino_padded_rows(*c, B, H, W);
    if (!big || r < big_rows || (r == big_rows && c->wino_m < big->wino_m)) { big = c; big_rows = r; }
  }
  if (!f4) return big ? big : L;
  if (!big) return f4;
  // a larger tile has to pay for its larger rounding error: only where it executes at least a tenth less
  return big_rows * 10 <= wino_padded_rows(*f4, B, H, W) * 9 ? big : f4;
}

// floats of the two Winograd scratch tensors (V: transformed input, M: GEMM output) for an input [B,H,W,*]
inline void wino_scratch_floats(const ConvLayer& L, int B, int H, int W, size_t* v, size_t* m) {
  int th, tw;
  long long n_tiles, m_pad;
  wino_geometry(B, H, W, L.d.dil, &th, &tw, &n_tiles, &m_pad, wino_gran_for(L, B, H, W), L.wino_m);
  *v = (size_t)L.wino_np() * m_pad * L.d.cin;
  *m = (size_t)L.wino_np() * m_pad * L.d.cout;
}

// whether this layer, run by launch_conv_layer on an input [B,H,W,*], can sum a producer's deferred split-K partial tiles
inline bool conv_layer_accepts_deferred(const ConvLayer& L, int B, int H, int W) {
  if (!L.has_wino || L.d.rs || L.wino.rs) return false;
  return wino_input_accepts_deferred(B, H, W, L.d.cin, L.d.dil, wino_gran_for(L, B, H, W), L.wino_m);
}

// One conv layer, as Winograd (input transform -> grouped GEMM -> output transform) when the layer carries
// that form and scratch is supplied, else as the direct kernel.
// `produced`: what the layer that wrote a.x left behind when it skipped its split-K reduce (common.h: DeferredSplit); consumed here.
inline int launch_conv_layer(const ConvLayer& L, const ConvArgs& a, float* wino_v, float* wino_m, hipStream_t s, DeferredSplit* produced = nullptr) {
  const bool deferred_in = produced && produced->valid;
  if (!(L.has_wino && !a.x2 && wino_v && wino_m)) {
    if (deferred_in) return fail(PEANUT_EINVAL, L.name + ": the producer deferred its split-K reduce to a layer that cannot sum it");
    return launch_conv(L.d, a, s);
  }
  int th, tw, rc;
  long long n_tiles, m_pad;
  const int gran = wino_gran_for(L, a.B, a.H, a.W);
  wino_geometry(a.B, a.H, a.W, L.d.dil, &th, &tw, &n_tiles, &m_pad, gran, L.wino_m);
  const long long np = L.wino_np();
  if (np * m_pad > 0x7fffffffLL) return fail(PEANUT_EINVAL, L.name + ": Winograd problem too large");
  rc = launch_wino_input(a.x, wino_v, a.B, a.H, a.W, L.d.cin, L.d.dil, s, gran, L.wino_m, 0, deferred_in ? produced : nullptr);
  if (produced) produced->valid = false;
  if (rc) return rc;
  ConvArgs g{};
  g.x = wino_v; g.y = wino_m;
  g.B = 1; g.H = 1; g.W = (int)(np * m_pad); g.c1 = L.d.cin; g.c2 = 0; g.Ho = 1; g.Wo = g.W;
  g.ws = a.ws; g.ws_floats = a.ws_floats;
  g.mt_per_group = (int)(m_pad / 128); g.w_group_stride = L.wino.rs ? L.wino_group_bytes : L.wino_group_floats;
  if ((rc = launch_conv(L.wino, g, s))) return rc;
  return launch_wino_output(wino_m, L.d.scale, L.d.shift, a.res, a.y, a.B, a.H, a.W, L.d.cout, L.d.dil, L.d.relu, s, gran, L.wino_m);
}

// ---- workspace arena with liveness-based reuse (offsets are planned on the host) ----
struct Arena {
  size_t top = 0;
  bool keep_all = false;
  std::vector<std::pair<size_t, size_t>> free_list;  // (offset, size)
  static size_t round_up(size_t n) { return (n + 255) & ~(size_t)255; }
  size_t alloc(size_t bytes) {
    bytes = round_up(bytes);
    int best = -1;
    for (size_t i = 0; i < free_list.size(); ++i)
      if (free_list[i].second >= bytes && (best < 0 || free_list[i].second < free_list[best].second)) best = (int)i;
    if (best >= 0) {
      const size_t off = free_list[best].first, sz = free_list[best].second;
      free_list.erase(free_list.begin() + best);
      if (sz > bytes) free_list.push_back({off + bytes, sz - bytes});
      return off;
    }
    const size_t off = top;
    top += bytes;
    return off;
  }
  void release(size_t off, size_t bytes) {
    if (keep_all) return;
    bytes = round_up(bytes);
    // coalesce with neighbours so the big stem buffers can be recycled for later stages
    for (bool merged = true; merged;) {
      merged = false;
      for (size_t i = 0; i < free_list.size(); ++i) {
        if (free_list[i].first + free_list[i].second == off) {
          off = free_list[i].first; bytes += free_list[i].second;
          free_list.erase(free_list.begin() + i); merged = true; break;
        }
        if (off + bytes == free_list[i].first) {
          bytes += free_list[i].second;
          free_list.erase(free_list.begin() + i); merged = true; break;
        }
      }
    }
    if (off + bytes == top) { top = off; return; }
    free_list.push_back({off, bytes});
  }
};

struct Act {  // an NHWC activation inside the workspace
  size_t off = 0, bytes = 0;
  int B = 0, H = 0, W = 0, C = 0;
};


inline Act make_act(Arena& a, int B, int H, int W, int C) {
  Act t;
  t.B = B; t.H = H; t.W = W; t.C = C;
  t.bytes = (size_t)B * H * W * C * sizeof(float);
  t.off = a.alloc(t.bytes);
  return t;
}

inline double conv_flops(const ConvLayer* L, const Act& out) {
  return 2.0 * (double)out.B * out.H * out.W * L->d.cout * L->cin_real * L->d.kh * L->d.kw;
}

// name -> tensor of a PyTorch/mmcv/detectron2 state dict handed over the C ABI
struct TensorMap {
  std::map<std::string, const peanut_tensor*> m;
  const peanut_tensor* get(const std::string& k, int ndim, const int64_t* shape, int* rc) const {
    auto it = m.find(k);
    if (it == m.end()) { *rc = fail(PEANUT_EWEIGHTS, "state dict is missing '" + k + "'"); return nullptr; }
    const peanut_tensor* t = it->second;
    bool ok = t->ndim == ndim && t->data != nullptr;
    for (int i = 0; ok && i < ndim; ++i) ok = t->shape[i] == shape[i];
    if (!ok) { *rc = fail(PEANUT_EWEIGHTS, "'" + k + "' has an unexpected shape"); return nullptr; }
    return t;
  }
};

// BN(eval) as y = x*alpha + beta, alpha = weight/sqrt(var+eps), beta = bias - mean*alpha (fp32)
inline int bn_fold_eps(const TensorMap& tm, const std::string& bn, int cout, float eps, float* scale, float* shift) {
  int rc = 0;
  const int64_t cshape[1] = {cout};
  const peanut_tensor* g = tm.get(bn + ".weight", 1, cshape, &rc); if (!g) return rc;
  const peanut_tensor* b = tm.get(bn + ".bias", 1, cshape, &rc); if (!b) return rc;
  const peanut_tensor* mu = tm.get(bn + ".running_mean", 1, cshape, &rc); if (!mu) return rc;
  const peanut_tensor* var = tm.get(bn + ".running_var", 1, cshape, &rc); if (!var) return rc;
  for (int n = 0; n < cout; ++n) {
    const float invstd = 1.0f / sqrtf(var->data[n] + eps);
    const float alpha = invstd * g->data[n];
    scale[n] = alpha;
    shift[n] = b->data[n] - mu->data[n] * alpha;
  }
  return 0;
}

}  // namespace peanut
