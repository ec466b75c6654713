// Types shared by the Mask R-CNN translation units (rcnn_api.hip: front end; rcnn_post.hip: proposal selection,
// ROI heads, detection selection, mask pasting -- the whole of peanut_rcnn_inference).
#pragma once
#include "net_common.h"

namespace peanut {

enum ROpKind { R_PREPROCESS, R_CONV, R_MAXPOOL, R_ADD_UP, R_SUBSAMPLE, R_RPN_FUSED };

struct ROp {
  ROpKind kind;
  std::string name, kernel;
  const ConvLayer* conv = nullptr;
  Act in, in2, res, out;      // in2: second source of a two-source pointwise layer (channels [in.C, in.C + in2.C))
  Act wino_v, wino_m;         // scratch of the Winograd form (R_CONV of a layer that carries one)
  bool has_res = false, has_wino = false, has_in2 = false;
  double flops = 0;
  float* ext_out = nullptr;   // filled at run time for ops that write a caller-owned output
  int ext_slot = -1;          // 0..4 = p2..p6, 5..9 = objectness, 10..14 = deltas
  int in_ext_slot = -1;       // the input is the caller-owned output of that slot (read there when the caller gave a buffer)
  bool rpn_level = false;     // one of the fifteen per-level RPN-head launches: skipped when the fused form (R_RPN_FUSED) runs
  bool side = false;          // FPN output conv of p3..p5: may run on the handle's side stream next to the top-down chain (round 5)
  bool join_side = false;     // the first op that reads a side op's output: the main stream waits for the side stream before it
  bool defer_ok = false;      // R_CONV: the next op is the Winograd layer that alone reads this output and can sum split-K partial tiles itself
                              // (common.h: DeferredSplit): this op may skip its split-K reduce (round 6, option defer_splitk)
};

// The RPN head on all five pyramid levels as ONE chain (round 5): the shared 3x3 conv as Winograd with the levels' tiles side by
// side in every position's rows -- five input transforms, ONE grouped GEMM, five output transforms into one [rows, C] buffer -- then
// the objectness and the anchor-delta layer as one pointwise GEMM each over all levels' rows: 12 launches instead of 25.
struct RpnFused {
  bool on = false;
  const ConvLayer* form = nullptr;            // the Winograd form all levels share (the one level p2's shape picks)
  Act in[5], t_all, obj_all, dl_all, v, m;    // level inputs (p2..p6), conv output of all levels, head outputs, Winograd scratch
  long long tile_off[5] = {0}, row_off[5] = {0}, rows = 0, m_pad_total = 0;
  int gran = 128;
};

struct RPlan {
  int B = 0, H = 0, W = 0, nh = 0, nw = 0, Hp = 0, Wp = 0;
  Act splitk, splitk_side;    // split-K scratch of the main chain / of the ops on the side stream
  size_t bytes = 0;
  std::vector<ROp> ops;
  int lvl_h[5] = {0}, lvl_w[5] = {0};
  RpnFused rpn;
  // fixed-point coefficients of the resize (rcnn_api.hip: resize_tables): per output column / row the first source
  // index, the tap count and `ks` int32 weights
  DevBuf rz_x, rz_y;
  int ksx = 0, ksy = 0;
};


}  // namespace peanut

struct peanut_rcnn {
  peanut::Options opts = peanut::default_options();   // this handle's tuning options (options.h)
  peanut_rcnn_cfg cfg{};
  std::vector<std::unique_ptr<peanut::ConvLayer>> convs;
  peanut::ConvLayer* stem = nullptr;
  peanut::ConvLayer* stem_s2d = nullptr;   // the same conv on the 2x2 space-to-depth input (rcnn_api.hip), when applicable
  struct Block { peanut::ConvLayer *shortcut, *c1, *c2, *c3, *c3s; };   // c3s: conv3 + stride-1 shortcut as one two-source GEMM
  std::vector<std::vector<Block>> stages;
  peanut::ConvLayer *lateral[4] = {nullptr}, *output[4] = {nullptr};   // levels 2..5
  peanut::ConvLayer *rpn_conv = nullptr, *rpn_obj = nullptr, *rpn_delta = nullptr;
  std::map<std::string, std::unique_ptr<peanut::RPlan>> plans;
  std::vector<std::string> probe_families;   // peanut_rcnn_probe_front: kernel family per op of the last probe
  peanut::DevBuf ws;
  // ROI heads (present when the state dict carries roi_heads.*): FC layers as 1x1 convs, mask head convs
  bool has_heads = false;
  peanut::ConvLayer *fc1 = nullptr, *fc2 = nullptr, *cls_score = nullptr, *bbox_pred = nullptr, *deconv = nullptr, *mask_pred = nullptr;
  std::vector<peanut::ConvLayer*> mask_fcn;
  struct PostBufs;                       // scratch of peanut_rcnn_inference (rcnn_post.hip)
  std::shared_ptr<PostBufs> post;
  // FPN side stream (rcnn_api.hip): the output convs of p5, p4, p3 next to the lateral / top-down chain that ends in p2's
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  ~peanut_rcnn() {
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (side) (void)hipStreamDestroy(side);
  }
};


// rcnn_post.hip: builds the ROI-head layers when the state dict carries roi_heads.* (called by peanut_rcnn_create)
int peanut_rcnn_build_heads(peanut_rcnn* h, const peanut::TensorMap& tm);
