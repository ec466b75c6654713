// Tuning options of the library -- kernel gates, Winograd form policy, launch-plan switches.
//
// Every option has a key ("pw256_mink"), a PEANUT_<KEY> environment variable that supplies its process default (read once,
// on first use) and a built-in default.  Handles (peanut_pred_t, peanut_conv_t, peanut_rcnn_t) SNAPSHOT the process
// defaults when they are created and carry their own copy from then on: peanut_*_set_option changes one handle,
// peanut_set_default_option the defaults later handles start from -- so two handles in one process can run different
// policies and tests need no environment games.  While an API call runs on a handle, that handle's copy is the calling
// thread's "active" set (OptionScope); the gates in the kernel sources read opt(OPT_...).
//
// Options that shape the UPLOADED weights (wino_m, wino_head_m, wino6_maxdil, wino5_mindil, wino_flush_ch, wino_min_cin,
// bn64_maxk, pw_bn64_maxk, fp32_bk, rs_conv, rs_bn64_maxk, rcnn_wino_m, rcnn_stem_s2d) are read while a handle is created: set them as
// defaults before the create call (the Python mirrors take `options=`); changing them on a live handle is refused.
#pragma once
#include <stdlib.h>
#include <string.h>

#include <string>

namespace peanut {

enum OptionId {
  OPT_PW_GLDS, OPT_PW256_MINK, OPT_PW256_MINTILES, OPT_PW256_PHASE, OPT_PW256_SKIP_PAD, OPT_PW256W_MINK, OPT_PW256W_MINTILES, OPT_PW256P_MINK, OPT_PW256P_MINTILES, OPT_PW256P_FLUSH, OPT_PW256P_ORDER, OPT_PW256P_STREAMK, OPT_PW256WP_MINK, OPT_PW256WP_MINTILES, OPT_PW256WP_NPRE, OPT_PW256WP_STAGGER, OPT_PW_ARES, OPT_PW_SKINNY,
  OPT_PW_ARES_MINUNITS, OPT_PATCH_MINTILES, OPT_STEM_NCHW, OPT_BN64_MAXK, OPT_PW_BN64_MAXK, OPT_PW64_MAXTILES, OPT_FP32_BK, OPT_NCHUNK, OPT_RES_PREFETCH, OPT_SPLIT_MODEL,
  OPT_RS_CONV, OPT_RS_BN64_MAXK, OPT_RS256_MINK, OPT_RS256_MINTILES, OPT_RS64_MAXK, OPT_RS64_MAXTILES,
  OPT_WINO_M, OPT_WINO_HEAD_M, OPT_WINO6_MAXDIL, OPT_WINO5_MINDIL, OPT_WINO_FLUSH_CH, OPT_WINO_MIN_CIN, OPT_WINO_NARROW_MINPIX, OPT_WINO_SMALL_MAXWG, OPT_DEFER_SPLITK,
  OPT_PPM_OVERLAP, OPT_PPM_GROUPED, OPT_PPM_TERM_ROWS, OPT_PPM_GROUP_ROWS, OPT_RCNN_WINO_M, OPT_RCNN_STEM_S2D, OPT_RCNN_RPN_FUSED, OPT_RCNN_FPN_OVERLAP, OPT_RCNN_RANK_SORT, OPT_RCNN_TOPK_SLICE, OPT_RCNN_NMS_LEVELS, OPT_FMM_LOCAL32, OPT_FMM_MAX_PASSES, OPT_FMM_BLOCKED, OPT_FMM_INNER,
  OPT_COUNT
};

struct OptionInfo {
  const char* key;
  long long def;
  bool upload_time;      // read while a handle is created (shapes the uploaded weights)
  const char* help;
};

inline const OptionInfo* option_table() {
  static const OptionInfo t[OPT_COUNT] = {
      {"pw_glds", 1, false, "fp32 1x1 convs / grouped GEMMs on the LDS-DMA kernels of conv_pw.hip (0: conv_igemm)"},
      {"pw256_mink", 1024, false, "fewest input channels for the 256 x 128 three-stage kernel"},
      {"pw256_mintiles", 256, false, "fewest 256 x 128 tiles for that kernel"},
      {"pw256_phase", 1, false, "the two waves of a SIMD request their LDS-DMA pieces half an iteration apart"},
      {"pw256_skip_pad", 1, false, "256 x 128 kernel, grouped GEMMs: waves whose rows are all group padding issue no MFMAs (0: compute the zero rows)"},
      {"pw256w_mink", 768, false, "fewest input channels for the 256 x 256 two-stage kernel (0: off)"},
      {"pw256w_mintiles", 1536, false, "fewest 256 x 256 tiles for that kernel"},
      {"pw256p_mink", 512, false, "fewest input channels for the persistent 256 x 128 kernel (conv_pw256p.hip; 0: off)"},
      {"pw256p_mintiles", 512, false, "fewest 256 x 128 tiles for that kernel"},
      {"pw256p_flush", 512, false, "most input channels of a Winograd position GEMM (two-level accumulation) that runs on the persistent kernel (0: none)"},
      {"pw256p_order", 1, false, "persistent 256 x 128 kernel: 1 = the workgroups of an XCD walk its run of tiles side by side (neighbours share operand panels in L2), 0 = one contiguous run per workgroup"},
      {"pw256p_streamk", 1, false, "persistent 256 x 128 kernel: the tail tiles' k-tiles as one stream in equal runs per workgroup (0: every tail tile cut into the same number of parts)"},
      {"pw256wp_mink", 512, false, "fewest input channels for the persistent 256 x 256 kernel (conv_pw256wp.hip; 0: off)"},
      {"pw256wp_mintiles", 768, false, "fewest 256 x 256 tiles for that kernel"},
      {"pw256wp_npre", 0, false, "persistent 256 x 256 kernel: accumulator blocks per in-place epilogue group (2 or 4; 0 = two with a residual, four without)"},
      {"pw256wp_stagger", 0, false, "persistent 256 x 256 kernel: spread of the workgroups' start times in sleeps of ~3.4 us (their tile boundaries -- 512 KiB of epilogue traffic per CU -- then fall at different times)"},
      {"pw_ares", 1, false, "K = 128 / 256 pointwise layers on the persistent A-resident kernel (conv_pw_ares.hip)"},
      {"pw_skinny", 1, false, "grouped pointwise launches with at most 64 data rows per group (the PSP pyramid's per-scale convs and Q tables at batch 1) on the skinny weight-streaming kernel (gemm_skinny.hip: 6 KiB of LDS, fits next to any other kernel; 0: the MFMA kernels on 128-row padded tiles)"},
      {"pw_ares_minunits", 512, false, "fewest (m-tile, n-tile) units for that kernel"},
      {"patch_mintiles", 1024, false, "3x3 convs of 16 / 32 input channels on the persistent LDS-patch kernel (conv_patch.hip) from this many 8 x 16 output tiles (0: off)"},
      {"stem_nchw", 1, false, "prediction forward: the first stem conv reads the NCHW input itself (conv_patch.hip, NCHW variant) instead of a layout pass + NHWC conv (fp32 mode, when the patch kernel takes the layer)"},
      {"bn64_maxk", 256, true, "128 x 64 tiles for layers with at most this many input channels"},
      {"pw_bn64_maxk", 128, true, "pointwise layers / Winograd position GEMMs: weights PACKED 64 wide up to this many input channels (wider layers: 128-wide packing, 64-wide tiles chosen per shape)"},
      {"pw64_maxtiles", 512, false, "a 128-wide-packed pointwise layer with at most bn64_maxk input channels runs 128 x 64 tiles while its 128 x 128 tiling has fewer tiles than this (two per CU)"},
      {"fp32_bk", 0, true, "16: force 16-channel k-tiles in conv_igemm (experiment)"},
      {"nchunk", 8, false, "n-tiles per chunk of the tile order (0: n fastest over all n-tiles)"},
      {"res_prefetch", 1, false, "request the first residual rows before the last k-tile's MFMAs"},
      {"split_model", 1, false, "tail split-K plan counts CUs (1) or resident slots (0)"},
      {"rs_conv", 1, true, "emulated modes: non-pointwise convs on conv_rs.hip (0: fp32 MFMA kernel)"},
      {"rs_bn64_maxk", 128, true, "emulated modes: 128 x 64 tiles up to this many input channels"},
      {"rs256_mink", 512, false, "emulated modes: fewest input channels for the 256 x 256 kernel"},
      {"rs256_mintiles", 256, false, "emulated modes: fewest 256 x 256 tiles for it"},
      {"rs64_maxk", 512, false, "emulated modes: 64 x 64 tiles up to this many input channels"},
      {"rs64_maxtiles", 128, false, "emulated modes: ... and below this many 128-row tiles"},
      {"wino_m", 0, true, "4 / 5 / 6: force one Winograd form on every eligible layer (0: policy)"},
      {"wino_head_m", 0, true, "4 / 5 / 6: Winograd form of the PSP bottleneck (0: policy)"},
      {"wino6_maxdil", 4, true, "largest dilation that gets an F(6x6) form"},
      {"wino5_mindil", 4, true, "smallest dilation that also gets an F(5x5) form (0: none)"},
      {"wino_flush_ch", 64, true, "channels per partial sum of the position GEMMs' two-level accumulation (0: off)"},
      {"wino_min_cin", 0, true, "fewest input channels of a Winograd layer (0: the planner's policy)"},
      {"wino_narrow_minpix", 20000, false, "layers under 128 channels take their Winograd form from this many input pixels on (round 6: 100 000 -> 20 000 -- with the small-problem transforms one 720 x 720 map's layer1 conv2, 32 400 pixels, gains 5 us per layer as F(6x6); profiles/r9e)"},
      {"wino_small_maxwg", 1024, false, "Winograd transforms: launches with fewer workgroups than this of the one-thread-per-tile kernels take the small-problem variants (a workgroup per tile and 64-channel slice, a thread per line, through LDS; bit-identical; 0: never)"},
      {"defer_splitk", 1, false, "a Bottleneck conv1 whose every tile is split along k leaves its partial tiles unsummed and conv2's (small-problem) Winograd input transform sums them: one launch fewer per block at batch 1, bit-identical (0: every conv runs its own reduce); read when a (B, H, W) plan is built"},
      {"ppm_overlap", -1, false, "pyramid branch of the PSP head on a side stream: 0 / 1, -1 = by size"},
      {"ppm_grouped", -1, false, "per-scale PSP GEMMs as one grouped launch: 0 / 1, -1 = by size"},
      {"ppm_term_rows", 1, false, "folded pyramid term: one wave per output row, no barrier in the row loop (0: the workgroup-wide two-phase kernel)"},
      {"ppm_group_rows", 0, false, "pyramid pooling, row pass: rows added up per workgroup (1..5; 0 = by size: as many as still leave ~768 workgroups)"},
      {"rcnn_wino_m", 0, true, "detector front end: 4 / 5 / 6 pins one Winograd form (0: per shape)"},
      {"rcnn_stem_s2d", 1, true, "detector stem 7x7 stride 2 as a space-to-depth 4x4 conv"},
      {"rcnn_rpn_fused", 1, false, "detector: the RPN head on all five pyramid levels as one chain (five input transforms, ONE grouped Winograd GEMM, five output transforms, one objectness and one anchor-delta GEMM over all levels' rows: 12 launches instead of 25); read when a (B, H, W) plan is built"},
      {"rcnn_fpn_overlap", 0, false, "detector: the FPN output convs of p5, p4, p3 on a side stream next to the lateral / top-down chain that ends in p2's (round 5 experiment, bit-identical; measured SLOWER at batch 1, 5.70 against 5.61 ms per frame, profiles/r8: off); read when a (B, H, W) plan is built"},
      {"rcnn_rank_sort", 1, false, "detector: the RPN candidates of an image are ordered by counting the larger keys (n / 64 workgroups) instead of one workgroup's bitonic sort (round 5; same order: the keys are distinct)"},
      {"rcnn_topk_slice", 20480, false, "detector: objectness logits per workgroup of the per-level top-k (a level is cut into up to 8 ranges whose k largest are merged by counting; 0: one workgroup per level, round 4)"},
      {"rcnn_nms_levels", 1, false, "detector: the proposal NMS level by level (five 16 x 16-word suppression blocks and five scans per image side by side, post-NMS top-k by counting) instead of one score-sorted list per image (round 5; same proposals)"},
      {"fmm_local32", 1, false, "goal solver: single-precision local solve inside a tile"},
      {"fmm_max_passes", 24, false, "goal solver: hard ceiling of the second-order ordering passes; they stop as soon as a pass changes nothing (6-10 passes on the agent's 960 x 960 map; peanut_goal_converged reports whether they reached their fixed point)"},
      {"fmm_blocked", 1, false, "goal solver: a wave relaxes its 8 x 8 block of the tile to convergence between workgroup barriers (round 5; 0: one Jacobi sweep of the whole tile per barrier pair)"},
      {"fmm_inner", 48, false, "goal solver: iterations a wave spends on its block before it publishes it (blocked rounds)"},
  };
  return t;
}

struct Options {
  long long v[OPT_COUNT];
};

inline int option_index(const char* key) {
  if (!key) return -1;
  std::string k(key);
  for (auto& c : k) c = (char)((c >= 'A' && c <= 'Z') ? c - 'A' + 'a' : c);
  if (k.rfind("peanut_", 0) == 0) k = k.substr(7);
  const OptionInfo* t = option_table();
  for (int i = 0; i < OPT_COUNT; ++i)
    if (k == t[i].key) return i;
  return -1;
}

// process defaults: built-in values overridden by PEANUT_<KEY> (read once), then by peanut_set_default_option
inline Options& default_options() {
  static Options d = [] {
    Options o;
    const OptionInfo* t = option_table();
    for (int i = 0; i < OPT_COUNT; ++i) {
      o.v[i] = t[i].def;
      std::string env = "PEANUT_";
      for (const char* c = t[i].key; *c; ++c) env += (char)((*c >= 'a' && *c <= 'z') ? *c - 'a' + 'A' : *c);
      const char* e = getenv(env.c_str());
      if (e && *e) o.v[i] = atoll(e);
    }
    return o;
  }();
  return d;
}

inline const Options*& active_options_ptr() {
  static thread_local const Options* p = nullptr;
  return p;
}

inline long long opt(OptionId id) {
  const Options* a = active_options_ptr();
  return (a ? a : &default_options())->v[id];
}

// RAII: the options of the handle an API call runs on become the calling thread's active set
struct OptionScope {
  const Options* saved;
  explicit OptionScope(const Options* o) : saved(active_options_ptr()) { active_options_ptr() = o; }
  ~OptionScope() { active_options_ptr() = saved; }
  OptionScope(const OptionScope&) = delete;
  OptionScope& operator=(const OptionScope&) = delete;
};

}  // namespace peanut
