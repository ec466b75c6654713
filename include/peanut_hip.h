/* peanut_hip -- C ABI of the MI355X-native PEANUT perception hot path (libpeanut_hip.so).
 *
 * The reference (ajzhai/PEANUT) has no plugin/FFI layer: its boundary for this path is three
 * Python call surfaces.  Each entry point below states the reference interface it replaces
 * (paths relative to the reference checkout).  Conventions:
 *   - every `const float* dev` / `float* dev` is a DEVICE pointer (e.g. torch.Tensor.data_ptr());
 *     the caller owns every buffer; `host` pointers are marked as such;
 *   - functions enqueue on the given hipStream_t (passed as void*; NULL = default stream) and do
 *     not synchronise unless stated;
 *   - return 0 on success, a negative PEANUT_E* code otherwise; peanut_last_error() returns a
 *     thread-local message for the last failure;
 *   - a handle is not thread-safe, distinct handles are; the process DEFAULTS of the tuning options (peanut_set_default_option) are
 *     one unlocked table that every peanut_*_create snapshots: callers that create handles on several threads while changing
 *     defaults serialise those calls themselves (the Python mirrors do, under one lock).
 */
#ifndef PEANUT_HIP_H_
#define PEANUT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PEANUT_OK 0
#define PEANUT_EINVAL (-2)   /* bad argument / unsupported configuration */
#define PEANUT_EHIP (-3)     /* HIP runtime error */
#define PEANUT_EWEIGHTS (-4) /* missing or mis-shaped tensor in the state dict */
#define PEANUT_ERANGE (-5)   /* precision FP16X3: a value left fp16's exponent range (the result would be NaN / dropped) */

const char* peanut_last_error(void);
/* kernel family the calling thread's most recent conv / GEMM launch selected, e.g. "conv_pw_glds_256x128", "conv_pw_glds_256x256", "conv_pw_ares_128x128" (for tests
 * and profiles: lets a parity test assert which kernel produced the result it checked) */
const char* peanut_last_conv_kernel(void);
/* library / ABI version and the arch it was compiled for ("gfx950") */
int peanut_abi_version(void);
const char* peanut_build_arch(void);
/* hash of the sources this library was compiled from (peanut_amd/build.py: source_hash), "" when built by other means: the
 * Python binding compares it with the sources lying next to it and rebuilds or refuses a stale library */
const char* peanut_source_hash(void);
/* Host-side test hook (no GPU needed): the pieces the weight packer of an emulated mode makes of n host values that
 * form ONE layer -- precision = PEANUT_PREC_BF16X3 / FP16X3 (two pieces) or BF16X6 (three).  pieces: [3][n] 16-bit
 * patterns (bf16 or fp16; plane 2 zero for the two-piece modes); *pack_scale = the power of two the values were multiplied
 * by first (1 for the bf16 modes).  tests/test_abi.py compares them with numpy's / torch's own roundings. */
int peanut_debug_weight_pieces(const float* values, int n, int precision, unsigned short* pieces, float* pack_scale);
/* Host-side test hook (no GPU needed): the Winograd weight transform U = G g G^T of a [cout][cin][3][3] layer for output
 * tiles of tile x tile (4, 5 or 6; csrc/winograd.hip), as the uploader computes it (double arithmetic, rounded once):
 * out [(tile + 2)^2][cout][cin].  tests/test_abi.py holds it against the Toom-Cook construction in exact rationals. */
int peanut_debug_wino_weights(const float* w_oihw, int cout, int cin, int tile, float* out);
/* Test hook (ABI 14): how many Winograd input transforms of this process have summed their producer's split-K partial tiles
 * themselves (csrc/common.h DeferredSplit; option defer_splitk): a Bottleneck conv1 at batch 1 then runs no reduce launch.  Replaces
 * nothing in the reference (resnet.py:267-307 conv1 -> bn1 -> relu -> conv2 are four module calls there); lets a test assert
 * that the fused path really ran while the results stay bit-identical. */
long long peanut_debug_deferred_splitk_count(void);
/* Test hook (ABI 14): an LDS canary.  Enqueues `workgroups` workgroups of 256 threads on `stream`, each of which fills `lds_bytes`
 * (<= 64 KiB, a multiple of 1024) of its LDS with a pattern and then, `rounds` times, sleeps a little and checks it; words found
 * changed are counted into *mismatches (device pointer, int, the caller zeroes it).  Run next to another kernel on a second stream it
 * shows whether that kernel writes LDS outside its own allocation (a workgroup that shares the CU would see it).  rounds < 0: |rounds|
 * rounds without the sleep and with a sweep of broadcast 16-byte reads over the whole array in each -- an LDS-heavy neighbour.  Nothing in
 * the reference corresponds to it. */
int peanut_debug_lds_canary(int workgroups, int lds_bytes, int rounds, int* mismatches, void* stream);
/* Test hook (ABI 15): a packed-FMA canary.  Workgroups of 256 threads that compute, `rounds` times, twelve 64-term dot products per thread
 * three ways on the same operands, two sums per packed instruction: `v_pk_fma_f32 ... op_sel:[0,1,0]` (the shared operand in the HIGH
 * register of its pair: the form hipcc emits when it packs scalar code), `v_pk_fma_f32 ... op_sel_hi:[1,0,1]` (the operand in the LOW
 * register) and two scalar v_fmac_f32.  mismatches (device pointer, TWO ints, zeroed by the caller): [0] sums where the first form differs
 * from the scalar one, [1] where the second does.  The same arithmetic: any difference is the hardware.  Measured on gfx950 (profiles/r9r):
 * [0] > 0 while fp16 / bf16 MFMA waves share the SIMD (the emulated modes' GEMM kernels), 0 otherwise; [1] always 0. */
int peanut_debug_pkfma_canary(int workgroups, int rounds, int* mismatches, void* stream);

/* ------------------------------------------------------------------------------------------
 * Tuning options (csrc/options.h): kernel gates, Winograd form policy, launch-plan switches -- named by key, e.g.
 * "pw256_mink"; peanut_option_list() returns one line per option: `key=default [create-time] help`.
 * The process defaults come from the PEANUT_<KEY> environment variables (read once) and peanut_set_default_option.  A
 * handle (peanut_pred_t / peanut_conv_t / peanut_rcnn_t) SNAPSHOTS the defaults when it is created and carries its own
 * copy: peanut_*_set_option changes that one handle (and drops its cached launch plans), so two handles in one process
 * can run different policies.  Options marked [create-time] shape the uploaded weights (Winograd forms, packing tiles):
 * set them as defaults before the create call; peanut_*_set_option refuses them.
 * ---------------------------------------------------------------------------------------- */
int peanut_set_default_option(const char* key, long long value);
int peanut_get_default_option(const char* key, long long* value);
const char* peanut_option_list(void);

/* ------------------------------------------------------------------------------------------
 * Stage 3 -- map-completion forward (PSPNet: ResNet-50-V1c-D8 + PSP head)
 * ---------------------------------------------------------------------------------------- */

/* Fields of nav/pred_model_cfg.py:2-42 that the inference path reads. */
typedef struct peanut_pred_cfg {
  int in_channels;       /* backbone.in_channels (14) */
  int num_classes;       /* decode_head.num_classes (6) */
  int strides[4];        /* backbone.strides (1,2,1,1) */
  int dilations[4];      /* backbone.dilations (1,1,2,4) */
  int contract_dilation; /* backbone.contract_dilation (True) */
  int pool_scales[8];    /* decode_head.pool_scales (1,2,3,6) */
  int n_pool_scales;
  int head_channels;     /* decode_head.channels (512) */
  int align_corners;     /* decode_head.align_corners (False) */
  float bn_eps;          /* nn.BatchNorm2d default 1e-5 */
  int precision;         /* PEANUT_PREC_*: arithmetic of the conv contractions (see below) */
  int fold_ppm;          /* 1: evaluate the pyramid half of the PSP bottleneck conv through linearity
                            (conv of a bilinear upsample of k*k vectors = bilinear blend of k*k folded
                            vectors; halves that conv's FLOPs, fp32 re-association only); 0: plain conv */
  int conv_algo;         /* PEANUT_ALGO_*: algorithm of the stride-1 3x3 convs with >= 64 input channels */
} peanut_pred_cfg;

/* PEANUT_ALGO_AUTO: Winograd with fp32 transforms, the form chosen per layer and shape -- F(6x6,3x3), F(5x5,3x3) (dilation-4
 * layers of the prediction backbone) or F(4x4,3x3) in the prediction model and the detector's front end, F(4x4,3x3) for
 * single convs, the detector heads and the PSP bottleneck of the emulated modes -- (what cuDNN/MIOpen pick for these layers
 * in the reference's own GPU runs): 4x / 4.6x / 5.1x fewer multiplies; well-conditioned interpolation points (0, +-3/4,
 * +-3/2, inf; 0, +-1/2, +-1, 3, inf; 0, +-1/2, +-1, +-2, inf) and position GEMMs that accumulate in two levels on the fp32
 * kernels (partial sums of 64 channels) keep the logits within 1e-5 max-abs of the reference golden vectors (direct form
 * 9e-6, bound 1e-3).
 * AUTO also runs conv3 and a stride-1 downsample / shortcut conv of a bottleneck block as ONE GEMM over [conv2 output | block
 * input] with the two BatchNorm scales folded into the weights (same function, another rounding order).
 * PEANUT_ALGO_DIRECT: every conv as the direct implicit GEMM (products summed exactly as an fmaf chain), every block in
 * the reference's op-for-op form (conv -> BN, conv -> BN, add, ReLU). */
#define PEANUT_ALGO_AUTO 0
#define PEANUT_ALGO_DIRECT 1

/* Conv arithmetic.  FP32: v_mfma_f32_32x32x2_f32 -- exact fp32 products, fp32 accumulation.  The pointwise GEMM families sum
 * a layer's products in ONE order and agree bit for bit with each other and with an fmaf chain in that order (only a tail
 * split-K cuts K differently by tile size); the summation order of the other kernels is kernel-dependent: the LDS-patch stem
 * kernel (csrc/conv_patch.hip) keeps two accumulator chains, the Winograd position GEMMs accumulate in two levels, and the
 * row groups of the pyramid pooling (option ppm_group_rows = 0: chosen from H * B, so that ~768 workgroups run) make the last
 * bits of the PSP pooling sums depend on the BATCH SIZE as well as on the position in the batch -- all within ~1e-6 relative.
 * BF16X6: fp32 emulated on the bf16 matrix cores.  Activations stay fp32 in HBM and LDS -- the forward's tensors,
 * Winograd transforms and fusions are exactly the fp32 mode's; every fp32 value is split into three bf16 pieces (3 x 8 =
 * 24 mantissa bits: the split is exact; the activations in registers after the fragment read, the weights once at load
 * time) and every product is rebuilt from the six piece products that matter (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo,
 * lo*hi; the dropped ones are <= 2^-24 relative, the size of fp32's own product rounding) on v_mfma_f32_32x32x16_bf16
 * with fp32 accumulation: fp32-class results, measured as close to a float64 run of the reference model as the
 * reference's own fp32 CPU path.  The 1x1 convs and the Winograd position GEMMs run on csrc/gemm_rs.hip, every other conv
 * (3x3 direct, strided, the stem) on csrc/conv_rs.hip -- the same arithmetic with im2col-free staging; the Winograd
 * transforms themselves are fp32 in every mode.
 * BF16X3: the same with two pieces and three products (~2^-16 relative per product: ~7e-5 max-abs on the logits,
 * bound 1e-3); an opt-in speed mode, not fp32-class.
 * FP16X3: two FP16 pieces per value (2 x 11 = 22 significand bits) and the three products hi*hi, hi*lo, lo*hi on
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation -- the cost of BF16X3 at close to BF16X6's accuracy (dropped / rounded
 * terms <= 2^-21.5 relative per product).  What it gives up is fp32's exponent range in the ACTIVATIONS of the emulated
 * layers: they must stay below 65504 in magnitude (Winograd-domain values included), and the low piece of a value below
 * 2^-3 is an fp16 subnormal (absolute error <= 3e-8, fp32's own rounding of a value near 0.5).  Weights are scaled by a
 * per-layer power of two before the split (undone exactly in the epilogue), so their range does not matter.  Opt-in. */
#define PEANUT_PREC_FP32 0
#define PEANUT_PREC_BF16X3 1
#define PEANUT_PREC_FP16X3 2
#define PEANUT_PREC_BF16X6 3

/* One entry of an mmcv/PyTorch state dict (HOST memory, fp32, contiguous, OIHW for convs). */
typedef struct peanut_tensor {
  const char* name;  /* e.g. "backbone.layer1.0.conv1.weight" */
  const float* data; /* host */
  int ndim;
  int64_t shape[4];
} peanut_tensor;

typedef struct peanut_pred peanut_pred_t;

/* Replaces init_segmentor (prediction/mmseg/apis/inference.py:12-40) as called from
 * PEANUT_Prediction_Model.__init__ (nav/agent/prediction.py:142-152): builds the layer table
 * from cfg, looks every tensor up BY NAME in the mmcv state-dict key schema
 * (auxiliary_head.* and num_batches_tracked entries are ignored), folds BatchNorm into a
 * per-channel scale/shift, repacks conv weights into the kernel tile layout and uploads them.
 * Synchronous. */
int peanut_pred_create(peanut_pred_t** out, const peanut_pred_cfg* cfg, const peanut_tensor* tensors,
                       int n_tensors);
void peanut_pred_destroy(peanut_pred_t* h);

/* Replaces run_inference + model(return_loss=False, rescale=True) + (optionally) expit:
 * nav/agent/prediction.py:112-137,155-158 -> prediction/mmseg/models/segmentors/
 * encoder_decoder.py:70-80,203-271.  in_dev: [B, in_channels, H, W] NCHW fp32;
 * out_dev: [B, num_classes, H, W] NCHW fp32 raw logits (apply_sigmoid=0, what simple_test
 * returns in the fork) or probabilities (apply_sigmoid=1, what get_prediction returns).
 * The activation workspace for a (B,H,W) shape is allocated on first use and cached. */
int peanut_pred_forward(peanut_pred_t* h, const float* in_dev, float* out_dev, int B, int H, int W,
                        int apply_sigmoid, void* stream);

/* Bytes of activation workspace a (B,H,W) forward needs (0 on error). */
size_t peanut_pred_workspace_bytes(peanut_pred_t* h, int B, int H, int W);

/* Conv FLOPs (2 x MACs over the 61 convolutions) of one HxW map. */
double peanut_pred_flops_per_map(peanut_pred_t* h, int H, int W);

/* Test/bisect hook: keep every intermediate of subsequent forwards in its own buffer
 * (keep=1) and fetch one by name after a forward.  Names: "stem0","stem1","stem2","pool",
 * "layer1".."layer4","ppm_table","bottleneck","logits_lowres".  *dev_out points into the
 * handle's workspace (NHWC; dims = {B,H,W,C}; "ppm_table" is {B,1,nbins,C}). */
int peanut_pred_debug_keep(peanut_pred_t* h, int keep);
int peanut_pred_debug_tensor(peanut_pred_t* h, const char* name, const float** dev_out, int dims[4]);
/* Same lookup, but copies the tensor into dst_dev (device, >= max_floats capacity) on `stream`;
 * dst_dev == NULL only reports dims. */
int peanut_pred_debug_read(peanut_pred_t* h, const char* name, float* dst_dev, size_t max_floats, int dims[4],
                           void* stream);

/* Event probe (used by bench.py for the roofline figure): while enabled, every forward records a
 * HIP event before/after each launch ON THE STREAM THE KERNELS RUN ON.  collect() synchronises
 * on those events, sums each op's elapsed time over all forwards since the last collect and
 * clears the probe.  names/kernels point to storage owned by the handle (valid until the plan
 * is dropped); kernels[i] is the kernel family op i launches (e.g. "conv_pw_glds_128x128");
 * flops[i] = FLOPs one launch of op i executes; bytes[i] = its algorithmic HBM bytes (every operand read
 * once, the result written once).  Returns the op count. */
int peanut_pred_probe_enable(peanut_pred_t* h, int enable);
int peanut_pred_probe_collect(peanut_pred_t* h, int max_ops, const char** names, const char** kernels,
                              double* ms_sum, double* flops, double* bytes, int* n_forwards);
/* hipGraph replay: with enable = 1 the launch sequence of a (B,H,W, in, out, apply_sigmoid, stream) combination is
 * captured on its second use and replayed as ONE hipGraphLaunch afterwards (~85 kernels per forward; measured on
 * MI355X the small-batch cases are bound by the dependent chain on the device, not by launches: no gain).
 * Up to 16 combinations are cached; growing the workspace drops them.  Results are identical.  Ignored while the probe or debug taps are on, and on the legacy default stream (NULL), which HIP
 * cannot capture: pass a created stream. */
int peanut_pred_use_graph(peanut_pred_t* h, int enable);
/* Tuning options of this handle (see above).  Not while the probe is enabled. */
int peanut_pred_set_option(peanut_pred_t* h, const char* key, long long value);
int peanut_pred_get_option(peanut_pred_t* h, const char* key, long long* value);

/* ------------------------------------------------------------------------------------------
 * Stage 2 -- egocentric -> allocentric semantic-map projection
 * ---------------------------------------------------------------------------------------- */

/* The `args` fields Semantic_Mapping.__init__ reads (nav/agent/mapping.py:15-37); python floats are
 * doubles here and are narrowed to fp32 exactly where torch narrows them. */
typedef struct peanut_map_cfg {
  int frame_height, frame_width;   /* 120, 160 */
  int map_resolution;              /* 5 (cm per cell) */
  int map_size_cm;                 /* 4800 */
  int global_downscaling;          /* 2 */
  int vision_range;                /* 100 */
  double hfov;                     /* 79.0 */
  int du_scale;                    /* 1 (1..8: depth pixels taken every du_scale-th row / column, depth_utils.py:129-149) */
  double cat_pred_threshold, exp_pred_threshold, map_pred_threshold;   /* 5.0, 1.0, 0.1 */
  int num_sem_categories;          /* 10 */
  double camera_height;            /* 0.88 (m) */
} peanut_map_cfg;

typedef struct peanut_map peanut_map_t;

/* Replaces Semantic_Mapping.__init__ (mapping.py:12-50): derives the constants (camera matrix,
 * z bins, paste window) and allocates the per-frame scratch (a few MB).  Synchronous. */
int peanut_map_create(peanut_map_t** out, const peanut_map_cfg* cfg);
void peanut_map_destroy(peanut_map_t* h);
/* dims = {C (= 4 + num_sem_categories), M (local map cells), vision_range, frame points} */
int peanut_map_dims(peanut_map_t* h, int dims[4]);

/* Replaces Semantic_Mapping.forward (mapping.py:52-179) for batch size 1.  All pointers are device
 * fp32: obs [1,C,h,w] (ch 3 = depth in cm, ch 4.. = semantic), pose_obs [3] = (dx, dy, dtheta),
 * maps_last [C,M,M], poses_inout [3] = (x m, y m, theta deg) updated IN PLACE (the reference's
 * returned pose_pred / current_poses both alias poses_last, mapping.py:143-160), fp_map_pred
 * [1,V,V], map_pred [C,M,M] (must not alias maps_last).  Enqueues 7 launches (all own kernels; the points are grouped by voxel without a sort:
 * count, claim a segment, fill, rank inside the segment), no host sync. */
int peanut_map_forward(peanut_map_t* h, const float* obs, const float* pose_obs, const float* maps_last,
                       float* poses_inout, float* fp_map_pred, float* map_pred, void* stream);
/* The bookkeeping Agent_State.update_local_map does on the local map after the projection
 * (nav/agent/agent_state.py:281-296), in one launch instead of eight small tensor operations:
 *   local_map[2, :, :] = 0;  local_map[2:4, r0:r1, c0:c1] = 1  (the trajectory square `loc - 2 : loc + 3`, passed as the
 *   NORMALISED Python slice, 0 <= r0, r1 <= m; empty when r0 >= r1);  local_map[1][selem_rows + cr, selem_cols + cc] = 1 for
 *   up to two centres (the agent's cell; the current goal when the agent is within goal_reached_dist of it), where the
 *   footprint is the non-zero cells of selem (device uint8 [(2R+1), (2R+1)], `disk(col_rad + 1)`) offset by -R.  Negative
 *   indices wrap like torch's; a footprint index outside [-m, m) is refused (PEANUT_EINVAL; torch raises IndexError).
 * local_map: device fp32 [channels, m, m], channels >= 4.  No host sync. */
int peanut_map_mark_agent(float* local_map, int channels, int m, int r0, int r1, int c0, int c1, const uint8_t* selem,
                          int selem_radius, int n_centres, const int* centres_rc, void* stream);
/* hipGraph replay of the step's launches, keyed on the seven pointer/stream arguments (an agent ping-pongs two map
 * buffers: two cached graphs). */
int peanut_map_use_graph(peanut_map_t* h, int enable);

/* ------------------------------------------------------------------------------------------
 * Stage 1 -- Mask R-CNN front end (preprocessing + ResNet-FPN backbone + RPN head)
 *
 * Replaces the dense-convolution part of detectron2's DefaultPredictor as built by
 * SemanticPredMaskRCNN.__init__ (nav/agent/utils/segmentation.py:30-38) from
 * COCO-InstSeg/mask_rcnn_R_101_cat9.yaml.  detectron2 is third party and absent from the reference
 * checkout: the module graph follows its published v0.6 definitions, parity is pinned only against
 * the restatement in oracle/rcnn_ref.py.  Proposal selection, NMS, ROIAlign and mask pasting are the operator
 * exports further down; peanut_rcnn_inference runs the whole detector.
 * ---------------------------------------------------------------------------------------- */
typedef struct peanut_rcnn_cfg {
  int depth;               /* RESNETS.DEPTH (101) */
  int stem_out;            /* RESNETS.STEM_OUT_CHANNELS (64) */
  int res2_out;            /* RESNETS.RES2_OUT_CHANNELS (256) */
  int stride_in_1x1;       /* RESNETS.STRIDE_IN_1X1 (true) */
  int fpn_out;             /* FPN.OUT_CHANNELS (256) */
  int num_anchors;         /* len(ANCHOR_GENERATOR.ASPECT_RATIOS[0]) (3) */
  int min_size, max_size;  /* INPUT.MIN_SIZE_TEST / MAX_SIZE_TEST (800, 1333) */
  int size_divisibility;   /* 32 */
  float pixel_mean[3], pixel_std[3]; /* BGR (103.53, 116.28, 123.675), (1, 1, 1) */
  float bn_eps;            /* FrozenBatchNorm2d eps 1e-5 */
  int precision;           /* PEANUT_PREC_* */
  int conv_algo;           /* PEANUT_ALGO_* (Winograd for the stride-1 3x3 convs with >= 128 input channels) */
  /* proposal generator and ROI heads (yaml :41-57, :163-253, :312); read by peanut_rcnn_inference only */
  float anchor_sizes[5];   /* ANCHOR_GENERATOR.SIZES, one per level p2..p6 (32, 64, 128, 256, 512) */
  float aspect_ratios[8];  /* ANCHOR_GENERATOR.ASPECT_RATIOS (0.5, 1, 2): num_anchors entries */
  int rpn_pre_nms_topk;    /* RPN.PRE_NMS_TOPK_TEST (1000; <= 1024) */
  int rpn_post_nms_topk;   /* RPN.POST_NMS_TOPK_TEST (1000) */
  float rpn_nms_thresh;    /* RPN.NMS_THRESH (0.7) */
  float rpn_bbox_weights[4];   /* RPN.BBOX_REG_WEIGHTS (1, 1, 1, 1) */
  int num_classes;         /* ROI_HEADS.NUM_CLASSES (9) */
  int box_pooler_resolution, mask_pooler_resolution;   /* 7, 14 */
  int fc_dim;              /* ROI_BOX_HEAD.FC_DIM (1024), NUM_FC 2 */
  int mask_conv_dim, num_mask_convs;                   /* ROI_MASK_HEAD.CONV_DIM (256), NUM_CONV (4) */
  float roi_bbox_weights[4];   /* ROI_BOX_HEAD.BBOX_REG_WEIGHTS (10, 10, 5, 5) */
  float score_thresh_test; /* ROI_HEADS.SCORE_THRESH_TEST (segmentation.py:33 sets it to sem_pred_prob_thr) */
  float nms_thresh_test;   /* ROI_HEADS.NMS_THRESH_TEST (0.5) */
  int detections_per_image;    /* TEST.DETECTIONS_PER_IMAGE (100) */
  float mask_threshold;    /* detector_postprocess (0.5) */
} peanut_rcnn_cfg;

typedef struct peanut_rcnn peanut_rcnn_t;

/* tensors: detectron2 state-dict entries (host fp32), names like "backbone.bottom_up.res4.7.conv2.weight",
 * "....conv2.norm.running_var", "backbone.fpn_lateral3.bias", "proposal_generator.rpn_head.conv.weight". */
int peanut_rcnn_create(peanut_rcnn_t** out, const peanut_rcnn_cfg* cfg, const peanut_tensor* tensors, int n_tensors);
void peanut_rcnn_destroy(peanut_rcnn_t* h);
int peanut_rcnn_set_option(peanut_rcnn_t* h, const char* key, long long value);
/* Geometry of a (B,H,W) input: resized (h,w), zero-padded (h,w), the 5 pyramid level sizes p2..p6 as
 * level_hw = {h2,w2,...,h6,w6}, workspace bytes, conv FLOPs per image.  Any output may be NULL. */
int peanut_rcnn_plan(peanut_rcnn_t* h, int B, int H, int W, int resized_hw[2], int padded_hw[2], int level_hw[10],
                     size_t* workspace_bytes, double* flops_per_image);
/* img_bgr: device uint8 [B,H,W,3] (what DefaultPredictor receives, segmentation.py:44-45).  Outputs
 * (device fp32 NHWC, each array has 5 entries for p2..p6, entries or whole arrays may be NULL):
 * pyramid[l] [B,h_l,w_l,fpn_out], objectness[l] [B,h_l,w_l,A], deltas[l] [B,h_l,w_l,4A].
 * The RPN head runs on all five levels as ONE chain (five Winograd input transforms, one grouped position GEMM, five output
 * transforms, one objectness and one anchor-delta GEMM over all levels' rows: 12 launches instead of 25; option rcnn_rpn_fused)
 * whenever objectness[0..4] -- and deltas[0..4] -- are consecutive pieces of ONE buffer in level order (objectness[l + 1] ==
 * objectness[l] + B*h_l*w_l*A), or are not asked for; scattered buffers get the level-by-level launches.  The small levels may
 * then take another Winograd tile size than on their own: results agree to ~1e-5. */
int peanut_rcnn_forward_front(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, float* const* pyramid,
                              float* const* objectness, float* const* deltas, void* stream);

/* Event probe of the front end (bench.py: stage-1 roofline): runs it `reps` times on `stream` with a HIP event after every
 * op, synchronises, and reports per op its name, the kernel family the launch picked (peanut_last_conv_kernel; "wino+..."
 * = Winograd transforms + that GEMM), the mean milliseconds and the direct-form conv FLOPs (0 for non-conv ops).  names /
 * kernels point to storage owned by the handle (valid until the next probe or plan change).  Returns the op count. */
int peanut_rcnn_probe_front(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, int reps, int max_ops,
                            const char** names, const char** kernels, double* ms, double* flops, void* stream);

/* Stage timing of peanut_rcnn_inference / peanut_rcnn_semantic (bench.py: the roofline of stage 1's BACK half -- proposal
 * selection, ROI heads, mask paste: nav/agent/utils/segmentation.py:41-62 around detectron2's GeneralizedRCNN.inference).  While
 * enabled, every call records a HIP event on its stream at the nine stage boundaries: front_end | rpn_selection | roi_align_7x7 |
 * box_head_fc | box_postprocess_nms | host_read_detection_counts | roi_align_14x14 | mask_head_convs | mask_probabilities_paste
 * (a call without detections ends after the sixth).  peanut_rcnn_stage_times reports, for the LAST call, each stage's name, the
 * roof that bounds it ("mfma" | "hbm" | "host"; static strings), its milliseconds and its algorithmic work (FLOPs for "mfma"
 * stages -- the front end's are the direct-form count of its convs (nominal: its Winograd layers execute fewer), the heads'
 * are the executed ones; bytes with every operand read once and every result written once for "hbm" stages); returns the
 * stage count. */
int peanut_rcnn_set_stage_timing(peanut_rcnn_t* h, int on);
int peanut_rcnn_stage_times(peanut_rcnn_t* h, int max_stages, const char** names, const char** bounds, double* ms, double* work);

/* The detector's input transform alone (DefaultPredictor.__call__: ResizeShortestEdge.get_transform(img).apply_image,
 * i.e. PIL.Image.resize(BILINEAR) for uint8 frames -- Pillow's two-pass fixed-point resample, restated bit for bit --
 * then GeneralizedRCNN.preprocess_image: (x - PIXEL_MEAN) / PIXEL_STD, zero padding to the size-divisible canvas).
 * img_bgr: device uint8 [B,H,W,3]; out_nchw: device float [B,3,Hp,Wp] (Hp, Wp from peanut_rcnn_plan).  The front end
 * computes the same pixels internally (in the layout its stem wants); this export exists for bisecting and parity. */
int peanut_rcnn_preprocess(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, float* out_nchw, void* stream);

/* The whole detector: what `DefaultPredictor(img)["instances"]` yields (nav/agent/utils/segmentation.py:45) --
 * GeneralizedRCNN.inference + detector_postprocess as configured by mask_rcnn_R_101_cat9.yaml -- for a batch of frames.
 * Needs a handle created with the roi_heads.* tensors in the state dict (box_head.fc1/fc2, box_predictor.cls_score /
 * bbox_pred, mask_head.mask_fcn1..N / deconv / predictor).  img_bgr: device uint8 [B,H,W,3].  Outputs: n_det_host
 * [B] (HOST ints: detections per image, at most detections_per_image each); the detections of all images back to
 * back, image-major, in decreasing score order inside an image: boxes device [sum n, 4] (x0, y0, x1, y1 in pixels of
 * the ORIGINAL frame), scores device [sum n], classes device int32 [sum n], masks device uint8 [sum n, H, W] (1 where
 * the pasted mask >= mask_threshold; pass NULL to skip the mask head).  The caller sizes the buffers for
 * B * detections_per_image instances.  Synchronises the stream once (to read the detection counts).  With precision
 * FP16X3 the RPN objectness, class scores, box deltas and mask logits are scanned for non-finite values and the call
 * returns PEANUT_ERANGE instead of an image without detections (one more 4-byte read behind the mask head). */
int peanut_rcnn_inference(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, int* n_det_host, float* boxes,
                          float* scores, int32_t* classes, uint8_t* masks, void* stream);
/* SemanticPredMaskRCNN.get_prediction (nav/agent/utils/segmentation.py:41-62) for a batch of frames: the detector as
 * above, then `semantic_input[:, :, cls] += pred_masks[j] * 1.` for every instance with cls in range(n_cats), score >=
 * sem_pred_prob_thr and, for the frame's goal category, score >= goal_thr -- evaluated per output pixel straight from
 * the 28 x 28 mask probabilities (same arithmetic as peanut_paste_masks), so the [n,H,W] instance masks are never
 * materialised.  semantic: device float [B,H,W,n_cats+1] (channel n_cats stays 0, as in the reference);
 * goal_cat_host: HOST int32 [B] (-1 = no goal gate) or NULL.  masks may be NULL (the usual case); the other outputs
 * are those of peanut_rcnn_inference. */
int peanut_rcnn_semantic(peanut_rcnn_t* h, const uint8_t* img_bgr, int B, int H, int W, int n_cats, float sem_pred_prob_thr,
                         float goal_thr, const int32_t* goal_cat_host, float* semantic, int* n_det_host, float* boxes,
                         float* scores, int32_t* classes, uint8_t* masks, void* stream);
/* Test / bisect hook: device pointer (and byte capacity) of a stage buffer of the last peanut_rcnn_inference call:
 * "rois" [B*cap,5], "roi_level", "roi_logit", "prop_count" [B], "cls" [B*cap,K+1], "bbox" [B*cap,4K], "det_in" /
 * "det_out" [B,D,4], "det_score", "det_cls", "det_count" [B], "mprobs" [sum n, 2P, 2P], "sel_idx", "sel_score", "nvalid". */
int peanut_rcnn_debug_stage(peanut_rcnn_t* h, const char* name, const void** dev, size_t* bytes);

/* The non-convolution operators of Mask R-CNN inference (detectron2 implements them natively in
 * `detectron2._C` / torchvision: ROIAlign, nms; paste_masks_in_image is a fused resample+threshold).
 * Restated from detectron2 v0.6's published definitions; parity pinned against oracle/rcnn_ref.py only.
 *
 * peanut_roi_align: ROIAlign over up to 4 NHWC pyramid levels in one launch.  feats[l] device [B,h_l,w_l,C],
 * feat_hw = {h_0,w_0,...}, scales[l] = 1/stride_l; rois device [N,5] = (batch, x0, y0, x1, y1) in input pixels,
 * levels device int32 [N]; sampling_ratio 0 = adaptive ceil(roi/pooled); aligned = ROIAlignV2's half-pixel
 * shift; out device [N,pooled,pooled,C]. */
int peanut_roi_align(const float* const* feats, const int* feat_hw, const float* scales, int n_levels, int C,
                     const float* rois, const int* levels, int n_rois, int pooled, int sampling_ratio, int aligned,
                     float* out, void* stream);
/* Greedy NMS of boxes ALREADY sorted by descending score (device [n,4] x0,y0,x1,y1); categories (device
 * int32 [n], may be NULL) restrict suppression to equal ids (= torchvision batched_nms); keep device
 * uint8 [n] (1 = kept); workspace device, peanut_nms_workspace_bytes(n) bytes. */
size_t peanut_nms_workspace_bytes(int n);
int peanut_nms(const float* boxes_sorted, const int* categories, int n, float iou_threshold, void* workspace,
               unsigned char* keep, void* stream);
/* The same for n_segments independent box lists stored back to back (one per image): segment k is
 * boxes_sorted[seg_offsets_host[k] .. seg_offsets_host[k+1]), sorted by descending score inside the segment.
 * workspace >= sum_k peanut_nms_workspace_bytes(n_k) bytes.  One pair of launches per 64 segments. */
int peanut_nms_segments(const float* boxes_sorted, const int* categories, const int* seg_offsets_host, int n_segments,
                        float iou_threshold, void* workspace, unsigned char* keep, void* stream);
/* paste_masks_in_image + threshold: masks device [n,M,M] probabilities, boxes device [n,4] in output-image
 * pixels -> out device uint8 [n,H,W] (1 where the bilinearly resampled mask >= threshold). */
int peanut_paste_masks(const float* masks, const float* boxes, int n, int M, int H, int W, float threshold,
                       unsigned char* out, void* stream);

/* Observation formatting, Agent_Helper._preprocess_obs/_preprocess_depth
 * (nav/agent/agent_helper.py:175-217): per-column invalid-depth fill, >0.99 -> far, metres -> cm
 * (f32(min_d*100.0) + (d*f32(max_d-min_d))*100 -- the two constants are formed in double like the
 * reference's Python floats, min_d / max_d are therefore doubles), then rows/cols ds//2::ds of RGB (the reference's PIL
 * NEAREST resize picks the same pixels), depth and semantics.  rgb [H,W,3] uint8, depth [H,W] fp32 in
 * [0,1] (0 = invalid), sem [H,W,ncat] fp32 -> obs [3+1+ncat, H/ds, W/ds] fp32. */
int peanut_preprocess_obs(const uint8_t* rgb, const float* depth, const float* sem, int H, int W, int ncat, int ds,
                          double min_d, double max_d, float* obs, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stage 1 -- per-instance mask accumulation of SemanticPredMaskRCNN.get_prediction
 * (nav/agent/utils/segmentation.py:47-60): for every detected instance j whose class is in
 * range(n_cats) and whose score passes sem_pred_prob_thr (and goal_thr when class == goal_cat),
 * out[:, :, class] += mask_j.  masks [n,H,W] uint8/bool, classes [n] int32, scores [n] fp32,
 * out [H,W,n_cats+1] fp32 (zeroed by the call; channel n_cats stays zero).  The detector that
 * produces masks/classes/scores is peanut_rcnn_* + peanut_roi_align / peanut_nms / peanut_paste_masks above
 * (or detectron2 itself: the accumulation only needs the three tensors).
 * ---------------------------------------------------------------------------------------- */
int peanut_seg_accumulate(const uint8_t* masks, const int32_t* classes, const float* scores, int n, int H, int W,
                          int n_cats, float sem_pred_prob_thr, float goal_thr, int goal_cat, float* out,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * Operator-level export: one fused conv (+BN scale/shift, +residual, +ReLU) on NHWC fp32.
 * Mirrors mmcv ConvModule / build_conv_layer+build_norm_layer call sites
 * (prediction/mmseg/models/backbones/resnet.py:164-209, decode_heads/psp_head.py:39-46,86-93).
 * ---------------------------------------------------------------------------------------- */
typedef struct peanut_conv peanut_conv_t;
/* w_oihw_host [cout][cin][kh][kw]; scale/shift host [cout] (NULL -> 1 / 0).  cin_pad = channel
 * count of the NHWC input buffer (multiple of 16, >= cin; extra channels must be zero-weighted,
 * which the packer guarantees).  precision = PEANUT_PREC_*: in the emulated modes a pointwise layer with >= 64 output
 * channels and the position GEMMs of a Winograd layer run on csrc/gemm_rs.hip, every other conv with a multiple of 16 input
 * channels on csrc/conv_rs.hip;
 * peanut_conv_precision() returns the PEANUT_PREC_* mode the layer actually runs in (no silent change of arithmetic).  conv_algo = PEANUT_ALGO_* (AUTO: stride-1 3x3 layers with
 * >= 128 input channels run as Winograd F(4x4,3x3), scratch allocated on first use per shape). */
int peanut_conv_create(peanut_conv_t** out, const float* w_oihw_host, const float* scale_host,
                       const float* shift_host, int cout, int cin, int cin_pad, int kh, int kw, int stride,
                       int pad, int dil, int relu, int precision, int conv_algo);
void peanut_conv_destroy(peanut_conv_t* c);
int peanut_conv_precision(peanut_conv_t* c);
int peanut_conv_set_option(peanut_conv_t* c, const char* key, long long value);
/* x_dev [B,H,W,cin_pad] (or split x_dev [..,c1] ++ x2_dev [..,cin_pad-c1] when x2_dev != NULL),
 * res_dev optional [B,Ho,Wo,cout], y_dev [B,Ho,Wo,cout]. */
int peanut_conv_forward(peanut_conv_t* c, const float* x_dev, const float* x2_dev, int c1, const float* res_dev,
                        float* y_dev, int B, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * Long-term goal selection: Agent_State.update_global_goal
 * (nav/agent/agent_state.py:376-415) and the geodesic distance transform behind it and behind
 * FMMPlanner.set_goal / set_multi_goal (nav/agent/utils/fmm_planner.py:55-75).  The reference calls scikit-fmm's
 * heap-ordered fast marching on the host (`skfmm.distance`, second order); here the same second-order upwind
 * discretisation is solved by tile-wise relaxation in stages that provably end (csrc/goal.hip; <= 0.14 cell from the
 * heap-ordered march on maze maps, gate 0.5).  scikit-fmm is an absent
 * third-party dependency: parity is pinned against the restatement in oracle/fmm_ref.c only (PARITY UNPINNED).
 * Distances are doubles, in cells.
 * ---------------------------------------------------------------------------------------- */
typedef struct peanut_goal peanut_goal_t;
/* full_h x full_w = the full map (960 x 960); col_rad = args.col_rad, the radius of the dilation disk
 * (skimage.morphology.disk(col_rad), agent_state.py:85).  Allocates ~35 B per cell.  Synchronous. */
int peanut_goal_create(peanut_goal_t** out, int full_h, int full_w, int col_rad);
void peanut_goal_destroy(peanut_goal_t* g);
/* Agent_State.reset (:94-105): forget the last distance weights (`self.dd_wt = None`). */
int peanut_goal_reset(peanut_goal_t* g);
/* relaxation rounds / second-order ordering passes the last solve took (diagnostics) */
int peanut_goal_rounds(peanut_goal_t* g);
int peanut_goal_passes(peanut_goal_t* g);
/* 1 when the last solve's ordering passes reached their fixed point (a pass that changed nothing), 0 when they stopped at
 * the pass cap (6) with the last pass still changing tiles: the field is then the last iterate, not the fixed point. */
int peanut_goal_converged(peanut_goal_t* g);
/* agent_state.py:382-386: trav = ~binary_dilation(rint(full_map[0]), disk(col_rad)); trav[collision_map == 1] = 0;
 * trav[visited_vis == 1] = 1.  full_obstacle device fp32 [H,W]; collision_map / visited_vis device uint8 [H,W] or
 * NULL; trav_out device uint8 [H,W] (NULL: kept inside the handle). */
int peanut_goal_traversible(peanut_goal_t* g, const float* full_obstacle, const uint8_t* collision_map,
                            const uint8_t* visited_vis, uint8_t* trav_out, void* stream);
/* FMMPlanner.set_goal (goal_mask NULL, one goal cell) / set_multi_goal (goal_mask device uint8 [H,W], 1 = goal;
 * pass goal_r = -1): traversible device uint8 [H,W] (0 = masked; goal cells are unmasked like
 * `traversible_ma[goal] = 0` does).  dist_out device double [H,W]; cells that are masked or never reached get
 * +inf (fill_mode 0) or max(reached) + 1 (fill_mode 1 = `ma.filled(dd, np.max(dd) + 1)`, fmm_planner.py:66,74).
 * Synchronises the stream (the solver polls a convergence counter). */
int peanut_fmm_distance(peanut_goal_t* g, const uint8_t* traversible, const uint8_t* goal_mask, int goal_r, int goal_c,
                        int fill_mode, double* dist_out, void* stream);
/* Optional first half of peanut_goal_select, for callers that produce target_pred AFTER the map is final -- in the agent
 * Agent_State.update_state runs update_prediction and then update_global_goal (nav/agent/agent_state.py:240-245), and the geodesic
 * field needs the map, not the prediction.  Call it when full_obstacle, collision_map and visited_vis are complete on `stream`
 * (same arguments as the select that follows): the traversible map, the solver's initialisation and the first batch of relaxation
 * rounds are enqueued on a stream of the handle behind that point of `stream`, without synchronising, so that they run beside
 * whatever the caller enqueues on `stream` next (the prediction forward).  The peanut_goal_select that follows with the same
 * inputs continues there and makes `stream` wait for the field before the weights are formed: results are those of a select
 * alone.  The caller must not write the three map inputs in between.  A select with other inputs, peanut_fmm_distance or
 * peanut_goal_reset lets the begun work run out and ignores it. */
int peanut_goal_select_begin(peanut_goal_t* g, const float* full_obstacle, const uint8_t* collision_map,
                             const uint8_t* visited_vis, const int lmb[4], int loc_r, int loc_c, void* stream);
/* The whole of update_global_goal: traversible map, geodesic field from the agent's cell
 * (clip(loc + lmb[0/2], 0, full - 1)), weights exp(-dd / (dist_weight_temperature / map_resolution)) over the local
 * window lmb = {gx1, gx2, gy1, gy2} with the "sum < 10: keep the last weights" rule, value = target_pred * weights
 * (temperature -1: target_pred alone; 0: frontier mode, exp(-dd'/100) with dd' = inf below 60) and its
 * first-occurrence argmax.  target_pred device fp32 [gx2-gx1, gy2-gy1].  Outputs (host): goal_rc_out = the argmax
 * cell in local-map coordinates (`np.unravel_index(value.argmax(), value.shape)`); stats_out (optional) =
 * {value max, sum of the fresh weights, 1 if the last weights were kept, relaxation rounds}; dist_out (optional,
 * device double [H,W], +inf = masked / unreachable) and value_out (optional, device double [w,h]) for tests.
 * The "avoid repeating the last goal" bookkeeping (:412-415) is host logic of the caller.  Synchronises. */
int peanut_goal_select(peanut_goal_t* g, const float* full_obstacle, const uint8_t* collision_map,
                       const uint8_t* visited_vis, const int lmb[4], int loc_r, int loc_c, const float* target_pred,
                       double dist_weight_temperature, int map_resolution, int goal_rc_out[2], double stats_out[4],
                       double* dist_out, double* value_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU: collation of the predicted maps for logging.  One process per GPU, maps /
 * episodes sharded with no data-path collective -- the reference shards by hand with --start_ep/--end_ep/
 * --sem_gpu_id (nav/arguments.py:15-20, nav/collect.py:37-50) and never communicates; this all-gather is the one
 * collective BASELINE.json's north_star adds.  RCCL over xGMI, bound at run time (the librccl.so already loaded
 * in the process, e.g. torch's, else ROCm's; PEANUT_RCCL_LIB overrides).
 * ---------------------------------------------------------------------------------------- */
#define PEANUT_COMM_ID_BYTES 128
typedef struct peanut_comm peanut_comm_t;
/* rank 0: create the id (ncclGetUniqueId), then hand the 128 bytes to every rank (host side: file, MPI, TCP store). */
int peanut_comm_unique_id(unsigned char id[PEANUT_COMM_ID_BYTES]);
/* every rank, on its own GPU (the current HIP device): ncclCommInitRank -- a collective call.  n_ranks == 1 needs
 * neither an id nor RCCL. */
int peanut_comm_create(peanut_comm_t** out, int n_ranks, int rank, const unsigned char id[PEANUT_COMM_ID_BYTES]);
void peanut_comm_destroy(peanut_comm_t* c);
int peanut_comm_info(peanut_comm_t* c, int* n_ranks, int* rank);
/* which RCCL library got bound ("" when none could be) */
const char* peanut_comm_backend(void);
/* local: this rank's [B_local,K,S,S] fp32 maps (device, `count` floats); all: [n_ranks * count] floats (device),
 * rank-major.  One ncclAllGather on `stream`, no staging copy, no synchronisation. */
int peanut_allgather_maps(peanut_comm_t* c, const float* local, float* all, size_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PEANUT_HIP_H_ */
