// A torch-free host of the WHOLE detector (include/peanut_hip.h: peanut_rcnn_create + peanut_rcnn_inference): what a
// C/C++ maintainer would write in place of detectron2's DefaultPredictor call (nav/agent/utils/segmentation.py:30-45).
// Driven by tests/test_c_host_gpu.py, which checks the outputs against the Python path bit for bit.
//
//   rcnn_host weights.bin image.bin out_prefix B H W depth pre_topk post_topk dets score_thresh
//
// weights.bin: int32 n; then n x { int32 name_len; char name[]; int32 ndim; int64 shape[4]; float data[prod] }
// image.bin:   uint8 [B,H,W,3] BGR.  Writes out_prefix.{counts,boxes,scores,classes,masks}.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "peanut_hip.h"

#define CHECK_HIP(e)                                                        \
  do {                                                                      \
    hipError_t _e = (e);                                                    \
    if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 2; } \
  } while (0)
#define CHECK_PEANUT(e)                                                     \
  do {                                                                      \
    int _rc = (e);                                                          \
    if (_rc != 0) { fprintf(stderr, "%s -> %d: %s\n", #e, _rc, peanut_last_error()); return 3; } \
  } while (0)

static bool dump(const std::string& path, const void* p, size_t bytes) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = bytes == 0 || fwrite(p, 1, bytes, f) == bytes;
  fclose(f);
  return ok;
}

int main(int argc, char** argv) {
  if (argc != 12) { fprintf(stderr, "usage: rcnn_host weights.bin image.bin out_prefix B H W depth pre_topk post_topk dets score_thresh\n"); return 1; }
  const int B = atoi(argv[4]), H = atoi(argv[5]), W = atoi(argv[6]);
  if (std::string(peanut_build_arch()) != "gfx950" || peanut_abi_version() < 5) { fprintf(stderr, "unexpected library\n"); return 1; }

  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int32_t n = 0;
  if (fread(&n, 4, 1, f) != 1) return 1;
  std::vector<std::string> names(n);
  std::vector<std::vector<float>> data(n);
  std::vector<peanut_tensor> tensors(n);
  for (int i = 0; i < n; ++i) {
    int32_t len = 0, ndim = 0;
    int64_t shape[4];
    if (fread(&len, 4, 1, f) != 1) return 1;
    names[i].resize(len);
    if (fread(&names[i][0], 1, len, f) != (size_t)len || fread(&ndim, 4, 1, f) != 1 || fread(shape, 8, 4, f) != 4) return 1;
    size_t count = 1;
    for (int d = 0; d < ndim; ++d) count *= (size_t)shape[d];
    data[i].resize(count);
    if (fread(data[i].data(), 4, count, f) != count) return 1;
    tensors[i].ndim = ndim;
    for (int d = 0; d < 4; ++d) tensors[i].shape[d] = shape[d];
  }
  fclose(f);
  for (int i = 0; i < n; ++i) { tensors[i].name = names[i].c_str(); tensors[i].data = data[i].data(); }

  // mask_rcnn_R_101_cat9.yaml, with the proposal / detection counts of the test
  peanut_rcnn_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.depth = atoi(argv[7]); cfg.stem_out = 64; cfg.res2_out = 256; cfg.stride_in_1x1 = 1; cfg.fpn_out = 256; cfg.num_anchors = 3;
  cfg.min_size = 800; cfg.max_size = 1333; cfg.size_divisibility = 32;
  const float mean[3] = {103.53f, 116.28f, 123.675f};
  for (int i = 0; i < 3; ++i) { cfg.pixel_mean[i] = mean[i]; cfg.pixel_std[i] = 1.0f; }
  cfg.bn_eps = 1e-5f; cfg.precision = PEANUT_PREC_FP32; cfg.conv_algo = PEANUT_ALGO_AUTO;
  const float sizes[5] = {32, 64, 128, 256, 512}, ratios[3] = {0.5f, 1.0f, 2.0f};
  for (int i = 0; i < 5; ++i) cfg.anchor_sizes[i] = sizes[i];
  for (int i = 0; i < 3; ++i) cfg.aspect_ratios[i] = ratios[i];
  cfg.rpn_pre_nms_topk = atoi(argv[8]); cfg.rpn_post_nms_topk = atoi(argv[9]); cfg.rpn_nms_thresh = 0.7f;
  for (int i = 0; i < 4; ++i) cfg.rpn_bbox_weights[i] = 1.0f;
  cfg.num_classes = 9; cfg.box_pooler_resolution = 7; cfg.mask_pooler_resolution = 14; cfg.fc_dim = 1024; cfg.mask_conv_dim = 256;
  cfg.num_mask_convs = 4;
  cfg.roi_bbox_weights[0] = cfg.roi_bbox_weights[1] = 10.0f; cfg.roi_bbox_weights[2] = cfg.roi_bbox_weights[3] = 5.0f;
  cfg.score_thresh_test = (float)atof(argv[11]); cfg.nms_thresh_test = 0.5f; cfg.detections_per_image = atoi(argv[10]); cfg.mask_threshold = 0.5f;

  peanut_rcnn_t* h = nullptr;
  CHECK_PEANUT(peanut_rcnn_create(&h, &cfg, tensors.data(), n));

  std::vector<uint8_t> img((size_t)B * H * W * 3);
  f = fopen(argv[2], "rb");
  if (!f || fread(img.data(), 1, img.size(), f) != img.size()) { fprintf(stderr, "bad image file\n"); return 1; }
  fclose(f);
  const int D = cfg.detections_per_image;
  uint8_t *d_img = nullptr, *d_masks = nullptr;
  float *d_boxes = nullptr, *d_scores = nullptr;
  int32_t* d_cls = nullptr;
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_HIP(hipMalloc((void**)&d_img, img.size()));
  CHECK_HIP(hipMalloc((void**)&d_boxes, (size_t)B * D * 16));
  CHECK_HIP(hipMalloc((void**)&d_scores, (size_t)B * D * 4));
  CHECK_HIP(hipMalloc((void**)&d_cls, (size_t)B * D * 4));
  CHECK_HIP(hipMalloc((void**)&d_masks, (size_t)B * D * H * W));
  CHECK_HIP(hipMemcpyAsync(d_img, img.data(), img.size(), hipMemcpyHostToDevice, stream));
  std::vector<int> counts(B);
  CHECK_PEANUT(peanut_rcnn_inference(h, d_img, B, H, W, counts.data(), d_boxes, d_scores, d_cls, d_masks, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  int total = 0;
  for (int b = 0; b < B; ++b) total += counts[b];
  std::vector<float> boxes((size_t)total * 4), scores(total);
  std::vector<int32_t> cls(total);
  std::vector<uint8_t> masks((size_t)total * H * W);
  if (total) {
    CHECK_HIP(hipMemcpy(boxes.data(), d_boxes, boxes.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(scores.data(), d_scores, scores.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(cls.data(), d_cls, cls.size() * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(masks.data(), d_masks, masks.size(), hipMemcpyDeviceToHost));
  }
  const std::string pre = argv[3];
  if (!dump(pre + ".counts.bin", counts.data(), counts.size() * 4) || !dump(pre + ".boxes.bin", boxes.data(), boxes.size() * 4) ||
      !dump(pre + ".scores.bin", scores.data(), scores.size() * 4) || !dump(pre + ".classes.bin", cls.data(), cls.size() * 4) ||
      !dump(pre + ".masks.bin", masks.data(), masks.size()))
    return 1;
  peanut_rcnn_destroy(h);
  printf("ok %d detections\n", total);
  return 0;
}
