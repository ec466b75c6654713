// Stage 1, S-2: per-instance mask accumulation of SemanticPredMaskRCNN.get_prediction
// (nav/agent/utils/segmentation.py:47-60).  The reference loops over <= 100 instances on the host,
// launching one `semantic_input[:, :, idx] += mask` per kept instance after a D2H sync of the classes;
// here the score/class gating runs on the device and ONE launch adds every kept mask in instance order
// (so overlapping instances of a class sum to 2, 3, ... exactly like the reference).
#include "../../include/peanut_hip.h"
#include "common.h"

namespace peanut {

__global__ __launch_bounds__(256) void seg_accumulate_kernel(const uint8_t* __restrict__ masks,
                                                             const int32_t* __restrict__ classes,
                                                             const float* __restrict__ scores, int n, int HW,
                                                             int n_cats, float thr, float goal_thr, int goal_cat,
                                                             float* __restrict__ out) {
  const int ch = n_cats + 1;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < HW;
       p += (long long)gridDim.x * blockDim.x) {
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
    for (int j = 0; j < n; ++j) {
      const int cls = classes[j];                       // wave-uniform -> scalar loads
      if (cls < 0 || cls >= n_cats) continue;           // `if class_idx in range(self.n_cats)`
      const float sc = scores[j];
      if (sc < thr) continue;                           // segmentation.py:53-54
      if (cls == goal_cat && sc < goal_thr) continue;   // :55-57
      const float m = masks[(size_t)j * HW + p] ? 1.f : 0.f;   // pred_masks[j] * 1.
#pragma unroll
      for (int c = 0; c < 32; ++c)
        if (c == cls) acc[c] += m;
    }
    for (int c = 0; c < ch; ++c) out[(size_t)p * ch + c] = c < 32 ? acc[c] : 0.f;
  }
}

}  // namespace peanut

extern "C" int peanut_seg_accumulate(const uint8_t* masks, const int32_t* classes, const float* scores, int n, int H,
                                     int W, int n_cats, float thr, float goal_thr, int goal_cat, float* out,
                                     void* stream) {
  using namespace peanut;
  if (!out || H < 1 || W < 1 || n < 0 || (n > 0 && (!masks || !classes || !scores)))
    return fail(PEANUT_EINVAL, "peanut_seg_accumulate: bad argument");
  if (n_cats < 1 || n_cats > 31) return fail(PEANUT_EINVAL, "peanut_seg_accumulate: 1 <= n_cats <= 31 required");
  const int HW = H * W;
  int grid = (HW + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(seg_accumulate_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, masks, classes, scores, n, HW,
                     n_cats, thr, goal_thr, goal_cat, out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("seg_accumulate: ") + hipGetErrorString(e));
}
