// Observation formatting for the map projection: Agent_Helper._preprocess_obs / _preprocess_depth
// (nav/agent/agent_helper.py:175-217).  The reference loops over the 640 image columns in Python on
// the host; here one workgroup per OUTPUT column does the column statistics (share of invalid pixels,
// column maximum), the invalid/too-far fills, the metres->cm conversion (same fp32 operation order as
// NumPy: 50 + (d*4.5)*100) and the ds-fold subsampling (rows/cols ds//2::ds) of depth, RGB and the
// semantic channels in a single launch, writing the [3+1+ncat, h, w] observation directly.
#include "../../include/peanut_hip.h"
#include "common.h"

#pragma clang fp contract(off)

namespace peanut {

__global__ __launch_bounds__(128) void preprocess_obs_kernel(const uint8_t* __restrict__ rgb,
                                                             const float* __restrict__ depth,
                                                             const float* __restrict__ sem, int H, int W, int ncat,
                                                             int ds, float base_cm, float span, float* __restrict__ obs) {
  __shared__ int s_zero;
  __shared__ unsigned s_max;   // depth >= 0, so the raw bit pattern orders like the float
  const int j = blockIdx.x;                 // output column
  const int i = ds / 2 + j * ds;            // source column
  const int h = H / ds, w = W / ds;
  if (threadIdx.x == 0) { s_zero = 0; s_max = 0u; }
  __syncthreads();
  int zeros = 0;
  float mx = 0.f;
  for (int r = threadIdx.x; r < H; r += blockDim.x) {
    const float d = depth[(size_t)r * W + i];
    zeros += d == 0.f;
    mx = fmaxf(mx, d);
  }
  atomicAdd(&s_zero, zeros);
  atomicMax(&s_max, __float_as_uint(fmaxf(mx, 0.f)));
  __syncthreads();
  // np.mean(invalid) > 0.9  ->  fill invalid with the column max, else with 100.0  (agent_helper.py:200-206)
  const bool mostly_invalid = ((double)s_zero / (double)H) > 0.9;
  const float fill = mostly_invalid ? __uint_as_float(s_max) : 100.0f;
  for (int k = threadIdx.x; k < h; k += blockDim.x) {
    const int r = ds / 2 + k * ds;
    float d = depth[(size_t)r * W + i];
    if (d == 0.f) d = fill;
    if (d > 0.99f) d = 0.f;                 // too far (:209-210)
    if (d == 0.f) d = 100.0f;               // (:212-213)
    const float cm = base_cm + ((d * span) * 100.0f);   // (:216) base_cm = f32(min_d*100.0), span = f32(max_d-min_d)
    const size_t o = (size_t)k * w + j;
    const size_t hw = (size_t)h * w;
    const uint8_t* px = rgb + ((size_t)r * W + i) * 3;
    obs[0 * hw + o] = (float)px[0];
    obs[1 * hw + o] = (float)px[1];
    obs[2 * hw + o] = (float)px[2];
    obs[3 * hw + o] = cm;
    const float* sp = sem + ((size_t)r * W + i) * ncat;
    for (int c = 0; c < ncat; ++c) obs[(4 + c) * hw + o] = sp[c];
  }
}

}  // namespace peanut

extern "C" int peanut_preprocess_obs(const uint8_t* rgb, const float* depth, const float* sem, int H, int W, int ncat,
                                     int ds, double min_d, double max_d, float* obs, void* stream) {
  using namespace peanut;
  if (!rgb || !depth || !sem || !obs) return fail(PEANUT_EINVAL, "peanut_preprocess_obs: null argument");
  if (ds < 1 || H % ds || W % ds || ncat < 1) return fail(PEANUT_EINVAL, "peanut_preprocess_obs: bad geometry");
  // the reference evaluates `min_d * 100.0` and `(max_d - min_d)` in Python doubles before NumPy narrows them to the
  // float32 of the depth array (agent_helper.py:216); do the same on the host
  const float base_cm = (float)(min_d * 100.0), span = (float)(max_d - min_d);
  hipLaunchKernelGGL(preprocess_obs_kernel, dim3(W / ds), dim3(128), 0, (hipStream_t)stream, rgb, depth, sem, H, W, ncat,
                     ds, base_cm, span, obs);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(PEANUT_EHIP, std::string("preprocess_obs: ") + hipGetErrorString(e));
}
