// Shared pieces of the implicit-GEMM conv kernels (fp32-MFMA and split-precision variants).
#pragma once
#include "common.h"

namespace peanut {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: no struct memcpy, stays in VGPRs

struct ConvKParams {
  const float* x;
  const float* x2;
  const float* w;
  const float* scale;
  const float* shift;
  const float* res;
  float* y;
  int H, W, c1, c2, Ho, Wo, cout;
  int kw, ntaps, stride, pad, dil, relu;
  int M, nkt, ntiles, HoWo;
};

template <int I>
struct IC { static constexpr int value = I; };
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IC<I>{});
    static_for<N, I + 1>(f);
  }
}

struct KIter {
  int tap, ky, kx, cbase;
  const float* wtile;
};


int launch_conv_split(const ConvKParams& p, int bn_tile, int fp16, hipStream_t stream);

// XCD-aware tile assignment (bijective for any grid size; guide T1): each XCD (private L2) owns a
// contiguous run of tiles, n-tile fastest.
__device__ __forceinline__ void xcd_tile(int ntiles, int* mt, int* nt) {
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
  const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  *mt = L / ntiles;
  *nt = L - *mt * ntiles;
}

}  // namespace peanut
