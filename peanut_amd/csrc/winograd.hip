// Winograd F(4x4, 3x3) -- and, further down, F(6x6, 3x3) and F(5x5, 3x3) -- for the stride-1 3x3 convolutions (fp32 transforms around the
// MFMA GEMM kernel).
//
//   Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A        (Lavin & Gray 2016; interpolation points 0, +-3/4, +-3/2, inf)
//
// 36 multiplies per 4x4 output tile and channel pair instead of 144: the contraction over input channels
// becomes 36 independent GEMMs [tiles x Cin] x [Cin x Cout], which run on conv_igemm's MFMA kernel as one
// launch (a 1x1 "conv" over 36*m_pad rows whose weight block is selected by the row group).  The two
// transforms here are HBM-bound streaming kernels: NHWC, one thread per (tile, 4-channel group), every
// access a 16-byte piece of a contiguous channel run.
//
// Dilation d is handled by decomposition: output pixels with equal (y mod d, x mod d) form d*d independent
// undilated problems on sub-sampled grids (padding 1 in sub-grid units = d in pixels).  Tiles are indexed
// t = (((b*d + oy)*d + ox)*th + ty)*tw + tx with a uniform th x tw per sub-grid; tiles or tile parts that
// fall outside the image read zeros and are not written back.
//
// Replaces (reference): the 3x3 conv2 of every stride-1 Bottleneck (prediction/mmseg/models/backbones/
// resnet.py:267-307, dilations 1/2/4) and the PSP bottleneck conv (models/decode_heads/psp_head.py:86-93).
// Numerics: fp32 throughout; on the seeded PSPNet the logits are as far from the fp64 reference as with direct
// convolutions (5e-6, |logit| <= 6.4); contract 1e-3.
#include <stdlib.h>

#include <atomic>

#include "common.h"
#include "conv_common.h"

namespace peanut {

namespace {

// Interpolation points 0, +-3/4, +-3/2, inf instead of Lavin & Gray's 0, +-1, +-2, inf: the transform constants stay
// exact in fp32 (dyadic rationals) and keep the even/odd structure, but the three matrices are better conditioned --
// on the seeded PSPNet the logits' distance to the fp64 reference drops from 1.5e-5 to the direct convolution's own
// 5e-6 (tests/test_pred_gpu.py::test_distance_to_the_fp64_reference).
//   B^T = [81/64 0 -45/16 0 1 0; 0 -27/16 -9/4 3/4 1 0; 0 27/16 -9/4 -3/4 1 0; 0 -27/32 -9/16 3/2 1 0;
//          0 27/32 -9/16 -3/2 1 0; 0 81/64 0 -45/16 0 1]
//   A^T = [1 1 1 1 1 0; 0 3/4 -3/4 3/2 -3/2 0; 0 9/16 9/16 9/4 9/4 0; 0 27/64 -27/64 27/8 -27/8 1]
//   G   = [64/81 0 0; -128/243 -32/81 -8/27; -128/243 32/81 -8/27; 32/243 16/81 8/27; 32/243 -16/81 8/27; 0 0 1]
// t = B^T d for one 6-vector
// a * k + c as ONE fused multiply-add per lane, spelled out.  The transform helpers below used to leave the choice to hipcc's
// contraction (-ffp-contract=fast): in `x * a - y * b` ONE product is fused and the other rounded, and which one turned out to depend
// on the instantiation the helper was inlined into (round 6: the input transform that sums split-K partial tiles came out 4.5e-6
// off on the logits of a 240 x 240 map in its f32x4 form only).  Every product that is not a power of two is now placed explicitly and
// contraction is off inside the helpers, so that every kernel that calls them computes the same bits.
template <typename V>
__device__ __forceinline__ V fmak(V a, float k, V c) { return __builtin_elementwise_fma(a, (V)k, c); }

__device__ __forceinline__ void bt6(const f32x4 d0, const f32x4 d1, const f32x4 d2, const f32x4 d3, const f32x4 d4,
                                    const f32x4 d5, f32x4& t0, f32x4& t1, f32x4& t2, f32x4& t3, f32x4& t4, f32x4& t5) {
#pragma clang fp contract(off)
  const f32x4 a = fmak(d2, -2.25f, d4), b = fmak(d3, 0.75f, -1.6875f * d1);
  const f32x4 c = fmak(d2, -0.5625f, d4), e = fmak(d3, 1.5f, -0.84375f * d1);
  t0 = fmak(d0, 1.265625f, fmak(d2, -2.8125f, d4));
  t1 = a + b;
  t2 = a - b;
  t3 = c + e;
  t4 = c - e;
  t5 = fmak(d1, 1.265625f, fmak(d3, -2.8125f, d5));
}

// y = A^T m for one 6-vector
__device__ __forceinline__ void at6(const f32x4 m0, const f32x4 m1, const f32x4 m2, const f32x4 m3, const f32x4 m4,
                                    const f32x4 m5, f32x4& y0, f32x4& y1, f32x4& y2, f32x4& y3) {
#pragma clang fp contract(off)
  const f32x4 s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
  y0 = m0 + s12 + s34;
  y1 = fmak(d12, 0.75f, 1.5f * d34);
  y2 = fmak(s12, 0.5625f, 2.25f * s34);
  y3 = fmak(d12, 0.421875f, fmak(d34, 3.375f, m5));
}

struct WinoGeom {
  int B, H, W, C;      // C = channels of the tensor this kernel touches (Cin for input, Cout for output)
  int d, th, tw;
  long long n_tiles, m_pad;
};

__device__ __forceinline__ void decode_tile(const WinoGeom& g, long long t, int& b, int& oy, int& ox, int& ty, int& tx) {
  tx = (int)(t % g.tw); t /= g.tw;
  ty = (int)(t % g.th); t /= g.th;
  ox = (int)(t % g.d); t /= g.d;
  oy = (int)(t % g.d);
  b = (int)(t / g.d);
}

// Thread -> (tile, 4-channel group), channel group fastest: a wave = 256 consecutive channels of one tile (1 KiB fp32 runs)
__device__ __forceinline__ bool decode_thread(long long idx, int groups, long long n_tiles, int& cg, long long& tile) {
  cg = (int)(idx % groups);
  tile = idx / groups;
  return tile < n_tiles;
}

// V[(i*6+j)][tile][c] = (B^T d B)[i][j]
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, const WinoGeom g) {
  const int groups = g.C >> 2;
  const long long total = g.n_tiles * groups;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int cg;
    long long tile;
    if (!decode_thread(idx, groups, g.n_tiles, cg, tile)) continue;
    int b, oy, ox, ty, tx;
    decode_tile(g, tile, b, oy, ox, ty, tx);
    const int r0 = oy + g.d * (4 * ty - 1), c0 = ox + g.d * (4 * tx - 1);
    const float* xb = x + (size_t)b * g.H * g.W * g.C + cg * 4;
    f32x4 p[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int r = r0 + i * g.d;
      const bool rok = (unsigned)r < (unsigned)g.H;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int c = c0 + j * g.d;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (rok && (unsigned)c < (unsigned)g.W) v = *reinterpret_cast<const f32x4*>(xb + ((size_t)r * g.W + c) * g.C);
        p[i][j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j)   // columns: p <- B^T p
      bt6(p[0][j], p[1][j], p[2][j], p[3][j], p[4][j], p[5][j], p[0][j], p[1][j], p[2][j], p[3][j], p[4][j], p[5][j]);
    float* vb = V + (size_t)tile * g.C + cg * 4;
    const size_t pos_stride = (size_t)g.m_pad * g.C;
#pragma unroll
    for (int i = 0; i < 6; ++i) {  // rows: (p B)[i][:]
      f32x4 t0, t1, t2, t3, t4, t5;
      bt6(p[i][0], p[i][1], p[i][2], p[i][3], p[i][4], p[i][5], t0, t1, t2, t3, t4, t5);
      float* o = vb + (size_t)(i * 6) * pos_stride;
      // (plain stores: non-temporal ones measured 2.5 % slower here, profiles/r3t)
      *reinterpret_cast<f32x4*>(o) = t0;
      *reinterpret_cast<f32x4*>(o + pos_stride) = t1;
      *reinterpret_cast<f32x4*>(o + 2 * pos_stride) = t2;
      *reinterpret_cast<f32x4*>(o + 3 * pos_stride) = t3;
      *reinterpret_cast<f32x4*>(o + 4 * pos_stride) = t4;
      *reinterpret_cast<f32x4*>(o + 5 * pos_stride) = t5;
    }
  }
}

// y[pixel][n] = relu(scale[n] * (A^T M A)[a][b] + shift[n] + res[pixel][n])
template <bool RES>
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mb, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ res,
                                                          float* __restrict__ y, const WinoGeom g, int relu) {
  const int groups = g.C >> 2;
  const long long total = g.n_tiles * groups;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int ng;
    long long tile;
    if (!decode_thread(idx, groups, g.n_tiles, ng, tile)) continue;
    int b, oy, ox, ty, tx;
    decode_tile(g, tile, b, oy, ox, ty, tx);
    const int r0 = oy + g.d * 4 * ty, c0 = ox + g.d * 4 * tx;
    if (r0 >= g.H || c0 >= g.W) continue;   // tile entirely outside its sub-grid
    const float* mb = Mb + (size_t)tile * g.C + ng * 4;
    const size_t pos_stride = (size_t)g.m_pad * g.C;
    f32x4 q[6][4];   // M A  (rows i, output columns)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float* s = mb + (size_t)(i * 6) * pos_stride;
      // M is read exactly once: non-temporal loads (1.36 -> 1.20 ms per batch-32 forward, profiles/r3t)
      const f32x4 m0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s));
      const f32x4 m1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s + pos_stride));
      const f32x4 m2 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s + 2 * pos_stride));
      const f32x4 m3 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s + 3 * pos_stride));
      const f32x4 m4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s + 4 * pos_stride));
      const f32x4 m5 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(s + 5 * pos_stride));
      at6(m0, m1, m2, m3, m4, m5, q[i][0], q[i][1], q[i][2], q[i][3]);
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ng * 4);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + ng * 4);
    const size_t img = (size_t)b * g.H * g.W;
    // the residual of output column j + 1 is requested before column j is finished (see wino6_output_kernel)
    f32x4 rn[4];
    auto res_column = [&](int j, f32x4 (&dst)[4]) __attribute__((always_inline)) {
      const int c = c0 + j * g.d;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int r = r0 + a * g.d;
        const bool ok = c < g.W && r < g.H;
        const size_t off = ok ? (img + (size_t)r * g.W + c) * g.C + ng * 4 : 0;
        dst[a] = *reinterpret_cast<const f32x4*>(res + off);
      }
    };
    if constexpr (RES) res_column(0, rn);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 rc[4];
      if constexpr (RES) {
#pragma unroll
        for (int a = 0; a < 4; ++a) rc[a] = rn[a];
        if (j + 1 < 4) res_column(j + 1, rn);
      }
      f32x4 o0, o1, o2, o3;
      at6(q[0][j], q[1][j], q[2][j], q[3][j], q[4][j], q[5][j], o0, o1, o2, o3);
      const int c = c0 + j * g.d;
      if (c >= g.W) continue;
      const f32x4 o[4] = {o0, o1, o2, o3};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int r = r0 + a * g.d;
        if (r >= g.H) continue;
        const size_t off = (img + (size_t)r * g.W + c) * g.C + ng * 4;
        f32x4 v = o[a] * sc + sh;
        if constexpr (RES) v += rc[a];
        if (relu) v = relu_keep_nan(v);
        *reinterpret_cast<f32x4*>(y + off) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// F(6x6, 3x3): 8x8 input tiles, 64 positions, 36 outputs per tile -- 1.78 multiplies per output instead of 2.25, and
// V / M are 1.78 x the activations instead of 2.25 x.  Interpolation points 0, +-1/2, +-1, +-2, inf (the constants are
// dyadic: exact in fp32); position order [0, 1/2, -1/2, 1, -1, 2, -2, inf].
//   B^T = [-1 0 21/4 0 -21/4 0 1 0; 0 2 4 -5/2 -5 1/2 1 0; 0 -2 4 5/2 -5 -1/2 1 0; 0 1 1 -17/4 -17/4 1 1 0;
//          0 -1 1 17/4 -17/4 -1 1 0; 0 1/2 1/4 -5/2 -5/4 2 1 0; 0 -1/2 1/4 5/2 -5/4 -2 1 0; 0 -1 0 21/4 0 -21/4 0 1]
//   A^T = [1 1 1 1 1 1 1 0; 0 1/2 -1/2 1 -1 2 -2 0; 0 1/4 1/4 1 1 4 4 0; 0 1/8 -1/8 1 -1 8 -8 0;
//          0 1/16 1/16 1 1 16 16 0; 0 1/32 -1/32 1 -1 32 -32 1]
//   G   = [-1 0 0; 32/45 16/45 8/45; 32/45 -16/45 8/45; -2/9 -2/9 -2/9; -2/9 2/9 -2/9; 1/90 1/45 2/45; 1/90 -1/45 2/45; 0 0 1]
// Operator-level rounding error about 3 x that of the F(4x4) form above with its 0, +-3/4, +-3/2 points and about that of
// F(4x4) with the classic 0, +-1, +-2 points (cin = 512, post-ReLU data: rms 3-5e-6 vs 1.1e-6 vs 2.7e-6 relative).
// One thread per (tile, 2-channel group): an 8x8 patch of 4-channel pieces would need 256 registers.
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void bt8(const f32x2 d0, const f32x2 d1, const f32x2 d2, const f32x2 d3, const f32x2 d4, const f32x2 d5,
                                    const f32x2 d6, const f32x2 d7, f32x2& t0, f32x2& t1, f32x2& t2, f32x2& t3, f32x2& t4,
                                    f32x2& t5, f32x2& t6, f32x2& t7) {
#pragma clang fp contract(off)
  // (products with a power of two are exact: only the 5, 2.5, 4.25, 1.25 and 5.25 terms need placing)
  const f32x2 e1 = 4.f * d2 + fmak(d4, -5.f, d6), o1 = 2.f * d1 + fmak(d3, -2.5f, 0.5f * d5);
  const f32x2 e2 = fmak(d4, -4.25f, d2) + d6, o2 = fmak(d3, -4.25f, d1) + d5;
  const f32x2 e3 = 0.25f * d2 + fmak(d4, -1.25f, d6), o3 = 0.5f * d1 + fmak(d3, -2.5f, 2.f * d5);
  const f32x2 r0 = fmak(d2 - d4, 5.25f, d6 - d0), r7 = fmak(d3 - d5, 5.25f, d7 - d1);
  t0 = r0; t1 = e1 + o1; t2 = e1 - o1; t3 = e2 + o2; t4 = e2 - o2; t5 = e3 + o3; t6 = e3 - o3; t7 = r7;
}

__device__ __forceinline__ void at8(const f32x2 m0, const f32x2 m1, const f32x2 m2, const f32x2 m3, const f32x2 m4, const f32x2 m5,
                                    const f32x2 m6, const f32x2 m7, f32x2& y0, f32x2& y1, f32x2& y2, f32x2& y3, f32x2& y4, f32x2& y5) {
#pragma clang fp contract(off)      // (every constant is a power of two: nothing to fuse that would change a bit)
  const f32x2 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4, s3 = m5 + m6, d3 = m5 - m6;
  y0 = m0 + s1 + s2 + s3;
  y1 = 0.5f * d1 + d2 + 2.f * d3;
  y2 = 0.25f * s1 + s2 + 4.f * s3;
  y3 = 0.125f * d1 + d2 + 8.f * d3;
  y4 = 0.0625f * s1 + s2 + 16.f * s3;
  y5 = 0.03125f * d1 + d2 + 32.f * d3 + m7;
}

// V[(i*8+j)][tile][c] = (B^T d B)[i][j]
__global__ __launch_bounds__(256) void wino6_input_kernel(const float* __restrict__ x, float* __restrict__ V, const WinoGeom g) {
  const int groups = g.C >> 1;
  const long long total = g.n_tiles * groups;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int cg;
    long long tile;
    if (!decode_thread(idx, groups, g.n_tiles, cg, tile)) continue;
    int b, oy, ox, ty, tx;
    decode_tile(g, tile, b, oy, ox, ty, tx);
    const int r0 = oy + g.d * (6 * ty - 1), c0 = ox + g.d * (6 * tx - 1);
    const float* xb = x + (size_t)b * g.H * g.W * g.C + cg * 2;
    f32x2 p[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = r0 + i * g.d;
      const bool rok = (unsigned)r < (unsigned)g.H;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = c0 + j * g.d;
        f32x2 v = {0.f, 0.f};
        if (rok && (unsigned)c < (unsigned)g.W) v = *reinterpret_cast<const f32x2*>(xb + ((size_t)r * g.W + c) * g.C);
        p[i][j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)   // columns: p <- B^T p
      bt8(p[0][j], p[1][j], p[2][j], p[3][j], p[4][j], p[5][j], p[6][j], p[7][j],
          p[0][j], p[1][j], p[2][j], p[3][j], p[4][j], p[5][j], p[6][j], p[7][j]);
    float* vb = V + (size_t)tile * g.C + cg * 2;
    const size_t pos_stride = (size_t)g.m_pad * g.C;
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // rows: (p B)[i][:]
      f32x2 t[8];
      bt8(p[i][0], p[i][1], p[i][2], p[i][3], p[i][4], p[i][5], p[i][6], p[i][7], t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
      float* o = vb + (size_t)(i * 8) * pos_stride;
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x2*>(o + j * pos_stride) = t[j];
    }
  }
}

// y[pixel][n] = relu(scale[n] * (A^T M A)[a][b] + shift[n] + res[pixel][n]),  M: 64 positions
template <bool RES>
__global__ __launch_bounds__(256) void wino6_output_kernel(const float* __restrict__ Mb, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ res,
                                                           float* __restrict__ y, const WinoGeom g, int relu) {
  const int groups = g.C >> 1;
  const long long total = g.n_tiles * groups;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int ng;
    long long tile;
    if (!decode_thread(idx, groups, g.n_tiles, ng, tile)) continue;
    int b, oy, ox, ty, tx;
    decode_tile(g, tile, b, oy, ox, ty, tx);
    const int r0 = oy + g.d * 6 * ty, c0 = ox + g.d * 6 * tx;
    if (r0 >= g.H || c0 >= g.W) continue;   // tile entirely outside its sub-grid
    const float* mb = Mb + (size_t)tile * g.C + ng * 2;
    const size_t pos_stride = (size_t)g.m_pad * g.C;
    f32x2 q[8][6];   // M A  (rows i, output columns)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float* s = mb + (size_t)(i * 8) * pos_stride;
      f32x2 m[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(s + j * pos_stride));
      at8(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], q[i][0], q[i][1], q[i][2], q[i][3], q[i][4], q[i][5]);
    }
    const f32x2 sc = *reinterpret_cast<const f32x2*>(scale + ng * 2);
    const f32x2 sh = *reinterpret_cast<const f32x2*>(shift + ng * 2);
    const size_t img = (size_t)b * g.H * g.W;
    // The residual of output column j + 1 is requested before column j is finished (round 4): with the loads sitting right
    // before their use, behind the runtime `res` test, every column waited out a memory round trip -- the PSP bottleneck's
    // transform (the only one with a residual: the folded pyramid term) ran at 3.8 TB/s against 6.0 for the same shape without.
    // Out-of-range elements read the tensor's first element (a valid address; the value is never used).
    f32x2 rn[6];
    auto res_column = [&](int j, f32x2 (&dst)[6]) __attribute__((always_inline)) {
      const int c = c0 + j * g.d;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const int r = r0 + a * g.d;
        const bool ok = c < g.W && r < g.H;
        const size_t off = ok ? (img + (size_t)r * g.W + c) * g.C + ng * 2 : 0;
        dst[a] = *reinterpret_cast<const f32x2*>(res + off);
      }
    };
    if constexpr (RES) res_column(0, rn);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      f32x2 rc[6];
      if constexpr (RES) {
#pragma unroll
        for (int a = 0; a < 6; ++a) rc[a] = rn[a];
        if (j + 1 < 6) res_column(j + 1, rn);
      }
      f32x2 o[6];
      at8(q[0][j], q[1][j], q[2][j], q[3][j], q[4][j], q[5][j], q[6][j], q[7][j], o[0], o[1], o[2], o[3], o[4], o[5]);
      const int c = c0 + j * g.d;
      if (c >= g.W) continue;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const int r = r0 + a * g.d;
        if (r >= g.H) continue;
        const size_t off = (img + (size_t)r * g.W + c) * g.C + ng * 2;
        f32x2 v = o[a] * sc + sh;
        if constexpr (RES) v += rc[a];
        if (relu) { v.x = relu_keep_nan(v.x); v.y = relu_keep_nan(v.y); }
        *reinterpret_cast<f32x2*>(y + off) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// F(5x5, 3x3): 7x7 input tiles, 49 positions, 25 outputs per tile -- 1.96 multiplies per output.  Between the other two
// in cost and in rounding error (operator level, cin = 512, post-ReLU data: rms 4.9e-6 relative against 1.8e-6 for the
// F(4x4) form and 6.4e-6 for F(6x6) in the same fp32 simulation); what it is FOR is divisibility: with dilation 4 a
// 60 x 60 map's sub-grids are 15 x 15 = 3 x 3 tiles of 5 with no padding at all -- 9 x 49 = 441 position-tiles against
// 16 x 36 = 576 (F(4x4), 16 of 15 rows used) or 9 x 64 = 576 (F(6x6), 18 of 15).
// Seven points cannot all be paired: 0, +-1/2, +-1, 3, inf (position order [0, 1/2, -1/2, 1, -1, 3, inf]); of the
// dyadic / small-integer choices for the unpaired point (+-1/4 .. +-3 with pairs +-1/2,+-1 / +-5/8,+-5/4 / +-3/4,+-3/2 /
// +-1,+-2) this one had the smallest rms error; all constants exact in fp32.
//   B^T = [-3/4 1/4 15/4 -5/4 -3 1 0; 0 3/2 5/2 -5/2 -5/2 1 0; 0 -3/2 7/2 1/2 -7/2 1 0; 0 3/4 1/2 -13/4 -2 1 0;
//          0 -3/4 1 11/4 -4 1 0; 0 1/4 0 -5/4 0 1 0; 0 -3/4 1/4 15/4 -5/4 -3 1]
//   A^T = [1 1 1 1 1 1 0; 0 1/2 -1/2 1 -1 3 0; 0 1/4 1/4 1 1 9 0; 0 1/8 -1/8 1 -1 27 0; 0 1/16 1/16 1 1 81 1]
//   G   = [-4/3 0 0; 16/15 8/15 4/15; 16/21 -8/21 4/21; -1/3 -1/3 -1/3; -1/6 1/6 -1/6; 1/210 1/70 3/70; 0 0 1]
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bt7(const f32x2 d0, const f32x2 d1, const f32x2 d2, const f32x2 d3, const f32x2 d4, const f32x2 d5,
                                    const f32x2 d6, f32x2& t0, f32x2& t1, f32x2& t2, f32x2& t3, f32x2& t4, f32x2& t5, f32x2& t6) {
  const f32x2 r0 = d5 - 0.75f * d0 + 0.25f * d1 + 3.75f * d2 - 1.25f * d3 - 3.f * d4;
  const f32x2 r1 = d5 + 1.5f * d1 + 2.5f * (d2 - d3 - d4);
  const f32x2 r2 = d5 - 1.5f * d1 + 3.5f * (d2 - d4) + 0.5f * d3;
  const f32x2 r3 = d5 + 0.75f * d1 + 0.5f * d2 - 3.25f * d3 - 2.f * d4;
  const f32x2 r4 = d5 - 0.75f * d1 + d2 + 2.75f * d3 - 4.f * d4;
  const f32x2 r5 = d5 + 0.25f * d1 - 1.25f * d3;
  const f32x2 r6 = d6 - 0.75f * d1 + 0.25f * d2 + 3.75f * d3 - 1.25f * d4 - 3.f * d5;
  t0 = r0; t1 = r1; t2 = r2; t3 = r3; t4 = r4; t5 = r5; t6 = r6;
}

__device__ __forceinline__ void at7(const f32x2 m0, const f32x2 m1, const f32x2 m2, const f32x2 m3, const f32x2 m4, const f32x2 m5,
                                    const f32x2 m6, f32x2& y0, f32x2& y1, f32x2& y2, f32x2& y3, f32x2& y4) {
  const f32x2 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
  y0 = m0 + s1 + s2 + m5;
  y1 = 0.5f * d1 + d2 + 3.f * m5;
  y2 = 0.25f * s1 + s2 + 9.f * m5;
  y3 = 0.125f * d1 + d2 + 27.f * m5;
  y4 = 0.0625f * s1 + s2 + 81.f * m5 + m6;
}

// V[(i*7+j)][tile][c] = (B^T d B)[i][j]
__global__ __launch_bounds__(256) void wino5_input_kernel(const float* __restrict__ x, float* __restrict__ V, const WinoGeom g) {
  const int groups = g.C >> 1;
  const long long total = g.n_tiles * groups;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int cg;
    long long tile;
    if (!decode_thread(idx, groups, g.n_tiles, cg, tile)) continue;
    int b, oy, ox, ty, tx;
    decode_tile(g, tile, b, oy, ox, ty, tx);
    const int r0 = oy + g.d * (5 * ty - 1), c0 = ox + g.d * (5 * tx - 1);
    const float* xb = x + (size_t)b * g.H * g.W * g.C + cg * 2;
    f32x2 p[7][7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int r = r0 + i * g.d;
      const bool rok = (unsigned)r < (unsigned)g.H;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int c = c0 + j * g.d;
        f32x2 v = {0.f, 0.f};
        if (rok && (unsigned)c < (unsigned)g.W) v = *reinterpret_cast<const f32x2*>(xb + ((size_t)r * g.W + c) * g.C);
        p[i][j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < 7; ++j)   // columns: p <- B^T p
      bt7(p[0][j], p[1][j], p[2][j], p[3][j], p[4][j], p[5][j], p[6][j],
          p[0][j], p[1][j], p[2][j], p[3][j], p[4][j], p[5][j], p[6][j]);
    float* vb = V + (size_t)tile * g.C + cg * 2;
    const size_t pos_stride = (size_t)g.m_pad * g.C;
#pragma unroll
    for (int i = 0; i < 7; ++i) {  // rows: (p B)[i][:]
      f32x2 t[7];
      bt7(p[i][0], p[i][1], p[i][2], p[i][3], p[i][4], p[i][5], p[i][6], t[0], t[1], t[2], t[3], t[4], t[5], t[6]);
      float* o = vb + (size_t)(i * 7) * pos_stride;
#pragma unroll
      for (int j = 0; j < 7; ++j) *reinterpret_cast<f32x2*>(o + j * pos_stride) = t[j];
    }
  }
}

// y[pixel][n] = relu(scale[n] * (A^T M A)[a][b] + shift[n] + res[pixel][n]),  M: 49 positions
__global__ __launch_bounds__(256) void wino5_output_kernel(const float* __restrict__ Mb, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ res,
                                                           float* __restrict__ y, const WinoGeom g, int relu) {
  const int groups = g.C >> 1;
  const long long total = g.n_tiles * groups;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int ng;
    long long tile;
    if (!decode_thread(idx, groups, g.n_tiles, ng, tile)) continue;
    int b, oy, ox, ty, tx;
    decode_tile(g, tile, b, oy, ox, ty, tx);
    const int r0 = oy + g.d * 5 * ty, c0 = ox + g.d * 5 * tx;
    if (r0 >= g.H || c0 >= g.W) continue;   // tile entirely outside its sub-grid
    const float* mb = Mb + (size_t)tile * g.C + ng * 2;
    const size_t pos_stride = (size_t)g.m_pad * g.C;
    f32x2 q[7][5];   // M A  (rows i, output columns)
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const float* s = mb + (size_t)(i * 7) * pos_stride;
      f32x2 m[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) m[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(s + j * pos_stride));
      at7(m[0], m[1], m[2], m[3], m[4], m[5], m[6], q[i][0], q[i][1], q[i][2], q[i][3], q[i][4]);
    }
    const f32x2 sc = *reinterpret_cast<const f32x2*>(scale + ng * 2);
    const f32x2 sh = *reinterpret_cast<const f32x2*>(shift + ng * 2);
    const size_t img = (size_t)b * g.H * g.W;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      f32x2 o[5];
      at7(q[0][j], q[1][j], q[2][j], q[3][j], q[4][j], q[5][j], q[6][j], o[0], o[1], o[2], o[3], o[4]);
      const int c = c0 + j * g.d;
      if (c >= g.W) continue;
#pragma unroll
      for (int a = 0; a < 5; ++a) {
        const int r = r0 + a * g.d;
        if (r >= g.H) continue;
        const size_t off = (img + (size_t)r * g.W + c) * g.C + ng * 2;
        f32x2 v = o[a] * sc + sh;
        if (res) v += *reinterpret_cast<const f32x2*>(res + off);
        if (relu) { v.x = relu_keep_nan(v.x); v.y = relu_keep_nan(v.y); }
        *reinterpret_cast<f32x2*>(y + off) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small-problem variants (round 6).  The kernels above give one thread a whole tile of a channel group: 64 (36) loads, the
// whole two-sided transform, 64 (36) stores -- a serial chain of ~8-10 us whatever the size, and a batch-1 layer has few such
// threads (one detector frame's res4 conv2: 108 tiles x 128 channel pairs = 54 workgroups on 256 CUs).  One detector frame
// runs 86 transforms, one 720 x 720 map 36: 0.84 ms of a 5.47 ms frame, 0.45 ms of a 3.4 ms map (profiles/r8z).
// Here a WORKGROUP takes one tile of 64 (128) channels and a thread one LINE of it: stage 1 the first-side transform of one
// column (input) / one row of positions (output), through LDS, stage 2 the second-side transform of one row / one output
// column.  8 (6) times the threads, an eighth (sixth) of the chain each.  The same bt8 / at8 / bt6 / at6 on the same
// operands in the same order: bit-identical to the kernels above (tests/test_conv_gpu.py holds the two to torch.equal).
// ---------------------------------------------------------------------------------------------------------------
// What a deferred producer left behind (common.h: DeferredSplit), as a kernel argument.  pixel_value() is the reduce kernel's
// expression (conv_common.h: conv_splitk_reduce_kernel) for the VEC channels starting at channel n of pixel row m: partial tiles
// summed in part order, times scale * alpha, plus shift, ReLU.
struct DeferredSrc {
  const float* partial;
  const float* scale;
  const float* shift;
  float alpha;
  int split_p, ntiles, relu;
};
// The NPIX pixels of one transform line (rows m[i], the same VEC channels starting at n; ok[i] = inside the image, others come out 0).
// The loop over the parts is the OUTER one: every pass issues NPIX independent loads (a loop over the parts per pixel had every load
// wait for the one before it -- profiles/r9d, r9h); SP = 2 / 4: fully unrolled, all loads leave together.  Per pixel the parts are
// added in part order, as the reduce kernel does.
template <typename VT, int SP, int NPIX>
__device__ __forceinline__ void deferred_line(const DeferredSrc& df, const long long (&m)[NPIX], const bool (&ok)[NPIX], int n, VT (&v)[NPIX]) {
  const int parts = SP ? SP : df.split_p;
  const int nt = n >> 7, col = n & 127;
  const float* base[NPIX];
#pragma unroll
  for (int i = 0; i < NPIX; ++i) {
    const long long mm = ok[i] ? m[i] : 0;                    // (a pixel outside the image reads a valid address; its value is dropped)
    base[i] = df.partial + ((size_t)((int)(mm >> 7) * df.ntiles + nt) * parts) * (128 * 128) + (int)(mm & 127) * 128 + col;
  }
#pragma unroll
  for (int i = 0; i < NPIX; ++i) v[i] = *reinterpret_cast<const VT*>(base[i]);
  if constexpr (SP != 0) {
#pragma unroll
    for (int s = 1; s < SP; ++s) {
      VT part[NPIX];
#pragma unroll
      for (int i = 0; i < NPIX; ++i) part[i] = *reinterpret_cast<const VT*>(base[i] + (size_t)s * (128 * 128));
#pragma unroll
      for (int i = 0; i < NPIX; ++i) v[i] += part[i];
    }
  } else {
    for (int s = 1; s < parts; ++s) {
      VT part[NPIX];
#pragma unroll
      for (int i = 0; i < NPIX; ++i) part[i] = *reinterpret_cast<const VT*>(base[i] + (size_t)s * (128 * 128));
#pragma unroll
      for (int i = 0; i < NPIX; ++i) v[i] += part[i];
    }
  }
  const VT sc = *reinterpret_cast<const VT*>(df.scale + n) * df.alpha;
  const VT sh = *reinterpret_cast<const VT*>(df.shift + n);
  // one fused multiply-add per channel, spelled out: what conv_splitk_reduce_kernel's `v * sc + sh` compiles to (v_pk_fma_f32)
  constexpr int NV = (int)(sizeof(VT) / sizeof(float));
#pragma unroll
  for (int i = 0; i < NPIX; ++i)
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      float t = __builtin_fmaf(v[i][e], sc[e], sh[e]);
      if (df.relu) t = relu_keep_nan(t);
      v[i][e] = ok[i] ? t : 0.f;
    }
}

// DEFER: -1 = the input tensor is read; 0 / 2 / 4 = the producer's split-K partial tiles are summed (any number of parts / exactly 2 / 4)
template <int LANES, int DEFER>
__global__ __launch_bounds__(8 * LANES) void wino6_input_small_kernel(const float* __restrict__ x, float* __restrict__ V, const WinoGeom g, const DeferredSrc df) {
  __shared__ f32x2 u[8][8][LANES];
  const int l = threadIdx.x % LANES, a = threadIdx.x / LANES;      // a: the column (stage 1), then the row (stage 2)
  const int slices = g.C / (2 * LANES);
  const long long tile = blockIdx.x / slices;
  const int cg = (int)(blockIdx.x % slices) * LANES + l;
  int b, oy, ox, ty, tx;
  decode_tile(g, tile, b, oy, ox, ty, tx);
  const int r0 = oy + g.d * (6 * ty - 1), c0 = ox + g.d * (6 * tx - 1);
  const float* xb = x + (size_t)b * g.H * g.W * g.C + cg * 2;
  {
    const int c = c0 + a * g.d;
    const bool cok = (unsigned)c < (unsigned)g.W;
    f32x2 p[8];
    if constexpr (DEFER >= 0) {
      long long m[8];
      bool ok[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = r0 + i * g.d;
        ok[i] = cok && (unsigned)r < (unsigned)g.H;
        m[i] = ((long long)b * g.H + r) * g.W + c;
      }
      deferred_line<f32x2, DEFER, 8>(df, m, ok, cg * 2, p);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = r0 + i * g.d;
        f32x2 v = {0.f, 0.f};
        if (cok && (unsigned)r < (unsigned)g.H) v = *reinterpret_cast<const f32x2*>(xb + ((size_t)r * g.W + c) * g.C);
        p[i] = v;
      }
    }
    bt8(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);      // column a: p <- B^T p
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i][a][l] = p[i];
  }
  __syncthreads();
  f32x2 q[8], t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = u[a][j][l];
  bt8(q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);          // row a: (p B)[a][:]
  const size_t pos_stride = (size_t)g.m_pad * g.C;
  float* o = V + (size_t)tile * g.C + cg * 2 + (size_t)(a * 8) * pos_stride;
#pragma unroll
  for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x2*>(o + j * pos_stride) = t[j];
}

template <bool RES, int LANES>
__global__ __launch_bounds__(8 * LANES) void wino6_output_small_kernel(const float* __restrict__ Mb, const float* __restrict__ scale,
                                                                      const float* __restrict__ shift, const float* __restrict__ res,
                                                                      float* __restrict__ y, const WinoGeom g, int relu) {
  __shared__ f32x2 qs[8][6][LANES];
  const int l = threadIdx.x % LANES, a = threadIdx.x / LANES;      // a: the row of positions (stage 1), then the output column (stage 2)
  const int slices = g.C / (2 * LANES);
  const long long tile = blockIdx.x / slices;
  const int ng = (int)(blockIdx.x % slices) * LANES + l;
  int b, oy, ox, ty, tx;
  decode_tile(g, tile, b, oy, ox, ty, tx);
  const int r0 = oy + g.d * 6 * ty, c0 = ox + g.d * 6 * tx;
  if (r0 >= g.H || c0 >= g.W) return;                               // tile entirely outside its sub-grid (the whole workgroup leaves)
  const size_t pos_stride = (size_t)g.m_pad * g.C;
  {
    const float* sp = Mb + (size_t)tile * g.C + ng * 2 + (size_t)(a * 8) * pos_stride;
    f32x2 m[8], q[6];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(sp + j * pos_stride));
    at8(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], q[0], q[1], q[2], q[3], q[4], q[5]);
#pragma unroll
    for (int j = 0; j < 6; ++j) qs[a][j][l] = q[j];
  }
  __syncthreads();
  if (a >= 6) return;
  const int c = c0 + a * g.d;
  const size_t img = (size_t)b * g.H * g.W;
  f32x2 rc[6];
  if constexpr (RES) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int r = r0 + k * g.d;
      const bool ok = c < g.W && r < g.H;
      const size_t off = ok ? (img + (size_t)r * g.W + c) * g.C + ng * 2 : 0;
      rc[k] = *reinterpret_cast<const f32x2*>(res + off);
    }
  }
  const f32x2 sc = *reinterpret_cast<const f32x2*>(scale + ng * 2);
  const f32x2 sh = *reinterpret_cast<const f32x2*>(shift + ng * 2);
  f32x2 o[6];
  at8(qs[0][a][l], qs[1][a][l], qs[2][a][l], qs[3][a][l], qs[4][a][l], qs[5][a][l], qs[6][a][l], qs[7][a][l], o[0], o[1], o[2], o[3], o[4], o[5]);
  if (c >= g.W) return;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int r = r0 + k * g.d;
    if (r >= g.H) continue;
    const size_t off = (img + (size_t)r * g.W + c) * g.C + ng * 2;
    f32x2 v = o[k] * sc + sh;
    if constexpr (RES) v += rc[k];
    if (relu) { v.x = relu_keep_nan(v.x); v.y = relu_keep_nan(v.y); }
    *reinterpret_cast<f32x2*>(y + off) = v;
  }
}

// F(4x4): a workgroup = one tile x LANES 4-channel groups, 6 lines
template <int LANES, int DEFER>
__global__ __launch_bounds__(6 * LANES) void wino4_input_small_kernel(const float* __restrict__ x, float* __restrict__ V, const WinoGeom g, const DeferredSrc df) {
  __shared__ f32x4 u[6][6][LANES];
  const int l = threadIdx.x % LANES, a = threadIdx.x / LANES;
  const int slices = g.C / (4 * LANES);
  const long long tile = blockIdx.x / slices;
  const int cg = (int)(blockIdx.x % slices) * LANES + l;
  int b, oy, ox, ty, tx;
  decode_tile(g, tile, b, oy, ox, ty, tx);
  const int r0 = oy + g.d * (4 * ty - 1), c0 = ox + g.d * (4 * tx - 1);
  const float* xb = x + (size_t)b * g.H * g.W * g.C + cg * 4;
  {
    const int c = c0 + a * g.d;
    const bool cok = (unsigned)c < (unsigned)g.W;
    f32x4 p[6];
    if constexpr (DEFER >= 0) {
      long long m[6];
      bool ok[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int r = r0 + i * g.d;
        ok[i] = cok && (unsigned)r < (unsigned)g.H;
        m[i] = ((long long)b * g.H + r) * g.W + c;
      }
      deferred_line<f32x4, DEFER, 6>(df, m, ok, cg * 4, p);
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int r = r0 + i * g.d;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (cok && (unsigned)r < (unsigned)g.H) v = *reinterpret_cast<const f32x4*>(xb + ((size_t)r * g.W + c) * g.C);
        p[i] = v;
      }
    }
    bt6(p[0], p[1], p[2], p[3], p[4], p[5], p[0], p[1], p[2], p[3], p[4], p[5]);
#pragma unroll
    for (int i = 0; i < 6; ++i) u[i][a][l] = p[i];
  }
  __syncthreads();
  f32x4 t0, t1, t2, t3, t4, t5;
  bt6(u[a][0][l], u[a][1][l], u[a][2][l], u[a][3][l], u[a][4][l], u[a][5][l], t0, t1, t2, t3, t4, t5);
  const size_t pos_stride = (size_t)g.m_pad * g.C;
  float* o = V + (size_t)tile * g.C + cg * 4 + (size_t)(a * 6) * pos_stride;
  *reinterpret_cast<f32x4*>(o) = t0;
  *reinterpret_cast<f32x4*>(o + pos_stride) = t1;
  *reinterpret_cast<f32x4*>(o + 2 * pos_stride) = t2;
  *reinterpret_cast<f32x4*>(o + 3 * pos_stride) = t3;
  *reinterpret_cast<f32x4*>(o + 4 * pos_stride) = t4;
  *reinterpret_cast<f32x4*>(o + 5 * pos_stride) = t5;
}

template <bool RES, int LANES>
__global__ __launch_bounds__(6 * LANES) void wino4_output_small_kernel(const float* __restrict__ Mb, const float* __restrict__ scale,
                                                                      const float* __restrict__ shift, const float* __restrict__ res,
                                                                      float* __restrict__ y, const WinoGeom g, int relu) {
  __shared__ f32x4 qs[6][4][LANES];
  const int l = threadIdx.x % LANES, a = threadIdx.x / LANES;
  const int slices = g.C / (4 * LANES);
  const long long tile = blockIdx.x / slices;
  const int ng = (int)(blockIdx.x % slices) * LANES + l;
  int b, oy, ox, ty, tx;
  decode_tile(g, tile, b, oy, ox, ty, tx);
  const int r0 = oy + g.d * 4 * ty, c0 = ox + g.d * 4 * tx;
  if (r0 >= g.H || c0 >= g.W) return;
  const size_t pos_stride = (size_t)g.m_pad * g.C;
  {
    const float* sp = Mb + (size_t)tile * g.C + ng * 4 + (size_t)(a * 6) * pos_stride;
    const f32x4 m0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp));
    const f32x4 m1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + pos_stride));
    const f32x4 m2 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + 2 * pos_stride));
    const f32x4 m3 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + 3 * pos_stride));
    const f32x4 m4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + 4 * pos_stride));
    const f32x4 m5 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sp + 5 * pos_stride));
    f32x4 q0, q1, q2, q3;
    at6(m0, m1, m2, m3, m4, m5, q0, q1, q2, q3);
    qs[a][0][l] = q0; qs[a][1][l] = q1; qs[a][2][l] = q2; qs[a][3][l] = q3;
  }
  __syncthreads();
  if (a >= 4) return;
  const int c = c0 + a * g.d;
  const size_t img = (size_t)b * g.H * g.W;
  f32x4 rc[4];
  if constexpr (RES) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + k * g.d;
      const bool ok = c < g.W && r < g.H;
      const size_t off = ok ? (img + (size_t)r * g.W + c) * g.C + ng * 4 : 0;
      rc[k] = *reinterpret_cast<const f32x4*>(res + off);
    }
  }
  const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ng * 4);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + ng * 4);
  f32x4 o0, o1, o2, o3;
  at6(qs[0][a][l], qs[1][a][l], qs[2][a][l], qs[3][a][l], qs[4][a][l], qs[5][a][l], o0, o1, o2, o3);
  if (c >= g.W) return;
  const f32x4 o[4] = {o0, o1, o2, o3};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + k * g.d;
    if (r >= g.H) continue;
    const size_t off = (img + (size_t)r * g.W + c) * g.C + ng * 4;
    f32x4 v = o[k] * sc + sh;
    if constexpr (RES) v += rc[k];
    if (relu) v = relu_keep_nan(v);
    *reinterpret_cast<f32x4*>(y + off) = v;
  }
}

// Which launches take the small-problem variants: fewer workgroups of the one-thread-per-tile kernels than wino_small_maxwg
// (options.h; default 1024 = four per CU -- above that those kernels stream at 5-6 TB/s and are the better form), whole 64-channel
// slices, and a grid that fits.  F(5x5) has no small variant (dilation-4 layers of 480 x 480 maps only).
int wino_small_lanes(long long n_tiles, int C, int m) {
  const long long maxwg = opt(OPT_WINO_SMALL_MAXWG);
  if (maxwg <= 0 || (m != 4 && m != 6)) return 0;
  const long long big_wgs = (n_tiles * (m == 4 ? C / 4 : C / 2) + 255) / 256;
  if (big_wgs >= maxwg) return 0;
  const int per32 = m == 4 ? 128 : 64, per16 = per32 / 2;
  const int lanes = C % per32 == 0 ? 32 : (C % per16 == 0 ? 16 : 0);
  if (!lanes || n_tiles * (C / (lanes * (m == 4 ? 4 : 2))) > 0x7fffffffLL) return 0;
  return lanes;
}

int grid_for(long long total) {
  long long blocks = (total + 255) / 256;
  if (blocks > 256LL * 64) blocks = 256LL * 64;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

void wino_geometry(int B, int H, int W, int dil, int* th, int* tw, long long* n_tiles, long long* m_pad, int gran, int m) {
  const int hs = (H + dil - 1) / dil, wsub = (W + dil - 1) / dil;   // largest sub-grid
  *th = (hs + m - 1) / m;
  *tw = (wsub + m - 1) / m;
  *n_tiles = (long long)B * dil * dil * *th * *tw;
  *m_pad = (*n_tiles + gran - 1) / gran * gran;                      // whole GEMM tiles per position
}

// U[(i*6+l)][n][c] = (G g G^T)[i][l], accumulated in double and rounded once
void wino_transform_weights(const float* w_oihw, int cout, int cin, float* out, int m) {
  static const double G4[6][3] = {{64.0 / 81, 0, 0},
                                  {-128.0 / 243, -32.0 / 81, -8.0 / 27},
                                  {-128.0 / 243, 32.0 / 81, -8.0 / 27},
                                  {32.0 / 243, 16.0 / 81, 8.0 / 27},
                                  {32.0 / 243, -16.0 / 81, 8.0 / 27},
                                  {0, 0, 1}};
  static const double G6[8][3] = {{-1, 0, 0},
                                  {32.0 / 45, 16.0 / 45, 8.0 / 45},
                                  {32.0 / 45, -16.0 / 45, 8.0 / 45},
                                  {-2.0 / 9, -2.0 / 9, -2.0 / 9},
                                  {-2.0 / 9, 2.0 / 9, -2.0 / 9},
                                  {1.0 / 90, 1.0 / 45, 2.0 / 45},
                                  {1.0 / 90, -1.0 / 45, 2.0 / 45},
                                  {0, 0, 1}};
  static const double G5[7][3] = {{-4.0 / 3, 0, 0},
                                  {16.0 / 15, 8.0 / 15, 4.0 / 15},
                                  {16.0 / 21, -8.0 / 21, 4.0 / 21},
                                  {-1.0 / 3, -1.0 / 3, -1.0 / 3},
                                  {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                  {1.0 / 210, 1.0 / 70, 3.0 / 70},
                                  {0, 0, 1}};
  const int n_in = m + 2;                                       // 6 (F(4x4)), 7 (F(5x5)) or 8 (F(6x6)) points per dimension
  const double (*G)[3] = m == 6 ? G6 : (m == 5 ? G5 : G4);
  const size_t plane = (size_t)cout * cin;
  for (int n = 0; n < cout; ++n)
    for (int c = 0; c < cin; ++c) {
      const float* g = w_oihw + ((size_t)n * cin + c) * 9;
      double t[8][3];
      for (int i = 0; i < n_in; ++i)
        for (int k = 0; k < 3; ++k) t[i][k] = G[i][0] * g[0 * 3 + k] + G[i][1] * g[1 * 3 + k] + G[i][2] * g[2 * 3 + k];
      for (int i = 0; i < n_in; ++i)
        for (int l = 0; l < n_in; ++l)
          out[(size_t)(i * n_in + l) * plane + (size_t)n * cin + c] =
              (float)(t[i][0] * G[l][0] + t[i][1] * G[l][1] + t[i][2] * G[l][2]);
    }
}

static std::atomic<long long> g_deferred_count{0};
long long wino_deferred_count() { return g_deferred_count.load(std::memory_order_relaxed); }

bool wino_input_accepts_deferred(int B, int H, int W, int C, int dil, int gran, int m) {
  if (C % 4 || m < 4 || m > 6 || C % 128 != 0) return false;      // (the producer's 128-wide n-tiles are whole)
  int th, tw;
  long long n_tiles, m_pad;
  wino_geometry(B, H, W, dil, &th, &tw, &n_tiles, &m_pad, gran, m);
  return wino_small_lanes(n_tiles, C, m) != 0;
}

int launch_wino_input(const float* x, float* V, int B, int H, int W, int C, int dil, hipStream_t s, int gran, int m, long long m_pad_total,
                      const DeferredSplit* df) {
  if (C % 4 || m < 4 || m > 6) return fail(-2, "wino_input: channels must be a multiple of 4, tiles 4x4, 5x5 or 6x6");
  WinoGeom g{B, H, W, C, dil, 0, 0, 0, 0};
  wino_geometry(B, H, W, dil, &g.th, &g.tw, &g.n_tiles, &g.m_pad, gran, m);
  if (m_pad_total > 0) g.m_pad = m_pad_total;     // this tensor's tiles are a sub-range of a position's rows (V already points at its first tile)
  const long long total = g.n_tiles * (m == 4 ? C / 4 : C / 2);
  const bool defer = df && df->valid;
  DeferredSrc src{};
  if (defer) {
    if (df->cout != C || (long long)df->M != (long long)B * H * W || !wino_small_lanes(g.n_tiles, C, m))
      return fail(-2, "wino_input: deferred split-K partials do not match this transform");
    src = DeferredSrc{df->partial, df->scale, df->shift, df->alpha, df->split_p, df->ntiles, df->relu};
    g_deferred_count.fetch_add(1, std::memory_order_relaxed);
  }
  if (const int lanes = wino_small_lanes(g.n_tiles, C, m)) {
    const unsigned grid = (unsigned)(g.n_tiles * (C / (lanes * (m == 4 ? 4 : 2))));
#define PEANUT_WINO_SMALL_IN(K, L, T)                                                                          \
    do {                                                                                                       \
      if (defer && src.split_p == 4) hipLaunchKernelGGL((K<L, 4>), dim3(grid), dim3(T), 0, s, x, V, g, src);   \
      else if (defer && src.split_p == 2) hipLaunchKernelGGL((K<L, 2>), dim3(grid), dim3(T), 0, s, x, V, g, src); \
      else if (defer) hipLaunchKernelGGL((K<L, 0>), dim3(grid), dim3(T), 0, s, x, V, g, src);                  \
      else hipLaunchKernelGGL((K<L, -1>), dim3(grid), dim3(T), 0, s, x, V, g, src);                            \
    } while (0)
    if (m == 6 && lanes == 32) PEANUT_WINO_SMALL_IN(wino6_input_small_kernel, 32, 256);
    else if (m == 6) PEANUT_WINO_SMALL_IN(wino6_input_small_kernel, 16, 128);
    else if (lanes == 32) PEANUT_WINO_SMALL_IN(wino4_input_small_kernel, 32, 192);
    else PEANUT_WINO_SMALL_IN(wino4_input_small_kernel, 16, 96);
#undef PEANUT_WINO_SMALL_IN
  } else if (m == 6) hipLaunchKernelGGL(wino6_input_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, V, g);
  else if (m == 5) hipLaunchKernelGGL(wino5_input_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, V, g);
  else hipLaunchKernelGGL(wino_input_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, V, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-3, std::string("wino_input launch: ") + hipGetErrorString(e));
  return 0;
}

int launch_wino_output(const float* Mb, const float* scale, const float* shift, const float* res, float* y,
                       int B, int H, int W, int C, int dil, int relu, hipStream_t s, int gran, int m, long long m_pad_total) {
  if (C % 4 || m < 4 || m > 6) return fail(-2, "wino_output: channels must be a multiple of 4, tiles 4x4, 5x5 or 6x6");
  WinoGeom g{B, H, W, C, dil, 0, 0, 0, 0};
  wino_geometry(B, H, W, dil, &g.th, &g.tw, &g.n_tiles, &g.m_pad, gran, m);
  if (m_pad_total > 0) g.m_pad = m_pad_total;
  const long long total = g.n_tiles * (m == 4 ? C / 4 : C / 2);
  if (const int lanes = wino_small_lanes(g.n_tiles, C, m)) {
    const unsigned grid = (unsigned)(g.n_tiles * (C / (lanes * (m == 4 ? 4 : 2))));
#define PEANUT_WINO_SMALL_OUT(K, R, L, T) hipLaunchKernelGGL((K<R, L>), dim3(grid), dim3(T), 0, s, Mb, scale, shift, res, y, g, relu)
    if (m == 6 && lanes == 32) { if (res) PEANUT_WINO_SMALL_OUT(wino6_output_small_kernel, true, 32, 256); else PEANUT_WINO_SMALL_OUT(wino6_output_small_kernel, false, 32, 256); }
    else if (m == 6) { if (res) PEANUT_WINO_SMALL_OUT(wino6_output_small_kernel, true, 16, 128); else PEANUT_WINO_SMALL_OUT(wino6_output_small_kernel, false, 16, 128); }
    else if (lanes == 32) { if (res) PEANUT_WINO_SMALL_OUT(wino4_output_small_kernel, true, 32, 192); else PEANUT_WINO_SMALL_OUT(wino4_output_small_kernel, false, 32, 192); }
    else { if (res) PEANUT_WINO_SMALL_OUT(wino4_output_small_kernel, true, 16, 96); else PEANUT_WINO_SMALL_OUT(wino4_output_small_kernel, false, 16, 96); }
#undef PEANUT_WINO_SMALL_OUT
  } else if (m == 6 && res) hipLaunchKernelGGL(wino6_output_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, Mb, scale, shift, res, y, g, relu);
  else if (m == 6) hipLaunchKernelGGL(wino6_output_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, Mb, scale, shift, res, y, g, relu);
  else if (m == 5) hipLaunchKernelGGL(wino5_output_kernel, dim3(grid_for(total)), dim3(256), 0, s, Mb, scale, shift, res, y, g, relu);
  else if (res) hipLaunchKernelGGL(wino_output_kernel<true>, dim3(grid_for(total)), dim3(256), 0, s, Mb, scale, shift, res, y, g, relu);
  else hipLaunchKernelGGL(wino_output_kernel<false>, dim3(grid_for(total)), dim3(256), 0, s, Mb, scale, shift, res, y, g, relu);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-3, std::string("wino_output launch: ") + hipGetErrorString(e));
  return 0;
}

}  // namespace peanut
