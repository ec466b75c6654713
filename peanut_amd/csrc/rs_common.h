// Device helpers shared by the emulated-fp32 kernels (gemm_rs.hip: pointwise layers and Winograd position GEMMs; conv_rs.hip:
// every other conv): an fp32 MFMA fragment peeled into bf16 (or fp16) pieces in registers.
//
// KIND of an emulation (ConvDesc::s_planes, the `planes` argument of the launchers and packers):
//   RS_BF16X3 = 2   two bf16 pieces per value, 3 products  (16 significand bits)
//   RS_BF16X6 = 3   three bf16 pieces, 6 products           (24 bits: fp32-class, fp32's exponent range)
//   RS_FP16X3 = 4   two fp16 pieces, 3 products             (22 bits: fp32-class inside fp16's exponent range, see below)
// fp16 pieces: x = h0 + h1 with h0 = fp16(x), h1 = fp16(x - h0); |x - h0 - h1| <= max(2^-23 |x|, 2^-25) -- the second
// bound is fp16's subnormal spacing, which h1 reaches for |x| < 2^-3.  Activations are split as they are (range
// |x| < 65504; an absolute floor of 3e-8 is fp32's own rounding at |x| = 0.5); weights are static, so the packer scales a
// layer's weights by a power of two that puts the largest at 2^13..2^14 (pack scale, undone exactly in the epilogue:
// ConvKParams::alpha) and the floor at 2^-38 of the largest weight.
#pragma once
#include "conv_common.h"

namespace peanut {
namespace {

// host side: next bf16 piece of v (round to nearest even), v <- the remainder (exact in fp32: the piece agrees with v in
// its leading bits).  The same rounding as v_cvt_pk_bf16_f32 on the device, so weights and activations split alike.
inline unsigned short bf16_piece_host(float& v) {
  unsigned bits;
  __builtin_memcpy(&bits, &v, 4);
  bits += 0x7fffu + ((bits >> 16) & 1u);
  bits &= 0xffff0000u;
  float piece;
  __builtin_memcpy(&piece, &bits, 4);
  v -= piece;
  return (unsigned short)(bits >> 16);
}

// host side: next fp16 piece of v (round to nearest even, subnormals kept), v <- the remainder (exact)
// (integer arithmetic: a host _Float16 cast goes through a libgcc soft-float call per value without F16C)
inline unsigned short f16_piece_host(float& v) {
  unsigned x;
  __builtin_memcpy(&x, &v, 4);
  const unsigned sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
  unsigned short h;
  float back;
  if (ax >= 0x7f800000u) {                       // inf / NaN
    h = (unsigned short)(sign | 0x7c00u | (ax > 0x7f800000u ? 0x200u : 0u));
    __builtin_memcpy(&back, &x, 4);
  } else if (ax >= 0x477ff000u) {                // >= 65520: rounds to inf
    h = (unsigned short)(sign | 0x7c00u);
    back = sign ? -__builtin_inff() : __builtin_inff();
  } else {
    unsigned m;
    if (ax >= 0x38800000u) {                     // normal fp16: drop 13 mantissa bits, round to nearest even
      m = ax - 0x38000000u;                      // rebias the exponent (127 -> 15)
      m = (m + 0xfffu + ((m >> 13) & 1u)) >> 13;
    } else if (ax >= 0x33000000u) {              // subnormal fp16: value / 2^-24 rounded to nearest even
      const int sh = 126 - (int)(ax >> 23);      // 14 .. 24: one more dropped bit per binade below 2^-14
      const unsigned mant = (ax & 0x7fffffu) | 0x800000u;
      const unsigned q = mant >> sh, rem = mant & ((1u << sh) - 1u), half = 1u << (sh - 1);
      m = q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u);
    } else {
      m = 0;                                     // below half the smallest subnormal
    }
    h = (unsigned short)(sign | m);
    // the piece as a float: exponent / mantissa back in place (exact)
    const unsigned e = (m >> 10) & 0x1fu, f = m & 0x3ffu;
    if (e == 0) {
      back = (float)f * 5.9604644775390625e-8f;            // f * 2^-24
      if (sign) back = -back;
    } else {
      const unsigned fb = (sign << 16) | ((e + 112u) << 23) | (f << 13);
      __builtin_memcpy(&back, &fb, 4);
    }
  }
  v -= back;
  return h;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int RS_BF16X3 = 2, RS_BF16X6 = 3, RS_FP16X3 = 4;
constexpr int rs_pieces(int kind) { return kind == RS_FP16X3 ? 2 : kind; }

// two fp32 -> two bf16 (round to nearest even) packed in one dword, first value in the low half
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// two fp32 -> two fp16 (round to nearest even) packed in one dword
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// the pieces of 8 consecutive k of one row: x = v0 | v1 (fp32) -> pc[q] = piece q of the 8 values (8 x 16 bits)
template <int KIND>
__device__ __forceinline__ void split_frag(const f32x4& v0, const f32x4& v1, u32x4 (&pc)[rs_pieces(KIND)]) {
  constexpr int NP = rs_pieces(KIND);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = i < 2 ? v0[2 * i] : v1[2 * i - 4], b = i < 2 ? v0[2 * i + 1] : v1[2 * i - 3];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      if constexpr (KIND == RS_FP16X3) {
        const unsigned pk = cvt_pk_f16(a, b);
        pc[q][i] = pk;
        if (q + 1 < NP) {
          const f16x2 h = __builtin_bit_cast(f16x2, pk);
          a -= (float)h[0];                          // exact
          b -= (float)h[1];
        }
      } else {
        const unsigned pk = cvt_pk_bf16(a, b);
        pc[q][i] = pk;
        if (q + 1 < NP) {
          a -= __uint_as_float(pk << 16);            // exact: the piece agrees with the value in its leading bits
          b -= __uint_as_float(pk & 0xffff0000u);
        }
      }
    }
  }
}

// acc += A-piece x B-piece on the matrix cores of the pieces' type
template <int KIND>
__device__ __forceinline__ f32x16 mfma_pieces(const u32x4& a, const u32x4& b, const f32x16& acc) {
  if constexpr (KIND == RS_FP16X3)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// host side, fp16 kinds: the power of two that puts the largest |w| of a layer into [2^13, 2^14) (1 for all-zero weights)
inline float rs_pack_scale(const float* w, size_t n, int kind) {
  if (kind != RS_FP16X3) return 1.f;
  float mx = 0.f;
  for (size_t i = 0; i < n; ++i) { const float a = w[i] < 0 ? -w[i] : w[i]; if (a > mx) mx = a; }
  if (!(mx > 0.f) || mx != mx || mx > 3e38f) return 1.f;
  int e;
  (void)__builtin_frexpf(mx, &e);                    // mx = f * 2^e, f in [0.5, 1)
  return __builtin_ldexpf(1.f, 14 - e);
}

// host side: piece q of a (scaled) weight
inline unsigned short rs_piece_host(float& v, int kind) { return kind == RS_FP16X3 ? f16_piece_host(v) : bf16_piece_host(v); }


}  // namespace
}  // namespace peanut
