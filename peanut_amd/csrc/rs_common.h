// Device helpers shared by the emulated-fp32 kernels (gemm_rs.hip: pointwise layers and Winograd position GEMMs; conv_rs.hip:
// every other conv): an fp32 MFMA fragment peeled into bf16 pieces in registers.
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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// two fp32 -> two bf16 (round to nearest even) packed in one dword, first value in the low half
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// the bf16 pieces of 8 consecutive k of one row: x = v0 | v1 (fp32) -> pc[q] = piece q of the 8 values
template <int NP>
__device__ __forceinline__ void split_frag(const f32x4& v0, const f32x4& v1, bf16x8 (&pc)[NP]) {
  u32x4 w[NP];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = i < 2 ? v0[2 * i] : v1[2 * i - 4], b = i < 2 ? v0[2 * i + 1] : v1[2 * i - 3];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const unsigned pk = cvt_pk_bf16(a, b);
      w[q][i] = pk;
      if (q + 1 < NP) {
        a -= __uint_as_float(pk << 16);            // exact: the piece agrees with the value in its leading bits
        b -= __uint_as_float(pk & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) pc[q] = __builtin_bit_cast(bf16x8, w[q]);
}


}  // namespace
}  // namespace peanut
