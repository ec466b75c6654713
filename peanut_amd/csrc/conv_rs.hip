// Implicit-GEMM convolution (3x3 and every other non-pointwise layer, any stride / dilation, one or two sources) emulated
// on the bf16 matrix cores: the register-split arithmetic of gemm_rs.hip with conv_igemm.hip's im2col-free staging.
//
// In the emulated-fp32 modes (bf16x6 / bf16x3 / fp16x3) round 2 left these layers -- the deep stem (resnet.py:591-624), layer1's and
// layer2.0's 3x3 convs (resnet.py:267-307; the wider stride-1 3x3 convs run as Winograd) -- on the fp32 MFMA kernel, where
// they are matrix-core-bound at ~100 TF/s: 2.8 ms of a 29 ms batch-32 step for 5 % of its FLOPs.  Here a k-tile is 16
// input channels of one filter tap: every thread gathers two 16-byte pieces of the shifted input pixels (out-of-image taps
// read a zero page through an arithmetic address select, as in conv_igemm.hip) and its share of the pre-split weight tile
// into VGPRs, one iteration ahead, and stores them to LDS in the image gemm_rs.hip reads -- A [128 rows][16 floats] with
// the 16-byte chunk c of row r at position c ^ ((r >> 2) & 3), B [plane][BN rows][16 bf16] with the halves of a row
// swapped for rows with bit 3 set -- then fragment read, split (rs_common.h: split_frag), 6 (3) MFMAs per product tile.
// Two LDS stages, one barrier per k-tile.  Weights: pack_weights_sx_conv, k-tile order = channel chunk outer, tap inner.
// KIND (rs_common.h) selects bf16 pieces (three: bf16x6, two: bf16x3) or two fp16 pieces (fp16x3).
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"
#include "rs_common.h"

namespace peanut {

namespace {

template <int BN, int WM, int WN, int KIND>
__global__ __launch_bounds__(256) void conv_rs_kernel(const ConvKParams p) {
  constexpr int NP = rs_pieces(KIND);
  constexpr int BM = 128, BK = 16;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int A_BYTES = BM * 64, B_BYTES = NP * BN * 32, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PER = (BM * 4) / 256;                          // 16-byte pieces of the A tile per thread (2)
  constexpr int B_CH = B_BYTES / 16, B_PER = (B_CH + 255) / 256; // 16-byte pieces of the weight tile per thread
  constexpr int CS = BN + 4;
  constexpr int EP = (BM * CS * 4 > 2 * STAGE) ? WM : 1;
  constexpr int ER = BM / EP;
  constexpr int SMEM_BYTES = (2 * STAGE > ER * CS * 4) ? 2 * STAGE : ER * CS * 4;
  static_assert(WM * WN == 4 && TM % 32 == 0 && TN % 32 == 0 && (NP == 2 || NP == 3), "tile configuration");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM_BYTES];

  const int tid = threadIdx.x;
  const Work wk = decode_work(p);
  const int mt = wk.mt, nt = wk.nt, nk = wk.kt1 - wk.kt0;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- staging coordinates: A piece j of this thread = (row idx / 4, 16-byte chunk idx % 4) ----
  int a_iy0[A_PER], a_ix0[A_PER], a_pix[A_PER], a_lds[A_PER];
  const int a_c4 = (tid & 3) * 4;
#pragma unroll
  for (int j = 0; j < A_PER; ++j) {
    const int idx = tid + 256 * j;
    const int row = idx >> 2;
    const int m = m0 + row;
    a_lds[j] = row * 64 + (((idx & 3) ^ ((row >> 2) & 3)) * 16);
    if (m < p.M) {
      const int b = m / p.HoWo;
      const int rem = m - b * p.HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[j] = oy * p.stride - p.pad;
      a_ix0[j] = ox * p.stride - p.pad;
      a_pix[j] = b * p.H * p.W;
    } else {
      a_iy0[j] = -(1 << 28);      // fails every bounds test -> zero row
      a_ix0[j] = 0;
      a_pix[j] = 0;
    }
  }
  // weight piece j: linear 16-byte piece idx of the packed tile [plane][row][2 halves]; LDS: the same with the half swapped
  int b_lds[B_PER];
#pragma unroll
  for (int j = 0; j < B_PER; ++j) {
    const int idx = tid + 256 * j;
    const int row = (idx >> 1) % BN;
    b_lds[j] = A_BYTES + ((idx & ~1) | ((idx & 1) ^ ((row >> 3) & 1))) * 16;
  }

  // k-tile iterator: channel chunk outer, filter tap inner (the order the weights are packed in)
  int tap = wk.kt0 % p.ntaps, cbase = (wk.kt0 / p.ntaps) * BK;
  int ky = tap / p.kw, kx = tap - ky * p.kw;
  const unsigned char* wtile = reinterpret_cast<const unsigned char*>(p.w) +
                               (p.mt_per_group ? (size_t)(mt / p.mt_per_group) * (size_t)p.w_group_stride : 0) +
                               ((size_t)nt * p.nkt + wk.kt0) * B_BYTES;
  f32x4 ra[A_PER], rb[B_PER];
  auto load_tile = [&]() __attribute__((always_inline)) {     // global -> registers, then advance the iterator
    const bool second = cbase >= p.c1;
    const float* src = second ? p.x2 : p.x;
    const int C = second ? p.c2 : p.c1, cb = second ? cbase - p.c1 : cbase;
    const int dy = ky * p.dil, dx = kx * p.dil;
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
      const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const float* ptr = src + (size_t)(a_pix[j] + iy * p.W + ix) * C + cb + a_c4;
      const unsigned long long msk = ok ? ~0ull : 0ull;        // arithmetic select: no exec-masked code in the loop
      const unsigned long long addr = ((unsigned long long)ptr & msk) | ((unsigned long long)p.zeros & ~msk);
      ra[j] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(addr);
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const int idx = tid + 256 * j;
      if (B_CH % 256 == 0 || idx < B_CH) rb[j] = *reinterpret_cast<const f32x4*>(wtile + (size_t)idx * 16);
    }
    wtile += B_BYTES;
    const int tap1 = tap + 1, kx1 = kx + 1;
    const bool wrap = tap1 == p.ntaps, kxw = kx1 == p.kw;
    tap = wrap ? 0 : tap1;
    ky = wrap ? 0 : (kxw ? ky + 1 : ky);
    kx = (wrap || kxw) ? 0 : kx1;
    cbase += wrap ? BK : 0;
  };
  auto store_tile = [&](unsigned char* stage) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < A_PER; ++j) *reinterpret_cast<f32x4*>(stage + a_lds[j]) = ra[j];
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const int idx = tid + 256 * j;
      if (B_CH % 256 == 0 || idx < B_CH) *reinterpret_cast<f32x4*>(stage + b_lds[j]) = rb[j];
    }
  };

  // ---- MFMA fragment coordinates (as gemm_rs.hip) ----
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const int af = (li >> 2) & 3;
  const int a_off0 = (wm * TM + li) * 64 + (((2 * hi) ^ af) * 16);
  const int a_off1 = (wm * TM + li) * 64 + (((2 * hi + 1) ^ af) * 16);
  const int b_row = A_BYTES + (wn * TN + li) * 32 + ((hi ^ ((li >> 3) & 1)) * 16);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  // ---- pipeline: k-tile kt is computed from LDS stage kt & 1 while k-tile kt + 1 sits in registers (stored to the other
  //      stage at the top of the iteration) and k-tile kt + 2 is requested ----
  load_tile();
  store_tile(smem);
  if (nk > 1) load_tile();
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* const cur = smem + (kt & 1) * STAGE;
    if (kt + 1 < nk) store_tile(smem + ((kt + 1) & 1) * STAGE);
    if (kt + 2 < nk) load_tile();
    u32x4 ap[MI][NP], bf[NP][NI];
#pragma unroll
    for (int t = 0; t < MI; ++t) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(cur + a_off0 + t * 32 * 64);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(cur + a_off1 + t * 32 * 64);
      split_frag<KIND>(v0, v1, ap[t]);
    }
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
      for (int u = 0; u < NI; ++u) bf[q][u] = *reinterpret_cast<const u32x4*>(cur + q * (BN * 32) + b_row + u * 32 * 32);
#pragma unroll
    for (int t = 0; t < MI; ++t)
#pragma unroll
      for (int u = 0; u < NI; ++u) {
        if constexpr (NP == 3) {
          acc[t][u] = mfma_pieces<KIND>(ap[t][2], bf[0][u], acc[t][u]);
          acc[t][u] = mfma_pieces<KIND>(ap[t][0], bf[2][u], acc[t][u]);
          acc[t][u] = mfma_pieces<KIND>(ap[t][1], bf[1][u], acc[t][u]);
        }
        acc[t][u] = mfma_pieces<KIND>(ap[t][1], bf[0][u], acc[t][u]);
        acc[t][u] = mfma_pieces<KIND>(ap[t][0], bf[1][u], acc[t][u]);
        acc[t][u] = mfma_pieces<KIND>(ap[t][0], bf[0][u], acc[t][u]);
      }
    __syncthreads();
  }
  conv_epilogue<BM, BN, WM, WN, EP>(p, wk, acc, reinterpret_cast<float*>(smem), m0, n0);
}

template <int BN, int WM, int WN, int KIND>
int launch_crs_t(const ConvKParams& p, float* ws, size_t ws_floats, hipStream_t stream) {
  static SlotCache slots;
  return launch_with_tail_split<decltype(&conv_rs_kernel<BN, WM, WN, KIND>), 128, BN>(&conv_rs_kernel<BN, WM, WN, KIND>, p, ws,
                                                                                      ws_floats, stream, &slots);
}

template <int KIND>
int launch_conv_rs_kind(const ConvKParams& p, int bn_tile, float* ws, size_t ws_floats, hipStream_t stream) {
  if (bn_tile == 128) return launch_crs_t<128, 2, 2, KIND>(p, ws, ws_floats, stream);
  if (bn_tile == 64) return launch_crs_t<64, 2, 2, KIND>(p, ws, ws_floats, stream);
  return launch_crs_t<32, 4, 1, KIND>(p, ws, ws_floats, stream);
}


}  // namespace

// bytes of the pre-split weights of a kh x kw layer: [n-tile][k-tile = chunk * taps + tap][plane][bn_tile][16 pieces];
// planes = the emulation kind (rs_common.h)
size_t sx_conv_packed_bytes(int cin_pad, int cout, int kh, int kw, int bn_tile, int planes) {
  const size_t ntiles = (cout + bn_tile - 1) / bn_tile;
  return ntiles * (size_t)(cin_pad / 16) * kh * kw * rs_pieces(planes) * bn_tile * 32;
}

void pack_weights_sx_conv(const float* w_oihw, int cout, int cin_real, int cin_pad, int kh, int kw, int bn_tile, int planes,
                          float wscale, void* out) {
  unsigned short* o = static_cast<unsigned short*>(out);
  const int ntaps = kh * kw, ntiles = (cout + bn_tile - 1) / bn_tile, nchunks = cin_pad / 16, nkt = nchunks * ntaps;
  const int np = rs_pieces(planes);
  for (int nt = 0; nt < ntiles; ++nt)
    for (int ch = 0; ch < nchunks; ++ch)
      for (int tap = 0; tap < ntaps; ++tap) {
        unsigned short* tile = o + ((size_t)nt * nkt + (size_t)ch * ntaps + tap) * np * bn_tile * 16;
        for (int r = 0; r < bn_tile; ++r)
          for (int e = 0; e < 16; ++e) {
            const int n = nt * bn_tile + r, c = ch * 16 + e;
            float v = (n < cout && c < cin_real) ? w_oihw[((size_t)n * cin_real + c) * ntaps + tap] * wscale : 0.f;
            for (int q = 0; q < np; ++q) tile[((size_t)q * bn_tile + r) * 16 + e] = rs_piece_host(v, planes);
          }
      }
}

// p.x / p.x2: fp32 NHWC sources, p.w: pack_weights_sx_conv weights, p.nkt = (cin / 16) * taps
int launch_conv_rs(const ConvKParams& p, int bn_tile, int planes, float* ws, size_t ws_floats, hipStream_t stream) {
  if (p.c1 % 16 || p.c2 % 16 || planes < RS_BF16X3 || planes > RS_FP16X3 || (bn_tile != 128 && bn_tile != 64 && bn_tile != 32))
    return fail(-2, "launch_conv_rs: 16-channel granularity, a known emulation kind, 128 / 64 / 32-row weight tiles");
  static const char* const names[3][3] = {{"conv_rs3_128x128", "conv_rs3_128x64", "conv_rs3_128x32"},
                                          {"conv_rs6_128x128", "conv_rs6_128x64", "conv_rs6_128x32"},
                                          {"conv_rs3h_128x128", "conv_rs3h_128x64", "conv_rs3h_128x32"}};
  note_kernel(names[planes - 2][bn_tile == 128 ? 0 : (bn_tile == 64 ? 1 : 2)]);
  if (planes == RS_BF16X6) return launch_conv_rs_kind<RS_BF16X6>(p, bn_tile, ws, ws_floats, stream);
  if (planes == RS_FP16X3) return launch_conv_rs_kind<RS_FP16X3>(p, bn_tile, ws, ws_floats, stream);
  return launch_conv_rs_kind<RS_BF16X3>(p, bn_tile, ws, ws_floats, stream);
}

}  // namespace peanut
