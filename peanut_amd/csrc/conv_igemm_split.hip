// Split-precision variant of the fused implicit-GEMM convolution: fp32 operands are split into a
// 16-bit head and a 16-bit tail (x = x_hi + x_lo, both bf16 -- or both fp16), and each fp32 product
// is rebuilt from three matrix-core products with fp32 accumulation:
//        a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi            (a_lo*b_lo ~ 2^-16 relative is dropped)
// on v_mfma_f32_32x32x16_{bf16,f16} (16x the fp32-MFMA rate, so ~5.3x after the 3 passes).
// Activations stay fp32 NHWC in HBM (the split happens in registers while staging to LDS), weights
// are split once at pack time; accumulators, BN scale/shift, residual and ReLU stay fp32.
// End-to-end deviation from the fp32 reference on the seeded PSPNet: ~1e-4 (bf16x3) / ~2e-5 (fp16x3)
// max-abs on logits, against the 1e-3 bound of BASELINE.json; gated by tests/test_pred_gpu.py.
// bf16x3 keeps fp32's exponent range; fp16x3 is more accurate but overflows above 65504.
//
// Same tiling, pipeline, epilogue and reference call sites as conv_igemm.hip.  Differences:
//   * LDS holds four 16-bit planes per stage (A_hi, A_lo, W_hi, W_lo), rows of BK=32 halves (64 B)
//     with the 16-byte slot index XOR-swizzled by ((row>>2)&3) -> ds_read_b128 conflict-free without
//     padding (2 stages x 32 KiB; the epilogue staging needs 66 KiB -> 2 workgroups per CU).
//   * one ds_read_b128 = the 8 k-values a lane feeds to one 32x32x16 MFMA.
#include "conv_common.h"

namespace peanut {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <bool FP16> struct Half;
template <> struct Half<false> {
  typedef __bf16 T; typedef bf16x4 V4; typedef bf16x8 V8;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Half<true> {
  typedef _Float16 T; typedef f16x4 V4; typedef f16x8 V8;
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// split 4 floats into 4 heads + 4 tails (round-to-nearest-even conversions)
template <bool FP16>
__device__ __forceinline__ void split4(const f32x4 v, u32x2* hi, u32x2* lo) {
  typedef typename Half<FP16>::T T;
  typedef typename Half<FP16>::V4 V4;
  V4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const T t = (T)v[e];
    h[e] = t;
    l[e] = (T)(v[e] - (float)t);
  }
  *hi = __builtin_bit_cast(u32x2, h);
  *lo = __builtin_bit_cast(u32x2, l);
}

template <int BM, int BN, int WM, int WN, bool FP16>
__global__ __launch_bounds__(256) void conv_igemm_split_kernel(const ConvKParams p) {
  constexpr int BK = 32;
  constexpr int KV = BK / 4;                 // fp32 float4 pieces per A row
  constexpr int RB = BK * 2;                 // bytes per LDS row of one 16-bit plane (64)
  constexpr int A_F4 = BM * KV;              // A: float4 pieces per k-tile
  constexpr int B_P = 2 * BN * (BK / 8);     // W: 16-byte pieces per k-tile (hi plane then lo plane)
  constexpr int A_PER = A_F4 / 256, B_PER = B_P / 256;
  static_assert(A_F4 % 256 == 0 && B_P % 256 == 0, "tile must divide over 256 threads");
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 32, NI = TN / 32;
  constexpr int PLANE_A = BM * RB, PLANE_B = BN * RB;          // bytes
  constexpr int STAGE = 2 * PLANE_A + 2 * PLANE_B;            // bytes per stage
  constexpr int CS = BN + 4;
  constexpr int SMEM_BYTES = (2 * STAGE > BM * CS * 4) ? 2 * STAGE : BM * CS * 4;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  typedef typename Half<FP16>::V8 V8;

  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];

  const int tid = threadIdx.x;
  const Work wk = decode_work(p);
  const int mt = wk.mt, nt = wk.nt, nk = wk.kt1 - wk.kt0;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread staging coordinates (A) ----
  int a_iy0[A_PER], a_ix0[A_PER], a_pix[A_PER];
  const int a_c4 = (tid % KV) * 4;           // first k of this thread's float4 inside the k-tile
#pragma unroll
  for (int j = 0; j < A_PER; ++j) {
    const int row = (tid + 256 * j) / KV;
    const int m = m0 + row;
    if (m < p.M) {
      const int b = m / p.HoWo;
      const int rem = m - b * p.HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_iy0[j] = oy * p.stride - p.pad;
      a_ix0[j] = ox * p.stride - p.pad;
      a_pix[j] = b * p.H * p.W;
    } else {
      a_iy0[j] = -(1 << 28);
      a_ix0[j] = 0;
      a_pix[j] = 0;
    }
  }
  // LDS byte offsets of this thread's pieces (swizzled 16-B slot = slot ^ ((row>>2)&3))
  int a_lds[A_PER], b_lds[B_PER];
#pragma unroll
  for (int j = 0; j < A_PER; ++j) {
    const int idx = tid + 256 * j, row = idx / KV, c4 = idx % KV;     // c4: which 4-k group (8 B of halves)
    a_lds[j] = row * RB + ((((c4 >> 1) ^ (row >> 2)) & 3) << 4) + (c4 & 1) * 8;
  }
#pragma unroll
  for (int j = 0; j < B_PER; ++j) {
    const int idx = tid + 256 * j;
    const int plane = idx / (BN * (BK / 8)), q = idx % (BN * (BK / 8));
    const int row = q / (BK / 8), s = q % (BK / 8);
    b_lds[j] = 2 * PLANE_A + plane * PLANE_B + row * RB + (((s ^ (row >> 2)) & 3) << 4);
  }

  f32x4 ra[A_PER];
  u32x4 rb[B_PER];
  int tap = wk.kt0 % p.ntaps, cbase = (wk.kt0 / p.ntaps) * BK;
  int ky = tap / p.kw, kx = tap - ky * p.kw;
  const unsigned char* wtile = reinterpret_cast<const unsigned char*>(p.w) +
                               (p.mt_per_group ? (size_t)(mt / p.mt_per_group) * p.w_group_stride * 4 : 0) +
                               ((size_t)nt * p.nkt + wk.kt0) * (BN * BK * 4);

  // Staging is software-pipelined and branch-free exactly like conv_igemm.hip: addresses of the next
  // k-tile are prepared one iteration ahead (pure ALU in the MFMA shadow), out-of-range taps read the zero
  // page through an arithmetic address select, the iterator advances with selects.
  unsigned long long a_addr[A_PER];
  const unsigned char* b_tile;
#define SPLIT_PREP_ADDR()                                                                               \
  {                                                                                                     \
    const bool second_ = cbase >= p.c1;                                                                 \
    const float* src = second_ ? p.x2 : p.x;                                                            \
    const int C = second_ ? p.c2 : p.c1, cb = second_ ? cbase - p.c1 : cbase;                           \
    const int dy = ky * p.dil, dx = kx * p.dil;                                                         \
    static_for<A_PER>([&](auto J) __attribute__((always_inline)) {                                      \
      constexpr int j = decltype(J)::value;                                                             \
      const int iy = a_iy0[j] + dy, ix = a_ix0[j] + dx;                                                 \
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;                     \
      const float* ptr = src + (size_t)(a_pix[j] + iy * p.W + ix) * C + cb + a_c4;                      \
      const unsigned long long m_ = ok ? ~0ull : 0ull;                                                  \
      a_addr[j] = ((unsigned long long)ptr & m_) | ((unsigned long long)p.zeros & ~m_);                 \
    });                                                                                                 \
    b_tile = wtile;                                                                                     \
    wtile += BN * BK * 4;                                                                               \
    const int tap1_ = tap + 1, kx1_ = kx + 1;                                                           \
    const bool wrap_ = tap1_ == p.ntaps, kxw_ = kx1_ == p.kw;                                           \
    tap = wrap_ ? 0 : tap1_;                                                                            \
    ky = wrap_ ? 0 : (kxw_ ? ky + 1 : ky);                                                              \
    kx = (wrap_ || kxw_) ? 0 : kx1_;                                                                    \
    cbase += wrap_ ? BK : 0;                                                                            \
  }

#define SPLIT_ISSUE_LOADS()                                                                             \
  {                                                                                                     \
    static_for<A_PER>([&](auto J) __attribute__((always_inline)) {                                      \
      constexpr int j = decltype(J)::value;                                                             \
      ra[j] = *reinterpret_cast<const f32x4*>(a_addr[j]);                                               \
    });                                                                                                 \
    static_for<B_PER>([&](auto J) __attribute__((always_inline)) {                                      \
      constexpr int j = decltype(J)::value;                                                             \
      rb[j] = *reinterpret_cast<const u32x4*>(b_tile + (size_t)(tid + 256 * j) * 16);                   \
    });                                                                                                 \
  }

#define SPLIT_STORE_TILES(stage)                                                                        \
  {                                                                                                     \
    unsigned char* st_ = (stage);                                                                       \
    static_for<A_PER>([&](auto J) __attribute__((always_inline)) {                                      \
      constexpr int j = decltype(J)::value;                                                             \
      u32x2 h, l;                                                                                       \
      split4<FP16>(ra[j], &h, &l);                                                                      \
      *reinterpret_cast<u32x2*>(st_ + a_lds[j]) = h;                                                    \
      *reinterpret_cast<u32x2*>(st_ + PLANE_A + a_lds[j]) = l;                                          \
    });                                                                                                 \
    static_for<B_PER>([&](auto J) __attribute__((always_inline)) {                                      \
      constexpr int j = decltype(J)::value;                                                             \
      *reinterpret_cast<u32x4*>(st_ + b_lds[j]) = rb[j];                                                \
    });                                                                                                 \
  }

  // ---- MFMA fragment coordinates ----
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  // lane (li, hi) reads k = 16*step + 8*hi .. +7 of row li: 16-B slot (2*step + hi), swizzled by the row
  int a_rd[MI], b_rd[NI], a_sw[MI], b_sw[NI];
#pragma unroll
  for (int t = 0; t < MI; ++t) {
    const int row = wm * TM + t * 32 + li;
    a_rd[t] = row * RB;
    a_sw[t] = (row >> 2) & 3;
  }
#pragma unroll
  for (int u = 0; u < NI; ++u) {
    const int row = wn * TN + u * 32 + li;
    b_rd[u] = 2 * PLANE_A + row * RB;
    b_sw[u] = (row >> 2) & 3;
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

#define SPLIT_COMPUTE(cur)                                                                              \
  _Pragma("unroll") for (int step = 0; step < BK / 16; ++step) {                                        \
    V8 ah[MI], al[MI], bh[NI], bl[NI];                                                                  \
    _Pragma("unroll") for (int t = 0; t < MI; ++t) {                                                    \
      const int off = a_rd[t] + ((((2 * step + hi) ^ a_sw[t]) & 3) << 4);                               \
      ah[t] = *reinterpret_cast<const V8*>((cur) + off);                                                \
      al[t] = *reinterpret_cast<const V8*>((cur) + PLANE_A + off);                                      \
    }                                                                                                   \
    _Pragma("unroll") for (int u = 0; u < NI; ++u) {                                                    \
      const int off = b_rd[u] + ((((2 * step + hi) ^ b_sw[u]) & 3) << 4);                               \
      bh[u] = *reinterpret_cast<const V8*>((cur) + off);                                                \
      bl[u] = *reinterpret_cast<const V8*>((cur) + PLANE_B + off);                                      \
    }                                                                                                   \
    /* term-major order: consecutive MFMAs hit DIFFERENT accumulators */                                \
    _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                      \
      _Pragma("unroll") for (int u = 0; u < NI; ++u) acc[t][u] = Half<FP16>::mfma(al[t], bh[u], acc[t][u]); \
    _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                      \
      _Pragma("unroll") for (int u = 0; u < NI; ++u) acc[t][u] = Half<FP16>::mfma(ah[t], bl[u], acc[t][u]); \
    _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                      \
      _Pragma("unroll") for (int u = 0; u < NI; ++u) acc[t][u] = Half<FP16>::mfma(ah[t], bh[u], acc[t][u]); \
  }

  SPLIT_PREP_ADDR();
  SPLIT_ISSUE_LOADS();                 // k-tile 0
  SPLIT_STORE_TILES(smem_raw);
  SPLIT_PREP_ADDR();
  if (nk > 1) SPLIT_ISSUE_LOADS();     // k-tile 1
  SPLIT_PREP_ADDR();                   // addresses of k-tile 2
  __syncthreads();

  int kt = 0;
  for (; kt + 2 < nk; ++kt) {          // steady state: no conditionals inside
    unsigned char* const nxt = smem_raw + ((kt + 1) & 1) * STAGE;
    const unsigned char* const cur = smem_raw + (kt & 1) * STAGE;
    SPLIT_STORE_TILES(nxt);
    SPLIT_ISSUE_LOADS();
    __builtin_amdgcn_sched_barrier(0);
    SPLIT_PREP_ADDR();
    SPLIT_COMPUTE(cur);
    __syncthreads();
  }
  if (kt + 1 < nk) {
    SPLIT_STORE_TILES(smem_raw + ((kt + 1) & 1) * STAGE);
    SPLIT_COMPUTE(smem_raw + (kt & 1) * STAGE);
    __syncthreads();
    ++kt;
  }
  SPLIT_COMPUTE(smem_raw + (kt & 1) * STAGE);
  __syncthreads();
#undef SPLIT_COMPUTE
#undef SPLIT_PREP_ADDR
#undef SPLIT_ISSUE_LOADS
#undef SPLIT_STORE_TILES

  // ---- epilogue shared with conv_igemm.hip (whole tile staged at once: the 64 KiB of pipeline buffers hold it) ----
  conv_epilogue<BM, BN, WM, WN, 1>(p, wk, acc, reinterpret_cast<float*>(smem_raw), m0, n0);
}

// ------------------------------------------------------------------------------------------------
// host side: 16-bit conversions (round-to-nearest-even, matching the device casts) and packing
static inline uint16_t f32_to_bf16(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)((u >> 16) | ((u & 0xffffu) ? 0x40 : 0));  // inf/nan
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f32_to_f16(float f) {
  const _Float16 h = (_Float16)f;   // host clang: IEEE RNE conversion
  uint16_t u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}
static inline float f16_to_f32(uint16_t u) {
  _Float16 h;
  __builtin_memcpy(&h, &u, 2);
  return (float)h;
}

// w_oihw -> [ntile][ktile]{ hi plane [BN][32], lo plane [BN][32] } of 16-bit values; same byte count as
// the fp32 packing (ktile = chunk * ntaps + tap).
void pack_conv_weights_split(const float* w, int cout, int cin_real, int cin_pad, int kh, int kw, int bn_tile,
                             int fp16, void* out_) {
  const int bk = 32, ntaps = kh * kw;
  const int cout_pad = (cout + bn_tile - 1) / bn_tile * bn_tile;
  const int nchunks = cin_pad / bk, nkt = nchunks * ntaps;
  uint16_t* out = reinterpret_cast<uint16_t*>(out_);
  for (int n = 0; n < cout_pad; ++n) {
    const int nt = n / bn_tile, nn = n % bn_tile;
    for (int ch = 0; ch < nchunks; ++ch)
      for (int tap = 0; tap < ntaps; ++tap) {
        uint16_t* tile = out + ((size_t)nt * nkt + (size_t)ch * ntaps + tap) * (2 * bn_tile * bk);
        uint16_t* hi = tile + (size_t)nn * bk;
        uint16_t* lo = tile + (size_t)bn_tile * bk + (size_t)nn * bk;
        for (int c = 0; c < bk; ++c) {
          const int ci = ch * bk + c;
          const float v = (n < cout && ci < cin_real) ? w[((size_t)n * cin_real + ci) * ntaps + tap] : 0.f;
          if (fp16) {
            hi[c] = f32_to_f16(v);
            lo[c] = f32_to_f16(v - f16_to_f32(hi[c]));
          } else {
            hi[c] = f32_to_bf16(v);
            lo[c] = f32_to_bf16(v - bf16_to_f32(hi[c]));
          }
        }
      }
  }
}

template <int BM, int BN, int WM, int WN, bool FP16>
static int launch_split_t(const ConvKParams& p, float* ws, size_t ws_floats, hipStream_t stream) {
  static int slots = 0;
  return launch_with_tail_split<decltype(&conv_igemm_split_kernel<BM, BN, WM, WN, FP16>), BM, BN>(
      &conv_igemm_split_kernel<BM, BN, WM, WN, FP16>, p, ws, ws_floats, stream, &slots);
}

int launch_conv_split(const ConvKParams& p, int bn_tile, int fp16, float* ws, size_t ws_floats, hipStream_t stream) {
  if (fp16) {
    if (bn_tile == 128) return launch_split_t<128, 128, 2, 2, true>(p, ws, ws_floats, stream);
    if (bn_tile == 64) return launch_split_t<128, 64, 2, 2, true>(p, ws, ws_floats, stream);
    if (bn_tile == 32) return launch_split_t<128, 32, 4, 1, true>(p, ws, ws_floats, stream);
  } else {
    if (bn_tile == 128) return launch_split_t<128, 128, 2, 2, false>(p, ws, ws_floats, stream);
    if (bn_tile == 64) return launch_split_t<128, 64, 2, 2, false>(p, ws, ws_floats, stream);
    if (bn_tile == 32) return launch_split_t<128, 32, 4, 1, false>(p, ws, ws_floats, stream);
  }
  return fail(-2, "launch_conv_split: unsupported tile configuration");
}

}  // namespace peanut
