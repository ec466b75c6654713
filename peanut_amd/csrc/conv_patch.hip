// conv_patch_kernel: the 3x3 convs with FEW input channels (16 / 32: the deep stem, prediction/mmseg/models/backbones/
// resnet.py:591-624) as a persistent im2col-free kernel with LDS-staged input patches.
//
// conv_igemm.hip runs these layers as a k-tile-per-tap implicit GEMM: every 128-pixel tile fetches its nine shifted
// views of the input from L2 (9 x the activation bytes through the vector L1) and the WHOLE weight tensor again (73 KB per
// 128 pixels for the 32 -> 64 layer: more bytes than the activations), behind nine barriers; 87-95 TF/s.  Here
//   * one workgroup per CU walks the output tiles (8 rows x 16 columns of output pixels, every n);
//   * the weights of all nine taps are put into LDS ONCE per workgroup;
//   * the input patch of a tile ((8s + 3 - s) x (16s + 3 - s) pixels for stride s, all channels) is fetched ONCE, by
//     LDS-DMA (global_load_lds_dwordx4), into one of two buffers while the previous tile computes; out-of-image pixels
//     come from the zero page (address select, no predication), so the zero padding costs nothing;
//   * a tap is a constant LDS offset: the nine taps read shifted windows of the same patch -- one barrier per TILE;
//   * the epilogue (scale / shift / ReLU, 128-byte row segments straight from the accumulator layout) of tile i is issued
//     from a second accumulator set after the first three taps of tile i + 1, so the loop's only wait (vmcnt(0) for the
//     next patch) never waits for a store.
// LDS rows are padded by one 16-byte slot (row = CIN + 4 floats), as in conv_igemm: conflict-free ds_read_b128 for the
// 2 x 16-pixel MFMA row blocks.  Same fragment layout and k order (tap outer; 8-channel groups; lanes 0-31 / 32-63 take
// channels 0-3 / 4-7 of a group) as conv_igemm_kernel: the results are bit-identical to it.
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"

namespace peanut {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int CIN, int BN, int STRIDE>
struct PatchCfg {
  static constexpr int TH = 8, TW = 16;                                   // output tile (4 waves x 2 rows, 16 columns)
  static constexpr int PH = (TH - 1) * STRIDE + 3, PW = (TW - 1) * STRIDE + 3;
  static constexpr int LS = CIN + 4, SLOTS = LS / 4;                      // floats / 16-byte slots per patch pixel (one pad slot)
  static constexpr int NS = PH * PW * SLOTS;                              // slots of a patch
  static constexpr int NP = (NS + 255) / 256;                             // LDS-DMA instructions per wave and patch
  static constexpr int PATCH_FLOATS = NP * 256 * 4;
  static constexpr int W_FLOATS = 9 * BN * LS;
  static constexpr int NI = BN / 32, NJ = CIN / 8;
};

template <int CIN, int BN, int STRIDE>
__global__ __launch_bounds__(256) void conv_patch_kernel(const ConvKParams p, int tiles_x, int tiles_y, int n_tiles) {
  using C = PatchCfg<CIN, BN, STRIDE>;
  constexpr int LS = C::LS, PW = C::PW, NI = C::NI, NJ = C::NJ, NP = C::NP;
  __shared__ __attribute__((aligned(1024))) float smem[C::W_FLOATS + 2 * C::PATCH_FLOATS];
  float* const wsm = smem + 2 * C::PATCH_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  const int G = gridDim.x;
  int tile = blockIdx.x;
  if (tile >= n_tiles) return;

  // ---- weights -> LDS, once: packed [tap][BN][CIN] (pack_conv_weights with bk = CIN: k-tile = tap) -> [tap][BN][LS] ----
  for (int i = tid; i < 9 * BN * (CIN / 4); i += 256) {
    const int row = i / (CIN / 4), c4 = i - row * (CIN / 4);
    *reinterpret_cast<f32x4*>(wsm + row * LS + c4 * 4) = *reinterpret_cast<const f32x4*>(p.w + (size_t)i * 4);
  }

  // ---- this lane's patch slots: piece j of wave w covers slots (j * 4 + w) * 64 .. + 63 (1 KiB of LDS per instruction) ----
  int s_py[NP], s_px[NP], s_c[NP];     // patch pixel (row, column) and channel offset of the slot; s_c < 0: pad slot / past the end
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int s = (j * 4 + wave) * 64 + lane;
    const int pp = s / C::SLOTS, c = s - pp * C::SLOTS;
    s_py[j] = pp / PW;
    s_px[j] = pp - s_py[j] * PW;
    s_c[j] = (s < C::NS && c < C::SLOTS - 1) ? c * 4 : -1;
  }
  const unsigned long long zero_addr = (unsigned long long)p.zeros;
  auto tile_origin = [&](int t, int* b, int* oy0, int* ox0) __attribute__((always_inline)) {
    const int per_img = tiles_x * tiles_y;
    *b = t / per_img;
    const int r = t - *b * per_img;
    const int ty = r / tiles_x;
    *oy0 = ty * C::TH;
    *ox0 = (r - ty * tiles_x) * C::TW;
  };
  auto request = [&](int t, float* buf) __attribute__((always_inline)) {
    int b, oy0, ox0;
    tile_origin(t, &b, &oy0, &ox0);
    const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
    const float* img = p.x + (size_t)b * p.H * p.W * CIN;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int iy = iy0 + s_py[j], ix = ix0 + s_px[j];
      const bool ok = s_c[j] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned long long a = (unsigned long long)(img + ((size_t)iy * p.W + ix) * CIN + s_c[j]);
      const unsigned long long m = ok ? ~0ull : 0ull;
      __builtin_amdgcn_global_load_lds((gptr_t)((a & m) | (zero_addr & ~m)), (lptr_t)(buf + (j * 4 + wave) * 256), 16, 0, 0);
    }
  };

  // ---- MFMA fragment coordinates: row li of the wave's 32-pixel block = tile pixel (2 * wave + li / 16, li % 16) ----
  const int trow = 2 * wave + (li >> 4), tcol = li & 15;
  const int a_base = ((trow * STRIDE) * PW + tcol * STRIDE) * LS + hi * 4;
  const int b_base = li * LS + hi * 4;

  f32x16 acc[NI], prev[NI];
#pragma unroll
  for (int u = 0; u < NI; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[u][r] = 0.f; prev[u][r] = 0.f; }
  bool prev_valid = false;
  int prev_b = 0, prev_oy0 = 0, prev_ox0 = 0;

  // scale / shift of this lane's columns (gate: cout == BN)
  float sc[NI], sh[NI];
#pragma unroll
  for (int u = 0; u < NI; ++u) { sc[u] = p.scale[u * 32 + li] * p.alpha; sh[u] = p.shift[u * 32 + li]; }
  const bool relu = p.relu != 0;

  auto store_prev = [&]() __attribute__((always_inline)) {
    // accumulator register r = block row (r & 3) + 8 * (r >> 2) + 4 * hi -> tile pixel; a wave instruction writes two pixels
    // x 32 consecutive channels = two whole 128-byte lines
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int oy = prev_oy0 + 2 * wave + (row >> 4), ox = prev_ox0 + (row & 15);
      if (oy < p.Ho && ox < p.Wo) {
        float* o = p.y + (((size_t)prev_b * p.Ho + oy) * p.Wo + ox) * BN + li;
#pragma unroll
        for (int u = 0; u < NI; ++u) {
          float v = prev[u][r] * sc[u] + sh[u];
          if (relu) v = relu_keep_nan(v);
          o[u * 32] = v;
        }
      }
    }
  };

  request(tile, smem);
  int cur = 0;
  for (; tile < n_tiles; tile += G) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this tile's patch (and stores issued two thirds of a tile ago)
    __syncthreads();                                      // every wave's pieces have landed; the other buffer is free
    const float* const patch = smem + cur * C::PATCH_FLOATS;
    if (tile + G < n_tiles) request(tile + G, smem + (cur ^ 1) * C::PATCH_FLOATS);
    __builtin_amdgcn_sched_barrier(0);

    f32x4 af[2][NJ], bf[2][NJ][NI];
#define PATCH_READ(set, tap)                                                                                          \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                                    \
    af[set][j] = *reinterpret_cast<const f32x4*>(patch + a_base + (((tap) / 3) * PW + (tap) % 3) * LS + j * 8);        \
    _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                                    \
      bf[set][j][u] = *reinterpret_cast<const f32x4*>(wsm + b_base + ((tap) * BN + u * 32) * LS + j * 8);             \
  }
#define PATCH_MFMA(set)                                                                                               \
  _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                                      \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                                  \
      _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                                  \
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][j][kk], bf[set][j][u][kk], acc[u], 0, 0, 0);
    PATCH_READ(0, 0)
    static_for<9>([&](auto T) __attribute__((always_inline)) {
      constexpr int tap = decltype(T)::value;
      if constexpr (tap + 1 < 9) { PATCH_READ((tap + 1) & 1, tap + 1) }
      __builtin_amdgcn_sched_barrier(0);
      PATCH_MFMA(tap & 1)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (tap == 2) {
        if (prev_valid) store_prev();
        __builtin_amdgcn_sched_barrier(0);
      }
    });
#undef PATCH_READ
#undef PATCH_MFMA
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      prev[u] = acc[u];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    }
    tile_origin(tile, &prev_b, &prev_oy0, &prev_ox0);
    prev_valid = true;
    cur ^= 1;
  }
  store_prev();
}

template <int CIN, int BN, int STRIDE>
int launch_patch_t(const ConvKParams& p, int B, int cus, hipStream_t stream) {
  using C = PatchCfg<CIN, BN, STRIDE>;
  const int tiles_x = (p.Wo + C::TW - 1) / C::TW, tiles_y = (p.Ho + C::TH - 1) / C::TH;
  const long long T = (long long)B * tiles_x * tiles_y;
  const int G = (int)std::min<long long>(T, cus);
  hipLaunchKernelGGL((conv_patch_kernel<CIN, BN, STRIDE>), dim3((unsigned)G), dim3(256), 0, stream, p, tiles_x, tiles_y, (int)T);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("conv_patch launch: ") + hipGetErrorString(e));
}

}  // namespace

// Which layers take it: 3 x 3, pad 1, no dilation, stride 1 or 2, ONE source of 16 or 32 (padded) channels packed with
// bk = cin (k-tile = tap), 32 or 64 output channels in one n-tile, no residual, no weight groups; at least patch_mintiles
// output tiles (below that a tile-per-workgroup launch of conv_igemm spreads better: every workgroup here stages all weights).
bool conv_patch_eligible(const ConvDesc& d, const ConvArgs& a) {
  const long long min_tiles = opt(OPT_PATCH_MINTILES);
  if (min_tiles <= 0) return false;
  if (d.kh != 3 || d.kw != 3 || d.pad != 1 || d.dil != 1 || (d.stride != 1 && d.stride != 2)) return false;
  if (a.c2 != 0 || a.res != nullptr || a.mt_per_group != 0) return false;
  if (!((d.cin == 32 && d.bk == 32) || (d.cin == 16 && d.bk == 16))) return false;
  if (d.cout != d.bn_tile || d.cout_pad != d.cout || (d.cout != 32 && d.cout != 64)) return false;
  if (d.cin == 16 && d.cout != 32) return false;
  if (d.cin == 32 && d.stride != 1) return false;          // a stride-2 patch of 32 channels (2 x 82 KB) does not fit LDS
  const long long tiles = (long long)a.B * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
  return tiles >= min_tiles && tiles < 0x7fffffffLL;
}

int launch_conv_patch(const ConvKParams& p, const ConvDesc& d, int B, hipStream_t stream) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
    return fail(-3, "conv_patch: no current device");
  if (d.cin == 32 && d.cout == 64 && d.stride == 1) { note_kernel("conv_patch_32x64s1"); return launch_patch_t<32, 64, 1>(p, B, cus, stream); }
  if (d.cin == 32 && d.cout == 32 && d.stride == 1) { note_kernel("conv_patch_32x32s1"); return launch_patch_t<32, 32, 1>(p, B, cus, stream); }
  if (d.cin == 16 && d.cout == 32 && d.stride == 1) { note_kernel("conv_patch_16x32s1"); return launch_patch_t<16, 32, 1>(p, B, cus, stream); }
  if (d.cin == 16 && d.cout == 32 && d.stride == 2) { note_kernel("conv_patch_16x32s2"); return launch_patch_t<16, 32, 2>(p, B, cus, stream); }
  return fail(-2, "conv_patch: unsupported configuration");
}

}  // namespace peanut
