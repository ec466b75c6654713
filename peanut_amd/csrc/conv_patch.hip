// conv_patch_kernel: the 3x3 convs with FEW input channels (16 / 32: the deep stem, prediction/mmseg/models/backbones/
// resnet.py:591-624) as a persistent im2col-free kernel with LDS-staged input patches.
//
// conv_igemm.hip runs these layers as a k-tile-per-tap implicit GEMM: every 128-pixel tile fetches its nine shifted
// views of the input from L2 (9 x the activation bytes through the vector L1) and the WHOLE weight tensor again (73 KB per
// 128 pixels for the 32 -> 64 layer: more bytes than the activations), behind nine barriers; 87-95 TF/s.  Here
//   * workgroups are persistent and walk the output tiles (TH rows x 16 columns of output pixels, every n);
//   * the weights of all nine taps are put into LDS ONCE per workgroup;
//   * the input patch of a tile (((TH - 1) s + 3) x (15 s + 3) pixels for stride s, all channels) is fetched ONCE: by LDS-DMA
//     (global_load_lds_dwordx4) from an NHWC tensor, or -- NCHW variant, the network's input layer: the separate layout
//     pass disappears -- through registers from the channel planes (loads coalesced along x, requested a tile ahead, written
//     to LDS channel-interleaved).  Out-of-image pixels come from the zero page (address select, no predication);
//   * a tap is a constant LDS offset: the nine taps read shifted windows of the same patch -- one barrier per TILE (two with
//     a single patch buffer);
//   * the epilogue (scale / shift / ReLU, 128-byte row segments straight from the accumulator layout) of tile i is issued
//     from a second accumulator set during tile i + 1, two rows after each tap: no store burst for the in-order issue of a
//     wave to stall on, and the loop's vmcnt(0) never waits for a store.
// Wave = one 32-pixel MFMA row block (2 rows x 16 columns of the tile) x 32 * NI output channels; a workgroup has
// (TH / 2) x NWN waves (NWN: waves side by side in n).  Eight waves (two per SIMD) wherever LDS allows: with one wave per SIMD
// every store / LDS-DMA issue stall idles the matrix pipe (profiles/r5r: 115 TF/s with four waves and the stores in one burst).
// LDS rows are padded by one 16-byte slot (row = CIN + 4 floats), as in conv_igemm: conflict-free ds_read_b128 for 8 / 16
// consecutive pixels.  Same fragment layout and k order (tap outer; 8-channel groups; lanes 0-31 / 32-63 take channels
// 0-3 / 4-7 of a group) as conv_igemm_kernel, and exact fp32 products -- but NOT the same summation order: the weight fragments
// of all nine taps live in REGISTERS (bw[9][NJ][NI], not LDS) and even / odd k-steps accumulate in two chains (acc0 / acc1) that
// are added at the end, so the result differs from conv_igemm's single chain in the last bits (~1e-6 relative; the tests hold the
// two kernels to 1e-5 of each other, and the fp32 stem's last bits therefore depend on patch_mintiles / stem_nchw).
#include <stdlib.h>

#include "common.h"
#include "conv_common.h"

namespace peanut {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int CIN, int BN, int STRIDE, int TH, int NWN, int DBUF>
struct PatchCfg {
  static constexpr int TW = 16;
  static constexpr int NW = (TH / 2) * NWN, NT = 64 * NW;                 // waves, threads
  static constexpr int PH = (TH - 1) * STRIDE + 3, PW = (TW - 1) * STRIDE + 3;
  static constexpr int LS = CIN + 4, SLOTS = LS / 4;                      // floats / 16-byte slots per patch pixel (one pad slot)
  // patch row pitch: a multiple of 64 floats, so that the second row of a wave's 2 x 16-pixel block starts on the bank the 17th
  // pixel of a straight run would (16 * LS = 0 mod 64): the 16-lane groups of a ds_read_b128 then see 16 distinct bank quads
  static constexpr int RP = (PW * LS + 63) / 64 * 64, RSLOTS = RP / 4;
  static constexpr int NPX = PH * PW;                                     // pixels of a patch
  static constexpr int NS = PH * RSLOTS;                                  // 16-byte slots of a patch (pitch padding included)
  static constexpr int NP = (NS + NT - 1) / NT;                           // LDS-DMA instructions per wave and patch
  static constexpr int PATCH_FLOATS = NP * NT * 4;
  static constexpr int NI = BN / 32 / NWN, NJ = CIN / 8;
  static constexpr int JH = 2, HS = NJ / JH, STEPS = 9 * HS;              // a step = JH 8-channel groups of one tap
  static constexpr int NPK = (NPX + NT - 1) / NT;                         // NCHW variant: patch pixels per thread
  static constexpr int SMEM_FLOATS = DBUF * PATCH_FLOATS;
};

// NCHW: p.x is the network input [B][creal][H][W] (creal <= 16 real channels); else NHWC [B][H][W][CIN]
template <int CIN, int BN, int STRIDE, int TH, int NWN, int DBUF, bool NCHW>
__global__ __launch_bounds__(64 * (TH / 2) * NWN) void conv_patch_kernel(const ConvKParams p, int tiles_x, int tiles_y, int n_tiles, int creal) {
  using C = PatchCfg<CIN, BN, STRIDE, TH, NWN, DBUF>;
  constexpr int LS = C::LS, PW = C::PW, PH = C::PH, RP = C::RP, NI = C::NI, NJ = C::NJ, NP = C::NP, NW = C::NW, NT = C::NT, NPK = C::NPK;
  constexpr int JH = C::JH, HS = C::HS, STEPS = C::STEPS;
  static_assert(!NCHW || CIN == 16, "the NCHW variant stages 16 channels per pixel");
  static_assert(DBUF == 2 || DBUF == 1, "patch buffers");
  static_assert(NJ % JH == 0 && 16 % STEPS != 1 && STEPS >= 8, "step layout");
  __shared__ __attribute__((aligned(1024))) float smem[C::SMEM_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave / NWN, wn = wave - wp * NWN;            // pixel group (tile rows 2 wp, 2 wp + 1), n part
  const int li = lane & 31, hi = lane >> 5;
  const int G = gridDim.x, bid = blockIdx.x;
  // workgroup b runs on XCD b % 8 (observed; used for speed only): the G / 8 workgroups of an XCD take consecutive tiles at every
  // step, so neighbouring tiles' halos meet in one L2
  const int lw = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
  int tile = lw;
  if (tile >= n_tiles) return;

  // ---- this lane's weight fragments, ALL taps, in registers for the life of the workgroup: packed [tap][BN][CIN]
  //      (pack_conv_weights with bk = CIN: k-tile = tap); lane (li, hi) holds channels 8 j + 4 hi .. + 3 of output row li ----
  f32x4 bw[9][NJ][NI];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int u = 0; u < NI; ++u)
        bw[t][j][u] = *reinterpret_cast<const f32x4*>(p.w + ((size_t)(t * BN + (wn * NI + u) * 32 + li)) * CIN + j * 8 + hi * 4);

  const unsigned long long zero_addr = (unsigned long long)p.zeros;
  auto tile_origin = [&](int t, int* b, int* oy0, int* ox0) __attribute__((always_inline)) {
    const int per_img = tiles_x * tiles_y;
    *b = t / per_img;
    const int r = t - *b * per_img;
    const int ty = r / tiles_x;
    *oy0 = ty * TH;
    *ox0 = (r - ty * tiles_x) * C::TW;
  };

  // ---- NHWC: this lane's patch slots; piece j of wave w covers slots (j * NW + w) * 64 .. + 63 (1 KiB of LDS per instruction).
  //      s_pos = patch row << 16 | patch column, or -1 for a pad slot; s_off = the slot's float offset from the patch origin ----
  int s_pos[NP], s_off[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int s = (j * NW + wave) * 64 + lane;
    const int py = s / C::RSLOTS, rem = s - py * C::RSLOTS;
    const int px = rem / C::SLOTS, c = rem - px * C::SLOTS;
    const bool ok = py < PH && px < PW && c < C::SLOTS - 1;
    s_pos[j] = ok ? (py << 16 | px) : -1;
    s_off[j] = (py * p.W + px) * CIN + c * 4;
  }
  auto request = [&](int b, int oy0, int ox0, float* buf) __attribute__((always_inline)) {
    const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
    const float* origin = p.x + ((size_t)b * p.H * p.W + (long long)iy0 * p.W + ix0) * CIN;      // may lie before the image: used with in-range offsets only
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + PH <= p.H && ix0 + PW <= p.W;              // uniform
    if (interior) {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const unsigned long long a = (unsigned long long)(origin + s_off[j]);
        const unsigned long long m = s_pos[j] >= 0 ? ~0ull : 0ull;
        __builtin_amdgcn_global_load_lds((gptr_t)((a & m) | (zero_addr & ~m)), (lptr_t)(buf + (j * NW + wave) * 256), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int iy = iy0 + (s_pos[j] >> 16), ix = ix0 + (s_pos[j] & 0xffff);
        const bool ok = s_pos[j] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const unsigned long long a = (unsigned long long)(origin + s_off[j]);
        const unsigned long long m = ok ? ~0ull : 0ull;
        __builtin_amdgcn_global_load_lds((gptr_t)((a & m) | (zero_addr & ~m)), (lptr_t)(buf + (j * NW + wave) * 256), 16, 0, 0);
      }
    }
  };

  // ---- NCHW: thread t stages patch pixels t, t + NT, ...: sixteen channel values each, one load per plane ----
  float stg[NCHW ? NPK : 1][16];
  auto load_regs = [&](int b, int oy0, int ox0) __attribute__((always_inline)) {
    const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
    const size_t plane = (size_t)p.H * p.W;
    const float* img = p.x + (size_t)b * creal * plane;
#pragma unroll
    for (int k = 0; k < (NCHW ? NPK : 1); ++k) {
      const int pp = tid + NT * k;
      const int py = pp / PW, px = pp - py * PW;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = pp < C::NPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const float* a = img + (size_t)iy * p.W + ix;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const unsigned long long m = (ok && c < creal) ? ~0ull : 0ull;
        const unsigned long long addr = ((unsigned long long)(a + (size_t)c * plane) & m) | (zero_addr & ~m);
        stg[k][c] = *reinterpret_cast<const __attribute__((address_space(1))) float*>(addr);
      }
    }
  };
  auto write_regs = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < (NCHW ? NPK : 1); ++k) {
      const int pp = tid + NT * k;
      const int py = pp / PW, px = pp - py * PW;
      if (pp < C::NPX) {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
          *reinterpret_cast<f32x4*>(buf + py * RP + px * LS + c4 * 4) = f32x4{stg[k][c4 * 4], stg[k][c4 * 4 + 1], stg[k][c4 * 4 + 2], stg[k][c4 * 4 + 3]};
      }
    }
  };

  // ---- MFMA fragment coordinates: row li of the wave's 32-pixel block = tile pixel (2 * wp + li / 16, li % 16) ----
  const int trow = 2 * wp + (li >> 4), tcol = li & 15;
  const int a_base = (trow * STRIDE) * RP + (tcol * STRIDE) * LS + hi * 4;

  // two accumulator chains per output block (even / odd k-steps of a group): consecutive MFMAs never depend on each other, so an
  // LDS read or a store issued between them costs an issue slot, not the dependent-accumulator latency
  f32x16 acc0[NI], acc1[NI], prev[NI];
#pragma unroll
  for (int u = 0; u < NI; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[u][r] = 0.f; acc1[u][r] = 0.f; prev[u][r] = 0.f; }
  bool prev_valid = false, prev_full = false;
  int prev_oy0 = 0, prev_ox0 = 0;
  float* prev_out = p.y;         // this lane's first output element of the previous tile: pixel (oy0 + 2 wp, ox0 + 4 hi), its first column

  // scale / shift of this lane's columns (gate: cout == BN)
  float sc[NI], sh[NI];
#pragma unroll
  for (int u = 0; u < NI; ++u) { sc[u] = p.scale[(wn * NI + u) * 32 + li] * p.alpha; sh[u] = p.shift[(wn * NI + u) * 32 + li]; }
  const bool relu = p.relu != 0;
  const int row_pitch = p.Wo * BN;

  // accumulator register r = block row (r & 3) + 8 * (r >> 2) + 4 * hi -> tile pixel (2 wp + (r >> 3), (r & 3) + 8 * ((r >> 2) & 1) + 4 hi);
  // a wave instruction writes two pixels x 32 consecutive channels = two whole 128-byte lines
  auto store_prev_row = [&](int r) __attribute__((always_inline)) {
    const int rr = r >> 3, cc = (r & 3) + 8 * ((r >> 2) & 1);
    float* o = prev_out + (rr ? row_pitch : 0) + cc * BN;
    const bool ok = prev_full || (prev_oy0 + 2 * wp + rr < p.Ho && prev_ox0 + 4 * hi + cc < p.Wo);
    if (ok) {
#pragma unroll
      for (int u = 0; u < NI; ++u) {
        float v = prev[u][r] * sc[u] + sh[u];
        if (relu) v = relu_keep_nan(v);
        o[u * 32] = v;
      }
    }
  };

  int cb, coy, cox;                       // origin of the current tile
  tile_origin(tile, &cb, &coy, &cox);
  if constexpr (NCHW) load_regs(cb, coy, cox);
  else if constexpr (DBUF == 2) request(cb, coy, cox, smem);
  int cur = 0;
  for (; tile < n_tiles; tile += G) {
    const bool more = tile + G < n_tiles;
    int nb = 0, noy = 0, nox = 0;
    if (more) tile_origin(tile + G, &nb, &noy, &nox);
    float* const patch_w = smem + cur * C::PATCH_FLOATS;
    if constexpr (DBUF == 2 && !NCHW) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's patch (the stores of the previous tile are long gone)
      __syncthreads();                                    // every wave's pieces have landed; the other buffer is free
    } else {
      if constexpr (DBUF == 1) __syncthreads();           // everybody is done with the previous tile's patch
      if constexpr (NCHW) {
        write_regs(patch_w);
        if (more) load_regs(nb, noy, nox);
      } else {
        request(cb, coy, cox, patch_w);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
    const float* const patch = patch_w;

    f32x4 af[2][JH];
#define PATCH_READ(set, step)                                                                                         \
  _Pragma("unroll") for (int jj = 0; jj < JH; ++jj)                                                                   \
    af[set][jj] = *reinterpret_cast<const f32x4*>(patch + a_base + (((step) / HS) / 3) * RP + (((step) / HS) % 3) * LS + (((step) % HS) * JH + jj) * 8);
#define PATCH_MFMA(set, step)                                                                                         \
  _Pragma("unroll") for (int jj = 0; jj < JH; ++jj)                                                                   \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                                  \
      _Pragma("unroll") for (int u = 0; u < NI; ++u) {                                                                \
        if ((kk & 1) == 0) acc0[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][jj][kk], bw[(step) / HS][((step) % HS) * JH + jj][u][kk], acc0[u], 0, 0, 0); \
        else acc1[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][jj][kk], bw[(step) / HS][((step) % HS) * JH + jj][u][kk], acc1[u], 0, 0, 0); \
      }
    PATCH_READ(0, 0)
    static_for<STEPS>([&](auto T) __attribute__((always_inline)) {
      constexpr int step = decltype(T)::value;
      if constexpr (step + 1 < STEPS) { PATCH_READ((step + 1) & 1, step + 1) }
      if constexpr (step == 0 && DBUF == 2 && !NCHW) {
        if (more) request(nb, noy, nox, smem + (cur ^ 1) * C::PATCH_FLOATS);   // under the first MFMAs, not ahead of them
      }
      __builtin_amdgcn_sched_barrier(0);
      PATCH_MFMA(step & 1, step)
      __builtin_amdgcn_sched_barrier(0);
      // the previous tile's 16 accumulator rows leave spread over the steps
      constexpr int R0 = step * 16 / STEPS, R1 = (step + 1) * 16 / STEPS;
      if constexpr (R1 > R0) {
        if (prev_valid) {
#pragma unroll
          for (int r = R0; r < R1; ++r) store_prev_row(r);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    });
#undef PATCH_READ
#undef PATCH_MFMA
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      prev[u] = acc0[u] + acc1[u];
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[u][r] = 0.f; acc1[u][r] = 0.f; }
    }
    prev_oy0 = coy; prev_ox0 = cox;
    prev_full = coy + TH <= p.Ho && cox + C::TW <= p.Wo;
    prev_out = p.y + (((size_t)cb * p.Ho + coy + 2 * wp) * p.Wo + cox + 4 * hi) * BN + wn * NI * 32 + li;
    prev_valid = true;
    cb = nb; coy = noy; cox = nox;
    if constexpr (DBUF == 2) cur ^= 1;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) store_prev_row(r);
}

template <int CIN, int BN, int STRIDE, int TH, int NWN, int DBUF, bool NCHW>
int launch_patch_t(const ConvKParams& p, int B, int cus, int per_cu, int creal, hipStream_t stream) {
  using C = PatchCfg<CIN, BN, STRIDE, TH, NWN, DBUF>;
  const int tiles_x = (p.Wo + C::TW - 1) / C::TW, tiles_y = (p.Ho + TH - 1) / TH;
  const long long T = (long long)B * tiles_x * tiles_y;
  const int G = (int)std::min<long long>(T, (long long)cus * per_cu);
  hipLaunchKernelGGL((conv_patch_kernel<CIN, BN, STRIDE, TH, NWN, DBUF, NCHW>), dim3((unsigned)G), dim3(C::NT), 0, stream, p, tiles_x, tiles_y,
                     (int)T, creal);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("conv_patch launch: ") + hipGetErrorString(e));
}

int device_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return cus;
}

// Which layers take it: 3 x 3, pad 1, no dilation, ONE source of 16 or 32 (padded) channels packed with bk = cin (k-tile =
// tap), 32 or 64 output channels in one n-tile, no residual, no weight groups; stride 1, or 2 with 16 channels (a stride-2
// patch of 32 channels does not fit LDS); at least patch_mintiles output tiles of 8 x 16 pixels (below that a
// tile-per-workgroup launch of conv_igemm spreads better: every workgroup here stages all the weights).
bool patch_layer_ok(const ConvDesc& d) {
  if (d.kh != 3 || d.kw != 3 || d.pad != 1 || d.dil != 1 || (d.stride != 1 && d.stride != 2)) return false;
  if (d.rs != 0) return false;
  if (!((d.cin == 32 && d.bk == 32) || (d.cin == 16 && d.bk == 16))) return false;
  if (d.cout != d.bn_tile || d.cout_pad != d.cout || (d.cout != 32 && d.cout != 64)) return false;
  if (d.cin == 16 && d.cout != 32) return false;
  if (d.cin == 32 && d.stride != 1) return false;
  return true;
}

}  // namespace

bool conv_patch_eligible(const ConvDesc& d, const ConvArgs& a) {
  const long long min_tiles = opt(OPT_PATCH_MINTILES);
  if (min_tiles <= 0 || !patch_layer_ok(d)) return false;
  if (a.c2 != 0 || a.res != nullptr || a.mt_per_group != 0) return false;
  const long long tiles = (long long)a.B * ((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
  return tiles >= min_tiles && tiles < 0x7fffffffLL;
}

// family name launch_conv_patch / launch_conv_patch_nchw note for a layer the gates admit (the planner's op tables)
const char* conv_patch_kernel_name(const ConvDesc& d, bool nchw) {
  if (nchw) return "conv_patch_nchw_16x32s2";
  if (d.cin == 32) return d.cout == 64 ? "conv_patch_32x64s1" : "conv_patch_32x32s1";
  return d.stride == 1 ? "conv_patch_16x32s1" : "conv_patch_16x32s2";
}

// the network's first conv straight from the NCHW input (option stem_nchw): 16-channel stride-2 layer, <= 16 real channels
bool conv_patch_nchw_eligible(const ConvDesc& d, int B, int Ho, int Wo, int creal) {
  const long long min_tiles = opt(OPT_PATCH_MINTILES);
  if (min_tiles <= 0 || opt(OPT_STEM_NCHW) == 0 || !patch_layer_ok(d)) return false;
  if (d.cin != 16 || d.stride != 2 || creal < 1 || creal > 16) return false;
  const long long tiles = (long long)B * ((Ho + 7) / 8) * ((Wo + 15) / 16);
  return tiles >= min_tiles && tiles < 0x7fffffffLL;
}

int launch_conv_patch(const ConvKParams& p, const ConvDesc& d, int B, hipStream_t stream) {
  const int cus = device_cus();
  if (cus < 1) return fail(-3, "conv_patch: no current device");
  note_kernel(conv_patch_kernel_name(d, false));
  if (d.cin == 32 && d.cout == 64) return launch_patch_t<32, 64, 1, 8, 2, 2, false>(p, B, cus, 1, 0, stream);
  if (d.cin == 32 && d.cout == 32) return launch_patch_t<32, 32, 1, 16, 1, 2, false>(p, B, cus, 1, 0, stream);
  if (d.cin == 16 && d.stride == 1) return launch_patch_t<16, 32, 1, 16, 1, 2, false>(p, B, cus, 1, 0, stream);
  if (d.cin == 16 && d.stride == 2) return launch_patch_t<16, 32, 2, 8, 1, 1, false>(p, B, cus, 2, 0, stream);
  return fail(-2, "conv_patch: unsupported configuration");
}

// x: [B][creal][H][W] fp32 (the network input as the caller hands it over); y: NHWC [B][Ho][Wo][32]
int launch_conv_patch_nchw(const ConvDesc& d, const float* x_nchw, int creal, float* y, int B, int H, int W, int Ho, int Wo, hipStream_t stream) {
  const int cus = device_cus();
  if (cus < 1) return fail(-3, "conv_patch: no current device");
  ConvKParams p{};
  p.x = x_nchw; p.x2 = x_nchw; p.w = d.w_packed; p.scale = d.scale; p.shift = d.shift; p.res = nullptr; p.y = y;
  p.zeros = zero_page();
  if (!p.zeros) return fail(-3, "conv_patch: zero page allocation failed");
  p.H = H; p.W = W; p.c1 = d.cin; p.c2 = 0; p.Ho = Ho; p.Wo = Wo; p.cout = d.cout;
  p.kw = 3; p.ntaps = 9; p.stride = d.stride; p.pad = 1; p.dil = 1; p.relu = d.relu;
  p.HoWo = Ho * Wo; p.M = B * Ho * Wo; p.alpha = 1.f;
  note_kernel(conv_patch_kernel_name(d, true));
  return launch_patch_t<16, 32, 2, 8, 1, 1, true>(p, B, cus, 2, creal, stream);
}

}  // namespace peanut
