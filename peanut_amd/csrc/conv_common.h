// Shared pieces of the implicit-GEMM conv kernels (fp32-MFMA and split-precision variants).
#pragma once
#include <stdlib.h>

#include <atomic>

#include "common.h"
#include "options.h"

namespace peanut {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: no struct memcpy, stays in VGPRs

// ReLU that lets NaN through, like torch.relu (fmaxf returns the other operand): an overflow of the fp16-piece mode, or
// a NaN in the input, must reach the output instead of turning into a plausible zero
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ f32x4 relu_keep_nan(f32x4 v) {
  return f32x4{relu_keep_nan(v.x), relu_keep_nan(v.y), relu_keep_nan(v.z), relu_keep_nan(v.w)};
}

struct ConvKParams {
  const float* x;
  const float* x2;
  const float* w;
  const float* scale;
  const float* shift;
  const float* res;
  const float* zeros;   // >= 16 bytes of zeros on the device (target of out-of-range loads)
  float* y;
  int H, W, c1, c2, Ho, Wo, cout;
  int kw, ntaps, stride, pad, dil, relu;
  int M, nkt, ntiles, HoWo;
  // tail split-K (see decode_work): the last `n_sp / split_p` tiles are cut into split_p k-ranges each
  int n_full, n_sp, split_p;
  float* partial;   // [n_sp][BM*BN] raw accumulator tiles of the split parts
  // grouped GEMM (Winograd positions): m-tile mt uses the weight block (mt / mt_per_group); 0 = one block
  int mt_per_group;
  long long w_group_stride;   // floats between consecutive weight blocks
  int ss_group_stride;        // floats between the groups' scale (and shift) blocks; 0 = shared
  int mtiles;                 // m-tiles of the launch (decode_work's n-chunked tile order)
  int nchunk;                 // n-tiles per chunk of that order; 0 = plain order (n fastest over all n-tiles)
  int res_prefetch;           // epilogue: request the first rows of the residual before the accumulators go through LDS
  float alpha;                // y = relu(alpha * acc * scale + shift + res): 1 / the pack scale of fp16-piece weights, else 1
  int ares_pbn;               // conv_pw_ares_kernel: the n-tile the weights were packed with (64 / 128)
  int phase_shift;            // conv_pw_glds256_kernel: waves 4-7 request their LDS-DMA pieces half an iteration after waves 0-3
  // conv_pw_glds256p_kernel, stream-K tail (round 4): the tail's sk_units k-tiles (tail tiles x nkt) are dealt out as ONE stream of
  // equal contiguous runs to the first sk_g workgroups; a run may end one tile and begin the next, a tile has at most sk_maxp
  // fragments (raw partial tiles j * sk_maxp + fragment index).  sk_units = 0: the uniform split above (split_p parts per tile)
  // sk_q: k-tiles per unit of the stream (2 for the two-level accumulation: fragments then start and end on a 64-channel boundary and
  // are never shorter than the two iterations its register epilogue needs)
  int sk_units, sk_maxp, sk_g, sk_q;
  int pack_bn;                // conv_pw_glds_kernel: rows of a PACKED weight tile when wider than the kernel's n-tile (128 for the 64-wide kernel on 128-wide packing); 0 = the kernel's
  int group_valid;            // grouped GEMM: rows of every weight group that hold data (the rest of the group's rows is padding); 0 = all
  int p_order;                // conv_pw_glds256p_kernel: item order (option pw256p_order)
  float* dump;                // conv_pw_glds256wp_kernel: one 256 x 256 tile of scratch behind the partial tiles (target of a workgroup's first, empty epilogue)
  int stagger;                // conv_pw_glds256wp_kernel: spread of the workgroups' start times, in sleeps of ~3.4 us (option pw256wp_stagger)
  int flush;                  // k-tiles per partial sum of the two-level fp32 accumulation (0: one running sum); see PEANUT_FLUSH_*
  DeferredSplit* defer;       // HOST pointer, never read on the device: launch_with_tail_split may skip its reduce and fill it (common.h)
  const int* group_rows;      // HOST pointer, never read on the device: data rows of each weight group (ConvArgs::group_rows)
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


const float* zero_page();   // per-device 4 KiB of zeros (allocated on first use)
// conv_pw.hip: fp32 pointwise convs / grouped GEMMs with LDS-DMA staging (PEANUT_PW_GLDS=0 disables)
bool conv_pw_enabled();
int launch_conv_pw(const ConvKParams& p, int bn_tile, float* ws, size_t ws_floats, hipStream_t stream);
// conv_pw_ares.hip: persistent A-resident kernel for the K = 128 / 256 pointwise layers and grouped GEMMs
int launch_conv_pw_ares(const ConvKParams& p, int bn_tile, hipStream_t stream);
// conv_pw256p.hip: persistent 256 x 128 kernel (p.ntiles = 128-wide n-tiles)
int launch_conv_pw256p(const ConvKParams& p, float* ws, size_t ws_floats, hipStream_t stream);
// conv_pw256wp.hip: persistent 256 x 256 kernel (returns 1 without launching when the tail's partial tiles do not fit the scratch)
int launch_conv_pw256wp(const ConvKParams& p, float* ws, size_t ws_floats, hipStream_t stream);
// conv_patch.hip: persistent LDS-patch kernel for the 3x3 convs with 16 / 32 input channels (gate: conv_patch_eligible)
int launch_conv_patch(const ConvKParams& p, const ConvDesc& d, int B, hipStream_t stream);
// gemm_rs.hip: pointwise layer / grouped GEMM emulated on the bf16 matrix cores: fp32 A (p.x, p.x2) split into bf16 pieces in
// registers, p.w = the weights' pre-split pieces, nkt = cin / 16
int launch_gemm_rs(const ConvKParams& p, int bn_tile, int planes, float* ws, size_t ws_floats, hipStream_t stream);
// conv_rs.hip: every other conv in the emulated-fp32 modes (implicit GEMM, p.w = pack_weights_sx_conv weights, nkt = cin / 16 * taps)
int launch_conv_rs(const ConvKParams& p, int bn_tile, int planes, float* ws, size_t ws_floats, hipStream_t stream);


// Work decomposition of one launch.
//  * XCD-aware (guide T1): workgroup b runs on XCD b % 8 (observed, used for speed only); every XCD gets
//    a contiguous run of logical work items so that its private L2 sees neighbouring tiles (n-tile
//    fastest -> co-resident workgroups share A panels).  Bijective for any grid size.
//  * Tail split-K: T tiles over S resident workgroup slots take ceil(T/S) "rounds"; when the last round
//    is mostly empty (e.g. 3600 tiles over 512 slots = 7.03 rounds) the t = T mod S tail tiles are each
//    cut into split_p k-ranges whose raw accumulators go to `partial` and are summed IN A FIXED ORDER
//    by conv_splitk_reduce_kernel (deterministic, no atomics).  The split parts are spread evenly over
//    the XCD runs (they are shorter than full tiles, so lumping them on one XCD would unbalance it).
struct Work { int mt, nt, kt0, kt1, item; };   // item >= 0: split part -> partial tile #item

// stream-K tail: run of workgroup w = units [U * w / G, U * (w + 1) / G); the workgroup whose run holds unit u
__host__ __device__ __forceinline__ int sk_owner(long long U, int G, long long u) {
  int w = (int)(u * G / U);
  while (w + 1 < G && U * (w + 1) / G <= u) ++w;
  while (w > 0 && U * w / G > u) --w;
  return w;
}

// tile -> (mt, nt).  Plain order: n fastest.  With more than `nchunk` n-tiles the n range is cut into chunks that are
// walked one after the other (all m-tiles of chunk 0, then chunk 1, ...): the 64 workgroups resident on an XCD then
// cover 8 m-tiles x 8 n-tiles instead of 4 x 16, i.e. 16 instead of 20 operand panels per round through its L2.
__device__ __forceinline__ void tile_to_mn(const ConvKParams& p, int tile, int* mt_out, int* nt_out) {
  int mt, nt;
  if (p.nchunk > 0 && p.ntiles > p.nchunk) {
    const int full = p.ntiles / p.nchunk, rem = p.ntiles - full * p.nchunk;
    const int per_chunk = p.mtiles * p.nchunk;
    if (tile < full * per_chunk) {
      const int c = tile / per_chunk, r = tile - c * per_chunk;
      mt = r / p.nchunk;
      nt = c * p.nchunk + (r - mt * p.nchunk);
    } else {
      const int r = tile - full * per_chunk;
      mt = r / rem;
      nt = full * p.nchunk + (r - mt * rem);
    }
  } else {
    mt = tile / p.ntiles;
    nt = tile - mt * p.ntiles;
  }
  *mt_out = mt;
  *nt_out = nt;
}

__device__ __forceinline__ Work decode_work(const ConvKParams& p) {
  const int G = gridDim.x, bid = blockIdx.x;
  const int q = G >> 3, r = G & 7, xcd = bid & 7, local = bid >> 3;
  const int sq = p.n_sp >> 3, sr = p.n_sp & 7;
  const int sp_x = sq + (xcd < sr), sp_before = xcd * sq + (xcd < sr ? xcd : sr);
  const int chunk_before = xcd * q + (xcd < r ? xcd : r);
  Work w;
  int tile;
  if (local < sp_x) {
    const int s = sp_before + local;
    const int j = s / p.split_p, part = s - j * p.split_p;
    tile = p.n_full + j;
    w.item = s;
    w.kt0 = (int)((long long)part * p.nkt / p.split_p);
    w.kt1 = (int)((long long)(part + 1) * p.nkt / p.split_p);
  } else {
    tile = (chunk_before - sp_before) + (local - sp_x);
    w.item = -1;
    w.kt0 = 0;
    w.kt1 = p.nkt;
  }
  tile_to_mn(p, tile, &w.mt, &w.nt);
  return w;
}

// ---- shared epilogue: y = relu(acc * scale[n] + shift[n] + res), or the raw partial tile of a split-K part ----
// The accumulators go through LDS (EP row slabs) so that global traffic is whole rows: each thread then handles
// 16-byte pieces (4 consecutive channels) -- a wave reads the residual and writes the output as contiguous row
// segments instead of 64 scalar accesses per lane.  The caller's k-loop must have retired every LDS read (barrier).
constexpr int kResPrefetchRows = 8;
struct ResPrefetch { f32x4 v[kResPrefetchRows]; bool on; };

// The first kResPrefetchRows residual rows a thread will add in conv_epilogue, requested early: a kernel may call this
// before its last k-tile's MFMAs (the loads then travel under them), conv_epilogue itself does it before the
// accumulators go through LDS otherwise.  Same thread -> (row, 4 channels) mapping as the epilogue's row loop.
template <int BM, int BN, int EP, int NT>
__device__ __forceinline__ ResPrefetch conv_res_prefetch(const ConvKParams& p, const Work& wk, int m0, int n0) {
  constexpr int NV = BN / 4, ROWS_PER_PASS = NT / NV, ER = BM / EP;
  const int tid = threadIdx.x;
  const int c4 = (tid % NV) * 4, r0 = tid / NV, n = n0 + c4;
  ResPrefetch r;
  r.on = p.res_prefetch && p.res && wk.item < 0 && (p.cout & 3) == 0 && n < p.cout;
  if (r.on) {
#pragma unroll
    for (int i = 0; i < kResPrefetchRows; ++i) {
      const int row = r0 + i * ROWS_PER_PASS, m = m0 + row;
      r.v[i] = (row < ER && m < p.M) ? *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.cout + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  return r;
}

template <int BM, int BN, int WM, int WN, int EP, int NT = 256, int MI = 0, int NI = 0>
__device__ __forceinline__ void conv_epilogue(const ConvKParams& p, const Work& wk, f32x16 (&acc)[MI][NI], float* smem,
                                              int m0, int n0, const ResPrefetch* early = nullptr) {
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int CS = BN + 4, ER = BM / EP;
  static_assert(MI == TM / 32 && NI == TN / 32, "accumulator shape");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  constexpr int NV = BN / 4;               // float4 per output row of the tile
  constexpr int ROWS_PER_PASS = NT / NV;   // rows covered by the NT threads per pass
  const int c4 = (tid % NV) * 4, r0 = tid / NV;
  const int n = n0 + c4;
  const int ss_off = (p.mt_per_group && p.ss_group_stride) ? (wk.mt / p.mt_per_group) * p.ss_group_stride : 0;
  const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + ss_off + n) * p.alpha;   // scale/shift are padded to cout_pad
  const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + ss_off + n);
  const bool vec_ok = (p.cout & 3) == 0;   // 16-byte aligned rows
  // the first PRE residual rows of this thread are requested early (conv_res_prefetch): their latency runs under the
  // last MFMAs / the LDS stores and the barrier instead of at the head of the row loop
  constexpr int PRE = kResPrefetchRows;
  const ResPrefetch rp = early ? *early : conv_res_prefetch<BM, BN, EP, NT>(p, wk, m0, n0);
  const f32x4 (&rpre)[PRE] = rp.v;
  const bool pre = rp.on;
#pragma unroll
  for (int ep = 0; ep < EP; ++ep) {
    if (EP == 1 || wm == ep) {
#pragma unroll
      for (int t = 0; t < MI; ++t)
#pragma unroll
        for (int u = 0; u < NI; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wm * TM + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi - ep * ER;
            smem[row * CS + wn * TN + u * 32 + li] = acc[t][u][r];
          }
    }
    __syncthreads();
    if (wk.item >= 0) {   // split-K part: raw accumulators, summed + finished by conv_splitk_reduce_kernel
      float* dst = p.partial + (size_t)wk.item * (BM * BN) + (size_t)ep * ER * BN;
      for (int row = r0; row < ER; row += ROWS_PER_PASS)
        *reinterpret_cast<f32x4*>(dst + row * BN + c4) = *reinterpret_cast<const f32x4*>(smem + row * CS + c4);
    } else {
      int row_first = r0;
      if (pre && ep == 0) {     // the rows whose residual is already on its way
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
          const int row = r0 + i * ROWS_PER_PASS, m = m0 + row;
          if (row < ER && m < p.M) {
            f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * CS + c4);
            v = v * sc + sh;
            v += rpre[i];
            if (p.relu) v = relu_keep_nan(v);
            const size_t o = (size_t)m * p.cout + n;
            *reinterpret_cast<f32x4*>(p.y + o) = v;
          }
        }
        row_first = r0 + PRE * ROWS_PER_PASS;
      }
#pragma unroll 4
      for (int row = row_first; row < ER; row += ROWS_PER_PASS) {
        const int m = m0 + ep * ER + row;
        if (m >= p.M) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * CS + c4);
        v = v * sc + sh;
        const size_t o = (size_t)m * p.cout + n;
        if (vec_ok) {
          if (n < p.cout) {
            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + o);
            if (p.relu) v = relu_keep_nan(v);
            *reinterpret_cast<f32x4*>(p.y + o) = v;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n + e < p.cout) {
              float x = v[e];
              if (p.res) x += p.res[o + e];
              if (p.relu) x = relu_keep_nan(x);
              p.y[o + e] = x;
            }
          }
        }
      }
    }
    if (ep + 1 < EP) __syncthreads();
  }
}

// Two-level fp32 accumulation (ConvKParams::flush = F > 0; conv_pw.hip): the matrix cores add a k-tile's 32 products per output onto
// one running sum, so the rounding error of a K-long contraction grows with the magnitude of that running sum (~ K in
// units of one product's spread).  Every F k-tiles the running sums are moved into a second accumulator set and restarted
// from zero: the partial sums stay small, and the error of a K = 2048 contraction drops about three-fold (fp32
// simulation of the Winograd position GEMMs, post-ReLU data: F(4x4) 4.1e-6 -> 0.95e-6 rms, F(6x6) 1.2e-5 -> 3.0e-6 with
// F = 2) -- which is what lets the larger-tile Winograd forms in where their A^T amplifies that error.  Cost per flush
// and wave: one v_add + one v_mov per accumulator register, against F * 64 MFMAs.
#define PEANUT_FLUSH_DECL()                                                        \
  f32x16 acc2[MI][NI];                                                             \
  _Pragma("unroll") for (int t = 0; t < MI; ++t)                                   \
    _Pragma("unroll") for (int u = 0; u < NI; ++u)                                 \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) acc2[t][u][r] = 0.f;          \
  int flush_left = p.flush > 0 ? p.flush : 0x7fffffff;
#define PEANUT_FLUSH_STEP()                                                        \
  if (--flush_left == 0) {                                                         \
    flush_left = p.flush;                                                          \
    _Pragma("unroll") for (int t = 0; t < MI; ++t)                                 \
      _Pragma("unroll") for (int u = 0; u < NI; ++u) {                             \
        acc2[t][u] += acc[t][u];                                                   \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;         \
      }                                                                            \
  }
#define PEANUT_FLUSH_FINISH()                                                      \
  if (p.flush > 0) {                                                               \
    _Pragma("unroll") for (int t = 0; t < MI; ++t)                                 \
      _Pragma("unroll") for (int u = 0; u < NI; ++u) acc[t][u] += acc2[t][u];      \
  }

// host side: pick split_p for T tiles over S resident workgroup slots on `cus` CUs; returns the number of tail tiles
// (0 = no split).  The tail's t tiles cut p ways put ceil(t * p / cus) workgroups on the busiest CU, and co-resident
// workgroups share that CU's matrix core, so the tail takes about
//     ceil(t * p / cus) * (nkt / p + fill)            k-tile times,  fill ~ 3 (pipeline fill + epilogue),
// plus, when split, the reduce launch (~5 us, 3 k-tile times).  (Counting slots instead of CUs -- two workgroups per
// CU as one "round" -- picked 7-way splits of 378 workgroups for the R-101 res4 layers at batch 1: half the CUs then
// carry two parts and the launch takes twice the balanced time.)
inline int plan_tail_split(int T, int S, int cus, int nkt, size_t tile_floats, size_t ws_floats, int* split_p) {
  *split_p = 1;
  if (S <= 0 || ws_floats == 0) return 0;
  const int t = T % S;
  if (t == 0) return 0;
  const bool cu_model = opt(OPT_SPLIT_MODEL) != 0;
  if (!cu_model || cus <= 0) {
    double best = 1.0;   // cost of the tail round without splitting, in tile-times
    int best_p = 1;
    for (int p = 2; p <= 32 && nkt / p >= 4; ++p) {
      if ((size_t)t * p * tile_floats > ws_floats) break;
      const int rounds = (int)(((long long)t * p + S - 1) / S);
      const double cost = (double)rounds / p * (1.0 + 3.0 * p / nkt);   // ~3 k-tiles of fill/drain per part
      if (cost < best - 0.02) { best = cost; best_p = p; }
    }
    if (best_p == 1) return 0;
    *split_p = best_p;
    return t;
  }
  auto stacked = [&](int p) { return (double)(((long long)t * p + cus - 1) / cus); };
  double best = stacked(1) * (nkt + 3.0);
  int best_p = 1;
  for (int p = 2; p <= 32 && nkt / p >= 4; ++p) {
    if ((size_t)t * p * tile_floats > ws_floats) break;
    const double cost = stacked(p) * ((double)nkt / p + 3.0) + 3.0;
    if (cost < best * 0.98) { best = cost; best_p = p; }
  }
  if (best_p == 1) return 0;
  *split_p = best_p;
  return t;
}

// sum the split_p partial tiles of every tail tile in order and apply the fused epilogue.
// grid = (tail tiles, BM / 16): each workgroup finishes a 16-row slab, so even a 16-tile tail spreads
// over 128 workgroups and the p partial reads of a slab are issued back to back.
template <int BM, int BN>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvKParams p) {
  const int j = blockIdx.x;                    // tail tile
  const int tile = p.n_full + j;
  int mt, nt;
  tile_to_mn(p, tile, &mt, &nt);
  const int m0 = mt * BM, n0 = nt * BN;
  constexpr int NV = BN / 4, SLAB = 16;
  // partial tiles of this tail tile: split_p of them, or -- stream-K -- one per workgroup whose run touches the tile
  int parts = p.split_p;
  const float* base = p.partial + (size_t)j * p.split_p * (BM * BN);
  if (p.sk_units > 0) {
    const int upt = p.nkt / p.sk_q;                      // units per tile
    const int w0 = sk_owner(p.sk_units, p.sk_g, (long long)j * upt), w1 = sk_owner(p.sk_units, p.sk_g, (long long)(j + 1) * upt - 1);
    parts = w1 - w0 + 1;
    base = p.partial + (size_t)j * p.sk_maxp * (BM * BN);
  }
  const bool vec_ok = (p.cout & 3) == 0;
  for (int i = threadIdx.x; i < SLAB * NV; i += 256) {
    const int row = blockIdx.y * SLAB + i / NV, c4 = (i % NV) * 4;
    const int m = m0 + row, n = n0 + c4;
    if (m >= p.M) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(base + row * BN + c4);
    for (int s = 1; s < parts; ++s) v += *reinterpret_cast<const f32x4*>(base + (size_t)s * (BM * BN) + row * BN + c4);
    const int ss_off = (p.mt_per_group && p.ss_group_stride) ? (mt / p.mt_per_group) * p.ss_group_stride : 0;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + ss_off + n) * p.alpha;
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + ss_off + n);
    // one fused multiply-add per channel, spelled out (what `v * sc + sh` compiled to anyway): the Winograd input transforms that sum
    // a producer's partial tiles themselves (winograd.hip: deferred_pixel) must reproduce this value bit for bit, whatever hipcc's
    // contraction decides in either instantiation (round 6: the f32x4 instantiation there had decided differently)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], sc[e], sh[e]);
    const size_t o = (size_t)m * p.cout + n;
    if (vec_ok) {
      if (n < p.cout) {
        if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + o);
        if (p.relu) v = relu_keep_nan(v);
        *reinterpret_cast<f32x4*>(p.y + o) = v;
      }
    } else {
      for (int e = 0; e < 4; ++e) {
        if (n + e < p.cout) {
          float x = v[e];
          if (p.res) x += p.res[o + e];
          if (p.relu) x = relu_keep_nan(x);
          p.y[o + e] = x;
        }
      }
    }
  }
}

// resident workgroup slots (CUs x occupancy) of one kernel, per device: 0 = not queried yet, -1 = unknown (never split).
// One object per kernel instantiation; the benign race of two threads filling an entry stores the same value twice.
struct SlotCache {
  static constexpr int kMaxDevices = 16;
  std::atomic<int> slots[kMaxDevices];
  std::atomic<int> cus[kMaxDevices];
  SlotCache() { for (int i = 0; i < kMaxDevices; ++i) { slots[i].store(0); cus[i].store(0); } }
};

// shared host launcher: occupancy query (once per instantiation and device), tail-split plan, launch, reduce
template <typename KernelT, int BM, int BN, int NT = 256>
int launch_with_tail_split(KernelT kernel, ConvKParams p, float* ws, size_t ws_floats, hipStream_t stream, SlotCache* cache) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SlotCache::kMaxDevices) return fail(-3, "conv launch: no current device");
  int slots = cache->slots[dev].load(std::memory_order_relaxed), cus = cache->cus[dev].load(std::memory_order_relaxed);
  if (slots == 0) {
    int occ = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, NT, 0) != hipSuccess || occ < 1)
      slots = -1;
    else
      slots = cus * occ;
    cache->cus[dev].store(cus, std::memory_order_relaxed);
    cache->slots[dev].store(slots, std::memory_order_relaxed);
  }
  const int mtiles = (p.M + BM - 1) / BM;
  const int T = mtiles * p.ntiles;
  int sp = 1;
  const int t = plan_tail_split(T, slots, cus, p.nkt, (size_t)BM * BN, ws ? ws_floats : 0, &sp);
  p.split_p = sp;
  p.n_sp = t * sp;
  p.n_full = T - t;
  p.partial = ws;
  p.mtiles = mtiles;
  p.nchunk = (int)opt(OPT_NCHUNK);
  p.res_prefetch = opt(OPT_RES_PREFETCH) != 0;
  // every tile split, 128 x 128 tiles in the plain order, no residual, one weight group: the consumer may sum the partial tiles itself
  bool deferred = false;
  if (p.defer) {
    DeferredSplit* df = p.defer;
    if (BM == 128 && BN == 128 && t > 0 && t == T && !p.res && p.mt_per_group == 0 && p.cout % 128 == 0 &&
        !(p.nchunk > 0 && p.ntiles > p.nchunk)) {
      df->valid = true;
      df->partial = ws; df->split_p = sp; df->ntiles = p.ntiles; df->M = p.M; df->cout = p.cout; df->relu = p.relu;
      df->scale = p.scale; df->shift = p.shift; df->alpha = p.alpha;
      deferred = true;
    }
    p.defer = nullptr;
  }
  hipLaunchKernelGGL(kernel, dim3((unsigned)(p.n_full + p.n_sp)), dim3(NT), 0, stream, p);
  if (t > 0 && !deferred) hipLaunchKernelGGL((conv_splitk_reduce_kernel<BM, BN>), dim3((unsigned)t, BM / 16), dim3(256), 0, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(-3, std::string("conv launch: ") + hipGetErrorString(e));
  return 0;
}

}  // namespace peanut
