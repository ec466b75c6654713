// Skinny grouped GEMM (round 6): Y[g][m][n] = relu(scale[g][n] * sum_k X[g][m][k] * W[g][k][n] + shift[g][n]) for a handful of rows m per
// group -- the per-scale 1x1 convs of the PSP pyramid (psp_head.py:39-46: 1, 4, 9 and 36 pooled vectors of 2048 channels -> 512) and
// the per-scale Q tables of the folded bottleneck (512 -> 9 x 512) at batch 1.
//
// Why: on the MFMA kernels each scale is padded to a 128-row tile (97 % padding), the launch is split along k to find 256 workgroups
// and a reduce launch follows -- 28 us + 41 us on one 240 x 240 map, and the tiles want 64 KiB of LDS, so that on a 720 x 720 map the two
// launches cannot start before the bottleneck's position GEMM (147 KiB of LDS per CU) has finished although they sit on the side stream.
// The work is weight streaming (17 + 38 MB once); the arithmetic is nothing (75 MFLOP).
//
// Here a workgroup takes one 128-wide n-tile of one group and a run of four k-tiles; thread t holds output column n = t & 127 and the
// k-half kh = t >> 7 of every k-tile (16 channels: one 64-byte piece of the packed weights, [n-tile][k-tile][128][32] as the MFMA
// kernels read them; the whole run's 16 pieces are requested at once), for at most 12 rows of the group.  X of the run goes through LDS
// once (6 KiB) and is read back as broadcast ds_read_b128; the kernel is compiled without packed fp32 instructions (see the note in the loop).  Every
// (k-run, kh) pair writes a raw partial
// [rows][N]; a second small launch adds the partials in order and applies scale / shift / ReLU.  Summation order differs from the MFMA
// kernels' (k ascending within a lane here, the matrix core's internal order there): results agree to rounding, not to the bit; both
// forms are held to the reference goldens (tests/test_pred_gpu.py).
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "conv_common.h"

namespace peanut {

namespace {

constexpr int kSkinnyMaxRows = 64;
constexpr int kSkinnyMaxGroups = 8;
constexpr int kSkinnyChunkRows = 12;
constexpr int kSkinnyMaxChunks = 24;

struct SkinnyParams {
  const float* x;        // [groups * group_stride_rows, K]
  const float* w;        // packed, group g at g * w_group_stride floats
  float* partial;        // [parts][groups][rows][N]
  int K, N, nkt, ntiles, rows, groups, group_stride_rows, ksplit;
  long long w_group_stride;
  int grp_rows[kSkinnyMaxGroups];      // rows of each group that hold data
  // blockIdx.y walks ROW CHUNKS: chunk c = rows [chunk_m0[c], chunk_m0[c] + chunk_rows[c]) of group chunk_g[c] (at most kSkinnyChunkRows rows:
  // the inner loop is VALU-bound, rows x 64 v_fmac per wave and k-run, so the 36-row scale is cut in three)
  unsigned char chunk_g[kSkinnyMaxChunks], chunk_m0[kSkinnyMaxChunks], chunk_rows[kSkinnyMaxChunks];
};

// X of the workgroup's k-run and row chunk is staged in LDS once ([row][k], 12 rows x 128 channels = 6 KiB; scalar loads straight from memory were
// the first version: 36 dependent s_load_dwordx16 per k-tile, 7 us per k-tile); the inner loop reads it back as broadcast ds_read_b128.
constexpr int kSkinnyRunTiles = 4;                 // k-tiles per workgroup
__global__ __launch_bounds__(256) PEANUT_NO_PK_F32 void gemm_skinny_kernel(const SkinnyParams p) {
  constexpr int RMAX = kSkinnyChunkRows;
  __shared__ f32x4 xs[RMAX * kSkinnyRunTiles * 8];          // [row][k / 4]
  const int tid = threadIdx.x;
  const int n = tid & 127;
  const int kh = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int nt = blockIdx.x % p.ntiles, ks = blockIdx.x / p.ntiles;
  const int g = p.chunk_g[blockIdx.y], mbase = p.chunk_m0[blockIdx.y], R = p.chunk_rows[blockIdx.y];
  const int kt0 = ks * kSkinnyRunTiles;
  const float* wp = p.w + (size_t)g * p.w_group_stride + ((size_t)nt * p.nkt + kt0) * (128 * 32) + n * 32 + kh * 16;
  f32x4 wc[kSkinnyRunTiles][4];
#pragma unroll
  for (int t = 0; t < kSkinnyRunTiles; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) wc[t][q] = *reinterpret_cast<const f32x4*>(wp + (size_t)t * (128 * 32) + q * 4);
  {
    const float* xg = p.x + ((size_t)g * p.group_stride_rows + mbase) * p.K + kt0 * 32;
    constexpr int PER_ROW = kSkinnyRunTiles * 8;             // 16-byte pieces per row of the run
    for (int i = tid; i < R * PER_ROW; i += 256) {
      const int m = i / PER_ROW, c = i - m * PER_ROW;
      xs[m * PER_ROW + c] = *reinterpret_cast<const f32x4*>(xg + (size_t)m * p.K + c * 4);
    }
  }
  // (__syncthreads() spelled out: HIP's wrapper is a function compiled WITH packed fp32 and would stay a call from this kernel)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  float acc[RMAX];
#pragma unroll
  for (int m = 0; m < RMAX; ++m) acc[m] = 0.f;
#pragma unroll
  for (int t = 0; t < kSkinnyRunTiles; ++t) {
#pragma unroll
    for (int m0 = 0; m0 < RMAX; m0 += 4) {
      if (m0 < R) {                                              // (wave-uniform: R is a kernel argument)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
          const int m = m0 + mm < R ? m0 + mm : R - 1;           // rows past the group's last repeat it (their sums are not stored)
          const f32x4* xr = xs + m * (kSkinnyRunTiles * 8) + t * 8 + kh * 4;      // the same address in every lane: a broadcast read
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 xv = xr[q];
            // (PEANUT_NO_PK_F32: left to itself hipcc packs this loop into v_pk_fma_f32 with op_sel, two rows per instruction -- the
            // form that returned wrong sums next to the emulated modes' gemm_rs kernel, common.h; unpacked it is one v_fmac_f32 per product)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[m0 + mm] = __builtin_fmaf(xv[e], wc[t][q][e], acc[m0 + mm]);
          }
        }
      }
    }
  }
  float* out = p.partial + (((size_t)(ks * 2 + kh) * p.groups + g) * p.rows + mbase) * (size_t)p.N + nt * 128 + n;
#pragma unroll
  for (int m = 0; m < RMAX; ++m)
    if (m < R) out[(size_t)m * p.N] = acc[m];
}

struct SkinnyEpi {
  const float* partial;
  const float* scale;
  const float* shift;
  float* y;
  int N, rows, groups, group_stride_rows, parts, ss_group_stride, relu;
  float alpha;
  int grp_rows[kSkinnyMaxGroups];
};

// one thread per (group, row, 4 columns): the partials added in part order, then the conv epilogue's expression
__global__ __launch_bounds__(256) void gemm_skinny_finish_kernel(const SkinnyEpi p) {
  const int nv = p.N / 4;
  const long long total = (long long)p.groups * p.rows * nv;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % nv) * 4;
    const int m = (int)((i / nv) % p.rows);
    const int g = (int)(i / ((long long)nv * p.rows));
    if (m >= p.grp_rows[g]) continue;
    const size_t plane = (size_t)p.groups * p.rows * p.N;
    const float* src = p.partial + ((size_t)g * p.rows + m) * p.N + c4;
    f32x4 v = *reinterpret_cast<const f32x4*>(src);
    for (int s0 = 1; s0 < p.parts; s0 += 8) {                  // eight parts' loads in flight, added in part order
      f32x4 part[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) part[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(s0 + u < p.parts ? s0 + u : p.parts - 1) * plane);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + u < p.parts) v += part[u];
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + (size_t)g * p.ss_group_stride + c4) * p.alpha;
    const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + (size_t)g * p.ss_group_stride + c4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = __builtin_fmaf(v[e], sc[e], sh[e]);
      if (p.relu) t = relu_keep_nan(t);
      v[e] = t;
    }
    *reinterpret_cast<f32x4*>(p.y + ((size_t)g * p.group_stride_rows + m) * p.N + c4) = v;
  }
}

// ---- LDS canary (include/peanut_hip.h: peanut_debug_lds_canary) ----
__global__ __launch_bounds__(256) void lds_canary_kernel(int words, int rounds, int* mismatches, int hammer) {
  extern __shared__ unsigned canary[];
  const unsigned salt = 0x9e3779b9u * (blockIdx.x + 1);
  for (int i = threadIdx.x; i < words; i += 256) canary[i] = salt ^ (unsigned)i;
  __syncthreads();
  int bad = 0;
  for (int r = 0; r < rounds; ++r) {
    if (rounds > 0 && !hammer) __builtin_amdgcn_s_sleep(64);
    for (int i = threadIdx.x; i < words; i += 256) {
      const unsigned v = canary[i];
      if (v != (salt ^ (unsigned)i)) { ++bad; canary[i] = salt ^ (unsigned)i; }
    }
    if (hammer) {      // keep the LDS pipe busy: broadcast 16-byte reads, as a compute kernel that stages a small operand there would
      const uint4* c4 = reinterpret_cast<const uint4*>(canary);
      unsigned acc = 0;
      for (int i = 0; i < words / 4; ++i) { const uint4 q = c4[i]; acc += q.x ^ q.y ^ q.z ^ q.w; }
      if (acc == 0x12345u) ++bad;
    }
  }
  if (bad) atomicAdd(mismatches, bad);
}

}  // namespace

// A grouped pointwise launch (one 128-row m-tile per weight group, `group_valid` data rows per group at most) that the skinny kernel
// takes: fp32, 128-wide packing, whole n-tiles, no residual, one running sum, a scratch that holds the partials.
bool gemm_skinny_takes(const ConvKParams& p, int bn_tile, size_t ws_floats) {
  if (opt(OPT_PW_SKINNY) == 0) return false;
  if (p.mt_per_group != 1 || p.group_valid <= 0 || p.group_valid > kSkinnyMaxRows || p.res || p.flush != 0 || p.c2 != 0 || p.stride != 1 ||
      p.ntaps != 1 || bn_tile != 128 || p.cout % 128 != 0 || p.M % 128 != 0 || p.M / 128 > kSkinnyMaxGroups || p.ss_group_stride == 0)
    return false;
  if (p.nkt % kSkinnyRunTiles != 0 || p.nkt / kSkinnyRunTiles > 64) return false;
  int nchunks = 0;      // the launch's chunk table is fixed-size: a shape that would overflow it stays with the MFMA kernels
  for (int g = 0; g < p.M / 128; ++g) {
    const int rows = p.group_rows ? std::min(std::max(p.group_rows[g], 1), p.group_valid) : p.group_valid;
    nchunks += (rows + kSkinnyChunkRows - 1) / kSkinnyChunkRows;
  }
  if (nchunks > kSkinnyMaxChunks) return false;
  return (size_t)2 * (p.nkt / kSkinnyRunTiles) * (p.M / 128) * p.group_valid * p.cout <= ws_floats;       // (k-runs x 2 halves of raw partials)
}

int launch_gemm_skinny(const ConvKParams& p, float* ws, size_t ws_floats, hipStream_t stream) {
  SkinnyParams a{};
  a.x = p.x; a.w = p.w; a.partial = ws;
  a.K = p.c1; a.N = p.cout; a.nkt = p.nkt; a.ntiles = p.cout / 128; a.rows = p.group_valid; a.groups = p.M / 128;
  a.group_stride_rows = 128; a.w_group_stride = p.w_group_stride;
  // the same rows for every group unless the caller said more (pred_api.hip: the pyramid's scales hold B * k^2 rows each)
  for (int g = 0; g < kSkinnyMaxGroups; ++g) a.grp_rows[g] = p.group_valid;
  // k-runs of kSkinnyRunTiles k-tiles: psp convs (K = 2048) 16 runs x 4 n-tiles x 4 scales = 256 workgroups, Q tables (K = 512) 4 x 36 x 4 = 576
  if (a.nkt % kSkinnyRunTiles != 0) return fail(-2, "gemm_skinny: k-tiles not a multiple of the run length");
  const int ksplit = a.nkt / kSkinnyRunTiles;
  a.ksplit = ksplit;
  const int parts = 2 * ksplit;
  if ((size_t)parts * a.groups * a.rows * a.N > ws_floats) return fail(-2, "gemm_skinny: scratch too small");
  if (p.group_rows) {
    for (int g = 0; g < a.groups; ++g) a.grp_rows[g] = std::min(std::max(p.group_rows[g], 1), a.rows);
  }
  int nchunks = 0;
  for (int g = 0; g < a.groups; ++g)
    for (int m0 = 0; m0 < a.grp_rows[g]; m0 += kSkinnyChunkRows) {
      if (nchunks >= kSkinnyMaxChunks) return fail(-2, "gemm_skinny: too many row chunks");
      a.chunk_g[nchunks] = (unsigned char)g; a.chunk_m0[nchunks] = (unsigned char)m0;
      a.chunk_rows[nchunks] = (unsigned char)std::min(kSkinnyChunkRows, a.grp_rows[g] - m0);
      ++nchunks;
    }
  note_kernel("gemm_skinny");
  hipLaunchKernelGGL(gemm_skinny_kernel, dim3((unsigned)(a.ntiles * ksplit), (unsigned)nchunks), dim3(256), 0, stream, a);
  SkinnyEpi e{};
  e.partial = ws; e.scale = p.scale; e.shift = p.shift; e.y = p.y;
  e.N = a.N; e.rows = a.rows; e.groups = a.groups; e.group_stride_rows = 128; e.parts = parts; e.ss_group_stride = p.ss_group_stride;
  e.relu = p.relu; e.alpha = p.alpha;
  for (int g = 0; g < kSkinnyMaxGroups; ++g) e.grp_rows[g] = a.grp_rows[g];
  const long long total = (long long)a.groups * a.rows * (a.N / 4);
  hipLaunchKernelGGL(gemm_skinny_finish_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, stream, e);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(-3, std::string("gemm_skinny launch: ") + hipGetErrorString(err));
  return 0;
}

// ---- packed-FMA canary (include/peanut_hip.h: peanut_debug_pkfma_canary) ----
// The sums of the skinny kernel's inner loop -- two rows per packed instruction -- three ways on the same operands:
//   risky:  v_pk_fma_f32 acc2, x2, w2, acc2 op_sel:[0,1,0]      the weight sits in the HIGH register of its pair, both halves pick it
//   safe:   v_pk_fma_f32 acc2, x2, w2, acc2 op_sel_hi:[1,0,1]   the weight sits in the LOW register, both halves pick it
//   scalar: two v_fmac_f32
// mismatches[0] counts sums where risky != scalar, mismatches[1] where safe != scalar.  Measured on gfx950 (profiles/r9r): both 0 alone
// and next to fp32 MFMA kernels; next to the emulated modes' fp16 / bf16 MFMA kernels only the risky form goes wrong (low halves,
// lanes 48-63) -- the form hipcc emits when it packs scalar code, and the reason for PEANUT_NO_PK_F32 (common.h).
__global__ __launch_bounds__(256) void pkfma_canary_kernel(int rounds, int* mismatches) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  int bad_risky = 0, bad_safe = 0;
  for (int r = 0; r < rounds; ++r) {
    f32x2 ar[6], as[6];
    float s0[6], s1[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) { ar[m] = f32x2{0.f, 0.f}; as[m] = f32x2{0.f, 0.f}; s0[m] = 0.f; s1[m] = 0.f; }
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
      const float w = (float)((((tid + r) * 64 + k) * 2246822519u >> 21) & 511) * (1.0f / 256.0f) - 1.0f;
      f32x2 w_hi = {0.f, w}, w_lo = {w, 0.f};
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        f32x2 x2;
        x2[0] = (float)((((2 * m) * 64 + k + r) * 2654435761u >> 20) & 1023) * (1.0f / 512.0f) - 1.0f;
        x2[1] = (float)((((2 * m + 1) * 64 + k + r) * 2654435761u >> 20) & 1023) * (1.0f / 512.0f) - 1.0f;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(ar[m]) : "v"(x2), "v"(w_hi));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(as[m]) : "v"(x2), "v"(w_lo));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s0[m]) : "v"(x2[0]), "v"(w));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s1[m]) : "v"(x2[1]), "v"(w));
      }
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      bad_risky += (ar[m][0] != s0[m] ? 1 : 0) + (ar[m][1] != s1[m] ? 1 : 0);
      bad_safe += (as[m][0] != s0[m] ? 1 : 0) + (as[m][1] != s1[m] ? 1 : 0);
    }
  }
  if (bad_risky) atomicAdd(mismatches, bad_risky);
  if (bad_safe) atomicAdd(mismatches + 1, bad_safe);
}

}  // namespace peanut

extern "C" int peanut_debug_pkfma_canary(int workgroups, int rounds, int* mismatches, void* stream) {
  using namespace peanut;
  if (workgroups < 1 || rounds < 1 || !mismatches) return fail(-2, "peanut_debug_pkfma_canary: bad argument");
  hipLaunchKernelGGL(pkfma_canary_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, rounds, mismatches);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("pkfma_canary launch: ") + hipGetErrorString(e));
}

extern "C" int peanut_debug_lds_canary(int workgroups, int lds_bytes, int rounds, int* mismatches, void* stream) {
  using namespace peanut;
  if (workgroups < 1 || lds_bytes < 1024 || lds_bytes > 65536 || lds_bytes % 1024 || rounds == 0 || !mismatches)
    return fail(-2, "peanut_debug_lds_canary: bad argument");
  hipLaunchKernelGGL(lds_canary_kernel, dim3((unsigned)workgroups), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, lds_bytes / 4, rounds < 0 ? -rounds : rounds,
                     mismatches, rounds < 0 ? 1 : 0);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(-3, std::string("lds_canary launch: ") + hipGetErrorString(e));
}
