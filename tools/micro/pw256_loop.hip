// Calibration of conv_pw_glds256_kernel's k-loop (csrc/conv_pw.hip): 256 x 128 tile, 8 waves, wave tile 64 x 64, three
// 48 KiB stages, 6 LDS-DMA pieces + 16 ds_read_b128 + 64 fp32 MFMAs per wave and k-tile, one barrier per k-tile --
// with the placement of the LDS-DMA requests varied (template V), operands L2-resident (every workgroup streams the same
// 384 KiB window).  One workgroup per CU, grid = CUs.
//   V = 0   requests at the top of the iteration (the product kernel)
//   V = 1   no requests after the prologue (upper bound of any placement)
//   V = 2   requests spread: one after every 8 MFMAs of k-groups 0..2
//   V = 3   waves 0-3 request at the top, waves 4-7 after their first 32 MFMAs (the two waves of a SIMD out of phase)
//   V = 4   all waves request after their first 16 MFMAs (fragment reads + MFMAs first)
//   V = 5   like 0 with s_setprio 1 around the MFMA groups
// and the same for a 256 x 256 tile (wave tile 64 x 128, two 64 KiB stages): T = 1.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int V>
__global__ __launch_bounds__(512) void loop256x128(const float* __restrict__ src, float* __restrict__ out, int nk) {
  constexpr int BM = 256, BN = 128, BK = 32, WN = 2, STAGES = 3;
  constexpr int TM = 64, TN = 64, MI = 2, NI = 2;
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;
  constexpr int A_INSTR = 4, B_INSTR = 2;
  __shared__ __attribute__((aligned(1024))) float smem[STAGES * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 3, lp = lane & 7;
  const float* a_src[A_INSTR];
  const float* b_src[B_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int r = (wave * A_INSTR + j) * 8 + lr;
    a_src[j] = src + r * BK + (lp ^ ((r >> 1) & 7)) * 4;
  }
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int r = (wave * B_INSTR + j) * 8 + lr;
    b_src[j] = src + A_FLOATS + r * BK + (lp ^ ((r >> 1) & 7)) * 4;
  }
  int wrap = 0;
#define DMA_A(stage, j) __builtin_amdgcn_global_load_lds((gptr_t)(a_src[j] + wrap * STAGE), (lptr_t)((stage) + (wave * A_INSTR + (j)) * 256), 16, 0, 0)
#define DMA_B(stage, j) __builtin_amdgcn_global_load_lds((gptr_t)(b_src[j] + wrap * STAGE), (lptr_t)((stage) + A_FLOATS + (wave * B_INSTR + (j)) * 256), 16, 0, 0)
#define DMA_TILE(stage) { DMA_A(stage, 0); DMA_A(stage, 1); DMA_A(stage, 2); DMA_A(stage, 3); DMA_B(stage, 0); DMA_B(stage, 1); wrap = (wrap + 1) & 7; }
#define BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const int swz = (li >> 1) & 7;
  int sw[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) sw[ks] = ((ks * 2 + hi) ^ swz) * 4;
  const int a_row = (wm * TM + li) * BK;
  const int b_row = A_FLOATS + (wn * TN + li) * BK;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  DMA_TILE(smem);
  DMA_TILE(smem + STAGE);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  BARRIER();
  int o_cur = 0, o_mid = STAGE, o_fill = 2 * STAGE;
  const bool late = (V == 3 && wave >= 4) || V == 4;
#define READ_GROUP(j)                                                                                    \
  _Pragma("unroll") for (int t = 0; t < MI; ++t) af[j][t] = *reinterpret_cast<const f32x4*>(cur + a_row + t * 32 * BK + sw[j]); \
  _Pragma("unroll") for (int u = 0; u < NI; ++u) bf[j][u] = *reinterpret_cast<const f32x4*>(cur + b_row + u * 32 * BK + sw[j]);
#define MFMA_KK(j, kk)                                                                                   \
  _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                         \
    _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                       \
      acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t][kk], bf[j][u][kk], acc[t][u], 0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const float* const cur = smem + o_cur;
    float* const fill = smem + o_fill;
    const bool more = (V != 1) && kt + 2 < nk;
    f32x4 af[4][MI], bf[4][NI];
    if (V == 0 || V == 5 || (V == 3 && !late)) { if (more) DMA_TILE(fill); }
    if (V == 2) {
      READ_GROUP(0); READ_GROUP(1); READ_GROUP(2); READ_GROUP(3);
      __builtin_amdgcn_sched_barrier(0);
      MFMA_KK(0, 0); MFMA_KK(0, 1);
      if (more) DMA_A(fill, 0);
      MFMA_KK(0, 2); MFMA_KK(0, 3);
      if (more) DMA_A(fill, 1);
      __builtin_amdgcn_sched_barrier(0);
      MFMA_KK(1, 0); MFMA_KK(1, 1);
      if (more) DMA_A(fill, 2);
      MFMA_KK(1, 2); MFMA_KK(1, 3);
      if (more) DMA_A(fill, 3);
      __builtin_amdgcn_sched_barrier(0);
      MFMA_KK(2, 0); MFMA_KK(2, 1);
      if (more) DMA_B(fill, 0);
      MFMA_KK(2, 2); MFMA_KK(2, 3);
      if (more) { DMA_B(fill, 1); wrap = (wrap + 1) & 7; }
      __builtin_amdgcn_sched_barrier(0);
      MFMA_KK(3, 0); MFMA_KK(3, 1); MFMA_KK(3, 2); MFMA_KK(3, 3);
    } else if (V == 3 || V == 4) {
      READ_GROUP(0); READ_GROUP(1); READ_GROUP(2); READ_GROUP(3);
      MFMA_KK(0, 0); MFMA_KK(0, 1); MFMA_KK(0, 2); MFMA_KK(0, 3);
      if (V == 3) { MFMA_KK(1, 0); MFMA_KK(1, 1); MFMA_KK(1, 2); MFMA_KK(1, 3); }
      __builtin_amdgcn_sched_barrier(0);
      if (late && more) DMA_TILE(fill);
      __builtin_amdgcn_sched_barrier(0);
      if (V == 4) { MFMA_KK(1, 0); MFMA_KK(1, 1); MFMA_KK(1, 2); MFMA_KK(1, 3); }
      MFMA_KK(2, 0); MFMA_KK(2, 1); MFMA_KK(2, 2); MFMA_KK(2, 3);
      MFMA_KK(3, 0); MFMA_KK(3, 1); MFMA_KK(3, 2); MFMA_KK(3, 3);
    } else {
      READ_GROUP(0); READ_GROUP(1); READ_GROUP(2); READ_GROUP(3);
      if (V == 5) __builtin_amdgcn_s_setprio(1);
      MFMA_KK(0, 0); MFMA_KK(0, 1); MFMA_KK(0, 2); MFMA_KK(0, 3);
      MFMA_KK(1, 0); MFMA_KK(1, 1); MFMA_KK(1, 2); MFMA_KK(1, 3);
      MFMA_KK(2, 0); MFMA_KK(2, 1); MFMA_KK(2, 2); MFMA_KK(2, 3);
      MFMA_KK(3, 0); MFMA_KK(3, 1); MFMA_KK(3, 2); MFMA_KK(3, 3);
      if (V == 5) __builtin_amdgcn_s_setprio(0);
    }
    if (more) { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    BARRIER();
    { const int t = o_cur; o_cur = o_mid; o_mid = o_fill; o_fill = t; }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[t][u][r];
  out[blockIdx.x * 512 + tid] = s;
#undef DMA_A
#undef DMA_B
#undef DMA_TILE
#undef READ_GROUP
#undef MFMA_KK
}

// 256 x 256 tile, wave tile 64 x 128 (MI 2, NI 4: 128 accumulator registers), two 64 KiB stages, 8 DMA pieces per wave
// and k-tile for 128 MFMAs.  V = 0: requests at the top; V = 1: none; V = 2: spread (one per 16 MFMAs).
template <int V>
__global__ __launch_bounds__(512) void loop256x256(const float* __restrict__ src, float* __restrict__ out, int nk) {
  constexpr int BM = 256, BN = 256, BK = 32, WN = 2, STAGES = 2;
  constexpr int TM = 64, TN = 128, MI = 2, NI = 4;
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;
  constexpr int A_INSTR = 4, B_INSTR = 4;
  __shared__ __attribute__((aligned(1024))) float smem[STAGES * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane >> 3, lp = lane & 7;
  const float* a_src[A_INSTR];
  const float* b_src[B_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int r = (wave * A_INSTR + j) * 8 + lr;
    a_src[j] = src + r * BK + (lp ^ ((r >> 1) & 7)) * 4;
  }
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int r = (wave * B_INSTR + j) * 8 + lr;
    b_src[j] = src + A_FLOATS + r * BK + (lp ^ ((r >> 1) & 7)) * 4;
  }
  int wrap = 0;
#define DMA_A(stage, j) __builtin_amdgcn_global_load_lds((gptr_t)(a_src[j] + wrap * STAGE), (lptr_t)((stage) + (wave * A_INSTR + (j)) * 256), 16, 0, 0)
#define DMA_B(stage, j) __builtin_amdgcn_global_load_lds((gptr_t)(b_src[j] + wrap * STAGE), (lptr_t)((stage) + A_FLOATS + (wave * B_INSTR + (j)) * 256), 16, 0, 0)
#define DMA_TILE(stage) { DMA_A(stage, 0); DMA_A(stage, 1); DMA_A(stage, 2); DMA_A(stage, 3); DMA_B(stage, 0); DMA_B(stage, 1); DMA_B(stage, 2); DMA_B(stage, 3); wrap = (wrap + 1) & 3; }
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, hi = lane >> 5;
  const int swz = (li >> 1) & 7;
  int sw[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) sw[ks] = ((ks * 2 + hi) ^ swz) * 4;
  const int a_row = (wm * TM + li) * BK;
  const int b_row = A_FLOATS + (wn * TN + li) * BK;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  DMA_TILE(smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BARRIER();
#define READ_GROUP(j)                                                                                    \
  _Pragma("unroll") for (int t = 0; t < MI; ++t) af[j][t] = *reinterpret_cast<const f32x4*>(cur + a_row + t * 32 * BK + sw[j]); \
  _Pragma("unroll") for (int u = 0; u < NI; ++u) bf[j][u] = *reinterpret_cast<const f32x4*>(cur + b_row + u * 32 * BK + sw[j]);
#define MFMA_KK(j, kk)                                                                                   \
  _Pragma("unroll") for (int t = 0; t < MI; ++t)                                                         \
    _Pragma("unroll") for (int u = 0; u < NI; ++u)                                                       \
      acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t][kk], bf[j][u][kk], acc[t][u], 0, 0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const float* const cur = smem + (kt & 1) * STAGE;
    float* const fill = smem + ((kt + 1) & 1) * STAGE;
    const bool more = (V != 1) && kt + 1 < nk;
    f32x4 af[2][MI], bf[2][NI];
    if (V == 0 && more) DMA_TILE(fill);
#pragma unroll
    for (int jj = 0; jj < 4; jj += 2) {
      _Pragma("unroll") for (int t = 0; t < MI; ++t) af[0][t] = *reinterpret_cast<const f32x4*>(cur + a_row + t * 32 * BK + sw[jj]);
      _Pragma("unroll") for (int u = 0; u < NI; ++u) bf[0][u] = *reinterpret_cast<const f32x4*>(cur + b_row + u * 32 * BK + sw[jj]);
      _Pragma("unroll") for (int t = 0; t < MI; ++t) af[1][t] = *reinterpret_cast<const f32x4*>(cur + a_row + t * 32 * BK + sw[jj + 1]);
      _Pragma("unroll") for (int u = 0; u < NI; ++u) bf[1][u] = *reinterpret_cast<const f32x4*>(cur + b_row + u * 32 * BK + sw[jj + 1]);
      MFMA_KK(0, 0); MFMA_KK(0, 1);
      if (V == 2 && more) { if (jj == 0) DMA_A(fill, 0); else DMA_B(fill, 0); }
      MFMA_KK(0, 2); MFMA_KK(0, 3);
      if (V == 2 && more) { if (jj == 0) DMA_A(fill, 1); else DMA_B(fill, 1); }
      MFMA_KK(1, 0); MFMA_KK(1, 1);
      if (V == 2 && more) { if (jj == 0) DMA_A(fill, 2); else DMA_B(fill, 2); }
      MFMA_KK(1, 2); MFMA_KK(1, 3);
      if (V == 2 && more) { if (jj == 0) DMA_A(fill, 3); else { DMA_B(fill, 3); wrap = (wrap + 1) & 3; } }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BARRIER();
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < MI; ++t)
#pragma unroll
    for (int u = 0; u < NI; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[t][u][r];
  out[blockIdx.x * 512 + tid] = s;
}

template <typename K>
void run(const char* name, K kernel, double flop_per_wg_ktile, const float* src, float* out, int cus, int nk) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(cus), dim3(512), 0, 0, src, out, nk);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kernel, dim3(cus), dim3(512), 0, 0, src, out, nk);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  hipError_t e = hipGetLastError();
  printf("{\"variant\": \"%s\", \"ms\": %.3f, \"tflops\": %.1f, \"err\": \"%s\"}\n", name, ms, flop_per_wg_ktile * cus * nk / ms / 1e9,
         e == hipSuccess ? "" : hipGetErrorString(e));
  fflush(stdout);
}

int main(int argc, char** argv) {
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int nk = argc > 1 ? atoi(argv[1]) : 1024;
  float *src, *out;
  const size_t src_floats = (size_t)16 * (256 + 256) * 32 + 4096;
  (void)hipMalloc(&src, src_floats * sizeof(float));
  {
    float* h = (float*)malloc(src_floats * sizeof(float));
    unsigned s = 12345u;
    for (size_t i = 0; i < src_floats; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)((s >> 9) & 0x3fff) / 16384.f; }   // dense values: realistic toggling
    (void)hipMemcpy(src, h, src_floats * sizeof(float), hipMemcpyHostToDevice);
    free(h);
  }
  (void)hipMalloc(&out, (size_t)cus * 512 * sizeof(float));
  const double f128 = 2.0 * 256 * 128 * 32, f256 = 2.0 * 256 * 256 * 32;
  for (int rep = 0; rep < 2; ++rep) {
    run("256x128 V0 top (product)", loop256x128<0>, f128, src, out, cus, nk);
    run("256x128 V1 no DMA", loop256x128<1>, f128, src, out, cus, nk);
    run("256x128 V2 spread", loop256x128<2>, f128, src, out, cus, nk);
    run("256x128 V3 waves out of phase", loop256x128<3>, f128, src, out, cus, nk);
    run("256x128 V4 after 16 MFMAs", loop256x128<4>, f128, src, out, cus, nk);
    run("256x128 V5 setprio", loop256x128<5>, f128, src, out, cus, nk);
    run("256x256 V0 top", loop256x256<0>, f256, src, out, cus, nk);
    run("256x256 V1 no DMA", loop256x256<1>, f256, src, out, cus, nk);
    run("256x256 V2 spread", loop256x256<2>, f256, src, out, cus, nk);
  }
  return 0;
}
