// Calibration of the fp32 conv k-loop: the same 64 MFMAs per k-tile as conv_igemm_kernel<128,128,32,2,2>, with the
// other ingredients of the loop switched on one at a time (template flags), 2 workgroups per CU like the real kernel.
//   F_LDS     fragments re-read from LDS every k-tile (16 ds_read_b128, padded rows, two halves)
//   F_BAR     one s_barrier per k-tile
//   F_STAGE   8 ds_write_b128 per k-tile (register -> LDS staging of the next tile)
//   F_GLOBAL  8 global_load_dwordx4 per k-tile feeding the staging registers
//   F_VALU    ~90 integer VALU ops per k-tile (stand-in for the address arithmetic)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LS = 36, STAGE = 256 * LS;   // floats

template <bool F_LDS, bool F_BAR, bool F_STAGE, bool F_GLOBAL, bool F_VALU, bool F_GLDS = false>
__global__ __launch_bounds__(256, 2) void loop_kernel(const float* __restrict__ src, float* __restrict__ out, int nk, int stride) {
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, hi = lane >> 5;
  const int a_off = (wm * 64 + li) * LS + hi * 4;
  const int b_off = 128 * LS + (wn * 64 + li) * LS + hi * 4;
  for (int i = tid; i < 2 * STAGE; i += 256) smem[i] = 0.001f * (i & 63);
  __syncthreads();
  f32x16 acc[2][2];
  for (int t = 0; t < 2; ++t) for (int u = 0; u < 2; ++u) for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  f32x4 af[4][2], bf[4][2];
  for (int j = 0; j < 4; ++j) for (int t = 0; t < 2; ++t) {
    af[j][t] = *reinterpret_cast<const f32x4*>(smem + a_off + t * 32 * LS + j * 8);
    bf[j][t] = *reinterpret_cast<const f32x4*>(smem + b_off + t * 32 * LS + j * 8);
  }
  f32x4 rg[8];
  for (int j = 0; j < 8; ++j) rg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* gp = src + ((size_t)blockIdx.x * 256 + tid) * 4;
  unsigned long long addr[4] = {1, 2, 3, 4};
  unsigned v0 = tid, v1 = tid * 3;
  for (int kt = 0; kt < nk; ++kt) {
    float* nxt = smem + ((kt + 1) & 1) * STAGE;
    const float* cur = smem + (kt & 1) * STAGE;
    if (F_STAGE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(nxt + ((tid + 256 * j) >> 3) * LS + (tid & 7) * 4) = rg[j];
    }
    if (F_GLOBAL) {
#pragma unroll
      for (int j = 0; j < 8; ++j) rg[j] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>((unsigned long long)(gp + (size_t)j * stride));
      gp += 8 * (size_t)stride;
    }
    if (F_GLDS) {   // 8 x global_load_lds_dwordx4 per thread: the next k-tile straight into LDS[nxt], no VGPR round trip
      float* dst = nxt + wave * 8 * 256;   // wave-uniform: 8 KB per wave per stage
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>((unsigned long long)(gp + (size_t)j * stride)),
                                         (__attribute__((address_space(3))) void*)(dst + j * 256), 16, 0, 0);
      gp += 8 * (size_t)stride;
    }
    if (F_LDS) {
#pragma unroll
      for (int j = 2; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          af[j][t] = *reinterpret_cast<const f32x4*>(cur + a_off + t * 32 * LS + j * 8);
          bf[j][t] = *reinterpret_cast<const f32x4*>(cur + b_off + t * 32 * LS + j * 8);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (F_VALU) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int r = 0; r < 5; ++r) {
          v0 = v0 * 0x9E3779B1u + v1;
          v1 = (v1 ^ (v0 >> 7)) + kt;
        }
        addr[q] = (addr[q] + (unsigned long long)v0 * v1) ^ (addr[q] >> 3);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t][kk], bf[j][u][kk], acc[t][u], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (F_BAR) __syncthreads();
    if (F_LDS) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          af[j][t] = *reinterpret_cast<const f32x4*>(nxt + a_off + t * 32 * LS + j * 8);
          bf[j][t] = *reinterpret_cast<const f32x4*>(nxt + b_off + t * 32 * LS + j * 8);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 2; j < 4; ++j)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j][t][kk], bf[j][u][kk], acc[t][u], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = (float)(addr[0] + addr[1] + addr[2] + addr[3]) + rg[0].x + rg[7].w;
  for (int t = 0; t < 2; ++t) for (int u = 0; u < 2; ++u) for (int r = 0; r < 16; ++r) s += acc[t][u][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <bool A, bool B, bool C, bool D, bool E, bool G = false>
void run(const char* name, const float* src, float* out, int cus, int nk, int stride) {
  const int grid = cus * 2 * 4;   // four rounds of 2 workgroups per CU
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((loop_kernel<A, B, C, D, E, G>), dim3(grid), dim3(256), 0, 0, src, out, nk, stride);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((loop_kernel<A, B, C, D, E, G>), dim3(grid), dim3(256), 0, 0, src, out, nk, stride);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  const double fl = (double)grid * 4 * nk * 64 * (32.0 * 32 * 2 * 2);
  const double us_per_ktile_pair = ms * 1e3 / 4 / nk;   // wall time for one k-tile of each of the 2 resident workgroups
  printf("{\"variant\": \"%s\", \"ms\": %.3f, \"tflops\": %.1f, \"us_per_ktile_pair\": %.3f}\n", name, ms, fl / ms / 1e9, us_per_ktile_pair);
}

int main() {
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int nk = 256, stride = 1 << 20;
  float *src, *out;
  (void)hipMalloc(&src, ((size_t)8 * nk * stride + (size_t)cus * 8 * 256 * 4 + 64) * sizeof(float) > ((size_t)6 << 30) ? ((size_t)6 << 30) : ((size_t)8 * nk * stride + (size_t)cus * 8 * 1024 + 64) * sizeof(float));
  (void)hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
  run<false, false, false, false, false>("mfma only (registers)", src, out, cus, nk, 0);
  run<true, false, false, false, false>("+ LDS fragment reads", src, out, cus, nk, 0);
  run<true, true, false, false, false>("+ LDS reads + barrier", src, out, cus, nk, 0);
  run<true, true, true, false, false>("+ LDS reads + barrier + ds_write staging", src, out, cus, nk, 0);
  run<true, true, true, true, false>("+ ... + global loads (same lines, L2 hits)", src, out, cus, nk, 0);
  run<true, true, true, false, true>("+ LDS + barrier + staging + VALU", src, out, cus, nk, 0);
  run<true, true, true, true, true>("everything", src, out, cus, nk, 0);
  run<true, true, false, false, false, true>("LDS reads + barrier + global_load_lds (no VGPR staging)", src, out, cus, nk, 0);
  run<false, false, false, false, true>("mfma + VALU only", src, out, cus, nk, 0);
  run<false, true, false, false, false>("mfma + barrier only", src, out, cus, nk, 0);
  return 0;
}
