// Calibration: what fp32 / bf16 MFMA rate does this MI355X sustain with nothing but matrix instructions in flight?
// (the roofline peak in MI355X_MICROARCH.md is the 2.4 GHz nameplate; a power- or clock-limited part sits below it)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int ACCS>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters, float a, float b) {
  f32x16 acc[ACCS];
  for (int i = 0; i < ACCS; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < ACCS; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ACCS>
__global__ __launch_bounds__(256) void k_bf16(float* out, int iters, short a0) {
  f32x16 acc[ACCS];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = a0; b[i] = a0 + 1; }
  for (int i = 0; i < ACCS; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < ACCS; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// same instruction stream, operands = per-lane pseudo-random fp32 values (what real activations/weights look like
// to the datapath) instead of two constants
template <int ACCS>
__global__ __launch_bounds__(256) void k_f32_random(float* out, int iters, const float* __restrict__ rnd) {
  f32x16 acc[ACCS];
  for (int i = 0; i < ACCS; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = rnd[(threadIdx.x * 16 + i) & 4095]; b[i] = rnd[(threadIdx.x * 16 + 8 + i) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 7], b[(u * 3 + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < ACCS; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class F>
double time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out;
  hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
  const int iters = 2000;
  for (int wgs_per_cu : {1, 2}) {
    const int grid = cus * wgs_per_cu;
    // fp32: 16*ACCS MFMA per iter per wave, each 32*32*2*2 flop
    double ms = time_ms([&] { hipLaunchKernelGGL(k_f32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); }, 5);
    double fl = (double)grid * 4 * iters * 16 * 4 * (32.0 * 32 * 2 * 2);
    printf("{\"kernel\": \"mfma_f32_32x32x2_f32 only\", \"waves_per_simd\": %d, \"ms\": %.3f, \"tflops\": %.1f}\n", wgs_per_cu, ms, fl / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_bf16<4>, dim3(grid), dim3(256), 0, 0, out, iters, (short)0x3f80); }, 5);
    fl = (double)grid * 4 * iters * 16 * 4 * (32.0 * 32 * 16 * 2);
    printf("{\"kernel\": \"mfma_f32_32x32x16_bf16 only\", \"waves_per_simd\": %d, \"ms\": %.3f, \"tflops\": %.1f}\n", wgs_per_cu, ms, fl / ms / 1e9);
  }
  {
    float* rnd;
    hipMalloc(&rnd, 4096 * sizeof(float));
    float h[4096];
    unsigned x = 12345;
    for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xffff) / 65536.0f * 2.f - 1.f + 1e-3f * (x & 255); }
    hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    const int grid2 = cus * 2;
    for (int rep = 0; rep < 2; ++rep) {
      double ms2 = time_ms([&] { hipLaunchKernelGGL(k_f32_random<4>, dim3(grid2), dim3(256), 0, 0, out, iters * (rep ? 20 : 1), rnd); }, 5);
      double fl2 = (double)grid2 * 4 * (iters * (rep ? 20.0 : 1.0)) * 16 * 4 * (32.0 * 32 * 2 * 2);
      printf("{\"kernel\": \"mfma_f32_32x32x2_f32 only, random operands%s\", \"waves_per_simd\": 2, \"ms\": %.3f, \"tflops\": %.1f}\n", rep ? ", sustained" : "", ms2, fl2 / ms2 / 1e9);
    }
  }
  // long run (~2 s) to see sustained clocks under power
  const int grid = cus * 2;
  double ms = time_ms([&] { hipLaunchKernelGGL(k_f32<4>, dim3(grid), dim3(256), 0, 0, out, iters * 20, 1.0f, 2.0f); }, 10);
  double fl = (double)grid * 4 * (iters * 20.0) * 16 * 4 * (32.0 * 32 * 2 * 2);
  printf("{\"kernel\": \"mfma_f32_32x32x2_f32 only, sustained\", \"waves_per_simd\": 2, \"ms\": %.3f, \"tflops\": %.1f}\n", ms, fl / ms / 1e9);
  return 0;
}
