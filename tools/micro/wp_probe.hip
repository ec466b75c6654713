// Stand-alone probe of the persistent 256 x 256 kernel (csrc/conv_pw256wp.hip, included as source): times one pointwise layer and,
// built with -DPEANUT_WP_TRACE, prints the length of every k-loop iteration of a few workgroups by kind (first / steady / last) from
// s_memtime stamps -- where a tile's time goes.  No correctness check here (tests/test_conv_gpu.py holds the kernel to F.conv2d).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I peanut_amd/csrc [-DPEANUT_WP_TRACE] [-DWP_VARIANT=n] tools/micro/wp_probe.hip -o tools/micro/build/wp_probe
//   wp_probe M K N residual(0/1) [reps] [c2]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../peanut_amd/csrc/conv_pw256wp.hip"

namespace peanut {
int fail(int code, const std::string& msg) { fprintf(stderr, "fail %d: %s\n", code, msg.c_str()); return code; }
void note_kernel(const char*) {}
}  // namespace peanut

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float lo, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f;      // [-1, 1)
    const float v = u * scale;
    p[i] = v < lo ? lo : v;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: wp_probe M K N residual [reps] [c2]\n"); return 2; }
  const long long M = atoll(argv[1]);
  const int K = atoi(argv[2]), N = atoi(argv[3]), has_res = atoi(argv[4]);
  const int reps = argc > 5 ? atoi(argv[5]) : 20;
  const int c2 = argc > 6 ? atoi(argv[6]) : 0;
  const int c1 = K - c2;
  float *x, *x2 = nullptr, *w, *ss, *res = nullptr, *y, *ws, *zeros;
  const size_t ws_floats = (size_t)48 << 20;
  CK(hipMalloc(&x, (size_t)M * c1 * 4));
  if (c2) CK(hipMalloc(&x2, (size_t)M * c2 * 4));
  CK(hipMalloc(&w, (size_t)N * K * 4));
  CK(hipMalloc(&ss, (size_t)2 * N * 4));
  if (has_res) CK(hipMalloc(&res, (size_t)M * N * 4));
  CK(hipMalloc(&y, (size_t)M * N * 4));
  CK(hipMalloc(&ws, ws_floats * 4 + (1 << 20)));
  CK(hipMalloc(&zeros, 4096));
  CK(hipMemset(zeros, 0, 4096));
  CK(hipMemset(ws, 0, ws_floats * 4 + (1 << 20)));
  fill_kernel<<<2048, 256>>>(x, (size_t)M * c1, 1u, 0.f, 1.7f);            // post-ReLU-like
  if (c2) fill_kernel<<<2048, 256>>>(x2, (size_t)M * c2, 2u, 0.f, 1.7f);
  fill_kernel<<<2048, 256>>>(w, (size_t)N * K, 3u, -10.f, 1.7f * sqrtf(2.0f / K));
  fill_kernel<<<64, 256>>>(ss, (size_t)N, 4u, 0.5f, 1.5f);
  fill_kernel<<<64, 256>>>(ss + N, (size_t)N, 5u, -1.f, 0.1f);
  if (has_res) fill_kernel<<<2048, 256>>>(res, (size_t)M * N, 6u, -10.f, 1.7f);
  CK(hipDeviceSynchronize());

  peanut::ConvKParams p{};
  p.x = x; p.x2 = x2 ? x2 : x; p.w = w; p.scale = ss; p.shift = ss + N; p.res = res; p.zeros = zeros; p.y = y;
  p.H = 1; p.W = (int)M; p.c1 = c1; p.c2 = c2; p.Ho = 1; p.Wo = (int)M; p.cout = N;
  p.kw = 1; p.ntaps = 1; p.stride = 1; p.pad = 0; p.dil = 1; p.relu = 1;
  p.HoWo = (int)M; p.M = (int)M; p.nkt = K / 32; p.ntiles = N / 128;
  p.split_p = 1; p.alpha = 1.f;
  if (!peanut::conv_pw_uses_256wp(N, M, 1, 0, 128, c1, c2, 0)) { fprintf(stderr, "shape not eligible\n"); return 3; }
  for (int i = 0; i < 3; ++i)
    if (int rc = peanut::launch_conv_pw256wp(p, ws, ws_floats, 0)) { fprintf(stderr, "launch rc %d\n", rc); return 4; }
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) peanut::launch_conv_pw256wp(p, ws, ws_floats, 0);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("{\"M\": %lld, \"K\": %d, \"N\": %d, \"res\": %d, \"c2\": %d, \"ms\": %.4f, \"tflops\": %.1f, \"variant\": %d}\n", M, K, N, has_res, c2, ms,
         2.0 * M * K * N / ms / 1e9,
#ifdef WP_VARIANT
         WP_VARIANT
#else
         0
#endif
  );
#ifdef PEANUT_WP_TRACE
  {
    // the kernel's trace area: behind the scratch proper; workgroup b < 16 wrote (stamp << 2 | kind) per iteration at [b * 1024 ...]
    std::vector<long long> tr(16 * 1024);
    CK(hipMemcpy(tr.data(), ws + ws_floats, tr.size() * 8, hipMemcpyDeviceToHost));
    double sum[3] = {0, 0, 0};
    long long cnt[3] = {0, 0, 0};
    for (int b = 0; b < 16; ++b) {
      const long long* t = tr.data() + b * 1024;
      const int n = (int)t[0];
      for (int i = 2; i <= n && i < 1024; ++i) {
        const int kind = (int)(t[i] & 3);
        const long long d = (t[i] >> 2) - (t[i - 1] >> 2);
        sum[kind] += (double)d; cnt[kind]++;
      }
      if (b == 0) {
        printf("wg0 iterations (kind:ticks): ");
        for (int i = 2; i <= n && i < 80; ++i) printf("%d:%lld ", (int)(t[i] & 3), (t[i] >> 2) - (t[i - 1] >> 2));
        printf("\n");
      }
    }
    printf("{\"trace\": \"mean s_memtime ticks per iteration by kind\", \"first\": %.1f, \"steady\": %.1f, \"last\": %.1f, \"n\": [%lld, %lld, %lld]}\n",
           cnt[0] ? sum[0] / cnt[0] : 0, cnt[1] ? sum[1] / cnt[1] : 0, cnt[2] ? sum[2] / cnt[2] : 0, cnt[0], cnt[1], cnt[2]);
  }
#endif
  return 0;
}
