// A torch-free host of the C ABI (include/peanut_hip.h): what a C/C++ maintainer of the reference would write to call
// the map-prediction forward -- hipMalloc'd buffers, a hipStream_t, plain structs.  Driven by
// tests/test_c_host_gpu.py, which checks that the output equals the Python path's bit for bit.
//
//   pred_host weights.bin input.bin output.bin B C H W apply_sigmoid precision
//
// weights.bin: int32 n; then n x { int32 name_len; char name[name_len]; int32 ndim; int64 shape[4]; float data[prod] }
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "peanut_hip.h"

#define CHECK_HIP(e)                                                        \
  do {                                                                      \
    hipError_t _e = (e);                                                    \
    if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); return 2; } \
  } while (0)
#define CHECK_PEANUT(e)                                                     \
  do {                                                                      \
    int _rc = (e);                                                          \
    if (_rc != 0) { fprintf(stderr, "%s -> %d: %s\n", #e, _rc, peanut_last_error()); return 3; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 10) { fprintf(stderr, "usage: pred_host weights.bin input.bin output.bin B C H W apply_sigmoid precision\n"); return 1; }
  const int B = atoi(argv[4]), C = atoi(argv[5]), H = atoi(argv[6]), W = atoi(argv[7]), sig = atoi(argv[8]), prec = atoi(argv[9]);
  if (std::string(peanut_build_arch()) != "gfx950" || peanut_abi_version() < 3) { fprintf(stderr, "unexpected library\n"); return 1; }

  // ---- state dict ----
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int32_t n = 0;
  if (fread(&n, 4, 1, f) != 1) return 1;
  std::vector<std::string> names(n);
  std::vector<std::vector<float>> data(n);
  std::vector<peanut_tensor> tensors(n);
  for (int i = 0; i < n; ++i) {
    int32_t len = 0, ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    if (fread(&len, 4, 1, f) != 1) return 1;
    names[i].resize(len);
    if (fread(&names[i][0], 1, len, f) != (size_t)len || fread(&ndim, 4, 1, f) != 1 || fread(shape, 8, 4, f) != 4) return 1;
    size_t cnt = 1;
    for (int d = 0; d < ndim; ++d) cnt *= (size_t)shape[d];
    data[i].resize(cnt);
    if (fread(data[i].data(), 4, cnt, f) != cnt) return 1;
    tensors[i].name = names[i].c_str();
    tensors[i].data = data[i].data();
    tensors[i].ndim = ndim;
    for (int d = 0; d < 4; ++d) tensors[i].shape[d] = shape[d];
  }
  fclose(f);

  // ---- nav/pred_model_cfg.py:2-42 ----
  peanut_pred_cfg cfg = {};
  cfg.in_channels = C; cfg.num_classes = 6;
  const int strides[4] = {1, 2, 1, 1}, dil[4] = {1, 1, 2, 4}, pools[4] = {1, 2, 3, 6};
  for (int i = 0; i < 4; ++i) { cfg.strides[i] = strides[i]; cfg.dilations[i] = dil[i]; cfg.pool_scales[i] = pools[i]; }
  cfg.contract_dilation = 1; cfg.n_pool_scales = 4; cfg.head_channels = 512; cfg.align_corners = 0; cfg.bn_eps = 1e-5f;
  cfg.precision = prec; cfg.fold_ppm = 1; cfg.conv_algo = PEANUT_ALGO_AUTO;

  peanut_pred_t* h = nullptr;
  CHECK_PEANUT(peanut_pred_create(&h, &cfg, tensors.data(), n));

  const size_t in_n = (size_t)B * C * H * W, out_n = (size_t)B * 6 * H * W;
  std::vector<float> in(in_n), out(out_n);
  f = fopen(argv[2], "rb");
  if (!f || fread(in.data(), 4, in_n, f) != in_n) { fprintf(stderr, "bad input file\n"); return 1; }
  fclose(f);
  float *d_in = nullptr, *d_out = nullptr;
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  CHECK_HIP(hipMalloc(&d_in, in_n * 4));
  CHECK_HIP(hipMalloc(&d_out, out_n * 4));
  CHECK_HIP(hipMemcpyAsync(d_in, in.data(), in_n * 4, hipMemcpyHostToDevice, stream));
  const size_t ws = peanut_pred_workspace_bytes(h, B, H, W);
  if (ws == 0) { fprintf(stderr, "workspace query failed: %s\n", peanut_last_error()); return 3; }
  for (int rep = 0; rep < 2; ++rep) CHECK_PEANUT(peanut_pred_forward(h, d_in, d_out, B, H, W, sig, stream));
  CHECK_HIP(hipMemcpyAsync(out.data(), d_out, out_n * 4, hipMemcpyDeviceToHost, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  f = fopen(argv[3], "wb");
  if (!f || fwrite(out.data(), 4, out_n, f) != out_n) { fprintf(stderr, "cannot write output\n"); return 1; }
  fclose(f);
  printf("ok workspace_bytes=%zu\n", ws);
  peanut_pred_destroy(h);
  (void)hipFree(d_in); (void)hipFree(d_out); (void)hipStreamDestroy(stream);
  return 0;
}
