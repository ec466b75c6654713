// Calibration for the split-precision GEMM on PRE-SPLIT operands: per k-tile (K = 32) a 128x128 workgroup tile with
// NP bf16 planes per operand (NP = 2: bf16x3, NP = 3: bf16x6), LDS-DMA staging (no VALU split in the loop),
// fragment reads from LDS, TERMS MFMA products per fragment pair, one barrier per k-tile, 2 workgroups per CU.
// Reports bf16 MFMA TFLOP/s and the fp32-equivalent rate (MFMA rate / TERMS).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NP, int TERMS, int BK, bool STREAM = false>
__global__ __launch_bounds__(256) void loop_kernel(const short* __restrict__ src, float* __restrict__ out, int nk) {
  // per stage: A planes NP x [128 rows][BK bf16] + B planes the same
  constexpr int PLANE = 128 * BK;                 // shorts
  constexpr int STAGE = 2 * NP * PLANE;           // shorts
  constexpr int PIECES = STAGE * 2 / 1024;        // 1 KiB DMA pieces per stage
  constexpr int PER_WAVE = PIECES / 4;
  constexpr int KS = BK / 16;                     // MFMA k-steps per k-tile
  __shared__ __attribute__((aligned(1024))) short smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, hi = lane >> 5;
  for (int i = tid; i < 2 * STAGE; i += 256) smem[i] = (short)(0x3c00 + (i & 63));
  __syncthreads();
  f32x16 acc[2][2];
  for (int t = 0; t < 2; ++t) for (int u = 0; u < 2; ++u) for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  const short* gp = src + ((size_t)blockIdx.x * 256 + tid) * 8;
  // STREAM: operands really travel -- per k-tile this workgroup's A block (NP*128*BK bf16, contiguous: K-blocked
  // layout) is unique, the B block is shared by all workgroups (weights), both advance every k-tile
  constexpr int OPB = NP * 128 * BK;              // shorts per operand block and k-tile
  const short* ga = src + (size_t)blockIdx.x * nk * OPB;
  const short* gb = src + (size_t)gridDim.x * nk * OPB;
  // fragment addresses: row (wm*64 + t*32 + li), 16-byte chunk (ks*2 + hi) of a BK*2-byte row
  const int a_row = (wm * 64 + li) * BK + hi * 8;
  const int b_row = NP * PLANE + (wn * 64 + li) * BK + hi * 8;
  for (int kt = 0; kt < nk; ++kt) {
    short* nxt = smem + ((kt + 1) & 1) * STAGE;
    const short* cur = smem + (kt & 1) * STAGE;
    if (STREAM) {
#pragma unroll
      for (int j = 0; j < PER_WAVE; ++j) {
        const int piece = wave * PER_WAVE + j;     // pieces [0, PIECES/2) = A block, the rest = B block
        const short* g = (piece < PIECES / 2 ? ga + (size_t)kt * OPB + piece * 512 : gb + (size_t)kt * OPB + (piece - PIECES / 2) * 512) + lane * 8;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(nxt + piece * 512), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < PER_WAVE; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(gp + (size_t)j * 2048), (lptr_t)(nxt + (wave * PER_WAVE + j) * 512), 16, 0, 0);
    }
    bf16x8 af[NP][KS][2], bf[NP][KS][2];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          af[p][ks][t] = *reinterpret_cast<const bf16x8*>(cur + p * PLANE + a_row + t * 32 * BK + ks * 16);
          bf[p][ks][t] = *reinterpret_cast<const bf16x8*>(cur + p * PLANE + b_row + t * 32 * BK + ks * 16);
        }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          // TERMS products per fragment pair: (0,0) (0,1) (1,0) [bf16x3]; + (1,1) (0,2) (2,0) [bf16x6]
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][ks][t], bf[0][ks][u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][ks][t], bf[1 % NP][ks][u], acc[t][u], 0, 0, 0);
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1 % NP][ks][t], bf[0][ks][u], acc[t][u], 0, 0, 0);
          if (TERMS >= 6) {
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1 % NP][ks][t], bf[1 % NP][ks][u], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][ks][t], bf[2 % NP][ks][u], acc[t][u], 0, 0, 0);
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2 % NP][ks][t], bf[0][ks][u], acc[t][u], 0, 0, 0);
          }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float s = 0;
  for (int t = 0; t < 2; ++t) for (int u = 0; u < 2; ++u) for (int r = 0; r < 16; ++r) s += acc[t][u][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int NP, int TERMS, int BK, bool STREAM = false>
void run(const char* name, const short* src, float* out, int cus) {
  const int nk = (STREAM ? 128 : 512) * 32 / BK;
  const int grid = cus * 2 * 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((loop_kernel<NP, TERMS, BK, STREAM>), dim3(grid), dim3(256), 0, 0, src, out, nk);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((loop_kernel<NP, TERMS, BK, STREAM>), dim3(grid), dim3(256), 0, 0, src, out, nk);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  const double mfma_fl = (double)grid * 4 * nk * (BK / 16) * 4 * TERMS * (32.0 * 32 * 16 * 2);
  printf("{\"variant\": \"%s\", \"ms\": %.3f, \"bf16_mfma_tflops\": %.1f, \"fp32_equivalent_tflops\": %.1f}\n", name, ms,
         mfma_fl / ms / 1e9, mfma_fl / TERMS / ms / 1e9);
}

int main() {
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  short* src;
  float* out;
  // STREAM needs (grid + 1) * nk * OPB shorts: 2049 * 256 * 6144 * 2 B = 6.4 GB at BK = 16, NP = 3
  (void)hipMalloc(&src, (size_t)7 << 30);
  (void)hipMemset(src, 0x3c, (size_t)7 << 30);
  (void)hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
  run<2, 3, 32>("bf16x3, 2 planes, BK=32", src, out, cus);
  run<2, 3, 64>("bf16x3, 2 planes, BK=64", src, out, cus);
  run<3, 6, 32>("bf16x6, 3 planes, BK=32", src, out, cus);
  run<3, 6, 16>("bf16x6, 3 planes, BK=16", src, out, cus);
  run<2, 3, 16>("bf16x3, 2 planes, BK=16", src, out, cus);
  run<3, 6, 16, true>("bf16x6, 3 planes, BK=16, operands streamed from HBM/L2", src, out, cus);
  run<2, 3, 16, true>("bf16x3, 2 planes, BK=16, operands streamed from HBM/L2", src, out, cus);
  run<2, 3, 32, true>("bf16x3, 2 planes, BK=32, operands streamed from HBM/L2", src, out, cus);
  return 0;
}
