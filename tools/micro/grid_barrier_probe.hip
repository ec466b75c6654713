// How much does a hand-rolled grid-wide barrier cost on MI355X?  (round 5: the cooperative-groups grid.sync() of this runtime
// measured tens of microseconds, profiles/r7k.)  One monotonically increasing counter; every workgroup adds 1 and spins until the
// counter reaches nblocks * phase.  hipcc --offload-arch=gfx950 -O3 -o /tmp/gbp tools/micro/grid_barrier_probe.hip && /tmp/gbp
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void probe(unsigned* counter, float* data, int nbar) {
  float v = data[blockIdx.x * 256 + threadIdx.x];
  for (int b = 0; b < nbar; ++b) {
    data[blockIdx.x * 256 + threadIdx.x] = v + 1.0f;
    grid_barrier(counter, gridDim.x * (b + 1));
    v = data[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];      // something another workgroup wrote
  }
  data[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
  unsigned* counter; float* data;
  hipMalloc(&counter, 4); hipMalloc(&data, 1024 * 256 * 4);
  hipMemset(data, 0, 1024 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {75, 128, 256, 512}) {
    for (int nbar : {0, 1, 8, 32}) {
      float best = 1e9;
      for (int rep = 0; rep < 20; ++rep) {
        hipMemset(counter, 0, 4);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, counter, data, nbar);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("grid %4d barriers %3d: %.1f us\n", grid, nbar, best * 1e3);
    }
  }
  float h[256]; hipMemcpy(h, data, sizeof(h), hipMemcpyDeviceToHost); printf("check %.0f\n", h[0]);
  return 0;
}
