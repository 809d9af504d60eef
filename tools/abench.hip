// Micro-benchmark: per-tile reservation with returning global atomics on a small contiguous
// cursor array (P entries), from G persistent workgroups, T tiles in total.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)
__global__ void __launch_bounds__(512) k_res(unsigned int* cur, unsigned int* sink, unsigned P, unsigned tiles, unsigned work) {
  __shared__ unsigned base[4096];
  unsigned acc = 0;
  for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
    // some ALU work standing in for the rest of the tile (~work iterations)
    unsigned x = threadIdx.x + t;
    for (unsigned i = 0; i < work; ++i) x = x * 1664525u + 1013904223u;
    acc += x;
    __syncthreads();
    for (unsigned p = threadIdx.x; p < P; p += 512) base[p] = atomicAdd(&cur[p], 16u);
    __syncthreads();
    acc += base[(threadIdx.x * 7) % P];
  }
  if (acc == 12345) sink[0] = acc;
}
int main(int argc, char** argv) {
  unsigned G = argc > 1 ? atoi(argv[1]) : 512;
  unsigned *cur, *sink; CK(hipMalloc(&cur, 4096 * 4 * 8)); CK(hipMalloc(&sink, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (unsigned P : {1024u, 2048u}) for (unsigned work : {0u, 2000u, 8000u}) {
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
      CK(hipMemset(cur, 0, 4096 * 4));
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k_res, dim3(G), dim3(512), 0, 0, cur, sink, P, 9155u, work);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("G=%u P=%u work=%u: %.3f ms for 9155 tiles (%.1f M atomics)\n", G, P, work, best, 9155.0 * P / 1e6);
  }
  return 0;
}
