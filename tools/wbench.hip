// Micro-benchmark: partition-scatter write patterns on MI355X.
// G persistent workgroups each own P private output streams (contiguous runs); per "tile"
// every stream receives one chunk of C bytes.  Lanes write 8-B records; C/8 consecutive lanes
// cover one chunk (coalesced), different chunks go to different streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)

__global__ void __launch_bounds__(256) k_w(uint64_t* out, uint32_t P, uint32_t recs_per_chunk, uint32_t n_tiles, uint64_t run_len /*records per (g,p)*/) {
  const uint32_t g = blockIdx.x;
  const uint32_t chunks_per_iter = 256 / recs_per_chunk;
  for (uint32_t t = 0; t < n_tiles; ++t) {
    for (uint32_t c0 = 0; c0 < P; c0 += chunks_per_iter) {
      uint32_t c = c0 + threadIdx.x / recs_per_chunk;
      if (c >= P) break;
      uint32_t p = (c * 2654435761u + t * 40503u + g * 7u) % P;
      uint64_t base = ((uint64_t)p * gridDim.x + g) * run_len + (uint64_t)t * recs_per_chunk;
      out[base + threadIdx.x % recs_per_chunk] = base + threadIdx.x;
    }
  }
}

int main(int argc, char** argv) {
  uint32_t G = argc > 1 ? atoi(argv[1]) : 512;
  uint64_t total_recs = 128ull << 20;  // 1 GiB of 8-B records
  uint64_t* d; CK(hipMalloc(&d, total_recs * 8 + (1<<20)));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (uint32_t P : {512u, 2048u}) {
    for (uint32_t rpc : {1u, 2u, 4u, 8u, 16u, 32u, 64u}) {
      uint64_t run_len = total_recs / ((uint64_t)P * G);
      uint32_t n_tiles = run_len / rpc;
      if (n_tiles == 0) continue;
      float best = 1e9;
      for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_w, dim3(G), dim3(256), 0, 0, d, P, rpc, n_tiles, run_len);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
      }
      double bytes = (double)P * G * n_tiles * rpc * 8;
      printf("G=%u P=%u chunk=%4uB tiles=%u  %.3f ms  %.1f GB/s\n", G, P, rpc * 8, n_tiles, best, bytes / best / 1e6);
    }
  }
  return 0;
}
