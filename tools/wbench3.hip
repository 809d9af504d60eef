// Micro-benchmark (round 4): what ONE CU's memory pipeline takes per wave-level store instruction when HBM is not
// the limit (few workgroups: 32 of 256 CUs busy).  One 1024-thread workgroup per CU, every thread issues `steps`
// stores per tile from registers, runs of RUN records (4 B each) go to 1024 block-interleaved regions like
// k_scatter32's.  Modes: width 1 / 2 / 4 records per lane; `half`: only lanes whose record index % 24 < 14 are
// active (the fixed-slot write-out's 58 % occupancy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)
constexpr int NT = 1024, P = 1024;
__device__ __forceinline__ uint64_t rec_at(uint32_t p, uint64_t at) { return (((at >> 10) * P + p) << 10) | (at & 1023u); }

template <int W, bool HALF>
__global__ void __launch_bounds__(NT, 4) k_w(uint32_t *out, uint32_t n_tiles, uint32_t steps, uint32_t run, uint32_t *sink) {
  uint32_t acc = threadIdx.x;
  for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    for (uint32_t s = 0; s < steps; ++s) {
      const uint32_t i = (threadIdx.x + s * NT) * W;  // first record of this lane
      const uint32_t p = (i / run) & (P - 1), j = i % run;
      const uint64_t at = (uint64_t)t * 64 + j;
      if (HALF && (i % 24) >= 14) continue;
      uint32_t *dst = out + rec_at(p, at);
      if (W == 1) *dst = acc;
      else if (W == 2) *reinterpret_cast<uint2 *>(dst) = make_uint2(acc, acc);
      else *reinterpret_cast<uint4 *>(dst) = make_uint4(acc, acc, acc, acc);
    }
    acc = acc * 1664525u + 1013904223u;
    __syncthreads();
  }
  if (acc == 12345u) sink[0] = acc;
}

int main(int argc, char **argv) {
  const uint32_t G = argc > 1 ? atoi(argv[1]) : 32;
  const uint32_t n_tiles = 31 * G * 4;
  uint32_t *d, *sink;
  CK(hipMalloc(&d, (size_t)P * ((size_t)n_tiles * 64 + 2048) * 4));
  CK(hipMalloc(&sink, 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  struct Cfg { int w; bool half; uint32_t steps, run; const char *name; };
  const Cfg cfgs[] = {
      {1, false, 16, 16, "16 x dword, full lanes, runs of 16"},
      {1, true, 24, 24, "24 x dword, 58% lanes (slots of 24)"},
      {2, false, 8, 16, " 8 x dwordx2, full lanes, runs of 16"},
      {4, false, 4, 16, " 4 x dwordx4, full lanes, runs of 16"},
      {1, false, 32, 16, "32 x dword (2 tiles' worth)"},
      {4, false, 8, 16, " 8 x dwordx4 (2 tiles' worth)"},
  };
  for (const Cfg &c : cfgs) {
    float best = 1e9;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(a));
#define L(W, H) hipLaunchKernelGGL((k_w<W, H>), dim3(G), dim3(NT), 0, 0, d, n_tiles, c.steps, c.run, sink)
      if (c.w == 1 && !c.half) L(1, false);
      else if (c.w == 1) L(1, true);
      else if (c.w == 2) L(2, false);
      else L(4, false);
      CK(hipEventRecord(b));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best) best = ms;
    }
    const double tiles_per_wg = (double)n_tiles / G;
    const double us_per_tile = best * 1e3 / tiles_per_wg;
    printf("G=%u  %-40s %.3f ms  %.2f us per tile  = %.0f cycles @2.4GHz per store instruction per CU\n", G, c.name, best, us_per_tile,
           us_per_tile * 2400.0 / (c.steps * 16.0));
  }
  return 0;
}
