// Micro-benchmark (round 4): two ways to write a scatter tile's partition runs from LDS to HBM on MI355X.
// One persistent 1024-thread workgroup per CU; per tile every one of P = 1024 partitions gets a run of
// RUN 4-byte records at its own place (a 64-B slot per (workgroup, tile) in the partition's region).
//   mode 0 (pairs, today's k_scatter32): lane i of the workgroup stores pair i of the tile's sorted records —
//           8 consecutive lanes cover one run, one 8-byte store per lane and step, 8 steps per tile;
//   mode 1 (thread per partition): thread t copies partition t's run by itself — dword head up to a 16-B
//           boundary, 16-byte body, dword tail; `mis` shifts the runs by 0..3 records to get heads and tails;
//   mode 2: thread per partition with 8-byte stores only (aligned runs).
// Records come out of LDS in all modes (ds_read at the position the mode reads from).  WORK: VALU iterations
// per thread and tile between the write phases (what the stores can drain under).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while(0)
constexpr int NT = 1024, RUN = 16, P = 1024;

// block-interleaved regions as in libshk (rec_slot): 1024-record blocks of all P regions side by side
__device__ __forceinline__ uint64_t rec_at(uint32_t p, uint64_t at) { return (((at >> 10) * P + p) << 10) | (at & 1023u); }
template <int MODE>
__global__ void __launch_bounds__(NT, 4) k_w(uint32_t *out, uint32_t n_tiles_total, uint32_t slots_per_region, uint32_t work, uint32_t mis, uint32_t *sink) {
  extern __shared__ __attribute__((aligned(16))) uint32_t recs[];
  for (int i = threadIdx.x; i < P * RUN + 64; i += NT) recs[i] = i * 2654435761u;
  __syncthreads();
  uint32_t acc = threadIdx.x;
  for (uint32_t t = blockIdx.x; t < n_tiles_total; t += gridDim.x) {
    for (uint32_t i = 0; i < work; ++i) acc = acc * 1664525u + 1013904223u;
    const uint64_t slot = t;  // one 64-B (+ slack) slot per tile in every region
    if (MODE == 0) {
#pragma unroll 2
      for (uint32_t i = threadIdx.x; i < P * RUN / 2; i += NT) {
        const uint32_t p = i / (RUN / 2), j = i % (RUN / 2);
        const uint2 v = *reinterpret_cast<const uint2 *>(&recs[2 * i]);
        uint32_t *dst = out + rec_at(p, slot * 32u + 2 * j);
        *reinterpret_cast<uint2 *>(dst) = v;
      }
    } else if (MODE == 3) {  // one record per lane and step: 16 consecutive lanes cover one run, 16 steps per tile
#pragma unroll 4
      for (uint32_t i = threadIdx.x; i < P * RUN; i += NT) {
        const uint32_t p = i / RUN, j = i % RUN;
        out[rec_at(p, slot * 32u + j + (mis ? ((p * 7u + t * 3u) & 3u) : 0u))] = recs[i];
      }
    } else if (MODE == 1) {
      const uint32_t p = threadIdx.x;
      const uint32_t sh = mis ? ((p * 7u + t * 3u) & 3u) : 0u;
      uint32_t *dst = out + rec_at(p, slot * 32u + sh);
      const uint32_t *src = &recs[p * RUN];
      uint32_t c = RUN;
      uint32_t head = (4u - sh) & 3u;
      for (uint32_t j = 0; j < head; ++j) dst[j] = src[j];
      dst += head, src += head, c -= head;
      for (; c >= 4; c -= 4, dst += 4, src += 4) {
        uint4 v;
        v.x = src[0], v.y = src[1], v.z = src[2], v.w = src[3];
        *reinterpret_cast<uint4 *>(dst) = v;
      }
      for (uint32_t j = 0; j < c; ++j) dst[j] = src[j];
    } else {
      const uint32_t p = threadIdx.x;
      uint32_t *dst = out + rec_at(p, slot * 32u);
      const uint32_t *src = &recs[p * RUN];
#pragma unroll
      for (uint32_t j = 0; j < RUN; j += 2) *reinterpret_cast<uint2 *>(dst + j) = *reinterpret_cast<const uint2 *>(src + j);
    }
    __syncthreads();
  }
  if (acc == 12345u) sink[0] = acc;
}

int main(int argc, char **argv) {
  const uint32_t G = argc > 1 ? atoi(argv[1]) : 256;
  const uint32_t n_tiles = 7936;  // config 2: 130 M k-mers / 16 Ki
  const uint32_t slots = n_tiles;
  uint32_t *d, *sink;
  const size_t bytes = (size_t)P * ((size_t)slots * 32 + 1024) * 4 + (1 << 20);
  CK(hipMalloc(&d, bytes));
  CK(hipMalloc(&sink, 4));
  const size_t LDS = (size_t)(P * RUN + 64) * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_w<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_w<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_w<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_w<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (uint32_t work : {0u, 300u, 1200u}) {
    for (int mode = 0; mode < 6; ++mode) {
      float best = 1e9;
      for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(a));
        if (mode == 0) hipLaunchKernelGGL(k_w<0>, dim3(G), dim3(NT), LDS, 0, d, n_tiles, slots, work, 0u, sink);
        else if (mode == 1) hipLaunchKernelGGL(k_w<1>, dim3(G), dim3(NT), LDS, 0, d, n_tiles, slots, work, 0u, sink);
        else if (mode == 2) hipLaunchKernelGGL(k_w<1>, dim3(G), dim3(NT), LDS, 0, d, n_tiles, slots, work, 1u, sink);
        else if (mode == 3) hipLaunchKernelGGL(k_w<2>, dim3(G), dim3(NT), LDS, 0, d, n_tiles, slots, work, 0u, sink);
        else if (mode == 4) hipLaunchKernelGGL(k_w<3>, dim3(G), dim3(NT), LDS, 0, d, n_tiles, slots, work, 0u, sink);
        else hipLaunchKernelGGL(k_w<3>, dim3(G), dim3(NT), LDS, 0, d, n_tiles, slots, work, 1u, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      const char *names[] = {"pairs (8 B, coalesced runs)", "thread/partition 16 B aligned", "thread/partition 16 B + head/tail", "thread/partition 8 B",
                             "single records (4 B, coalesced runs)", "single records, runs shifted 0-3"};
      const double rec_bytes = (double)n_tiles * P * RUN * 4;
      printf("G=%u work=%4u  %-36s %.3f ms  %.0f GB/s of records\n", G, work, names[mode], best, rec_bytes / best / 1e6);
    }
  }
  return 0;
}
