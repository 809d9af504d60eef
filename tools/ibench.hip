// Issue-rate probe for a few VALU integer ops on gfx950 (experiment helper, not product code).
// hipcc --offload-arch=gfx950 -O3 -o tools/ibench tools/ibench.hip && tools/ibench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 4096;
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
  uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x55, d = a + 7;
  uint64_t q = ((uint64_t)a << 32) | b, r = ((uint64_t)c << 32) | d;
  for (int i = 0; i < ITER; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) { a = (a * b) ^ d; b = (b * c) ^ a; c = (c * d) ^ b; d = (d * a) ^ c; }  // mul_lo + xor
      if (OP == 1) { a = __umul24(a, b) ^ d; b = __umul24(b, c) ^ a; c = __umul24(c, d) ^ b; d = __umul24(d, a) ^ c; }  // mul24 + xor
      if (OP == 2) { a = (a ^ (a >> 15)) + 1; b = (b ^ (b >> 15)) + 1; c = (c ^ (c >> 15)) + 1; d = (d ^ (d >> 15)) + 1; }  // shift xor add
      if (OP == 3) { q = (q >> 3) + r; r = (r >> 5) + q; }  // 64-bit shift + add
      if (OP == 4) { a += (q < r); q += 0x100000001ull * a; r += 3; }  // 64-bit compare
      if (OP == 5) { a = __builtin_amdgcn_alignbit(a, b, 7); b = __builtin_amdgcn_alignbit(b, c, 9); c = __builtin_amdgcn_alignbit(c, d, 11); d = __builtin_amdgcn_alignbit(d, a, 13); }
      if (OP == 7) { a = __umulhi(a, b) ^ d; b = __umulhi(b, c) ^ a; c = __umulhi(c, d) ^ b; d = __umulhi(d, a) ^ c; }
      if (OP == 8) { a = (a ^ (a >> 15)) * 0x2C1B3C6Du; b = (b ^ (b >> 15)) * 0x2C1B3C6Du; c = (c ^ (c >> 15)) * 0x2C1B3C6Du; d = (d ^ (d >> 15)) * 0x2C1B3C6Du; }
      if (OP == 9) { a = __umul24(a ^ (a >> 15), 0x3779B1u); b = __umul24(b ^ (b >> 15), 0x3779B1u); c = __umul24(c ^ (c >> 15), 0x3779B1u); d = __umul24(d ^ (d >> 15), 0x3779B1u); }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (uint32_t)q + (uint32_t)r;
}
template <int OP>
void run(const char *name, double ops_per_iter, uint32_t *d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 4 * 4;  // 4 WGs of 4 waves per CU → 4 waves per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 2u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 16 waves over the run (4 resident at a time, 4 rounds)
  double wave_instr = (double)grid * 4 * ITER * 8 * ops_per_iter;  // wave-level instructions
  double per_simd = wave_instr / 1024;
  printf("%-22s %.3f ms  → %.2f ns per wave-instr per SIMD (%.1f cycles @2.4GHz)\n", name, ms,
         ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
  uint32_t *d; hipMalloc(&d, 256 * 4 * 4 * 256 * 4);
  run<0>("mul_lo_u32+xor (8)", 8, d);
  run<1>("mul_u32_u24+xor (8)", 8, d);
  run<2>("lshr,xor,add (12)", 12, d);
  run<3>("64b shr+add (2x~4)", 8, d);
  run<4>("64b cmp etc", 6, d);
  run<5>("alignbit (4)", 4, d);
  run<7>("mul_hi_u32+xor (8)", 8, d);
  run<8>("lshr,xor,mul_lo (12)", 12, d);
  run<9>("lshr,xor,mul24 (12)", 12, d);
  return 0;
}
