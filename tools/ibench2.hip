// Per-instruction issue cost on gfx950 (experiment helper, not product code): inline-asm chains of ONE
// instruction, 8 independent accumulators per lane, 4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o tools/ibench2 tools/ibench2.hip && tools/ibench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 2048;
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed) {
  uint32_t a[8], b = threadIdx.x * 2654435761u + seed;
  uint64_t q[8];
  for (int i = 0; i < 8; ++i) { a[i] = b + i * 77; q[i] = ((uint64_t)(b ^ i) << 32) | (b + i); }
  uint32_t m = seed | 1;
  for (int it = 0; it < ITER; ++it) {
#define ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(m));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "s"(m));
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "s"(m));
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "s"(m) : "vcc");
#define SHR64(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q[i]));
#define CMP64(i) asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(q[i]), "v"(q[(i + 1) & 7]) : "vcc");
#define CMP32(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
#define ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b));
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(a[i]));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
#define SUBCO(i) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(b));
#define MAD32(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "s"(m));
    if (OP == 0) { REP8(ADD) }
    if (OP == 1) { REP8(MULLO) }
    if (OP == 2) { REP8(MULHI) }
    if (OP == 3) { REP8(MAD24) }
    if (OP == 4) { REP8(MAD64) }
    if (OP == 5) { REP8(SHR64) }
    if (OP == 6) { REP8(CMP64) }
    if (OP == 7) { REP8(CMP32) }
    if (OP == 8) { REP8(ALIGN) }
    if (OP == 9) { REP8(CNDM) }
    if (OP == 10) { REP8(BFE) }
    if (OP == 11) { REP8(LSHLADD) }
    if (OP == 12) { REP8(SUBCO) }
    if (OP == 13) { REP8(ADD3) }
    if (OP == 14) { REP8(BFI) }
  }
  uint32_t s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + (uint32_t)q[i] + (uint32_t)(q[i] >> 32);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char *name, uint32_t *d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 4;  // 4 WGs of 4 waves per CU → 4 waves per SIMD, one round
  hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 1u);
  hipEventRecord(e0);
  for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, 2u + r);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= 4;
  const double per_simd = 4.0 * ITER * 8;  // wave-instructions per SIMD
  printf("%-16s %.3f ms  → %.2f ns per wave-instr per SIMD\n", name, ms, ms * 1e6 / per_simd);
}
int main() {
  uint32_t *d; hipMalloc(&d, 256 * 4 * 256 * 4);
  run<0>("v_add_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_hi_u32", d); run<3>("v_mad_u32_u24", d);
  run<4>("v_mad_u64_u32", d); run<5>("v_lshrrev_b64", d); run<6>("v_cmp_lt_u64", d); run<7>("v_cmp_lt_u32", d);
  run<8>("v_alignbit_b32", d); run<9>("v_cndmask_b32", d); run<10>("v_bfe_u32", d); run<11>("v_lshl_add_u32", d);
  run<12>("v_sub_co_u32", d); run<13>("v_add3_u32", d); run<14>("v_bfi_b32", d);
  return 0;
}
