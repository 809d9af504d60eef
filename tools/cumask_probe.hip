// Probe (round 4): how hipExtStreamCreateWithCUMask's bits map to (XCC, CU) on MI355X — which workgroups of a 2048-WG
// launch on a masked stream run where.  Prints, per mask, the number of distinct (xcc, se, cu) triples seen per XCC.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;} } while(0)
__global__ void k_where(uint32_t *out) {
  if (threadIdx.x == 0) {
    uint32_t xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
    uint32_t hwid = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((32 - 1) << 11));
    out[blockIdx.x * 2] = xcc;
    out[blockIdx.x * 2 + 1] = hwid;
  }
  // stay a while so that every enabled CU gets work
  unsigned long long t0 = clock64();
  while (clock64() - t0 < 200000) {}
}
int main() {
  const int G = 4096;
  uint32_t *d;
  CK(hipMalloc(&d, G * 8));
  std::vector<uint32_t> h(G * 2);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("multiProcessorCount %d\n", prop.multiProcessorCount);
  struct M { const char *name; std::vector<uint32_t> mask; };
  std::vector<M> masks;
  masks.push_back({"all 256", std::vector<uint32_t>(8, 0xFFFFFFFFu)});
  masks.push_back({"bits 0-31 only", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"bits 0-7 only", {0xFFu, 0, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"every 8th bit (0,8,16,...)", std::vector<uint32_t>(8, 0x01010101u)});
  masks.push_back({"all but bits 0-15", {0xFFFF0000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}});
  masks.push_back({"all but every 16th bit", std::vector<uint32_t>(8, 0xFFFEFFFEu)});
  for (auto &m : masks) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)m.mask.size(), m.mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", m.name, hipGetErrorString(e)); continue; }
    CK(hipMemsetAsync(d, 0xFF, G * 8, st));
    hipLaunchKernelGGL(k_where, dim3(G), dim3(256), 0, st, d);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost));
    std::map<uint32_t, std::set<uint32_t>> per;
    for (int i = 0; i < G; ++i) {
      uint32_t hw = h[2 * i + 1];
      uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      per[h[2 * i]].insert((se << 8) | (sh << 4) | cu);
    }
    size_t tot = 0;
    printf("%-28s:", m.name);
    for (auto &kv : per) { printf(" xcc%u=%zu", kv.first, kv.second.size()); tot += kv.second.size(); }
    printf("  total %zu CUs\n", tot);
    CK(hipStreamDestroy(st));
  }
  return 0;
}
