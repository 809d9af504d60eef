// Times the FASTQ front-end alone (shk_fastq_* of libshk): g++ -O2 -o tools/reader_bench tools/reader_bench.cpp -Lsharkmer_amd/csrc -lshk -Wl,-rpath,$PWD/sharkmer_amd/csrc
#include "../include/shk.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv) {
  std::vector<const char *> paths(argv + 1, argv + argc);
  const uint64_t cap = 256ull << 20, max_seqs = 1000000;
  uint8_t *bases = (uint8_t *)malloc(cap);
  uint64_t *offs = (uint64_t *)malloc((max_seqs + 1) * 8);
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    shk_fastq *r = nullptr;
    shk_fastq_open(paths.data(), (uint32_t)paths.size(), 0, 0, &r);
    uint64_t tot = 0, reads = 0;
    for (;;) {
      uint64_t n = 0;
      int rc = shk_fastq_next_batch(r, bases, cap, offs, max_seqs, &n);
      if (rc != 0) { printf("error %s\n", shk_fastq_error(r)); return 1; }
      tot += offs[n];
      reads += n;
      int done = 0;
      shk_fastq_stats(r, nullptr, nullptr, nullptr, &done);
      if (done) break;
    }
    shk_fastq_close(r);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%llu reads %.3f s  %.2f Gbases/s\n", (unsigned long long)reads, dt, tot / dt / 1e9);
  }
  return 0;
}
