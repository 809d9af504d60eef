// Times the FASTQ front-end alone (shk_fastq_* of libshk), ASCII batches or packed ones:
//   g++ -O2 -o tools/reader_bench tools/reader_bench.cpp -Lsharkmer_amd/csrc -lshk -Wl,-rpath,$PWD/sharkmer_amd/csrc
//   tools/reader_bench [--packed] file...
#include "../include/shk.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int main(int argc, char **argv) {
  bool packed = false;
  std::vector<const char *> paths;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--packed")) packed = true;
    else paths.push_back(argv[i]);
  }
  const uint64_t cap = 64ull << 20, max_seqs = 1000000;
  uint8_t *bases = (uint8_t *)malloc(cap);
  uint32_t *nmask = (uint32_t *)malloc(cap / 8 + 64);
  uint64_t *offs = (uint64_t *)malloc((max_seqs + 1) * 8);
  memset(bases, 1, cap);  // (touched once: the timed runs do not pay for page faults)
  memset(nmask, 1, cap / 8 + 64);
  for (int rep = 0; rep < 4; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    shk_fastq *r = nullptr;
    shk_fastq_open(paths.data(), (uint32_t)paths.size(), 0, 0, &r);
    uint64_t tot = 0, reads = 0;
    for (;;) {
      uint64_t n = 0;
      int rc = packed ? shk_fastq_next_batch_packed(r, bases, nmask, cap, offs, max_seqs, &n) : shk_fastq_next_batch(r, bases, cap, offs, max_seqs, &n);
      if (rc != 0) { printf("error %s\n", shk_fastq_error(r)); return 1; }
      tot += offs[n];
      reads += n;
      int done = 0;
      shk_fastq_stats(r, nullptr, nullptr, nullptr, &done);
      if (done) break;
    }
    shk_fastq_close(r);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%s %llu reads %.3f s  %.2f Gbases/s\n", packed ? "packed" : "ascii", (unsigned long long)reads, dt, tot / dt / 1e9);
    fflush(stdout);
  }
  return 0;
}
