// front_driver — a native caller of the host-side C ABI (FASTQ front-end + packer), for the sanitizer builds that
// cannot live inside a Python process (ThreadSanitizer): reads the given files the way shk_run_files does — batch
// after batch — packs every batch, and prints one digest line a test compares with what the ordinary library gives
// through ctypes.  usage: front_driver MAX_READS VALIDATE_EVERY MAX_SEQS MAX_BASES file...
#include "../../include/shk.h"

#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
  if (argc < 6) return 2;
  const uint64_t max_reads = strtoull(argv[1], nullptr, 10), validate_every = strtoull(argv[2], nullptr, 10);
  const uint64_t max_seqs = strtoull(argv[3], nullptr, 10), max_bases = strtoull(argv[4], nullptr, 10);
  shk_fastq *r = nullptr;
  if (shk_fastq_open(argv + 5, (uint32_t)(argc - 5), max_reads, validate_every, &r) != SHK_OK) return 3;
  std::vector<uint8_t> bases(max_bases), packed(max_bases / 4 + 8);
  std::vector<uint32_t> nmask(max_bases / 32 + 8);
  std::vector<uint64_t> offs(max_seqs + 1);
  uint64_t h = 1469598103934665603ull, hp = h, n_total = 0;
  auto mix = [](uint64_t &x, const uint8_t *p, size_t n) {
    for (size_t i = 0; i < n; ++i) x = (x ^ p[i]) * 1099511628211ull;
  };
  for (;;) {
    uint64_t n = 0;
    const int rc = shk_fastq_next_batch(r, bases.data(), max_bases, offs.data(), max_seqs, &n);
    if (rc != SHK_OK) {
      printf("error %d %s\n", rc, shk_fastq_error(r));
      shk_fastq_close(r);
      return 0;
    }
    n_total += n;
    mix(h, bases.data(), offs[n]);
    for (uint64_t i = 0; i <= n; ++i) mix(h, (const uint8_t *)&offs[i], 8);
    const int prc = shk_pack_reads(bases.data(), offs[n], packed.data(), nmask.data(), 4);
    if (prc == SHK_OK) {
      mix(hp, packed.data(), (offs[n] + 3) / 4);
      mix(hp, (const uint8_t *)nmask.data(), (offs[n] + 31) / 32 * 4);
    } else {
      mix(hp, (const uint8_t *)shk_run_error(), 8);
    }
    int done = 0;
    shk_fastq_stats(r, nullptr, nullptr, nullptr, &done);
    if (done) break;
  }
  uint64_t nr = 0, nb = 0;
  int rm = 0;
  shk_fastq_stats(r, &nr, &nb, &rm, nullptr);
  printf("ok reads %" PRIu64 " bases %" PRIu64 " max %d hash %016" PRIx64 " packed %016" PRIx64 "\n", nr, nb, rm, h, hp);
  shk_fastq_close(r);
  return n_total == nr ? 0 : 4;
}
