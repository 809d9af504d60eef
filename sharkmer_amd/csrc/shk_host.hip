// shk_host.hip — shk_run_files: ingest_reads + consolidate_and_histogram + write_stats for local files
// (src/main.rs:74-78, 112-197 without sPCR), feeding libshk's device path in super-batches through pinned buffers.
// The FASTQ front-end, the writers and the packer it drives are plain C++ (shk_front.cpp, shk_inflate.cpp — also
// built under sanitizers, `make san`); only what touches HIP (pinned buffers, device memory in use) lives here.
#include "../../include/shk.h"
#include "shk_front.h"

#include <hip/hip_runtime.h>
#include <sys/stat.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using shk::fmt;
#define g_run_error (shk::run_error())

extern "C" {

int shk_run_files(const shk_run_config *rc, shk_run_stats *out_stats) {
  if (!rc) return SHK_ERR_BAD_ARG;
  const bool trace = getenv("SHK_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  double t_mark = t_start;
  auto mark = [&](const char *what) {
    if (trace) {
      const double t = now();
      fprintf(stderr, "[shk] run_files: %-28s %7.2f ms (at %7.2f)\n", what, (t - t_mark) * 1e3, (t - t_start) * 1e3);
      t_mark = t;
    }
  };
  g_run_error.clear();
  int v = shk_validate_args(rc->k, rc->histo_max, rc->sample);
  if (v != SHK_OK) return v;
  std::string dir = rc->outdir ? rc->outdir : "./";  // main.rs:74-78
  if (dir.empty() || dir.back() != '/') dir += '/';
  {  // create_dir_all (main.rs:80-84)
    std::string acc;
    bool ok = true;
    for (size_t i = 0; i <= dir.size() && ok; ++i) {
      if (i == dir.size() || dir[i] == '/') {
        if (!acc.empty() && acc != "." && acc != "/" && mkdir(acc.c_str(), 0777) != 0 && errno != EEXIST) ok = false;
      }
      if (i < dir.size()) acc += dir[i];
    }
    struct stat st;
    if (!ok || stat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) {
      g_run_error = fmt("Failed to create output directory: %s", dir.c_str());
      return SHK_ERR_IO;
    }
  }
  shk_fastq *rd = nullptr;
  v = shk_fastq_open_ex(rc->inputs, rc->n_inputs, rc->max_reads, rc->validate_every, rc->fastq_flags, &rd);
  if (v != SHK_OK) return v;
  shk_config cfg{};
  cfg.k = rc->k;
  cfg.chunks = rc->chunks;
  cfg.histo_max = rc->histo_max;
  cfg.device = rc->device;
  cfg.table_capacity_hint = rc->table_capacity_hint;
  if (rc->n_devices > 1) {  // one context over several devices; a single device id is just that device
    cfg.n_devices = rc->n_devices;
    cfg.device_ids = rc->device_ids;
  } else if (rc->n_devices == 1 && rc->device_ids) {
    cfg.device = rc->device_ids[0];
  }
  shk_ctx *ctx = nullptr;
  v = shk_create(&cfg, &ctx);
  if (v != SHK_OK) {
    g_run_error = shk_last_error(nullptr);
    shk_fastq_close(rd);
    return v;
  }
  mark("reader open + context create");
  // Super-batches through pinned buffers, in the 2-bit PACKED input format: the front-end's copy-out threads pack the
  // sequences straight out of the parsed file (shk_fastq_next_batch_packed), so a batch crosses PCIe at 0.3 B per base
  // and is never written as ASCII on the host at all; the engine stripes by its own running read index, which is
  // exactly drain_batch's cadence (io.rs:340-343,355-361) whatever the batch size.  SHK_RUN_ASCII=1: the ASCII path.
  const bool ascii = getenv("SHK_RUN_ASCII") != nullptr || rc->n_devices > 1;  // (a multi-device context deals ASCII host batches to its devices)
  const uint64_t max_seqs = rc->batch_reads ? rc->batch_reads : 1000000;
  // (pinning host memory costs ≈ 45 µs per MB each way: two 256 MB buffers were 40 ms of a 1.2 Gbase job; batches of
  // 64 M bases keep every launch large enough, and a read that is longer gets larger buffers when it shows up)
  uint64_t cap_bases = rc->batch_bases ? rc->batch_bases : (64ull << 20);
  // two batch buffers: while the engine takes batch i (copy + count), the front-end fills batch i+1
  struct BatchBuf {
    uint8_t *data = nullptr;   // ASCII bases, or the packed stream
    uint32_t *nmask = nullptr;
    uint64_t *offs = nullptr;
  } bb2[2];
  auto data_bytes = [&](uint64_t cap) { return ascii ? cap : ((cap + 3) / 4 + 7) / 8 * 8 + 64; };
  auto alloc_buf = [&](BatchBuf &b, bool with_offs) {
    b.data = (uint8_t *)shk_alloc_pinned(data_bytes(cap_bases));
    if (!ascii) b.nmask = (uint32_t *)shk_alloc_pinned(((cap_bases + 31) / 32 + 16) * 4);
    if (with_offs) b.offs = (uint64_t *)shk_alloc_pinned((max_seqs + 1) * 8);
    return b.data && (ascii || b.nmask) && b.offs;
  };
  auto free_data = [&](BatchBuf &b) {
    shk_free_pinned(b.data);
    shk_free_pinned(b.nmask);
    b.data = nullptr;
    b.nmask = nullptr;
  };
  std::thread ingest_th;
  int ingest_rc = SHK_OK;
  auto join_ingest = [&]() {
    if (ingest_th.joinable()) ingest_th.join();
    return ingest_rc;
  };
  // Tearing down costs as much as a fifth of a 1.2-Gbase job — unmapping the file and joining the reader's threads
  // (7–12 ms), unpinning the batch buffers (5 ms), freeing the table (5 ms) — so the three go side by side, and the
  // reader is closed in the background as soon as its last batch is out (close_reader_early).
  std::thread close_th;
  uint64_t nrr = 0, nbr = 0;
  auto close_reader_early = [&]() {
    if (!rd) return;
    shk_fastq_stats(rd, &nrr, &nbr, nullptr, nullptr);
    shk_fastq *r0 = rd;
    rd = nullptr;
    close_th = std::thread([r0, trace, now] {
      const double t0 = now();
      shk_fastq_close(r0);
      if (trace) fprintf(stderr, "[shk] run_files:   (reader closed in %.2f ms)\n", (now() - t0) * 1e3);
    });
  };
  auto cleanup = [&]() {
    (void)join_ingest();
    close_reader_early();
    std::thread unpin([&] {
      const double t0 = now();
      for (int i = 0; i < 2; ++i) {
        free_data(bb2[i]);
        shk_free_pinned(bb2[i].offs);
      }
      if (trace) fprintf(stderr, "[shk] run_files:   (buffers unpinned in %.2f ms)\n", (now() - t0) * 1e3);
    });
    {
      const double t0 = now();
      shk_destroy(ctx);
      if (trace) fprintf(stderr, "[shk] run_files:   (context destroyed in %.2f ms)\n", (now() - t0) * 1e3);
    }
    unpin.join();
    if (close_th.joinable()) close_th.join();
    mark("  buffers unpinned, context destroyed, reader closed");
  };
  if (!alloc_buf(bb2[0], true)) {
    cleanup();
    g_run_error = "pinned buffer allocation failed";
    return SHK_ERR_NOMEM;
  }
  auto next_batch = [&](BatchBuf &b, uint64_t *n) {
    return ascii ? shk_fastq_next_batch(rd, b.data, cap_bases, b.offs, max_seqs, n)
                 : shk_fastq_next_batch_packed(rd, b.data, b.nmask, cap_bases, b.offs, max_seqs, n);
  };
  mark("first pinned buffers");
  for (int cur = 0;; cur ^= 1) {  // any batch size keeps the striping: the engine counts reads itself
    if (!bb2[cur].data) {  // (the second pair is only allocated when there is a second batch)
      if (!alloc_buf(bb2[cur], true)) {
        cleanup();
        g_run_error = "pinned buffer allocation failed";
        return SHK_ERR_NOMEM;
      }
    }
    uint64_t n = 0;
    v = next_batch(bb2[cur], &n);
    while (v == SHK_ERR_BAD_ARG && !rc->batch_bases && cap_bases < (16ull << 30) &&
           strstr(shk_fastq_error(rd), "longer than the batch buffer")) {  // a very long read: larger buffers, same call again
      if (join_ingest() != SHK_OK) break;  // (the other buffer was still being read by the engine; its outcome comes first, below)
      cap_bases *= 4;
      bool ok = true;
      for (int i = 0; i < 2 && ok; ++i) {
        if (!bb2[i].data) continue;  // (not in use yet: allocated at the new size when it is)
        free_data(bb2[i]);
        ok = alloc_buf(bb2[i], false);
      }
      if (!ok) {
        cleanup();
        g_run_error = "pinned buffer allocation failed";
        return SHK_ERR_NOMEM;
      }
      v = next_batch(bb2[cur], &n);
    }
    // the batch before this one: its outcome comes first, as in the reference, which ingests what it has read
    // before it reads on (io.rs:340-343)
    const int prev = join_ingest();
    if (prev != SHK_OK) {
      g_run_error = shk_last_error(ctx);
      cleanup();
      return prev;
    }
    if (v != SHK_OK) {
      g_run_error = shk_fastq_error(rd);
      cleanup();
      return v;
    }
    if (n) {
      const BatchBuf b = bb2[cur];
      ingest_rc = SHK_OK;
      if (ascii)
        ingest_th = std::thread([&ingest_rc, ctx, b, n] { ingest_rc = shk_ingest_reads(ctx, b.data, b.offs, n); });
      else
        ingest_th = std::thread([&ingest_rc, ctx, b, n] { ingest_rc = shk_ingest_packed(ctx, b.data, b.nmask, b.offs, n); });
    }
    int done = 0;
    shk_fastq_stats(rd, nullptr, nullptr, nullptr, &done);
    if (done) break;
  }
  mark("batches read and handed over");
  v = join_ingest();
  if (v != SHK_OK) {
    g_run_error = shk_last_error(ctx);
    cleanup();
    return v;
  }
  mark("last ingest joined");
  close_reader_early();  // (every batch is out and counted: the mapping and the reader's threads go while the histogram is written —
                         // not earlier: unmapping locks the address space, and the engine's last copies would wait for it)
  v = shk_finalize(ctx);
  if (v != SHK_OK) {
    g_run_error = shk_last_error(ctx);
    cleanup();
    return v;
  }
  mark("finalize");
  shk_counters cn{};
  shk_get_counters(ctx, &cn);
  const char *version = rc->version ? rc->version : "3.1.0";
  std::string sample = rc->sample;
  if (rc->chunks > 0) {  // io.rs:1051-1094
    std::vector<uint64_t> h((size_t)rc->chunks * (rc->histo_max + 2));
    shk_histograms(ctx, h.data());
    v = shk_write_histo((dir + sample + ".histo").c_str(), version, rc->k, rc->chunks, rc->histo_max, h.data());
    if (v == SHK_OK)
      v = shk_write_final_histo((dir + sample + ".final.histo").c_str(), version, rc->k, rc->chunks,
                                rc->histo_max, h.data());
    if (v != SHK_OK) {
      g_run_error = "Failed to create histogram file";
      cleanup();
      return v;
    }
  }
  shk_run_stats st{};
  st.sharkmer_version = version;
  st.command = rc->command ? rc->command : "";
  st.sample = rc->sample;
  st.kmer_length = rc->k;
  st.chunks = rc->chunks;
  st.n_reads_read = nrr;
  st.n_bases_read = nbr;
  st.n_subreads_ingested = cn.n_reads_ingested;
  st.n_bases_ingested = cn.n_bases_ingested;
  st.n_kmers = cn.n_kmers_ingested;
  st.has_histogram = rc->chunks > 0;
  st.n_singleton_kmers = cn.n_singleton_kmers;
  st.n_multi_kmers = cn.n_kmers_ingested >= cn.n_singleton_kmers ? cn.n_kmers_ingested - cn.n_singleton_kmers : 0;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) st.peak_memory_bytes = total_b - free_b;  // device bytes in use
  v = shk_write_stats_yaml((dir + sample + ".stats.yaml").c_str(), &st);
  if (out_stats) *out_stats = st;
  if (out_stats) {  // strings owned by the caller's config, not by us
    out_stats->sharkmer_version = nullptr;
    out_stats->command = nullptr;
    out_stats->sample = nullptr;
  }
  mark("output files");
  cleanup();
  mark("cleanup");
  if (v != SHK_OK) g_run_error = "Failed to create stats file";
  return v;
}

}  // extern "C"
