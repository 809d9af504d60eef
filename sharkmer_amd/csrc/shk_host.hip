// shk_host.hip — the host side that sits either side of the counting path,
// in C++ because the reference's host is compiled code (Rust is absent from the image):
//
//   * FASTQ(.gz) front-end restating read_fastq / open_fastq_reader / validate_fastq_record
//     (src/io.rs:161-198, 271-352, 598-625): 4 lines per record, CRLF tolerant, record 0 and
//     every validate_every-th record validated, identical error texts, state carried across
//     files so 1000-read batches span file boundaries (io.rs:498-512), --max-reads cut-off
//     (io.rs:345-348).  It parses; it never counts.
//   * writers for {sample}.histo, {sample}.final.histo (io.rs:1009-1014, 1051-1094) and
//     {sample}.stats.yaml (stats.rs:27-45,186-193; field order of RunStats).
//   * shk_run_files: ingest_reads + consolidate_and_histogram + write_stats for local files
//     (main.rs:112-197 without sPCR), feeding libshk's device path in super-batches through
//     pinned buffers.
//
// Compiled into libshk.so together with shk_engine.hip; everything here is plain host code.
#include "../../include/shk.h"

#include <hip/hip_runtime.h>
#include <zlib.h>

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

std::string fmt(const char *f, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  return buf;
}

}  // namespace

// ---- FASTQ reader --------------------------------------------------------------------------------
// Read-ahead: a background thread does the gzread()s (file I/O + inflate, the slowest stage of the
// host front-end) into a small ring of blocks while the caller's thread splits lines, fills the
// batch and feeds the GPU.  The parser's view is unchanged: a stream of bytes, EOF, or a read error.
struct ReadAhead {
  static constexpr int NB = 4;
  static constexpr size_t BLOCK = 4u << 20;
  std::vector<char> blk[NB];
  int len[NB];                 // bytes in the block; 0 = EOF marker; < 0 = gzread error
  unsigned head = 0, tail = 0; // produced / consumed block counts
  bool stop = false;
  std::mutex m;
  std::condition_variable cv;
  std::thread th;
  gzFile f = nullptr;
  int fd = -1;  // ≥ 0: plain file, read() straight into the blocks (no pass through zlib's buffer)

  void start(gzFile file, int raw_fd) {
    f = file;
    fd = raw_fd;
    head = tail = 0;
    stop = false;
    for (auto &b : blk)
      if (b.size() != BLOCK) b.resize(BLOCK);
    th = std::thread([this] {
      for (;;) {
        unsigned slot;
        {
          std::unique_lock<std::mutex> lk(m);
          cv.wait(lk, [this] { return stop || head - tail < (unsigned)NB; });
          if (stop) return;
          slot = head % NB;
        }
        int n;
        if (fd >= 0) {
          ssize_t got = 0, r = 0;  // fill the block: short reads would only shrink the blocks
          while ((size_t)got < BLOCK && (r = ::read(fd, blk[slot].data() + got, BLOCK - (size_t)got)) > 0) got += r;
          n = r < 0 ? -1 : (int)got;
        } else {
          n = gzread(f, blk[slot].data(), (unsigned)BLOCK);
        }
        {
          std::lock_guard<std::mutex> lk(m);
          len[slot] = n;
          ++head;
        }
        cv.notify_all();
        if (n <= 0) return;  // EOF or error: nothing more to produce
      }
    });
  }
  // next block for the consumer (releases the previous one); n = its length, 0 = EOF, < 0 = error
  const char *next(bool release_prev, int *n) {
    std::unique_lock<std::mutex> lk(m);
    if (release_prev) {
      ++tail;
      cv.notify_all();
    }
    cv.wait(lk, [this] { return head > tail; });
    const unsigned slot = tail % NB;
    *n = len[slot];
    return blk[slot].data();
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

struct shk_fastq {
  std::vector<std::string> paths;
  size_t file_idx = 0;
  gzFile f = nullptr;
  int fd = -1;  // plain (not gzip) regular file: read without zlib
  std::string cur_name;
  ReadAhead ra;
  const char *buf = nullptr;  // current read-ahead block
  bool have_block = false;
  size_t buf_pos = 0, buf_len = 0;
  bool file_eof = false;
  uint64_t max_reads = 0, validate_every = 0;
  uint64_t n_reads_read = 0, n_bases_read = 0;  // FastqReadState, io.rs:205-206
  bool reached_max = false, done = false, pending = false;  // pending: line[] holds an undelivered record
  std::string err;
  int err_code = 0;
  std::string line[4];

  void close_file() {
    ra.shutdown();
    if (f) gzclose(f);
    if (fd >= 0) ::close(fd);
    f = nullptr;
    fd = -1;
    have_block = false;
  }
  ~shk_fastq() { close_file(); }

  // BufRead::lines(): split on '\n', strip one trailing '\r'.  1 = line, 0 = EOF, -1 = I/O error.
  // keep = false: the line's text is not needed (header / separator / quality of a record that is
  // not validated, io.rs:321-332), only that it exists.
  int next_line(std::string &out, bool keep = true) {
    out.clear();
    bool got_any = false;
    for (;;) {
      if (buf_pos == buf_len) {
        if (file_eof) break;
        int n = 0;
        buf = ra.next(have_block, &n);
        have_block = true;
        if (n < 0) return -1;
        if (n == 0) {
          file_eof = true;
          buf_pos = buf_len = 0;
          break;
        }
        buf_pos = 0;
        buf_len = (size_t)n;
      }
      const char *p = buf + buf_pos;
      const char *nl = (const char *)memchr(p, '\n', buf_len - buf_pos);
      if (nl) {
        if (keep) {
          out.append(p, nl - p);
          if (!out.empty() && out.back() == '\r') out.pop_back();
        }
        buf_pos += (size_t)(nl - p) + 1;
        return 1;
      }
      if (keep) out.append(p, buf_len - buf_pos);
      got_any = got_any || buf_len > buf_pos;
      buf_pos = buf_len;
    }
    if (!out.empty() || got_any) return 1;  // last line without newline
    return 0;
  }

  int open_next() {  // open_fastq_reader, io.rs:598-625 (gzread passes plain files through)
    close_file();
    if (file_idx >= paths.size()) return 0;
    cur_name = paths[file_idx++];
    if (cur_name != "-") {  // a file that does not start with the gzip magic is read directly
      fd = ::open(cur_name.c_str(), O_RDONLY);
      unsigned char magic[2] = {0, 0};
      if (fd >= 0 && (::pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b))) {
        ::close(fd);
        fd = -1;
      }
    }
    if (fd < 0) {
      f = cur_name == "-" ? gzdopen(0, "rb") : gzopen(cur_name.c_str(), "rb");
      if (!f) {
        err = fmt("Failed to open file: %s", cur_name.c_str());
        err_code = SHK_ERR_IO;
        return -1;
      }
      gzbuffer(f, 1 << 20);
    }
    buf_pos = buf_len = 0;
    file_eof = false;
    ra.start(f, fd);
    if (cur_name == "-") cur_name = "stdin";
    return 1;
  }

  int fail(int code, const std::string &m) {
    err = m;
    err_code = code;
    return code;
  }

  // validate_fastq_record, io.rs:161-198
  int validate() {
    const unsigned long long rec = n_reads_read + 1;
    const std::string &h = line[0], &sep = line[2];
    if (!h.empty() && h[0] == '>')
      return fail(SHK_ERR_FASTQ,
                  fmt("Input appears to be FASTA format, not FASTQ (record %llu starts with '>'). "
                      "sharkmer requires FASTQ input with quality scores.",
                      rec));
    if (h.empty() || h[0] != '@')
      return fail(SHK_ERR_FASTQ, fmt("FASTQ record %llu has invalid header (expected '@', got '%c'): %s", rec,
                                     h.empty() ? ' ' : h[0], h.c_str()));
    if (sep.empty() || sep[0] != '+')
      return fail(SHK_ERR_FASTQ,
                  fmt("FASTQ record %llu has invalid separator line (expected '+', got '%c'): %s", rec,
                      sep.empty() ? ' ' : sep[0], sep.c_str()));
    if (line[3].size() != line[1].size())
      return fail(SHK_ERR_FASTQ, fmt("FASTQ record %llu has mismatched sequence (%zu) and quality (%zu) lengths",
                                     rec, line[1].size(), line[3].size()));
    return SHK_OK;
  }

  // One record into line[0..3].  1 = record, 0 = end of this file, <0 = error
  int next_record() {
    // io.rs:321-332: only record 0 and every validate_every-th are looked at beyond their sequence
    const bool keep = n_reads_read == 0 || (validate_every > 0 && n_reads_read % validate_every == 0);
    int g = next_line(line[0], keep);
    if (g == 0) return 0;
    static const char *role[4] = {"header", "sequence", "separator", "quality"};
    if (g < 0)
      return fail(SHK_ERR_IO, fmt("Failed to read %s line of record %llu in %s", role[0],
                                  (unsigned long long)n_reads_read + 1, cur_name.c_str()));
    for (int i = 1; i < 4; ++i) {
      g = next_line(line[i], keep || i == 1);
      if (g == 0)  // io.rs:291-317
        return fail(SHK_ERR_FASTQ, fmt("Truncated FASTQ record at record %llu in %s: missing %s line",
                                       (unsigned long long)n_reads_read + 1, cur_name.c_str(), role[i]));
      if (g < 0)
        return fail(SHK_ERR_IO, fmt("Failed to read %s line of record %llu in %s", role[i],
                                    (unsigned long long)n_reads_read + 1, cur_name.c_str()));
    }
    return 1;
  }
};

extern "C" {

int shk_fastq_open(const char *const *paths, uint32_t n_paths, uint64_t max_reads, uint64_t validate_every,
                   shk_fastq **out) {
  if (!out) return SHK_ERR_BAD_ARG;
  auto *r = new shk_fastq();
  for (uint32_t i = 0; i < n_paths; ++i) r->paths.emplace_back(paths[i]);
  if (n_paths == 0) r->paths.emplace_back("-");  // stdin, io.rs:517-537
  r->max_reads = max_reads;
  r->validate_every = validate_every;
  *out = r;
  return SHK_OK;
}

void shk_fastq_close(shk_fastq *r) { delete r; }
const char *shk_fastq_error(const shk_fastq *r) { return r ? r->err.c_str() : ""; }

int shk_fastq_stats(const shk_fastq *r, uint64_t *n_reads_read, uint64_t *n_bases_read, int *reached_max,
                    int *done) {
  if (!r) return SHK_ERR_BAD_ARG;
  if (n_reads_read) *n_reads_read = r->n_reads_read;
  if (n_bases_read) *n_bases_read = r->n_bases_read;
  if (reached_max) *reached_max = r->reached_max;
  if (done) *done = r->done && !r->pending;
  return SHK_OK;
}

// Fill (bases, offsets) with up to max_seqs sequences / max_bases bytes in input order.
// offsets[0] = 0.  *n_seqs = 0 with SHK_OK means end of input.  Sequences are never split; a
// sequence longer than bases_cap is an error.
int shk_fastq_next_batch(shk_fastq *r, uint8_t *bases, uint64_t bases_cap, uint64_t *offsets,
                         uint64_t max_seqs, uint64_t *n_seqs) {
  if (!r || !bases || !offsets || !n_seqs) return SHK_ERR_BAD_ARG;
  *n_seqs = 0;
  offsets[0] = 0;
  if (r->err_code) return r->err_code;
  uint64_t used = 0, n = 0;
  while (!r->done && n < max_seqs) {
    if (!r->f && r->fd < 0 && !r->pending) {
      int o = r->open_next();
      if (o < 0) return r->err_code;
      if (o == 0) {
        r->done = true;
        break;
      }
    }
    if (!r->pending) {
      int g = r->next_record();
      if (g < 0) return g;
      if (g == 0) {  // this file is exhausted; state persists into the next one (io.rs:498-512)
        r->close_file();
        continue;
      }
      // io.rs:321-332
      const bool should_validate =
          r->n_reads_read == 0 || (r->validate_every > 0 && r->n_reads_read % r->validate_every == 0);
      if (should_validate) {
        int v = r->validate();
        if (v != SHK_OK) return v;
      }
    }
    const std::string &seq = r->line[1];
    if (seq.size() > bases_cap) return r->fail(SHK_ERR_BAD_ARG, "sequence longer than the batch buffer");
    if (used + seq.size() > bases_cap) {  // does not fit: deliver it first thing next call
      r->pending = true;
      break;
    }
    r->pending = false;
    memcpy(bases + used, seq.data(), seq.size());
    used += seq.size();
    offsets[++n] = used;
    r->n_bases_read += seq.size();  // io.rs:335 (N included)
    r->n_reads_read += 1;           // io.rs:337
    if (r->max_reads > 0 && r->n_reads_read >= r->max_reads) {  // io.rs:345-348
      r->reached_max = true;
      r->done = true;
    }
  }
  *n_seqs = n;
  return SHK_OK;
}

// ---- writers ---------------------------------------------------------------------------------------

int shk_write_histo(const char *path, const char *version, uint32_t k, uint32_t chunks, uint64_t histo_max,
                    const uint64_t *histo /* chunks × (histo_max+2) */) {
  if (!path || !histo || chunks == 0) return SHK_ERR_BAD_ARG;
  FILE *f = fopen(path, "w");
  if (!f) return SHK_ERR_IO;
  const uint64_t len = histo_max + 2;
  std::string out = fmt("# sharkmer %s k=%u chunks=%u\n", version, k, chunks);  // io.rs:1009-1014
  out += "count";
  for (uint32_t c = 1; c <= chunks; ++c) out += fmt("\tchunk_%u", c);
  out += '\n';
  for (uint64_t i = 1; i < len; ++i) {  // io.rs:1066-1073: rows 1..=histo_max+1
    out += std::to_string(i);
    for (uint32_t c = 0; c < chunks; ++c) {
      out += '\t';
      out += std::to_string(histo[(uint64_t)c * len + i]);
    }
    out += '\n';
    if (out.size() > (1u << 20)) {
      fwrite(out.data(), 1, out.size(), f);
      out.clear();
    }
  }
  fwrite(out.data(), 1, out.size(), f);
  return fclose(f) == 0 ? SHK_OK : SHK_ERR_IO;
}

int shk_write_final_histo(const char *path, const char *version, uint32_t k, uint32_t chunks,
                          uint64_t histo_max, const uint64_t *histo) {
  if (!path || !histo || chunks == 0) return SHK_ERR_BAD_ARG;
  FILE *f = fopen(path, "w");
  if (!f) return SHK_ERR_IO;
  const uint64_t len = histo_max + 2;
  const uint64_t *last = histo + (uint64_t)(chunks - 1) * len;
  std::string out = fmt("# sharkmer %s k=%u chunks=%u\n", version, k, chunks);
  out += "count\tfrequency\n";  // io.rs:1085
  for (uint64_t i = 1; i < len; ++i) out += fmt("%llu\t%llu\n", (unsigned long long)i, (unsigned long long)last[i]);
  fwrite(out.data(), 1, out.size(), f);
  return fclose(f) == 0 ? SHK_OK : SHK_ERR_IO;
}

// serde_yaml_ng plain-scalar rule, conservatively: quote when the text could be read as
// something other than a string
static std::string yaml_str(const std::string &s) {
  bool plain = !s.empty();
  static const char *specials = "-?:,[]{}#&*!|>'\"%@`";
  if (plain && (strchr(specials, s[0]) || s[0] == ' ' || s.back() == ' ')) plain = false;
  if (plain && (s.find(": ") != std::string::npos || s.find(" #") != std::string::npos ||
                s.find('\n') != std::string::npos || s.back() == ':'))
    plain = false;
  if (plain) {
    static const char *kw[] = {"null", "Null", "NULL", "~", "true", "True", "TRUE", "false", "False", "FALSE"};
    for (auto w : kw)
      if (s == w) plain = false;
    char *end = nullptr;
    strtod(s.c_str(), &end);
    if (end && *end == 0) plain = false;  // looks like a number
  }
  if (plain) return s;
  std::string q = "'";
  for (char ch : s) {
    if (ch == '\'') q += "''";
    else q += ch;
  }
  q += "'";
  return q;
}

int shk_write_stats_yaml(const char *path, const shk_run_stats *st) {
  if (!path || !st) return SHK_ERR_BAD_ARG;
  FILE *f = fopen(path, "w");
  if (!f) return SHK_ERR_IO;
  // RunStats field order, stats.rs:27-45; Option fields skipped when None, pcr_results when empty
  fprintf(f, "sharkmer_version: %s\n", yaml_str(st->sharkmer_version ? st->sharkmer_version : "").c_str());
  fprintf(f, "command: %s\n", yaml_str(st->command ? st->command : "").c_str());
  fprintf(f, "sample: %s\n", yaml_str(st->sample ? st->sample : "").c_str());
  fprintf(f, "kmer_length: %u\n", st->kmer_length);
  fprintf(f, "chunks: %u\n", st->chunks);
  fprintf(f, "n_reads_read: %llu\n", (unsigned long long)st->n_reads_read);
  fprintf(f, "n_bases_read: %llu\n", (unsigned long long)st->n_bases_read);
  fprintf(f, "n_subreads_ingested: %llu\n", (unsigned long long)st->n_subreads_ingested);
  fprintf(f, "n_bases_ingested: %llu\n", (unsigned long long)st->n_bases_ingested);
  fprintf(f, "n_kmers: %llu\n", (unsigned long long)st->n_kmers);
  if (st->has_histogram) {  // main.rs:192-193
    fprintf(f, "n_multi_kmers: %llu\n", (unsigned long long)st->n_multi_kmers);
    fprintf(f, "n_singleton_kmers: %llu\n", (unsigned long long)st->n_singleton_kmers);
  }
  fprintf(f, "peak_memory_bytes: %llu\n", (unsigned long long)st->peak_memory_bytes);
  return fclose(f) == 0 ? SHK_OK : SHK_ERR_IO;
}

// ---- whole run over local files (main.rs:74-78,112-197 minus sPCR) ---------------------------------------

static thread_local std::string g_run_error;
const char *shk_run_error(void) { return g_run_error.c_str(); }

int shk_validate_args(uint32_t k, uint64_t histo_max, const char *sample) {
  // cli.rs:659-673 and cli.rs:645-652, same messages
  if (!(k < 32)) {
    g_run_error = "k must be less than 32 due to use of 64 bit integers to encode kmers";
    return SHK_ERR_BAD_ARG;
  }
  if (!(k > 0)) {
    g_run_error = "k must be greater than 0";
    return SHK_ERR_BAD_ARG;
  }
  if (k % 2 != 1) {
    g_run_error = "k must be odd";
    return SHK_ERR_BAD_ARG;
  }
  if (!(histo_max > 0)) {
    g_run_error = "histo_max must be greater than 0";
    return SHK_ERR_BAD_ARG;
  }
  if (histo_max > 1000000) {
    g_run_error = fmt("histo_max must not exceed 1000000, got %llu", (unsigned long long)histo_max);
    return SHK_ERR_BAD_ARG;
  }
  if (!sample) {
    g_run_error = "--sample is required. Provide a sample name as output file prefix.\n"
                  "When using --ena, the sample name can be derived automatically from ENA metadata.";
    return SHK_ERR_BAD_ARG;
  }
  for (const char *p = sample; *p; ++p) {
    unsigned char ch = (unsigned char)*p;
    bool ok = (ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || ch == '_' ||
              ch == '-' || ch == '.' || ch >= 0x80;
    if (!ok) {
      g_run_error = fmt("Sample name '%s' contains characters that are unsafe for filenames. "
                        "Use only alphanumeric characters, hyphens, underscores, and periods.",
                        sample);
      return SHK_ERR_BAD_ARG;
    }
  }
  return SHK_OK;
}

// ---- 2-bit packing on the host (the reference's Read::from_str layout, encoding.rs:60-95, over the whole
// batch as ONE sequence, + the N mask kmers_from_ascii's N handling needs, encoding.rs:346-352) -------------
void shk_packed_sizes(uint64_t n_bases, uint64_t *packed_bytes, uint64_t *nmask_words) {
  if (packed_bytes) *packed_bytes = (n_bases + 3) / 4;
  if (nmask_words) *nmask_words = (n_bases + 31) / 32;
}

int shk_pack_reads(const uint8_t *bases, uint64_t n_bases, uint8_t *packed, uint32_t *nmask, uint32_t n_threads) {
  g_run_error.clear();
  if (n_bases == 0) return SHK_OK;
  if (!bases || !packed || !nmask) {
    g_run_error = "null buffer";
    return SHK_ERR_BAD_ARG;
  }
  // 256-entry table: 0-3 the code, 4 = N, 0xFF = invalid (encoding.rs:341-356)
  static const struct Lut {
    uint8_t t[256];
    Lut() {
      memset(t, 0xFF, sizeof t);
      t[(unsigned)'A'] = 0, t[(unsigned)'C'] = 1, t[(unsigned)'G'] = 2, t[(unsigned)'T'] = 3, t[(unsigned)'N'] = 4;
    }
  } lut;
  const uint64_t n_words = (n_bases + 31) / 32;  // a thread's share is whole 32-base groups: no shared byte or word
  uint32_t T = n_threads ? n_threads : std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
  if (n_words < 4096) T = 1;
  T = (uint32_t)std::min<uint64_t>(T, n_words);
  std::vector<uint64_t> bad(T, ~0ull);  // first offender of each share: position << 8 | byte
  auto work = [&](uint32_t t) {
    const uint64_t w0 = n_words * t / T, w1 = n_words * (t + 1) / T;
    for (uint64_t w = w0; w < w1; ++w) {
      const uint64_t p0 = w * 32;
      const uint32_t n = (uint32_t)std::min<uint64_t>(32, n_bases - p0);
      uint64_t bits = 0;
      uint32_t nm = 0;
      for (uint32_t i = 0; i < n; ++i) {
        const uint8_t c = lut.t[bases[p0 + i]];
        if (c == 0xFF) {
          if (bad[t] == ~0ull) bad[t] = ((p0 + i) << 8) | bases[p0 + i];
          continue;
        }
        nm |= (uint32_t)(c >> 2) << i;
        bits |= (uint64_t)(c & 3u) << (62 - 2 * i);
      }
      nmask[w] = nm;
      for (uint32_t j = 0; j < (n + 3) / 4; ++j) packed[w * 8 + j] = (uint8_t)(bits >> (56 - 8 * j));
    }
  };
  if (T == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  uint64_t first = ~0ull;
  for (uint64_t b : bad) first = std::min(first, b);
  if (first != ~0ull) {  // identical text to encoding.rs:353-356
    g_run_error = fmt("Invalid character '%c' in sequence. Only ACGTN allowed.", (char)(first & 0xFF));
    return SHK_ERR_INVALID_CHAR;
  }
  return SHK_OK;
}

int shk_run_files(const shk_run_config *rc, shk_run_stats *out_stats) {
  if (!rc) return SHK_ERR_BAD_ARG;
  g_run_error.clear();
  int v = shk_validate_args(rc->k, rc->histo_max, rc->sample);
  if (v != SHK_OK) return v;
  std::string dir = rc->outdir ? rc->outdir : "./";  // main.rs:74-78
  if (dir.empty() || dir.back() != '/') dir += '/';
  {
    std::string cmd = "mkdir -p '" + dir + "'";
    if (system(cmd.c_str()) != 0) {
      g_run_error = fmt("Failed to create output directory: %s", dir.c_str());
      return SHK_ERR_IO;
    }
  }
  shk_fastq *rd = nullptr;
  v = shk_fastq_open(rc->inputs, rc->n_inputs, rc->max_reads, rc->validate_every, &rd);
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
  // super-batches through pinned buffers; the engine stripes by its own running read index,
  // which is exactly drain_batch's cadence (io.rs:340-343,355-361) whatever the batch size
  const uint64_t max_seqs = rc->batch_reads ? rc->batch_reads : 1000000;
  const uint64_t cap_bases = rc->batch_bases ? rc->batch_bases : (256ull << 20);
  uint8_t *bases = (uint8_t *)shk_alloc_pinned(cap_bases);
  uint64_t *offs = (uint64_t *)shk_alloc_pinned((max_seqs + 1) * 8);
  auto cleanup = [&]() {
    shk_free_pinned(bases);
    shk_free_pinned(offs);
    shk_destroy(ctx);
    shk_fastq_close(rd);
  };
  if (!bases || !offs) {
    cleanup();
    g_run_error = "pinned buffer allocation failed";
    return SHK_ERR_NOMEM;
  }
  for (;;) {  // any batch size keeps the striping: the engine counts reads itself
    uint64_t n = 0;
    v = shk_fastq_next_batch(rd, bases, cap_bases, offs, max_seqs, &n);
    if (v != SHK_OK) {
      g_run_error = shk_fastq_error(rd);
      cleanup();
      return v;
    }
    if (n) {
      v = shk_ingest_reads(ctx, bases, offs, n);
      if (v != SHK_OK) {
        g_run_error = shk_last_error(ctx);
        cleanup();
        return v;
      }
    }
    int done = 0;
    shk_fastq_stats(rd, nullptr, nullptr, nullptr, &done);
    if (done) break;
  }
  v = shk_finalize(ctx);
  if (v != SHK_OK) {
    g_run_error = shk_last_error(ctx);
    cleanup();
    return v;
  }
  shk_counters cn{};
  shk_get_counters(ctx, &cn);
  uint64_t nrr = 0, nbr = 0;
  shk_fastq_stats(rd, &nrr, &nbr, nullptr, nullptr);
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
  cleanup();
  if (v != SHK_OK) g_run_error = "Failed to create stats file";
  return v;
}

}  // extern "C"
