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
#include <sched.h>
#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
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
// Front-end at speed, order-preserving.  Every input file has a PRODUCER that turns
// it into chunks of sequences (bases back to back + lengths) and notes what is wrong with which record;
// the caller's thread (shk_fastq_next_batch) CONSUMES the files strictly in input order, so the global
// read index — and with it the 1000-read chunk striping (io.rs:340-361), the validation cadence
// (io.rs:321-332) and --max-reads (io.rs:345-348) — is exactly the sequential reader's:
//   * plain files: mmap + a parallel parse of 128 MiB windows by a pool of threads — newline positions
//     per thread share, a prefix sum gives every share its line number (a FASTQ record is exactly four lines,
//     BufRead::lines: io.rs:271-352), then the sequence lines are sized and copied out in parallel;
//   * gzip streams and stdin: one inflate + line-split thread per stream; the streams of LATER files run
//     ahead (up to 8 at a time, each bounded by its queue), which is the only parallelism zlib allows.
// A record is only ever looked at beyond its sequence when the cadence says so; producers check every
// record (it costs nothing next to the line split) but only record WHAT they found — the consumer raises
// a flaw when the flawed record turns out to be one the reference would have validated, with the reference's
// text and the global record number.
namespace {

// CPUs this process may really keep busy: the hardware's, the affinity mask's, and the container's CFS quota
// (cgroup v2 cpu.max / v1 cpu.cfs_quota_us) — whichever is smallest.
static uint32_t usable_cpus() {
  uint32_t n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<uint32_t>(n, (uint32_t)std::max(1, CPU_COUNT(&set)));
  auto read2 = [](const char *path, long long &a, long long &b) {
    FILE *f = fopen(path, "r");
    if (!f) return false;
    char x[64] = {0}, y[64] = {0};
    const int got = fscanf(f, "%63s %63s", x, y);
    fclose(f);
    if (got < 1 || !strcmp(x, "max")) return false;
    a = atoll(x);
    b = got > 1 ? atoll(y) : 0;
    return a > 0;
  };
  long long q = 0, per = 0;
  if (read2("/sys/fs/cgroup/cpu.max", q, per) && per > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, q / per));
  else if (read2("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q, per)) {
    long long p2 = 0, dummy = 0;
    if (read2("/sys/fs/cgroup/cpu/cpu.cfs_period_us", p2, dummy) && p2 > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, q / p2));
  }
  return n;
}

struct Pool {  // a small persistent pool: parallel_for over [0, n)
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::function<void(uint32_t)> job;
  uint32_t n_jobs = 0, next = 0, running = 0;
  uint64_t gen = 0;
  bool quit = false;
  explicit Pool(uint32_t T) {
    for (uint32_t t = 0; t < T; ++t) th.emplace_back([this] { run(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
    }
    cv_job.notify_all();
    for (auto &t : th) t.join();
  }
  void run() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv_job.wait(lk, [&] { return quit || next < n_jobs; });
      if (quit) return;
      const uint32_t i = next++;
      ++running;
      lk.unlock();
      job(i);
      lk.lock();
      if (--running == 0 && next >= n_jobs) cv_done.notify_all();
    }
  }
  std::mutex use;  // one parallel_for at a time (the file producers and the consumer share the pool)
  void parallel_for(uint32_t n, std::function<void(uint32_t)> f) {
    if (n == 0) return;
    std::lock_guard<std::mutex> only(use);
    std::unique_lock<std::mutex> lk(m);
    job = std::move(f);
    n_jobs = n;
    next = 0;
    cv_job.notify_all();
    cv_done.wait(lk, [&] { return next >= n_jobs && running == 0; });
    n_jobs = 0;
  }
};

enum FlawKind : uint8_t { FLAW_FASTA, FLAW_HEADER, FLAW_SEP, FLAW_LEN };
struct Flaw {  // what validate_fastq_record (io.rs:161-198) would say about local record `rec` of a file
  uint64_t rec;
  FlawKind kind;
  std::string text;  // the header or separator line
  size_t seq_len = 0, qual_len = 0;
};
struct SeqChunk {
  std::vector<uint8_t> bases;  // stream producers: the sequences back to back
  // plain-file producers hand on no copy of the sequences, only where they lie in the mapped file: line ℓ of
  // the window is [ℓ ? nl[ℓ-1]+1 : 0, nl[ℓ]) from `src` on, record r's sequence line ℓ = 4r+1; the consumer's
  // thread pool copies them straight into the caller's batch buffer
  const char *src = nullptr;
  std::vector<uint32_t> nl;
  bool last_unterminated = false;
  std::vector<uint32_t> lens;
  std::vector<Flaw> flaws;   // ascending by rec
  uint64_t first_rec = 0;    // local index (within the file) of the chunk's first record
  size_t bytes() const { return bases.size() + lens.size() * 4 + nl.size() * 4; }
  void seq_line(size_t r, const char **p, size_t *len) const {
    const size_t l = 4 * r + 1;
    const size_t s0 = (size_t)nl[l - 1] + 1, s1 = nl[l];
    size_t n = s1 - s0;
    if (!(last_unterminated && l == nl.size() - 1) && n && src[s1 - 1] == '\r') --n;
    *p = src + s0;
    *len = n;
  }
};
// how a file ended
struct FileEnd {
  int kind = 0;  // 0 clean EOF, 1 truncated record (missing line `role`), 2 read error (line `role`), 3 cannot open
  int role = 0;
};

// first failing check of validate_fastq_record on a record's four lines
static bool find_flaw(const char *h, size_t hl, const char *sq, size_t sl, const char *sp, size_t spl, const char *q, size_t ql,
                      uint64_t rec, Flaw *out) {
  (void)sq;
  (void)q;
  if (hl && h[0] == '>') {
    *out = Flaw{rec, FLAW_FASTA, std::string(), 0, 0};
    return true;
  }
  if (!hl || h[0] != '@') {
    *out = Flaw{rec, FLAW_HEADER, std::string(h, hl), 0, 0};
    return true;
  }
  if (!spl || sp[0] != '+') {
    *out = Flaw{rec, FLAW_SEP, std::string(sp, spl), 0, 0};
    return true;
  }
  if (ql != sl) {
    *out = Flaw{rec, FLAW_LEN, std::string(), sl, ql};
    return true;
  }
  return false;
}

struct Producer {
  std::string path, name;  // name: what error messages call it ("stdin" for "-")
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::deque<SeqChunk> q;
  size_t q_bytes = 0;
  bool finished = false, cancel = false;
  FileEnd end;
  static constexpr size_t Q_MAX = 384u << 20;  // bytes buffered ahead per file
  const char *map = nullptr;  // plain files: the mapping the chunks point into (released with the producer)
  size_t map_size = 0;
  int map_fd = -1;
  ~Producer() {
    if (map) munmap((void *)map, map_size);
    if (map_fd >= 0) ::close(map_fd);
  }

  void push(SeqChunk &&c) {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return cancel || q_bytes < Q_MAX; });
    if (cancel) return;
    q_bytes += c.bytes();
    q.emplace_back(std::move(c));
    cv.notify_all();
  }
  void finish(FileEnd e) {
    std::lock_guard<std::mutex> lk(m);
    end = e;
    finished = true;
    cv.notify_all();
  }
  bool cancelled() {
    std::lock_guard<std::mutex> lk(m);
    return cancel;
  }

  // ---- gzip / stdin / anything zlib reads: one thread, inflate + line split ------------------------------
  void run_stream() {
    gzFile f = path == "-" ? gzdopen(0, "rb") : gzopen(path.c_str(), "rb");
    if (!f) return finish(FileEnd{3, 0});
    gzbuffer(f, 1 << 20);
    std::vector<char> buf(4u << 20);
    std::string line[4];
    int li = 0;           // which line of the record is being assembled
    bool partial = false;  // line[li] holds the beginning of an unterminated line
    uint64_t rec = 0;
    SeqChunk c;
    c.first_rec = 0;
    auto flush = [&]() {
      if (!c.lens.empty()) {
        SeqChunk out = std::move(c);
        c = SeqChunk();
        c.first_rec = rec;
        push(std::move(out));
      }
    };
    auto end_line = [&](bool strip_cr) {
      std::string &l = line[li];
      if (strip_cr && !l.empty() && l.back() == '\r') l.pop_back();
      if (li == 3) {
        Flaw fl;
        if (find_flaw(line[0].data(), line[0].size(), line[1].data(), line[1].size(), line[2].data(), line[2].size(),
                      line[3].data(), line[3].size(), rec, &fl))
          c.flaws.emplace_back(std::move(fl));
        c.bases.insert(c.bases.end(), line[1].begin(), line[1].end());
        c.lens.push_back((uint32_t)line[1].size());
        ++rec;
        for (auto &x : line) x.clear();
        li = 0;
        if (c.bases.size() >= (8u << 20)) flush();
      } else {
        ++li;
      }
      partial = false;
    };
    for (;;) {
      if (cancelled()) break;
      const int n = gzread(f, buf.data(), (unsigned)buf.size());
      if (n < 0) {
        flush();
        gzclose(f);
        return finish(FileEnd{2, li});
      }
      if (n == 0) break;
      const char *p = buf.data(), *e = p + n;
      while (p < e) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!nl) {
          line[li].append(p, (size_t)(e - p));
          partial = true;
          break;
        }
        line[li].append(p, (size_t)(nl - p));
        p = nl + 1;
        end_line(true);
      }
    }
    gzclose(f);
    if (partial) end_line(false);  // a last line without '\n' is a line (BufRead::lines); its '\r', if any, stays
    flush();
    finish(li == 0 ? FileEnd{0, 0} : FileEnd{1, li});
  }

  // ---- plain files: mmap + parallel parse -----------------------------------------------------------------
  std::vector<std::vector<uint32_t>> tl_nl;  // run_plain: the newline positions each pool thread found in its share of the window
  void run_plain(int fd, Pool *pool, uint32_t T) {
    struct stat st;
    if (fstat(fd, &st) != 0) {
      ::close(fd);
      return finish(FileEnd{3, 0});
    }
    const size_t size = (size_t)st.st_size;
    if (size == 0) {
      ::close(fd);
      return finish(FileEnd{0, 0});
    }
    const char *data = (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (data == MAP_FAILED) {  // (a pipe or a special file: the stream reader takes it)
      ::close(fd);
      return run_stream();
    }
    (void)madvise((void *)data, size, MADV_SEQUENTIAL);
    map = data;
    map_size = size;
    map_fd = fd;
    size_t WINDOW = 128u << 20;
    if (const char *wk = getenv("SHK_FASTQ_WINDOW_KB")) WINDOW = std::max<size_t>(1, (size_t)atoll(wk)) << 10;  // test hook
    size_t pos = 0;       // start of the first line not yet delivered (a record boundary)
    uint64_t rec = 0;     // local index of the record that starts at pos
    FileEnd fe{0, 0};
    while (pos < size && !cancelled()) {
      const size_t wend = std::min(size, pos + WINDOW);
      const size_t wlen = wend - pos;
      const bool dbg = getenv("SHK_FASTQ_DEBUG") != nullptr;
      auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
      const double t_a = now();
      // 1. newline positions: every share of the window is scanned ONCE — a thread keeps the positions it finds in
      // a list of its own (kept across windows: no allocation after the first) — and, their places among all the
      // lines known from a prefix sum, the lists are copied into the window's line table (4 B per line, ≈ 5 % of
      // the window's bytes; scanning every share twice instead cost a second pass over the file: 2.6 of 7 ms per
      // 128 MB window)
      std::vector<size_t> first(T + 1, 0);
      if (tl_nl.size() != T) tl_nl.assign(T, {});
      pool->parallel_for(T, [&](uint32_t t) {
        const char *base = data + pos;
        const char *p = base + wlen * t / T, *e = base + wlen * (t + 1) / T;
        std::vector<uint32_t> &v = tl_nl[t];
        v.clear();
        while (p < e) {
          const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
          if (!nl) break;
          v.push_back((uint32_t)(nl - base));
          p = nl + 1;
        }
        first[t + 1] = v.size();
      });
      const double t_b = now();
      for (uint32_t t = 0; t < T; ++t) first[t + 1] += first[t];
      size_t M = first[T];  // complete ('\n'-terminated) lines in the window
      std::vector<uint32_t> NL;
      NL.reserve(M + 1);
      NL.resize(M + 1);     // (value-initialised by one thread)
      pool->parallel_for(T, [&](uint32_t t) {
        if (!tl_nl[t].empty()) memcpy(NL.data() + first[t], tl_nl[t].data(), tl_nl[t].size() * 4);
      });
      bool last_unterminated = false;
      if (wend == size && (M == 0 ? wlen > 0 : (size_t)NL[M - 1] + 1 < wlen)) {  // a last line without '\n' (it has ≥ 1 byte)
        NL[M] = (uint32_t)wlen;
        ++M;
        last_unterminated = true;
      }
      const size_t R = M / 4;  // whole records in the window
      if (R == 0 && wend < size) {  // a record longer than the window: look at a longer stretch
        WINDOW *= 2;
        continue;
      }
      // line ℓ = [ℓ ? NL[ℓ-1]+1 : 0, NL[ℓ]) relative to pos, '\r' stripped (not from an unterminated last line)
      auto line_at = [&](size_t l, const char **p, size_t *len) {
        const size_t s0 = l ? (size_t)NL[l - 1] + 1 : 0, s1 = NL[l];
        size_t n = s1 - s0;
        const bool strip = !(last_unterminated && l == M - 1);
        if (strip && n && data[pos + s1 - 1] == '\r') --n;
        *p = data + pos + s0;
        *len = n;
      };
      const double t_c = now();
      SeqChunk c;
      c.first_rec = rec;
      c.lens.resize(R);
      std::vector<std::vector<Flaw>> fl(T);
      // 2. sequence lengths + flaws, records shared out evenly (the sequences themselves stay where they are)
      pool->parallel_for(T, [&](uint32_t t) {
        const size_t r0 = R * t / T, r1 = R * (t + 1) / T;
        for (size_t r = r0; r < r1; ++r) {
          const char *h, *sq, *sp, *ql;
          size_t hl, sl, spl, qll;
          line_at(4 * r, &h, &hl);
          line_at(4 * r + 1, &sq, &sl);
          line_at(4 * r + 2, &sp, &spl);
          line_at(4 * r + 3, &ql, &qll);
          c.lens[r] = (uint32_t)sl;
          Flaw f;
          if (find_flaw(h, hl, sq, sl, sp, spl, ql, qll, rec + r, &f)) fl[t].emplace_back(std::move(f));
        }
      });
      for (uint32_t t = 0; t < T; ++t)
        for (auto &f : fl[t]) c.flaws.emplace_back(std::move(f));
      c.src = data + pos;
      c.last_unterminated = last_unterminated;
      const size_t next_pos = wend == size ? size : pos + (size_t)NL[4 * R - 1] + 1;
      const size_t M_all = M;
      NL.resize(4 * R);  // (only whole records' lines are handed on)
      c.nl = std::move(NL);
      if (c.last_unterminated && 4 * R != M_all) c.last_unterminated = false;  // the unterminated line is not among them
      rec += R;
      const double t_d = now();
      if (R) push(std::move(c));
      if (dbg) fprintf(stderr, "[fastq window %zu MB] scan %.1f ms  gather %.1f ms  lens %.1f ms  push(wait) %.1f ms\n", wlen >> 20,
                       (t_b - t_a) * 1e3, (t_c - t_b) * 1e3, (t_d - t_c) * 1e3, (now() - t_d) * 1e3);
      if (wend == size && (M_all % 4)) fe = FileEnd{1, (int)(M_all % 4)};  // the file ends inside a record
      pos = next_pos;
    }
    finish(fe);
  }
};

}  // namespace

struct shk_fastq {
  std::vector<std::string> paths;
  std::vector<std::unique_ptr<Producer>> prod;  // one per path; started up to LOOKAHEAD files ahead of the consumer
  size_t file_idx = 0;        // the file being consumed
  size_t started = 0;         // producers started so far
  std::unique_ptr<Pool> pool, cpool;  // the producers' pool (window parse) and the consumer's (copy-out): they overlap
  uint32_t T = 1;
  SeqChunk cur;               // the chunk being handed out
  bool have_cur = false;
  size_t cur_seq = 0, cur_byte = 0, cur_flaw = 0;
  uint64_t file_rec = 0;      // local index of the next record of the current file
  uint64_t max_reads = 0, validate_every = 0;
  uint64_t n_reads_read = 0, n_bases_read = 0;  // FastqReadState, io.rs:205-206
  bool reached_max = false, done = false;
  std::string err;
  int err_code = 0;
  static constexpr size_t LOOKAHEAD = 8;

  ~shk_fastq() { stop_all(); }
  void stop_all() {
    for (auto &p : prod)
      if (p) {
        {
          std::lock_guard<std::mutex> lk(p->m);
          p->cancel = true;
        }
        p->cv.notify_all();
      }
    for (auto &p : prod)
      if (p && p->th.joinable()) p->th.join();
  }
  int fail(int code, const std::string &m) {
    err = m;
    err_code = code;
    return code;
  }
  void start_producers() {
    if (!pool) {
      // The parsing pool is the pipeline's critical path (the producer never waits for the consumer: a 128 MB window
      // is split in ≈5 ms, its sequences are copied out in less), so it gets every CPU this process may really use
      // — a container's CFS quota counts, not the host's core count — and the copying pool a quarter of that
      // (measured on a 16-CPU quota, Gbases/s of an 8 M-read plain file, three runs each, parse/copy threads: 8/8
      // 8.6, 12/4 9.0, 16/4 9.4 and 8.8, 14/2 7.1, 24/4 9.6, 24/8 10.0, 32/4 8.7 — a quota is an AVERAGE over 100 ms:
      // bursts wider than it are not throttled, and the pools idle half of the time).
      const uint32_t usable = usable_cpus();
      const char *ev = getenv("SHK_FASTQ_THREADS");
      const char *evc = getenv("SHK_FASTQ_COPY_THREADS");
      T = ev && atoi(ev) > 0 ? (uint32_t)atoi(ev) : std::max(2u, std::min(24u, usable + usable / 2));
      pool.reset(new Pool(T));
      cpool.reset(new Pool(evc && atoi(evc) > 0 ? (uint32_t)atoi(evc) : std::max(2u, std::min(8u, usable / 2))));
    }
    // plain files go through the shared pool one after the other (each is parsed at memory speed); streams
    // get a thread each and run ahead
    while (started < paths.size() && started < file_idx + LOOKAHEAD) {
      auto p = std::make_unique<Producer>();
      p->path = paths[started];
      p->name = p->path == "-" ? "stdin" : p->path;
      Producer *pp = p.get();
      int fd = -1;
      if (pp->path != "-") {  // a file that does not start with the gzip magic is read directly (open_fastq_reader, io.rs:598-625)
        fd = ::open(pp->path.c_str(), O_RDONLY);
        unsigned char magic[2] = {0, 0};
        struct stat st;
        const bool regular = fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
        if (fd >= 0 && (!regular || (st.st_size >= 2 && (::pread(fd, magic, 2, 0) != 2 || (magic[0] == 0x1f && magic[1] == 0x8b))))) {
          ::close(fd);
          fd = -1;
        }
      }
      if (fd >= 0) {
        // (plain files share the pool: a second one starts when the first is done — the pool is not re-entrant)
        Producer *prev = plain_tail;
        plain_tail = pp;
        Pool *pl = pool.get();
        const uint32_t t = T;
        pp->th = std::thread([pp, prev, fd, pl, t] {
          if (prev) {
            std::unique_lock<std::mutex> lk(prev->m);
            prev->cv.wait(lk, [&] { return prev->finished || prev->cancel; });
          }
          pp->run_plain(fd, pl, t);
        });
      } else {
        pp->th = std::thread([pp] { pp->run_stream(); });
      }
      prod.emplace_back(std::move(p));
      ++started;
    }
  }
  Producer *plain_tail = nullptr;

  // next chunk of the current file into `cur`; 1 = got one, 0 = file finished (its end is in *fe)
  int next_chunk(FileEnd *fe) {
    Producer *p = prod[file_idx].get();
    std::unique_lock<std::mutex> lk(p->m);
    p->cv.wait(lk, [&] { return !p->q.empty() || p->finished; });
    if (!p->q.empty()) {
      cur = std::move(p->q.front());
      p->q.pop_front();
      p->q_bytes -= cur.bytes();
      p->cv.notify_all();
      have_cur = true;
      cur_seq = cur_byte = cur_flaw = 0;
      return 1;
    }
    *fe = p->end;
    return 0;
  }
  // validate_fastq_record's message for a flaw, with the GLOBAL record number (io.rs:161-198)
  int raise(const Flaw &f) {
    const unsigned long long rec = n_reads_read + 1;
    switch (f.kind) {
      case FLAW_FASTA:
        return fail(SHK_ERR_FASTQ, fmt("Input appears to be FASTA format, not FASTQ (record %llu starts with '>'). "
                                       "sharkmer requires FASTQ input with quality scores.", rec));
      case FLAW_HEADER:
        return fail(SHK_ERR_FASTQ, fmt("FASTQ record %llu has invalid header (expected '@', got '%c'): %s", rec,
                                       f.text.empty() ? ' ' : f.text[0], f.text.c_str()));
      case FLAW_SEP:
        return fail(SHK_ERR_FASTQ, fmt("FASTQ record %llu has invalid separator line (expected '+', got '%c'): %s", rec,
                                       f.text.empty() ? ' ' : f.text[0], f.text.c_str()));
      default:
        return fail(SHK_ERR_FASTQ, fmt("FASTQ record %llu has mismatched sequence (%zu) and quality (%zu) lengths", rec,
                                       f.seq_len, f.qual_len));
    }
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
  if (done) *done = r->done;
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
  static const char *role[4] = {"header", "sequence", "separator", "quality"};
  uint64_t used = 0, n = 0;
  while (!r->done && n < max_seqs) {
    if (r->file_idx >= r->paths.size()) {
      r->done = true;
      break;
    }
    r->start_producers();
    if (!r->have_cur) {
      FileEnd fe;
      if (r->next_chunk(&fe) == 0) {  // this file is exhausted; state persists into the next one (io.rs:498-512)
        Producer *p = r->prod[r->file_idx].get();
        if (fe.kind == 3) return r->fail(SHK_ERR_IO, fmt("Failed to open file: %s", p->path.c_str()));
        if (fe.kind == 1)  // io.rs:291-317
          return r->fail(SHK_ERR_FASTQ, fmt("Truncated FASTQ record at record %llu in %s: missing %s line",
                                             (unsigned long long)r->n_reads_read + 1, p->name.c_str(), role[fe.role]));
        if (fe.kind == 2)
          return r->fail(SHK_ERR_IO, fmt("Failed to read %s line of record %llu in %s", role[fe.role],
                                         (unsigned long long)r->n_reads_read + 1, p->name.c_str()));
        if (p->th.joinable()) p->th.join();
        ++r->file_idx;
        r->file_rec = 0;
        continue;
      }
    }
    // hand out what fits of the current chunk
    SeqChunk &c = r->cur;
    const size_t seq_begin = r->cur_seq;
    const uint64_t n_begin = n;
    bool stop = false;
    while (r->cur_seq < c.lens.size() && n < max_seqs) {
      const uint64_t len = c.lens[r->cur_seq];
      // io.rs:321-332: record 0 and every validate_every-th are validated — by global record index
      const bool cadence = r->n_reads_read == 0 || (r->validate_every > 0 && r->n_reads_read % r->validate_every == 0);
      const uint64_t local = c.first_rec + r->cur_seq;
      while (r->cur_flaw < c.flaws.size() && c.flaws[r->cur_flaw].rec < local) ++r->cur_flaw;
      if (cadence && r->cur_flaw < c.flaws.size() && c.flaws[r->cur_flaw].rec == local) return r->raise(c.flaws[r->cur_flaw]);
      // (a sequence that no batch of this size can hold: reported when it is the FIRST of a batch, with nothing
      // consumed — the caller may come back with a larger buffer; shk_run_files does)
      if (len > bases_cap && n == n_begin && used == 0) {
        r->err = "sequence longer than the batch buffer";  // (not sticky: the same call with a larger buffer goes on)
        return SHK_ERR_BAD_ARG;
      }
      if (used + len > bases_cap) {  // does not fit: it is delivered first thing next call
        stop = true;
        break;
      }
      if (!c.src) memcpy(bases + used, c.bases.data() + r->cur_byte, len);  // (a plain file's sequences: copied out below, in parallel)
      used += len;
      r->cur_byte += len;
      ++r->cur_seq;
      offsets[++n] = used;
      r->n_bases_read += len;  // io.rs:335 (N included)
      r->n_reads_read += 1;    // io.rs:337
      if (r->max_reads > 0 && r->n_reads_read >= r->max_reads) {  // io.rs:345-348
        r->reached_max = true;
        r->done = true;
        stop = true;
        break;
      }
    }
    if (c.src && r->cur_seq > seq_begin) {  // the sequences of records [seq_begin, cur_seq): mapped file → batch buffer
      const size_t cnt = r->cur_seq - seq_begin;
      const uint32_t TT = cnt >= 4096 ? r->T : 1;
      auto copy = [&](uint32_t t) {
        const size_t j0 = cnt * t / TT, j1 = cnt * (t + 1) / TT;
        for (size_t j = j0; j < j1; ++j) {
          const char *sq;
          size_t sl;
          c.seq_line(seq_begin + j, &sq, &sl);
          memcpy(bases + offsets[n_begin + j], sq, sl);
        }
      };
      if (TT == 1) copy(0);
      else r->cpool->parallel_for(TT, copy);
    }
    if (r->done) r->stop_all();
    if (r->cur_seq >= c.lens.size()) {
      r->have_cur = false;
      r->cur = SeqChunk();
    }
    if (stop) break;
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
  // (pinning host memory costs ≈ 45 µs per MB each way: two 256 MB buffers were 40 ms of a 1.2 Gbase job; 64 MB
  // batches keep every launch large enough, and a read that is longer gets larger buffers when it shows up)
  uint64_t cap_bases = rc->batch_bases ? rc->batch_bases : (64ull << 20);
  // two batch buffers: while the engine takes batch i (copy + count), the front-end fills batch i+1
  uint8_t *bases2[2] = {(uint8_t *)shk_alloc_pinned(cap_bases), nullptr};
  uint64_t *offs2[2] = {(uint64_t *)shk_alloc_pinned((max_seqs + 1) * 8), nullptr};
  std::thread ingest_th;
  int ingest_rc = SHK_OK;
  auto join_ingest = [&]() {
    if (ingest_th.joinable()) ingest_th.join();
    return ingest_rc;
  };
  auto cleanup = [&]() {
    (void)join_ingest();
    for (int i = 0; i < 2; ++i) {
      shk_free_pinned(bases2[i]);
      shk_free_pinned(offs2[i]);
    }
    shk_destroy(ctx);
    shk_fastq_close(rd);
  };
  if (!bases2[0] || !offs2[0]) {
    cleanup();
    g_run_error = "pinned buffer allocation failed";
    return SHK_ERR_NOMEM;
  }
  for (int cur = 0;; cur ^= 1) {  // any batch size keeps the striping: the engine counts reads itself
    if (!bases2[cur]) {  // (the second pair is only allocated when there is a second batch)
      bases2[cur] = (uint8_t *)shk_alloc_pinned(cap_bases);
      offs2[cur] = (uint64_t *)shk_alloc_pinned((max_seqs + 1) * 8);
      if (!bases2[cur] || !offs2[cur]) {
        cleanup();
        g_run_error = "pinned buffer allocation failed";
        return SHK_ERR_NOMEM;
      }
    }
    uint64_t n = 0;
    v = shk_fastq_next_batch(rd, bases2[cur], cap_bases, offs2[cur], max_seqs, &n);
    while (v == SHK_ERR_BAD_ARG && !rc->batch_bases && cap_bases < (16ull << 30) &&
           strstr(shk_fastq_error(rd), "longer than the batch buffer")) {  // a very long read: larger buffers, same call again
      if (join_ingest() != SHK_OK) break;  // (the other buffer was still being read by the engine; its outcome comes first, below)
      cap_bases *= 4;
      for (int i = 0; i < 2; ++i) {
        if (!bases2[i]) continue;  // (not in use yet: allocated at the new size when it is)
        shk_free_pinned(bases2[i]);
        bases2[i] = (uint8_t *)shk_alloc_pinned(cap_bases);
        if (!bases2[i]) {
          cleanup();
          g_run_error = "pinned buffer allocation failed";
          return SHK_ERR_NOMEM;
        }
      }
      v = shk_fastq_next_batch(rd, bases2[cur], cap_bases, offs2[cur], max_seqs, &n);
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
      uint8_t *bb = bases2[cur];
      uint64_t *oo = offs2[cur];
      ingest_rc = SHK_OK;
      ingest_th = std::thread([&ingest_rc, ctx, bb, oo, n] { ingest_rc = shk_ingest_reads(ctx, bb, oo, n); });
    }
    int done = 0;
    shk_fastq_stats(rd, nullptr, nullptr, nullptr, &done);
    if (done) break;
  }
  v = join_ingest();
  if (v != SHK_OK) {
    g_run_error = shk_last_error(ctx);
    cleanup();
    return v;
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
