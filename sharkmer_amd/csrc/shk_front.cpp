// shk_front.cpp — the host side that sits either side of the counting path, in plain C++
// (no HIP call in this file: it is also built with -fsanitize=address,undefined and -fsanitize=thread, `make san`):
//
//   * FASTQ(.gz) front-end restating read_fastq / open_fastq_reader / validate_fastq_record / stream_io_error
//     (src/io.rs:161-198, 213-265, 271-352, 598-625): 4 lines per record through BufRead::lines (CRLF tolerant, a line
//     that is not UTF-8 is an I/O error), record 0 and every validate_every-th record validated, identical error
//     texts, state carried across files so that 1000-read batches span file boundaries (io.rs:498-512), --max-reads
//     (io.rs:345-348); gzip by extension (.gz / .gzip) or by the 1f 8b magic, decoded the way flate2's GzDecoder
//     does — the FIRST member only (shk_inflate.h).  It parses; it never counts.
//   * the 2-bit packer (the reference's Read::from_str layout, encoding.rs:60-95) for batches that cross PCIe packed;
//   * writers for {sample}.histo, {sample}.final.histo (io.rs:1009-1014, 1051-1094) and {sample}.stats.yaml
//     (stats.rs:27-45,186-193; field order of RunStats); validate_args (cli.rs:645-677).
#include "../../include/shk.h"
#include "shk_front.h"
#include "shk_inflate.h"

#include <sched.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>

namespace shk {

std::string fmt(const char *f, ...) {
  va_list ap, ap2;
  va_start(ap, f);
  va_copy(ap2, ap);
  char buf[1024];
  const int n = vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  std::string out;
  if (n < (int)sizeof buf) {
    out.assign(buf, n > 0 ? (size_t)n : 0);
  } else {
    out.resize((size_t)n + 1);
    vsnprintf(&out[0], out.size(), f, ap2);
    out.resize((size_t)n);
  }
  va_end(ap2);
  return out;
}

uint32_t usable_cpus() {
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

std::string &run_error() {
  static thread_local std::string e;
  return e;
}

Pool::Pool(uint32_t T) {
  for (uint32_t t = 0; t < T; ++t) th.emplace_back([this] { run(); });
}
Pool::~Pool() {
  {
    std::lock_guard<std::mutex> lk(m);
    quit = true;
  }
  cv_job.notify_all();
  for (auto &t : th) t.join();
}
void Pool::run() {
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
void Pool::parallel_for(uint32_t n, std::function<void(uint32_t)> f) {
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

}  // namespace shk

using namespace shk;

// ---- FASTQ reader --------------------------------------------------------------------------------
// Front-end at speed, order-preserving.  Every input file has a PRODUCER: a SOURCE turns the
// file into WINDOWS of text — runs of whole lines: slices of the mapped file, buffers read from a pipe, buffers a
// gzip member is inflated into — and the producer's thread parses window after window with a pool of threads
// (newline positions per thread share, a prefix sum gives every line its number — a FASTQ record is exactly four
// lines, BufRead::lines: io.rs:271-352 — then lengths and flaws per record in parallel), handing on CHUNKS that say
// where the sequences lie.  The caller's thread (shk_fastq_next_batch) consumes the files strictly in input order,
// so the global read index — and with it the 1000-read chunk striping (io.rs:340-361), the validation cadence
// (io.rs:321-332) and --max-reads (io.rs:345-348) — is exactly the sequential reader's.  Producers of LATER files run
// ahead (up to 8 at a time, each bounded by its queue).
//
// What comes out, and when an error comes out, is what the reference would have INGESTED: it drains 1000 reads at a
// time (io.rs:340-343) and whatever it has read beyond the last full thousand is lost with the run when reading fails
// (`?` to main) — so a flawed record, a line that is not UTF-8, a stream that ends early or a file that cannot be
// opened at global record R only ever follows reads [0, ⌊R/1000⌋·1000), and an invalid base among THOSE is reported
// first (it was met when its thousand was drained), one among the rest never.  The consumer therefore SCOUTS ahead of
// what it hands out: reads are delivered up to the last thousand known to have been read whole, everything once the
// input is known to end cleanly.
namespace {

struct Buf {
  std::vector<uint8_t> v;
};
struct Window {
  std::shared_ptr<Buf> hold;  // what keeps the bytes alive (null: the producer's mapping)
  const char *p = nullptr;
  size_t n = 0;
  bool last = false;  // the file's last window (any other ends with '\n')
};

// Rust's str::from_utf8 (Unicode Table 3-7: no overlong forms, no surrogates, nothing above U+10FFFF)
static bool valid_utf8(const uint8_t *s, size_t n) {
  size_t i = 0;
  while (i < n) {
    const uint8_t c = s[i];
    if (c < 0x80) {
      ++i;
      continue;
    }
    uint8_t lo = 0x80, hi = 0xBF;
    size_t more;
    if (c >= 0xC2 && c <= 0xDF) more = 1;
    else if (c == 0xE0) more = 2, lo = 0xA0;
    else if (c >= 0xE1 && c <= 0xEC) more = 2;
    else if (c == 0xED) more = 2, hi = 0x9F;
    else if (c == 0xEE || c == 0xEF) more = 2;
    else if (c == 0xF0) more = 3, lo = 0x90;
    else if (c >= 0xF1 && c <= 0xF3) more = 3;
    else if (c == 0xF4) more = 3, hi = 0x8F;
    else return false;
    if (n - i <= more) return false;
    if (s[i + 1] < lo || s[i + 1] > hi) return false;
    for (size_t j = 2; j <= more; ++j)
      if ((s[i + j] & 0xC0) != 0x80) return false;
    i += more + 1;
  }
  return true;
}
static bool any_high_bit(const uint8_t *s, size_t n) {
  uint64_t acc = 0;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    memcpy(&w, s + i, 8);
    acc |= w;
  }
  uint8_t a = 0;
  for (; i < n; ++i) a |= s[i];
  return ((acc & 0x8080808080808080ull) | (a & 0x80)) != 0;
}

// ---- sources ---------------------------------------------------------------------------------------------------
struct Source {
  size_t window = 128u << 20;
  virtual ~Source() {}
  virtual bool next(Window *w) = 0;  // false: no window is left
  // What reading on behind the last byte reports (IO_NONE: a clean end of file), given the CRC-32 and the length of
  // all the windows' bytes.  Valid once the window with `last` set has been handed out.
  virtual IoError finish(uint32_t, uint64_t) { return IoError(); }
  virtual bool wants_crc() const { return false; }
  virtual void cancel() {}
};

struct MappedSource final : Source {  // a regular file, not gzip: slices of the mapping, cut at line ends
  const char *data;
  size_t size, pos = 0;
  MappedSource(const char *d, size_t n) : data(d), size(n) {}
  bool next(Window *w) override {
    if (pos >= size) return false;
    size_t want = window;
    for (;;) {
      const size_t end = size - pos <= want ? size : pos + want;
      if (end == size) {
        *w = Window{nullptr, data + pos, size - pos, true};
        pos = size;
        return true;
      }
      if (const void *nl = memrchr(data + pos, '\n', end - pos)) {
        const size_t cut = (size_t)((const char *)nl - data) + 1;
        *w = Window{nullptr, data + pos, cut - pos, false};
        pos = cut;
        return true;
      }
      want *= 2;  // a line longer than the window
    }
  }
};

struct FdSource final : Source {  // a pipe, stdin, anything that cannot be mapped: read() into buffers
  int fd;
  bool own_fd;
  std::vector<uint8_t> carry;  // bytes read but not handed out yet (a partial line, or what was read to sniff the magic)
  bool eof = false;
  IoError err;
  std::string err_text;
  FdSource(int f, bool own, std::vector<uint8_t> &&pre) : fd(f), own_fd(own), carry(std::move(pre)) {}
  ~FdSource() override {
    if (own_fd && fd >= 0) ::close(fd);
  }
  bool next(Window *w) override {
    if (eof) return false;
    auto b = std::make_shared<Buf>();
    size_t cap = carry.size() + window;
    b->v.resize(cap);
    if (!carry.empty()) memcpy(b->v.data(), carry.data(), carry.size());
    size_t have = carry.size();
    carry.clear();
    for (;;) {
      while (have < cap) {
        const ssize_t r = ::read(fd, b->v.data() + have, cap - have);
        if (r < 0) {
          if (errno == EINTR) continue;
          err_text = strerror(errno);
          err = IoError{IO_OTHER, err_text.c_str()};
          eof = true;
          break;
        }
        if (r == 0) {
          eof = true;
          break;
        }
        have += (size_t)r;
      }
      if (eof) {
        *w = Window{b, (const char *)b->v.data(), have, true};
        return true;
      }
      if (const void *nl = memrchr(b->v.data(), '\n', have)) {
        const size_t cut = (size_t)((const uint8_t *)nl - b->v.data()) + 1;
        carry.assign(b->v.data() + cut, b->v.data() + have);
        *w = Window{b, (const char *)b->v.data(), cut, false};
        return true;
      }
      cap *= 2;  // a line longer than the buffer
      b->v.resize(cap);
    }
  }
  IoError finish(uint32_t, uint64_t) override { return err; }
};

// One gzip member inflated into buffers by a thread of its own, running ahead of the parse by a few windows.  A
// buffer starts with the last 32 KiB (at least) of the one before — the decoder's history — which is also where an
// unfinished last line is carried over, so that every window is whole lines.
struct GzSource final : Source {
  GzMember gz;
  std::shared_ptr<void> owner;  // the compressed bytes (a mapping or a vector)
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::deque<Window> q;
  bool done = false, stop = false, started = false;
  InflateStatus final_status = INF_TRUNCATED;
  static constexpr size_t DEPTH = 3;
  GzSource(const uint8_t *p, const uint8_t *e, std::shared_ptr<void> own) : owner(std::move(own)) {
    window = 16u << 20;
    gz.open(p, e);
  }
  ~GzSource() override {
    cancel();
    if (th.joinable()) th.join();
  }
  void cancel() override {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
  }
  bool wants_crc() const override { return true; }
  void emit(Window &&w) {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return stop || q.size() < DEPTH; });
    if (stop) return;
    q.emplace_back(std::move(w));
    cv.notify_all();
  }
  void run() {
    if (gz.header_error.kind != IO_NONE) {  // nothing can be read: an empty last window, the error follows it
      emit(Window{nullptr, "", 0, true});
      std::lock_guard<std::mutex> lk(m);
      done = true;
      cv.notify_all();
      return;
    }
    size_t cap = window + 32768 + Inflater::OUT_SLACK;
    auto b = std::make_shared<Buf>();
    b->v.resize(cap);
    size_t pos = 0, line0 = 0;  // decoded so far in this buffer; start of the bytes not handed out yet
    for (;;) {
      {
        std::lock_guard<std::mutex> lk(m);
        if (stop) break;
      }
      const InflateStatus st = gz.inf.run(b->v.data(), &pos, b->v.size());
      if (st != INF_OUTPUT_FULL) {
        final_status = st;
        emit(Window{b, (const char *)b->v.data() + line0, pos - line0, true});
        break;
      }
      const void *nl = memrchr(b->v.data() + line0, '\n', pos - line0);
      if (!nl) {  // a line longer than the window: a bigger buffer, everything moves along
        auto nb = std::make_shared<Buf>();
        nb->v.resize(b->v.size() * 2);
        memcpy(nb->v.data(), b->v.data(), pos);
        b = std::move(nb);
        continue;
      }
      const size_t cut = (size_t)((const uint8_t *)nl - b->v.data()) + 1;
      const size_t keep = std::min(pos, std::max<size_t>(32768, pos - cut));
      auto nb = std::make_shared<Buf>();
      nb->v.resize(std::max(cap, keep + window + Inflater::OUT_SLACK));
      memcpy(nb->v.data(), b->v.data() + pos - keep, keep);
      emit(Window{b, (const char *)b->v.data() + line0, cut - line0, false});
      line0 = keep - (pos - cut);
      pos = keep;
      b = std::move(nb);
    }
    std::lock_guard<std::mutex> lk(m);
    done = true;
    cv.notify_all();
  }
  bool next(Window *w) override {
    std::unique_lock<std::mutex> lk(m);
    if (!started) {
      started = true;
      th = std::thread([this] { run(); });
    }
    cv.wait(lk, [&] { return stop || !q.empty() || done; });
    if (q.empty()) return false;
    *w = std::move(q.front());
    q.pop_front();
    cv.notify_all();
    return true;
  }
  IoError finish(uint32_t crc, uint64_t total) override { return gz.finish(final_status, crc, total); }
};

// ---- chunks ------------------------------------------------------------------------------------------------------
enum FlawKind : uint8_t { FLAW_FASTA, FLAW_HEADER, FLAW_SEP, FLAW_LEN, FLAW_UTF8 };
struct Flaw {  // what reading (FLAW_UTF8: always) or validate_fastq_record (io.rs:161-198: when the cadence says so)
               // would say about local record `rec` of a file
  uint64_t rec;
  FlawKind kind;
  uint8_t role = 0;   // FLAW_UTF8: which of the record's four lines
  std::string text;   // the header or separator line
  size_t seq_len = 0, qual_len = 0;
};
struct SeqChunk {
  // Where the sequences lie: line ℓ of the window is [ℓ ? nl[ℓ-1]+1 : 0, nl[ℓ]) from `src` on; the chunk's records are
  // — first, if has_lead, the record begun in the window before, its sequence copied into lead_seq — then the whole
  // records from line `line0` on (record j's sequence is line line0 + 4j + 1).  The consumer's thread pool copies
  // them straight into the caller's batch buffer.
  std::shared_ptr<Buf> hold;
  const char *src = nullptr;
  std::vector<uint32_t> nl;
  size_t line0 = 0;
  bool last_unterminated = false;  // line nl.size()-1 has no '\n' (its '\r', if any, stays: BufRead::lines)
  bool has_lead = false;
  std::string lead_seq;
  std::vector<uint32_t> lens;  // all records, the lead first
  std::vector<Flaw> flaws;     // ascending by rec
  uint64_t first_rec = 0;      // local index (within the file) of the chunk's first record
  size_t bytes() const { return lens.size() * 4 + nl.size() * 4 + lead_seq.size() + (hold ? hold->v.size() : 0); }
  void seq_line(size_t j, const char **p, size_t *len) const {  // j: index among the whole records
    const size_t l = line0 + 4 * j + 1;
    const size_t s0 = (size_t)nl[l - 1] + 1, s1 = nl[l];
    size_t n = s1 - s0;
    if (!(last_unterminated && l == nl.size() - 1) && n && src[s1 - 1] == '\r') --n;
    *p = src + s0;
    *len = n;
  }
};
// how a file ended
enum EndKind { END_CLEAN = 0, END_TRUNCATED, END_STREAM, END_OPEN, END_UTF8, END_PEEK };
struct FileEnd {
  EndKind kind = END_CLEAN;
  int role = 0;  // which line of the record was being read
  IoError err;   // END_STREAM
};

// first failing check of validate_fastq_record on a record's four lines
static bool find_flaw(const char *h, size_t hl, size_t sl, const char *sp, size_t spl, size_t ql, uint64_t rec, Flaw *out) {
  if (hl && h[0] == '>') {
    *out = Flaw{rec, FLAW_FASTA, 0, std::string(), 0, 0};
    return true;
  }
  if (!hl || h[0] != '@') {
    *out = Flaw{rec, FLAW_HEADER, 0, std::string(h, hl), 0, 0};
    return true;
  }
  if (!spl || sp[0] != '+') {
    *out = Flaw{rec, FLAW_SEP, 0, std::string(sp, spl), 0, 0};
    return true;
  }
  if (ql != sl) {
    *out = Flaw{rec, FLAW_LEN, 0, std::string(), sl, ql};
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
  static constexpr size_t Q_MAX = 256u << 20;  // bytes buffered ahead per file
  const char *map = nullptr;  // plain files: the mapping the chunks point into (released with the producer)
  size_t map_size = 0;
  std::unique_ptr<Source> src;
  Pool *pool = nullptr;
  uint32_t T = 1;
  ~Producer() {
    src.reset();
    if (map) munmap((void *)map, map_size);
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
  void request_cancel() {
    {
      std::lock_guard<std::mutex> lk(m);
      cancel = true;
    }
    cv.notify_all();
    // (src is created by the producer's own thread; it looks at `cancel` between windows and a GzSource is told then)
  }

  // open_fastq_reader (io.rs:598-625): gzip when the NAME ends in .gz / .gzip, or when the file starts with 1f 8b
  bool open_source() {
    const size_t window_env = [] {
      const char *wk = getenv("SHK_FASTQ_WINDOW_KB");  // test hook
      return wk ? std::max<size_t>(1, (size_t)atoll(wk)) << 10 : (size_t)0;
    }();
    auto ends_with = [&](const char *s) {
      const size_t n = strlen(s);
      return path.size() >= n && path.compare(path.size() - n, n, s) == 0;
    };
    auto with_window = [&](Source *s) {
      if (window_env) s->window = window_env;
      src.reset(s);
      return true;
    };
    if (path == "-") return with_window(new FdSource(0, false, {}));  // stdin is read as it is (io.rs:517-537)
    const bool gz_ext = ends_with(".gz") || ends_with(".gzip");
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) {
      end = FileEnd{END_OPEN, 0, {}};
      return false;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || S_ISDIR(st.st_mode)) {
      ::close(fd);
      end = FileEnd{END_PEEK, 0, {}};
      return false;
    }
    if (S_ISREG(st.st_mode)) {
      const size_t size = (size_t)st.st_size;
      const char *data = size ? (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : "";
      if (data != MAP_FAILED) {
        ::close(fd);
        if (size) {
          (void)madvise((void *)data, size, MADV_SEQUENTIAL);
          map = data;
          map_size = size;
        }
        const bool magic = size >= 2 && (uint8_t)data[0] == 0x1f && (uint8_t)data[1] == 0x8b;
        if (gz_ext || magic) return with_window(new GzSource((const uint8_t *)data, (const uint8_t *)data + size, nullptr));
        return with_window(new MappedSource(data, size));
      }
    }
    // not mappable: sniff the first bytes, then either stream it or — gzip — take all of it into memory
    std::vector<uint8_t> pre(1 << 16);
    size_t have = 0;
    while (have < 2) {
      const ssize_t r = ::read(fd, pre.data() + have, pre.size() - have);
      if (r < 0 && errno == EINTR) continue;
      if (r <= 0) break;
      have += (size_t)r;
    }
    pre.resize(have);
    const bool magic = have >= 2 && pre[0] == 0x1f && pre[1] == 0x8b;
    if (!(gz_ext || magic)) return with_window(new FdSource(fd, true, std::move(pre)));
    auto all = std::make_shared<std::vector<uint8_t>>(std::move(pre));
    for (;;) {
      const size_t old = all->size();
      all->resize(old + (4u << 20));
      const ssize_t r = ::read(fd, all->data() + old, all->size() - old);
      if (r < 0 && errno == EINTR) {
        all->resize(old);
        continue;
      }
      all->resize(old + (r > 0 ? (size_t)r : 0));
      if (r <= 0) break;
    }
    ::close(fd);
    return with_window(new GzSource(all->data(), all->data() + all->size(), all));
  }

  // ---- the parse ---------------------------------------------------------------------------------------------------
  std::vector<std::vector<uint32_t>> tl_nl;  // the newline positions each pool thread found in its share of the window
  // the lines of a record begun in one window and not finished there (at most three), as BufRead::lines yields them
  std::vector<std::string> carry;
  std::vector<uint8_t> carry_valid;  // are they UTF-8
  uint64_t rec = 0;                  // local index of the next record
  uint32_t crc = 0;
  uint64_t total_bytes = 0;
  IoError end_err;  // what the source reports behind its last byte

  void parse_window(const Window &w) {
    const char *base = w.p;
    const size_t n = w.n;
    const bool dbg = getenv("SHK_FASTQ_DEBUG") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_a = now();
    // 1. newline positions: every share of the window is scanned ONCE — a thread keeps the positions it finds in a list
    // of its own (kept across windows: no allocation after the first) — and, their places among all the lines known
    // from a prefix sum, the lists are copied into the window's line table (4 B per line, ≈ 5 % of the window's bytes).
    // The same pass notes whether a share has any byte ≥ 0x80 (only then is anything checked for UTF-8) and, for a gzip
    // member, sums the share's CRC-32.
    const uint32_t TT = n >= (1u << 16) ? T : 1;
    std::vector<size_t> first(TT + 1, 0);
    std::vector<uint8_t> high(TT, 0);
    std::vector<uint32_t> crcs(TT, 0);
    if (tl_nl.size() < TT) tl_nl.resize(TT);
    const bool want_crc = src->wants_crc();
    auto scan = [&](uint32_t t) {
      const size_t a = n * t / TT, b = n * (t + 1) / TT;
      const char *p = base + a, *e = base + b;
      std::vector<uint32_t> &v = tl_nl[t];
      v.clear();
      while (p < e) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!nl) break;
        v.push_back((uint32_t)(nl - base));
        p = nl + 1;
      }
      first[t + 1] = v.size();
      high[t] = any_high_bit((const uint8_t *)base + a, b - a);
      if (want_crc) crcs[t] = crc32_update(0, (const uint8_t *)base + a, b - a);
    };
    if (TT == 1) scan(0);
    else pool->parallel_for(TT, scan);
    const double t_b = now();
    bool any_high = false;
    for (uint32_t t = 0; t < TT; ++t) {
      first[t + 1] += first[t];
      any_high |= high[t] != 0;
      if (want_crc) crc = crc32_combine(crc, crcs[t], n * (t + 1) / TT - n * t / TT);
    }
    total_bytes += n;
    size_t M = first[TT];  // complete ('\n'-terminated) lines in the window
    std::vector<uint32_t> NL(M + 1);
    auto gather = [&](uint32_t t) {
      if (!tl_nl[t].empty()) memcpy(NL.data() + first[t], tl_nl[t].data(), tl_nl[t].size() * 4);
    };
    if (TT == 1) gather(0);
    else pool->parallel_for(TT, gather);
    bool last_unterminated = false;
    if (w.last) {
      end_err = src->finish(crc, total_bytes);
      // a last line without '\n' (it has ≥ 1 byte) is a line — unless reading on fails: then what had been read of it
      // goes with the error (read_line returns the Err)
      if (end_err.kind == IO_NONE && (M == 0 ? n > 0 : (size_t)NL[M - 1] + 1 < n)) {
        NL[M] = (uint32_t)n;
        ++M;
        last_unterminated = true;
      }
    }
    NL.resize(M);
    // line ℓ = [ℓ ? NL[ℓ-1]+1 : 0, NL[ℓ]), '\r' stripped (not from an unterminated last line)
    auto line_at = [&](size_t l, const char **p, size_t *len) {
      const size_t s0 = l ? (size_t)NL[l - 1] + 1 : 0, s1 = NL[l];
      size_t k = s1 - s0;
      if (!(last_unterminated && l == M - 1) && k && base[s1 - 1] == '\r') --k;
      *p = base + s0;
      *len = k;
    };
    SeqChunk c;
    c.first_rec = rec;
    // 2. the record begun in the window before: its remaining lines are this window's first
    size_t l0 = 0;
    if (!carry.empty()) {
      while (carry.size() < 4 && l0 < M) {
        const char *p;
        size_t len;
        line_at(l0++, &p, &len);
        carry.emplace_back(p, len);
        carry_valid.push_back(!any_high || valid_utf8((const uint8_t *)p, len));
      }
      if (carry.size() == 4) {
        Flaw f;
        bool flawed = false;
        for (int i = 0; i < 4 && !flawed; ++i)
          if (!carry_valid[i]) {
            f = Flaw{rec, FLAW_UTF8, (uint8_t)i, std::string(), 0, 0};
            flawed = true;
          }
        if (!flawed)
          flawed = find_flaw(carry[0].data(), carry[0].size(), carry[1].size(), carry[2].data(), carry[2].size(), carry[3].size(), rec, &f);
        if (flawed) c.flaws.emplace_back(std::move(f));
        c.has_lead = true;
        c.lead_seq = std::move(carry[1]);
        carry.clear();
        carry_valid.clear();
      }
    }
    const size_t R = (M - l0) / 4;  // whole records from line l0 on
    const size_t n_lead = c.has_lead ? 1 : 0;
    c.lens.resize(n_lead + R);
    if (c.has_lead) c.lens[0] = (uint32_t)c.lead_seq.size();
    const double t_c = now();
    // 3. sequence lengths + flaws, records shared out evenly (the sequences themselves stay where they are)
    const uint32_t TR = R >= 4096 ? T : 1;
    std::vector<std::vector<Flaw>> fl(TR);
    auto per_record = [&](uint32_t t) {
      const size_t r0 = R * t / TR, r1 = R * (t + 1) / TR;
      for (size_t r = r0; r < r1; ++r) {
        const char *h, *sq, *sp, *ql;
        size_t hl, sl, spl, qll;
        line_at(l0 + 4 * r, &h, &hl);
        line_at(l0 + 4 * r + 1, &sq, &sl);
        line_at(l0 + 4 * r + 2, &sp, &spl);
        line_at(l0 + 4 * r + 3, &ql, &qll);
        c.lens[n_lead + r] = (uint32_t)sl;
        Flaw f;
        if (any_high) {  // BufRead::lines: a line that is not UTF-8 is an error of the read, whatever the cadence
          const char *lp[4] = {h, sq, sp, ql};
          const size_t ll[4] = {hl, sl, spl, qll};
          int bad = -1;
          for (int i = 0; i < 4 && bad < 0; ++i)
            if (any_high_bit((const uint8_t *)lp[i], ll[i]) && !valid_utf8((const uint8_t *)lp[i], ll[i])) bad = i;
          if (bad >= 0) {
            fl[t].emplace_back(Flaw{rec + n_lead + r, FLAW_UTF8, (uint8_t)bad, std::string(), 0, 0});
            continue;
          }
        }
        if (find_flaw(h, hl, sl, sp, spl, qll, rec + n_lead + r, &f)) fl[t].emplace_back(std::move(f));
      }
    };
    if (TR == 1) per_record(0);
    else pool->parallel_for(TR, per_record);
    for (uint32_t t = 0; t < TR; ++t)
      for (auto &f : fl[t]) c.flaws.emplace_back(std::move(f));
    // 4. the lines of a record that goes on in the next window
    for (size_t l = l0 + 4 * R; l < M; ++l) {
      const char *p;
      size_t len;
      line_at(l, &p, &len);
      carry.emplace_back(p, len);
      carry_valid.push_back(!any_high || valid_utf8((const uint8_t *)p, len));
    }
    c.hold = w.hold;
    c.src = base;
    c.line0 = l0;
    c.last_unterminated = last_unterminated && l0 + 4 * R == M;  // (else the unterminated line is not among the records')
    NL.resize(l0 + 4 * R);
    c.nl = std::move(NL);
    rec += n_lead + R;
    const double t_d = now();
    if (n_lead + R) push(std::move(c));
    if (dbg)
      fprintf(stderr, "[fastq window %zu MB] scan %.1f ms  gather %.1f ms  lens %.1f ms  push(wait) %.1f ms\n", n >> 20, (t_b - t_a) * 1e3,
              (t_c - t_b) * 1e3, (t_d - t_c) * 1e3, (now() - t_d) * 1e3);
  }

  void run() {
    if (!open_source()) return finish(end);
    Window w;
    bool saw_last = false;
    while (!cancelled() && src->next(&w)) {
      parse_window(w);
      saw_last = w.last;
      w = Window();
    }
    if (cancelled()) {
      src->cancel();
      return finish(FileEnd{});
    }
    if (!saw_last) end_err = src->finish(crc, total_bytes);  // (an empty file: no window at all)
    // how the file ends, in the order a line-by-line reader meets it: the lines of an unfinished last record (one that is
    // not UTF-8 fails its own read), then whatever reading on reports, then the record being short of lines
    // (io.rs:287-318)
    for (size_t i = 0; i < carry.size(); ++i)
      if (!carry_valid[i]) return finish(FileEnd{END_UTF8, (int)i, {}});
    if (end_err.kind != IO_NONE) return finish(FileEnd{END_STREAM, (int)carry.size(), end_err});
    if (!carry.empty()) return finish(FileEnd{END_TRUNCATED, (int)carry.size(), {}});
    finish(FileEnd{});
  }
};

}  // namespace

struct shk_fastq {
  std::vector<std::string> paths;
  std::vector<std::unique_ptr<Producer>> prod;  // one per path; started up to LOOKAHEAD files ahead of the scout
  size_t started = 0;                           // producers started so far
  std::unique_ptr<Pool> pool, cpool;            // the producers' pool (window parse) and the consumer's (copy-out): they overlap
  uint32_t T = 1;
  uint64_t max_reads = 0, validate_every = 0;
  // -- the scout: what is known about the input ahead of what has been handed out
  struct Held {
    SeqChunk c;
    size_t n = 0;     // records of the chunk that count (--max-reads or an error may cut it short)
    size_t next = 0;  // the next one to hand out
  };
  std::deque<Held> held;
  size_t scout_file = 0;
  uint64_t scout_n = 0;  // records known to be readable (global)
  enum { RUNNING, AT_END, AT_MAX, FAILED } state = RUNNING;
  std::string pending_err;  // FAILED: what the reference reports, once everything it had drained is out
  int pending_code = 0;
  // -- FastqReadState, io.rs:205-206, of what has been handed out
  uint64_t n_reads_read = 0, n_bases_read = 0;
  bool reached_max = false, done = false;
  std::string err;
  int err_code = 0;
  static constexpr size_t LOOKAHEAD = 8;

  ~shk_fastq() { stop_all(); }
  void stop_all() {
    for (auto &p : prod)
      if (p) p->request_cancel();
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
      // — a container's CFS quota counts, not the host's core count — and the copying pool half of that
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
    while (started < paths.size() && started < scout_file + LOOKAHEAD) {
      auto p = std::make_unique<Producer>();
      p->path = paths[started];
      p->name = p->path == "-" ? "stdin" : p->path;
      p->pool = pool.get();
      p->T = T;
      Producer *pp = p.get();
      pp->th = std::thread([pp] { pp->run(); });
      prod.emplace_back(std::move(p));
      ++started;
    }
  }

  // io.rs:321-322: record 0 and every validate_every-th are validated — by GLOBAL record index
  bool cadence(uint64_t g) const { return g == 0 || (validate_every > 0 && g % validate_every == 0); }
  void fail_at(int code, std::string &&m) {
    state = FAILED;
    pending_code = code;
    pending_err = std::move(m);
    stop_all();
  }
  // stream_io_error (io.rs:213-265) for a local source
  static std::string stream_error(const char *role, const IoError &e, const std::string &name, uint64_t n_read) {
    if (e.kind == IO_UNEXPECTED_EOF)
      return fmt("Local read stream ended unexpectedly while reading %s line of record %llu in %s (I/O error: %s \xE2\x80\x94 kind %s). "
                 "The file may be truncated or corrupted.",
                 role, (unsigned long long)n_read + 1, name.c_str(), e.text, io_kind_name(e.kind));
    return fmt("Failed to read %s line of record %llu in %s: %s (kind %s)", role, (unsigned long long)n_read + 1, name.c_str(), e.text,
               io_kind_name(e.kind));
  }
  // validate_fastq_record's message for a flaw (io.rs:161-198) — or the failed read's, for a line that is not UTF-8 —
  // with the GLOBAL record number
  std::string flaw_text(const Flaw &f, uint64_t g, const std::string &name) const {
    static const char *role[4] = {"header", "sequence", "separator", "quality"};
    const unsigned long long recno = g + 1;
    auto first_char = [](const std::string &s) -> std::string {  // header.chars().next().unwrap_or(' ')
      if (s.empty()) return " ";
      const uint8_t c = (uint8_t)s[0];
      const size_t n = c < 0x80 ? 1 : c < 0xE0 ? 2 : c < 0xF0 ? 3 : 4;
      return s.substr(0, n);
    };
    switch (f.kind) {
      case FLAW_UTF8:
        return stream_error(role[f.role], IoError{IO_INVALID_DATA, "stream did not contain valid UTF-8"}, name, g);
      case FLAW_FASTA:
        return fmt("Input appears to be FASTA format, not FASTQ (record %llu starts with '>'). "
                   "sharkmer requires FASTQ input with quality scores.", recno);
      case FLAW_HEADER:
        return fmt("FASTQ record %llu has invalid header (expected '@', got '%s'): %s", recno, first_char(f.text).c_str(), f.text.c_str());
      case FLAW_SEP:
        return fmt("FASTQ record %llu has invalid separator line (expected '+', got '%s'): %s", recno, first_char(f.text).c_str(),
                   f.text.c_str());
      default:
        return fmt("FASTQ record %llu has mismatched sequence (%zu) and quality (%zu) lengths", recno, f.seq_len, f.qual_len);
    }
  }

  // One step of the scout: the next chunk of the file it is in — or that file's end.
  void scout_more() {
    static const char *role[4] = {"header", "sequence", "separator", "quality"};
    if (scout_file >= paths.size()) {
      state = AT_END;
      return;
    }
    start_producers();
    Producer *p = prod[scout_file].get();
    std::unique_lock<std::mutex> lk(p->m);
    p->cv.wait(lk, [&] { return !p->q.empty() || p->finished; });
    if (p->q.empty()) {  // this file is exhausted; state persists into the next one (io.rs:498-512)
      const FileEnd fe = p->end;
      lk.unlock();
      switch (fe.kind) {
        case END_OPEN: return fail_at(SHK_ERR_IO, fmt("Failed to open file: %s", p->path.c_str()));
        case END_PEEK: return fail_at(SHK_ERR_IO, "Failed to peek at file");
        case END_TRUNCATED:  // io.rs:291-317
          return fail_at(SHK_ERR_FASTQ, fmt("Truncated FASTQ record at record %llu in %s: missing %s line", (unsigned long long)scout_n + 1,
                                            p->name.c_str(), role[fe.role]));
        case END_STREAM: return fail_at(SHK_ERR_IO, stream_error(role[fe.role], fe.err, p->name, scout_n));
        case END_UTF8:
          return fail_at(SHK_ERR_IO, stream_error(role[fe.role], IoError{IO_INVALID_DATA, "stream did not contain valid UTF-8"}, p->name, scout_n));
        default: break;
      }
      if (p->th.joinable()) p->th.join();
      ++scout_file;
      return;
    }
    Held h;
    h.c = std::move(p->q.front());
    p->q.pop_front();
    p->q_bytes -= h.c.bytes();
    p->cv.notify_all();
    lk.unlock();
    h.n = h.c.lens.size();
    bool at_max = false;
    if (max_reads > 0 && scout_n + h.n >= max_reads) {  // io.rs:345-348: nothing behind the last read is looked at
      h.n = (size_t)(max_reads - scout_n);
      at_max = true;
    }
    const Flaw *hit = nullptr;
    for (const Flaw &f : h.c.flaws) {
      const uint64_t j = f.rec - h.c.first_rec;
      if (j >= h.n) break;
      if (f.kind == FLAW_UTF8 || cadence(scout_n + j)) {
        hit = &f;
        break;
      }
    }
    if (hit) {
      const uint64_t j = hit->rec - h.c.first_rec;
      std::string msg = flaw_text(*hit, scout_n + j, p->name);
      const int code = hit->kind == FLAW_UTF8 ? SHK_ERR_IO : SHK_ERR_FASTQ;
      h.n = (size_t)j;
      scout_n += h.n;
      if (h.n) held.emplace_back(std::move(h));
      return fail_at(code, std::move(msg));
    }
    scout_n += h.n;
    if (h.n) held.emplace_back(std::move(h));
    if (at_max) {
      state = AT_MAX;
      stop_all();
    }
  }
  // how many reads may be handed out, all in all
  uint64_t limit() const {
    if (state == AT_END || state == AT_MAX) return scout_n;
    return scout_n / SHK_READS_PER_BATCH * SHK_READS_PER_BATCH;
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
  uint64_t used = 0, n = 0;
  bool stop = false;
  while (!stop && n < max_seqs) {
    if (r->n_reads_read == r->limit()) {
      if (r->state == shk_fastq::RUNNING) {
        r->scout_more();
        continue;
      }
      break;
    }
    while (!r->held.empty() && r->held.front().next == r->held.front().n) r->held.pop_front();
    shk_fastq::Held &h = r->held.front();
    const SeqChunk &c = h.c;
    // hand out what fits of this chunk
    const size_t seq_begin = h.next;
    const uint64_t n_begin = n;
    const uint64_t room = std::min<uint64_t>(max_seqs - n, r->limit() - r->n_reads_read);
    size_t take = (size_t)std::min<uint64_t>(room, h.n - h.next);
    for (size_t j = 0; j < take; ++j) {
      const uint64_t len = c.lens[seq_begin + j];
      // (a sequence that no batch of this size can hold: reported when it is the FIRST of a batch, with nothing
      // consumed — the caller may come back with a larger buffer; shk_run_files does)
      if (len > bases_cap && n == 0) {
        r->err = "sequence longer than the batch buffer";  // (not sticky: the same call with a larger buffer goes on)
        return SHK_ERR_BAD_ARG;
      }
      if (used + len > bases_cap) {  // does not fit: it is delivered first thing next call
        take = j;
        stop = true;
        break;
      }
      used += len;
      offsets[++n] = used;
    }
    if (take) {  // the sequences of records [seq_begin, seq_begin + take): where they lie → batch buffer
      size_t j0 = 0;
      if (c.has_lead && seq_begin == 0) {
        memcpy(bases + offsets[n_begin], c.lead_seq.data(), c.lead_seq.size());
        j0 = 1;
      }
      const size_t lead = c.has_lead ? 1 : 0;
      const size_t cnt = take - j0;
      const uint32_t TT = cnt >= 4096 ? r->cpool->size() : 1;
      auto copy = [&](uint32_t t) {
        const size_t a = j0 + cnt * t / TT, b = j0 + cnt * (t + 1) / TT;
        for (size_t j = a; j < b; ++j) {
          const char *sq;
          size_t sl;
          c.seq_line(seq_begin + j - lead, &sq, &sl);
          memcpy(bases + offsets[n_begin + j], sq, sl);
        }
      };
      if (TT == 1) copy(0);
      else r->cpool->parallel_for(TT, copy);
      h.next += take;
      r->n_reads_read += take;                    // io.rs:337
      r->n_bases_read += used - offsets[n_begin];  // io.rs:335 (N included)
    }
  }
  if (r->n_reads_read == r->limit() && r->state != shk_fastq::RUNNING) {
    if (r->state == shk_fastq::FAILED) {
      if (n == 0) return r->fail(r->pending_code, r->pending_err);
    } else {
      r->reached_max = r->state == shk_fastq::AT_MAX;
      r->done = true;
      r->held.clear();
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

const char *shk_run_error(void) { return run_error().c_str(); }

int shk_validate_args(uint32_t k, uint64_t histo_max, const char *sample) {
  // cli.rs:659-673 and cli.rs:645-652, same messages
  std::string &e = run_error();
  if (!(k < 32)) {
    e = "k must be less than 32 due to use of 64 bit integers to encode kmers";
    return SHK_ERR_BAD_ARG;
  }
  if (!(k > 0)) {
    e = "k must be greater than 0";
    return SHK_ERR_BAD_ARG;
  }
  if (k % 2 != 1) {
    e = "k must be odd";
    return SHK_ERR_BAD_ARG;
  }
  if (!(histo_max > 0)) {
    e = "histo_max must be greater than 0";
    return SHK_ERR_BAD_ARG;
  }
  if (histo_max > 1000000) {
    e = fmt("histo_max must not exceed 1000000, got %llu", (unsigned long long)histo_max);
    return SHK_ERR_BAD_ARG;
  }
  if (!sample) {
    e = "--sample is required. Provide a sample name as output file prefix.\n"
        "When using --ena, the sample name can be derived automatically from ENA metadata.";
    return SHK_ERR_BAD_ARG;
  }
  for (const char *p = sample; *p; ++p) {
    unsigned char ch = (unsigned char)*p;
    bool ok = (ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || ch == '_' ||
              ch == '-' || ch == '.' || ch >= 0x80;
    if (!ok) {
      e = fmt("Sample name '%s' contains characters that are unsafe for filenames. "
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
  std::string &err = run_error();
  err.clear();
  if (n_bases == 0) return SHK_OK;
  if (!bases || !packed || !nmask) {
    err = "null buffer";
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
  uint32_t T = n_threads ? n_threads : std::max(1u, std::min(32u, usable_cpus()));
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
    err = fmt("Invalid character '%s' in sequence. Only ACGTN allowed.", byte_as_char((uint8_t)(first & 0xFF)).c_str());
    return SHK_ERR_INVALID_CHAR;
  }
  return SHK_OK;
}

}  // extern "C"
