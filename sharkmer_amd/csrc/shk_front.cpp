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

#include <immintrin.h>
#include <sched.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <new>

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

// A window's bytes when they are not the mapped file's: a block that is neither zero-filled (std::vector would: 16 MiB
// per inflated window, a tenth of the inflate thread's time) nor handed back to the allocator after every window (a block
// this large comes from mmap: thousands of page faults each time) — a few blocks of each size in use are kept for reuse.
struct Buf {
  struct Bytes {
    uint8_t *p = nullptr;
    size_t n = 0;
    uint8_t *data() const { return p; }
    size_t size() const { return n; }
    void resize(size_t m) {  // (contents are NOT kept: every caller copies what it needs itself)
      release(p, n);
      p = acquire(m);
      n = m;
    }
    ~Bytes() { release(p, n); }
    Bytes() = default;
    Bytes(const Bytes &) = delete;
    Bytes &operator=(const Bytes &) = delete;
  } v;

 private:
  struct Cache {
    std::mutex m;
    std::vector<std::pair<size_t, uint8_t *>> blocks;
    ~Cache() {
      for (auto &b : blocks) free(b.second);
    }
  };
  static Cache &cache() {
    static Cache c;
    return c;
  }
  static uint8_t *acquire(size_t n) {
    if (!n) return nullptr;
    {
      Cache &c = cache();
      std::lock_guard<std::mutex> lk(c.m);
      for (size_t i = 0; i < c.blocks.size(); ++i)
        if (c.blocks[i].first == n) {
          uint8_t *p = c.blocks[i].second;
          c.blocks.erase(c.blocks.begin() + (long)i);
          return p;
        }
    }
    uint8_t *p = (uint8_t *)malloc(n);
    if (!p) throw std::bad_alloc();
    return p;
  }
  static void release(uint8_t *p, size_t n) {
    if (!p) return;
    Cache &c = cache();
    {
      std::lock_guard<std::mutex> lk(c.m);
      if (c.blocks.size() < 12) {
        c.blocks.emplace_back(n, p);
        return;
      }
    }
    free(p);
  }
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

// An array that is not value-initialised (the line tables of a 128 MiB window are megabytes: zero-filling them, by one
// thread, was a third of the window's parse).
template <class T>
struct UBuf {
  std::unique_ptr<T[]> p;
  size_t n = 0, cap = 0;
  void resize_discard(size_t m) {  // (contents are lost when it has to grow)
    if (m > cap) {
      p.reset(new T[m]);
      cap = m;
    }
    n = m;
  }
  T *data() { return p.get(); }
  const T *data() const { return p.get(); }
  size_t size() const { return n; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
};
constexpr uint32_t NL_POS = 0x7FFFFFFFu, NL_CR = 0x80000000u;  // a line-table entry: position of the '\n' | "the byte before it is '\r'"

// One pass over a share [a, b) of a window of n bytes: for every '\n' in it an entry (its position relative to `base`,
// NL_CR when the byte in front of it is '\r' — what BufRead::lines strips with it) appended to `nl`, and the byte
// behind it — the first byte of the next line, which is all validate_fastq_record looks at in a header or separator
// (io.rs:169-189) — appended to `fb` (0 at the window's end); returns whether any byte of the share has its top bit
// set.  With these the per-record pass never touches the file's bytes again.  The AVX2 form takes 64 bytes per step —
// the compare masks of two vectors as one 64-bit word, an entry per set bit — and reads the top bits off the same
// vectors (vpmovmskb IS the top bits); memchr line by line (a FASTQ record is four short lines) and a second pass for
// the top bits took three times as long.
static inline void scan_emit(const char *base, size_t n, size_t i, uint32_t *nl, uint8_t *fb, size_t at) {
  nl[at] = (uint32_t)i | ((i && base[i - 1] == '\r') ? NL_CR : 0u);
  fb[at] = i + 1 < n ? (uint8_t)base[i + 1] : (uint8_t)0;
}
static bool scan_share_scalar(const char *base, size_t n, size_t a, size_t b, std::vector<uint32_t> &nl, std::vector<uint8_t> &fb) {
  const char *p = base + a, *e = base + b;
  while (p < e) {
    const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
    if (!q) break;
    nl.push_back(0);
    fb.push_back(0);
    scan_emit(base, n, (size_t)(q - base), nl.data(), fb.data(), nl.size() - 1);
    p = q + 1;
  }
  return any_high_bit((const uint8_t *)base + a, b - a);
}
__attribute__((target("avx2,bmi,bmi2"))) static bool scan_share_avx2(const char *base, size_t n, size_t a, size_t b, std::vector<uint32_t> &nl,
                                                                   std::vector<uint8_t> &fb) {
  size_t i = a;
  uint32_t hi = 0;
  const __m256i nlv = _mm256_set1_epi8('\n');
  // room for the worst case of one step (64 entries) is kept ahead of the write cursor
  size_t at = nl.size();
  size_t cap = std::max<size_t>(nl.capacity(), at + 4096);
  nl.resize(cap);
  fb.resize(cap);
  uint32_t *w = nl.data();
  uint8_t *f = fb.data();
  for (; i + 64 <= b; i += 64) {
    const __m256i v0 = _mm256_loadu_si256((const __m256i *)(base + i));
    const __m256i v1 = _mm256_loadu_si256((const __m256i *)(base + i + 32));
    hi |= (uint32_t)_mm256_movemask_epi8(_mm256_or_si256(v0, v1));
    uint64_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v0, nlv)) |
                 ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v1, nlv)) << 32);
    if (m) {
      if (at + 64 > cap) {
        cap *= 2;
        nl.resize(cap);
        fb.resize(cap);
        w = nl.data();
        f = fb.data();
      }
      do {
        scan_emit(base, n, i + (size_t)_tzcnt_u64(m), w, f, at++);
        m = _blsr_u64(m);
      } while (m);
    }
  }
  nl.resize(at);
  fb.resize(at);
  bool h = hi != 0;
  if (i < b) h |= scan_share_scalar(base, n, i, b, nl, fb);
  return h;
}
static bool have_avx2() {  // (SHK_NO_AVX2: the portable forms, for the tests)
  static const bool on = !getenv("SHK_NO_AVX2") && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
  return on;
}
static bool scan_share(const char *base, size_t n, size_t a, size_t b, std::vector<uint32_t> &nl, std::vector<uint8_t> &fb) {
  return have_avx2() ? scan_share_avx2(base, n, a, b, nl, fb) : scan_share_scalar(base, n, a, b, nl, fb);
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
  // (Dropping a window's pages from the page table as soon as its chunk has been handed out — MADV_DONTNEED, so that
  // the final munmap has nothing left to do — was measured: the job got 8 ms slower, not 8 ms faster; every call shoots
  // down the TLBs of the thirty-odd threads that share the address space.  The mapping goes in one piece, at close.)
  bool next(Window *w) override {
    if (pos >= size) return false;
    // (the first window is a small one: the consumer — and behind it the GPU — has something to do 4 ms sooner)
    size_t want = pos == 0 ? std::min<size_t>(window, 16u << 20) : window;
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
      {
        auto nb = std::make_shared<Buf>();
        nb->v.resize(cap);
        memcpy(nb->v.data(), b->v.data(), have);
        b = std::move(nb);
      }
    }
  }
  IoError finish(uint32_t, uint64_t) override { return err; }
};

// One gzip member inflated into buffers by a thread of its own, running ahead of the parse by a few windows.  A
// buffer starts with the last 32 KiB (at least) of the one before — the decoder's history — which is also where an
// unfinished last line is carried over, so that every window is whole lines.
// One member at a time gets the many-thread decoder.  The reader starts the producers of up to LOOKAHEAD files ahead, so
// that slow (one-thread) streams run side by side; eight large .gz files would otherwise each bring their own sixteen
// workers — 128 threads on 16 CPUs, all files crawling, the one the consumer is waiting for among them.  Tickets are
// drawn in file order when the sources are made and served in that order: file i + 1's decode begins the moment file
// i's ends (file i's windows are still being parsed and handed out then).  A source that does not use the parallel
// decoder hands its ticket back.
struct PgzGate {
  std::mutex m;
  std::condition_variable cv;
  uint64_t next = 0, serving = 0;
  std::vector<uint64_t> returned;  // tickets given back before their turn
  uint64_t draw() {
    std::lock_guard<std::mutex> lk(m);
    return next++;
  }
  void skip_returned() {  // (m held)
    for (bool again = true; again;) {
      again = false;
      for (size_t i = 0; i < returned.size(); ++i)
        if (returned[i] == serving) {
          returned.erase(returned.begin() + (long)i);
          ++serving;
          again = true;
          break;
        }
    }
  }
  bool enter(uint64_t t, const std::atomic<bool> &abort) {  // false: the caller was cancelled while it waited
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return serving == t || abort.load(); });
    return serving == t;
  }
  void poke() {
    std::lock_guard<std::mutex> lk(m);
    cv.notify_all();
  }
  void leave(uint64_t t) {  // after enter(t), or instead of it
    std::lock_guard<std::mutex> lk(m);
    if (serving == t) {
      ++serving;
    } else {
      returned.push_back(t);
    }
    skip_returned();
    cv.notify_all();
  }
};
// (The gate belongs to ONE reader — shk_fastq::gate — not to the process: tickets only mean something inside one
// reader's file order, and a source holds its turn until its member is decoded, which only advances while THAT reader
// is consumed.  With a process-wide gate a second reader on a large .gz waited for the first to be drained — one thread
// taking two readers in lockstep, as read_fastq_paired does with R1 / R2 (io.rs:629-700), never came back.)

struct GzSource final : Source {
  GzMember gz;
  PgzGate *gate;            // the reader's gate
  uint64_t ticket;          // this source's turn at the many-thread decoder
  bool ticket_done = false;
  std::atomic<bool> gate_abort{false};  // cancel(): stop waiting for the turn
  void ticket_leave() {
    if (!ticket_done) {
      ticket_done = true;
      gate->leave(ticket);
    }
  }
  std::shared_ptr<void> owner;  // the compressed bytes (a mapping or a vector)
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::deque<Window> q;
  bool done = false, stop = false, started = false;
  InflateStatus final_status = INF_TRUNCATED;
  // NOT the reference (flate2::read::GzDecoder reads ONE member, and so does this reader unless asked otherwise): on
  // request — shk_fastq_open_ex, SHK_FASTQ_GZIP_ALL_MEMBERS — every member, as flate2::read::MultiGzDecoder would:
  // what follows a member's trailer is the next member's header; each member's CRC-32 and ISIZE are checked where it
  // ends (by this thread: wants_crc() is then false), its error is the stream's.
  bool all_members = false;
  IoError members_err;
  static constexpr size_t DEPTH = 3;
  const uint8_t *file_base;  // the file's first byte (file offsets decide where flate2's reads begin)
  GzSource(const uint8_t *p, const uint8_t *e, std::shared_ptr<void> own, PgzGate *gt, uint64_t turn) : gate(gt), ticket(turn), owner(std::move(own)), file_base(p) {
    window = 16u << 20;
    gz.open(p, e);
  }
  // ---- what the reference has SEEN of a stream when its decoder reports damage --------------------------------------
  // flate2's zio::read (GzDecoder over its own 32 KiB BufReader, under std's 8 KiB BufReader, io.rs:606-617) returns
  // `Err("corrupt deflate stream")` for the whole call in which the decoder met the damage: what that call had already
  // written into the caller's 8 KiB buffer is lost with it.  A call begins where the one before it ended — when the
  // 8 KiB were full, when the 32 KiB of compressed input ran out, or at a member's end — so of the `n_ok` bytes in
  // front of the damage the reference's line reader gets
  //     seen = o + 8192 · ⌊(n_ok − o) / 8192⌋,   o = the output in front of the last such input (or member) boundary.
  // HOLD_BACK: no window is handed on while fewer than this many decoded bytes lie behind its end, so that the cut
  // never has to take back anything (the reader parses, and hands out, what the windows hold).
  static constexpr size_t HOLD_BACK = 8192;
  // bytes a decoder gets out of [from, to) alone (a stream cut short at `to`: everything whose bits lie in front of it)
  static uint64_t output_of_prefix(const uint8_t *from, const uint8_t *to) {
    Inflater inf;
    inf.seek(from, to, 0);
    std::vector<uint8_t> buf((1u << 20) + 32768 + Inflater::OUT_SLACK);
    uint64_t total = 0;
    size_t pos = 0;
    for (;;) {
      const size_t before = pos;
      const InflateStatus st = inf.run(buf.data(), &pos, buf.size());
      total += pos - before;
      if (st != INF_OUTPUT_FULL) return total;
      const size_t keep = std::min<size_t>(pos, 32768);
      memmove(buf.data(), buf.data() + pos - keep, keep);
      pos = keep;
    }
  }
  // n_ok: the stream's bytes in front of the damage; member_out0: the stream's bytes in front of the damaged member;
  // deflate0: where that member's DEFLATE data begins; fail: the input byte the decoder stood at
  uint64_t reference_has_seen(uint64_t n_ok, uint64_t member_out0, const uint8_t *deflate0, const uint8_t *fail) const {
    const uint64_t fail_off = (uint64_t)(fail - file_base), b_off = fail_off & ~(uint64_t)32767;
    uint64_t o = member_out0;
    if (b_off > (uint64_t)(deflate0 - file_base)) o += output_of_prefix(deflate0, file_base + b_off);
    if (o > n_ok) o = n_ok;  // (cannot be: the prefix decodes like the whole)
    return o + (n_ok - o) / 8192 * 8192;
  }
  ~GzSource() override {
    cancel();
    if (th.joinable()) th.join();
    ticket_leave();  // (never started, or stopped early)
  }
  void cancel() override {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv.notify_all();
    gate_abort.store(true);
    gate->poke();
  }
  bool wants_crc() const override { return !all_members; }
  void emit(Window &&w) {
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return stop || q.size() < DEPTH; });
    if (stop) return;
    q.emplace_back(std::move(w));
    cv.notify_all();
  }
  // ---- one member, many threads -------------------------------------------------------------------------------------
  // A DEFLATE stream is one long dependency chain (every block's matches reach into the 32 KiB before it), and one
  // thread inflates ≈1.2 GB/s of FASTQ — 0.6 Gbases/s, three orders of magnitude under the counting rate.  The
  // two-pass scheme of parallel gzip readers (pugz, rapidgzip) breaks the chain:
  //   1. the compressed bytes are cut into chunks; a worker per chunk FINDS a block boundary behind its cut (a bit
  //      position where a dynamic-Huffman header parses, its block decodes and another header follows) and decodes from
  //      there SPECULATIVELY into 16-bit symbols — a byte, or "byte w of the 32 KiB in front of my entry point, which I
  //      do not know" (Inflater::run_symbols) — up to the first block boundary behind the next cut;
  //   2. this thread goes through the chunks in order.  A chunk's result counts only if its entry point is EXACTLY
  //      where the verified decoding before it ended; what lies between (a stored or fixed block the finder does not
  //      look for, a chunk whose speculation failed) is decoded here, the ordinary way.  With the window in front of a
  //      chunk known, its symbols become bytes — any of them independently of the others, so the last 32 KiB (the next
  //      chunk's window) at once, here, and all of them by the workers, chunks side by side.
  // What comes out — bytes, their order, the final status and where the stream ended — is the sequential decoder's:
  // nothing speculative is ever handed on unverified, and an error is only ever reported by a verified position.
  struct SpecResult {
    UBuf<uint16_t> sym;
    size_t n = 0;
    bool found = false, ready = false;
    uint64_t start_bit = 0, end_bit = 0;
    InflateStatus status = INF_CORRUPT;
    const uint8_t *after = nullptr;
    double seconds = 0, find_seconds = 0;  // (debug: the worker's time on this chunk, and of it the search for the entry point)
  };
  struct Workers {  // a small two-priority pool of the decoder's own
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> hi, lo;
    bool quit = false;
    // n threads take whatever there is, urgent work first; n_urgent more take urgent work only — a chunk's bytes are
    // wanted NOW (the windows go out in order), and a worker deep in a 10 ms speculative decode is no help with that
    Workers(uint32_t n, uint32_t n_urgent) {
      for (uint32_t i = 0; i < n + n_urgent; ++i)
        th.emplace_back([this, only_urgent = i >= n] {
          for (;;) {
            std::function<void()> f;
            {
              std::unique_lock<std::mutex> lk(m);
              cv.wait(lk, [&] { return quit || !hi.empty() || (!only_urgent && !lo.empty()); });
              if (hi.empty() && (only_urgent || lo.empty())) {
                if (quit) return;
                continue;
              }
              auto &q = hi.empty() ? lo : hi;
              f = std::move(q.front());
              q.pop_front();
            }
            f();
          }
        });
    }
    void post(bool urgent, std::function<void()> f) {
      {
        std::lock_guard<std::mutex> lk(m);
        (urgent ? hi : lo).emplace_back(std::move(f));
      }
      cv.notify_all();
    }
    ~Workers() {
      {
        std::lock_guard<std::mutex> lk(m);
        quit = true;
        lo.clear();  // (speculation nobody will look at)
      }
      cv.notify_all();
      for (auto &t : th) t.join();
    }
  };
  static bool plausible_header_at(const uint8_t *origin, const uint8_t *end, uint64_t bit) {
    // BFINAL = 0, BTYPE = 2 (bits 0, 0, 1), HLIT ≤ 29, HDIST ≤ 29: what an encoder's dynamic, non-final block starts with
    const uint8_t *p = origin + (bit >> 3);
    if (end - p < 4) return false;
    uint32_t w;
    memcpy(&w, p, 4);
    w >>= (uint32_t)(bit & 7);
    return (w & 7u) == 4u && ((w >> 3) & 31u) <= 29u && ((w >> 8) & 31u) <= 29u;
  }
  // The speculative decode of chunk `i`: entry point at or behind bit lo, output up to the first block boundary at or
  // behind bit hi.
  static void speculate(SpecResult *r, const uint8_t *origin, const uint8_t *end, uint64_t lo, uint64_t hi, size_t guess) {
    const auto t_begin = std::chrono::steady_clock::now();
    struct Note {
      SpecResult *r;
      std::chrono::steady_clock::time_point t0;
      ~Note() { r->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    } note{r, t_begin};
    Inflater inf;
    size_t cap = guess + 65536;
    r->sym.resize_discard(cap);
    auto grow = [&](size_t keep) {
      UBuf<uint16_t> nb;
      nb.resize_discard(cap * 2);
      memcpy(nb.data(), r->sym.data(), keep * 2);
      r->sym = std::move(nb);
      cap *= 2;
    };
    const uint64_t total_bits = (uint64_t)(end - origin) * 8;
    for (uint64_t p = lo; p < hi && p + 64 < total_bits; ++p) {
      if (!plausible_header_at(origin, end, p)) continue;
      inf.seek(origin, end, p);
      if (inf.read_block_header() != INF_OUTPUT_FULL || inf.state != 2 || inf.last_block || !inf.hdr_plausible) continue;
      // its block must decode, and a header must follow it
      size_t pos = 0;
      bool between = false;
      InflateStatus st;
      for (;;) {
        st = inf.run_symbols(r->sym.data(), &pos, cap, origin, p + 1, &between);
        if (st == INF_OUTPUT_FULL && !between) {
          grow(pos);
          continue;
        }
        break;
      }
      if (st != INF_OUTPUT_FULL || !between) continue;
      {
        Inflater peek = inf;
        const InflateStatus hs = peek.read_block_header();
        if (hs == INF_CORRUPT) continue;
      }
      // accepted: go on to the first block boundary at or behind `hi`
      r->found = true;
      r->start_bit = p;
      r->find_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
      for (;;) {
        st = inf.run_symbols(r->sym.data(), &pos, cap, origin, hi, &between);
        if (st == INF_OUTPUT_FULL && !between) {
          grow(pos);
          continue;
        }
        break;
      }
      r->n = pos;
      r->status = st;
      r->end_bit = inf.bit_position(origin);
      if (st == INF_STREAM_END) r->after = inf.input_after_stream();
      return;
    }
  }

  void run_parallel(uint32_t n_threads, size_t chunk_bytes) {
    const uint8_t *origin = gz.inf.in, *end = gz.inf.in_end;
    const size_t n_chunks = ((size_t)(end - origin) + chunk_bytes - 1) / chunk_bytes;
    std::vector<SpecResult> spec(n_chunks);
    std::mutex rm;
    std::condition_variable rcv;
    Workers pool(n_threads, std::max(2u, n_threads / 4));
    size_t issued = 1;  // (chunk 0 starts at the stream's start: decoded here, the ordinary way)
    const size_t ahead = (size_t)n_threads + 2;
    const size_t guess = chunk_bytes * 6;
    auto issue_up_to = [&](size_t upto) {
      for (; issued < n_chunks && issued <= upto; ++issued) {
        const size_t i = issued;
        pool.post(false, [&, i] {
          speculate(&spec[i], origin, end, (uint64_t)i * chunk_bytes * 8, (uint64_t)(i + 1) * chunk_bytes * 8, guess);
          {
            std::lock_guard<std::mutex> lk(rm);
            spec[i].ready = true;
          }
          rcv.notify_all();
        });
      }
    };
    issue_up_to(ahead);

    // a piece of output on its way out: [window w_len][n bytes] in one block; `pending` jobs still write into it
    // (a count the workers take down and this thread sleeps on — it used to spin with yield())
    struct Pending {
      std::mutex m;
      std::condition_variable cv;
      int n = 0;
      void set(int k) {
        std::lock_guard<std::mutex> lk(m);
        n = k;
      }
      void done() {
        std::lock_guard<std::mutex> lk(m);
        if (--n <= 0) cv.notify_all();
      }
      void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return n <= 0; });
      }
    };
    struct Piece {
      std::shared_ptr<Buf> buf;
      size_t w_len = 0, n = 0;
      std::shared_ptr<Pending> pending;
    };
    std::deque<Piece> out_q;
    std::vector<uint8_t> W;       // the last ≤ 32 KiB of everything decoded so far
    std::vector<uint8_t> carry;   // the bytes of the last, unfinished line (they are also the tail of W when they fit)
    uint64_t verified = 0;        // bit position everything in front of which has been decoded and checked
    InflateStatus status = INF_OUTPUT_FULL;
    const uint8_t *after = nullptr;
    uint64_t emitted_out = 0;  // bytes of the pieces that have left out_q
    bool cancelled = false;
    size_t n_spec_used = 0, n_here = 0;  // chunks taken from the workers / stretches decoded by this thread
    uint64_t bytes_spec = 0, bytes_here = 0;
    double t_workers = 0, t_find = 0, t_wait_spec = 0, t_wait_resolve = 0, t_emit = 0;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

    auto is_stopped = [&] {
      std::lock_guard<std::mutex> lk(m);
      return stop;
    };
    // hand the front piece on, as a window of whole lines (its last line's beginning is carried into the next)
    auto emit_front = [&](bool last) {
      Piece pc = std::move(out_q.front());
      out_q.pop_front();
      {
        const double t0 = tnow();
        if (pc.pending) pc.pending->wait();
        t_wait_resolve += tnow() - t0;
      }
      const double t_e0 = tnow();
      struct EmitNote {
        double &acc, t0;
        decltype(tnow) &now;
        ~EmitNote() { acc += now() - t0; }
      } emit_note{t_emit, t_e0, tnow};
      uint8_t *base = pc.buf->v.data();
      size_t a = pc.w_len, b = pc.w_len + pc.n;
      if (!carry.empty()) {  // the line begun in the piece before: it lies right in front of this piece's bytes
        if (carry.size() <= pc.w_len && memcmp(base + pc.w_len - carry.size(), carry.data(), carry.size()) == 0) {
          a = pc.w_len - carry.size();
        } else {  // (longer than the window in front: a block of its own)
          auto nb = std::make_shared<Buf>();
          nb->v.resize(carry.size() + pc.n + 64);
          memcpy(nb->v.data(), carry.data(), carry.size());
          memcpy(nb->v.data() + carry.size(), base + pc.w_len, pc.n);
          pc.buf = nb;
          base = nb->v.data();
          a = 0;
          b = carry.size() + pc.n;
        }
      }
      if (last) {
        carry.clear();
        emit(Window{pc.buf, (const char *)base + a, b - a, true});
        return;
      }
      const void *nl = b > a ? memrchr(base + a, '\n', b - a) : nullptr;
      if (!nl) {  // no line ends in this piece: all of it is carried on
        carry.assign(base + a, base + b);
        return;
      }
      const size_t cut = (size_t)((const uint8_t *)nl - base) + 1;
      carry.assign(base + cut, base + b);
      emit(Window{pc.buf, (const char *)base + a, cut - a, false});
    };
    auto new_piece_block = [&](size_t n) {
      auto b = std::make_shared<Buf>();
      b->v.resize(W.size() + n + Inflater::OUT_SLACK + 64);
      if (!W.empty()) memcpy(b->v.data(), W.data(), W.size());
      return b;
    };
    auto roll_window = [&](const uint8_t *bytes, size_t n) {  // W := last 32 KiB of W ++ bytes
      if (n >= 32768) {
        W.assign(bytes + n - 32768, bytes + n);
      } else {
        const size_t keep = std::min(W.size(), (size_t)32768 - n);
        std::vector<uint8_t> nw(W.end() - (long)keep, W.end());
        nw.insert(nw.end(), bytes, bytes + n);
        W.swap(nw);
      }
    };
    // the ordinary decoder from the verified position on, up to the first block boundary at or behind `to_bit`
    auto decode_here = [&](uint64_t to_bit) {
      Inflater inf;
      inf.seek(origin, end, verified);
      inf.stop_origin = origin;
      inf.stop_bit = to_bit;
      size_t cap = std::max<size_t>(chunk_bytes * 6, 1u << 20);
      auto b = new_piece_block(cap);
      const size_t w_len = W.size();
      size_t pos = w_len;
      InflateStatus st;
      for (;;) {
        st = inf.run(b->v.data(), &pos, w_len + cap);
        if (st == INF_OUTPUT_FULL && !inf.stopped_between_blocks) {  // more room
          auto nb = std::make_shared<Buf>();
          nb->v.resize(w_len + cap * 2 + Inflater::OUT_SLACK + 64);
          memcpy(nb->v.data(), b->v.data(), pos);
          b = std::move(nb);
          cap *= 2;
          if (is_stopped()) {
            cancelled = true;
            break;
          }
          continue;
        }
        break;
      }
      Piece pc;
      pc.buf = b;
      pc.w_len = w_len;
      pc.n = pos - w_len;
      roll_window(b->v.data() + w_len, pc.n);
      ++n_here;
      bytes_here += pc.n;
      out_q.emplace_back(std::move(pc));
      verified = inf.bit_position(origin);
      if (st != INF_OUTPUT_FULL) {
        status = st;
        if (st == INF_STREAM_END) after = inf.input_after_stream();
      }
    };
    // a verified chunk: its window is known now — the next chunk's window at once, its bytes by the workers
    auto resolve = [&](SpecResult &r) {
      const size_t n = r.n;
      auto b = new_piece_block(n);
      const size_t w_len = W.size();
      // symbol 256 + w means byte w of a FULL 32 KiB window; with less decoded so far (w_len < 32768) the window's first
      // 32768 − w_len places do not exist: a reference to one is no valid stream (the ordinary decoder says "distance
      // too far back"): such a chunk cannot have been reached by a valid stream, and a verified one never has any
      const size_t shift = 32768 - w_len;
      auto byte_of = [shift, base = b->v.data()](uint16_t s2) -> uint8_t { return s2 < 256 ? (uint8_t)s2 : base[(size_t)(s2 - 256) - shift]; };
      (void)byte_of;
      Piece pc;
      pc.buf = b;
      pc.w_len = w_len;
      pc.n = n;
      pc.pending = std::make_shared<Pending>();
      // the next window first (the last 32 KiB of this chunk's bytes), by this thread
      {
        const size_t t0 = n > 32768 ? n - 32768 : 0;
        std::vector<uint8_t> tail(n - t0);
        const uint16_t *sy = r.sym.data();
        const uint8_t *wb = b->v.data();
        bool bad = false;
        for (size_t j = t0; j < n; ++j) {
          const uint16_t s2 = sy[j];
          if (s2 < 256) tail[j - t0] = (uint8_t)s2;
          else if ((size_t)(s2 - 256) >= shift) tail[j - t0] = wb[(size_t)(s2 - 256) - shift];
          else bad = true, tail[j - t0] = 0;
        }
        (void)bad;
        roll_window(tail.data(), tail.size());
      }
      // all of it, in slices, by the workers
      const size_t slice = 1u << 20;
      const size_t n_slices = (n + slice - 1) / slice;
      pc.pending->set((int)n_slices);
      auto keep = std::make_shared<SpecResult>(std::move(r));
      for (size_t sl = 0; sl < n_slices; ++sl) {
        const size_t j0 = sl * slice, j1 = std::min(n, j0 + slice);
        auto pend = pc.pending;
        pool.post(true, [keep, b, w_len, shift, j0, j1, pend] {
          const uint16_t *sy = keep->sym.data();
          uint8_t *base = b->v.data();
          uint8_t *dst = base + w_len;
          for (size_t j = j0; j < j1; ++j) {
            const uint16_t s2 = sy[j];
            dst[j] = s2 < 256 ? (uint8_t)s2 : ((size_t)(s2 - 256) >= shift ? base[(size_t)(s2 - 256) - shift] : (uint8_t)0);
          }
          pend->done();
        });
      }
      out_q.emplace_back(std::move(pc));
    };
    // (a piece goes out only while HOLD_BACK decoded bytes lie behind it: see reference_has_seen)
    auto drain = [&](size_t keep_n) {
      while (out_q.size() > keep_n && !cancelled) {
        size_t behind = 0;
        for (size_t i = 1; i < out_q.size() && behind < HOLD_BACK; ++i) behind += out_q[i].n;
        if (behind < HOLD_BACK) break;
        emitted_out += out_q.front().n;
        emit_front(false);
      }
    };

    // A chunk whose speculation is passed over gives its symbol buffer back at once (12 MB or more each, touched by
    // the worker): a member where the finder keeps missing — stored or fixed blocks, false block starts — would
    // otherwise hold one per chunk until its end, memory growing with the file.
    auto drop_symbols = [](SpecResult &x) {
      UBuf<uint16_t> none;
      std::swap(x.sym, none);
      x.n = 0;
    };
    // chunk 0, then chunk after chunk
    decode_here((uint64_t)chunk_bytes * 8);
    size_t next = 1;
    while (status == INF_OUTPUT_FULL && !cancelled) {
      if (is_stopped()) {
        cancelled = true;
        break;
      }
      drain(6);
      if (next >= n_chunks) {  // behind the last cut: the rest of the stream, the ordinary way
        decode_here(~0ull);
        continue;
      }
      issue_up_to(next + ahead);
      SpecResult &r = spec[next];
      {
        const double t0 = tnow();
        std::unique_lock<std::mutex> lk(rm);
        rcv.wait(lk, [&] { return r.ready; });
        t_wait_spec += tnow() - t0;
        t_workers += r.seconds;
        t_find += r.find_seconds;
      }
      if (r.found && r.start_bit == verified) {
        // (a symbol that points in front of everything decoded so far cannot come from a valid stream entered at a
        // verified position — checked all the same, over the whole chunk, before anything of it is used)
        bool sane = true;
        if (W.size() < 32768) {
          const size_t shift = 32768 - W.size();
          for (size_t j = 0; j < r.n && sane; ++j) sane = r.sym[j] < 256 || (size_t)(r.sym[j] - 256) >= shift;
        }
        if (sane) {
          const uint64_t e2 = r.end_bit;
          const InflateStatus st2 = r.status;
          const uint8_t *after2 = r.after;
          ++n_spec_used;
          bytes_spec += r.n;
          resolve(r);
          verified = e2;
          if (st2 != INF_OUTPUT_FULL) {
            status = st2;
            after = after2;
          }
          ++next;
          continue;
        }
        decode_here((uint64_t)(next + 1) * chunk_bytes * 8);  // (the ordinary decoder says what is wrong with it)
        drop_symbols(r);
        ++next;
        continue;
      }
      if (r.found && r.start_bit > verified) {  // something the finder does not look for lies in between
        decode_here(r.start_bit);
        if (verified != r.start_bit) {  // (it was no block boundary after all: that chunk's result is void)
          drop_symbols(r);
          ++next;
        }
        continue;
      }
      // no entry point found in this chunk, or one in front of where the stream really stands: the ordinary way
      if ((uint64_t)(next + 1) * chunk_bytes * 8 > verified) decode_here((uint64_t)(next + 1) * chunk_bytes * 8);
      drop_symbols(r);
      ++next;
    }
    if (!cancelled) {
      if (out_q.empty()) {  // (nothing at all was decoded)
        Piece pc;
        pc.buf = std::make_shared<Buf>();
        pc.buf->v.resize(64);
        out_q.emplace_back(std::move(pc));
      }
      final_status = status == INF_OUTPUT_FULL ? INF_TRUNCATED : status;
      if (final_status == INF_CORRUPT) {  // the reference's reader has not seen all of what lies in front of the damage
        uint64_t n_ok = emitted_out;
        for (const Piece &pc : out_q) n_ok += pc.n;
        uint64_t cut = n_ok - reference_has_seen(n_ok, 0, origin, origin + (verified >> 3));
        while (cut && !out_q.empty()) {  // (off the tail: resolved pieces first — their bytes are being written by workers)
          Piece &pc = out_q.back();
          const size_t t = (size_t)std::min<uint64_t>(cut, pc.n);
          pc.n -= t;
          cut -= t;
          if (pc.n == 0 && out_q.size() > 1) {
            if (pc.pending) pc.pending->wait();
            out_q.pop_back();
          } else if (pc.n == 0) {
            break;
          }
        }
      }
      while (out_q.size() > 1) emit_front(false);
      if (final_status == INF_STREAM_END && after) {  // where the trailer lies (GzMember::finish)
        gz.inf.in = after;
        gz.inf.bitcnt = 0;
      }
      emit_front(true);
    }
    if (getenv("SHK_FASTQ_DEBUG"))
      fprintf(stderr, "[gzip member, %u threads, %zu chunks of %zu KiB] %zu chunks (%.1f MB) from the workers, %zu stretches (%.1f MB) decoded in order\n",
              n_threads, n_chunks, chunk_bytes >> 10, n_spec_used, bytes_spec / 1e6, n_here, bytes_here / 1e6),
          fprintf(stderr, "   workers %.0f ms in all (%.0f ms of it looking for entry points); this thread waited %.0f ms for chunks, %.0f ms for their bytes, %.0f ms handing windows on\n",
                  t_workers * 1e3, t_find * 1e3, t_wait_spec * 1e3, t_wait_resolve * 1e3, t_emit * 1e3);
    // (the workers are joined by ~Workers: what is still queued of the speculation is dropped)
  }

  void run() {
    if (gz.header_error.kind != IO_NONE) {  // nothing can be read: an empty last window, the error follows it
      ticket_leave();
      emit(Window{nullptr, "", 0, true});
      std::lock_guard<std::mutex> lk(m);
      done = true;
      cv.notify_all();
      return;
    }
    if (all_members) {
      ticket_leave();
      run_members();
      std::lock_guard<std::mutex> lk(m);
      done = true;
      cv.notify_all();
      return;
    }
    {
      // several threads on the one member when it is large enough to pay (test hooks: SHK_PGZ_*)
      const char *e_min = getenv("SHK_PGZ_MIN_KB"), *e_chunk = getenv("SHK_PGZ_CHUNK_KB"), *e_thr = getenv("SHK_PGZ_THREADS");
      const size_t min_bytes = e_min ? (size_t)atoll(e_min) << 10 : (size_t)8 << 20;
      const size_t chunk = std::max<size_t>(e_chunk ? (size_t)atoll(e_chunk) << 10 : (size_t)1 << 20, 512);
      // (three quarters of the CPUs: the window parse and the copy-out run beside the decoder — on the 16-CPU box one .gz
      // file gives 2.5-2.6 Gbases/s with 8 threads, 2.9-3.3 with 12, 2.2-3.0 with 16, 2.1 with 24)
      const uint32_t thr = e_thr ? (uint32_t)atoi(e_thr) : std::max(2u, std::min(32u, usable_cpus() * 3 / 4));
      if (thr > 1 && (size_t)(gz.inf.in_end - gz.inf.in) >= min_bytes) {
        if (gate->enter(ticket, gate_abort)) run_parallel(thr, chunk);
        ticket_leave();
        std::lock_guard<std::mutex> lk(m);
        done = true;
        cv.notify_all();
        return;
      }
    }
    ticket_leave();  // (one thread is all this member gets: no turn needed)
    size_t cap = window + 32768 + Inflater::OUT_SLACK;
    auto b = std::make_shared<Buf>();
    b->v.resize(cap);
    size_t pos = 0, line0 = 0;  // decoded so far in this buffer; start of the bytes not handed out yet
    uint64_t gone = 0;          // bytes of the stream that have left the buffer's front
    const uint8_t *const deflate0 = gz.inf.in;
    for (;;) {
      {
        std::lock_guard<std::mutex> lk(m);
        if (stop) break;
      }
      const InflateStatus st = gz.inf.run(b->v.data(), &pos, b->v.size());
      if (st != INF_OUTPUT_FULL) {
        final_status = st;
        size_t end = pos;
        if (st == INF_CORRUPT) {  // the reference's reader has not seen all of what lies in front of the damage
          const uint64_t seen = reference_has_seen(gone + pos, 0, deflate0, gz.inf.in - (gz.inf.bitcnt >> 3));
          end = (size_t)(seen - gone);  // (≥ line0: the last HOLD_BACK bytes were never handed on)
        }
        emit(Window{b, (const char *)b->v.data() + line0, end - line0, true});
        break;
      }
      const size_t held = std::min(pos - line0, HOLD_BACK);
      const void *nl = memrchr(b->v.data() + line0, '\n', pos - held - line0);
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
      gone += pos - keep;
      pos = keep;
      b = std::move(nb);
    }
    std::lock_guard<std::mutex> lk(m);
    done = true;
    cv.notify_all();
  }
  // every member, one after the other, by this thread (a bgzip file's 64 KB members, `cat a.gz b.gz`)
  void run_members() {
    const size_t cap = window + 32768 + Inflater::OUT_SLACK;
    auto b = std::make_shared<Buf>();
    b->v.resize(cap);
    size_t pos = 0, line0 = 0, crc_from = 0;  // decoded so far in this buffer; first byte not handed out; first byte not in the CRC yet
    uint32_t crc = 0;
    uint64_t member_out = 0;
    uint64_t gone = 0;                        // bytes of the stream that have left the buffer's front
    const uint8_t *deflate0 = gz.inf.in;      // where the member being decoded begins
    for (;;) {
      {
        std::lock_guard<std::mutex> lk(m);
        if (stop) break;
      }
      gz.inf.history_bytes = member_out;  // (a member's matches reach back into that member only)
      const InflateStatus st = gz.inf.run(b->v.data(), &pos, b->v.size());
      crc = crc32_update(crc, b->v.data() + crc_from, pos - crc_from);
      member_out += pos - crc_from;
      crc_from = pos;
      if (st == INF_STREAM_END) {
        members_err = gz.finish(st, crc, member_out);
        const uint8_t *nxt = gz.inf.input_after_stream() + 8, *fe = gz.file_end;
        if (members_err.kind == IO_NONE && nxt < fe) {  // another member: its header, its own history and checks
          gz = GzMember();
          gz.open(nxt, fe);
          members_err = gz.header_error;
          crc = 0;
          member_out = 0;
          deflate0 = gz.inf.in;
          if (members_err.kind == IO_NONE) continue;
        }
        emit(Window{b, (const char *)b->v.data() + line0, pos - line0, true});
        break;
      }
      if (st != INF_OUTPUT_FULL) {  // a short or corrupt body
        members_err = gz.finish(st, crc, member_out);
        size_t end = pos;
        if (st == INF_CORRUPT) {  // (see reference_has_seen: a member's end is where a read ends, too)
          const uint64_t seen = reference_has_seen(gone + pos, gone + pos - member_out, deflate0, gz.inf.in - (gz.inf.bitcnt >> 3));
          end = (size_t)(seen - gone);
        }
        emit(Window{b, (const char *)b->v.data() + line0, end - line0, true});
        break;
      }
      const size_t held = std::min(pos - line0, HOLD_BACK);
      const void *nl = memrchr(b->v.data() + line0, '\n', pos - held - line0);
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
      gone += pos - keep;
      pos = keep;
      crc_from = keep;
      b = std::move(nb);
    }
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
  IoError finish(uint32_t crc, uint64_t total) override { return all_members ? members_err : gz.finish(final_status, crc, total); }
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
  const char *src_end = nullptr;   // one past the window's last byte (a 32-byte load must not start within 31 bytes of it)
  UBuf<uint32_t> nl;  // entries: position | NL_CR
  size_t line0 = 0;
  bool last_unterminated = false;  // line nl.size()-1 has no '\n' (its '\r', if any, stays: BufRead::lines)
  bool has_lead = false;
  std::string lead_seq;
  UBuf<uint32_t> lens;  // all records, the lead first
  std::vector<Flaw> flaws;     // ascending by rec
  uint64_t first_rec = 0;      // local index (within the file) of the chunk's first record
  size_t bytes() const { return lens.size() * 4 + nl.size() * 4 + lead_seq.size() + (hold ? hold->v.size() : 0); }
  void seq_line(size_t j, const char **p, size_t *len) const {  // j: index among the whole records
    const size_t l = line0 + 4 * j + 1;
    const size_t s0 = (size_t)(nl[l - 1] & NL_POS) + 1, s1 = nl[l] & NL_POS;
    *p = src + s0;
    *len = s1 - s0 - (nl[l] >> 31);  // (an unterminated last line's entry never carries NL_CR: its '\r', if any, stays)
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

// Unmapping a file that was read through a mapping walks its whole page table with the address space locked: 8-9 ms for
// 2.5 GB — a tenth of the job that read it, and everybody else's page faults wait meanwhile.  Mappings are therefore
// handed to ONE background thread that takes them down in 32 MB pieces (the lock is held a tenth of a millisecond at a
// time) while the caller goes on; the thread is joined, its queue drained, when the library is unloaded or the process
// ends.
struct Unmapper {
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::pair<char *, size_t>> q;
  bool stop = false;
  std::thread th;
  void run() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [&] { return stop || !q.empty(); });
      if (q.empty()) return;  // (stop, and nothing left)
      std::pair<char *, size_t> r = q.front();
      q.pop_front();
      lk.unlock();
      constexpr size_t PIECE = 32u << 20;
      while (r.second) {
        const size_t n = std::min(r.second, PIECE);
        (void)munmap(r.first, n);
        r.first += n;
        r.second -= n;
      }
      lk.lock();
    }
  }
  void give(const char *p, size_t n) {
    std::lock_guard<std::mutex> lk(m);
    if (!th.joinable()) th = std::thread([this] { run(); });
    q.emplace_back(const_cast<char *>(p), n);
    cv.notify_one();
  }
  ~Unmapper() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
      cv.notify_one();
    }
    if (th.joinable()) th.join();
  }
};
static Unmapper &unmapper() {
  static Unmapper u;
  return u;
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
  bool gz_all_members = false;  // (shk_fastq_open_ex: not the reference's behaviour)
  PgzGate *gate = nullptr;      // the reader's gate (shk_fastq::gate)
  uint64_t pgz_ticket = 0;      // this file's turn at the many-thread gzip decoder (drawn in file order by the reader)
  bool ticket_given = false;    // … handed to a GzSource; otherwise given back as soon as the source is known
  const char *map = nullptr;  // plain files: the mapping the chunks point into (released with the producer)
  size_t map_size = 0;
  std::unique_ptr<Source> src;
  Pool *pool = nullptr;
  uint32_t T = 1;
  ~Producer() {
    src.reset();
    if (map) unmapper().give(map, map_size);  // (page-aligned: a whole mapping; nobody points into it any more)
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
        if (gz_ext || magic) {
          auto *g = new GzSource((const uint8_t *)data, (const uint8_t *)data + size, nullptr, gate, pgz_ticket);
          ticket_given = true;
          g->all_members = gz_all_members;
          return with_window(g);
        }
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
    auto *g = new GzSource(all->data(), all->data() + all->size(), all, gate, pgz_ticket);
    ticket_given = true;
    g->all_members = gz_all_members;
    return with_window(g);
  }

  // ---- the parse ---------------------------------------------------------------------------------------------------
  // what each pool thread found in its share of the window (kept across windows: no allocation after the first)
  std::vector<std::vector<uint32_t>> tl_nl;
  std::vector<std::vector<uint8_t>> tl_fb;
  UBuf<uint8_t> FB;  // first byte of the line behind every '\n' of the window
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
    // 1. the line table: every share of the window is scanned ONCE (scan_share) into lists of the scanning thread's own
    // and, their places among all the lines known from a prefix sum, the lists are copied into the window's tables (5 B
    // per line, ≈ 6 % of the window's bytes).  The same pass notes whether a share has any byte ≥ 0x80 (only then is
    // anything checked for UTF-8) and, for a gzip member, sums the share's CRC-32.
    const uint32_t TT = n >= (1u << 16) ? T : 1;
    std::vector<size_t> first(TT + 1, 0);
    std::vector<uint8_t> high(TT, 0);
    std::vector<uint32_t> crcs(TT, 0);
    if (tl_nl.size() < TT) tl_nl.resize(TT), tl_fb.resize(TT);
    const bool want_crc = src->wants_crc();
    auto scan = [&](uint32_t t) {
      const size_t a = n * t / TT, b = n * (t + 1) / TT;
      tl_nl[t].clear();
      tl_fb[t].clear();
      high[t] = scan_share(base, n, a, b, tl_nl[t], tl_fb[t]);
      first[t + 1] = tl_nl[t].size();
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
    UBuf<uint32_t> NL;
    NL.resize_discard(M + 1);
    FB.resize_discard(M + 1);
    auto gather = [&](uint32_t t) {
      if (!tl_nl[t].empty()) {
        memcpy(NL.data() + first[t], tl_nl[t].data(), tl_nl[t].size() * 4);
        memcpy(FB.data() + first[t], tl_fb[t].data(), tl_fb[t].size());
      }
    };
    if (TT == 1) gather(0);
    else pool->parallel_for(TT, gather);
    bool last_unterminated = false;
    if (w.last) {
      end_err = src->finish(crc, total_bytes);
      // a last line without '\n' (it has ≥ 1 byte) is a line — unless reading on fails: then what had been read of it
      // goes with the error (read_line returns the Err)
      if (end_err.kind == IO_NONE && (M == 0 ? n > 0 : (size_t)(NL[M - 1] & NL_POS) + 1 < n)) {
        NL[M] = (uint32_t)n;  // (no NL_CR: the '\r' of an unterminated line stays, BufRead::lines)
        FB[M] = 0;
        ++M;
        last_unterminated = true;
      }
    }
    // line ℓ = [ℓ ? NL[ℓ-1]+1 : 0, NL[ℓ]), a '\r' in front of the '\n' stripped
    auto line_start = [&](size_t l) -> size_t { return l ? (size_t)(NL[l - 1] & NL_POS) + 1 : 0; };
    auto line_len = [&](size_t l) -> size_t { return (size_t)(NL[l] & NL_POS) - line_start(l) - (NL[l] >> 31); };
    auto line_first = [&](size_t l) -> uint8_t { return l ? FB[l - 1] : (uint8_t)base[0]; };  // (only looked at when the line is not empty)
    auto line_at = [&](size_t l, const char **p, size_t *len) {
      *p = base + line_start(l);
      *len = line_len(l);
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
    c.lens.resize_discard(n_lead + R);
    if (c.has_lead) c.lens[0] = (uint32_t)c.lead_seq.size();
    const double t_c = now();
    // 3. sequence lengths + flaws, records shared out evenly: arithmetic on the tables alone (the file's bytes are only
    // looked at again for the text of a flaw, or when the window has bytes ≥ 0x80)
    const uint32_t TR = R >= 4096 ? T : 1;
    std::vector<std::vector<Flaw>> fl(TR);
    auto per_record = [&](uint32_t t) {
      const size_t r0 = R * t / TR, r1 = R * (t + 1) / TR;
      for (size_t r = r0; r < r1; ++r) {
        const size_t l = l0 + 4 * r;
        const size_t hl = line_len(l), sl = line_len(l + 1), spl = line_len(l + 2), qll = line_len(l + 3);
        c.lens[n_lead + r] = (uint32_t)sl;
        if (any_high) {  // BufRead::lines: a line that is not UTF-8 is an error of the read, whatever the cadence
          const size_t ll[4] = {hl, sl, spl, qll};
          int bad = -1;
          for (int i = 0; i < 4 && bad < 0; ++i) {
            const uint8_t *lp = (const uint8_t *)base + line_start(l + i);
            if (any_high_bit(lp, ll[i]) && !valid_utf8(lp, ll[i])) bad = i;
          }
          if (bad >= 0) {
            fl[t].emplace_back(Flaw{rec + n_lead + r, FLAW_UTF8, (uint8_t)bad, std::string(), 0, 0});
            continue;
          }
        }
        // validate_fastq_record's checks (io.rs:169-196) off the tables; the lines themselves only for a flaw's text
        const bool h_ok = hl && line_first(l) == '@', sp_ok = spl && line_first(l + 2) == '+';
        if (!h_ok || !sp_ok || qll != sl) {
          Flaw f;
          if (find_flaw(base + line_start(l), hl, sl, base + line_start(l + 2), spl, qll, rec + n_lead + r, &f)) fl[t].emplace_back(std::move(f));
        }
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
    c.src_end = base + n;
    c.line0 = l0;
    c.last_unterminated = last_unterminated && l0 + 4 * R == M;  // (else the unterminated line is not among the records')
    NL.n = l0 + 4 * R;
    c.nl = std::move(NL);
    rec += n_lead + R;
    const double t_d = now();
    if (n_lead + R) push(std::move(c));
    if (dbg)
      fprintf(stderr, "[fastq window %zu MB] scan %.1f ms  gather %.1f ms  lens %.1f ms  push(wait) %.1f ms\n", n >> 20, (t_b - t_a) * 1e3,
              (t_c - t_b) * 1e3, (t_d - t_c) * 1e3, (now() - t_d) * 1e3);
  }

  void run() {
    const bool opened = open_source();
    if (!ticket_given) gate->leave(pgz_ticket);  // (not a gzip file, or it could not be opened)
    if (!opened) return finish(end);
    Window w;
    bool saw_last = false;
    while (!cancelled() && src->next(&w)) {
      if (w.n > NL_POS)  // (a single line of 2 GiB: the line tables hold 31-bit positions)
        return finish(FileEnd{END_STREAM, (int)carry.size(), IoError{IO_OTHER, "line longer than 2 GiB"}});
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

// ---- 2-bit packing straight out of the parsed windows ---------------------------------------------------------
// The batch's concatenated bases as ONE sequence in the layout of the reference's Read::from_str (encoding.rs:60-95:
// four bases per byte, the first in the top two bits; A 00, C 01, G 10, T 11) + an N mask (bit p % 32 of word p / 32;
// an N is 00 in the stream) — what shk_pack_reads makes of an ASCII batch, made here while the sequences are copied
// out of the mapped file, so that a batch crosses PCIe at 0.3 B per base and is never written as ASCII at all.
namespace {

// up to 32 bases → their codes as a big-endian 64-bit word (first base on top, missing bases 0), the N bits and the
// bits of bytes outside ACGTN (bit i = base i)
struct Conv32 {
  uint64_t be;
  uint32_t nbits, bad;
};
static inline Conv32 conv32_scalar(const uint8_t *p, uint32_t n) {
  static const struct Lut {
    uint8_t t[256];
    Lut() {
      memset(t, 0xFF, sizeof t);
      t[(unsigned)'A'] = 0, t[(unsigned)'C'] = 1, t[(unsigned)'G'] = 2, t[(unsigned)'T'] = 3, t[(unsigned)'N'] = 4;
    }
  } lut;
  Conv32 r{0, 0, 0};
  for (uint32_t i = 0; i < n; ++i) {
    const uint8_t c = lut.t[p[i]];
    if (c == 0xFF) r.bad |= 1u << i;
    else if (c == 4) r.nbits |= 1u << i;
    else r.be |= (uint64_t)c << (62 - 2 * i);
  }
  return r;
}
__attribute__((target("avx2,bmi,bmi2"))) static inline Conv32 conv32_avx2(const uint8_t *p, uint32_t n) {
  const __m256i v = _mm256_loadu_si256((const __m256i *)p);
  const __m256i isn = _mm256_cmpeq_epi8(v, _mm256_set1_epi8('N'));
  const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('A')), _mm256_cmpeq_epi8(v, _mm256_set1_epi8('C'))),
                                     _mm256_or_si256(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('G')), _mm256_or_si256(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('T')), isn)));
  const uint32_t live = n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u;
  Conv32 r;
  r.nbits = (uint32_t)_mm256_movemask_epi8(isn) & live;
  r.bad = ~(uint32_t)_mm256_movemask_epi8(ok) & live;
  // A 41 C 43 G 47 T 54: bits 2..1 are 00 01 11 10 — the code is t ^ (t >> 1); an N (and anything else) becomes 00
  __m256i t = _mm256_and_si256(_mm256_srli_epi16(v, 1), _mm256_set1_epi8(3));
  t = _mm256_xor_si256(t, _mm256_and_si256(_mm256_srli_epi16(t, 1), _mm256_set1_epi8(1)));
  t = _mm256_andnot_si256(isn, t);
  // four codes → one byte, first on top: (c0·64 + c1·16) + (c2·4 + c3)
  const __m256i w16 = _mm256_maddubs_epi16(t, _mm256_set1_epi32(0x01041040));
  const __m256i w32 = _mm256_madd_epi16(w16, _mm256_set1_epi16(1));
  const __m256i b = _mm256_shuffle_epi8(w32, _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1,
                                                               -1, -1, -1, -1, -1, -1, -1, -1, -1));
  const uint64_t le = (uint32_t)_mm256_extract_epi32(b, 0) | ((uint64_t)(uint32_t)_mm256_extract_epi32(b, 4) << 32);  // bytes in stream order
  uint64_t be = __builtin_bswap64(le);
  if (n < 32) be &= ~0ull << (64 - 2 * n);  // (bytes past the sequence's end were converted too)
  r.be = be;
  return r;
}

// Appends bases to a packed stream at any base position: the words are big-endian 64-bit groups of 32 bases.
struct PackWriter {
  uint8_t *packed;
  uint32_t *nmask;
  uint64_t word;   // index of the 32-base word being filled
  uint32_t fill;   // bases it holds
  uint64_t acc;    // … on top
  uint32_t nacc;   // their N bits (low `fill` bits)
  uint64_t first_bad = ~0ull;  // (position << 8) | byte of the first byte outside ACGTN
  bool avx2;
  void open(uint8_t *pk, uint32_t *nm, uint64_t pos, bool merge_first) {
    packed = pk, nmask = nm;
    word = pos >> 5, fill = (uint32_t)(pos & 31);
    acc = 0, nacc = 0;
    if (fill && merge_first) {  // the word's first bases are there already (the chunk before wrote them, zeros behind them)
      uint64_t be;
      memcpy(&be, packed + word * 8, 8);
      acc = __builtin_bswap64(be);
      nacc = nmask[word];
    }
    avx2 = have_avx2();
  }
  inline void put(const Conv32 &c, uint32_t n, uint64_t pos, const uint8_t *src) {
    if (c.bad && first_bad == ~0ull) {
      const uint32_t i = (uint32_t)__builtin_ctz(c.bad);
      first_bad = ((pos + i) << 8) | src[i];
    }
    acc |= fill ? c.be >> (2 * fill) : c.be;
    const uint64_t nb = (uint64_t)c.nbits << fill;
    nacc |= (uint32_t)nb;
    if (fill + n >= 32) {
      const uint64_t be = __builtin_bswap64(acc);
      memcpy(packed + word * 8, &be, 8);
      nmask[word] = nacc;
      ++word;
      acc = fill ? c.be << (64 - 2 * fill) : 0;
      nacc = (uint32_t)(nb >> 32);
      fill = fill + n - 32;
    } else {
      fill += n;
    }
  }
  // `len` bases at `p` (absolute position `pos`); lim: one past the last byte that may be read
  void append(const uint8_t *p, size_t len, uint64_t pos, const uint8_t *lim) {
    size_t i = 0;
    if (avx2) {
      for (; i + 32 <= len; i += 32) put(conv32_avx2(p + i, 32), 32, pos + i, p + i);
      if (i < len) {
        const uint32_t n = (uint32_t)(len - i);
        if (p + i + 32 <= lim) {
          put(conv32_avx2(p + i, n), n, pos + i, p + i);
        } else {
          uint8_t tmp[32] = {0};
          memcpy(tmp, p + i, n);
          put(conv32_avx2(tmp, n), n, pos + i, p + i);
        }
      }
    } else {
      for (; i < len; i += 32) {
        const uint32_t n = (uint32_t)std::min<size_t>(32, len - i);
        put(conv32_scalar(p + i, n), n, pos + i, p + i);
      }
    }
  }
  void close() {  // a last, partial word: its unused bits are zero (the next chunk may go on in it)
    if (fill) {
      const uint64_t be = __builtin_bswap64(acc);
      memcpy(packed + word * 8, &be, 8);
      nmask[word] = nacc;
    }
  }
};

}  // namespace

struct shk_fastq {
  std::vector<std::string> paths;
  PgzGate gate;  // this reader's files take the many-thread gzip decoder in turn (declared before `prod`: the sources hand their tickets back when they go)
  std::vector<std::unique_ptr<Producer>> prod;  // one per path; started up to LOOKAHEAD files ahead of the scout
  size_t started = 0;                           // producers started so far
  std::unique_ptr<Pool> pool, cpool;            // the producers' pool (window parse) and the consumer's (copy-out): they overlap
  uint32_t T = 1;
  uint64_t max_reads = 0, validate_every = 0;
  bool gz_all_members = false;
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
      // Two pools: the producers' (window parse: ≈1.5 ms per 128 MiB window) and the consumer's (prefix sums, copy-out or
      // packing: the larger share of the work since the parse became lean).  Each gets as many threads as this process may
      // really keep busy — a container's CFS quota counts, not the host's core count; they overlap only while a
      // producer is ahead of the consumer.
      const uint32_t usable = usable_cpus();
      const char *ev = getenv("SHK_FASTQ_THREADS");
      const char *evc = getenv("SHK_FASTQ_COPY_THREADS");
      T = ev && atoi(ev) > 0 ? (uint32_t)atoi(ev) : std::max(2u, std::min(32u, usable));
      pool.reset(new Pool(T));
      cpool.reset(new Pool(evc && atoi(evc) > 0 ? (uint32_t)atoi(evc) : std::max(2u, std::min(32u, usable + usable / 2))));
    }
    while (started < paths.size() && started < scout_file + LOOKAHEAD) {
      auto p = std::make_unique<Producer>();
      p->path = paths[started];
      p->name = p->path == "-" ? "stdin" : p->path;
      p->pool = pool.get();
      p->T = T;
      p->gz_all_members = gz_all_members;
      p->gate = &gate;
      p->pgz_ticket = gate.draw();
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
  return shk_fastq_open_ex(paths, n_paths, max_reads, validate_every, 0, out);
}

int shk_fastq_open_ex(const char *const *paths, uint32_t n_paths, uint64_t max_reads, uint64_t validate_every, uint32_t flags,
                      shk_fastq **out) {
  if (!out) return SHK_ERR_BAD_ARG;
  auto *r = new shk_fastq();
  r->gz_all_members = (flags & SHK_FASTQ_GZIP_ALL_MEMBERS) != 0;
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

// Fill (bases, offsets) — or (packed, nmask, offsets) — with up to max_seqs sequences / bases_cap bases in input
// order.  offsets[0] = 0.  *n_seqs = 0 with SHK_OK means end of input.  Sequences are never split; a sequence longer
// than bases_cap is an error.
static int next_batch_impl(shk_fastq *r, uint8_t *bases, uint8_t *packed, uint32_t *nmask, uint64_t bases_cap, uint64_t *offsets,
                           uint64_t max_seqs, uint64_t *n_seqs) {
  *n_seqs = 0;
  offsets[0] = 0;
  if (r->err_code) return r->err_code;
  uint64_t used = 0, n = 0;
  bool stop = false;
  uint64_t first_bad = ~0ull;
  const bool dbg = getenv("SHK_FASTQ_DEBUG") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_wait = 0, t_sum = 0, t_move = 0;
  while (!stop && n < max_seqs) {
    if (r->n_reads_read == r->limit()) {
      if (r->state == shk_fastq::RUNNING) {
        const double t0 = now();
        r->scout_more();
        t_wait += now() - t0;
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
    // (a sequence that no batch of this size can hold: reported when it is the FIRST of a batch, with nothing
    // consumed — the caller may come back with a larger buffer; shk_run_files does)
    if (take && n == 0 && c.lens[seq_begin] > bases_cap) {
      r->err = "sequence longer than the batch buffer";  // (not sticky: the same call with a larger buffer goes on)
      return SHK_ERR_BAD_ARG;
    }
    // how many of them fit, and their offsets: a prefix sum over the lengths — by the copy pool when there are many
    // (a million additions by the caller's thread alone were a third of a batch's time)
    const double t_s0 = now();
    {
      const uint32_t TS = take >= 65536 ? r->cpool->size() : 1;
      std::vector<uint64_t> part(TS + 1, 0);
      auto sum = [&](uint32_t t) {
        const size_t a = take * t / TS, b = take * (t + 1) / TS;
        uint64_t sacc = 0;
        for (size_t j = a; j < b; ++j) sacc += c.lens[seq_begin + j];
        part[t + 1] = sacc;
      };
      if (TS == 1) sum(0);
      else r->cpool->parallel_for(TS, sum);
      for (uint32_t t = 0; t < TS; ++t) part[t + 1] += part[t];
      if (used + part[TS] > bases_cap) {  // not all of them: the first that does not fit is delivered first thing next call
        uint32_t t = 0;
        while (used + part[t + 1] <= bases_cap) ++t;
        size_t j = take * t / TS;
        uint64_t u = used + part[t];
        while (u + c.lens[seq_begin + j] <= bases_cap) u += c.lens[seq_begin + j++];
        take = j;
        stop = true;
      }
      if (!stop && TS > 1) {
        auto wr = [&](uint32_t t) {
          const size_t a = take * t / TS, b = take * (t + 1) / TS;
          uint64_t u = used + part[t];
          for (size_t j = a; j < b; ++j) {
            u += c.lens[seq_begin + j];
            offsets[n + 1 + j] = u;
          }
        };
        r->cpool->parallel_for(TS, wr);
      } else {
        uint64_t u = used;
        for (size_t j = 0; j < take; ++j) {
          u += c.lens[seq_begin + j];
          offsets[n + 1 + j] = u;
        }
      }
      if (take) used = offsets[n + take];
      n += take;
    }
    if (!take) continue;
    const double t_m0 = now();
    t_sum += t_m0 - t_s0;
    const size_t lead = c.has_lead ? 1 : 0;
    // record j of this hand-out (j in [0, take)): where its sequence lies
    auto seq_of = [&](size_t j, const char **p, size_t *len, const char **lim) {
      if (lead && seq_begin + j == 0) {
        *p = c.lead_seq.data(), *len = c.lead_seq.size(), *lim = c.lead_seq.data() + c.lead_seq.size();
      } else {
        c.seq_line(seq_begin + j - lead, p, len);
        *lim = c.src_end;
      }
    };
    const uint32_t TT = take >= 4096 ? r->cpool->size() : 1;
    if (!packed) {  // ASCII: record by record
      auto copy = [&](uint32_t t) {
        const size_t a = take * t / TT, b = take * (t + 1) / TT;
        for (size_t j = a; j < b; ++j) {
          const char *sq, *lim;
          size_t sl;
          seq_of(j, &sq, &sl, &lim);
          memcpy(bases + offsets[n_begin + j], sq, sl);
        }
      };
      if (TT == 1) copy(0);
      else r->cpool->parallel_for(TT, copy);
    } else {
      // packed: the hand-out's base range, cut at multiples of 32 bases so that no two threads share a word; the first
      // cut goes on in the word the hand-out before left unfinished
      const uint64_t B0 = offsets[n_begin], B1 = used;
      std::vector<uint64_t> bad(TT, ~0ull);
      auto pack = [&](uint32_t t) {
        uint64_t P0 = t == 0 ? B0 : (B0 + (B1 - B0) * t / TT) & ~31ull, P1 = t + 1 == TT ? B1 : (B0 + (B1 - B0) * (t + 1) / TT) & ~31ull;
        if (t > 0 && P0 < B0) P0 = B0;  // (tiny ranges: a cut may round below the start)
        if (P1 < P0) P1 = P0;
        if (P0 >= P1) return;
        // the record that holds position P0
        size_t lo = 0, hi = take;
        while (lo + 1 < hi) {
          const size_t mid = (lo + hi) / 2;
          if (offsets[n_begin + mid] <= P0) lo = mid;
          else hi = mid;
        }
        PackWriter w;
        w.open(packed, nmask, P0, /*merge_first=*/true);  // (only the part that starts at the hand-out's own, unaligned start finds a word begun: every other cut is a multiple of 32)
        for (size_t j = lo; j < take && offsets[n_begin + j] < P1; ++j) {
          const char *sq, *lim;
          size_t sl;
          seq_of(j, &sq, &sl, &lim);
          const uint64_t r0 = offsets[n_begin + j];
          const uint64_t a = std::max(P0, r0), b = std::min<uint64_t>(P1, r0 + sl);
          if (a < b) w.append((const uint8_t *)sq + (a - r0), (size_t)(b - a), a, (const uint8_t *)lim);
        }
        w.close();
        bad[t] = w.first_bad;
      };
      if (TT == 1) pack(0);
      else r->cpool->parallel_for(TT, pack);
      for (uint64_t v : bad) first_bad = std::min(first_bad, v);
    }
    t_move += now() - t_m0;
    h.next += take;
    r->n_reads_read += take;                    // io.rs:337
    r->n_bases_read += used - offsets[n_begin];  // io.rs:335 (N included)
    if (first_bad != ~0ull) break;
  }
  if (first_bad != ~0ull)  // identical text to encoding.rs:353-356 — these reads are ones the reference would have drained
    return r->fail(SHK_ERR_INVALID_CHAR, fmt("Invalid character '%s' in sequence. Only ACGTN allowed.", byte_as_char((uint8_t)(first_bad & 0xFF)).c_str()));
  if (r->n_reads_read == r->limit() && r->state != shk_fastq::RUNNING) {
    if (r->state == shk_fastq::FAILED) {
      if (n == 0) return r->fail(r->pending_code, r->pending_err);
    } else {
      r->reached_max = r->state == shk_fastq::AT_MAX;
      r->done = true;
      r->held.clear();
    }
  }
  if (dbg)
    fprintf(stderr, "[fastq batch %llu reads, %llu bases] waited for the parse %.1f ms  offsets %.1f ms  %s %.1f ms\n", (unsigned long long)n,
            (unsigned long long)used, t_wait * 1e3, t_sum * 1e3, packed ? "pack" : "copy", t_move * 1e3);
  *n_seqs = n;
  return SHK_OK;
}

int shk_fastq_next_batch(shk_fastq *r, uint8_t *bases, uint64_t bases_cap, uint64_t *offsets,
                         uint64_t max_seqs, uint64_t *n_seqs) {
  if (!r || !bases || !offsets || !n_seqs) return SHK_ERR_BAD_ARG;
  return next_batch_impl(r, bases, nullptr, nullptr, bases_cap, offsets, max_seqs, n_seqs);
}

int shk_fastq_next_batch_packed(shk_fastq *r, uint8_t *packed, uint32_t *nmask, uint64_t bases_cap, uint64_t *offsets,
                                uint64_t max_seqs, uint64_t *n_seqs) {
  if (!r || !packed || !nmask || !offsets || !n_seqs) return SHK_ERR_BAD_ARG;
  return next_batch_impl(r, nullptr, packed, nmask, bases_cap, offsets, max_seqs, n_seqs);
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
  const bool avx2 = have_avx2();
  auto work = [&](uint32_t t) {
    uint64_t w0 = n_words * t / T;
    const uint64_t w1 = n_words * (t + 1) / T;
    if (avx2) {  // whole 32-base words by the front-end's converter (32 bases per step); a last, partial word below
      const uint64_t wf = std::min(w1, n_bases / 32);
      for (; w0 < wf; ++w0) {
        const Conv32 c = conv32_avx2(bases + w0 * 32, 32);
        if (c.bad && bad[t] == ~0ull) {
          const uint32_t i = (uint32_t)__builtin_ctz(c.bad);
          bad[t] = ((w0 * 32 + i) << 8) | bases[w0 * 32 + i];
        }
        const uint64_t be = __builtin_bswap64(c.be);
        memcpy(packed + w0 * 8, &be, 8);
        nmask[w0] = c.nbits;
      }
    }
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
