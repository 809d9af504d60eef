// shk_engine.hip — host side of libshk: the C ABI of include/shk.h over the gfx950
// kernels in shk_device.hip.h.  One context = one GPU = one HIP stream.
//
// Reference call sites this replaces (paths under /root/reference):
//   src/io.rs:355-361   drain_batch → Chunk::ingest_seq        → shk_ingest_batch/_reads
//   src/io.rs:1021-1028 extend_with_histogram + get_vector      → shk_finalize/_histograms
//   src/io.rs:545-552   per-chunk totals                        → shk_get_counters
// There is no CPU fallback anywhere in this file.
#include "../../include/shk.h"
#include "shk_device.hip.h"
#include "shk_front.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace shk;

struct shk_ctx;
static int count_tiles(shk_ctx *c, const BatchRef &b, uint64_t sub_kmers_ub, bool prezeroed);
static int prepare_cursors(shk_ctx *c, bool multi, bool defer, uint32_t *n_words);
static int acc_prepare(shk_ctx *c, uint64_t kmers_ub, int64_t lane_one, uint64_t lane_add_exact = 0);
static bool first_launch_defers(shk_ctx *c, uint64_t kmers_ub, bool multi);
static int settle(shk_ctx *c);
static int settle_light(shk_ctx *c);
static int flush_acc(shk_ctx *c);
static int replay_held_spills(shk_ctx *c);
static uint64_t acc_records_est(const shk_ctx *c, uint64_t kmers_ub);
struct XchgOut {
  uint64_t layout_bases;  // in: what every rank sizes its segments for (0: this batch)
  void *d_records, *d_cursors;
  shk_xchg_layout lay;
  uint64_t n_foreign;
  // the wide round (shk_xchg_wide_scatter_device): whole k-mers + lanes grouped by owner, counts on the host
  bool wide = false;
  void *d_kmers = nullptr, *d_lanes = nullptr;
  uint64_t *counts = nullptr;
  bool late_settle = false;  // shk_xchg_scatter_device looks at the previous absorb's outcome AFTER this launch (below)
  bool two_calls = false;    // shk_xchg_scatter_begin: the statistics' copy is queued behind the launch, not waited for
};
static int xchg_scatter_launch(shk_ctx *c, const BatchRef &b, uint64_t kmers_ub, XchgOut *xo);
static int xw_scatter_launch(shk_ctx *c, const BatchRef &b, uint64_t kmers_ub, XchgOut *xo);
static int xchg_prepare_cursors(shk_ctx *c, uint32_t *n_words);
static bool trace_on() {
  static const bool on = getenv("SHK_TRACE") != nullptr;
  return on;
}
#define SHK_TRACEF(...)                                  \
  do {                                                   \
    if (trace_on()) fprintf(stderr, "[shk] " __VA_ARGS__); \
  } while (0)
static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return v ? atoi(v) : dflt;
}

namespace {

thread_local std::string g_create_error;

// ---- a process-wide cache of device and pinned-host blocks -------------------------------------------------
// hipMalloc / hipFree and hipHostMalloc / hipHostFree map and unmap memory: ≈ 45-110 µs per MB of pinned memory, a few
// hundred µs per device block, all of it serialised on the process's address space.  A job of 1.2 Gbases out of a FASTQ
// file spent 17-19 ms of its 80 taking a context and its batch buffers down again, and a few more setting them up.
// Blocks that are given back are therefore kept (per device, a bounded amount) and handed to the next request of about
// their size — the next context of the process starts with warm memory.  Nothing in the engine relies on what a fresh
// block holds.  shk_release_cached_memory() gives everything back; SHK_NO_MEM_CACHE=1 turns the cache off.
struct MemCache {
  struct Blk {
    int dev;  // device (−1: pinned host memory)
    size_t bytes;
    void *p;
  };
  std::mutex m;
  std::vector<Blk> blocks;
  size_t dev_total = 0, host_total = 0;
  std::unordered_map<void *, size_t> pinned_out;  // shk_alloc_pinned blocks in the caller's hands → their size
  // … and idle streams: destroying the stream a context's host copies ran on takes 6-7 ms (its queues go with it)
  std::vector<std::pair<int, hipStream_t>> streams;
};
static MemCache &mem_cache() {
  static MemCache *c = new MemCache;  // (never destroyed: no HIP call may run from a static destructor at exit)
  return *c;
}
constexpr size_t DEV_CACHE_MAX = 4ull << 30, DEV_BLOCK_MAX = 1ull << 30, HOST_CACHE_MAX = 512ull << 20, HOST_BLOCK_MAX = 160ull << 20;
static bool mem_cache_on() {
  static const bool on = getenv("SHK_NO_MEM_CACHE") == nullptr;
  return on;
}
// a cached block of at least `bytes` (and not much more) for device `dev` (−1: pinned host), or nullptr
static void *cache_take(int dev, size_t bytes, size_t *got) {
  if (!mem_cache_on()) return nullptr;
  MemCache &mc = mem_cache();
  std::lock_guard<std::mutex> lk(mc.m);
  size_t best = ~(size_t)0;
  for (size_t i = 0; i < mc.blocks.size(); ++i) {
    const MemCache::Blk &b = mc.blocks[i];
    if (b.dev == dev && b.bytes >= bytes && b.bytes <= bytes + bytes / 2 + 65536 && (best == ~(size_t)0 || b.bytes < mc.blocks[best].bytes)) best = i;
  }
  if (best == ~(size_t)0) return nullptr;
  const MemCache::Blk b = mc.blocks[best];
  mc.blocks.erase(mc.blocks.begin() + (long)best);
  (dev < 0 ? mc.host_total : mc.dev_total) -= b.bytes;
  *got = b.bytes;
  return b.p;
}
static bool cache_give(int dev, void *p, size_t bytes) {
  if (!mem_cache_on()) return false;
  MemCache &mc = mem_cache();
  std::lock_guard<std::mutex> lk(mc.m);
  size_t &total = dev < 0 ? mc.host_total : mc.dev_total;
  if (bytes > (dev < 0 ? HOST_BLOCK_MAX : DEV_BLOCK_MAX) || total + bytes > (dev < 0 ? HOST_CACHE_MAX : DEV_CACHE_MAX) || mc.blocks.size() >= 256) return false;
  mc.blocks.push_back(MemCache::Blk{dev, bytes, p});
  total += bytes;
  return true;
}
static hipError_t stream_take(hipStream_t *st) {  // a non-blocking stream on the current device
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (mem_cache_on()) {
    MemCache &mc = mem_cache();
    std::lock_guard<std::mutex> lk(mc.m);
    for (size_t i = 0; i < mc.streams.size(); ++i)
      if (mc.streams[i].first == dev) {
        *st = mc.streams[i].second;
        mc.streams.erase(mc.streams.begin() + (long)i);
        return hipSuccess;
      }
  }
  return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
static void stream_give(hipStream_t st) {  // (idle: the caller has synchronised it)
  if (!st) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (mem_cache_on()) {
    MemCache &mc = mem_cache();
    std::lock_guard<std::mutex> lk(mc.m);
    if (mc.streams.size() < 16) {
      mc.streams.emplace_back(dev, st);
      return;
    }
  }
  (void)hipStreamDestroy(st);
}
static hipError_t dev_alloc(void **p, size_t bytes, size_t *got) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if ((*p = cache_take(dev, bytes, got))) return hipSuccess;
  *got = bytes;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess && mem_cache_on()) {  // out of memory with blocks lying idle in the cache: give them back and try again
    (void)hipGetLastError();
    shk_release_cached_memory();
    e = hipMalloc(p, bytes);
  }
  return e;
}
static void dev_free(void *p, size_t bytes) {
  if (!p) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  // (hipFree waits for the device; a block that goes to the cache instead must be as idle as one that is freed)
  if (mem_cache_on() && bytes <= DEV_BLOCK_MAX && hipDeviceSynchronize() == hipSuccess && cache_give(dev, p, bytes)) return;
  (void)hipFree(p);
}
static hipError_t host_alloc(void **p, size_t bytes, size_t *got) {
  if ((*p = cache_take(-1, bytes, got))) return hipSuccess;
  *got = bytes;
  return hipHostMalloc(p, bytes, hipHostMallocDefault);
}
static void host_free(void *p, size_t bytes) {
  if (!p) return;
  if (cache_give(-1, p, bytes)) return;
  (void)hipHostFree(p);
}

struct DevBuf {  // grow-only device scratch
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    hipError_t e = dev_alloc(&p, bytes + bytes / 8 + 4096, &cap);
    if (e != hipSuccess) p = nullptr, cap = 0;
    return e;
  }
  hipError_t ensure_exact(size_t bytes) {  // (a buffer that is planned once from the free memory: no eighth on top)
    if (bytes <= cap) return hipSuccess;
    release();
    hipError_t e = dev_alloc(&p, bytes + 4096, &cap);
    if (e != hipSuccess) p = nullptr, cap = 0;
    return e;
  }
  void release() {
    dev_free(p, cap);
    p = nullptr;
    cap = 0;
  }
};

struct HostBuf {  // grow-only pinned host scratch
  void *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    hipError_t e = host_alloc(&p, bytes + bytes / 8 + 4096, &cap);
    if (e != hipSuccess) p = nullptr, cap = 0;
    return e;
  }
  void release() {
    host_free(p, cap);
    p = nullptr;
    cap = 0;
  }
};

struct TimedEvent {
  int kid;
  hipEvent_t a, b;
  bool a_shared;  // `a` is the previous timer's `b` (back-to-back launches share the event)
};

}  // namespace

struct shk_group;
struct shk_ctx {
  shk_config cfg{};
  shk_group *group = nullptr;  // a multi-device context (cfg.n_devices > 1): everything lives in the group
  uint32_t n_lanes = 1;
  // owner share (cfg.n_owners > 1): this context holds the k-mers of owner `owner_id` only
  uint32_t n_owners = 1, owner_bits = 0, owner_id = 0;
  // cfg.n_owners == 1 said explicitly: a "share" that is the whole key space — the context takes the owner-layout
  // route and the shk_xchg_* rounds like any owner share (one segment), so that the exchange path of a W-rank job runs
  // unchanged in a world of one (OwnerCounter over a one-rank communicator; configs[3] on a single card)
  bool share_w1 = false;
  bool is_share() const { return owner_bits != 0 || share_w1; }
  int64_t xchg_lane_fixed = -1;  // the next shk_xchg_scatter_device sends every read to this chunk lane (multi-device shk_ingest_batch)
  uint32_t n_cus = 256;  // compute units of the device (multiProcessorCount)
  uint32_t n_cus_scatter = 256;  // … that the persistent scatter takes (n_cus − shk_config.reserve_cus)
  hipStream_t stream = nullptr;
  // table
  TableRef tb{};
  // device state
  // Control block: every small piece of device state in ONE allocation, mirrored in pinned host
  // memory with the same layout, so that a reset is one fill launch and the end of a run one copy:
  //   [DevStats][lane_bases × n_lanes][pad to 16 B] | [HistoTotals][lane sums × n_lanes][4 words][pad to 16 B][hist × chunks·(histo_max+2)]
  // Everything from HistoTotals on is a SUM over contexts (what shk_finalize_begin hands out for an in-place
  // reduction); the LIVE per-lane base counters the counting kernels add to lie in front of it, so that a repeated
  // _begin / reduce / _end never sums a sum — k_fin_extras copies them into the summed part every time.
  uint8_t *d_ctl = nullptr, *h_ctl = nullptr;
  // The histogram a FRESH page pass left on its way out (k_pages32<true, true>): rows per page in fh_partial / fh_tot,
  // good for finalize as long as nothing else has touched the table since (tb_fresh — every writer's and reader's
  // first call — says so; so do a grow and a reset).
  DevBuf fh_partial, fh_tot;
  bool fused_valid = false, fused_off = false;
  bool flush_for_finalize = false;  // the settle in progress is finalize's own (the last page pass of the job, as far as anybody knows)
  uint32_t fused_pages = 0;
  size_t ctl_bytes = 0, ctl_hist_off = 0, ctl_tot_off = 0, ctl_alloc = 0, ctl_alloc_h = 0;
  unsigned long long *d_lane_sum = nullptr, *h_lane_sum = nullptr;
  DevStats *d_stats = nullptr;
  DevStats *h_stats = nullptr;
  unsigned long long *d_lane_bases = nullptr, *h_lane_bases = nullptr;
  // 4 words behind the lane counters for shk_finalize_begin/_end: reads ingested, bases read, "my scan ran over a
  // table that has to be repaired", the caller's word — everything from d_tot to the end of the block is a SUM
  unsigned long long *d_extra = nullptr, *h_extra = nullptr;
  bool fin_scanned = false, fin_was_unsettled = false, fin_summed = false;
  unsigned long long *d_hist = nullptr;
  HistoTotals *d_tot = nullptr;
  HistoTotals *h_totp = nullptr;
  HistoTotals h_tot{};         // the totals of the last finalize
  const uint64_t *h_hist = nullptr;  // → the mirror's histogram
  // scratch
  hipStream_t copy_stream = nullptr;
  static constexpr int NST = 6;   // staging sets of a host-buffer ingest: NST − 1 slices' copies are queued ahead of the count
  hipEvent_t copy_done[NST] = {};
  int stage_last = -1;            // the staging set the last slice of the previous host-buffer ingest took (its count may still be in flight)
  bool zero_count_keys = false;   // some key may have been inserted with count 0 (shk_insert_counts, merges): k_histo reads the keys
  bool lds_attr_scatter = false, lds_attr_rescatter = false, lds_attr_scatter_own = false, lds_attr_scatter64 = false;  // hipFuncSetAttribute done for this context's device
  HostBuf h_rebased[NST];           // pinned staging of a slice's re-based offsets (a pageable source would make the copy synchronous)
  DevBuf in_bases, in_offsets, st_bases[NST], st_offsets[NST], startbits, tiles, spillA, spillB, misc, part, part2, part3, part_meta;
  DevBuf pk_stage[NST], nm_stage[NST], nz_dev[NST], pk_ascii;
  DevBuf xw_kmers, xw_lanes, xw_count;  // the wide exchange round's output (shk_xchg_wide_scatter_device)
  HostBuf hp_pk[NST], hp_nm[NST];  // ASCII host batches packed on the host (ingest_host): a slice's 2-bit stream and N mask, pinned
  HostBuf nz_host[NST];             // … and the non-zero words of a slice's N mask, when they are few (index, word)  // packed input: the staged streams of a slice; a whole batch unpacked (device-resident packed ingest)
  DevBuf xbuf, xspill;            // owner layout: the level-1 records of a launch by [owner][lane][super-page]; the foreign spill list
  DevBuf xbuf_alt, part_meta_alt;  // the OTHER exchange buffer and cursor block: shk_xchg_scatter_device takes the two in turn
  uint64_t xspill_cap = 0;
  // host counters
  std::vector<uint64_t> lane_reads;
  uint64_t n_reads_read = 0, n_bases_read = 0;
  uint64_t n_grows = 0, n_spilled = 0, n_inserted = 0;
  uint64_t own_p0 = 0, own_p1 = 0;  // owned page range for finalize (0,0 = all)
  bool own_set = false;
  // shk_reset does not clear the table: the first page pass that covers every page and lane writes it whole
  // (k_pages32<true>); anything else that touches the table first clears it then (tb_fresh)
  bool tb_stale = false;
  // the histogram and totals on the device still hold what the last finalize read back: cleared by the next
  // reset, by the next ingest's k_mark_starts on its way, or by the next scan itself — not by a launch per finalize
  bool hist_dirty = false;
  uint64_t job_idx = 0;      // shk_reset calls since shk_reset_timings (SHK_FLAG_TIMING_SAMPLED)
  bool timing_now = true;
  std::vector<uint64_t> held_keys;   // spilled records taken off the device during a grouped flush (flush_acc)
  std::vector<uint32_t> held_lanes, held_counts;
  uint32_t own_share_n = 0, own_share_id = 0;  // … or as a share of the pages, resolved when a scan is launched
  bool finalized = false, poisoned = false;
  bool finalize_redone = false;  // the last finalize repeated its histogram scan after repairing spills
  bool hist_ready = false;  // finalize got as far as the histograms and totals (then failed an invariant, io.rs:1042-1047: the reference has its histo_vecs by then)
  bool unsettled = false;  // a counting launch whose outcome the host has not looked at yet
  uint64_t unsettled_spill_cap = 0;
  int poison_code = 0;
  std::string err;
  // timing
  std::vector<TimedEvent> events;
  std::vector<hipEvent_t> event_pool;
  hipEvent_t done_ev = nullptr;   // finalize: "the control block has been copied back"
  hipEvent_t xs_ev = nullptr;     // shk_xchg_scatter_begin: "the statistics behind the scatter have been copied back"
  bool xs_pending = false, xs_late = false;
  uint64_t n_absorbs = 0, xs_absorbs = 0;  // absorbs launched so far / when the pending scatter was launched
  hipEvent_t chain_ev = nullptr;  // end event of the last timer (see ScopedTimer)
  uint64_t cur_blocks = 1;        // 1000-read blocks in the batch being counted (ingest_core)
  double cur_kmer_ratio = 1.0;    // (k-mers ÷ bases) of the batch being counted if every read has ≥ k bases: n_bases − (k−1)·n_seqs over n_bases
  // Deferred page passes (see count_tiles): partitioned records of several batches wait here
  DevBuf acc_buf, acc_cur;        // page regions (4-B records block-interleaved, or 8-B records lane by lane) and their cursors
  bool acc_rec32 = true;          // which kind of record the regions were planned for
  uint64_t acc_budget_max = 0;    // records the regions were sized for (a window's budget stays below)
  bool acc_active = false;        // the regions hold records that k_pages32 has not counted yet
  uint32_t acc_lp = 0, acc_cap = 0, acc_region_lanes = 1;  // geometry the regions were planned for
  uint64_t acc_records_ub = 0, acc_budget = 0;              // records in the regions (upper bound) / allowed
  std::vector<uint64_t> acc_lane_ub;                        // … per chunk lane (a lane's regions hold a lane's share)
  uint64_t acc_lane_budget = 0;                             // records one lane's regions were sized for
  uint64_t acc_spill_cap = 0;     // ONE spill list per accumulation window: every launch of it uses this capacity
  uint64_t acc_nd0 = 0;           // distinct keys when the window was planned
  double acc_new_frac = 1.0;      // new keys per record in the last window (1 = nothing known yet)
  bool chain_from_mark = false;   // nothing was enqueued between k_mark_starts' timer and the first scatter
  shk_timings timings{};
};

namespace {

int fail(shk_ctx *c, int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c)
    c->err = buf;
  else
    g_create_error = buf;
  return code;
}

#define HIPC(c, expr)                                                                       \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return fail(c, e__ == hipErrorOutOfMemory ? SHK_ERR_NOMEM : SHK_ERR_HIP,              \
                  "HIP error %s at %s:%d (%s)", hipGetErrorString(e__), __FILE__, __LINE__, \
                  #expr);                                                                   \
  } while (0)

// HIP-event timer around one launch.  Back-to-back launches can CHAIN: a timer constructed with
// chain = true starts from the end event of the previous timer (c->chain_ev, valid only if nothing
// else has been enqueued on the stream since) instead of recording an event of its own — every
// event record costs the stream a few µs, and the bench times four kernels per step.
struct ScopedTimer {
  shk_ctx *c;
  int kid;
  hipEvent_t a = nullptr, b = nullptr;
  bool shared = false;
  ScopedTimer(shk_ctx *c_, int kid_, bool chain = false) : c(c_), kid(kid_) {
    if (!(c->cfg.flags & SHK_FLAG_TIMING) || !c->timing_now) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!c->event_pool.empty()) {
        e = c->event_pool.back();
        c->event_pool.pop_back();
      } else {
        (void)hipEventCreate(&e);
      }
      return e;
    };
    if (chain && c->chain_ev) {
      a = c->chain_ev;
      shared = true;
    } else {
      a = get();
      (void)hipEventRecord(a, c->stream);
    }
    b = get();
  }
  ~ScopedTimer() {
    if (!a) return;
    (void)hipEventRecord(b, c->stream);
    c->events.push_back({kid, a, b, shared});
    c->chain_ev = b;  // whoever enqueues anything else before the next timer must not chain
  }
};

void resolve_timings(shk_ctx *c) {
  c->chain_ev = nullptr;  // its event goes back to the pool below
  for (auto &ev : c->events) {
    (void)hipEventSynchronize(ev.b);
    float ms = 0;
    if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
      c->timings.ms[ev.kid] += ms;
      c->timings.launches[ev.kid] += 1;
    }
    if (!ev.a_shared) c->event_pool.push_back(ev.a);
    c->event_pool.push_back(ev.b);
  }
  c->events.clear();
}

inline uint32_t grid_for(uint64_t n_items, uint32_t per_block, uint32_t cap_blocks) {
  uint64_t g = (n_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap_blocks) g = cap_blocks;
  return (uint32_t)g;
}

// One launch: table t ← empty (keys = EMPTY, counts = 0) and, with `ctl`, the control block ←
// initial state.
int fill_state(shk_ctx *c, const TableRef &t, bool ctl) {
  FillSegs f{};
  f.ptr[0] = t.keys;
  f.n16[0] = t.cap * sizeof(uint64_t) / 16;
  f.val[0] = ~0ull;
  f.ptr[1] = t.vals;
  f.n16[1] = t.cap * sizeof(uint32_t) * t.n_lanes / 16;
  if (ctl) {  // all zero, except the first 16-B word = DevStats.bad = ~0 ("no invalid byte")
    f.ptr[2] = c->d_ctl;
    f.n16[2] = 1;
    f.val[2] = ~0ull;
    f.ptr[3] = c->d_ctl + 16;
    f.n16[3] = c->ctl_bytes / 16 - 1;
  }
  const uint64_t total = f.n16[0] + f.n16[1] + f.n16[2] + f.n16[3];
  hipLaunchKernelGGL(k_fill, dim3(grid_for(total, WG * 4, 4096)), dim3(WG), 0, c->stream, f);
  HIPC(c, hipGetLastError());
  return SHK_OK;
}

// The table as everybody but a FRESH page pass needs it: cleared, if the last reset left that for later.
static void fused_drop(shk_ctx *c) {
  if (!c->fused_valid) return;
  c->fused_valid = false;
  c->hist_dirty = true;  // (the pass may have added high bins to the histogram itself: whoever scans next clears it first)
}
int tb_fresh(shk_ctx *c) {
  fused_drop(c);
  if (!c->tb_stale) return SHK_OK;
  c->tb_stale = false;
  return fill_state(c, c->tb, false);
}

static void free_table(TableRef &t) {
  // (a block from the cache may be larger than what was asked for: it goes back under the size it was asked at — the
  // cache matches requests against that, and the table's arrays only ever ask for powers of two times a lane count)
  dev_free(t.keys, t.cap * sizeof(uint64_t));
  dev_free(t.vals, t.cap * sizeof(uint32_t) * t.n_lanes);
  t.keys = nullptr;
  t.vals = nullptr;
}
int alloc_table(shk_ctx *c, uint32_t log_pages, TableRef *out, bool cleared = true) {
  TableRef t{};
  t.log_pages = log_pages;
  t.n_lanes = c->n_lanes;
  t.key_bits = 2 * c->cfg.k;
  t.owner_bits = c->owner_bits;
  t.owner_id = c->owner_id;
  t.cap = (uint64_t)PAGE_SLOTS << log_pages;
  size_t got = 0;  // (the table's arrays are asked for and given back at their exact sizes)
  HIPC(c, dev_alloc((void **)&t.keys, t.cap * sizeof(uint64_t), &got));
  hipError_t e = dev_alloc((void **)&t.vals, t.cap * sizeof(uint32_t) * t.n_lanes, &got);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    dev_free(t.keys, t.cap * sizeof(uint64_t));
    return fail(c, SHK_ERR_NOMEM, "out of device memory allocating %llu-slot table (%u lanes)",
                (unsigned long long)t.cap, t.n_lanes);
  }
  *out = t;
  return cleared ? fill_state(c, t, false) : SHK_OK;
}

int read_stats(shk_ctx *c) {
  HIPC(c, hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(DevStats), hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  return SHK_OK;
}

int grow_to(shk_ctx *c, uint32_t new_log_pages) {
  if (new_log_pages <= c->tb.log_pages) return SHK_OK;
  if (c->acc_active) {  // the waiting records were partitioned for the current geometry
    int rcf = settle(c);
    if (rcf != SHK_OK) return rcf;
    if (new_log_pages <= c->tb.log_pages) return SHK_OK;
  }
  fused_drop(c);
  TableRef nt{};
  int rc = env_int("SHK_TEST_GROW_NOMEM", 0) && c->acc_buf.p ? SHK_ERR_NOMEM : alloc_table(c, new_log_pages, &nt);  // (test hook: the first try fails)
  if (rc == SHK_ERR_NOMEM && c->acc_buf.p && !c->acc_active) {
    // The waiting regions are empty here (settled above) and may hold most of the card (a window planned under a
    // capacity hint that said the table would not grow takes the free memory but 48 GiB): give them back and try
    // again; the next window plans them anew beside the larger table.
    SHK_TRACEF("grow_to: no room for the %u-page table beside %.1f GiB of waiting regions -> regions given back\n", 1u << new_log_pages, c->acc_buf.cap / 1073741824.0);
    c->acc_buf.release();
    c->acc_budget_max = 0;
    rc = alloc_table(c, new_log_pages, &nt);
    if (rc == SHK_OK) c->err.clear();
  }
  if (rc != SHK_OK) return rc;
  if (c->tb_stale) {
    c->tb_stale = false;  // (nothing to carry over: the new table is the cleared one)
  } else {
    ScopedTimer t(c, SHK_K_GROW);
    hipLaunchKernelGGL(k_grow, dim3(grid_for(c->tb.cap, WG, 8192)), dim3(WG), 0, c->stream, c->tb, nt);
  }
  HIPC(c, hipStreamSynchronize(c->stream));
  free_table(c->tb);
  c->tb = nt;
  c->n_grows++;
  return SHK_OK;
}

uint32_t log_pages_for(uint64_t want_slots, uint32_t owner_bits = 0) {
  uint32_t lp = 0;
  while (((uint64_t)PAGE_SLOTS << lp) < want_slots && lp + owner_bits < MAX_LOG_PAGES) lp++;
  return lp;
}

// Make room so that `expect_new` further distinct k-mers keep the load ≤ 1/2.
int ensure_capacity(shk_ctx *c, uint64_t expect_new) {
  uint64_t need = (c->h_stats->n_distinct + expect_new) * 2;
  if (need <= c->tb.cap) return SHK_OK;
  return grow_to(c, log_pages_for(need, c->owner_bits));
}

SpillRef spill_ref(DevBuf &b, uint64_t cap) {
  SpillRef s{};
  uint8_t *p = (uint8_t *)b.p;
  s.keys = (uint64_t *)p;
  s.lanes = (uint32_t *)(p + cap * 8);
  s.counts = (uint32_t *)(p + cap * 12);
  s.cap = cap;
  return s;
}

// After a counting launch: re-insert the spilled records one by one (global atomics, exact) until none
// remain.  Most spills need no bigger table — a record whose key sits too far from home for a tag, a
// page region that overflowed on skewed input — so the first round inserts into the table as it
// is unless every spilled record being a new key would take the load past 1/2; what spills again
// found its page full, and from then on the table at least doubles per round.
int drain_spill(shk_ctx *c, uint64_t spill_cap) {
  DevBuf *cur = &c->spillA, *nxt = &c->spillB;
  if (c->h_stats->spill_count > 0) {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  for (uint32_t round = 0; c->h_stats->spill_count > 0; ++round) {
    uint64_t n = c->h_stats->spill_count;
    if (n > spill_cap)
      return fail(c, SHK_ERR_INVARIANT, "spill list overflow (%llu > %llu)", (unsigned long long)n,
                  (unsigned long long)spill_cap);
    c->n_spilled += n;
    if (env_int("SHK_DEBUG_SPILL", 0)) {  // experiment hook: show what spilled
      const uint64_t m = std::min<uint64_t>(n, 48);
      std::vector<uint64_t> hk(m);
      std::vector<uint32_t> hl(m);
      SpillRef in0 = spill_ref(*cur, spill_cap);
      (void)hipMemcpy(hk.data(), in0.keys, m * 8, hipMemcpyDeviceToHost);
      (void)hipMemcpy(hl.data(), in0.lanes, m * 4, hipMemcpyDeviceToHost);
      for (uint64_t i = 0; i < m; ++i) {
        const uint64_t y = mix_key(hk[i], 2 * c->cfg.k);
        fprintf(stderr, "[spill %llu/%llu] key %011llx lane %u page %llu rec %08llx\n", (unsigned long long)i, (unsigned long long)n,
                (unsigned long long)hk[i], hl[i], (unsigned long long)(y >> (2 * c->cfg.k - c->tb.log_pages)), (unsigned long long)(y & 0xFFFFFFFFull));
      }
    }
    uint32_t lp = log_pages_for((c->h_stats->n_distinct + n) * 2, c->owner_bits);
    if (round > 0) {
      if (c->tb.log_pages + c->owner_bits >= MAX_LOG_PAGES)
        return fail(c, SHK_ERR_NOMEM, "table full: %llu distinct k-mers do not fit 2^%u pages",
                    (unsigned long long)c->h_stats->n_distinct, MAX_LOG_PAGES - c->owner_bits);
      lp = std::max(lp, c->tb.log_pages + 1);
    }
    int rc = grow_to(c, lp);
    if (rc != SHK_OK) return rc;
    HIPC(c, nxt->ensure(n * 16));
    SpillRef in = spill_ref(*cur, spill_cap), out = spill_ref(*nxt, n);
    HIPC(c, hipMemsetAsync(&c->d_stats->spill_count, 0, sizeof(unsigned long long), c->stream));
    {
      ScopedTimer t(c, SHK_K_INSERT);
      hipLaunchKernelGGL(k_insert, dim3(grid_for(n, WG, 4096)), dim3(WG), 0, c->stream, in.keys,
                         in.lanes, in.counts, n, 0u, c->tb, c->d_stats, out);
    }
    rc = read_stats(c);
    if (rc != SHK_OK) return rc;
    std::swap(cur, nxt);
    spill_cap = n;
  }
  if (cur != &c->spillA) std::swap(c->spillA, c->spillB);
  return SHK_OK;
}

constexpr uint64_t MAX_SUB_BASES = 1ull << 28;  // bases per counting launch (bounds scratch)

// Core ingest over device-resident input.  lane_fixed >= 0: every read to that chunk lane
// (drain_batch, io.rs:356-358); lane_fixed < 0: stripe by running read index
// (read i → chunk (i/1000) % n_chunks, io.rs:340-343,355-361).
int ingest_core(shk_ctx *c, const uint8_t *d_bases, const uint64_t *d_offsets, uint64_t n_seqs,
                uint64_t n_bases, int64_t lane_fixed, XchgOut *xo = nullptr, uint64_t off_bias = 0) {
  if (c->poisoned) return fail(c, c->poison_code, "%s", c->err.c_str());
  if (!(xo && xo->late_settle)) {
    int rc0 = settle_light(c);  // the previous launch's spill list / scratch must be done with
    if (rc0 != SHK_OK) return rc0;
  }
  c->finalized = c->hist_ready = false;
  c->chain_from_mark = false;
  const uint64_t g0 = c->n_reads_read;
  const uint32_t NL = c->n_lanes;
  // host-side bookkeeping that depends only on read indices
  uint64_t n_blocks = 1;
  const uint64_t first = (g0 % 1000 == 0) ? 1000 : 1000 - g0 % 1000;
  const bool striped = lane_fixed < 0 && NL > 1;
  if (striped) {
    n_blocks = n_seqs <= first ? 1 : 1 + (n_seqs - first + 999) / 1000;
    for (uint64_t j = 0; j < n_blocks; ++j) {
      uint64_t r0 = j == 0 ? 0 : first + (j - 1) * 1000;
      uint64_t r1 = std::min(first + j * 1000, n_seqs);
      c->lane_reads[((g0 + r0) / 1000) % NL] += r1 - r0;
    }
  } else {
    c->lane_reads[lane_fixed < 0 ? 0 : (uint32_t)lane_fixed] += n_seqs;
  }
  if (lane_fixed < 0) c->n_reads_read += n_seqs;  // explicit-lane batches do not advance striping
  c->n_bases_read += n_bases;
  if (n_seqs == 0 || n_bases == 0) return xo ? (xo->wide ? xw_scatter_launch(c, BatchRef{}, 0, xo) : xchg_scatter_launch(c, BatchRef{}, 0, xo)) : SHK_OK;

  // 1. read-start bitmap (+ tile list when the batch spans several chunk lanes)
  const size_t sb_words = (size_t)(n_bases / 32 + 3);
  HIPC(c, c->startbits.ensure(sb_words * 4));
  // (no memset: k_mark_starts writes every word of the bitmap)
  uint64_t n_tiles_ub = (n_bases + TILE_T - 1) / TILE_T;
  const bool multi = striped && n_blocks > 1;
  if (multi) {
    n_tiles_ub += n_blocks;
    HIPC(c, c->tiles.ensure(n_tiles_ub * sizeof(TileDesc)));
  }
  // The table geometry of the first counting launch is fixed here, so that k_mark_starts can
  // clear that launch's partition cursors and the spill counter on its way (capacity heuristic
  // when no hint was given: assume ≥ 4× coverage; the spill path keeps the result exact whatever
  // the truth is).
  // (a batch of at most MAX_SUB_BASES bases is ONE counting launch even when the partial tiles at its 1000-read
  // block boundaries take the tile count past MAX_SUB_BASES / TILE_T: a second launch would be a second page pass)
  const uint64_t tiles_per_sub = n_bases <= MAX_SUB_BASES ? std::max<uint64_t>(MAX_SUB_BASES / TILE_T, n_tiles_ub) : MAX_SUB_BASES / TILE_T;
  {
    const uint64_t first_kmers_ub = std::min(tiles_per_sub, n_tiles_ub) * TILE_T;
    int rc = ensure_capacity(c, c->cfg.table_capacity_hint ? 0 : (first_kmers_ub / 4) >> c->owner_bits);
    if (rc != SHK_OK) return rc;
  }
  c->cur_blocks = n_blocks;
  {
    const double cut = (double)(c->cfg.k - 1) * (double)n_seqs;
    c->cur_kmer_ratio = cut < (double)n_bases ? std::max(0.125, 1.0 - cut / (double)n_bases) : 0.125;
  }
  uint32_t n_cursor_words = 0;
  if (xo && xo->wide) {  // the wide exchange round: no partition cursors at all
    if (n_tiles_ub > tiles_per_sub) return fail(c, SHK_ERR_BAD_ARG, "an exchange batch takes at most %llu bases", (unsigned long long)MAX_SUB_BASES);
    HIPC(c, c->part_meta.ensure(64));
  } else if (xo) {  // exchange round: level-1 scatter only, every owner's records (shk_xchg_scatter_device)
    if (n_tiles_ub > tiles_per_sub) return fail(c, SHK_ERR_BAD_ARG, "an exchange batch takes at most %llu bases", (unsigned long long)MAX_SUB_BASES);
    int rc = xchg_prepare_cursors(c, &n_cursor_words);
    if (rc != SHK_OK) return rc;
  } else {
    const uint64_t first_kmers_ub = std::min(tiles_per_sub, n_tiles_ub) * TILE_T;
    const bool defer = first_launch_defers(c, first_kmers_ub, striped && n_blocks > 1);
    int rc = defer ? acc_prepare(c, acc_records_est(c, first_kmers_ub), striped && n_blocks > 1 ? -1 : (int64_t)(lane_fixed >= 0 ? (uint64_t)lane_fixed : (striped ? (g0 / 1000) % NL : 0))) : (c->acc_active ? settle(c) : SHK_OK);  // (may flush: launch + settle)
    if (rc == SHK_OK) rc = prepare_cursors(c, striped && n_blocks > 1, defer && first_launch_defers(c, first_kmers_ub, striped && n_blocks > 1), &n_cursor_words);
    if (rc != SHK_OK) return rc;
  }
  {
    ScopedTimer t(c, SHK_K_MARK);
    hipLaunchKernelGGL(k_mark_starts, dim3((uint32_t)((n_seqs + WG - 1) / WG)), dim3(WG), 0,
                       c->stream, d_offsets, n_seqs, n_bases, (uint32_t *)c->startbits.p, (uint64_t)sb_words,
                       (unsigned int *)c->part_meta.p, n_cursor_words,
                       xo && xo->late_settle ? nullptr : &c->d_stats->spill_count,  // (the previous absorb's spills are still to be looked at)
                       off_bias,
                       (uint4 *)c->d_tot, c->hist_dirty ? (uint32_t)(sizeof(HistoTotals) / 16) : 0u, (uint4 *)c->d_hist,
                       c->hist_dirty ? (uint32_t)((c->ctl_bytes - c->ctl_hist_off) / 16) : 0u);
    c->hist_dirty = false;
    if (multi)
      hipLaunchKernelGGL(k_build_tiles, dim3(1), dim3(TB_WG), 0, c->stream, d_offsets, n_seqs, g0,
                         NL, n_blocks, (TileDesc *)c->tiles.p, c->d_stats, off_bias);
  }
  c->chain_from_mark = true;  // (reset by the first scatter; nothing is enqueued in between)
  BatchRef b{};
  b.bases = d_bases;
  b.startbits = (const uint32_t *)c->startbits.p;
  b.n_bases = n_bases;
  b.tiles = multi ? (const TileDesc *)c->tiles.p : nullptr;
  b.stats = c->d_stats;
  b.n_tiles_single = (n_bases + TILE_T - 1) / TILE_T;
  b.tile_first = 0;
  b.tile_count = n_tiles_ub;
  b.lane0 = lane_fixed >= 0 ? (uint32_t)lane_fixed : (striped ? (uint32_t)((g0 / 1000) % NL) : 0u);
  b.k = (int)c->cfg.k;

  if (xo) return xo->wide ? xw_scatter_launch(c, b, n_tiles_ub * TILE_T, xo) : xchg_scatter_launch(c, b, n_tiles_ub * TILE_T, xo);
  // 3. count, in sub-ranges of tiles
  for (uint64_t ta = 0; ta < n_tiles_ub; ta += tiles_per_sub) {
    uint64_t tn = std::min(tiles_per_sub, n_tiles_ub - ta);
    uint64_t sub_kmers_ub = tn * TILE_T;
    b.tile_first = ta;
    b.tile_count = tn;
    if (ta) {
      int rcs = settle_light(c);
      if (rcs != SHK_OK) return rcs;
    }
    int rc = ta ? ensure_capacity(c, c->cfg.table_capacity_hint ? 0 : (sub_kmers_ub / 4) >> c->owner_bits) : SHK_OK;
    if (rc != SHK_OK) return rc;
    rc = count_tiles(c, b, sub_kmers_ub, /*prezeroed=*/ta == 0);
    if (rc != SHK_OK) return rc;
  }
  return SHK_OK;
}

}  // namespace

// count_tiles: one counting pass over b's tile range; picks the paged (LDS) or the direct
// (global atomics) path, then repairs spills and keeps the load ≤ 1/2.
// Paged (LDS) counting is possible when every page has an LDS histogram slot and the lane
// rounds stay few; it pays when the batch is at least comparable to the table, because every
// touched page is read and written once per pass.
static bool paged_feasible(const shk_ctx *c) {
  // one level up to MAX_PARTS pages, two levels (super-pages of ≤ MAX_PARTS pages) beyond
  // (Chunk lanes: 16 until round 4 — "the lane rounds stay few".  Nothing in the paged passes depends on it: with
  // 40 lanes config 2's batch runs at 118 Gbases/s paged against 19.5 through the global atomics, with 100 — what
  // the reference's own historical runs used, sharkmer_viewer/tests/data/Cordagalma.stats — see DESIGN.md §8.)
  return c->tb.log_pages + c->owner_bits <= 20 && c->n_lanes <= (uint32_t)env_int("SHK_PAGED_MAX_LANES", (int)shk::SC32_MAX_LANES);
}
static bool paged_pays(const shk_ctx *c, uint64_t sub_kmers_ub) {
  return c->tb.log_pages >= 8 && sub_kmers_ub >= c->tb.cap / 2;
}

constexpr int SC_NT = 512;  // threads of the partition count / sorted scatter workgroups

static uint32_t region_cap(uint64_t n_records_ub, uint64_t n_regions, uint64_t pads) {
  // mean load + 25 % + worst-case padding + slack, even
  uint64_t cap = n_records_ub / n_regions + n_records_ub / n_regions / 4 + pads + 1024;
  return (uint32_t)std::min<uint64_t>((cap + 1) & ~1ull, 0x7FFFFFF0ull);
}

// Partition geometry of a paged counting pass over the current table.
// One level: one partition per page.  More than MAX_PARTS pages: level 1 groups the records by
// super-page (2^log_sub consecutive pages), level 2 (k_part_rescatter) by page.
struct PartGeom {
  uint32_t lp, n_pages, log_p1, log_sub, P1;
  uint32_t lw, lpg;  // owner bits; page bits of the virtual global table (lp + lw)
  bool two_level;
  uint32_t cursor_words() const { return P1 + (two_level ? n_pages : 0); }
};
// An owner share (lw > 0) is the slice of one owner of a virtual table of 2^(lp + lw) pages: the
// partition levels are laid over THAT table — level 1 fans out over its top log_p1 bits, owner bits
// included, so that the same pass serves the exchange between owners — and always in the two-level
// form (log_sub may be 0).
static PartGeom part_geom(const shk_ctx *c) {
  PartGeom g{};
  g.lp = c->tb.log_pages;
  g.lw = c->owner_bits;
  g.lpg = g.lp + g.lw;
  g.n_pages = 1u << g.lp;
  const uint32_t lvl1_log = (uint32_t)env_int("SHK_LEVEL1_LOG", 10);                      // test hooks: force
  const uint32_t two_level_min = (uint32_t)env_int("SHK_TWO_LEVEL_MIN_PAGES", MAX_PARTS);  // the two-level path
  g.two_level = c->is_share() || g.n_pages > std::min<uint32_t>(two_level_min, (uint32_t)MAX_PARTS);
  g.log_p1 = g.two_level ? std::max(std::min(lvl1_log, g.lpg), g.lw) : g.lpg;
  g.log_sub = g.lpg - g.log_p1;
  g.P1 = 1u << g.log_p1;
  return g;
}

// 4-byte records (k_scatter32 / k_part_scatter_sorted<.., true> + k_pages32) when a record fits:
// 11 ≤ 2k - log_pages ≤ 32 (the low bits of the mixed key below the page bits); with two levels
// the level-1 record, 2k - log_p1 bits, must fit as well.
static bool use_rec32(const shk_ctx *c, const PartGeom &g) {
  const uint32_t rbits = 2 * c->cfg.k >= g.lpg ? 2 * c->cfg.k - g.lpg : 0;
  const uint32_t r1_bits = 2 * c->cfg.k >= g.log_p1 ? 2 * c->cfg.k - g.log_p1 : 0;
  return rbits >= 11 && rbits <= 32 && r1_bits <= 32 && env_int("SHK_REC32", 1) != 0;
}
// (SHK_SC32_NT / SHK_SC32_WGS: build-time shape of k_scatter32 — threads per workgroup (a tile is 16 bases per
// thread) and workgroups per CU)
#ifndef SHK_SC32_NT
#define SHK_SC32_NT 1024
#endif
#ifndef SHK_SC32_WGS
#define SHK_SC32_WGS 1
#endif
constexpr int SC32_NT = SHK_SC32_NT, SC32_TT = 16 * SHK_SC32_NT;  // k_scatter32: 64 KiB of records + 64 KiB of entries in LDS,
constexpr size_t SC32_LDS_MAX = 160 * 1024 - 2048;  // one workgroup per CU (its static LDS is < 2 KiB: per-lane base counts of up to 128 chunk lanes)
static size_t scatter32_lds(uint32_t P1) {
  return (size_t)SC32_TT * 8 + (size_t)P1 * 12 + 32;  // entries (+ P1/2 holes at most) + records + three words per partition (+ the walk's 8 spare counters when P1 < 8)
}
// records-in-LDS scatter (k_scatter32: one 1024-thread workgroup per CU) when its LDS footprint fits
static bool use_scatter32(const shk_ctx *c, const PartGeom &g) {
  return use_rec32(c, g) && scatter32_lds(g.P1) <= SC32_LDS_MAX && env_int("SHK_SCATTER32_LDS", 1) != 0;
}
// ALL-LANES mode: a batch that spans several chunk lanes is partitioned in ONE scatter pass (regions
// and cursors per (lane, page)) and counted by ONE k_pages32 launch that keeps a page in LDS for all
// its lanes — instead of one scatter + one page pass per lane.  One level, k_scatter32 only.
static bool use_all_lanes(const shk_ctx *c, const PartGeom &g, bool multi) {
  return multi && !g.two_level && use_scatter32(c, g) && c->n_lanes <= shk::SC32_MAX_LANES && env_int("SHK_ALL_LANES", 1) != 0;
}
// The OWNER LAYOUT route (xl_count): ONE level-1 scatter for the tiles of every chunk lane into regions
// ordered [owner][lane][super-page], then ONE level-2 pass per owner segment into the waiting (lane, page)
// regions.  It is what an owner share always takes, and what a whole-key-space context takes for a
// deferred two-level pass over several lanes (instead of a scatter + re-scatter per lane).
static bool xl_feasible(const shk_ctx *c, const PartGeom &g) {
  return g.two_level && use_scatter32(c, g) && (1u << g.log_sub) <= (uint32_t)MAX_PARTS && c->n_lanes <= shk::SC32_MAX_LANES &&
         g.lpg <= MAX_LOG_PAGES;
}
static bool xl64_feasible(const shk_ctx *c, const PartGeom &g);
// … and the same layout with 8-byte records for a share whose records do not fit a word (k > 21): its own ingest
static bool xl64_route(const shk_ctx *c, const PartGeom &g) { return c->is_share() && !xl_feasible(c, g) && xl64_feasible(c, g); }
static bool xl_route(const shk_ctx *c, const PartGeom &g, bool multi, bool defer) {
  if (!xl_feasible(c, g)) return false;
  if (c->is_share()) return true;
  return defer && multi && env_int("SHK_XL", 1) != 0;
}

// Which way a counting launch goes.
//   PATH_PAGED  : partition, then count every page in LDS right away — pays when the batch is at
//                 least comparable to the table (every page is streamed through LDS once per pass)
//   PATH_DEFER  : partition only; the 4-byte records wait in the page regions, which go on filling
//                 over the following launches, and ONE page pass counts them when their number
//                 has become comparable to the table (or when anybody needs the table).  This is
//                 what keeps small batches into a large table (the usual case: reads stream in,
//                 the table holds a genome) at the partition rate instead of the rate of scattered
//                 global atomics
//   PATH_DIRECT : global atomics; whatever the paged paths cannot take (> 16 lanes, records that
//                 do not fit 4 bytes on a table too small for an immediate pass, …)
enum CountPath { PATH_DIRECT, PATH_PAGED, PATH_DEFER };
static bool paged_feasible(const shk_ctx *c);
static bool paged_pays(const shk_ctx *c, uint64_t sub_kmers_ub);
static CountPath count_path(const shk_ctx *c, uint64_t sub_kmers_ub, bool multi = false) {
  if (!paged_feasible(c) || (c->cfg.flags & SHK_FLAG_FORCE_DIRECT)) return PATH_DIRECT;
  // a batch over several chunk lanes into a two-level table: the owner-layout route (ONE level-1 pass, ONE
  // level-2 pass, ONE page launch that keeps a page's tags in LDS for all its lanes) instead of a scatter,
  // a re-scatter and a page pass per lane, each re-reading the keys — also when the batch is large enough
  // to be counted at once (the records just wait for the page pass that finalize, or the budget, asks for)
  if (multi && c->n_lanes > 1 && !c->is_share() && env_int("SHK_XL", 1) != 0 && env_int("SHK_DEFER", 1) != 0) {
    const PartGeom g2 = part_geom(c);
    if (g2.two_level && xl_feasible(c, g2)) return PATH_DEFER;
  }
  if (c->is_share()) {  // an owner share: an owner layout (4- or 8-byte records) + deferred page passes, or global atomics (all drop foreign k-mers)
    const PartGeom gs = part_geom(c);
    return (xl_feasible(c, gs) || xl64_feasible(c, gs)) && env_int("SHK_DEFER", 1) != 0 ? PATH_DEFER : PATH_DIRECT;
  }
  if ((c->cfg.flags & SHK_FLAG_FORCE_PAGED) || paged_pays(c, sub_kmers_ub)) return PATH_PAGED;
  const PartGeom g = part_geom(c);
  if (env_int("SHK_DEFER", 1) == 0) return PATH_DIRECT;
  if (use_scatter32(c, g)) return PATH_DEFER;
  // 8-byte records (k-mers too long for a 4-byte remainder, e.g. every table at k = 31) wait in page
  // regions of their own kind; tables of a few pages stay with the global atomics
  if (!use_rec32(c, g) && c->tb.log_pages >= 8) return PATH_DEFER;
  return PATH_DIRECT;
}

// Make room for `kmers_ub` more records in the accumulation regions: plan them if there are none,
// count what is waiting first (flush) if these would not fit any more.
// What a launch of ≤ est records adds to each chunk lane's regions at most: everything to one lane
// (lane_one ≥ 0: an explicit-chunk batch, or a batch inside one 1000-read block), or — striped over the
// blocks of the batch (lane_one < 0) — a lane's share of the blocks.
static uint64_t acc_lane_add(const shk_ctx *c, uint64_t est, int64_t lane_one, uint64_t lane_add_exact = 0) {
  const uint32_t NL = c->n_lanes;
  if (lane_add_exact) return std::min(est, lane_add_exact);  // (the caller knows a bound: an exchange segment's regions per lane)
  if (NL == 1 || lane_one >= 0) return est;
  const uint64_t nb = std::max<uint64_t>(c->cur_blocks, 1);
  return std::min<uint64_t>(est, est / nb * ((nb + NL - 1) / NL) + 2 * TILE_T);
}
static void acc_book(shk_ctx *c, uint64_t est, int64_t lane_one, uint64_t lane_add_exact = 0) {
  c->acc_records_ub += est;
  const uint64_t add = acc_lane_add(c, est, lane_one, lane_add_exact);
  if (c->acc_lane_ub.size() != c->n_lanes) c->acc_lane_ub.assign(c->n_lanes, 0);
  for (uint32_t l = 0; l < c->n_lanes; ++l)
    if (lane_one < 0 || (uint32_t)lane_one == l || c->n_lanes == 1) c->acc_lane_ub[l] += add;
}
static int acc_prepare(shk_ctx *c, uint64_t kmers_ub, int64_t lane_one, uint64_t lane_add_exact) {
  if (c->acc_active) {
    bool full = c->acc_lp != c->tb.log_pages || c->acc_records_ub + kmers_ub > c->acc_budget;
    const uint64_t add = acc_lane_add(c, kmers_ub, lane_one, lane_add_exact);
    for (uint32_t l = 0; l < c->n_lanes && !full && l < c->acc_lane_ub.size(); ++l)
      if ((lane_one < 0 || (uint32_t)lane_one == l) && c->acc_lane_ub[l] + add > c->acc_lane_budget) full = true;
    if (full) {
      SHK_TRACEF("acc_prepare: window full (%llu booked + %llu > %llu, or a lane's %llu + %llu > %llu) -> flush\n", (unsigned long long)c->acc_records_ub,
                 (unsigned long long)kmers_ub, (unsigned long long)c->acc_budget, (unsigned long long)(c->acc_lane_ub.empty() ? 0 : c->acc_lane_ub[0]),
                 (unsigned long long)add, (unsigned long long)c->acc_lane_budget);
      int rc = settle(c);  // flush + settle (may grow the table)
      if (rc != SHK_OK) return rc;
    }
  }
  if (c->acc_active) return SHK_OK;
  const uint64_t n_pages = 1ull << c->tb.log_pages;
  const uint32_t NL = c->n_lanes;
  // Budget: half a table's worth of records (the new keys among them cannot take the load past 1
  // before the next flush looks at it), at least this batch, at most what memory allows.
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  // Budget = the records after which the table should be looked at again: as many as would take the
  // pages to 80 % full if new keys kept arriving at the last window's rate (with a margin; at first
  // every record is assumed to bring a new key), between a quarter of and eight times the table.
  const uint64_t nd = c->h_stats->n_distinct;
  const double room = 0.8 * (double)c->tb.cap - (double)nd;
  const double per_rec = std::max(0.02, std::min(1.0, c->acc_new_frac * 1.3));
  uint64_t budget = room > 0 ? (uint64_t)(room / per_rec) : 0;
  budget = std::min<uint64_t>(std::max<uint64_t>(budget, c->tb.cap / 4), c->tb.cap * 8);
  // A capacity hint is the caller's word on how many distinct k-mers (of this context's share) there will be:
  // when even all of them fit the pages at ≤ 80 %, no window has to end for the table's sake — one page pass
  // (which streams the whole table: 45 ms on a 2^33-slot one) counts whatever the regions can hold.  (A hint
  // that was too low costs time, not exactness: new keys beyond a page's fill limit take the spill path.)
  if (c->cfg.table_capacity_hint && (double)c->cfg.table_capacity_hint <= 0.8 * (double)c->tb.cap) budget = ~0ull >> 1;
  budget = std::max<uint64_t>(budget, kmers_ub);
  if (env_int("SHK_DEFER_BUDGET", 0) > 0)  // test hook: flush early and often
    budget = std::max<uint64_t>((uint64_t)env_int("SHK_DEFER_BUDGET", 0), kmers_ub);
  const PartGeom g = part_geom(c);
  const bool rec32 = use_rec32(c, g);
  // The regions are sized ONCE per table geometry, for the largest window this table may get, and
  // every window's budget stays below that: freeing and re-allocating tens of GB between windows
  // stalls the host for seconds (the driver wipes VRAM that changes hands).
  if (!(c->acc_buf.p && c->acc_lp == c->tb.log_pages && c->acc_rec32 == rec32 && c->acc_region_lanes == NL &&
        kmers_ub <= c->acc_budget_max && acc_lane_add(c, kmers_ub, lane_one, lane_add_exact) <= c->acc_lane_budget)) {
    // lanes each get a full-size region set; 8-byte records also need k_pages' miss queues
    // regions are 1.25 × the window (× 1.5 over several lanes: each lane's share + 50 %); k_pages' miss queues are a
    // fixed MISS_PAGE_MAX entries per page once a region is longer than that (they were booked at 8 B per RECORD, which
    // kept configs[2]'s 12 G records from ever being one window)
    const uint64_t rec_bytes = rec32 ? (NL > 1 ? 8ull : 5ull) + 1 : (NL > 1 ? 15ull : 10ull) + 1;
    const uint64_t miss_bytes = rec32 ? 0ull : n_pages * (uint64_t)MISS_PAGE_MAX * 8ull;
    // two thirds of what is free — or, when a capacity hint says the table will not have to grow (its double is what the
    // third is kept for), all of it but 48 GiB (the spill list's 16, the exchange buffers, staging), and the regions'
    // bytes reckoned as they will be allocated (1.25 x, over several lanes 1.5 x 1.25 x the records, + 1024 and a rounding
    // block per region) instead of a byte per record for the rest: configs[4]'s share — 74 rounds booked at 260 M records
    // = 19.2 G over ten lanes beside a 51 GB table and the reads — was 10 % short of ONE window, and its second page pass
    // read and wrote the whole table again (the pass 53.6 -> 38 ms of a 159 ms job)
    const bool hinted0 = c->cfg.table_capacity_hint && (double)c->cfg.table_capacity_hint <= 0.8 * (double)c->tb.cap;
    const uint64_t keep = (uint64_t)env_int("SHK_WINDOW_KEEP_GIB", 48) << 30;
    const bool roomy = hinted0 && free_b > keep && free_b - keep > free_b / 3 * 2;
    const uint64_t mem_avail = (roomy ? free_b - keep : free_b / 3 * 2) + c->acc_buf.cap;
    uint64_t mem_records = (mem_avail > miss_bytes ? mem_avail - miss_bytes : 0ull) / rec_bytes;
    if (roomy) {
      const uint64_t fixed = (uint64_t)NL * n_pages * (1024 + (1ull << RB_LOG)) * (rec32 ? 4 : 8);
      const uint64_t eighths = (rec32 ? 4ull : 8ull) * (NL > 1 ? 15 : 10);  // bytes per record x 8
      mem_records = std::max<uint64_t>(mem_records, mem_avail > miss_bytes + fixed ? (mem_avail - miss_bytes - fixed) / eighths * 8 : 0ull);
    }
    SHK_TRACEF("acc_prepare: %.1f of %.1f GiB free, %.1f GiB for the regions = %llu records\n", free_b / 1073741824.0, total_b / 1073741824.0,
               mem_avail / 1073741824.0, (unsigned long long)mem_records);
    // up to eight tables' worth of records while that is a few GiB, two and a half tables' worth beyond — and eight
    // again when a capacity hint says that no window will have to end for the table's sake: every page pass streams
    // the whole table (24 GB in and out on configs[2]'s 2^30 slots), so the job should make as few as memory allows
    // (configs[2]: 12 G records in two passes instead of five with eight tables' worth — round 3 — and in ONE with sixteen,
    // memory allowing — round 4: page passes 31.2 → 28.5 ms, the job 117.5 → 122 Gbases/s;
    // twenty-four, since the rounds of an exchange book their segments' capacity, 12 % over what arrives: configs[4]'s share)
    const uint64_t few_gib = (8ull << 30) / (rec32 ? 4 : 8) / NL;
    const bool hinted = c->cfg.table_capacity_hint && (double)c->cfg.table_capacity_hint <= 0.8 * (double)c->tb.cap;
    uint64_t bmax = std::max<uint64_t>(std::min<uint64_t>(c->tb.cap * 8, few_gib), hinted && env_int("SHK_WIDE_WINDOW", 1) ? c->tb.cap * (uint64_t)env_int("SHK_WINDOW_TABLES", 24) : c->tb.cap * 5 / 2);
    if (env_int("SHK_ACC_MAX_MRECORDS", 0) > 0) bmax = std::min<uint64_t>(bmax, (uint64_t)env_int("SHK_ACC_MAX_MRECORDS", 0) << 20);  // test hook: small windows (several contexts on one card)
    bmax = std::max<uint64_t>(std::min(bmax, mem_records), kmers_ub);
    // a lane's regions take a lane's share of the window (+ 50 %: blocks of uneven read lengths), but at
    // least what one launch can put into a single lane
    const uint64_t lane_max = NL > 1 ? std::max<uint64_t>(std::min(bmax, bmax / NL * 3 / 2), acc_lane_add(c, kmers_ub, lane_one, lane_add_exact)) : bmax;
    c->acc_lane_budget = lane_max;
    uint64_t cap = lane_max / n_pages + lane_max / n_pages / 4 + 1024;
    cap = (cap + (1u << RB_LOG) - 1) & ~(uint64_t)((1u << RB_LOG) - 1);
    if (cap > 0x7FFFF000ull) cap = 0x7FFFF000ull;
    if (!rec32 && !g.two_level) {  // the one-level 8-byte scatter addresses a lane's regions with 32-bit byte offsets
      const uint64_t lim = ((0xFFFFFFFFull / 8 / n_pages) >> RB_LOG) << RB_LOG;
      if (cap > lim) cap = lim;
      if (cap == 0) return fail(c, SHK_ERR_INVARIANT, "no room for accumulation regions");
    }
    c->acc_rec32 = rec32;
    c->acc_cap = (uint32_t)cap;
    c->acc_budget_max = bmax;
    HIPC(c, roomy ? c->acc_buf.ensure_exact((size_t)NL * n_pages * cap * (rec32 ? 4 : 8)) : c->acc_buf.ensure((size_t)NL * n_pages * cap * (rec32 ? 4 : 8)));
    if (!rec32)  // k_pages' miss queues, here rather than at the first flush
      HIPC(c, c->part2.ensure((uint64_t)n_pages * std::min<uint64_t>(cap + MISS_SLACK, MISS_PAGE_MAX) * 8));
  }
  budget = std::min(budget, c->acc_budget_max);
  SHK_TRACEF("acc_prepare: window of %llu records (max %llu), lane budget %llu, region cap %u x %llu pages x %u lanes, rec32 %d, first add %llu\n",
             (unsigned long long)budget, (unsigned long long)c->acc_budget_max, (unsigned long long)c->acc_lane_budget, c->acc_cap,
             (unsigned long long)n_pages, NL, (int)c->acc_rec32, (unsigned long long)kmers_ub);
  HIPC(c, c->acc_cur.ensure((size_t)NL * n_pages * 4 + 64));
  HIPC(c, hipMemsetAsync(c->acc_cur.p, 0, (size_t)NL * n_pages * 4, c->stream));
  c->acc_lp = c->tb.log_pages;
  c->acc_region_lanes = NL;
  c->acc_budget = budget;
  c->acc_records_ub = 0;
  c->acc_lane_ub.assign(NL, 0);
  c->acc_nd0 = nd;
  // The spill list bounds how many pages one launch of the window's page pass may cover (flush_acc: every record of
  // every page of a group could spill) — 2^28 entries (4 GiB) cut configs[4]'s ten-lane pass into 252 launches with a
  // host round trip each; with the regions allocated and 48 GiB and more still free the list may be 2^30 entries.
  size_t free_now = 0, total_now = 0;
  (void)hipMemGetInfo(&free_now, &total_now);
  // (what the list already holds counts as free for it; not where SHK_ACC_MAX_MRECORDS says that several contexts share
  // the card — eight of them each taking 16 GiB on sight of the same free memory is an out-of-memory)
  // (a list that is that long already stays in use whatever came to lie beside it since)
  const uint64_t spill_max = (c->spillA.cap >= (16ull << 30) || free_now + c->spillA.cap >= (40ull << 30)) && env_int("SHK_ACC_MAX_MRECORDS", 0) == 0 &&
                                     env_int("SHK_BIG_SPILL", 1)
                                 ? 1ull << 30 : 1ull << 28;
  c->acc_spill_cap = std::min<uint64_t>(std::max<uint64_t>(budget, kmers_ub), spill_max);
  return SHK_OK;
}

static bool first_launch_defers(shk_ctx *c, uint64_t kmers_ub, bool multi) { return count_path(c, kmers_ub, multi) == PATH_DEFER; }

// Bytes of the per-launch cursor buffer (part_meta).  One size for everybody who asks: the buffer
// must not be reallocated between k_mark_starts (which clears cursors in it) and the partition launch.
static bool xl64_feasible(const shk_ctx *c, const PartGeom &g);
static size_t cursor_buf_bytes(const shk_ctx *c, const PartGeom &g, bool multi) {
  const bool lanes = use_all_lanes(c, g, multi) || xl_feasible(c, g) || xl64_feasible(c, g);  // (the owner layouts: n_lanes · P1 words, whatever the number of segments)
  return ((size_t)(lanes ? c->n_lanes : 1) * g.P1 + g.n_pages) * 4 + 64;
}

// Room for the partition cursors of the next paged pass; *n_words = how many k_mark_starts clears.
static int prepare_cursors(shk_ctx *c, bool multi, bool defer, uint32_t *n_words) {
  const PartGeom g = part_geom(c);
  HIPC(c, c->part_meta.ensure(cursor_buf_bytes(c, g, multi)));
  if (defer && (xl_route(c, g, multi, defer) || xl64_route(c, g))) {  // one segment's cursors: [lane][super-page of the share]
    *n_words = c->n_lanes << (g.log_p1 - g.lw);
    return SHK_OK;
  }
  if (defer) {  // the page regions' cursors persist; only a level-1 pass has per-launch cursors
    *n_words = g.two_level ? g.P1 : 0;
    return SHK_OK;
  }
  const size_t lanes = use_all_lanes(c, g, multi) ? c->n_lanes : 1;
  *n_words = (uint32_t)(lanes * g.P1) + (g.two_level ? g.n_pages : 0);
  return SHK_OK;
}

// k_scatter32 is compiled in eight variants — 64-bit offsets into the accumulation buffer or not; the
// fan-out of 1024 (one partition per thread: what every table from 8 M slots up gets) as a compile-time
// constant or any fan-out at run time; one chunk lane or all of them in one pass — and asks for more
// than 64 KiB of dynamic LDS, which has to be allowed per function and device.
template <bool W, int LP, bool A>
static hipError_t scatter32_variant(shk_ctx *c, bool set_attr, uint32_t G, size_t lds, const BatchRef &b, uint32_t log_p1,
                                    uint32_t lane_filter, unsigned int *cursor, uint32_t cap, uint32_t *buf, SpillRef sp,
                                    unsigned long long *dbg, uint32_t NL) {
  if (set_attr)
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scatter32<SC32_NT, SC32_TT, W, LP, A>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC32_LDS_MAX);
  hipLaunchKernelGGL((k_scatter32<SC32_NT, SC32_TT, W, LP, A>), dim3(G), dim3(SC32_NT), lds, c->stream, b, log_p1,
                     lane_filter, cursor, cap, buf, c->d_stats, c->d_lane_bases, sp, dbg, NL, OwnerCfg{});
  return hipSuccess;
}
static int launch_scatter32(shk_ctx *c, bool wide, uint32_t G, size_t lds, const BatchRef &b, uint32_t log_p1,
                            uint32_t lane_filter, unsigned int *cursor, uint32_t cap, uint32_t *buf, SpillRef sp,
                            unsigned long long *dbg, uint32_t NL) {
  const bool all = lane_filter == 0xFFFFFFFFu, p10 = log_p1 == 10;
  auto each = [&](bool set_attr, bool w, bool p, bool a) -> hipError_t {
#define SHK_V(W, LP, A) scatter32_variant<W, LP, A>(c, set_attr, G, lds, b, log_p1, lane_filter, cursor, cap, buf, sp, dbg, NL)
    if (w) return p ? (a ? SHK_V(true, 10, true) : SHK_V(true, 10, false)) : (a ? SHK_V(true, 0, true) : SHK_V(true, 0, false));
    return p ? (a ? SHK_V(false, 10, true) : SHK_V(false, 10, false)) : (a ? SHK_V(false, 0, true) : SHK_V(false, 0, false));
#undef SHK_V
  };
  if (!c->lds_attr_scatter) {
    for (int v = 0; v < 8; ++v) HIPC(c, each(true, v & 1, v & 2, v & 4));
    c->lds_attr_scatter = true;
  }
  HIPC(c, each(false, wide, p10, all));
  return SHK_OK;
}

// k_scatter64: the 8-byte-record scatter with k_scatter32's shape (records carried in registers, sorted in LDS).
// (Two 512-thread workgroups per CU on 8 Ki-position tiles and 512 partitions — the shape that lets two workgroups'
// phases overlap without doubling the reservations per k-mer — measured 12.2 ms against 11.4-11.6 per 24 M reads of
// config 3, the level-2 pass 3 % slower on top: not kept.)
static size_t scatter64_lds(uint32_t P1) { return (size_t)(SC32_TT + P1) * 8 + (size_t)P1 * 12 + (size_t)(SC32_NT + 2) * 8 + 32; }
static bool use_scatter64(const shk_ctx *c, const PartGeom &g) {
  return !use_rec32(c, g) && g.log_p1 >= 3 && g.P1 <= (uint32_t)SC32_NT && c->cfg.k >= 18 && SC32_NT == 1024 && scatter64_lds(g.P1) <= SC32_LDS_MAX &&
         env_int("SHK_SCATTER64", 1) != 0;
}
template <int LAYOUT>
static hipError_t scatter64_variant(shk_ctx *c, bool set_attr, uint32_t G, size_t lds, const BatchRef &b, uint32_t log_p1, uint32_t lane,
                                    unsigned int *cursor, uint32_t cap, uint64_t *buf, SpillRef sp, unsigned long long *dbg,
                                    OwnerCfg own = OwnerCfg{}, uint32_t n_region_lanes = 1) {
  if (set_attr)
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scatter64<1024, 16384, LAYOUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC32_LDS_MAX);
  hipLaunchKernelGGL((k_scatter64<1024, 16384, LAYOUT>), dim3(G), dim3(1024), lds, c->stream, b, log_p1, lane, cursor, cap, buf, c->d_stats, c->d_lane_bases, sp, dbg,
                     own, n_region_lanes);
  return hipSuccess;
}
static int scatter64_attrs(shk_ctx *c, const BatchRef &b, SpillRef sp) {
  if (c->lds_attr_scatter64) return SHK_OK;
  HIPC(c, scatter64_variant<S64_LINEAR>(c, true, 0, 0, b, 0, 0, nullptr, 0, nullptr, sp, nullptr));
  HIPC(c, scatter64_variant<S64_INTERLEAVED>(c, true, 0, 0, b, 0, 0, nullptr, 0, nullptr, sp, nullptr));
  HIPC(c, scatter64_variant<S64_OWNER>(c, true, 0, 0, b, 0, 0, nullptr, 0, nullptr, sp, nullptr));
  c->lds_attr_scatter64 = true;
  return SHK_OK;
}
// interleaved: the level-1 buffer of a two-level pass (what k_part_rescatter reads tile by tile); otherwise page regions
static int launch_scatter64(shk_ctx *c, uint32_t G, const BatchRef &b, uint32_t log_p1, uint32_t lane, unsigned int *cursor, uint32_t cap,
                            uint64_t *buf, SpillRef sp, unsigned long long *dbg, bool interleaved) {
  const size_t lds = scatter64_lds(1u << log_p1);
  int rc = scatter64_attrs(c, b, sp);
  if (rc != SHK_OK) return rc;
  HIPC(c, interleaved ? scatter64_variant<S64_INTERLEAVED>(c, false, G, lds, b, log_p1, lane, cursor, cap, buf, sp, dbg)
                      : scatter64_variant<S64_LINEAR>(c, false, G, lds, b, log_p1, lane, cursor, cap, buf, sp, dbg));
  return SHK_OK;
}

// ---- the owner layout -------------------------------------------------------------------------------
struct XlPlan {
  uint32_t log_p1w, n_grp, cap1, n_seg;  // super-page bits of a share; regions per segment; records per region; segments
  uint64_t seg_recs;                     // records per segment
};
// Geometry of one launch's level-1 buffer: every (lane, partition) region gets the per-lane share of the
// launch's k-mers (blocks of 1000 reads go round the lanes) + 50 % for uneven read lengths + 25 % + slack;
// what still overflows takes the spill path.
static XlPlan xl_plan(const shk_ctx *c, const PartGeom &g, uint64_t sub_kmers_ub, bool keep_all) {
  XlPlan x{};
  const uint32_t NL = c->n_lanes;
  x.log_p1w = g.log_p1 - g.lw;
  x.n_grp = NL << x.log_p1w;
  uint64_t lane_kmers_ub = sub_kmers_ub;
  if (NL > 1) {
    const uint64_t nb = std::max<uint64_t>(c->cur_blocks, 1);
    const uint64_t per_lane_blocks = (nb + NL - 1) / NL;
    lane_kmers_ub = std::min<uint64_t>(sub_kmers_ub, sub_kmers_ub / nb * per_lane_blocks * 3 / 2 + 2 * TILE_T);
  }
  x.cap1 = (region_cap(lane_kmers_ub, g.P1, 0) + (1u << RB_LOG) - 1u) & ~((1u << RB_LOG) - 1u);
  x.n_seg = keep_all ? 1u << g.lw : 1u;
  x.seg_recs = (uint64_t)x.n_grp * x.cap1;
  return x;
}

template <int LP>
static hipError_t scatter32_own_variant(shk_ctx *c, bool set_attr, uint32_t G, size_t lds, const BatchRef &b, uint32_t log_p1,
                                        unsigned int *cursor, uint32_t cap, uint32_t *buf, SpillRef sp, uint32_t NL, OwnerCfg own) {
  if (set_attr)
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scatter32<SC32_NT, SC32_TT, false, LP, true, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC32_LDS_MAX);
  hipLaunchKernelGGL((k_scatter32<SC32_NT, SC32_TT, false, LP, true, true>), dim3(G), dim3(SC32_NT), lds, c->stream, b, log_p1,
                     0xFFFFFFFFu, cursor, cap, buf, c->d_stats, c->d_lane_bases, sp, (unsigned long long *)nullptr, NL, own);
  return hipSuccess;
}

// Level-1 scatter of b's tiles into the owner layout (c->xbuf, cursors at the start of c->part_meta).
// keep_all: every owner's records, one segment per owner, overflow to the foreign spill list; otherwise
// only this context's own records, one segment, overflow to `sp`.
static int xl_scatter(shk_ctx *c, const BatchRef &b, const PartGeom &g, const XlPlan &x, SpillRef sp, bool keep_all,
                      bool prezeroed) {
  const uint32_t NL = c->n_lanes;
  const uint64_t total = (uint64_t)x.n_seg * x.seg_recs;
  if (total * 4 > 0xFFFFFFFFull)
    return fail(c, SHK_ERR_INVARIANT, "level-1 buffer of one launch exceeds 4 GiB (%llu records)", (unsigned long long)total);
  HIPC(c, c->xbuf.ensure(total * 4));
  unsigned int *cursor = (unsigned int *)c->part_meta.p;
  const size_t n_words = (size_t)x.n_seg * x.n_grp;
  if (c->part_meta.cap < n_words * 4) return fail(c, SHK_ERR_INVARIANT, "cursor buffer too small for the owner layout");
  if (!prezeroed) HIPC(c, hipMemsetAsync(cursor, 0, n_words * 4, c->stream));
  if (!c->lds_attr_scatter_own) {
    HIPC(c, scatter32_own_variant<0>(c, true, 0, 0, b, 0, nullptr, 0, nullptr, sp, NL, OwnerCfg{}));
    HIPC(c, scatter32_own_variant<10>(c, true, 0, 0, b, 0, nullptr, 0, nullptr, sp, NL, OwnerCfg{}));
    c->lds_attr_scatter_own = true;
  }
  OwnerCfg own{};
  own.log_w = g.lw;
  own.keep = keep_all ? 0xFFFFFFFFu : c->owner_id;
  own.seg_recs = keep_all ? (uint32_t)x.seg_recs : 0u;
  const uint32_t G = std::min<uint32_t>(grid_for(b.tile_count * (TILE_T / SC32_TT), 1, (uint32_t)env_int("SHK_PART_G", 512)), c->n_cus_scatter * SHK_SC32_WGS);
  const size_t lds = scatter32_lds(g.P1);
  {
    ScopedTimer t(c, SHK_K_SCATTER, /*chain=*/prezeroed && c->chain_from_mark);
    c->chain_from_mark = false;
    if (g.log_p1 == 10)
      HIPC(c, scatter32_own_variant<10>(c, false, G, lds, b, g.log_p1, cursor, x.cap1, (uint32_t *)c->xbuf.p, sp, NL, own));
    else
      HIPC(c, scatter32_own_variant<0>(c, false, G, lds, b, g.log_p1, cursor, x.cap1, (uint32_t *)c->xbuf.p, sp, NL, own));
  }
  return SHK_OK;
}

// k_part_rescatter32 on the context's stream.  (A 16384-record tile — 64-byte runs where 2^10 pages share a super-page
// — was tried for the 2^33-slot tables: the kernel itself is slower, 662 against 613 µs per 1.7 M reads; an apparent
// gain in the exchange path was the first process on a fresh box waiting longer in its collectives.)
static uint32_t rs32_tile(uint32_t) { return (uint32_t)RS32_TILE; }
static int launch_rescatter32(shk_ctx *c, uint32_t n_src_regions, const uint32_t *src_buf, const unsigned int *src_cursor, uint32_t cap1,
                              uint32_t log_sub, uint32_t r1_bits, unsigned int *dst_cursor, uint32_t dst_cap, uint32_t *dst_buf,
                              uint32_t lane, SpillRef sp, uint64_t dst_region_base, uint64_t n_dst_total, uint32_t log_src_lane,
                              uint32_t region_hi, uint64_t dst_lane_stride) {
  const uint32_t tile = rs32_tile(log_sub), tiles_per_region = (cap1 + tile - 1) / tile;
  const size_t lds = (size_t)tile * 6 + ((size_t)12 << log_sub);  // records + 16-bit entries + three words per page
  if (lds > SC32_LDS_MAX) return fail(c, SHK_ERR_INVARIANT, "level-2 pass: %zu bytes of LDS", lds);
  if (lds > 64 * 1024 && !c->lds_attr_rescatter) {  // > 64 KiB of dynamic LDS has to be asked for
    HIPC(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_part_rescatter32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC32_LDS_MAX));
    c->lds_attr_rescatter = true;
  }
  hipLaunchKernelGGL(k_part_rescatter32, dim3(n_src_regions * tiles_per_region), dim3(RS32_NT), lds, c->stream, src_buf, src_cursor, cap1,
                     tiles_per_region, log_sub, r1_bits, 2 * c->cfg.k, dst_cursor, dst_cap, dst_buf, lane, c->d_stats, sp, dst_region_base,
                     n_dst_total, log_src_lane, region_hi, dst_lane_stride);
  HIPC(c, hipGetLastError());
  return SHK_OK;
}

// Level-2 pass over ONE owner segment (this context's share: `regions` = [lane][super-page] regions of
// `cap1` records each, fill levels in src_cursor) into the waiting (lane, page) regions.
static int xl_absorb(shk_ctx *c, const PartGeom &g, const uint32_t *src_buf, const unsigned int *src_cursor, uint32_t cap1,
                     uint32_t n_grp, SpillRef sp) {
  const uint32_t NL = c->n_lanes, n_pages = g.n_pages;
  const uint32_t log_p1w = g.log_p1 - g.lw;
  if (n_grp != NL << log_p1w) return fail(c, SHK_ERR_BAD_ARG, "segment has %u regions, this context expects %u", n_grp, NL << log_p1w);
  if (!c->acc_cur.p || c->acc_lp != g.lp || !c->acc_rec32 || c->acc_region_lanes != NL)
    return fail(c, SHK_ERR_INVARIANT, "accumulation regions not planned for the owner layout");
  const uint32_t r1_bits = 2 * c->cfg.k - g.log_p1;
  ScopedTimer t(c, SHK_K_PSCAN, /*chain=*/true);
  return launch_rescatter32(c, n_grp, src_buf, src_cursor, cap1, g.log_sub, r1_bits, (unsigned int *)c->acc_cur.p, c->acc_cap,
                            (uint32_t *)c->acc_buf.p, 0u, sp, 0ull, (uint64_t)NL * n_pages, log_p1w, c->owner_id << log_p1w,
                            (uint64_t)n_pages);
}

// One deferred counting launch on the owner-layout route: scatter (own records only), then absorb.
static int xl_count(shk_ctx *c, const BatchRef &b, const PartGeom &g, uint64_t sub_kmers_ub, SpillRef sp, bool prezeroed) {
  const XlPlan x = xl_plan(c, g, sub_kmers_ub, /*keep_all=*/false);
  int rc = xl_scatter(c, b, g, x, sp, /*keep_all=*/false, prezeroed);
  if (rc != SHK_OK) return rc;
  return xl_absorb(c, g, (const uint32_t *)c->xbuf.p, (const unsigned int *)c->part_meta.p, x.cap1, x.n_grp, sp);
}

// ---- the owner layout with 8-byte records (k > 21: the mixed key's remainder does not fit a word) -------------------
// The same segments — [owner][lane][super-page] regions of cap1 records, their fill levels beside them — with the
// canonical k-mer as the record: k_scatter64 in its owner layout at the sender (a launch per chunk lane present in the
// batch), k_part_rescatter at the receiver (a launch per lane of a segment) into the waiting (lane, page) regions that
// k_pages counts.  The exchange rounds use it with every owner's records kept (a segment each), a share's own ingest
// with its own records only (one segment: xl64_count).
static bool xl64_feasible(const shk_ctx *c, const PartGeom &g) {
  // (≤ 32 chunk lanes: a round's segments — lanes · 2^log_p1 regions of at least 1024 records — stay below the 4 GiB the scatter addresses)
  return g.two_level && use_scatter64(c, g) && g.log_sub <= 10 && c->n_lanes <= 32 && g.lpg <= MAX_LOG_PAGES &&
         env_int("SHK_DEFER", 1) != 0 && env_int("SHK_XL64", 1) != 0;
}
static int xl64_scatter(shk_ctx *c, const BatchRef &b, const PartGeom &g, const XlPlan &x, SpillRef sp, bool keep_all = true, bool prezeroed = true) {
  const uint32_t NL = c->n_lanes;
  const uint64_t total = (uint64_t)x.n_seg * x.seg_recs;
  if (total * 8 > 0xFFFFFFFFull)
    return fail(c, SHK_ERR_INVARIANT, "level-1 buffer of one launch exceeds 4 GiB (%llu 8-byte records)", (unsigned long long)total);
  HIPC(c, c->xbuf.ensure(total * 8));
  unsigned int *cursor = (unsigned int *)c->part_meta.p;
  if (c->part_meta.cap < (size_t)x.n_seg * x.n_grp * 4) return fail(c, SHK_ERR_INVARIANT, "cursor buffer too small for the owner layout");
  int rc = scatter64_attrs(c, b, sp);
  if (rc != SHK_OK) return rc;
  if (!prezeroed) HIPC(c, hipMemsetAsync(cursor, 0, (size_t)x.n_seg * x.n_grp * 4, c->stream));
  OwnerCfg own{};
  own.log_w = g.lw;
  own.keep = keep_all ? 0xFFFFFFFFu : c->owner_id;
  own.seg_recs = keep_all ? (uint32_t)x.seg_recs : 0u;
  const uint32_t G = std::min<uint32_t>(grid_for(b.tile_count, 1, (uint32_t)env_int("SHK_PART_G", 512)), c->n_cus_scatter);
  const size_t lds = scatter64_lds(g.P1);
  // a batch inside one 1000-read block is one lane's; a tile list (k_build_tiles) may hold every lane's tiles
  const uint32_t lane_lo = b.tiles ? 0u : b.lane0, lane_hi = b.tiles ? NL : b.lane0 + 1;
  for (uint32_t lane = lane_lo; lane < lane_hi; ++lane) {
    ScopedTimer t(c, SHK_K_SCATTER, /*chain=*/lane == lane_lo && c->chain_from_mark);
    c->chain_from_mark = false;
    HIPC(c, scatter64_variant<S64_OWNER>(c, false, G, lds, b, g.log_p1, lane, cursor, x.cap1, (uint64_t *)c->xbuf.p, sp, nullptr, own, NL));
  }
  return SHK_OK;
}
// Level-2 pass over ONE owner segment of 8-byte records into the waiting (lane, page) regions.
static int xl64_absorb(shk_ctx *c, const PartGeom &g, const uint64_t *src_buf, const unsigned int *src_cursor, uint32_t cap1, uint32_t n_grp,
                       SpillRef sp) {
  const uint32_t NL = c->n_lanes, n_pages = g.n_pages;
  const uint32_t log_p1w = g.log_p1 - g.lw, S1w = 1u << log_p1w;
  if (n_grp != NL << log_p1w) return fail(c, SHK_ERR_BAD_ARG, "segment has %u regions, this context expects %u", n_grp, NL << log_p1w);
  if (!c->acc_cur.p || c->acc_lp != g.lp || c->acc_rec32 || c->acc_region_lanes != NL)
    return fail(c, SHK_ERR_INVARIANT, "accumulation regions not planned for 8-byte records");
  const uint32_t S = 1u << g.log_sub, tiles_per_region = (cap1 + RS_TILE - 1) / RS_TILE;
  const size_t lds_rs = (size_t)RS_TILE * 8 + (((size_t)RS_TILE + S) * 2 + 15) / 16 * 16 + (size_t)S * 12;
  RescatterList ls{};  // (not a list: the owner bits of a k-mer's page are this share's, the rest is the page)
  ls.owner_bits = g.lw;
  ls.owner_id = c->tb.owner_id;
  for (uint32_t lane = 0; lane < NL; ++lane) {
    ScopedTimer t(c, SHK_K_PSCAN, /*chain=*/true);
    hipLaunchKernelGGL(k_part_rescatter, dim3(S1w * tiles_per_region), dim3(RS_NT), lds_rs, c->stream, src_buf + (size_t)lane * S1w * cap1,
                       src_cursor + (size_t)lane * S1w, cap1, tiles_per_region, g.lp, g.log_sub, 2 * c->cfg.k,
                       (unsigned int *)c->acc_cur.p + (size_t)lane * n_pages, c->acc_cap,
                       (uint64_t *)c->acc_buf.p + (size_t)lane * n_pages * c->acc_cap, lane, c->d_stats, sp, 0u, ls);
  }
  HIPC(c, hipGetLastError());
  return SHK_OK;
}
// One deferred counting launch of a share's OWN ingest at k > 21: scatter (own records only, one segment), then absorb.
static int xl64_count(shk_ctx *c, const BatchRef &b, const PartGeom &g, uint64_t sub_kmers_ub, SpillRef sp, bool prezeroed) {
  const XlPlan x = xl_plan(c, g, sub_kmers_ub, /*keep_all=*/false);
  int rc = xl64_scatter(c, b, g, x, sp, /*keep_all=*/false, prezeroed);
  if (rc != SHK_OK) return rc;
  return xl64_absorb(c, g, (const uint64_t *)c->xbuf.p, (const unsigned int *)c->part_meta.p, x.cap1, x.n_grp, sp);
}
// the record of an exchange segment at this geometry: 4 bytes (the 4-byte owner layout), 8 (the one above), 0: neither
static uint32_t xchg_rec_bytes(const shk_ctx *c, const PartGeom &g) {
  if (xl_feasible(c, g) && count_path(c, 0) == PATH_DEFER) return 4;
  return xl64_feasible(c, g) ? 8u : 0u;
}

// ---- exchange rounds between owner shares (shk_xchg_*) ------------------------------------------------
// The exchange layout depends only on (layout_bases, n_lanes, geometry), never on the batch at hand: every
// rank of a round must come out with the same segment size.
static XlPlan xchg_plan(const shk_ctx *c, const PartGeom &g, uint64_t layout_bases) {
  XlPlan x{};
  const uint32_t NL = c->n_lanes;
  x.log_p1w = g.log_p1 - g.lw;
  x.n_grp = NL << x.log_p1w;
  const uint64_t B = (layout_bases + TILE_T - 1) / TILE_T * TILE_T;
  // What crosses the links is the segments as they are sized, full or not, so they are sized tightly: a lane's
  // share of a round is its blocks' (1000-read blocks go round the lanes: + 10 % and two blocks' worth of slack
  // for uneven read lengths), a region's share of that the mean + 5 % + 8 σ of a Poisson count + a block.  What
  // still overflows goes to the foreign spill list — exact, just slower.
  const uint64_t lane_kmers_ub = NL > 1 ? std::min<uint64_t>(B, B / NL * 11 / 10 + 262144) : B;
  const double mean = (double)lane_kmers_ub / (double)g.P1;
  const uint64_t cap = (uint64_t)(mean * 1.05 + 8.0 * std::sqrt(mean)) + 1024;
  x.cap1 = (uint32_t)std::min<uint64_t>((cap + (1u << RB_LOG) - 1u) & ~(uint64_t)((1u << RB_LOG) - 1u), 0x7FFFF000ull);
  x.n_seg = 1u << g.lw;
  x.seg_recs = (uint64_t)x.n_grp * x.cap1;
  return x;
}
static int xchg_check(shk_ctx *c, const PartGeom &g) {
  if (!c->is_share()) return fail(c, SHK_ERR_STATE, "not an owner share (shk_config.n_owners = 0: say 1 for a share that is the whole key space)");
  if (!xchg_rec_bytes(c, g))
    return fail(c, SHK_ERR_STATE,
                "the owner exchange needs 4-byte records (2k - %u ≤ 32) or k ≥ 18 on a two-level table, and ≤ 128 chunk lanes at this "
                "table geometry; take the wide round or merge the tables at finalize instead", g.log_p1);
  return SHK_OK;
}
static int xchg_prepare_cursors(shk_ctx *c, uint32_t *n_words) {
  const PartGeom g = part_geom(c);
  int rc = xchg_check(c, g);
  if (rc != SHK_OK) return rc;
  HIPC(c, c->part_meta.ensure(cursor_buf_bytes(c, g, true)));
  *n_words = c->n_lanes * g.P1;  // every segment's cursors
  return SHK_OK;
}
static SpillRef xspill_ref(shk_ctx *c) {
  SpillRef sp = spill_ref(c->xspill, c->xspill_cap);
  sp.count = &c->d_stats->scratch[0];
  return sp;
}
// What an exchange scatter left in the statistics (h_stats read back behind it).
static int xchg_scatter_outcome(shk_ctx *c, uint64_t *n_foreign) {
  if (c->h_stats->bad != ~0ull) {
    c->poisoned = true;
    c->poison_code = SHK_ERR_INVALID_CHAR;
    return fail(c, SHK_ERR_INVALID_CHAR, "Invalid character '%s' in sequence. Only ACGTN allowed.", shk::byte_as_char((uint8_t)(c->h_stats->bad & 0xFF)).c_str());
  }
  if (c->h_stats->scratch[0] > c->xspill_cap) return fail(c, SHK_ERR_INVARIANT, "foreign spill list overflow");
  *n_foreign = c->h_stats->scratch[0];
  return SHK_OK;
}

static int xchg_scatter_launch(shk_ctx *c, const BatchRef &b, uint64_t kmers_ub, XchgOut *xo) {
  const PartGeom g = part_geom(c);
  int rc = xchg_check(c, g);
  if (rc != SHK_OK) return rc;
  const uint64_t lb = xo->layout_bases ? xo->layout_bases : std::max<uint64_t>(kmers_ub, TILE_T);
  if (b.n_bases > lb) return fail(c, SHK_ERR_BAD_ARG, "batch larger than the exchange layout it is to use");
  const XlPlan x = xchg_plan(c, g, lb);
  // the foreign spill list: whatever it holds already stays (the caller drains it between rounds)
  const uint64_t pending = c->h_stats->scratch[0];
  const uint64_t want_cap = pending + std::max<uint64_t>(kmers_ub, 1);
  if (want_cap > c->xspill_cap) {
    DevBuf nb;
    HIPC(c, nb.ensure(want_cap * 16));
    const uint64_t ncap = nb.cap / 16;
    if (pending) {  // (rare: skewed input two rounds in a row without a drain)
      SpillRef o = spill_ref(c->xspill, c->xspill_cap), n = spill_ref(nb, ncap);
      HIPC(c, hipMemcpyAsync(n.keys, o.keys, pending * 8, hipMemcpyDeviceToDevice, c->stream));
      HIPC(c, hipMemcpyAsync(n.lanes, o.lanes, pending * 4, hipMemcpyDeviceToDevice, c->stream));
      HIPC(c, hipMemcpyAsync(n.counts, o.counts, pending * 4, hipMemcpyDeviceToDevice, c->stream));
      HIPC(c, hipStreamSynchronize(c->stream));
    }
    c->xspill.release();
    c->xspill = nb;
    c->xspill_cap = ncap;
  }
  const uint32_t rec_bytes = xchg_rec_bytes(c, g);
  if (kmers_ub) {
    rc = rec_bytes == 8 ? xl64_scatter(c, b, g, x, xspill_ref(c)) : xl_scatter(c, b, g, x, xspill_ref(c), /*keep_all=*/true, /*prezeroed=*/true);
    if (rc != SHK_OK) return rc;
  } else {  // an empty batch still takes part in the round: all-zero cursors
    HIPC(c, c->xbuf.ensure((uint64_t)x.n_seg * x.seg_recs * rec_bytes));
    HIPC(c, c->part_meta.ensure(cursor_buf_bytes(c, g, true)));
    HIPC(c, hipMemsetAsync(c->part_meta.p, 0, (size_t)x.n_seg * x.n_grp * 4, c->stream));
  }
  xo->d_records = c->xbuf.p;
  xo->d_cursors = c->part_meta.p;
  xo->lay.n_owners = c->n_owners;
  xo->lay.n_lanes = c->n_lanes;
  xo->lay.log_p1 = g.log_p1;
  xo->lay.regions = x.n_grp;
  xo->lay.region_cap = x.cap1;
  xo->lay.record_bytes = rec_bytes;
  xo->lay.segment_records = x.seg_recs;
  if (xo->two_calls) {  // the outcome is fetched by shk_xchg_scatter_end; what is launched in between runs behind the scatter
    if (!c->xs_ev) HIPC(c, hipEventCreateWithFlags(&c->xs_ev, hipEventDisableTiming));
    HIPC(c, hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(DevStats), hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipEventRecord(c->xs_ev, c->stream));
    return SHK_OK;
  }
  rc = read_stats(c);  // (synchronises: the segments are complete when this call returns)
  if (rc != SHK_OK) return rc;
  return xchg_scatter_outcome(c, &xo->n_foreign);
}

// The wide exchange round: validate (k_scan), count the batch's k-mers by owner, lay the owners' segments out back to
// back, write k-mers and lanes.  Synchronous: the counts come back to the host between the two passes.
static int xw_scatter_launch(shk_ctx *c, const BatchRef &b, uint64_t kmers_ub, XchgOut *xo) {
  const uint32_t W = c->n_owners;
  for (uint32_t o = 0; o < W; ++o) xo->counts[o] = 0;
  xo->d_kmers = xo->d_lanes = nullptr;
  if (!kmers_ub) return SHK_OK;
  HIPC(c, c->xw_kmers.ensure(kmers_ub * 8 + 64));
  HIPC(c, c->xw_lanes.ensure(kmers_ub * 4 + 64));
  HIPC(c, c->xw_count.ensure(64 * 8));
  unsigned long long *d_cnt = (unsigned long long *)c->xw_count.p;
  HIPC(c, hipMemsetAsync(d_cnt, 0, 64 * 8, c->stream));
  {  // validate + count bases first (encoding.rs:353-356, 374-376); k_xw_scatter tests stats->bad
    ScopedTimer t(c, SHK_K_SCAN);
    hipLaunchKernelGGL(k_scan, dim3(grid_for(b.tile_count, 1, 4096)), dim3(WG), 0, c->stream, b, c->d_stats, c->d_lane_bases);
  }
  const uint32_t G = grid_for(b.tile_count, 1, 256 * 8);
  {
    ScopedTimer t(c, SHK_K_PCOUNT);
    hipLaunchKernelGGL(k_xw_scatter<false>, dim3(G), dim3(WG), 0, c->stream, b, 2 * c->cfg.k, c->owner_bits, c->d_stats, d_cnt,
                       (uint64_t *)nullptr, (uint32_t *)nullptr);
  }
  unsigned long long h_cnt[64];
  HIPC(c, hipMemcpyAsync(h_cnt, d_cnt, (size_t)W * 8, hipMemcpyDeviceToHost, c->stream));
  int rc = read_stats(c);  // (synchronises)
  if (rc != SHK_OK) return rc;
  if (c->h_stats->bad != ~0ull) {
    c->poisoned = true;
    c->poison_code = SHK_ERR_INVALID_CHAR;
    return fail(c, SHK_ERR_INVALID_CHAR, "Invalid character '%s' in sequence. Only ACGTN allowed.", shk::byte_as_char((uint8_t)(c->h_stats->bad & 0xFF)).c_str());
  }
  unsigned long long base[64], at = 0;
  for (uint32_t o = 0; o < W; ++o) {
    base[o] = at;
    at += h_cnt[o];
    xo->counts[o] = h_cnt[o];
  }
  if (at > kmers_ub) return fail(c, SHK_ERR_INVARIANT, "more k-mers than end positions in an exchange batch");
  HIPC(c, hipMemcpyAsync(d_cnt, base, (size_t)W * 8, hipMemcpyHostToDevice, c->stream));
  {
    ScopedTimer t(c, SHK_K_SCATTER);
    hipLaunchKernelGGL(k_xw_scatter<true>, dim3(G), dim3(WG), 0, c->stream, b, 2 * c->cfg.k, c->owner_bits, c->d_stats, d_cnt,
                       (uint64_t *)c->xw_kmers.p, (uint32_t *)c->xw_lanes.p);
  }
  HIPC(c, hipStreamSynchronize(c->stream));  // (`base` lives on this stack; and the arrays are complete when the call returns)
  xo->d_kmers = c->xw_kmers.p;
  xo->d_lanes = c->xw_lanes.p;
  return SHK_OK;
}

// A FRESH page pass over every page and lane may leave the histogram behind (k_pages32<true, true>): room for its rows.
// (Not for a context that reports a slot range of its table — own_set, the merge-at-finalize counter — or whose
// histogram was asked to stay with the scan, SHK_FUSED_HIST=0.)
static bool fused_hist_wanted(const shk_ctx *c) {
  // (one lane on a large table: the scan's 4 B per slot cost less than the pass's extra work — measured on a 30 Mb
  // genome, 8192 pages: pages + 0.06 ms for a scan of 0.077; from two lanes on, and on small tables, the pass wins)
  if (c->n_lanes == 1 && c->tb.log_pages > 11 && env_int("SHK_FUSED_HIST", 1) != 2) return false;
  // (the rows: 2 KiB per page and column — many lanes on a large table would be tens of GB: the scan then)
  const uint64_t rows_bytes = ((uint64_t)1 << c->tb.log_pages) * std::max<uint32_t>(c->cfg.chunks, 1) * FH_BINS * 4;
  if (rows_bytes > (4ull << 30)) return false;
  return !c->own_set && !c->zero_count_keys && !c->fused_off && env_int("SHK_FUSED_HIST", 1) != 0;
}
static int fused_hist_prepare(shk_ctx *c, uint32_t n_pages, FusedHist *fh) {
  const uint32_t n_cols = c->cfg.chunks;
  HIPC(c, c->fh_partial.ensure((size_t)n_pages * std::max<uint32_t>(n_cols, 1) * FH_BINS * 4));
  HIPC(c, c->fh_tot.ensure((size_t)n_pages * 4 * 8));
  fh->partial = (uint32_t *)c->fh_partial.p;
  fh->ptot = (unsigned long long *)c->fh_tot.p;
  fh->hist = c->d_hist;
  fh->histo_max = c->cfg.histo_max;
  fh->n_cols = n_cols;
  return SHK_OK;
}

static int xl64_count(shk_ctx *c, const BatchRef &b, const PartGeom &g, uint64_t sub_kmers_ub, SpillRef sp, bool prezeroed);
static int paged_count(shk_ctx *c, const BatchRef &b, uint64_t sub_kmers_ub, SpillRef sp, bool prezeroed, bool defer) {
  const PartGeom pg = part_geom(c);
  if (xl_route(c, pg, b.tiles != nullptr, defer)) return xl_count(c, b, pg, sub_kmers_ub, sp, prezeroed);
  if (defer && xl64_route(c, pg)) return xl64_count(c, b, pg, sub_kmers_ub, sp, prezeroed);
  if (pg.lw) return fail(c, SHK_ERR_INVARIANT, "an owner share has no paged path besides the owner layout");
  const uint32_t lp = pg.lp, n_pages = pg.n_pages, log_p1 = pg.log_p1, log_sub = pg.log_sub, P1 = pg.P1;
  const bool two_level = pg.two_level;
  const uint32_t g_cap = (uint32_t)env_int("SHK_PART_G", 512);
  const uint32_t G = grid_for(b.tile_count, 1, g_cap);
  if (two_level && (1u << log_sub) > (uint32_t)MAX_PARTS)
    return fail(c, SHK_ERR_BAD_ARG, "table too large for the two-level partition");
  // Every region is filled by per-tile reservations (one returning atomic per non-empty
  // (tile, region)); a region that still overflows (skewed input: one k-mer making up a large
  // share of the batch) sends the excess through the spill list — exact either way.
  const uint32_t r1_bits = 2 * c->cfg.k >= log_p1 ? 2 * c->cfg.k - log_p1 : 0;
  const bool rec32 = use_rec32(c, pg);
  const bool multi = b.tiles != nullptr;
  bool all_lanes = use_all_lanes(c, pg, multi) && !defer;  // (deferred: always (lane, page) regions, see below)
  const uint32_t NL = c->n_lanes;
  // per-lane share of the batch's k-mers in ALL-LANES mode: blocks go round the lanes, so a lane
  // holds at most ceil(blocks / lanes) of them; half as much again for uneven read lengths
  // (what still overflows a region takes the spill path)
  uint64_t lane_kmers_ub = sub_kmers_ub;
  if (all_lanes) {
    const uint64_t nb = std::max<uint64_t>(c->cur_blocks, 1);
    const uint64_t per_lane_blocks = (nb + NL - 1) / NL;
    lane_kmers_ub = std::min<uint64_t>(sub_kmers_ub, sub_kmers_ub / nb * per_lane_blocks * 3 / 2 + 2 * TILE_T);
  }
  // (4-byte-record regions are block-interleaved, rec_slot: whole blocks of 2^RB_LOG records)
  const uint64_t pads = rec32 ? 0 : b.tile_count;  // (only 8-B record runs are padded, once per (tile, region) at most)
  // (the level-1 regions of k_scatter64 are interleaved in blocks of RS_TILE records: whole blocks)
  const bool il64 = !rec32 && two_level && use_scatter64(c, pg) && env_int("SHK_S64_INTERLEAVE", 1) != 0;
  const uint32_t cap1_unit = il64 ? (uint32_t)RS_TILE : 1u << RB_LOG;
  uint32_t cap1 = (region_cap(lane_kmers_ub, P1, pads) + cap1_unit - 1u) & ~(cap1_unit - 1u);
  if (all_lanes && (uint64_t)NL * P1 * cap1 * 4 > 0xFFFFFFFFull) {  // 32-bit byte offsets: fall back to a pass per lane
    all_lanes = false;
    cap1 = (region_cap(sub_kmers_ub, P1, pads) + (1u << RB_LOG) - 1u) & ~((1u << RB_LOG) - 1u);
  }
  const uint32_t region_lanes = all_lanes ? NL : 1;
  const uint32_t rs_tile = rec32 ? rs32_tile(log_sub) : (uint32_t)RS_TILE;
  const uint32_t tiles_per_region = (cap1 + rs_tile - 1) / rs_tile;
  const uint32_t cap_pg =
      two_level ? (region_cap(sub_kmers_ub, n_pages, rec32 ? 0 : tiles_per_region) + (1u << RB_LOG) - 1u) & ~((1u << RB_LOG) - 1u) : cap1;
  DevBuf &buf_pg = defer ? c->acc_buf : (two_level ? c->part3 : c->part);  // what k_pages reads
  if (defer && (c->acc_lp != lp || !c->acc_cur.p)) return fail(c, SHK_ERR_INVARIANT, "accumulation regions not planned");
  if (!(defer && !two_level) && (uint64_t)region_lanes * P1 * cap1 * (rec32 ? 4 : 8) > 0xFFFFFFFFull)  // 32-bit byte offsets in the scatter
    return fail(c, SHK_ERR_INVARIANT, "partition buffer of one launch exceeds 4 GiB");
  if (!(defer && !two_level)) HIPC(c, c->part.ensure((uint64_t)region_lanes * P1 * cap1 * (rec32 ? 4 : 8)));
  if (two_level && !defer) HIPC(c, c->part3.ensure((uint64_t)n_pages * cap_pg * (rec32 ? 4 : 8)));
  if (!rec32 && !defer)
    HIPC(c, c->part2.ensure((uint64_t)n_pages * std::min<uint64_t>((uint64_t)cap_pg + MISS_SLACK, MISS_PAGE_MAX) * 8));  // k_pages miss queues
  if (defer && rec32 != c->acc_rec32) return fail(c, SHK_ERR_INVARIANT, "accumulation regions planned for the other record size");
  HIPC(c, c->part_meta.ensure(cursor_buf_bytes(c, pg, multi)));
  unsigned int *cursor1 = (unsigned int *)c->part_meta.p;
  unsigned int *cursor_pg = two_level ? cursor1 + P1 : cursor1;
  unsigned long long *dbg = nullptr;
#ifdef SHK_PHASE_TIMING
  HIPC(c, c->misc.ensure((size_t)G * 64));
  dbg = (unsigned long long *)c->misc.p;
#endif
  const size_t lds_sorted = (size_t)sort_region_bytes(P1) + (size_t)PACK_WORDS * 8 + (size_t)P1 * 12 + 32;  // (+ the walk's 8 spare counters when P1 < 8)
  const uint32_t S = 1u << log_sub;
  const size_t lds_rs = (size_t)RS_TILE * 8 + (((size_t)RS_TILE + S) * 2 + 15) / 16 * 16 + (size_t)S * 12;
  const size_t lds_s32 = scatter32_lds(P1);
  const bool lds32 = use_scatter32(c, pg);
  const uint32_t lane_lo = multi ? 0 : b.lane0, lane_hi = multi ? c->n_lanes : b.lane0 + 1;
  // deferred: page regions and cursors are the accumulation ones and persist; only a level-1 pass
  // has cursors of its own
  const size_t n_cursor_words = defer ? (two_level ? P1 : 0) : (size_t)region_lanes * P1 + (two_level ? n_pages : 0);
  const bool one_pass = all_lanes || (defer && !two_level && rec32);  // every lane's tiles in a single scatter launch
  const uint32_t acc_wide = (uint64_t)NL * n_pages * c->acc_cap * 4 > 0xFFFFFFFFull;
  // one pass per chunk lane — or a single pass for all of them (`lane` = ~0 below)
  for (uint32_t lane = lane_lo; lane < (one_pass ? lane_lo + 1 : lane_hi); ++lane) {
    if (n_cursor_words && !(prezeroed && lane == lane_lo))  // the first pass's cursors were cleared by k_mark_starts
      HIPC(c, hipMemsetAsync(cursor1, 0, n_cursor_words * 4, c->stream));
    {
      ScopedTimer t(c, SHK_K_SCATTER, /*chain=*/prezeroed && lane == lane_lo && c->chain_from_mark);
      c->chain_from_mark = false;
      if (rec32 && lds32) {
        int rcl;
        if (defer && !two_level)  // straight into the accumulation regions, (lane, page) layout
          rcl = launch_scatter32(c, acc_wide != 0, std::min<uint32_t>(G, c->n_cus_scatter * SHK_SC32_WGS), lds_s32, b, log_p1, 0xFFFFFFFFu,
                                 (unsigned int *)c->acc_cur.p, c->acc_cap, (uint32_t *)c->acc_buf.p, sp, dbg, NL);
        else
          rcl = launch_scatter32(c, false, std::min<uint32_t>(G, c->n_cus_scatter * SHK_SC32_WGS), lds_s32, b, log_p1,
                                 all_lanes ? 0xFFFFFFFFu : lane, cursor1, cap1, (uint32_t *)c->part.p, sp, dbg, NL);
        if (rcl != SHK_OK) return rcl;
      } else if (rec32)
        hipLaunchKernelGGL((k_part_scatter_sorted<SC_NT, true>), dim3(G), dim3(SC_NT), lds_sorted, c->stream,
                           b, log_p1, lane, cursor1, cap1, c->part.p, c->d_stats, c->d_lane_bases, sp, dbg);
      else if (use_scatter64(c, pg)) {  // 8-byte records, k_scatter32's machine (deferred one-level: straight into this lane's accumulation regions)
        const bool acc1 = defer && !two_level;
        int rcl = launch_scatter64(c, std::min<uint32_t>(G, c->n_cus_scatter), b, log_p1, lane, acc1 ? (unsigned int *)c->acc_cur.p + (size_t)lane * n_pages : cursor1,
                                   acc1 ? c->acc_cap : cap1, acc1 ? (uint64_t *)c->acc_buf.p + (size_t)lane * n_pages * c->acc_cap : (uint64_t *)c->part.p, sp, dbg,
                                   il64);
        if (rcl != SHK_OK) return rcl;
      } else if (defer && !two_level)  // straight into this lane's accumulation regions (8-byte records)
        hipLaunchKernelGGL((k_part_scatter_sorted<SC_NT, false>), dim3(G), dim3(SC_NT), lds_sorted, c->stream,
                           b, log_p1, lane, (unsigned int *)c->acc_cur.p + (size_t)lane * n_pages, c->acc_cap,
                           (void *)((uint64_t *)c->acc_buf.p + (size_t)lane * n_pages * c->acc_cap), c->d_stats,
                           c->d_lane_bases, sp, dbg);
      else
        hipLaunchKernelGGL((k_part_scatter_sorted<SC_NT, false>), dim3(G), dim3(SC_NT), lds_sorted, c->stream,
                           b, log_p1, lane, cursor1, cap1, c->part.p, c->d_stats, c->d_lane_bases, sp, dbg);
    }
    if (two_level) {
      ScopedTimer t(c, SHK_K_PSCAN, /*chain=*/true);  // timer slot reused: the level-2 re-scatter
      if (rec32 && defer) {  // append to this lane's accumulation regions
        int rcl = launch_rescatter32(c, P1, (const uint32_t *)c->part.p, (const unsigned int *)cursor1, cap1, log_sub, r1_bits,
                                     (unsigned int *)c->acc_cur.p, c->acc_cap, (uint32_t *)c->acc_buf.p, lane, sp, (uint64_t)lane * n_pages,
                                     (uint64_t)NL * n_pages, 31u, 0u, 0ull);
        if (rcl != SHK_OK) return rcl;
      } else if (rec32) {
        int rcl = launch_rescatter32(c, P1, (const uint32_t *)c->part.p, (const unsigned int *)cursor1, cap1, log_sub, r1_bits, cursor_pg,
                                     cap_pg, (uint32_t *)buf_pg.p, lane, sp, 0ull, (uint64_t)n_pages, 31u, 0u, 0ull);
        if (rcl != SHK_OK) return rcl;
      }
      else if (defer)  // append to this lane's accumulation regions (8-byte records)
        hipLaunchKernelGGL(k_part_rescatter, dim3(P1 * tiles_per_region), dim3(RS_NT), lds_rs, c->stream,
                           (const uint64_t *)c->part.p, (const unsigned int *)cursor1, cap1, tiles_per_region, lp,
                           log_sub, 2 * c->cfg.k, (unsigned int *)c->acc_cur.p + (size_t)lane * n_pages, c->acc_cap,
                           (uint64_t *)c->acc_buf.p + (size_t)lane * n_pages * c->acc_cap, lane, c->d_stats, sp, il64 ? 1u : 0u);
      else
        hipLaunchKernelGGL(k_part_rescatter, dim3(P1 * tiles_per_region), dim3(RS_NT), lds_rs, c->stream,
                           (const uint64_t *)c->part.p, (const unsigned int *)cursor1, cap1, tiles_per_region, lp,
                           log_sub, 2 * c->cfg.k, cursor_pg, cap_pg, (uint64_t *)buf_pg.p, lane, c->d_stats, sp, il64 ? 1u : 0u);
    }
    if (!defer) {
      const uint32_t l_lo = all_lanes ? 0u : lane, l_hi = all_lanes ? NL : lane + 1;
      const bool fresh = c->tb_stale && rec32 && l_lo == 0 && l_hi == c->n_lanes && !env_int("SHK_NO_FRESH", 0);
      if (!fresh) {
        int rcf = tb_fresh(c);
        if (rcf != SHK_OK) return rcf;
      }
      FusedHist fh{};
      const bool fuse = rec32 && fresh && fused_hist_wanted(c);
      if (fuse) {
        int rch = fused_hist_prepare(c, n_pages, &fh);
        if (rch != SHK_OK) return rch;
      }
      ScopedTimer t(c, SHK_K_PAGES, /*chain=*/true);
      if (rec32 && fresh) {
        if (fuse && l_hi - l_lo == 1)
          hipLaunchKernelGGL((k_pages32<true, 2>), dim3(n_pages), dim3(PG_WG), 0, c->stream, c->tb, l_lo, l_hi,
                             all_lanes ? n_pages : 0u, (uint32_t)region_lanes * n_pages,
                             (const unsigned int *)cursor_pg, cap_pg, (const uint32_t *)buf_pg.p, c->d_stats, sp, 0u, fh);
        else if (fuse)
          hipLaunchKernelGGL((k_pages32<true, 1>), dim3(n_pages), dim3(PG_WG), 0, c->stream, c->tb, l_lo, l_hi,
                             all_lanes ? n_pages : 0u, (uint32_t)region_lanes * n_pages,
                             (const unsigned int *)cursor_pg, cap_pg, (const uint32_t *)buf_pg.p, c->d_stats, sp, 0u, fh);
        else
          hipLaunchKernelGGL((k_pages32<true, 0>), dim3(n_pages), dim3(PG_WG), 0, c->stream, c->tb, l_lo, l_hi,
                             all_lanes ? n_pages : 0u, (uint32_t)region_lanes * n_pages,
                             (const unsigned int *)cursor_pg, cap_pg, (const uint32_t *)buf_pg.p, c->d_stats, sp, 0u, fh);
        c->tb_stale = false;
        c->fused_valid = fuse;
        c->fused_pages = n_pages;
      } else if (rec32)
        hipLaunchKernelGGL((k_pages32<false, 0>), dim3(n_pages), dim3(PG_WG), 0, c->stream, c->tb, l_lo, l_hi,
                           all_lanes ? n_pages : 0u, (uint32_t)region_lanes * n_pages,
                           (const unsigned int *)cursor_pg, cap_pg, (const uint32_t *)buf_pg.p, c->d_stats, sp);
      else
        hipLaunchKernelGGL((k_pages<false, false>), dim3(n_pages), dim3(PG_WG), 0, c->stream, c->tb, lane, lane + 1, 0u,
                           (const unsigned int *)cursor_pg, cap_pg, (const uint64_t *)buf_pg.p,
                           (uint64_t *)c->part2.p, c->d_stats, sp);
    }
#ifdef SHK_PHASE_TIMING
    {
      std::vector<unsigned long long> h((size_t)G * 8);
      HIPC(c, hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
      double ph[8] = {0};
      for (uint32_t g = 0; g < G; ++g)
        for (int i = 0; i < 8; ++i) ph[i] += (double)h[(size_t)g * 8 + i];
      double tot = 0;
      for (int i = 0; i < 8; ++i) tot += ph[i];
      fprintf(stderr, "[phase cycles/WG] stage %.0f pack %.0f walk %.0f scan %.0f place %.0f write %.0f | share:",
              ph[0] / G, ph[1] / G, ph[2] / G, ph[3] / G, ph[4] / G, ph[5] / G);
      for (int i = 0; i < 8; ++i) fprintf(stderr, " %.1f%%", 100 * ph[i] / tot);
      fprintf(stderr, "\n");
    }
#endif
  }
  return SHK_OK;
}

// Records a launch of ≤ kmers_ub k-mer occurrences adds to the waiting regions: all of them, or — an
// owner share keeps the k-mers of one owner in W, owners being hash bits — about 1/W of them (+ 1/8 +
// slack; a region that overflows all the same spills, exactly).
static uint64_t acc_records_est(const shk_ctx *c, uint64_t kmers_ub) {
  // (kmers_ub counts one record per base of the launch's tiles; a batch of reads of ≥ k bases has k-1 fewer per
  // read — cur_kmer_ratio, from the batch's read count.  An estimate that is too low costs nothing but time:
  // a region that overflows spills, exactly.)
  const uint64_t est = (uint64_t)((double)kmers_ub * c->cur_kmer_ratio) + 65536;
  if (!c->owner_bits) return std::min(kmers_ub, est);
  const uint64_t share = std::min(kmers_ub, est) >> c->owner_bits;
  return std::min<uint64_t>(kmers_ub, share + share / 8 + 65536);
}

static int count_tiles(shk_ctx *c, const BatchRef &b, uint64_t sub_kmers_ub, bool prezeroed) {
  CountPath path = count_path(c, sub_kmers_ub, b.tiles != nullptr);
  if (path == PATH_DEFER) {  // (the first launch of an ingest was planned before k_mark_starts)
    int rc = acc_prepare(c, acc_records_est(c, sub_kmers_ub), b.tiles ? -1 : (int64_t)b.lane0);
    if (rc != SHK_OK) return rc;
    path = count_path(c, sub_kmers_ub, b.tiles != nullptr);  // a flush may have grown the table
    if (path == PATH_DEFER && !c->acc_cur.p) return fail(c, SHK_ERR_INVARIANT, "accumulation regions missing");
  }
  if (path != PATH_DEFER && c->acc_active) {  // the other paths work on the table itself
    int rc = settle(c);
    if (rc != SHK_OK) return rc;
  }
  // (deferred: the partition launches of a window and its page pass append to one spill list)
  const uint64_t spill_cap = path == PATH_DEFER ? std::max<uint64_t>(c->acc_spill_cap, sub_kmers_ub) : sub_kmers_ub;
  if (path == PATH_DEFER) c->acc_spill_cap = spill_cap;
  HIPC(c, c->spillA.ensure(spill_cap * 16));
  SpillRef sp = spill_ref(c->spillA, spill_cap);
  if (!prezeroed)
    HIPC(c, hipMemsetAsync(&c->d_stats->spill_count, 0, sizeof(unsigned long long), c->stream));
  if (path != PATH_DIRECT) {
    int rc = paged_count(c, b, sub_kmers_ub, sp, prezeroed, path == PATH_DEFER);
    if (rc != SHK_OK) return rc;
    if (path == PATH_DEFER) {
      c->acc_active = true;
      acc_book(c, acc_records_est(c, sub_kmers_ub), b.tiles ? -1 : (int64_t)b.lane0);
    }
  } else {
    {  // validate + count bases first (encoding.rs:353-356, 374-376); k_direct tests stats->bad
      ScopedTimer t(c, SHK_K_SCAN);
      hipLaunchKernelGGL(k_scan, dim3(grid_for(b.tile_count, 1, 4096)), dim3(WG), 0, c->stream, b,
                         c->d_stats, c->d_lane_bases);
    }
    {
      int rcf = tb_fresh(c);
      if (rcf != SHK_OK) return rcf;
    }
    ScopedTimer t(c, SHK_K_DIRECT);
    hipLaunchKernelGGL(k_direct, dim3(grid_for(b.tile_count, 1, 256 * 8)), dim3(WG), 0, c->stream, b,
                       c->tb, c->d_stats, sp);
  }
  // Nothing is waited for here: the host looks at the launch's outcome (invalid byte, spilled
  // records, load factor) in settle(), at the latest before the next launch or at finalize.
  c->unsettled = true;
  c->unsettled_spill_cap = spill_cap;
  return SHK_OK;
}

// Outcome of the last counting launch, h_stats already read back and synchronised.
// An owned range given as a share (shk_set_owner_share) follows the table when it grows.
static void own_resolve(shk_ctx *c) {
  if (!c->own_share_n) return;
  const uint64_t per = (1ull << c->tb.log_pages) / c->own_share_n;
  c->own_p0 = per * c->own_share_id;
  c->own_p1 = per * (c->own_share_id + 1);
}

static int settle_checked(shk_ctx *c) {
  c->unsettled = false;
  if (c->h_stats->bad != ~0ull) {
    // identical text to encoding.rs:353-356
    c->poisoned = true;
    c->poison_code = SHK_ERR_INVALID_CHAR;
    return fail(c, SHK_ERR_INVALID_CHAR, "Invalid character '%s' in sequence. Only ACGTN allowed.",
                shk::byte_as_char((uint8_t)(c->h_stats->bad & 0xFF)).c_str());
  }
  int rc = drain_spill(c, c->unsettled_spill_cap);
  if (rc != SHK_OK) return rc;
  if (!c->held_keys.empty() && !c->acc_active) {  // what a grouped flush took off the device goes back in now
    rc = replay_held_spills(c);
    if (rc != SHK_OK) return rc;
  }
  // keep the load factor ≤ 1/2 for the next launch
  if (c->h_stats->n_distinct * 2 > c->tb.cap) {
    rc = grow_to(c, log_pages_for(c->h_stats->n_distinct * 4, c->owner_bits));
    if (rc != SHK_OK) return rc;
  }
  return SHK_OK;
}

// Spilled records that were taken off the device in the middle of a grouped flush (below): they go back in, through
// the ordinary repair path (which may grow the table), once no records wait for the old geometry any more.
static int hold_spills(shk_ctx *c, uint64_t spill_cap) {
  const uint64_t n = c->h_stats->spill_count;
  if (n == 0) return SHK_OK;
  if (n > spill_cap)
    return fail(c, SHK_ERR_INVARIANT, "spill list overflow (%llu > %llu)", (unsigned long long)n, (unsigned long long)spill_cap);
  SpillRef sp = spill_ref(c->spillA, spill_cap);
  const size_t at = c->held_keys.size();
  c->held_keys.resize(at + n);
  c->held_lanes.resize(at + n);
  c->held_counts.resize(at + n);
  HIPC(c, hipMemcpyAsync(c->held_keys.data() + at, sp.keys, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipMemcpyAsync(c->held_lanes.data() + at, sp.lanes, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipMemcpyAsync(c->held_counts.data() + at, sp.counts, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipMemsetAsync(&c->d_stats->spill_count, 0, sizeof(unsigned long long), c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  c->h_stats->spill_count = 0;
  return SHK_OK;
}
static int replay_held_spills(shk_ctx *c) {
  // (a flush that spilled a quarter of a table's worth of records: whatever the caller's capacity hint promised is
  // not what the data hold — from here on windows end when the evidence says so)
  if (c->held_keys.size() > c->tb.cap / 4) c->cfg.table_capacity_hint = 0;
  while (!c->held_keys.empty()) {
    const uint64_t cap = std::max<uint64_t>(std::min<uint64_t>(c->held_keys.size(), 1ull << 26), 1);
    const uint64_t n = std::min<uint64_t>(c->held_keys.size(), cap);
    HIPC(c, c->spillA.ensure(cap * 16));
    SpillRef sp = spill_ref(c->spillA, cap);
    const size_t at = c->held_keys.size() - n;  // (from the back: the vectors shrink as they are replayed)
    HIPC(c, hipMemcpyAsync(sp.keys, c->held_keys.data() + at, n * 8, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(sp.lanes, c->held_lanes.data() + at, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipMemcpyAsync(sp.counts, c->held_counts.data() + at, n * 4, hipMemcpyHostToDevice, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    c->held_keys.resize(at);
    c->held_lanes.resize(at);
    c->held_counts.resize(at);
    c->h_stats->spill_count = n;
    int rc = drain_spill(c, cap);
    if (rc != SHK_OK) return rc;
  }
  std::vector<uint64_t>().swap(c->held_keys);
  std::vector<uint32_t>().swap(c->held_lanes);
  std::vector<uint32_t>().swap(c->held_counts);
  return SHK_OK;
}

// Count the records that are waiting in the accumulation regions: one k_pages32 launch over every
// lane's regions, then the cursors go back to zero.  The launch's outcome is looked at by the
// settle that follows.
//
// EXACTNESS WHATEVER THE TABLE'S SIZE: a page pass spills at most the records it reads, and a region holds at
// most acc_cap of them, so a launch over G pages can put at most G · lanes · acc_cap entries on the spill list.
// When that bound for ALL pages exceeds the list (a window far larger than the list — or a capacity hint that was
// far too low: every new key of every page spills), the flush goes over the pages in GROUPS whose bound fits;
// after each group the host takes what was spilled off the device (hold_spills: pinned-size independent, it is
// ordinary host memory) — the table cannot be grown while records still wait for its geometry — and puts it back
// through the ordinary repair path once the last group is done (replay_held_spills, from settle_checked).
static int flush_acc(shk_ctx *c) {
  if (!c->acc_active) return SHK_OK;
  const uint32_t n_pages = 1u << c->acc_lp;
  if (c->acc_lp != c->tb.log_pages) return fail(c, SHK_ERR_INVARIANT, "table geometry changed under waiting records");
  const uint32_t NL = c->acc_region_lanes;
  // what the page pass spills (new keys that find their page full) goes behind what the window's
  // partition launches may have spilled already: same list, same capacity, the counter runs on
  const uint64_t spill_cap = c->acc_spill_cap;
  if (c->spillA.cap < spill_cap * 16) return fail(c, SHK_ERR_INVARIANT, "spill list of the accumulation window missing");
  SpillRef sp = spill_ref(c->spillA, spill_cap);
  const bool fresh = c->tb_stale && c->acc_rec32 && NL == c->n_lanes && !env_int("SHK_NO_FRESH", 0);
  // (8-byte records: k_pages<FK, FV> — the pass zeroes each lane's counts page by page and writes the keys whole,
  // instead of a fill of the table beforehand that it would then read back: 26 GB of configs[2]'s traffic)
  const bool fresh8 = c->tb_stale && !c->acc_rec32 && NL == c->n_lanes && !env_int("SHK_NO_FRESH", 0) && env_int("SHK_FRESH8", 1) != 0;
  if (fresh8) fused_drop(c);
  if (!fresh && !fresh8) {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  // pages per launch so that even if every record of every one of them spilled the list would hold them
  const uint64_t per_page_ub = (uint64_t)NL * c->acc_cap;  // (every lane's regions of a page in one launch, either record width)
  uint64_t ppg = std::max<uint64_t>(spill_cap / std::max<uint64_t>(per_page_ub, 1), 1);
  if (env_int("SHK_FLUSH_GROUP_PAGES", 0) > 0) ppg = (uint64_t)env_int("SHK_FLUSH_GROUP_PAGES", 0);  // test hook
  // (a window whose records all fit the list — the usual case — is one launch whatever the regions could hold)
  const bool grouped = ppg < n_pages && (c->acc_records_ub > spill_cap || env_int("SHK_FLUSH_GROUP_PAGES", 0) > 0);
  if (!grouped) ppg = n_pages;
  if (grouped) {  // the partition launches' own spills first: the groups need the whole list
    int rc = read_stats(c);
    if (rc != SHK_OK) return rc;
    if (c->h_stats->bad != ~0ull) return SHK_OK;  // (the settle that follows reports it; nothing is counted)
    rc = hold_spills(c, spill_cap);
    if (rc != SHK_OK) return rc;
  }
  if (!c->acc_rec32)
    HIPC(c, c->part2.ensure((uint64_t)n_pages * std::min<uint64_t>((uint64_t)c->acc_cap + MISS_SLACK, MISS_PAGE_MAX) * 8));  // (planned with the regions)
  FusedHist fh{};
  // (only the flush a finalize asked for: a window that ends on its budget is followed by another, whose pass would
  // overtake this one's histogram — configs[4]'s share, ten lanes: 4.5 ms of fold for nothing)
  const bool fuse = fresh && c->acc_rec32 && c->flush_for_finalize && fused_hist_wanted(c);
  if (fuse) {
    int rch = fused_hist_prepare(c, n_pages, &fh);
    if (rch != SHK_OK) return rch;
  }
  for (uint64_t p0 = 0; p0 < n_pages; p0 += ppg) {
    const uint32_t gp = (uint32_t)std::min<uint64_t>(ppg, n_pages - p0);
    if (c->acc_rec32) {
      ScopedTimer t(c, SHK_K_PAGES);
      if (fresh && fuse && NL == 1)
        hipLaunchKernelGGL((k_pages32<true, 2>), dim3(gp), dim3(PG_WG), 0, c->stream, c->tb, 0u, NL, 0u,
                           NL * n_pages, (const unsigned int *)c->acc_cur.p, c->acc_cap, (const uint32_t *)c->acc_buf.p,
                           c->d_stats, sp, (uint32_t)p0, fh);
      else if (fresh && fuse)
        hipLaunchKernelGGL((k_pages32<true, 1>), dim3(gp), dim3(PG_WG), 0, c->stream, c->tb, 0u, NL, NL > 1 ? n_pages : 0u,
                           NL * n_pages, (const unsigned int *)c->acc_cur.p, c->acc_cap, (const uint32_t *)c->acc_buf.p,
                           c->d_stats, sp, (uint32_t)p0, fh);
      else if (fresh)
        hipLaunchKernelGGL((k_pages32<true, 0>), dim3(gp), dim3(PG_WG), 0, c->stream, c->tb, 0u, NL, NL > 1 ? n_pages : 0u,
                           NL * n_pages, (const unsigned int *)c->acc_cur.p, c->acc_cap, (const uint32_t *)c->acc_buf.p,
                           c->d_stats, sp, (uint32_t)p0, fh);
      else
        hipLaunchKernelGGL((k_pages32<false, 0>), dim3(gp), dim3(PG_WG), 0, c->stream, c->tb, 0u, NL, NL > 1 ? n_pages : 0u,
                           NL * n_pages, (const unsigned int *)c->acc_cur.p, c->acc_cap, (const uint32_t *)c->acc_buf.p,
                           c->d_stats, sp, (uint32_t)p0);
    } else {  // 8-byte records: every lane's regions of a page in one launch (the keys stay in LDS over the lanes)
      ScopedTimer t(c, SHK_K_PAGES);
      const unsigned int *cur = (const unsigned int *)c->acc_cur.p;
      const uint64_t *buf = (const uint64_t *)c->acc_buf.p;
      if (fresh8)  // the table's first page pass: nothing is read, keys and counts are written whole
        hipLaunchKernelGGL((k_pages<true, true>), dim3(gp), dim3(PG_WG), 0, c->stream, c->tb, 0u, NL, n_pages, cur, c->acc_cap, buf, (uint64_t *)c->part2.p, c->d_stats, sp, (uint32_t)p0);
      else
        hipLaunchKernelGGL((k_pages<false, false>), dim3(gp), dim3(PG_WG), 0, c->stream, c->tb, 0u, NL, n_pages, cur, c->acc_cap, buf, (uint64_t *)c->part2.p, c->d_stats, sp, (uint32_t)p0);
      if (grouped) {
        int rc = read_stats(c);
        if (rc == SHK_OK) rc = hold_spills(c, spill_cap);
        if (rc != SHK_OK) return rc;
      }
    }
    if (grouped && c->acc_rec32) {
      int rc = read_stats(c);
      if (rc == SHK_OK) rc = hold_spills(c, spill_cap);
      if (rc != SHK_OK) return rc;
    }
  }
  if (c->acc_rec32 || fresh8) c->tb_stale = false;
  if (fuse) c->fused_valid = true, c->fused_pages = n_pages;
  HIPC(c, hipMemsetAsync(c->acc_cur.p, 0, (size_t)NL * n_pages * 4, c->stream));
  c->acc_active = false;
  c->acc_records_ub = 0;
  c->acc_lane_ub.assign(NL, 0);
  c->unsettled = true;
  c->unsettled_spill_cap = spill_cap;
  return SHK_OK;
}

// Full settle: everything that was launched has been looked at and repaired, and the table holds
// every record that was handed over (records waiting for a deferred page pass are counted first).
static int settle(shk_ctx *c) {
  if (c->acc_active) {
    if (c->unsettled) {  // the partition launches that filled the regions: an invalid byte stops everything
      int rc = read_stats(c);
      if (rc != SHK_OK) return rc;
      if (c->h_stats->bad != ~0ull) return settle_checked(c);  // poisons the context; nothing is counted
    }
    const uint64_t window_records = c->acc_records_ub;
    int rc = flush_acc(c);  // (its spills join the partition launches' on the window's spill list)
    if (rc != SHK_OK) return rc;
    rc = read_stats(c);
    if (rc != SHK_OK) return rc;
    if (window_records)  // what the next window's budget is planned with (acc_prepare)
    {
      const uint64_t nd1 = c->h_stats->n_distinct;
      c->acc_new_frac = (double)((nd1 > c->acc_nd0 ? nd1 - c->acc_nd0 : 0) + c->h_stats->spill_count) /
                        (double)window_records;
    }
    return settle_checked(c);
  }
  if (!c->unsettled) return SHK_OK;
  int rc = read_stats(c);
  if (rc != SHK_OK) return rc;
  return settle_checked(c);
}

// Between launches of an ingest: look at the last launch's outcome, but leave records that are
// waiting for a deferred page pass where they are unless something has to be repaired.
static int settle_light(shk_ctx *c) {
  if (!c->unsettled) return SHK_OK;
  if (!c->acc_active) return settle(c);
  int rc = read_stats(c);
  if (rc != SHK_OK) return rc;
  if (c->h_stats->bad != ~0ull || c->h_stats->spill_count > 0) {
    SHK_TRACEF("settle_light: last launch spilled %llu records (window %llu booked) -> settle\n", (unsigned long long)c->h_stats->spill_count,
               (unsigned long long)c->acc_records_ub);
    return settle(c);  // (re-reads the stats; rare)
  }
  c->unsettled = false;  // a clean partition launch: nothing to repair, the table was not touched
  return SHK_OK;
}

#include "shk_group.hip.h"

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int shk_abi_version(void) { return SHK_ABI_VERSION; }

const char *shk_last_error(const shk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int shk_create(const shk_config *cfg, shk_ctx **out) {
  if (!cfg || !out) return fail(nullptr, SHK_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  // cli.rs:662-673 (the "k odd" rule is the CLI's, cli.rs:667)
  if (!(cfg->k < 32))
    return fail(nullptr, SHK_ERR_BAD_ARG,
                "k must be less than 32 due to use of 64 bit integers to encode kmers");
  if (!(cfg->k > 0)) return fail(nullptr, SHK_ERR_BAD_ARG, "k must be greater than 0");
  if (!(cfg->histo_max > 0)) return fail(nullptr, SHK_ERR_BAD_ARG, "histo_max must be greater than 0");
  if (!(cfg->histo_max <= 1000000))
    return fail(nullptr, SHK_ERR_BAD_ARG, "histo_max must not exceed 1000000, got %llu",
                (unsigned long long)cfg->histo_max);
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(nullptr, SHK_ERR_NO_DEVICE, "no HIP device available (libshk has no CPU fallback)");
  if (cfg->n_devices > 1) return group_create(cfg, out);  // one owner share per device (shk_group.hip.h)
  if (cfg->n_devices == 1 && cfg->device_ids) {
    shk_config c1 = *cfg;
    c1.device = cfg->device_ids[0];
    c1.n_devices = 0;
    c1.device_ids = nullptr;
    return shk_create(&c1, out);
  }
  if (cfg->device < 0 || cfg->device >= n_dev)
    return fail(nullptr, SHK_ERR_NO_DEVICE, "device %d out of range (have %d)", cfg->device, n_dev);
  if (hipSetDevice(cfg->device) != hipSuccess)
    return fail(nullptr, SHK_ERR_NO_DEVICE, "hipSetDevice(%d) failed", cfg->device);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
    return fail(nullptr, SHK_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, SHK_ERR_NO_DEVICE, "device %d is %s; libshk is built for gfx950 only",
                cfg->device, prop.gcnArchName);

  if (cfg->n_owners > 1) {
    if (cfg->n_owners > 64 || (cfg->n_owners & (cfg->n_owners - 1)))
      return fail(nullptr, SHK_ERR_BAD_ARG, "n_owners must be a power of two ≤ 64, got %u", cfg->n_owners);
    if (cfg->owner_id >= cfg->n_owners)
      return fail(nullptr, SHK_ERR_BAD_ARG, "owner_id %u out of range (n_owners %u)", cfg->owner_id, cfg->n_owners);
  }
  shk_ctx *c = new shk_ctx();
  c->cfg = *cfg;
  c->cfg.device_ids = nullptr;
  c->cfg.n_devices = 0;
  c->n_lanes = cfg->chunks == 0 ? 1 : cfg->chunks;  // io.rs:378
  c->share_w1 = cfg->n_owners == 1;
  if (cfg->n_owners > 1) {
    c->n_owners = cfg->n_owners;
    c->owner_id = cfg->owner_id;
    while ((1u << c->owner_bits) < c->n_owners) c->owner_bits++;
    if (2 * cfg->k < c->owner_bits + 1) {
      delete c;
      return fail(nullptr, SHK_ERR_BAD_ARG, "k = %u leaves no key bits below %u owner bits", cfg->k, c->owner_bits);
    }
  }
  if (prop.multiProcessorCount > 0) c->n_cus = (uint32_t)prop.multiProcessorCount;
  c->lane_reads.assign(c->n_lanes, 0);
  auto bail = [&](int code) {
    g_create_error = c->err;
    shk_destroy(c);
    return code;
  };
#define HIPB(expr)                                                                           \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      fail(c, SHK_ERR_HIP, "HIP error %s in %s", hipGetErrorString(e__), #expr);             \
      return bail(e__ == hipErrorOutOfMemory ? SHK_ERR_NOMEM : SHK_ERR_HIP);                 \
    }                                                                                        \
  } while (0)
  {
    // compute units left to others (shk_config.reserve_cus): the PERSISTENT kernel of the context — k_scatter32, one
    // 1024-thread workgroup with 144 KiB of LDS per CU for the whole launch — starts that many workgroups fewer.  (A CU
    // mask on the context's stream — hipExtStreamCreateWithCUMask; bit i = CU i / 8 of XCD i % 8, tools/cumask_probe.hip
    // — was built and measured first: it also holds the many-workgroup kernels off the reserved CUs, and it costs the
    // config-4 share 17 to 32 % at a world of one — 126 Gbases/s unmasked, 86 / 87 / 105 / 92 with 8 / 16 / 32 / 64 CUs
    // masked off: the masked queue dispatches slowly, the level-2 pass loses 30 % to 3 % fewer CUs.  Not kept.)
    uint32_t r = cfg->reserve_cus == SHK_RESERVE_NONE ? 0u : cfg->reserve_cus;
    if (cfg->reserve_cus == 0 && cfg->n_owners > 1) r = (uint32_t)std::max(0, env_int("SHK_RESERVE_CUS", 16));
    c->n_cus_scatter = c->n_cus - std::min((r + 7u) & ~7u, c->n_cus > 64 ? c->n_cus - 64 : 0u);
    HIPB(stream_take(&c->stream));
  }
  HIPB(hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming));
  const size_t hist_n = (size_t)std::max<uint32_t>(cfg->chunks, 1) * (cfg->histo_max + 2);
  {
    c->ctl_tot_off = (sizeof(DevStats) + sizeof(unsigned long long) * c->n_lanes + 15) & ~(size_t)15;  // (behind the live lane counters)
    size_t off = c->ctl_tot_off + sizeof(HistoTotals) + sizeof(unsigned long long) * (c->n_lanes + 4);  // (+ lane sums, d_extra)
    c->ctl_hist_off = (off + 15) & ~(size_t)15;
    c->ctl_bytes = (c->ctl_hist_off + hist_n * sizeof(unsigned long long) + 15) & ~(size_t)15;
  }
  HIPB(dev_alloc((void **)&c->d_ctl, c->ctl_bytes, &c->ctl_alloc));
  HIPB(host_alloc((void **)&c->h_ctl, c->ctl_bytes, &c->ctl_alloc_h));
  memset(c->h_ctl, 0, c->ctl_bytes);
  c->d_stats = (DevStats *)c->d_ctl;
  c->h_stats = (DevStats *)c->h_ctl;
  c->d_lane_bases = (unsigned long long *)(c->d_ctl + sizeof(DevStats));
  c->h_lane_bases = (unsigned long long *)(c->h_ctl + sizeof(DevStats));
  c->d_tot = (HistoTotals *)(c->d_ctl + c->ctl_tot_off);
  c->h_totp = (HistoTotals *)(c->h_ctl + c->ctl_tot_off);
  c->d_lane_sum = (unsigned long long *)(c->d_ctl + c->ctl_tot_off + sizeof(HistoTotals));
  c->h_lane_sum = (unsigned long long *)(c->h_ctl + c->ctl_tot_off + sizeof(HistoTotals));
  c->d_extra = c->d_lane_sum + c->n_lanes;
  c->h_extra = c->h_lane_sum + c->n_lanes;
  c->d_hist = (unsigned long long *)(c->d_ctl + c->ctl_hist_off);
  c->h_hist = (const uint64_t *)(c->h_ctl + c->ctl_hist_off);
  c->h_stats->bad = ~0ull;
  uint64_t want = cfg->table_capacity_hint ? cfg->table_capacity_hint * 2 : (1ull << 20);
  if (c->is_share() && 2 * cfg->k > 32 + c->owner_bits && 2 * cfg->k - 32 <= (uint32_t)env_int("SHK_LEVEL1_LOG", 10))
    // an owner share starts with enough pages for 4-byte exchange records (2k − level-1 bits ≤ 32, the
    // level-1 fan-out being at most the page bits of the virtual table): k = 21 → 2^10 pages over all owners
    want = std::max<uint64_t>(want, (uint64_t)PAGE_SLOTS << (2 * cfg->k - 32 - c->owner_bits));
  // (the table's memory is not cleared here either: see tb_stale)
  const bool lazy = !env_int("SHK_NO_FRESH", 0);
  int rc = alloc_table(c, log_pages_for(want, c->owner_bits), &c->tb, !lazy);
  if (rc == SHK_OK) {
    TableRef none{};
    rc = fill_state(c, none, true);
    c->tb_stale = lazy;
  }
  if (rc != SHK_OK) return bail(rc);
  HIPB(hipStreamSynchronize(c->stream));
#undef HIPB
  *out = c;
  return SHK_OK;
}

void shk_destroy(shk_ctx *c) {
  if (!c) return;
  if (c->group) return group_destroy(c);
  (void)hipSetDevice(c->cfg.device);
  const auto t_d0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (trace_on()) fprintf(stderr, "[shk] destroy: %-22s at %7.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_d0).count());
  };
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  resolve_timings(c);
  for (auto e : c->event_pool) (void)hipEventDestroy(e);
  lap("synced, events gone");
  if (c->tb.keys) free_table(c->tb);
  lap("table given back");
  if (c->done_ev) (void)hipEventDestroy(c->done_ev);
  if (c->xs_ev) (void)hipEventDestroy(c->xs_ev);
  if (c->d_ctl) dev_free(c->d_ctl, c->ctl_alloc);
  if (c->h_ctl) host_free(c->h_ctl, c->ctl_alloc_h);
  c->in_bases.release();
  c->in_offsets.release();
  for (int i = 0; i < shk_ctx::NST; ++i) c->st_bases[i].release(), c->st_offsets[i].release();
  for (int i = 0; i < shk_ctx::NST; ++i) {
    c->h_rebased[i].release();
    if (c->copy_done[i]) (void)hipEventDestroy(c->copy_done[i]);
  }
  lap("control block, staging");
  if (c->copy_stream) {
    (void)hipStreamSynchronize(c->copy_stream);
    stream_give(c->copy_stream);
  }
  lap("copy stream destroyed");
  c->startbits.release();
  c->tiles.release();
  c->spillA.release();
  c->spillB.release();
  c->misc.release();
  c->part.release();
  c->part2.release();
  c->part3.release();
  c->part_meta.release();
  c->acc_buf.release();
  c->acc_cur.release();
  c->xbuf.release();
  c->xbuf_alt.release();
  c->part_meta_alt.release();
  c->xspill.release();
  c->xw_kmers.release();  // (the wide exchange round's output: 12 B per k-mer of the largest batch)
  c->xw_lanes.release();
  c->xw_count.release();
  c->fh_partial.release();
  c->fh_tot.release();
  for (int i = 0; i < shk_ctx::NST; ++i) c->pk_stage[i].release(), c->nm_stage[i].release(), c->nz_dev[i].release(), c->nz_host[i].release(), c->hp_pk[i].release(), c->hp_nm[i].release();
  c->pk_ascii.release();
  lap("scratch given back");
  if (c->stream) stream_give(c->stream);  // (synchronised at the top)
  lap("stream destroyed");
  delete c;
}

int shk_reset(shk_ctx *c) {
  if (!c) return SHK_ERR_BAD_ARG;
  if (c->group) {
    shk_group *g = c->group;
    for (uint32_t d = 0; d < g->D; ++d) {
      const int rc = shk_reset(g->ctx[d]);
      if (rc != SHK_OK) return group_fail(c, g, rc, d);
    }
    g->next_read = 0;
    g->finalized = g->hist_ready = false;
    c->err.clear();
    return SHK_OK;
  }
  HIPC(c, hipSetDevice(c->cfg.device));
  if (c->xs_pending) {  // (a round that was given up between shk_xchg_scatter_begin and _end: its copy of the statistics lands first)
    (void)hipEventSynchronize(c->xs_ev);
    c->xs_pending = false;
  }
  {
    // the control block (stats, totals, histogram); the TABLE is cleared by whoever touches it first — the first
    // page pass writes it whole instead, with the counts in it (k_pages32<true>)
    TableRef none{};
    int rc = fill_state(c, env_int("SHK_NO_FRESH", 0) ? c->tb : none, true);
    if (rc != SHK_OK) return rc;
    c->tb_stale = !env_int("SHK_NO_FRESH", 0);
    c->hist_dirty = false;
    c->fused_valid = false;
  }
  c->job_idx++;
  c->timing_now = !(c->cfg.flags & SHK_FLAG_TIMING_SAMPLED) || ((c->job_idx - 1) % 4 == 0);
  memset(c->h_stats, 0, sizeof(DevStats));
  c->h_stats->bad = ~0ull;
  // no host sync: everything later is ordered behind these on the engine stream; h_stats is
  // re-read (read_stats) before the host looks at it again
  std::fill(c->lane_reads.begin(), c->lane_reads.end(), 0);
  c->n_reads_read = c->n_bases_read = 0;
  c->n_inserted = 0;
  c->own_set = false;
  c->own_share_n = 0;
  c->held_keys.clear();
  c->held_lanes.clear();
  c->held_counts.clear();
  c->zero_count_keys = false;
  c->finalized = c->poisoned = c->hist_ready = false;
  c->unsettled = false;  // the memsets above are ordered behind any launch still in flight
  c->acc_active = false;  // (the regions' cursors are cleared when they are planned again; the regions themselves —
  c->acc_records_ub = 0;  // tens of GB on a large table — stay as they were sized: re-planning them from the free memory
                          // of the moment re-allocated them, which stalls the host for seconds)
  c->acc_new_frac = 1.0;
  c->poison_code = 0;
  c->err.clear();
  return SHK_OK;
}

// Host buffers → device → count.  A large batch is streamed in slices of whole reads through two
// device staging buffers: slice i+1 crosses PCIe on the copy stream while slice i is being
// counted on the engine stream (BASELINE.json configs[2]: "streamed chunks with copy/compute
// overlap").  Striping is unaffected: ingest_core advances the running read index per slice.
// packed != nullptr: the batch comes as the 2-bit stream + N mask of shk_pack_reads (offsets count BASES of
// that stream); every slice's share of both crosses PCIe instead of its ASCII bytes and is unpacked in HBM.
// The non-zero words of an N mask as (index, word) pairs — on a few threads: the caller is between two slices of
// a host-buffer ingest, with the link busy and nothing else to do.  Returns the number of pairs, or ~0 when
// there are more than `cap` (the mask then crosses the link as it is).
static size_t nmask_nonzero(const uint32_t *w, size_t n, uint32_t *pairs, size_t cap) {
  const unsigned T = (unsigned)std::max(1, std::min(8, (int)(n >> 18)));  // ≥ 1 MiB of mask per thread
  std::vector<size_t> cnt(T, 0);
  const size_t seg = cap / T;
  auto work = [&](unsigned t) {
    const size_t a = n * t / T, b = n * (t + 1) / T;
    uint32_t *out = pairs + 2 * seg * t;
    size_t k = 0;
    size_t i = a;
    for (; i < b && (i & 1); ++i)
      if (w[i]) { if (k < seg) out[2 * k] = (uint32_t)i, out[2 * k + 1] = w[i]; ++k; }
    for (; i + 8 <= b; i += 8) {  // (64-bit loads: the mask is 4-byte aligned, i is even here — and the host tolerates the rest)
      uint64_t q[4];
      memcpy(q, w + i, 32);
      if ((q[0] | q[1] | q[2] | q[3]) == 0) continue;
      for (size_t j = i; j < i + 8; ++j)
        if (w[j]) { if (k < seg) out[2 * k] = (uint32_t)j, out[2 * k + 1] = w[j]; ++k; }
    }
    for (; i < b; ++i)
      if (w[i]) { if (k < seg) out[2 * k] = (uint32_t)i, out[2 * k + 1] = w[i]; ++k; }
    cnt[t] = k;
  };
  if (T == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
  }
  size_t total = 0;
  for (unsigned t = 0; t < T; ++t) {
    if (cnt[t] > seg) return ~(size_t)0;
    if (t && cnt[t]) memmove(pairs + 2 * total, pairs + 2 * seg * t, cnt[t] * 8);
    total += cnt[t];
  }
  return total;
}

static int ingest_host(shk_ctx *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_seqs,
                       int64_t lane_fixed, const uint8_t *packed = nullptr, const uint32_t *nmask = nullptr) {
  if (!c) return SHK_ERR_BAD_ARG;
  if (n_seqs && (!offsets)) return fail(c, SHK_ERR_BAD_ARG, "null offsets");
  HIPC(c, hipSetDevice(c->cfg.device));
  if (n_seqs == 0) return ingest_core(c, nullptr, nullptr, 0, 0, lane_fixed);
  const uint64_t n_bases_all = offsets[n_seqs] - offsets[0];
  if (n_bases_all && !bases && !packed) return fail(c, SHK_ERR_BAD_ARG, "null bases");
  if (packed && !nmask) return fail(c, SHK_ERR_BAD_ARG, "null N mask");
  // Slices of 128 M bases of ASCII: the link is the bottleneck there and the first slice's copy is the one nothing
  // overlaps (config 3 streamed from pinned memory, 4 M reads per call: 11.7 ms per call at 128 M, 12.1 at 256 M, 14.0
  // at 512 M).  A packed batch moves 0.3 B per base: the counting is the bottleneck, every slice costs a host round
  // trip, and 256 M bases per slice are fastest (5.3 ms per call; 128 M: 5.6-6.0, 512 M: 6.6) — the first slice a
  // quarter of that.
  // An ASCII batch is PACKED ON THE HOST, slice by slice, when the host has the cores for it (2-bit stream + N mask by
  // shk_pack_reads' AVX2 converter, ≈ 77 Gbases/s on 16 cores, into pinned staging, by a thread of its own that runs
  // ahead of the copies and the counting): the link moves 52 GB/s = 52 Gbases/s of ASCII out of pinned memory and 41
  // out of pageable memory, and 0.3 B per base of a packed slice — and an invalid byte is found, and reported with the
  // same text, before anything of its slice is copied.  SHK_HOST_PACK=0 / 1 pins the choice.
  const int hp_env = env_int("SHK_HOST_PACK", -1);
  bool hp = false;
  if (!packed && bases && n_bases_all) {
    if (hp_env >= 0) {
      hp = hp_env != 0;
    } else if (shk::usable_cpus() >= 12 && n_bases_all >= (8u << 20) && n_bases_all <= (1ull << 31)) {
      // (up to 2^31 bases a call: the call's packed staging is pinned memory, 3/8 of a byte per base.)  By default only
      // for a batch in PAGEABLE memory — 41 → 50-52 Gbases/s; out of pinned memory the gain over the link's 52 is small
      // (55; config 3's stream 43.6 against 41.7: in line the packing runs at ≈ 55, not 77 — it shares the host's
      // memory bus with the copies) and a busy host would turn it into a loss
      hipPointerAttribute_t at{};
      if (hipPointerGetAttributes(&at, bases) == hipSuccess) hp = at.type != hipMemoryTypeHost && at.type != hipMemoryTypeManaged && at.type != hipMemoryTypeDevice;
      else (void)hipGetLastError(), hp = true;  // (ordinary host memory: not an error)
    }
  }
  const uint64_t slice_kb = (uint64_t)env_int("SHK_SLICE_KB", packed ? 256 << 10 : 128 << 10);  // test hook: tiny slices
  const uint64_t slice_bases = slice_kb << 10;
  // slice boundaries at read boundaries, ≈ slice_bases each
  std::vector<uint64_t> cut{0};
  while (cut.back() < n_seqs) {
    uint64_t lo = cut.back(), hi = n_seqs;
    const uint64_t limit = offsets[lo] + (lo == 0 && (packed || hp) ? std::max<uint64_t>(slice_bases / 4, 1) : slice_bases);
    if (offsets[n_seqs] > limit) {  // largest hi with offsets[hi] ≤ limit, at least one read
      hi = (uint64_t)(std::upper_bound(offsets + lo, offsets + n_seqs + 1, limit) - offsets) - 1;
      if (hi <= lo) hi = lo + 1;
    }
    cut.push_back(hi);
  }
  const size_t n_slices = cut.size() - 1;
  if (!c->copy_stream) HIPC(c, stream_take(&c->copy_stream));
  if (!c->copy_done[0])
    for (int i = 0; i < shk_ctx::NST; ++i) HIPC(c, hipEventCreateWithFlags(&c->copy_done[i], hipEventDisableTiming));
  bool offsets_pinned = false;
  {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, offsets) == hipSuccess) offsets_pinned = at.type == hipMemoryTypeHost;
    else (void)hipGetLastError();  // (ordinary host memory: not an error)
  }
  // NST staging sets, taken in turn across slices and across calls: while slice i is counted, the copies of slices
  // i+1 … i+NST−1 are queued on the copy stream, so the link does not wait for the host to come back from a counting
  // launch — nor from a deferred page pass of a large table, which holds the host for several slices' worth of copy
  // time (config 3: ≈ 10-20 ms against 5 ms per slice) — and the set a call starts with is never the one the previous
  // call's last launch (which nobody may have waited for: SHK_FLAG_DEFER_ERRORS) is still reading.
  // A call of few slices takes few sets (at least two: one may still be read by the previous call's last launch):
  // a stream of one-slice calls — shk_run_files' 64 M-base batches — allocates two sets, not six (and frees two:
  // tearing the context down was 14 ms of an 80 ms job with all six in use).
  const int NST = (int)std::min<size_t>(shk_ctx::NST, std::max<size_t>(n_slices + 1, 2));
  const uint32_t s0 = c->stage_last >= 0 ? (uint32_t)(c->stage_last + 1) % (uint32_t)NST : 0u;
  auto set_of = [&](size_t i) { return (int)((s0 + i) % (uint32_t)NST); };
  auto bases_of = [&](int sel) -> DevBuf & { return c->st_bases[sel]; };
  auto offs_of = [&](int sel) -> DevBuf & { return c->st_offsets[sel]; };
  // Host packing: ONE thread of its own packs the call's slices, in order, each into its own stretch of the call's
  // pinned staging (shk_pack_reads: all the host's cores per slice), and says how far it has come; this thread sends
  // what is packed and counts what has arrived — a deferred page pass that holds it up does not hold the packing up.
  std::vector<size_t> hp_pk_off, hp_nm_off;
  std::mutex hp_m;
  std::condition_variable hp_cv;
  size_t hp_done = 0;      // slices packed so far
  int hp_rc = SHK_OK;      // … or the packer's error (an invalid byte: the run is over)
  std::string hp_err;
  bool hp_quit = false;
  std::thread hp_thread;
  struct HpJoin {  // (every way out of this function goes past the packer)
    std::thread &t;
    std::mutex &m;
    bool &quit;
    ~HpJoin() {
      {
        std::lock_guard<std::mutex> lk(m);
        quit = true;
      }
      if (t.joinable()) t.join();
    }
  } hp_join{hp_thread, hp_m, hp_quit};
  if (hp) {
    size_t pk_total = 0, nm_total = 0;
    for (size_t j = 0; j < n_slices; ++j) {
      const uint64_t nb = offsets[cut[j + 1]] - offsets[cut[j]];
      hp_pk_off.push_back(pk_total);
      hp_nm_off.push_back(nm_total);
      pk_total += (size_t)((nb / 4 + 64 + 63) & ~63ull);
      nm_total += (size_t)(((nb / 32 + 2) * 4 + 63) & ~63ull);
    }
    HIPC(c, c->hp_pk[0].ensure(pk_total));  // (free: every copy out of them was waited for before the call that made it returned)
    HIPC(c, c->hp_nm[0].ensure(nm_total));
    hp_thread = std::thread([&] {
      for (size_t j = 0; j < n_slices; ++j) {
        {
          std::lock_guard<std::mutex> lk(hp_m);
          if (hp_quit) return;
        }
        const uint64_t o0 = offsets[cut[j]], nb = offsets[cut[j + 1]] - o0;
        int prc = SHK_OK;
        if (nb) prc = shk_pack_reads(bases + o0, nb, (uint8_t *)c->hp_pk[0].p + hp_pk_off[j], (uint32_t *)((uint8_t *)c->hp_nm[0].p + hp_nm_off[j]), 0);
        std::lock_guard<std::mutex> lk(hp_m);
        if (prc != SHK_OK) {
          hp_rc = prc;
          hp_err = shk_run_error();  // (this thread's own message buffer)
          hp_cv.notify_all();
          return;
        }
        hp_done = j + 1;
        hp_cv.notify_all();
      }
    });
  }
  // slices [0, n) packed?  block: wait for slice n − 1 (or the packer's error, which is this call's)
  auto hp_packed = [&](bool block, size_t need) -> size_t {
    std::unique_lock<std::mutex> lk(hp_m);
    if (block) hp_cv.wait(lk, [&] { return hp_done >= need || hp_rc != SHK_OK; });
    return hp_done;
  };
  size_t hp_issued = 0;  // slices whose copies have been queued (host packing)
  auto issue_copy = [&](size_t i) -> int {
    const int bsel = set_of(i);
    const uint64_t r0 = cut[i], r1 = cut[i + 1];
    const uint64_t o0 = offsets[r0], nb = offsets[r1] - o0, ns = r1 - r0;
    DevBuf &db = bases_of(bsel);
    DevBuf &dof = offs_of(bsel);
    HIPC(c, db.ensure(nb + 64));
    HIPC(c, dof.ensure((ns + 1) * 8));
    const uint8_t *pk_src = packed;
    const uint32_t *nm_src = nmask;
    uint64_t o_rel = o0;  // the slice's first base in the packed streams
    if (nb && hp) {  // (the packer thread has finished this slice: the caller checked)
      pk_src = (const uint8_t *)c->hp_pk[0].p + hp_pk_off[i];
      nm_src = (const uint32_t *)((const uint8_t *)c->hp_nm[0].p + hp_nm_off[i]);
      o_rel = 0;
    }
    if (nb && pk_src) {  // the slice's bytes of the 2-bit stream and words of the N mask (+ pad: k_unpack reads a few bytes on)
      const uint64_t b0 = o_rel >> 2, b1 = (o_rel + nb + 3) >> 2, w0 = o_rel >> 5, w1 = (o_rel + nb + 31) >> 5;
      HIPC(c, c->pk_stage[bsel].ensure(b1 - b0 + 16));
      HIPC(c, c->nm_stage[bsel].ensure((w1 - w0 + 2) * 4));
      HIPC(c, hipMemcpyAsync(c->pk_stage[bsel].p, pk_src + b0, b1 - b0, hipMemcpyHostToDevice, c->copy_stream));
      // the N mask: a bit per base, a third of the slice's bytes — and nearly all zeros on most data.  When at
      // most 1/16 of its words are non-zero they cross the link as a list and the rest is cleared on the device.
      const size_t nw = (size_t)(w1 - w0), list_cap = nw / 16 + 64;
      size_t nz = ~(size_t)0;
      if (nw >= (1u << 16) && env_int("SHK_NMASK_SPARSE", 1)) {
        HIPC(c, c->nz_host[bsel].ensure(list_cap * 8));
        nz = nmask_nonzero(nm_src + w0, nw, (uint32_t *)c->nz_host[bsel].p, list_cap);
      }
      if (nz != ~(size_t)0) {
        HIPC(c, hipMemsetAsync(c->nm_stage[bsel].p, 0, nw * 4, c->copy_stream));
        if (nz) {
          HIPC(c, c->nz_dev[bsel].ensure(nz * 8));
          HIPC(c, hipMemcpyAsync(c->nz_dev[bsel].p, c->nz_host[bsel].p, nz * 8, hipMemcpyHostToDevice, c->copy_stream));
          hipLaunchKernelGGL(k_nmask_sparse, dim3((uint32_t)((nz + WG - 1) / WG)), dim3(WG), 0, c->copy_stream,
                             (uint32_t *)c->nm_stage[bsel].p, (const uint2 *)c->nz_dev[bsel].p, (uint32_t)nz);
        }
      } else {
        HIPC(c, hipMemcpyAsync(c->nm_stage[bsel].p, nm_src + w0, nw * 4, hipMemcpyHostToDevice, c->copy_stream));
      }
    } else if (nb)
      HIPC(c, hipMemcpyAsync(db.p, bases + o0, nb, hipMemcpyHostToDevice, c->copy_stream));
    // the caller's own offsets, as they are (the kernels that read them subtract the slice's first: off_bias) —
    // straight from the caller's memory when that is pinned, through a pinned staging copy otherwise (a pageable
    // source makes the copy synchronous and holds up the slice's other transfers)
    const uint64_t *osrc = offsets + r0;
    if (!offsets_pinned) {
      HIPC(c, c->h_rebased[bsel].ensure((ns + 1) * 8));  // (free: the copy that read it last has completed, see below)
      memcpy(c->h_rebased[bsel].p, osrc, (ns + 1) * 8);
      osrc = (const uint64_t *)c->h_rebased[bsel].p;
    }
    HIPC(c, hipMemcpyAsync(dof.p, osrc, (ns + 1) * 8, hipMemcpyHostToDevice, c->copy_stream));
    HIPC(c, hipEventRecord(c->copy_done[bsel], c->copy_stream));
    return SHK_OK;
  };
  // (sets s0 … s0+NST−2 were last read by launches that a later launch of the previous call has waited for)
  int rc = SHK_OK;
  // host packing: queue the copies of the slices that are packed, in order, up to slice max_j (what the staging sets
  // allow at this point); need > 0: wait until `need` slices are packed first
  auto hp_issue = [&](size_t max_j, size_t need) -> int {
    const size_t have = hp_packed(need > 0, need);
    if (have < need) {  // the packer gave up: encoding.rs:353-356, the run is over
      c->poisoned = true;
      c->poison_code = hp_rc;
      return fail(c, hp_rc, "%s", hp_err.c_str());
    }
    while (hp_issued < have && hp_issued < n_slices && hp_issued <= max_j) {
      const int r = issue_copy(hp_issued);
      if (r != SHK_OK) return r;
      ++hp_issued;
    }
    return SHK_OK;
  };
  for (size_t i = 0; !hp && i < std::min<size_t>(n_slices, (size_t)NST - 1) && rc == SHK_OK; ++i) rc = issue_copy(i);
  if (rc != SHK_OK) {
    (void)hipStreamSynchronize(c->copy_stream);
    return rc;
  }
  const bool trace = env_int("SHK_HOST_TRACE", 0) != 0;
  const auto t_begin = std::chrono::steady_clock::now();
  auto now_us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(); };
  for (size_t i = 0; i < n_slices; ++i) {
    const int bsel = set_of(i);
    if (hp) {  // slice i must be on its way; slice i − 1's count may still be reading set (i − 1) % NST = slice i + NST − 1's
      rc = hp_issue(i + (size_t)NST - 2, i + 1);
      if (rc != SHK_OK) {
        (void)hipStreamSynchronize(c->copy_stream);
        return rc;
      }
    }
    const double ta = now_us();
    HIPC(c, hipEventSynchronize(c->copy_done[bsel]));  // `rebased[bsel]` consumed; data resident
    if (trace) fprintf(stderr, "[slice %zu] copy waited %.0f us (at %.0f)", i, now_us() - ta, now_us());
    // slice i-1 (launched without a host sync) read the set that slice i+NST-1 is about to overwrite
    rc = settle_light(c);
    if (rc != SHK_OK) {
      (void)hipStreamSynchronize(c->copy_stream);
      return rc;
    }
    HIPC(c, hipStreamSynchronize(c->stream));
    if (trace) fprintf(stderr, " count(i-1) done at %.0f\n", now_us());
    if (hp) {
      rc = hp_issue(i + (size_t)NST - 1, 0);  // (whatever else is packed by now)
    } else if (i + NST - 1 < n_slices) {
      rc = issue_copy(i + NST - 1);
    }
    if (rc != SHK_OK) {
      (void)hipStreamSynchronize(c->copy_stream);
      return rc;
    }
    const uint64_t r0 = cut[i], r1 = cut[i + 1];
    DevBuf &db = bases_of(bsel);
    DevBuf &dof = offs_of(bsel);
    if ((packed || hp) && offsets[r1] > offsets[r0]) {  // 2-bit stream + N mask → the slice's ASCII bytes, in HBM
      const uint64_t o0 = hp ? 0 : offsets[r0], nb = offsets[r1] - offsets[r0];
      // (the staged copies start at byte o0/4 resp. word o0/32 of the streams; what k_unpack reads past their
      // end only ever feeds positions ≥ nb, which it does not write)
      hipLaunchKernelGGL(k_unpack, dim3((uint32_t)((nb + 16ull * WG - 1) / (16ull * WG))), dim3(WG), 0, c->stream,
                         (const uint8_t *)c->pk_stage[bsel].p, (uint32_t)(o0 & 3), (const uint32_t *)c->nm_stage[bsel].p,
                         (uint32_t)(o0 & 31), nb, (uint8_t *)db.p);
    }
    rc = ingest_core(c, (const uint8_t *)db.p, (const uint64_t *)dof.p, r1 - r0, offsets[r1] - offsets[r0],
                     lane_fixed, nullptr, offsets[r0]);
    if (rc != SHK_OK) {
      (void)hipStreamSynchronize(c->copy_stream);  // do not leave a copy reading `rebased` behind
      return rc;
    }
    if (hp) {  // (and what has been packed while this slice was launched)
      rc = hp_issue(i + (size_t)NST - 1, 0);
      if (rc != SHK_OK) {
        (void)hipStreamSynchronize(c->copy_stream);
        return rc;
      }
    }
  }
  c->stage_last = set_of(n_slices - 1);
  // A host-buffer ingest reports its errors (an invalid byte) before returning — unless the caller asked for
  // the device-buffer behaviour (SHK_FLAG_DEFER_ERRORS): then the last slice's launch is looked at by the next
  // call, whose first copies run under it.
  if (c->cfg.flags & SHK_FLAG_DEFER_ERRORS) return SHK_OK;
  // A packed batch cannot hold an invalid byte (two bits per base + the N mask: every value is a base), and its
  // copies have all completed (each slice's was waited for above): nothing the caller has to hear about before the
  // next call, so the last slice's count stays in flight (config 3 packed: 5.3 → 4.0 ms per 4 M-read call).
  if (packed || hp) return SHK_OK;  // (hp: every byte was looked at when it was packed)
  return settle_light(c);  // host-buffer ingest reports its errors (an invalid byte) before returning
}

int shk_ingest_batch(shk_ctx *c, uint32_t chunk_id, const uint8_t *bases, const uint64_t *offsets,
                     uint64_t n_seqs) {
  if (!c) return SHK_ERR_BAD_ARG;
  if (chunk_id >= c->n_lanes)
    return fail(c, SHK_ERR_BAD_ARG, "chunk_id %u out of range (n_chunks %u)", chunk_id, c->n_lanes);
  if (c->group) return group_ingest(c, bases, offsets, n_seqs, (int64_t)chunk_id);
  return ingest_host(c, bases, offsets, n_seqs, (int64_t)chunk_id);
}

int shk_ingest_reads(shk_ctx *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_seqs) {
  if (c && c->group) return group_ingest(c, bases, offsets, n_seqs, -1);
  return ingest_host(c, bases, offsets, n_seqs, -1);
}

// ---- 2-bit packed input (the reference's Read::from_str layout, encoding.rs:60-95, + an N mask) ----------
int shk_ingest_packed(shk_ctx *c, const uint8_t *packed, const uint32_t *nmask, const uint64_t *offsets, uint64_t n_seqs) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "a multi-device context takes ASCII host buffers");
  if (n_seqs && !packed) return fail(c, SHK_ERR_BAD_ARG, "null packed stream");
  return ingest_host(c, nullptr, offsets, n_seqs, -1, packed, nmask);
}

int shk_pack_reads_device(shk_ctx *c, const void *d_bases, uint64_t n_bases, void *d_packed, void *d_nmask) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  if (n_bases == 0) return SHK_OK;
  if (!d_bases || !d_packed || !d_nmask) return fail(c, SHK_ERR_BAD_ARG, "null buffer");
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcs = settle(c);  // (stats->bad below belongs to this call alone)
    if (rcs != SHK_OK) return rcs;
  }
  hipLaunchKernelGGL(k_pack, dim3((uint32_t)((n_bases + 32ull * WG - 1) / (32ull * WG))), dim3(WG), 0, c->stream,
                     (const uint8_t *)d_bases, n_bases, (uint8_t *)d_packed, (uint32_t *)d_nmask, c->d_stats);
  int rc = read_stats(c);
  if (rc != SHK_OK) return rc;
  if (c->h_stats->bad != ~0ull) {  // encoding.rs:353-356; the context's table was not touched: no poisoning
    const std::string bad = shk::byte_as_char((uint8_t)(c->h_stats->bad & 0xFF));
    HIPC(c, hipMemsetAsync(&c->d_stats->bad, 0xFF, 8, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
    c->h_stats->bad = ~0ull;
    return fail(c, SHK_ERR_INVALID_CHAR, "Invalid character '%s' in sequence. Only ACGTN allowed.", bad.c_str());
  }
  return SHK_OK;
}

int shk_unpack_reads_device(shk_ctx *c, const void *d_packed, const void *d_nmask, uint64_t n_bases, void *d_bases) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  if (n_bases == 0) return SHK_OK;
  if (!d_bases || !d_packed || !d_nmask) return fail(c, SHK_ERR_BAD_ARG, "null buffer");
  HIPC(c, hipSetDevice(c->cfg.device));
  hipLaunchKernelGGL(k_unpack, dim3((uint32_t)((n_bases + 16ull * WG - 1) / (16ull * WG))), dim3(WG), 0, c->stream,
                     (const uint8_t *)d_packed, 0u, (const uint32_t *)d_nmask, 0u, n_bases, (uint8_t *)d_bases);
  HIPC(c, hipGetLastError());
  return SHK_OK;
}

int shk_ingest_packed_device(shk_ctx *c, const void *d_packed, const void *d_nmask, const void *d_offsets, uint64_t n_seqs,
                             uint64_t n_bases) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "a multi-device context takes host buffers");
  if (!c) return SHK_ERR_BAD_ARG;
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcs = settle_light(c);  // (the previous launch may still be reading the unpacked copy of ITS batch)
    if (rcs != SHK_OK) return rcs;
    HIPC(c, hipStreamSynchronize(c->stream));
  }
  HIPC(c, c->pk_ascii.ensure(n_bases + 64));
  if (n_bases) {
    int rc = shk_unpack_reads_device(c, d_packed, d_nmask, n_bases, c->pk_ascii.p);
    if (rc != SHK_OK) return rc;
  }
  return ingest_core(c, (const uint8_t *)c->pk_ascii.p, (const uint64_t *)d_offsets, n_seqs, n_bases, -1);
}

int shk_set_read_index(shk_ctx *c, uint64_t next_read_index) {
  if (!c) return SHK_ERR_BAD_ARG;
  if (c->group) c->group->next_read = next_read_index;
  c->n_reads_read = next_read_index;
  return SHK_OK;
}

int shk_ingest_reads_device(shk_ctx *c, const void *d_bases, const void *d_offsets, uint64_t n_seqs,
                            uint64_t n_bases) {
  if (!c) return SHK_ERR_BAD_ARG;
  if (c->group) return fail(c, SHK_ERR_STATE, "a multi-device context takes host buffers (shk_ingest_reads / shk_ingest_batch)");
  HIPC(c, hipSetDevice(c->cfg.device));
  return ingest_core(c, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_seqs, n_bases, -1);
}

int shk_insert_counts(shk_ctx *c, uint32_t chunk_id, const uint64_t *kmers, const uint32_t *counts,
                      uint64_t n) {
  if (!c) return SHK_ERR_BAD_ARG;
  if (c->group) {  // every share is offered every k-mer and keeps what it owns
    shk_group *g = c->group;
    g->finalized = g->hist_ready = false;
    for (uint32_t d = 0; d < g->D; ++d) {
      const int rc = shk_insert_counts(g->ctx[d], chunk_id, kmers, counts, n);
      if (rc != SHK_OK) return group_fail(c, g, rc, d);
    }
    return SHK_OK;
  }
  if (c->poisoned) return fail(c, c->poison_code, "%s", c->err.c_str());
  {
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  if (chunk_id >= c->n_lanes) return fail(c, SHK_ERR_BAD_ARG, "chunk_id out of range");
  if (n == 0) return SHK_OK;
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  c->finalized = c->hist_ready = false;
  const uint64_t kmax = c->cfg.k >= 32 ? ~0ull : ((1ull << (2 * c->cfg.k)) - 1);
  for (uint64_t i = 0; i < n; ++i) {
    if (kmers[i] > kmax) return fail(c, SHK_ERR_BAD_ARG, "kmer %llu does not fit k=%u",
                                     (unsigned long long)kmers[i], c->cfg.k);
    if (counts[i] == 0) c->zero_count_keys = true;  // counting.rs:152-154 creates the entry all the same
  }
  int rc = ensure_capacity(c, n);
  if (rc != SHK_OK) return rc;
  c->n_inserted += n;
  HIPC(c, c->misc.ensure(n * 12));
  uint64_t *dk = (uint64_t *)c->misc.p;
  uint32_t *dc = (uint32_t *)((uint8_t *)c->misc.p + n * 8);
  HIPC(c, hipMemcpyAsync(dk, kmers, n * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemcpyAsync(dc, counts, n * 4, hipMemcpyHostToDevice, c->stream));
  HIPC(c, c->spillA.ensure(n * 16));
  SpillRef sp = spill_ref(c->spillA, n);
  HIPC(c, hipMemsetAsync(&c->d_stats->spill_count, 0, sizeof(unsigned long long), c->stream));
  {
    ScopedTimer t(c, SHK_K_INSERT);
    hipLaunchKernelGGL(k_insert, dim3(grid_for(n, WG, 4096)), dim3(WG), 0, c->stream, dk,
                       (const uint32_t *)nullptr, dc, n, chunk_id, c->tb, c->d_stats, sp);
  }
  rc = read_stats(c);
  if (rc != SHK_OK) return rc;
  return drain_spill(c, n);
}

// settle_light's look at what was unsettled when an exchange scatter was launched without waiting (the absorb in front of
// it), on the statistics read back behind the scatter.  absorbs_since: launches that those statistics may not cover.
static int xchg_late_settle(shk_ctx *c, bool late, bool absorbs_since) {
  if (!c->unsettled || !(late || absorbs_since)) return SHK_OK;
  if (c->h_stats->spill_count > 0) {
    SHK_TRACEF("xchg_scatter: an absorb spilled %llu records -> settle\n", (unsigned long long)c->h_stats->spill_count);
    return settle(c);
  }
  if (!absorbs_since) c->unsettled = false;
  return SHK_OK;
}

// The exchange scatter, in one call (the outcome waited for) or in two (shk_xchg_scatter_begin / _end).
static int xchg_scatter_call(shk_ctx *c, const void *d_bases, const void *d_offsets, uint64_t n_seqs, uint64_t n_bases, uint64_t layout_bases,
                             void **d_records, void **d_cursors, shk_xchg_layout *layout, uint64_t *n_foreign_spilled, bool two_calls) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c || !d_records || !d_cursors || !layout) return SHK_ERR_BAD_ARG;
  HIPC(c, hipSetDevice(c->cfg.device));
  if (c->xs_pending) return fail(c, SHK_ERR_STATE, "an exchange scatter is begun and not ended (shk_xchg_scatter_end)");
  if (n_bases > SHK_XCHG_MAX_BASES || (layout_bases && layout_bases > SHK_XCHG_MAX_BASES))
    return fail(c, SHK_ERR_BAD_ARG, "an exchange batch takes at most %llu bases", (unsigned long long)SHK_XCHG_MAX_BASES);
  XchgOut xo{};
  xo.layout_bases = layout_bases;
  xo.two_calls = two_calls;
  // Two exchange buffers, taken in turn: what this call hands out stays valid until the next call BUT ONE, so the
  // caller can have round r's segments on the links (another stream, peers pulling) while round r+1 is scattered.
  // What is unsettled here, between two rounds, is the previous round's ABSORB (a level-2 pass into the waiting page
  // regions; its spills go to the window's list, which this scatter — foreign spills on a list of its own — does not
  // touch, and it is in front of this scatter on the stream).  Waiting for it before launching cost a host round trip
  // with the GPU idle every round; the scatter's own read of the outcome, below, sees the absorb's too.
  xo.late_settle = c->unsettled && c->acc_active && !c->poisoned && env_int("SHK_XCHG_LATE_SETTLE", 1) != 0;
  if (!xo.late_settle) {
    int rcs = settle_light(c);
    if (rcs != SHK_OK) return rcs;
  }
  std::swap(c->xbuf, c->xbuf_alt);
  std::swap(c->part_meta, c->part_meta_alt);
  int rc = ingest_core(c, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_seqs, n_bases, c->xchg_lane_fixed, &xo);
  if (rc != SHK_OK) return rc;
  *d_records = xo.d_records;
  *d_cursors = xo.d_cursors;
  *layout = xo.lay;
  if (two_calls) {
    c->xs_pending = true;
    c->xs_late = xo.late_settle;
    c->xs_absorbs = c->n_absorbs;
    return SHK_OK;
  }
  if (n_foreign_spilled) *n_foreign_spilled = xo.n_foreign;
  return xchg_late_settle(c, xo.late_settle, /*absorbs_since=*/false);
}

int shk_xchg_scatter_device(shk_ctx *c, const void *d_bases, const void *d_offsets, uint64_t n_seqs, uint64_t n_bases,
                            uint64_t layout_bases, void **d_records, void **d_cursors, shk_xchg_layout *layout,
                            uint64_t *n_foreign_spilled) {
  return xchg_scatter_call(c, d_bases, d_offsets, n_seqs, n_bases, layout_bases, d_records, d_cursors, layout, n_foreign_spilled, false);
}

int shk_xchg_scatter_begin(shk_ctx *c, const void *d_bases, const void *d_offsets, uint64_t n_seqs, uint64_t n_bases,
                           uint64_t layout_bases, void **d_records, void **d_cursors, shk_xchg_layout *layout) {
  return xchg_scatter_call(c, d_bases, d_offsets, n_seqs, n_bases, layout_bases, d_records, d_cursors, layout, nullptr, true);
}

int shk_xchg_scatter_end(shk_ctx *c, uint64_t *n_foreign_spilled) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  if (!c->xs_pending) return fail(c, SHK_ERR_STATE, "no exchange scatter is begun (shk_xchg_scatter_begin)");
  c->xs_pending = false;
  HIPC(c, hipSetDevice(c->cfg.device));
  HIPC(c, hipEventSynchronize(c->xs_ev));  // (a later, synchronous read of the statistics — an absorb that ended a window — only makes them newer)
  if (c->poisoned) return fail(c, c->poison_code, "%s", c->err.c_str());
  uint64_t nf = 0;
  int rc = xchg_scatter_outcome(c, &nf);
  if (rc != SHK_OK) return rc;
  if (n_foreign_spilled) *n_foreign_spilled = nf;
  return xchg_late_settle(c, c->xs_late, c->n_absorbs != c->xs_absorbs);
}

int shk_xchg_wide_scatter_device(shk_ctx *c, const void *d_bases, const void *d_offsets, uint64_t n_seqs, uint64_t n_bases,
                                 void **d_kmers, void **d_lanes, uint64_t *counts) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c || !d_kmers || !d_lanes || !counts) return SHK_ERR_BAD_ARG;
  if (!c->is_share()) return fail(c, SHK_ERR_STATE, "not an owner share (shk_config.n_owners = 0: say 1 for a share that is the whole key space)");
  HIPC(c, hipSetDevice(c->cfg.device));
  if (n_bases > SHK_XCHG_MAX_BASES) return fail(c, SHK_ERR_BAD_ARG, "an exchange batch takes at most %llu bases", (unsigned long long)SHK_XCHG_MAX_BASES);
  XchgOut xo{};
  xo.wide = true;
  xo.counts = counts;
  int rc = ingest_core(c, (const uint8_t *)d_bases, (const uint64_t *)d_offsets, n_seqs, n_bases, c->xchg_lane_fixed, &xo);
  if (rc != SHK_OK) return rc;
  *d_kmers = xo.d_kmers;
  *d_lanes = xo.d_lanes;
  return SHK_OK;
}

int shk_xchg_feasible(shk_ctx *c) {
  if (!c || c->group || !c->is_share()) return 0;
  const PartGeom g = part_geom(c);
  return xchg_rec_bytes(c, g) ? 1 : 0;
}

int shk_xchg_absorb(shk_ctx *c, const void *d_records, const void *d_cursors, const shk_xchg_layout *lay) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c || !lay) return SHK_ERR_BAD_ARG;
  if (c->poisoned) return fail(c, c->poison_code, "%s", c->err.c_str());
  HIPC(c, hipSetDevice(c->cfg.device));
  PartGeom g = part_geom(c);
  int rc = xchg_check(c, g);
  if (rc != SHK_OK) return rc;
  if (lay->n_owners != c->n_owners || lay->n_lanes != c->n_lanes || lay->log_p1 != g.log_p1 ||
      lay->regions != c->n_lanes << (g.log_p1 - g.lw) || lay->region_cap == 0 || (lay->region_cap & ((1u << RB_LOG) - 1u)) ||
      lay->segment_records != (uint64_t)lay->regions * lay->region_cap || (lay->record_bytes ? lay->record_bytes : 4u) != xchg_rec_bytes(c, g))
    return fail(c, SHK_ERR_BAD_ARG, "exchange segment layout does not match this context (owners %u/%u, lanes %u/%u, level-1 bits %u/%u, record bytes %u/%u)",
                lay->n_owners, c->n_owners, lay->n_lanes, c->n_lanes, lay->log_p1, g.log_p1, lay->record_bytes ? lay->record_bytes : 4u, xchg_rec_bytes(c, g));
  const bool rec8 = xchg_rec_bytes(c, g) == 8;
  if (!d_records || !d_cursors) return fail(c, SHK_ERR_BAD_ARG, "null segment");
  c->finalized = c->hist_ready = false;
  // a segment's regions are sized 1.25 × (1.5 ×) their expected fill: 4/5 of it bounds what it holds in practice
  uint64_t est = lay->segment_records / 5 * 4 + 1024;
  if (c->acc_active && c->acc_spill_cap) est = std::min<uint64_t>(est, c->acc_spill_cap);
  const uint64_t lane_bound = lay->segment_records / lay->n_lanes;  // a lane's regions of the segment cannot hold more
  rc = acc_prepare(c, est, -1, lane_bound);  // (may count what is waiting first: launch + settle)
  if (rc != SHK_OK) return rc;
  g = part_geom(c);  // a settle may have grown the table
  rc = xchg_check(c, g);
  if (rc != SHK_OK) return rc;
  const uint64_t spill_cap = std::max<uint64_t>(c->acc_spill_cap, est);
  c->acc_spill_cap = spill_cap;
  if (c->spillA.cap < spill_cap * 16) {
    if (c->unsettled) {  // (its list is about to move)
      rc = settle(c);
      if (rc != SHK_OK) return rc;
      rc = acc_prepare(c, est, -1, lane_bound);
      if (rc != SHK_OK) return rc;
      g = part_geom(c);
    }
    HIPC(c, c->spillA.ensure(spill_cap * 16));
  }
  SpillRef sp = spill_ref(c->spillA, spill_cap);
  rc = rec8 ? xl64_absorb(c, g, (const uint64_t *)d_records, (const unsigned int *)d_cursors, lay->region_cap, lay->regions, sp)
            : xl_absorb(c, g, (const uint32_t *)d_records, (const unsigned int *)d_cursors, lay->region_cap, lay->regions, sp);
  if (rc != SHK_OK) return rc;
  if (trace_on()) {  // (debugging aid: where did the segment's records go?)
    const size_t nreg = (size_t)c->n_lanes << c->tb.log_pages;
    std::vector<unsigned int> cur(nreg), src(lay->regions);
    unsigned long long spilled = 0;
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemcpy(cur.data(), c->acc_cur.p, nreg * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(src.data(), d_cursors, (size_t)lay->regions * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&spilled, &c->d_stats->spill_count, 8, hipMemcpyDeviceToHost);
    unsigned long long sum = 0, ssum = 0;
    unsigned int mx = 0, mn = ~0u, over = 0, smx = 0;
    for (unsigned int v : cur) sum += v, mx = std::max(mx, v), mn = std::min(mn, v), over += v > c->acc_cap;
    for (size_t i = 0, shown = 0; i < cur.size() && shown < 6; ++i)
      if (cur[i] > c->acc_cap) {
        SHK_TRACEF("  page %zu (sub %zu of super-page %zu): %u records\n", i, i & ((1u << g.log_sub) - 1u), i >> g.log_sub, cur[i]);
        ++shown;
      }
    for (unsigned int v : src) ssum += v, smx = std::max(smx, v);
    SHK_TRACEF("absorb: segment holds %llu records (fullest region %u of cap %u); page cursors: sum %llu min %u max %u, %u over cap %u; spill_count %llu (list cap %llu)\n",
               ssum, smx, lay->region_cap, sum, mn, mx, over, c->acc_cap, spilled, (unsigned long long)spill_cap);
  }
  c->acc_active = true;
  acc_book(c, est, -1, lane_bound);
  c->n_absorbs++;
  c->unsettled = true;
  c->unsettled_spill_cap = spill_cap;
  return SHK_OK;
}

int shk_xchg_spill(shk_ctx *c, void **d_kmers, void **d_lanes, void **d_counts, uint64_t *n) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c || !n) return SHK_ERR_BAD_ARG;
  HIPC(c, hipSetDevice(c->cfg.device));
  int rc = read_stats(c);
  if (rc != SHK_OK) return rc;
  const SpillRef sp = spill_ref(c->xspill, c->xspill_cap);
  if (d_kmers) *d_kmers = sp.keys;
  if (d_lanes) *d_lanes = sp.lanes;
  if (d_counts) *d_counts = sp.counts;
  *n = c->h_stats->scratch[0];
  return SHK_OK;
}

int shk_xchg_spill_clear(shk_ctx *c) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  HIPC(c, hipSetDevice(c->cfg.device));
  HIPC(c, hipMemsetAsync(&c->d_stats->scratch[0], 0, sizeof(unsigned long long), c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  c->h_stats->scratch[0] = 0;
  return SHK_OK;
}

// A LIST of k-mer occurrences (count 1 each, with their chunk lanes) through the paged passes instead of the global
// atomics of k_insert: what the receiver of a wide exchange round holds (k > 21 between owner shares: 8-byte k-mers
// cross the links).  The list is partitioned like a batch of reads — level 1 by k_part_rescatter in its list mode (one
// pass per chunk lane: k-mers of other lanes, and of other owners, are skipped), level 2 into the waiting (lane, page)
// regions — and counted by the page pass the window's budget or a finalize asks for.  A k = 31 share of 300 M
// distinct k-mers absorbed 14 G k-mers/s through k_insert; see DESIGN.md §6 for what it does this way.
static bool insert_list_paged_ok(const shk_ctx *c, const void *d_counts, uint64_t n) {
  if (d_counts || n < (uint64_t)env_int("SHK_INSERT_PAGED_MIN", 1 << 20) || env_int("SHK_INSERT_PAGED", 1) == 0) return false;
  if (!paged_feasible(c) || (c->cfg.flags & SHK_FLAG_FORCE_DIRECT) || env_int("SHK_DEFER", 1) == 0) return false;
  const PartGeom g = part_geom(c);
  if (use_rec32(c, g) || !g.two_level || c->tb.log_pages < 8) return false;  // (4-byte records have the owner layout; a table of one level is small)
  if (g.log_p1 - g.lw > 10 || g.log_sub > 10) return false;  // (fan-outs of ≤ 1024: the kernel's LDS stays below 64 KiB)
  return n < (1ull << 32);
}
static int insert_list_paged(shk_ctx *c, const uint64_t *d_kmers, const uint32_t *d_lanes, uint64_t n) {
  int rc = settle_light(c);  // the last launch's outcome (its spills would be overwritten by this one's)
  if (rc != SHK_OK) return rc;
  const uint32_t NL = c->n_lanes;
  // (every lane's regions may get all of the list: nothing is known about its lanes)
  rc = acc_prepare(c, n, NL == 1 ? 0 : -1, n);
  if (rc != SHK_OK) return rc;
  if (!insert_list_paged_ok(c, nullptr, n)) return SHK_ERR_STATE;  // (a flush in acc_prepare grew the table out of the route: the caller falls back)
  const PartGeom g = part_geom(c);
  const uint32_t lp = g.lp, n_pages = g.n_pages, S1_log = g.log_p1 - g.lw, S1 = 1u << S1_log, log_sub = g.log_sub, S2 = 1u << log_sub;
  const uint32_t tiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
  const uint32_t cap1 = (region_cap(n, S1, tiles) + 1u) & ~1u;
  HIPC(c, c->part.ensure((uint64_t)S1 * cap1 * 8));
  HIPC(c, c->part_meta.ensure(std::max<size_t>(cursor_buf_bytes(c, g, false), (size_t)S1 * 4 + 64)));
  unsigned int *cursor1 = (unsigned int *)c->part_meta.p;
  const uint64_t spill_cap = std::max<uint64_t>(c->acc_spill_cap, n);
  c->acc_spill_cap = spill_cap;
  HIPC(c, c->spillA.ensure(spill_cap * 16));
  SpillRef sp = spill_ref(c->spillA, spill_cap);
  HIPC(c, hipMemsetAsync(&c->d_stats->spill_count, 0, sizeof(unsigned long long), c->stream));
  auto lds_for = [](uint32_t S) { return (size_t)RS_TILE * 8 + (((size_t)RS_TILE + S) * 2 + 15) / 16 * 16 + (size_t)S * 12; };
  const uint32_t tiles_per_region = (cap1 + RS_TILE - 1) / RS_TILE;
  for (uint32_t lane = 0; lane < NL; ++lane) {
    HIPC(c, hipMemsetAsync(cursor1, 0, (size_t)S1 * 4, c->stream));
    RescatterList ls{};
    ls.lanes = NL > 1 ? d_lanes : nullptr;
    ls.n = n;
    ls.sub_shift = log_sub;
    ls.owner_bits = g.lw;
    ls.owner_id = c->tb.owner_id;
    {
      ScopedTimer t(c, SHK_K_SCATTER);  // level 1: list → the share's super-page regions
      hipLaunchKernelGGL(k_part_rescatter, dim3(tiles), dim3(RS_NT), lds_for(S1), c->stream, d_kmers, (const unsigned int *)nullptr, 0u, tiles, lp, S1_log,
                         2 * c->cfg.k, cursor1, cap1, (uint64_t *)c->part.p, lane, c->d_stats, sp, 0u, ls);
    }
    {
      ScopedTimer t(c, SHK_K_PSCAN, /*chain=*/true);  // level 2: → this lane's waiting page regions
      RescatterList l2{};
      l2.owner_bits = g.lw;
      l2.owner_id = c->tb.owner_id;
      hipLaunchKernelGGL(k_part_rescatter, dim3(S1 * tiles_per_region), dim3(RS_NT), lds_for(S2), c->stream, (const uint64_t *)c->part.p,
                         (const unsigned int *)cursor1, cap1, tiles_per_region, lp, log_sub, 2 * c->cfg.k,
                         (unsigned int *)c->acc_cur.p + (size_t)lane * n_pages, c->acc_cap,
                         (uint64_t *)c->acc_buf.p + (size_t)lane * n_pages * c->acc_cap, lane, c->d_stats, sp, 0u, l2);
    }
  }
  HIPC(c, hipGetLastError());
  c->acc_active = true;
  acc_book(c, n, NL == 1 ? 0 : -1, n);
  c->unsettled = true;
  c->unsettled_spill_cap = spill_cap;
  return SHK_OK;
}

int shk_insert_device(shk_ctx *c, const void *d_kmers, const void *d_lanes, const void *d_counts, uint64_t n) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  if (c->poisoned) return fail(c, c->poison_code, "%s", c->err.c_str());
  HIPC(c, hipSetDevice(c->cfg.device));
  if (d_kmers && insert_list_paged_ok(c, d_counts, n) && (d_lanes || c->n_lanes == 1)) {
    fused_drop(c);
    c->finalized = c->hist_ready = false;
    const int rcp = insert_list_paged(c, (const uint64_t *)d_kmers, (const uint32_t *)d_lanes, n);
    if (rcp == SHK_OK) {
      c->n_inserted += n;
      return SHK_OK;
    }
    if (rcp != SHK_ERR_STATE) return rcp;  // (SHK_ERR_STATE: not this way after all — the general path below)
  }
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  {
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  if (n == 0) return SHK_OK;
  if (!d_kmers) return fail(c, SHK_ERR_BAD_ARG, "null k-mers");
  c->finalized = c->hist_ready = false;
  int rc = ensure_capacity(c, n >> c->owner_bits);
  if (rc != SHK_OK) return rc;
  c->n_inserted += n;
  HIPC(c, c->spillA.ensure(n * 16));
  SpillRef sp = spill_ref(c->spillA, n);
  HIPC(c, hipMemsetAsync(&c->d_stats->spill_count, 0, sizeof(unsigned long long), c->stream));
  {
    ScopedTimer t(c, SHK_K_INSERT);
    hipLaunchKernelGGL(k_insert, dim3(grid_for(n, WG, 4096)), dim3(WG), 0, c->stream, (const uint64_t *)d_kmers,
                       (const uint32_t *)d_lanes, (const uint32_t *)d_counts, n, 0u, c->tb, c->d_stats, sp);
  }
  rc = read_stats(c);
  if (rc != SHK_OK) return rc;
  return drain_spill(c, n);
}

void *shk_stream(shk_ctx *c) { return c && !c->group ? (void *)c->stream : nullptr; }

int shk_sync(shk_ctx *c) {
  if (!c) return SHK_ERR_BAD_ARG;
  if (c->group) {
    for (uint32_t d = 0; d < c->group->D; ++d) {
      const int rc = shk_sync(c->group->ctx[d]);
      if (rc != SHK_OK) return group_fail(c, c->group, rc, d);
    }
    return SHK_OK;
  }
  HIPC(c, hipSetDevice(c->cfg.device));
  HIPC(c, hipStreamSynchronize(c->stream));
  if (c->poisoned) return fail(c, c->poison_code, "%s", c->err.c_str());
  return settle(c);
}

// The front half of finalize: everything up to and including the launch of the histogram scan.
static int finalize_scan(shk_ctx *c) {
  if (c->poisoned) return fail(c, c->poison_code, "%s", c->err.c_str());
  HIPC(c, hipSetDevice(c->cfg.device));
  uint64_t n_reads = 0;
  for (auto v : c->lane_reads) n_reads += v;
  if (n_reads == 0 && c->n_inserted == 0 && !c->own_set && !c->is_share())  // io.rs:578-580 (owner shares: the caller looks at the sum over the shares)
    return fail(c, SHK_ERR_NO_READS,
                "No reads were ingested. Check that input files contain valid FASTQ records.");
  if (c->acc_active) {  // records still waiting for their page pass
    c->flush_for_finalize = true;
    int rcf = settle(c);
    c->flush_for_finalize = false;
    if (rcf != SHK_OK) return rcf;
  }
  // The scan below is queued optimistically behind a counting launch nobody has looked at yet; if that launch
  // spilled, the scan is repeated over the repaired table.  A context whose LAST finalize had to do that (the
  // same input shape tends to spill the same few records again: one k-mer ≥ 7 buckets from home is enough)
  // looks first this time — one host round trip instead of a second scan of the table.
  if (c->unsettled && c->finalize_redone) {
    const uint64_t before = c->n_spilled;
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
    c->finalize_redone = c->n_spilled != before;  // (nothing spilled this time: back to the optimistic order)
  }
  // the histogram the (one) fresh page pass left behind, if the table is still as that pass wrote it
  const bool fused = c->fused_valid && !c->tb_stale && !c->own_set && !c->zero_count_keys && c->fused_pages == (1u << c->tb.log_pages);
  if (!fused) {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  const uint32_t n_cols = c->cfg.chunks;
  const uint64_t hlen = c->cfg.histo_max + 2;
  if (fused) {
    // (d_tot is zero and d_hist holds the pass's bins ≥ FH_BINS only: both were cleared before that pass — by the
    // reset, or by the k_mark_starts of its ingest — and nobody has written them since, or fused_valid would be off)
    ScopedTimer t(c, SHK_K_HISTO_ROWS);
    const uint32_t n_pages = c->fused_pages;
    const uint32_t gx = std::max<uint32_t>((n_cols * FH_BINS + WG - 1) / WG, 1), gy = std::min<uint32_t>(std::max<uint32_t>(n_pages / 8, 1), 1024);  // (eight rows per thread up to 8 Ki pages; 1024 slices beyond)
    hipLaunchKernelGGL(k_hist_reduce, dim3(gx, gy), dim3(WG), 0, c->stream, (const uint32_t *)c->fh_partial.p, (const unsigned long long *)c->fh_tot.p, n_pages,
                       n_cols, (unsigned long long)hlen, c->d_hist, c->d_tot);
    c->fused_valid = false;  // (its bins are in the histogram now: a second scan would add them again)
    c->hist_dirty = false;
    c->fin_scanned = true;
    c->fin_was_unsettled = c->unsettled;
    c->fin_summed = false;
    return SHK_OK;
  }
  if (c->hist_dirty) {  // nothing has come between the last read-back and this scan: histogram + totals back to zero now
    FillSegs f{};
    f.ptr[0] = c->d_tot;
    f.n16[0] = sizeof(HistoTotals) / 16;
    f.ptr[1] = c->d_hist;
    f.n16[1] = (c->ctl_bytes - c->ctl_hist_off) / 16;
    hipLaunchKernelGGL(k_fill, dim3(grid_for(f.n16[0] + f.n16[1], WG * 4, 1024)), dim3(WG), 0, c->stream, f);
    c->hist_dirty = false;
  }
  // d_hist and d_tot are zero here
  uint64_t s0 = 0, s1 = c->tb.cap;
  if (c->own_set) {
    own_resolve(c);
    s0 = c->own_p0 << PAGE_LOG;
    s1 = c->own_p1 << PAGE_LOG;
  }
  uint32_t lds_bins = n_cols ? std::max<uint32_t>(32, 16384 / n_cols) : 0;
  if (lds_bins > hlen) lds_bins = (uint32_t)hlen;
  if (n_cols && (uint64_t)lds_bins * n_cols * 4 > 65536) lds_bins = 65536 / 4 / n_cols;
  {
    ScopedTimer t(c, SHK_K_HISTO);
    if (c->zero_count_keys)  // a key may sit in the table with count 0: occupancy has to come from the keys
      hipLaunchKernelGGL(k_histo<true>, dim3(grid_for(s1 - s0, HISTO_WG * 16, 256)), dim3(HISTO_WG),
                         (size_t)lds_bins * n_cols * 4, c->stream, c->tb, s0, s1, c->cfg.histo_max,
                         n_cols, lds_bins, c->d_hist, c->d_tot);
    else
      hipLaunchKernelGGL(k_histo<false>, dim3(grid_for(s1 - s0, HISTO_WG * 16, 256)), dim3(HISTO_WG),
                         (size_t)lds_bins * n_cols * 4, c->stream, c->tb, s0, s1, c->cfg.histo_max,
                         n_cols, lds_bins, c->d_hist, c->d_tot);
  }
  c->fin_scanned = true;
  c->fin_was_unsettled = c->unsettled;
  c->fin_summed = false;
  return SHK_OK;
}

// The back half: one copy brings back the whole control block (launch outcome, totals, non-N base counts,
// histogram), and one host sync serves both the last counting launch and the scan.
static int finalize_fetch(shk_ctx *c) {
  if (c->ctl_bytes <= (1u << 20) && env_int("SHK_CTL_OUT_KERNEL", 1) != 0) {  // (a few columns: a kernel's stores over the link; more: the copy engine)
    memset(c->h_ctl + c->ctl_hist_off, 0, c->ctl_bytes - c->ctl_hist_off);  // (nobody reads the host copy between here and the wait below)
    const uint32_t n8 = (uint32_t)(c->ctl_bytes / 8);
    hipLaunchKernelGGL(k_ctl_out, dim3((n8 + WG - 1) / WG), dim3(WG), 0, c->stream, (const unsigned long long *)c->d_ctl, (unsigned long long *)c->h_ctl, n8,
                       (uint32_t)(c->ctl_hist_off / 8));
  } else {
    HIPC(c, hipMemcpyAsync(c->h_ctl, c->d_ctl, c->ctl_bytes, hipMemcpyDeviceToHost, c->stream));
  }
  HIPC(c, hipEventRecord(c->done_ev, c->stream));  // the host waits for the copy, not for what follows it
  c->hist_dirty = true;  // (cleared by whoever comes next: reset, an ingest, or the next scan)
  {
    // poll first: the blocking wait's wake-up alone costs the host tens of µs, a tenth of a whole
    // 1 M-read job; a run that takes longer than the poll window falls through to the blocking wait
    hipError_t q = hipErrorNotReady;
    for (int spin = 0; spin < 20000 && (q = hipEventQuery(c->done_ev)) == hipErrorNotReady; ++spin) {
    }
    if (q == hipErrorNotReady) q = hipEventSynchronize(c->done_ev);
    HIPC(c, q);
  }
  c->h_tot = *c->h_totp;
  {  // a fused page pass's adds past its LDS bins ride in the upper half of the saturation word; many of them (deep coverage: same-line
     // global adds, one after the other) and this context's later jobs leave the histogram to k_histo
    const unsigned long long n_high = c->h_tot.any_saturated >> 32;
    c->h_tot.any_saturated = (c->h_tot.any_saturated & 0xFFFFFFFFull) != 0;
    if (n_high > (1ull << 16)) c->fused_off = true;
  }
  c->fin_scanned = false;
  return SHK_OK;
}

int shk_finalize(shk_ctx *c) {
  if (!c) return SHK_ERR_BAD_ARG;
  if (c->group) return group_finalize(c);
  if (c->poisoned) return fail(c, c->poison_code, "%s", c->err.c_str());
  if (c->finalized) return SHK_OK;
  if (c->fin_scanned) return fail(c, SHK_ERR_STATE, "shk_finalize between shk_finalize_begin and shk_finalize_end");
  int rc = finalize_scan(c);
  if (rc != SHK_OK) return rc;
  // The scan was queued optimistically; if the launch before it turns out to have spilled records (or hit an
  // invalid byte) it is settled now and the scan repeated over the repaired table.
  rc = finalize_fetch(c);
  if (rc != SHK_OK) return rc;
  if (c->fin_was_unsettled) {
    const bool redo = c->h_stats->bad != ~0ull || c->h_stats->spill_count > 0;
    int rcs = settle_checked(c);
    if (rcs != SHK_OK) return rcs;
    c->finalize_redone = redo;
    if (redo) return shk_finalize(c);
  }
  c->hist_ready = true;
  const uint32_t n_cols = c->cfg.chunks;
  const uint64_t hlen = c->cfg.histo_max + 2;
  if (!c->own_set) {
    // io.rs:1042-1047 (and :1150-1155 for chunks==0)
    if (c->h_tot.n_hashed != c->h_tot.n_lane_sum)
      return fail(c, SHK_ERR_INVARIANT,
                  "The total count of hashed kmers (%llu) does not equal the number of ingested kmers (%llu)",
                  (unsigned long long)c->h_tot.n_hashed, (unsigned long long)c->h_tot.n_lane_sum);
    if (n_cols > 0) {
      // io.rs:1114-1132: histogram totals of the last column vs the table
      const uint64_t *last = c->h_hist + (size_t)(n_cols - 1) * hlen;
      uint64_t nu = 0;
      for (uint64_t i = 1; i < hlen; ++i) nu += last[i];
      if (nu != c->h_tot.n_unique)
        return fail(c, SHK_ERR_INVARIANT,
                    "The total count of unique kmers in the histogram (%llu) does not equal the total count of hashed kmers (%llu)",
                    (unsigned long long)nu, (unsigned long long)c->h_tot.n_unique);
    }
  }
  c->finalized = true;
  return SHK_OK;
}

int shk_finalize_begin(shk_ctx *c, uint64_t user_word, void **d_sum, uint64_t *n_words) {
  if (!c || !d_sum || !n_words) return SHK_ERR_BAD_ARG;
  if (c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  c->finalized = c->hist_ready = false;
  int rc = finalize_scan(c);
  if (rc != SHK_OK) return rc;
  uint64_t n_reads = 0;
  for (auto v : c->lane_reads) n_reads += v;
  hipLaunchKernelGGL(k_fin_extras, dim3(1), dim3(64), 0, c->stream, c->d_extra, (unsigned long long)n_reads,
                     (unsigned long long)c->n_bases_read, (unsigned long long)user_word, (const DevStats *)c->d_stats,
                     c->fin_was_unsettled ? 1u : 0u, (const unsigned long long *)c->d_lane_bases, c->d_lane_sum, c->n_lanes);
  HIPC(c, hipGetLastError());
  *d_sum = c->d_tot;
  *n_words = (c->ctl_bytes - c->ctl_tot_off) / 8;
  return SHK_OK;
}

int shk_finalize_end(shk_ctx *c, int *again, uint64_t *user_sum) {
  if (!c || !again) return SHK_ERR_BAD_ARG;
  if (c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c->fin_scanned) return fail(c, SHK_ERR_STATE, "shk_finalize_end without shk_finalize_begin");
  *again = 0;
  int rc = finalize_fetch(c);
  if (rc != SHK_OK) return rc;
  if (c->fin_was_unsettled) {
    const bool redo = c->h_stats->bad != ~0ull || c->h_stats->spill_count > 0;
    int rcs = settle_checked(c);  // (repairs this context's table if it was the one)
    if (rcs != SHK_OK) return rcs;
    c->finalize_redone = redo;
  }
  if (user_sum) *user_sum = c->h_extra[3];
  if (c->h_extra[2] > 0) {  // somebody's scan — the same sum on every rank — ran over an incomplete table
    *again = 1;
    return SHK_OK;
  }
  c->fin_summed = true;
  c->hist_ready = true;
  c->finalized = true;
  return SHK_OK;
}

int shk_histograms(shk_ctx *c, uint64_t *out) {
  if (!c) return SHK_ERR_BAD_ARG;
  if (c->group) {
    if (!c->group->finalized && !c->group->hist_ready) return fail(c, SHK_ERR_STATE, "shk_histograms before shk_finalize");
    if (c->cfg.chunks == 0) return SHK_OK;
    if (!out) return fail(c, SHK_ERR_BAD_ARG, "null output");
    memcpy(out, c->group->hist.data(), c->group->hist.size() * sizeof(uint64_t));
    return SHK_OK;
  }
  if (!c->finalized && !c->hist_ready) return fail(c, SHK_ERR_STATE, "shk_histograms before shk_finalize");
  if (c->cfg.chunks == 0) return SHK_OK;
  if (!out) return fail(c, SHK_ERR_BAD_ARG, "null output");
  memcpy(out, c->h_hist, (size_t)c->cfg.chunks * (c->cfg.histo_max + 2) * sizeof(uint64_t));
  return SHK_OK;
}

int shk_get_counters(shk_ctx *c, shk_counters *o) {
  if (!c || !o) return SHK_ERR_BAD_ARG;
  if (c->group) return group_counters(c, o);
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  memset(o, 0, sizeof *o);
  for (auto v : c->lane_reads) o->n_reads_ingested += v;
  o->n_bases_read = c->n_bases_read;
  if (c->fin_summed && c->finalized) {  // (shk_finalize_end: what came back is the sum the caller's reduction left)
    o->n_reads_ingested = c->h_extra[0];
    o->n_bases_read = c->h_extra[1];
  }
  if (!c->finalized && !c->hist_ready) {  // (finalize has just brought the whole control block back)
    HIPC(c, hipMemcpyAsync(c->h_lane_bases, c->d_lane_bases, sizeof(unsigned long long) * c->n_lanes,
                           hipMemcpyDeviceToHost, c->stream));
    HIPC(c, hipStreamSynchronize(c->stream));
  }
  {  // (after shk_finalize_end: the job-wide sums the reduction left; else this context's own counters)
    const unsigned long long *lb = c->fin_summed && c->finalized ? c->h_lane_sum : c->h_lane_bases;
    for (uint32_t l = 0; l < c->n_lanes; ++l) o->n_bases_ingested += lb[l];
  }
  if (c->finalized || c->hist_ready) {
    o->n_kmers_ingested = c->h_tot.n_lane_sum;
    o->n_unique_kmers = c->h_tot.n_unique;
    o->n_hashed_kmers = c->h_tot.n_hashed;
    o->any_saturated = (uint32_t)c->h_tot.any_saturated;
    if (c->cfg.chunks > 0)
      o->n_singleton_kmers = c->h_hist[(size_t)(c->cfg.chunks - 1) * (c->cfg.histo_max + 2) + 1];
  }
  o->n_chunks = c->n_lanes;
  o->table_capacity = c->tb.cap;
  o->n_grows = c->n_grows;
  o->n_spilled = c->n_spilled;
  return SHK_OK;
}

int shk_get_timings(shk_ctx *c, shk_timings *o) {
  if (!c || !o) return SHK_ERR_BAD_ARG;
  if (c->group) {  // summed over the devices
    memset(o, 0, sizeof *o);
    for (uint32_t d = 0; d < c->group->D; ++d) {
      shk_timings t{};
      (void)shk_get_timings(c->group->ctx[d], &t);
      for (int i = 0; i < SHK_N_KERNELS; ++i) o->ms[i] += t.ms[i], o->launches[i] += t.launches[i];
    }
    return SHK_OK;
  }
  (void)hipSetDevice(c->cfg.device);
  resolve_timings(c);
  *o = c->timings;
  return SHK_OK;
}

int shk_reset_timings(shk_ctx *c) {
  if (!c) return SHK_ERR_BAD_ARG;
  if (c->group) {
    for (uint32_t d = 0; d < c->group->D; ++d) (void)shk_reset_timings(c->group->ctx[d]);
    return SHK_OK;
  }
  (void)hipSetDevice(c->cfg.device);
  resolve_timings(c);
  memset(&c->timings, 0, sizeof c->timings);
  c->job_idx = 0;
  c->timing_now = true;
  return SHK_OK;
}

int shk_export_table(shk_ctx *c, uint64_t *kmers, uint32_t *counts, uint64_t cap, uint64_t *n_out) {
  if (!c || !n_out) return SHK_ERR_BAD_ARG;
  if (c->group) {  // the shares are disjoint: one after the other
    uint64_t at = 0;
    for (uint32_t d = 0; d < c->group->D; ++d) {
      uint64_t n = 0;
      const uint64_t room = at < cap ? cap - at : 0;
      const int rc = shk_export_table(c->group->ctx[d], kmers ? kmers + std::min(at, cap) : nullptr,
                                      counts ? counts + std::min(at, cap) : nullptr, room, &n);
      if (rc != SHK_OK) return group_fail(c, c->group, rc, d);
      at += n;
    }
    *n_out = at;
    return SHK_OK;
  }
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  HIPC(c, c->misc.ensure(cap * 12 + 16));
  uint8_t *p = (uint8_t *)c->misc.p;
  unsigned long long *dn = (unsigned long long *)p;
  uint64_t *dk = (uint64_t *)(p + 16);
  uint32_t *dc = (uint32_t *)(p + 16 + cap * 8);
  HIPC(c, hipMemsetAsync(dn, 0, 8, c->stream));
  uint64_t s0 = 0, s1 = c->tb.cap;
  if (c->own_set) {
    own_resolve(c);
    s0 = c->own_p0 << PAGE_LOG;
    s1 = c->own_p1 << PAGE_LOG;
  }
  {
    ScopedTimer t(c, SHK_K_EXPORT);
    hipLaunchKernelGGL(k_export, dim3(grid_for(s1 - s0, WG * 4, 4096)), dim3(WG), 0, c->stream, c->tb, s0,
                       s1, dk, dc, cap, dn);
  }
  unsigned long long n = 0;
  HIPC(c, hipMemcpyAsync(&n, dn, 8, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  *n_out = n;
  uint64_t m = std::min<uint64_t>(n, cap);
  if (m && kmers) HIPC(c, hipMemcpy(kmers, dk, m * 8, hipMemcpyDeviceToHost));
  if (m && counts) HIPC(c, hipMemcpy(counts, dc, m * 4, hipMemcpyDeviceToHost));
  return SHK_OK;
}

int shk_lookup(shk_ctx *c, const uint64_t *kmers, uint32_t *counts, uint64_t n, int canonical) {
  if (c && c->group) {  // exactly one share owns a k-mer; the others answer 0
    std::vector<uint32_t> part(n);
    std::fill(counts, counts + n, 0u);
    for (uint32_t d = 0; d < c->group->D; ++d) {
      const int rc = shk_lookup(c->group->ctx[d], kmers, part.data(), n, canonical);
      if (rc != SHK_OK) return group_fail(c, c->group, rc, d);
      for (uint64_t i = 0; i < n; ++i) counts[i] += part[i];
    }
    return SHK_OK;
  }
  if (!c) return SHK_ERR_BAD_ARG;
  if (n == 0) return SHK_OK;
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  HIPC(c, c->misc.ensure(n * 12));
  uint64_t *dk = (uint64_t *)c->misc.p;
  uint32_t *dc = (uint32_t *)((uint8_t *)c->misc.p + n * 8);
  HIPC(c, hipMemcpyAsync(dk, kmers, n * 8, hipMemcpyHostToDevice, c->stream));
  {
    ScopedTimer t(c, SHK_K_LOOKUP);
    hipLaunchKernelGGL(k_lookup, dim3((uint32_t)((n + WG - 1) / WG)), dim3(WG), 0, c->stream, c->tb, dk, dc,
                       n, canonical, (int)c->cfg.k);
  }
  HIPC(c, hipMemcpyAsync(counts, dc, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  return SHK_OK;
}

int shk_find_oligos(shk_ctx *c, const uint64_t *oligos, uint32_t n_oligos, uint32_t oligo_len,
                    uint32_t min_count, uint64_t *kmers, uint32_t *counts, uint64_t cap, uint64_t *n_out) {
  if (c && c->group) {  // the shares are disjoint: one after the other
    uint64_t at = 0;
    for (uint32_t d = 0; d < c->group->D; ++d) {
      uint64_t n = 0;
      const uint64_t room = at < cap ? cap - at : 0;
      const int rc = shk_find_oligos(c->group->ctx[d], oligos, n_oligos, oligo_len, min_count, kmers ? kmers + std::min(at, cap) : nullptr,
                                     counts ? counts + std::min(at, cap) : nullptr, room, &n);
      if (rc != SHK_OK) return group_fail(c, c->group, rc, d);
      at += n;
    }
    if (n_out) *n_out = at;
    return SHK_OK;
  }
  if (!c || !n_out) return SHK_ERR_BAD_ARG;
  const uint32_t k = c->cfg.k;
  // the reference asserts these (primers.rs:169-186)
  if (n_oligos == 0 || !oligos) return fail(c, SHK_ERR_BAD_ARG, "find_oligos_in_kmers called with no oligos");
  if (!(oligo_len > 0 && oligo_len < k))
    return fail(c, SHK_ERR_BAD_ARG, "oligo length %u out of range for k=%u (must be 1..k-1); trim must be < k",
                oligo_len, k);
  if (n_oligos > 3000) return fail(c, SHK_ERR_BAD_ARG, "too many oligos (%u > 3000)", n_oligos);
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  auto rc_of = [](uint64_t x, int len) {  // reverse complement of a len-base value (host side)
    uint64_t r = 0;
    for (int i = 0; i < len; ++i) {
      r = (r << 2) | (3 - (x & 3));
      x >>= 2;
    }
    return r;
  };
  std::vector<uint64_t> fwd(n_oligos), rc(n_oligos);
  for (uint32_t i = 0; i < n_oligos; ++i) {
    fwd[i] = oligos[i] << (2 * (k - oligo_len));  // primers.rs:189-192
    rc[i] = rc_of(oligos[i], (int)oligo_len);      // primers.rs:206-209
  }
  std::sort(fwd.begin(), fwd.end());
  std::sort(rc.begin(), rc.end());
  HIPC(c, c->misc.ensure((size_t)n_oligos * 16 + cap * 12 + 64));
  uint8_t *p = (uint8_t *)c->misc.p;
  unsigned long long *dn = (unsigned long long *)p;
  uint64_t *dsets = (uint64_t *)(p + 16);
  uint64_t *dk = dsets + 2 * (size_t)n_oligos;
  uint32_t *dc = (uint32_t *)(dk + cap);
  HIPC(c, hipMemsetAsync(dn, 0, 8, c->stream));
  HIPC(c, hipMemcpyAsync(dsets, fwd.data(), (size_t)n_oligos * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemcpyAsync(dsets + n_oligos, rc.data(), (size_t)n_oligos * 8, hipMemcpyHostToDevice, c->stream));
  uint64_t s0 = 0, s1 = c->tb.cap;
  if (c->own_set) {
    own_resolve(c);
    s0 = c->own_p0 << PAGE_LOG;
    s1 = c->own_p1 << PAGE_LOG;
  }
  {
    ScopedTimer t(c, SHK_K_LOOKUP);
    hipLaunchKernelGGL(k_find_oligos, dim3(grid_for(s1 - s0, WG * 8, 2048)), dim3(WG), (size_t)n_oligos * 16,
                       c->stream, c->tb, s0, s1, (int)k, (int)oligo_len, min_count, (const uint64_t *)dsets,
                       (const uint64_t *)(dsets + n_oligos), n_oligos, dk, dc, cap, dn);
  }
  unsigned long long n = 0;
  HIPC(c, hipMemcpyAsync(&n, dn, 8, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));  // also keeps fwd/rc alive until their copies ran
  *n_out = n;
  uint64_t m = std::min<uint64_t>(n, cap);
  if (m && kmers) HIPC(c, hipMemcpy(kmers, dk, m * 8, hipMemcpyDeviceToHost));
  if (m && counts) HIPC(c, hipMemcpy(counts, dc, m * 4, hipMemcpyDeviceToHost));
  return SHK_OK;
}

int shk_filter_reads(shk_ctx *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_seqs,
                     const uint64_t *primer_kmers, uint64_t n_kmers, uint8_t *out_matches) {
  if (c && c->group) {  // stateless: any device will do
    const int rc = shk_filter_reads(c->group->ctx[0], bases, offsets, n_seqs, primer_kmers, n_kmers, out_matches);
    return rc == SHK_OK ? rc : group_fail(c, c->group, rc, 0);
  }
  if (!c || (n_seqs && (!offsets || !out_matches))) return SHK_ERR_BAD_ARG;
  if (n_seqs == 0) return SHK_OK;
  HIPC(c, hipSetDevice(c->cfg.device));
  const uint32_t k = c->cfg.k;
  // the union of the primer k-mers (read_filter.rs:24-41) as an open-addressing set at load ≤ 1/2
  uint64_t cap = 16;
  while (cap < 2 * n_kmers) cap <<= 1;
  if (cap > (1ull << 31)) return fail(c, SHK_ERR_BAD_ARG, "primer k-mer set too large");
  std::vector<uint64_t> set(cap, ~0ull);
  for (uint64_t j = 0; j < n_kmers; ++j) {
    const uint64_t key = primer_kmers[j];
    if (2 * k < 64 && (key >> (2 * k)) != 0)
      return fail(c, SHK_ERR_BAD_ARG, "primer k-mer %llu does not fit %u bases", (unsigned long long)key, k);
    for (uint64_t sl = set_hash(key) & (cap - 1);; sl = (sl + 1) & (cap - 1)) {
      if (set[sl] == key) break;
      if (set[sl] == ~0ull) {
        set[sl] = key;
        break;
      }
    }
  }
  const uint64_t n_bases = offsets[n_seqs];
  HIPC(c, c->in_bases.ensure(n_bases + 16));
  HIPC(c, c->in_offsets.ensure((n_seqs + 2) * 8));
  HIPC(c, c->misc.ensure(cap * 8 + n_seqs));
  uint64_t *dset = (uint64_t *)c->misc.p;
  uint8_t *dout = (uint8_t *)c->misc.p + cap * 8;
  {
    int rcs = settle(c);  // the staging buffers may still feed a counting launch
    if (rcs != SHK_OK) return rcs;
  }
  HIPC(c, hipStreamSynchronize(c->stream));
  if (n_bases) HIPC(c, hipMemcpyAsync(c->in_bases.p, bases, n_bases, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemcpyAsync(c->in_offsets.p, offsets, (n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemcpyAsync(dset, set.data(), cap * 8, hipMemcpyHostToDevice, c->stream));
  {
    ScopedTimer t(c, SHK_K_LOOKUP);
    hipLaunchKernelGGL(k_filter_reads, dim3((uint32_t)((n_seqs + WG - 1) / WG)), dim3(WG), 0, c->stream,
                       (const uint8_t *)c->in_bases.p, (const uint64_t *)c->in_offsets.p, n_seqs, (int)k,
                       (const uint64_t *)dset, (uint32_t)(cap - 1), dout);
  }
  HIPC(c, hipMemcpyAsync(out_matches, dout, n_seqs, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));  // (also keeps `set` alive until its copy ran)
  return SHK_OK;
}

int shk_kmers_from_reads(shk_ctx *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_seqs,
                         uint64_t *kmers, uint64_t kmers_cap, uint32_t *n_kmers, uint8_t *bad_byte) {
  if (c && c->group) {  // stateless: any device will do
    const int rc = shk_kmers_from_reads(c->group->ctx[0], bases, offsets, n_seqs, kmers, kmers_cap, n_kmers, bad_byte);
    return rc == SHK_OK ? rc : group_fail(c, c->group, rc, 0);
  }
  if (!c || (n_seqs && (!offsets || !n_kmers || !bad_byte))) return SHK_ERR_BAD_ARG;
  if (n_seqs == 0) return SHK_OK;
  HIPC(c, hipSetDevice(c->cfg.device));
  const uint32_t k = c->cfg.k;
  // koff(i): every read gets room for the most k-mers it can yield
  std::vector<uint64_t> koff(n_seqs + 1);
  koff[0] = 0;
  for (uint64_t i = 0; i < n_seqs; ++i) {
    if (offsets[i + 1] < offsets[i]) return fail(c, SHK_ERR_BAD_ARG, "offsets must be non-decreasing");
    const uint64_t len = offsets[i + 1] - offsets[i];
    koff[i + 1] = koff[i] + (len >= k ? len - k + 1 : 0);
  }
  const uint64_t n_total = koff[n_seqs];
  if (n_total > kmers_cap || (n_total && !kmers))
    return fail(c, SHK_ERR_BAD_ARG, "kmers_cap %llu is below the %llu k-mers these reads can yield",
                (unsigned long long)kmers_cap, (unsigned long long)n_total);
  const uint64_t n_bases = offsets[n_seqs];
  HIPC(c, c->in_bases.ensure(n_bases + 16));
  HIPC(c, c->in_offsets.ensure((n_seqs + 2) * 8));
  const uint64_t o_koff = 0, o_kmers = (n_seqs + 1) * 8, o_n = o_kmers + (n_total + 1) * 8,
                 o_bad = o_n + n_seqs * 4;
  HIPC(c, c->misc.ensure(o_bad + n_seqs));
  char *m = (char *)c->misc.p;
  {
    int rcs = settle(c);  // the staging buffers may still feed a counting launch
    if (rcs != SHK_OK) return rcs;
  }
  HIPC(c, hipStreamSynchronize(c->stream));
  if (n_bases) HIPC(c, hipMemcpyAsync(c->in_bases.p, bases, n_bases, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemcpyAsync(c->in_offsets.p, offsets, (n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemcpyAsync(m + o_koff, koff.data(), (n_seqs + 1) * 8, hipMemcpyHostToDevice, c->stream));
  {
    ScopedTimer t(c, SHK_K_LOOKUP);
    hipLaunchKernelGGL(k_kmers_from_reads, dim3((uint32_t)((n_seqs + WG - 1) / WG)), dim3(WG), 0, c->stream,
                       (const uint8_t *)c->in_bases.p, (const uint64_t *)c->in_offsets.p,
                       (const uint64_t *)(m + o_koff), n_seqs, (int)k, (uint64_t *)(m + o_kmers),
                       (uint32_t *)(m + o_n), (uint8_t *)(m + o_bad));
  }
  // a read's span is copied back whole; only its first n_kmers[i] entries mean anything
  if (n_total) HIPC(c, hipMemcpyAsync(kmers, m + o_kmers, n_total * 8, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipMemcpyAsync(n_kmers, m + o_n, n_seqs * 4, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipMemcpyAsync(bad_byte, m + o_bad, n_seqs, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));  // (also keeps `koff` alive until its copy ran)
  return SHK_OK;
}

int shk_table_geometry(shk_ctx *c, uint64_t *n_pages, uint32_t *page_slots, uint32_t *n_lanes) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcs = settle(c);  // a pending spill may still grow the table
    if (rcs != SHK_OK) return rcs;
  }
  if (n_pages) *n_pages = 1ull << c->tb.log_pages;
  if (page_slots) *page_slots = PAGE_SLOTS;
  if (n_lanes) *n_lanes = c->n_lanes;
  return SHK_OK;
}

int shk_table_reserve_pages(shk_ctx *c, uint64_t n_pages) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  uint32_t lp = 0;
  while ((1ull << lp) < n_pages) lp++;
  c->finalized = c->hist_ready = false;
  return grow_to(c, lp);
}

int shk_table_device_ptrs(shk_ctx *c, void **d_keys, void **d_vals) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  HIPC(c, hipStreamSynchronize(c->stream));
  if (d_keys) *d_keys = c->tb.keys;
  if (d_vals) *d_vals = c->tb.vals;
  return SHK_OK;
}

int shk_merge_pages(shk_ctx *c, uint64_t p0, uint64_t p1, const void *d_keys, const void *d_vals,
                    uint64_t vals_lane_stride) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  if (p1 <= p0) return SHK_OK;
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  c->finalized = c->hist_ready = false;
  c->zero_count_keys = true;  // (a peer's table may hold keys inserted with count 0: keep reading the keys)
  const uint64_t n_slots = (p1 - p0) << PAGE_LOG;
  // worst case every peer key is new here
  HIPC(c, c->spillA.ensure(n_slots * c->n_lanes * 16));
  SpillRef sp = spill_ref(c->spillA, n_slots * c->n_lanes);
  HIPC(c, hipMemsetAsync(&c->d_stats->spill_count, 0, sizeof(unsigned long long), c->stream));
  {
    ScopedTimer t(c, SHK_K_MERGE);
    hipLaunchKernelGGL(k_merge, dim3(grid_for(n_slots, WG, 8192)), dim3(WG), 0, c->stream, c->tb,
                       n_slots, vals_lane_stride, (const uint64_t *)d_keys, (const uint32_t *)d_vals,
                       c->d_stats, sp, 0ull, ~0u);
  }
  int rc = read_stats(c);
  if (rc != SHK_OK) return rc;
  return drain_spill(c, n_slots * c->n_lanes);
}

int shk_owner_counts(shk_ctx *c, uint32_t n_owners, uint64_t *counts) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c || !counts || n_owners == 0) return SHK_ERR_BAD_ARG;
  const uint64_t n_pages = 1ull << c->tb.log_pages;
  if (n_pages % n_owners) return fail(c, SHK_ERR_BAD_ARG, "%llu pages do not split over %u owners",
                                      (unsigned long long)n_pages, n_owners);
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  HIPC(c, c->misc.ensure((size_t)n_owners * 16));
  unsigned long long *dc = (unsigned long long *)c->misc.p;
  HIPC(c, hipMemsetAsync(dc, 0, (size_t)n_owners * 8, c->stream));
  const uint32_t bpo = std::min<uint32_t>(1024, std::max<uint32_t>(16, 2048 / n_owners));  // blocks per owner
  hipLaunchKernelGGL(k_owner_counts, dim3(n_owners * bpo), dim3(WG), 0, c->stream, c->tb,
                     c->tb.cap / n_owners, bpo, dc);
  HIPC(c, hipMemcpyAsync(counts, dc, (size_t)n_owners * 8, hipMemcpyDeviceToHost, c->stream));
  HIPC(c, hipStreamSynchronize(c->stream));
  return SHK_OK;
}

int shk_compact_owners(shk_ctx *c, uint32_t n_owners, const uint64_t *seg_offsets, void *d_keys, void *d_vals,
                       uint64_t vals_lane_stride, int32_t skip_owner) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c || !seg_offsets || n_owners == 0) return SHK_ERR_BAD_ARG;
  const uint64_t n_pages = 1ull << c->tb.log_pages;
  if (n_pages % n_owners) return fail(c, SHK_ERR_BAD_ARG, "%llu pages do not split over %u owners",
                                      (unsigned long long)n_pages, n_owners);
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  HIPC(c, c->misc.ensure((size_t)n_owners * 16));
  unsigned long long *doff = (unsigned long long *)c->misc.p, *dcur = doff + n_owners;
  HIPC(c, hipMemcpyAsync(doff, seg_offsets, (size_t)n_owners * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemsetAsync(dcur, 0, (size_t)n_owners * 8, c->stream));
  const uint32_t bpo = std::min<uint32_t>(1024, std::max<uint32_t>(16, 2048 / n_owners));  // blocks per owner
  hipLaunchKernelGGL(k_compact_owners, dim3(n_owners * bpo), dim3(WG), 0, c->stream, c->tb,
                     c->tb.cap / n_owners, (const unsigned long long *)doff, dcur, (uint64_t *)d_keys,
                     (uint32_t *)d_vals, vals_lane_stride, skip_owner < 0 ? ~0u : (uint32_t)skip_owner, bpo,
                     (const unsigned long long *)nullptr, 0ull, (unsigned long long *)nullptr);
  HIPC(c, hipStreamSynchronize(c->stream));  // the caller hands the buffers to a collective next
  return SHK_OK;
}

int shk_compact_owners_packed(shk_ctx *c, uint32_t n_owners, const uint64_t *counts, void *d_buf, int32_t skip_owner) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c || !counts || n_owners == 0) return SHK_ERR_BAD_ARG;
  const uint64_t n_pages = 1ull << c->tb.log_pages;
  if (n_pages % n_owners) return fail(c, SHK_ERR_BAD_ARG, "%llu pages do not split over %u owners",
                                      (unsigned long long)n_pages, n_owners);
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  // device: [seg offsets (entries) × W][cursors × W][counts × W], staged through pinned memory
  HIPC(c, c->misc.ensure((size_t)n_owners * 24));
  HIPC(c, c->h_rebased[0].ensure((size_t)n_owners * 16));
  unsigned long long *h = (unsigned long long *)c->h_rebased[0].p;
  unsigned long long run = 0;
  for (uint32_t o = 0; o < n_owners; ++o) {
    h[o] = run;
    h[n_owners + o] = counts[o];
    run += counts[o];
  }
  unsigned long long *doff = (unsigned long long *)c->misc.p, *dcur = doff + n_owners, *dcnt = dcur + n_owners;
  HIPC(c, hipMemcpyAsync(doff, h, (size_t)n_owners * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemcpyAsync(dcnt, h + n_owners, (size_t)n_owners * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemsetAsync(dcur, 0, (size_t)n_owners * 8, c->stream));
  const uint32_t bpo = std::min<uint32_t>(1024, std::max<uint32_t>(16, 2048 / n_owners));  // blocks per owner
  hipLaunchKernelGGL(k_compact_owners, dim3(n_owners * bpo), dim3(WG), 0, c->stream, c->tb,
                     c->tb.cap / n_owners, (const unsigned long long *)doff, dcur, (uint64_t *)d_buf,
                     (uint32_t *)nullptr, 0ull, skip_owner < 0 ? ~0u : (uint32_t)skip_owner, bpo,
                     (const unsigned long long *)dcnt, 0ull, (unsigned long long *)nullptr);
  HIPC(c, hipGetLastError());
  return SHK_OK;  // (asynchronous on the context's stream: run the collective on shk_stream())
}

int shk_compact_owners_fixed(shk_ctx *c, uint32_t n_owners, uint64_t capacity, void *d_buf, int32_t skip_owner) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c || n_owners == 0 || capacity == 0 || !d_buf) return SHK_ERR_BAD_ARG;
  const uint64_t n_pages = 1ull << c->tb.log_pages;
  if (n_pages % n_owners) return fail(c, SHK_ERR_BAD_ARG, "%llu pages do not split over %u owners",
                                      (unsigned long long)n_pages, n_owners);
  HIPC(c, hipSetDevice(c->cfg.device));
  if (c->acc_active) {  // records still waiting for their page pass
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  // (Otherwise nothing is waited for: a counting launch nobody has looked at yet may have spilled records, in which
  // case the table read here is incomplete — k_piece_headers sees that on the device and poisons every header, no
  // rank merges anything, and the finalize that follows repairs the table before the exchange is repeated.)
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  HIPC(c, c->misc.ensure((size_t)n_owners * 24));
  HIPC(c, c->h_rebased[0].ensure((size_t)n_owners * 16));
  unsigned long long *h = (unsigned long long *)c->h_rebased[0].p;
  for (uint32_t o = 0; o < n_owners; ++o) {
    h[o] = (unsigned long long)o * capacity;  // every piece at its fixed place
    h[n_owners + o] = capacity;
  }
  unsigned long long *doff = (unsigned long long *)c->misc.p, *dcur = doff + n_owners, *dcnt = dcur + n_owners;
  const size_t piece_bytes = 8 + capacity * (8 + 4 * (size_t)c->n_lanes);
  HIPC(c, hipMemcpyAsync(doff, h, (size_t)n_owners * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemcpyAsync(dcnt, h + n_owners, (size_t)n_owners * 8, hipMemcpyHostToDevice, c->stream));
  HIPC(c, hipMemsetAsync(dcur, 0, (size_t)n_owners * 8, c->stream));
  // unused places read as EMPTY k-mers, which the merge skips
  HIPC(c, hipMemsetAsync(d_buf, 0xFF, (size_t)n_owners * piece_bytes, c->stream));
  HIPC(c, hipMemsetAsync(&c->d_stats->scratch[1], 0, 16, c->stream));  // [1]: fullest range here, [2]: … anywhere (merge)
  const uint32_t bpo = std::min<uint32_t>(1024, std::max<uint32_t>(16, 2048 / n_owners));  // blocks per owner
  hipLaunchKernelGGL(k_compact_owners, dim3(n_owners * bpo), dim3(WG), 0, c->stream, c->tb,
                     c->tb.cap / n_owners, (const unsigned long long *)doff, dcur, (uint64_t *)d_buf,
                     (uint32_t *)nullptr, 0ull, skip_owner < 0 ? ~0u : (uint32_t)skip_owner, bpo,
                     (const unsigned long long *)dcnt, 8ull, &c->d_stats->scratch[1]);
  hipLaunchKernelGGL(k_piece_headers, dim3((n_owners + 63) / 64), dim3(64), 0, c->stream, (uint32_t *)d_buf, n_owners,
                     (unsigned long long)(piece_bytes / 4), (const DevStats *)c->d_stats);
  HIPC(c, hipGetLastError());
  return SHK_OK;
}

int shk_merge_pieces_max(shk_ctx *c, uint64_t *max_count) {
  if (!c || !max_count) return SHK_ERR_BAD_ARG;
  if (c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  HIPC(c, hipSetDevice(c->cfg.device));
  if (!c->finalized && !c->hist_ready) {  // (a finalize has just brought the control block back otherwise)
    int rc = read_stats(c);
    if (rc != SHK_OK) return rc;
  }
  *max_count = c->h_stats->scratch[2];
  return SHK_OK;
}

static int merge_launch(shk_ctx *c, const void *d_keys, const void *d_vals, uint64_t n, uint64_t vals_lane_stride,
                        uint64_t piece_cap, uint32_t skip_piece);

int shk_merge_entries(shk_ctx *c, const void *d_keys, const void *d_vals, uint64_t n, uint64_t vals_lane_stride) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  if (n == 0) return SHK_OK;
  return merge_launch(c, d_keys, d_vals, n, vals_lane_stride, 0, ~0u);
}

int shk_merge_pieces(shk_ctx *c, const void *d_buf, uint32_t n_pieces, uint64_t capacity, int32_t skip_piece) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c || !d_buf) return SHK_ERR_BAD_ARG;
  if (n_pieces == 0 || capacity == 0) return SHK_OK;
  return merge_launch(c, d_buf, nullptr, (uint64_t)n_pieces * capacity, capacity, capacity, skip_piece < 0 ? ~0u : (uint32_t)skip_piece);
}

static int merge_launch(shk_ctx *c, const void *d_keys, const void *d_vals, uint64_t n, uint64_t vals_lane_stride,
                        uint64_t piece_cap, uint32_t skip_piece) {
  HIPC(c, hipSetDevice(c->cfg.device));
  {
    int rcf = tb_fresh(c);
    if (rcf != SHK_OK) return rcf;
  }
  // Fixed-capacity pieces behind a counting launch nobody has looked at yet: nothing is waited for.  If that
  // launch spilled, the senders' headers are poisoned and k_merge touches nothing; if not, what the merge spills
  // goes on the same list (same capacity, the counter runs on) and the finalize that follows repairs it.
  // (Only when that list could take the merge's own worst case — every entry spilling on every lane, which is what
  // W pieces' worth of new keys do to pages sized for the local shard alone: k_merge drops what does not fit the list,
  // and the settle that follows would fail the job with "spill list overflow" where the exact-count protocol would
  // have finished.  A counting launch's list has a place per k-mer of the launch, so this holds whenever the pieces
  // are no larger than the batch.)
  const bool ride_on = piece_cap && c->unsettled && !c->acc_active && c->unsettled_spill_cap >= n * c->n_lanes &&
                       c->spillA.cap >= c->unsettled_spill_cap * 16;
  if (!ride_on) {
    int rcs = settle(c);
    if (rcs != SHK_OK) return rcs;
  }
  c->finalized = c->hist_ready = false;
  c->zero_count_keys = true;  // (a peer's table may hold keys inserted with count 0: keep reading the keys)
  const uint64_t spill_cap = ride_on ? c->unsettled_spill_cap : n * c->n_lanes;  // worst case every entry spills on every lane
  if (!ride_on) {
    HIPC(c, c->spillA.ensure(spill_cap * 16));
    HIPC(c, hipMemsetAsync(&c->d_stats->spill_count, 0, sizeof(unsigned long long), c->stream));
  }
  SpillRef sp = spill_ref(c->spillA, spill_cap);
  {
    ScopedTimer t(c, SHK_K_MERGE);
    hipLaunchKernelGGL(k_merge, dim3(grid_for(n, WG, 8192)), dim3(WG), 0, c->stream, c->tb, n, vals_lane_stride,
                       (const uint64_t *)d_keys, (const uint32_t *)d_vals, c->d_stats, sp, piece_cap, skip_piece);
  }
  // (nothing is waited for: the outcome — spilled entries, load factor — is looked at by the next call that
  // needs the table, at the latest finalize)
  c->unsettled = true;
  c->unsettled_spill_cap = spill_cap;
  return SHK_OK;
}

int shk_set_owned_pages(shk_ctx *c, uint64_t p0, uint64_t p1) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  if (p1 < p0 || p1 > (1ull << c->tb.log_pages)) return fail(c, SHK_ERR_BAD_ARG, "bad page range");
  c->own_p0 = p0;
  c->own_p1 = p1;
  c->own_set = true;
  c->own_share_n = 0;
  c->finalized = c->hist_ready = false;
  return SHK_OK;
}

int shk_set_owner_share(shk_ctx *c, uint32_t n_owners, uint32_t owner) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c) return SHK_ERR_BAD_ARG;
  if (n_owners == 0 || (n_owners & (n_owners - 1)) || owner >= n_owners || n_owners > (1ull << c->tb.log_pages))
    return fail(c, SHK_ERR_BAD_ARG, "bad owner share %u of %u", owner, n_owners);
  c->own_share_n = n_owners;  // (no settle: the range is worked out from the page count of the moment a scan is launched)
  c->own_share_id = owner;
  c->own_set = true;
  c->finalized = c->hist_ready = false;
  return SHK_OK;
}

void *shk_alloc_pinned(size_t bytes) {
  void *p = nullptr;
  size_t got = 0;
  if (host_alloc(&p, bytes ? bytes : 1, &got) != hipSuccess) return nullptr;
  MemCache &mc = mem_cache();
  std::lock_guard<std::mutex> lk(mc.m);
  mc.pinned_out[p] = got;
  return p;
}
void shk_free_pinned(void *p) {
  if (!p) return;
  size_t bytes = 0;
  {
    MemCache &mc = mem_cache();
    std::lock_guard<std::mutex> lk(mc.m);
    auto it = mc.pinned_out.find(p);
    if (it != mc.pinned_out.end()) {
      bytes = it->second;
      mc.pinned_out.erase(it);
    }
  }
  if (bytes) host_free(p, bytes);
  else (void)hipHostFree(p);
}
void shk_release_cached_memory(void) {
  std::vector<MemCache::Blk> blocks;
  std::vector<std::pair<int, hipStream_t>> streams;
  {
    MemCache &mc = mem_cache();
    std::lock_guard<std::mutex> lk(mc.m);
    blocks.swap(mc.blocks);
    mc.dev_total = mc.host_total = 0;
    streams.swap(mc.streams);
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (const MemCache::Blk &b : blocks) {
    if (b.dev < 0) {
      (void)hipHostFree(b.p);
    } else {
      (void)hipSetDevice(b.dev);
      (void)hipFree(b.p);
    }
  }
  for (auto &st : streams) {
    (void)hipSetDevice(st.first);
    (void)hipStreamDestroy(st.second);
  }
  (void)hipSetDevice(cur);
}
void *shk_alloc_device(shk_ctx *c, size_t bytes) {
  if (c && c->group) return nullptr;
  if (!c) return nullptr;
  (void)hipSetDevice(c->cfg.device);
  void *p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
  return p;
}
void shk_free_device(shk_ctx *c, void *p) {
  if (c && c->group) return;
  if (!c || !p) return;
  (void)hipSetDevice(c->cfg.device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(p);
}

int shk_synth_reads_device(shk_ctx *c, const shk_synth *spec, uint64_t first_read, uint64_t n_reads,
                           void *d_bases, void *d_offsets) {
  if (c && c->group) return fail(c, SHK_ERR_STATE, "not available on a multi-device context");
  if (!c || !spec) return SHK_ERR_BAD_ARG;
  if (spec->read_len == 0 || spec->genome_len < spec->read_len)
    return fail(c, SHK_ERR_BAD_ARG, "bad synth spec");
  HIPC(c, hipSetDevice(c->cfg.device));
  SynthSpec s{spec->seed_genome, spec->seed_reads, spec->genome_len, spec->read_len,
              spec->sub_per_64k, spec->n_per_64k, 0};
  uint64_t total = n_reads * spec->read_len;
  uint64_t items = std::max<uint64_t>(total, n_reads + 1);
  uint64_t threads = (items + 7) / 8;
  {
    ScopedTimer t(c, SHK_K_SYNTH);
    hipLaunchKernelGGL(k_synth, dim3((uint32_t)((threads + WG - 1) / WG)), dim3(WG), 0, c->stream, s,
                       first_read, n_reads, (uint8_t *)d_bases, (uint64_t *)d_offsets);
  }
  HIPC(c, hipStreamSynchronize(c->stream));
  return SHK_OK;
}

}  // extern "C"
