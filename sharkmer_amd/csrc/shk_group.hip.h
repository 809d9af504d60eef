// shk_group.hip.h — the MULTI-DEVICE context (shk_config.n_devices > 1; SURVEY.md §8b `n_devices,
// device_ids`, §8e).  Included by shk_engine.hip; everything here drives the per-device contexts through
// the public C ABI of include/shk.h.
//
// One owner share per device (n_owners = n_devices, owner_id = the device's rank): every device holds 1/D
// of the key space for the whole run, so no table is ever exchanged and a load that does not fit one card
// (BASELINE configs[4]: 10 lanes × a 3 Gb genome) fits D of them.  A host batch is cut into D contiguous
// runs of reads; worker thread d (one per device, started once, never restarted — nothing re-execs after
// the GPU is initialised) copies run d to its card, tells the engine the global index of its first read
// (read i → chunk (i / 1000) % n_chunks, io.rs:340-361, whatever card counts it), runs the level-1 pass
// (shk_xchg_scatter_device) and, after a barrier, pulls its own segment out of every peer's exchange buffer
// (hipMemcpyPeerAsync: the copy engines over xGMI; D−1 pulls per card run concurrently, one per link of
// the fully connected mesh) and absorbs it (shk_xchg_absorb).  Foreign spills (records that overflowed a
// region on skewed input) are pulled and inserted the same way.  At finalize every device scans its own
// share and the host adds the D histograms and totals (bins are additive over disjoint key sets:
// KmerCounts::extend, counting.rs:157-166; io.rs:1023-1047 on the sums).
//
// Multi-PROCESS runs (one process per GPU) do the same rounds over RCCL: sharkmer_amd/dist.py, OwnerCounter.
// When 4-byte records do not fit (k > 21 at the default fan-out) the context cannot be created
// (SHK_ERR_BAD_ARG): use one context per device and merge at finalize (shk_merge_*).
#pragma once

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

struct shk_group {
  uint32_t D = 0;
  shk_config cfg{};
  std::vector<int32_t> dev_ids;
  std::vector<shk_ctx *> ctx;
  // worker threads: one per device, parked between jobs
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  uint64_t generation = 0;
  uint32_t n_done = 0;
  bool quit = false;
  std::function<int(uint32_t)> job;
  std::vector<int> job_rc;
  // barrier used INSIDE jobs (all D workers take part in every barrier of a job)
  std::mutex bm;
  std::condition_variable bcv;
  uint32_t b_count = 0;
  uint64_t b_gen = 0;
  // per-device staging
  std::vector<DevBuf> in_bases, in_offsets, recv_rec, recv_cur, recv_spill;
  std::vector<HostBuf> rebased;
  // k-mers that do not fit the owner layout's 4-byte records (k > 21 at the default fan-out): the WIDE round — whole
  // 64-bit k-mers and their chunk lanes grouped by owner (shk_xchg_wide_scatter_device), every device pulls its piece
  // of every peer's arrays and inserts it (shk_insert_device).  Round 4: such a context used to be refused.
  bool wide = false;
  std::vector<void *> wk_ptr, wl_ptr;
  std::vector<std::vector<uint64_t>> wcounts;  // [sender][owner]
  std::vector<DevBuf> recv_wk, recv_wl;
  // what the workers publish in a round
  std::vector<void *> rec_ptr, cur_ptr;
  std::vector<shk_xchg_layout> lay;
  std::vector<uint64_t> n_foreign, sp_n;
  std::vector<void *> sp_k, sp_l, sp_c;
  std::vector<int> status;
  // run state
  uint64_t next_read = 0;  // global index of the next read (shk_set_read_index)
  bool finalized = false, hist_ready = false;
  std::vector<uint64_t> hist;
  shk_counters tot{};

  // Returns whether any worker had failed WHEN THE LAST ONE ARRIVED — one answer for all of them, taken inside the
  // barrier, so that either every worker leaves the round early or none does (a worker that looked at status[] on its
  // own after waking could see a peer's failure from the NEXT phase, skip the next barrier and leave that peer waiting
  // in it for ever).  A worker writes only its own status[] entry, before it takes the barrier's lock.
  bool b_failed = false;
  bool barrier() {
    std::unique_lock<std::mutex> lk(bm);
    const uint64_t g = b_gen;
    if (++b_count == D) {
      b_count = 0;
      b_failed = any_failed();
      ++b_gen;
      bcv.notify_all();
    } else {
      bcv.wait(lk, [&] { return b_gen != g; });
    }
    return b_failed;
  }
  bool any_failed() const {
    for (int s : status)
      if (s != SHK_OK) return true;
    return false;
  }
  // run job(d) on every worker; returns the first failing device's code (its message → *who)
  int run(std::function<int(uint32_t)> f, uint32_t *who = nullptr) {
    {
      std::lock_guard<std::mutex> lk(m);
      job = std::move(f);
      n_done = 0;
      ++generation;
    }
    cv_job.notify_all();
    {
      std::unique_lock<std::mutex> lk(m);
      cv_done.wait(lk, [&] { return n_done == D; });
    }
    for (uint32_t d = 0; d < D; ++d)
      if (job_rc[d] != SHK_OK) {
        if (who) *who = d;
        return job_rc[d];
      }
    return SHK_OK;
  }
  void worker(uint32_t d) {
    (void)hipSetDevice(dev_ids[d]);
    uint64_t seen = 0;
    for (;;) {
      std::function<int(uint32_t)> f;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_job.wait(lk, [&] { return quit || generation != seen; });
        if (quit) return;
        seen = generation;
        f = job;
      }
      const int rc = f(d);
      {
        std::lock_guard<std::mutex> lk(m);
        job_rc[d] = rc;
        ++n_done;
      }
      cv_done.notify_all();
    }
  }
};

namespace {

int group_fail(shk_ctx *top, shk_group *g, int code, uint32_t who) {
  top->err = who < g->D ? g->ctx[who]->err : std::string("multi-device context failure");
  if (top->err.empty()) top->err = "device " + std::to_string(who) + " failed";
  return code;
}

int group_create(const shk_config *cfg, shk_ctx **out) {
  const uint32_t D = cfg->n_devices;
  if (D > 64 || (D & (D - 1))) return fail(nullptr, SHK_ERR_BAD_ARG, "n_devices must be a power of two ≤ 64, got %u", D);
  if (!cfg->device_ids) return fail(nullptr, SHK_ERR_BAD_ARG, "n_devices = %u but device_ids is null", D);
  if (cfg->n_owners > 1) return fail(nullptr, SHK_ERR_BAD_ARG, "a multi-device context assigns the owner shares itself");
  shk_ctx *top = new shk_ctx();
  top->cfg = *cfg;
  top->n_lanes = cfg->chunks == 0 ? 1 : cfg->chunks;
  shk_group *g = new shk_group();
  top->group = g;
  g->D = D;
  g->cfg = *cfg;
  g->dev_ids.assign(cfg->device_ids, cfg->device_ids + D);
  g->cfg.device_ids = g->dev_ids.data();
  top->cfg.device_ids = g->dev_ids.data();
  g->ctx.assign(D, nullptr);
  for (uint32_t d = 0; d < D; ++d) {
    shk_config c1 = *cfg;
    c1.n_devices = 0;
    c1.device_ids = nullptr;
    c1.device = g->dev_ids[d];
    c1.n_owners = D;
    c1.owner_id = d;
    c1.table_capacity_hint = cfg->table_capacity_hint ? (cfg->table_capacity_hint + D - 1) / D : 0;
    int rc = shk_create(&c1, &g->ctx[d]);
    if (rc == SHK_OK) {  // 4-byte records at this geometry, or the wide round for the whole context (every device alike)
      const PartGeom pg = part_geom(g->ctx[d]);
      if (xchg_check(g->ctx[d], pg) != SHK_OK) {
        g->wide = true;
        g->ctx[d]->err.clear();
      }
    }
    if (rc != SHK_OK) {
      for (auto *c : g->ctx)
        if (c) shk_destroy(c);
      delete g;
      delete top;
      return rc;
    }
  }
  // peers read each other's exchange buffers with hipMemcpyPeerAsync; direct access where the platform has it
  for (uint32_t a = 0; a < D; ++a)
    for (uint32_t b = 0; b < D; ++b)
      if (g->dev_ids[a] != g->dev_ids[b]) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, g->dev_ids[a], g->dev_ids[b]) == hipSuccess && can) {
          (void)hipSetDevice(g->dev_ids[a]);
          (void)hipDeviceEnablePeerAccess(g->dev_ids[b], 0);  // (already enabled: an error we ignore)
          (void)hipGetLastError();
        }
      }
  g->in_bases.resize(D);
  g->in_offsets.resize(D);
  g->recv_rec.resize(D);
  g->recv_cur.resize(D);
  g->recv_spill.resize(D);
  g->rebased.resize(D);
  g->wk_ptr.assign(D, nullptr);
  g->wl_ptr.assign(D, nullptr);
  g->wcounts.assign(D, std::vector<uint64_t>(D, 0));
  g->recv_wk.resize(D);
  g->recv_wl.resize(D);
  g->rec_ptr.assign(D, nullptr);
  g->cur_ptr.assign(D, nullptr);
  g->lay.assign(D, shk_xchg_layout{});
  g->n_foreign.assign(D, 0);
  g->sp_n.assign(D, 0);
  g->sp_k.assign(D, nullptr);
  g->sp_l.assign(D, nullptr);
  g->sp_c.assign(D, nullptr);
  g->status.assign(D, SHK_OK);
  g->job_rc.assign(D, SHK_OK);
  for (uint32_t d = 0; d < D; ++d) g->th.emplace_back([g, d] { g->worker(d); });
  *out = top;
  return SHK_OK;
}

void group_destroy(shk_ctx *top) {
  shk_group *g = top->group;
  {
    std::lock_guard<std::mutex> lk(g->m);
    g->quit = true;
  }
  g->cv_job.notify_all();
  for (auto &t : g->th)
    if (t.joinable()) t.join();
  for (uint32_t d = 0; d < g->D; ++d) {
    (void)hipSetDevice(g->dev_ids[d]);
    if (g->ctx[d]) (void)hipStreamSynchronize(g->ctx[d]->stream);
    g->in_bases[d].release();
    g->in_offsets[d].release();
    g->recv_rec[d].release();
    g->recv_cur[d].release();
    g->recv_spill[d].release();
    g->recv_wk[d].release();
    g->recv_wl[d].release();
    g->rebased[d].release();
    if (g->ctx[d]) shk_destroy(g->ctx[d]);
  }
  delete g;
  delete top;
}

// One exchange round: device d counts reads [cut[d], cut[d+1]) of the host batch.
int group_round(shk_ctx *top, const uint8_t *bases, const uint64_t *offsets, const std::vector<uint64_t> &cut,
                uint64_t first_read, uint64_t layout_bases, int64_t lane_fixed) {
  shk_group *g = top->group;
  const uint32_t D = g->D;
  std::fill(g->status.begin(), g->status.end(), SHK_OK);
  auto job = [&](uint32_t d) -> int {
    shk_ctx *c = g->ctx[d];
    auto hipc = [&](hipError_t e) {
      if (e != hipSuccess && g->status[d] == SHK_OK)
        g->status[d] = fail(c, e == hipErrorOutOfMemory ? SHK_ERR_NOMEM : SHK_ERR_HIP, "HIP error %s in a multi-device round", hipGetErrorString(e));
    };
    // ---- phase 1: my run of reads → my card → level-1 pass over every owner's records
    const uint64_t r0 = cut[d], r1 = cut[d + 1], ns = r1 - r0;
    const uint64_t o0 = offsets[r0], nb = offsets[r1] - o0;
    hipc(g->in_bases[d].ensure(nb + 64));
    hipc(g->in_offsets[d].ensure((ns + 1) * 8));
    hipc(g->rebased[d].ensure((ns + 1) * 8));
    if (g->status[d] == SHK_OK) {
      uint64_t *reb = (uint64_t *)g->rebased[d].p;
      if (nb) hipc(hipMemcpyAsync(g->in_bases[d].p, bases + o0, nb, hipMemcpyHostToDevice, c->stream));
      for (uint64_t j = 0; j <= ns; ++j) reb[j] = offsets[r0 + j] - o0;
      hipc(hipMemcpyAsync(g->in_offsets[d].p, reb, (ns + 1) * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (g->wide) {  // ---- the wide round: whole k-mers grouped by owner, pulled and inserted by their owners
      if (g->status[d] == SHK_OK) {
        if (lane_fixed >= 0) c->xchg_lane_fixed = lane_fixed;
        else (void)shk_set_read_index(c, first_read + r0);
        g->status[d] = shk_xchg_wide_scatter_device(c, g->in_bases[d].p, g->in_offsets[d].p, ns, nb, &g->wk_ptr[d], &g->wl_ptr[d],
                                                    g->wcounts[d].data());
        c->xchg_lane_fixed = -1;
      }
      if (g->barrier()) return g->status[d];
      for (uint32_t i = 0; i < D && g->status[d] == SHK_OK; ++i) {
        const uint32_t s = (d + i) % D;  // (my own piece first: no copy)
        const uint64_t n = g->wcounts[s][d];
        if (!n) continue;
        uint64_t off = 0;
        for (uint32_t o = 0; o < d; ++o) off += g->wcounts[s][o];
        const char *src_k = (const char *)g->wk_ptr[s] + off * 8, *src_l = (const char *)g->wl_ptr[s] + off * 4;
        if (s == d) {
          g->status[d] = shk_insert_device(c, src_k, src_l, nullptr, n);
        } else {
          hipc(g->recv_wk[d].ensure(n * 8));
          hipc(g->recv_wl[d].ensure(n * 4));
          hipc(hipMemcpyPeerAsync(g->recv_wk[d].p, g->dev_ids[d], src_k, g->dev_ids[s], n * 8, c->stream));
          hipc(hipMemcpyPeerAsync(g->recv_wl[d].p, g->dev_ids[d], src_l, g->dev_ids[s], n * 4, c->stream));
          if (g->status[d] == SHK_OK) g->status[d] = shk_insert_device(c, g->recv_wk[d].p, g->recv_wl[d].p, nullptr, n);  // (synchronous)
        }
      }
      hipc(hipStreamSynchronize(c->stream));
      (void)g->barrier();  // everybody has pulled what it owns: the peers may scatter again
      return g->status[d];
    }
    if (g->status[d] == SHK_OK) {
      if (lane_fixed >= 0) c->xchg_lane_fixed = lane_fixed;  // drain_batch with an explicit chunk (io.rs:356-358)
      else (void)shk_set_read_index(c, first_read + r0);
      g->status[d] = shk_xchg_scatter_device(c, g->in_bases[d].p, g->in_offsets[d].p, ns, nb, layout_bases, &g->rec_ptr[d],
                                             &g->cur_ptr[d], &g->lay[d], &g->n_foreign[d]);
      c->xchg_lane_fixed = -1;
    }
    if (g->barrier()) return g->status[d];
    bool foreign = false;
    for (uint32_t s = 0; s < D; ++s) foreign |= g->n_foreign[s] > 0;
    // ---- phase 2: pull my segment out of every peer's buffer and absorb it
    const shk_xchg_layout L = g->lay[d];
    const size_t seg_bytes = (size_t)L.segment_records * (L.record_bytes == 8 ? 8 : 4), cur_bytes = (size_t)L.regions * 4;
    hipc(g->recv_rec[d].ensure(seg_bytes));
    hipc(g->recv_cur[d].ensure(cur_bytes));
    for (uint32_t i = 0; i < D && g->status[d] == SHK_OK; ++i) {
      const uint32_t s = (d + i) % D;  // (start with my own segment: no copy; then one peer after the other)
      const char *src_rec = (const char *)g->rec_ptr[s] + (size_t)d * seg_bytes;
      const char *src_cur = (const char *)g->cur_ptr[s] + (size_t)d * cur_bytes;
      if (g->lay[s].segment_records != L.segment_records || g->lay[s].regions != L.regions) {
        g->status[d] = fail(c, SHK_ERR_INVARIANT, "devices disagree on the exchange layout");
        break;
      }
      if (s == d) {
        g->status[d] = shk_xchg_absorb(c, src_rec, src_cur, &L);
      } else {
        hipc(hipMemcpyPeerAsync(g->recv_rec[d].p, g->dev_ids[d], src_rec, g->dev_ids[s], seg_bytes, c->stream));
        hipc(hipMemcpyPeerAsync(g->recv_cur[d].p, g->dev_ids[d], src_cur, g->dev_ids[s], cur_bytes, c->stream));
        if (g->status[d] == SHK_OK) g->status[d] = shk_xchg_absorb(c, g->recv_rec[d].p, g->recv_cur[d].p, &L);
      }
    }
    hipc(hipStreamSynchronize(c->stream));  // my pulls are done: the peers may scatter again
    if (foreign && g->status[d] == SHK_OK) {
      uint64_t n = 0;
      g->status[d] = shk_xchg_spill(c, &g->sp_k[d], &g->sp_l[d], &g->sp_c[d], &n);
      g->sp_n[d] = n;
    }
    if (g->barrier()) return g->status[d];
    // ---- phase 3 (skewed input only): every device inserts what it owns of every spill list
    if (foreign) {
      for (uint32_t i = 0; i < D && g->status[d] == SHK_OK; ++i) {
        const uint32_t s = (d + i) % D;
        const uint64_t n = g->sp_n[s];
        if (!n) continue;
        if (s == d) {
          g->status[d] = shk_insert_device(c, g->sp_k[s], g->sp_l[s], g->sp_c[s], n);
          continue;
        }
        hipc(g->recv_spill[d].ensure(n * 16));
        char *p = (char *)g->recv_spill[d].p;
        hipc(hipMemcpyPeerAsync(p, g->dev_ids[d], g->sp_k[s], g->dev_ids[s], n * 8, c->stream));
        hipc(hipMemcpyPeerAsync(p + n * 8, g->dev_ids[d], g->sp_l[s], g->dev_ids[s], n * 4, c->stream));
        hipc(hipMemcpyPeerAsync(p + n * 12, g->dev_ids[d], g->sp_c[s], g->dev_ids[s], n * 4, c->stream));
        if (g->status[d] == SHK_OK) g->status[d] = shk_insert_device(c, p, p + n * 8, p + n * 12, n);  // (synchronous)
      }
      (void)g->barrier();  // everybody has read my list
      if (g->status[d] == SHK_OK) g->status[d] = shk_xchg_spill_clear(c);
    }
    return g->status[d];
  };
  uint32_t who = 0;
  const int rc = g->run(job, &who);
  return rc == SHK_OK ? SHK_OK : group_fail(top, g, rc, who);
}

// Host buffers → rounds.  Every round deals D contiguous runs of whole reads, each ≤ R bases.
int group_ingest(shk_ctx *top, const uint8_t *bases, const uint64_t *offsets, uint64_t n_seqs, int64_t lane_fixed) {
  shk_group *g = top->group;
  const uint32_t D = g->D;
  if (n_seqs && !offsets) return fail(top, SHK_ERR_BAD_ARG, "null offsets");
  g->finalized = g->hist_ready = false;
  if (n_seqs == 0) return SHK_OK;
  const uint64_t R = std::min<uint64_t>((uint64_t)env_int("SHK_GROUP_ROUND_KB", (int)(SHK_XCHG_MAX_BASES >> 10)) << 10, SHK_XCHG_MAX_BASES);
  const uint64_t total = offsets[n_seqs] - offsets[0];
  if (total && !bases) return fail(top, SHK_ERR_BAD_ARG, "null bases");
  const uint64_t target = std::max<uint64_t>(std::min<uint64_t>(R, (total + D - 1) / D), 1);
  const uint64_t call_first = g->next_read;  // global index of this call's first read
  uint64_t r = 0;
  while (r < n_seqs) {
    std::vector<uint64_t> cut(D + 1, r);
    uint64_t round_max = 0;
    for (uint32_t d = 0; d < D; ++d) {
      uint64_t a = cut[d], b = a;
      if (a < n_seqs) {
        const uint64_t limit = offsets[a] + target;  // largest b with offsets[b] ≤ limit, at least one read
        b = (uint64_t)(std::upper_bound(offsets + a, offsets + n_seqs + 1, limit) - offsets) - 1;
        if (b <= a) b = a + 1;
        if (offsets[b] - offsets[a] > SHK_XCHG_MAX_BASES)
          return fail(top, SHK_ERR_BAD_ARG, "a single read of %llu bases exceeds what one exchange round takes", (unsigned long long)(offsets[b] - offsets[a]));
      }
      cut[d + 1] = b;
      round_max = std::max(round_max, offsets[b] - offsets[a]);
    }
    // (cut[] counts from the start of this call: so does the global index handed on)
    const int rc = group_round(top, bases, offsets, cut, call_first, std::max<uint64_t>(round_max, 1), lane_fixed);
    if (rc != SHK_OK) return rc;
    r = cut[D];
  }
  if (lane_fixed < 0) g->next_read = call_first + n_seqs;
  return SHK_OK;
}

int group_finalize(shk_ctx *top) {
  shk_group *g = top->group;
  if (g->finalized) return SHK_OK;
  const uint32_t D = g->D;
  const size_t hn = (size_t)g->cfg.chunks * (g->cfg.histo_max + 2);
  std::vector<std::vector<uint64_t>> h(D, std::vector<uint64_t>(hn));
  std::vector<shk_counters> cn(D);
  uint32_t who = 0;
  int rc = g->run([&](uint32_t d) -> int {
    int r = shk_finalize(g->ctx[d]);
    if (r != SHK_OK) return r;
    if (hn) r = shk_histograms(g->ctx[d], h[d].data());
    if (r == SHK_OK) r = shk_get_counters(g->ctx[d], &cn[d]);
    return r;
  }, &who);
  if (rc != SHK_OK) return group_fail(top, g, rc, who);
  g->hist.assign(hn, 0);
  shk_counters t{};
  for (uint32_t d = 0; d < D; ++d) {
    for (size_t i = 0; i < hn; ++i) g->hist[i] += h[d][i];
    t.n_reads_ingested += cn[d].n_reads_ingested;
    t.n_bases_read += cn[d].n_bases_read;
    t.n_bases_ingested += cn[d].n_bases_ingested;
    t.n_kmers_ingested += cn[d].n_kmers_ingested;
    t.n_unique_kmers += cn[d].n_unique_kmers;
    t.n_hashed_kmers += cn[d].n_hashed_kmers;
    t.any_saturated |= cn[d].any_saturated;
    t.table_capacity += cn[d].table_capacity;
    t.n_grows += cn[d].n_grows;
    t.n_spilled += cn[d].n_spilled;
  }
  t.n_chunks = top->n_lanes;
  if (g->cfg.chunks > 0) t.n_singleton_kmers = g->hist[(size_t)(g->cfg.chunks - 1) * (g->cfg.histo_max + 2) + 1];
  g->tot = t;
  g->hist_ready = true;
  if (t.n_reads_ingested == 0)  // io.rs:578-580, on the whole job
    return fail(top, SHK_ERR_NO_READS, "No reads were ingested. Check that input files contain valid FASTQ records.");
  if (t.n_hashed_kmers != t.n_kmers_ingested)  // io.rs:1042-1047
    return fail(top, SHK_ERR_INVARIANT, "The total count of hashed kmers (%llu) does not equal the number of ingested kmers (%llu)",
                (unsigned long long)t.n_hashed_kmers, (unsigned long long)t.n_kmers_ingested);
  g->finalized = true;
  return SHK_OK;
}

int group_counters(shk_ctx *top, shk_counters *o) {
  shk_group *g = top->group;
  if (g->finalized || g->hist_ready) {
    *o = g->tot;
    return SHK_OK;
  }
  memset(o, 0, sizeof *o);
  for (uint32_t d = 0; d < g->D; ++d) {
    shk_counters c1{};
    int rc = shk_get_counters(g->ctx[d], &c1);
    if (rc != SHK_OK) return group_fail(top, g, rc, d);
    o->n_reads_ingested += c1.n_reads_ingested;
    o->n_bases_read += c1.n_bases_read;
    o->n_bases_ingested += c1.n_bases_ingested;
    o->table_capacity += c1.table_capacity;
    o->n_grows += c1.n_grows;
    o->n_spilled += c1.n_spilled;
  }
  o->n_chunks = top->n_lanes;
  return SHK_OK;
}

}  // namespace
