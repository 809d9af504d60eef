"""world_size-2 gloo tests (CPU) of the multi-GPU merge logic in sharkmer_amd/dist.py: batch
sharding, geometry agreement, owner-range all_to_all, merge, owned-shard histogram, all_reduce.
The per-rank engine is a numpy stand-in with the same duck-typed interface KmerEngine offers
(the HIP engine itself needs a GPU); k-mers come from the oracle's numpy extractor, and the
result is checked against the oracle run over ALL reads."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

EMPTY = np.int64(-1)


def _hash32(key: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        h = (key.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)
    return h.astype(np.uint64)


class NumpyPagedEngine:
    """Stand-in for KmerEngine: a dict of per-lane counts, materialised on demand as the same
    paged arrays (page = top hash bits) the HIP table uses."""
    PAGE_SLOTS = 64

    def __init__(self, k, chunks, histo_max):
        self.k, self.chunks, self.histo_max = k, chunks, histo_max
        self.n_lanes = max(chunks, 1)
        self.counts = {}
        self.log_pages = 2
        self.n_reads = self.n_bases = self.n_valid = 0
        self.owned = None

    # ---- ingest (oracle-side extraction; chunk of read i = (i // 1000) % n_lanes) ------------
    def ingest(self, orc, bases, offsets, first_read_index):
        offsets = np.asarray(offsets, dtype=np.int64)
        seg = bases[offsets[0]:offsets[-1]]
        rel = offsets - offsets[0]
        kmers, rid = orc.canonical_kmers_numpy(seg, rel, self.k, return_read_id=True)
        lanes = ((first_read_index + rid) // 1000) % self.n_lanes
        for key, lane in zip(kmers.tolist(), lanes.tolist()):
            v = self.counts.setdefault(key, np.zeros(self.n_lanes, dtype=np.int64))
            v[lane] = min(v[lane] + 1, 0xFFFFFFFF)
        self.n_reads += len(offsets) - 1
        self.n_bases += len(seg)
        self.n_valid += int((seg != ord("N")).sum())

    # ---- the interface DistCounter uses -----------------------------------------------------------
    def _page(self, key):
        return int(_hash32(np.array([key], dtype=np.uint64))[0]) >> (32 - self.log_pages)

    def _fits(self):
        load = {}
        for key in self.counts:
            p = self._page(key)
            load[p] = load.get(p, 0) + 1
        return all(v <= self.PAGE_SLOTS for v in load.values())

    def table_geometry(self):
        while not self._fits():
            self.log_pages += 1
        return 1 << self.log_pages, self.PAGE_SLOTS, self.n_lanes

    def reserve_pages(self, n_pages):
        while (1 << self.log_pages) < n_pages:
            self.log_pages += 1
        while not self._fits():
            self.log_pages += 1

    def table_tensors(self):
        P, S = 1 << self.log_pages, self.PAGE_SLOTS
        keys = np.full(P * S, EMPTY, dtype=np.int64)
        vals = np.zeros((self.n_lanes, P * S), dtype=np.int32)
        fill = [0] * P
        for key, v in self.counts.items():
            p = self._page(key)
            i = p * S + fill[p]
            fill[p] += 1
            keys[i] = key
            vals[:, i] = np.array(v, dtype=np.uint32).view(np.int32)
        return torch.from_numpy(keys), torch.from_numpy(vals)

    def merge_page_tensors(self, p0, p1, keys_t, vals_t):
        keys = keys_t.numpy()
        vals = vals_t.numpy().view(np.uint32)
        for i in np.nonzero(keys != EMPTY)[0]:
            key = int(keys[i])
            assert p0 <= self._page(key) < p1, "peer sent a key outside my range"
            v = self.counts.setdefault(key, np.zeros(self.n_lanes, dtype=np.int64))
            v[:] = np.minimum(v + vals[:, i].astype(np.int64), 0xFFFFFFFF)

    def owner_counts(self, n_owners):
        P = 1 << self.log_pages
        per = P // n_owners
        out = np.zeros(n_owners, dtype=np.uint64)
        for key in self.counts:
            out[self._page(key) // per] += 1
        return out

    def compact_owner_tensors(self, counts, skip_owner=-1):
        P = 1 << self.log_pages
        per = P // len(counts)
        order = sorted((key for key in self.counts if self._page(key) // per != skip_owner),
                       key=lambda key: (self._page(key) // per, key))
        assert [sum(1 for key in order if self._page(key) // per == o) for o in range(len(counts))] == \
            [int(c) for c in counts]
        keys = np.array(order, dtype=np.uint64).view(np.int64) if order else np.zeros(0, dtype=np.int64)
        vals = np.zeros((self.n_lanes, len(order)), dtype=np.int32)
        for i, key in enumerate(order):
            vals[:, i] = np.array(self.counts[key], dtype=np.uint32).view(np.int32)
        return torch.from_numpy(keys.copy()), torch.from_numpy(vals)

    def merge_entry_tensors(self, keys_t, vals_t):
        keys = keys_t.numpy().view(np.uint64)
        vals = vals_t.numpy().view(np.uint32)
        for i in range(len(keys)):
            key = int(keys[i])
            v = self.counts.setdefault(key, np.zeros(self.n_lanes, dtype=np.int64))
            v[:] = np.minimum(v + vals[:, i].astype(np.int64), 0xFFFFFFFF)

    def set_owned_pages(self, p0, p1):
        self.owned = (p0, p1)

    def finalize(self):
        hm = self.histo_max
        self.hist = np.zeros((self.chunks, hm + 2), dtype=np.uint64)
        self.n_unique = self.n_hashed = self.n_lane_sum = 0
        for key, v in self.counts.items():
            if self.owned and not (self.owned[0] <= self._page(key) < self.owned[1]):
                continue
            self.n_unique += 1
            self.n_lane_sum += int(v.sum())
            cum = 0
            for l in range(self.n_lanes):
                cum = min(cum + int(v[l]), 0xFFFFFFFF)
                if l < self.chunks and cum > 0:
                    self.hist[l, min(cum, hm + 1)] += 1
            self.n_hashed += cum

    def histograms(self):
        return self.hist

    def counters(self):
        return {"n_reads_ingested": self.n_reads, "n_bases_read": self.n_bases,
                "n_bases_ingested": self.n_valid, "n_kmers_ingested": self.n_lane_sum,
                "n_unique_kmers": self.n_unique, "n_hashed_kmers": self.n_hashed, "any_saturated": 0}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, k, chunks, histo_max, n_reads, out_dir, dense=False):
    import sys
    os.environ["SHK_DIST_DENSE"] = "1" if dense else "0"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import sharkmer_amd as sa
    from sharkmer_amd.dist import DistCounter, shard_batches
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = sa.SynthSpec(genome_len=3000, sub_per_64k=400, n_per_64k=100)
    bases, offsets = sa.synth_reads(spec, 0, n_reads)
    eng = NumpyPagedEngine(k, chunks, histo_max)
    for first, n in shard_batches(n_reads, rank, world):
        eng.ingest(orc, bases, offsets[first:first + n + 1], first)
    dc = DistCounter(eng, dist)
    hist = dc.finalize_histograms()
    np.save(os.path.join(out_dir, f"hist_{rank}.npy"), hist)
    np.save(os.path.join(out_dir, f"tot_{rank}.npy"),
            np.array([dc.totals[x] for x in ("n_reads_ingested", "n_bases_read", "n_bases_ingested",
                                             "n_kmers_ingested", "n_unique_kmers")], dtype=np.int64))
    dist.destroy_process_group()


@pytest.mark.parametrize("k,chunks,n_reads,dense,max_msg", [(21, 1, 2500, False, 0), (15, 3, 4321, False, 0), (9, 0, 1800, False, 0),
                                                              (15, 3, 4321, True, 0), (15, 3, 4321, False, 2048), (15, 3, 4321, True, 4096)])
def test_two_rank_merge_matches_single_oracle(orc, tmp_path, monkeypatch, k, chunks, n_reads, dense, max_msg):
    """Compact exchange (occupied entries, uneven all_to_all splits) and the dense one (page ranges).  max_msg: the
    message limit of sharkmer_amd.dist pinned so low that every exchange goes in pieces (grouped send/recv pairs over
    slices, the own part by a copy) — the path that keeps messages below what RCCL was measured to deliver whole."""
    import sharkmer_amd as sa
    if max_msg:
        monkeypatch.setenv("SHK_DIST_MAX_MESSAGE", str(max_msg))
    histo_max = 40
    port = _free_port()
    mp.spawn(_worker, args=(2, port, k, chunks, histo_max, n_reads, str(tmp_path), dense), nprocs=2, join=True)
    spec = sa.SynthSpec(genome_len=3000, sub_per_64k=400, n_per_64k=100)
    bases, offsets = sa.synth_reads(spec, 0, n_reads)
    ref = orc.run_batch(bases, offsets, k, chunks, histo_max)
    h0 = np.load(tmp_path / "hist_0.npy")
    h1 = np.load(tmp_path / "hist_1.npy")
    assert np.array_equal(h0, h1)
    assert np.array_equal(h0, ref.histograms())
    t0 = np.load(tmp_path / "tot_0.npy")
    st = ref.stats
    assert list(t0) == [st["n_reads_ingested"], st["n_bases_read"], st["n_bases_ingested"],
                        st["n_kmers_ingested"], st["n_unique_kmers"]]


def _skewed_merge_worker(rank, world, port, out_dir):
    """Rank 0 holds many keys of owner 1's range; everything else anybody holds is a handful of keys."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["SHK_DIST_MAX_MESSAGE"] = "4096"   # bytes: 512 keys
    os.environ["SHK_DIST_DENSE"] = "0"
    from sharkmer_amd.dist import DistCounter
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    eng = NumpyPagedEngine(21, 1, 40)
    eng.log_pages = 6   # 64 pages of 64 slots, 16 pages per owner
    rng = np.random.default_rng(1234)   # the same stream on every rank: every rank can name every key
    pool = rng.integers(1, 1 << 42, size=6000, dtype=np.int64)
    owner = (_hash32(pool.astype(np.uint64)) >> np.uint64(32 - 6)).astype(np.int64) // 16
    big = pool[owner == 1][:700]                      # 700 keys x 8 B = 5600 B > the limit: only the part 0 -> 1
    small = [pool[owner == o][700:700 + 5] for o in range(world)]
    mine = {int(x): 1 for o in range(world) for x in small[o]}
    if rank == 0:
        mine.update({int(x): 2 for x in big})
    for key, c in mine.items():
        eng.counts[key] = np.array([c], dtype=np.int64)
    dc = DistCounter(eng, dist)
    hist = dc.finalize_histograms()
    np.save(os.path.join(out_dir, f"hist_{rank}.npy"), hist)
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_four_rank_merge_with_one_oversized_part_takes_one_route_everywhere(tmp_path):
    """Only the part rank 0 -> rank 1 is above the message limit; ranks 2 and 3 see nothing but small parts.  All four
    must still take the same route through exchange_parts (send/recv pairs): a rank that chose all_to_all_single on
    its own view left the others waiting (a hang under gloo, mispaired messages under RCCL).  The largest part of the
    whole exchange rides in the (page count, counts) all_to_all that every rank already makes."""
    world = 4
    port = _free_port()
    mp.spawn(_skewed_merge_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    hs = [np.load(tmp_path / f"hist_{r}.npy") for r in range(world)]
    for h in hs[1:]:
        assert np.array_equal(h, hs[0])
    # 20 small keys held by all four ranks with count 1 each -> merged count 4; 700 keys with count 2 on rank 0 only
    want = np.zeros_like(hs[0])
    want[0, 4] = 20
    want[0, 2] = 700
    assert np.array_equal(hs[0], want)


def test_shard_batches_cover_every_read_once():
    from sharkmer_amd.dist import shard_batches
    for n, w in [(0, 2), (999, 2), (1000, 2), (2500, 2), (10_001, 8), (123_456, 4)]:
        seen = np.zeros(n, dtype=np.int32)
        for r in range(w):
            for first, cnt in shard_batches(n, r, w):
                assert first % 1000 == 0 and 0 < cnt <= 1000
                seen[first:first + cnt] += 1
        assert (seen == 1).all()


# ---- key-space-partitioned ingest (sharkmer_amd.dist.OwnerCounter) under gloo ----------------------------------

MIX_M32, MIX_M64 = 0xC2B2AE35, 0x9E3779B97F4A7C15  # shk_device.hip.h: mix_key


def _mix(x, bits):
    return (x * (MIX_M32 if bits <= 42 else MIX_M64)) & ((1 << bits) - 1)


def _unmix(y, bits):
    return (y * pow(MIX_M32 if bits <= 42 else MIX_M64, -1, 1 << 64)) & ((1 << bits) - 1)


class _Layout:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class NumpyOwnerEngine:
    """Stand-in for an owner-share KmerEngine (shk_config.n_owners): speaks the exchange wire format of
    include/shk.h — per owner one segment of `regions` = [lane][super-page] regions of `region_cap` 4-byte
    records (the low bits of the mixed key below the level-1 bits), block-interleaved in 1024-record blocks,
    fill levels as one u32 per region; overflow goes to the foreign spill list."""

    def __init__(self, k, chunks, histo_max, n_owners, owner_id, log_p1, region_cap=1024):
        self.k, self.chunks, self.histo_max = k, chunks, histo_max
        self.n_lanes = max(chunks, 1)
        self.W, self.me = n_owners, owner_id
        self.lw = n_owners.bit_length() - 1
        self.log_p1 = log_p1
        self.log_p1w = log_p1 - self.lw
        self.cap = region_cap
        self.counts = {}
        self.spill = []
        self.next_read = 0
        self.n_reads = self.n_bases = self.n_valid = 0

    def xchg_feasible(self):
        """(shk_xchg_feasible) the low bits of the mixed key below the level-1 bits fit a 4-byte record"""
        return 2 * self.k - self.log_p1 <= 32

    def set_read_index(self, i):
        self.next_read = i

    def stream(self):
        return 0

    def xchg_wide_scatter_tensors(self, bases, offsets, n_seqs, n_bases):
        """(shk_xchg_wide_scatter_device) whole k-mers + lanes grouped by owner, and how many each owner gets."""
        from oracle import oracle as orc
        offsets = np.asarray(offsets, dtype=np.int64)
        segb = bases[offsets[0]:offsets[-1]]
        bad = segb[~np.isin(segb, np.frombuffer(b"ACGTN", dtype=np.uint8))]
        if len(bad):
            raise RuntimeError(f"Invalid character '{chr(bad[0])}' in sequence. Only ACGTN allowed.")
        kmers, rid = orc.canonical_kmers_numpy(segb, offsets - offsets[0], self.k, return_read_id=True)
        lanes = (((self.next_read + rid) // 1000) % self.n_lanes).astype(np.int32)
        bits = 2 * self.k
        own = np.array([(_mix(key, bits) >> (bits - self.lw)) if self.lw else 0 for key in kmers.tolist()], dtype=np.int64)
        order = np.argsort(own, kind="stable")[::-1] if len(own) else np.zeros(0, dtype=np.int64)
        order = order[np.argsort(own[order], kind="stable")]   # (grouped by owner, any order within: the engine's is arbitrary too)
        self.n_reads += n_seqs
        self.n_bases += len(segb)
        self.n_valid += int((segb != ord("N")).sum())
        counts = np.bincount(own, minlength=self.W).tolist()
        return torch.from_numpy(kmers.astype(np.int64)[order].copy()), torch.from_numpy(lanes[order].copy()), counts

    def _slot(self, g, n_grp, j):
        return (((j >> 10) * n_grp + g) << 10) | (j & 1023)

    def xchg_scatter_tensors(self, bases, offsets, n_seqs, n_bases, layout_bases=0):
        from oracle import oracle as orc
        n_grp = self.n_lanes << self.log_p1w
        seg = n_grp * self.cap
        rec = np.zeros(self.W * seg, dtype=np.int32)
        cur = np.zeros(self.W * n_grp, dtype=np.int32)
        bits, r1 = 2 * self.k, 2 * self.k - self.log_p1
        if n_seqs:
            offsets = np.asarray(offsets, dtype=np.int64)
            segb = bases[offsets[0]:offsets[-1]]
            if np.any(~np.isin(segb, np.frombuffer(b"ACGTN", dtype=np.uint8))):
                bad = segb[~np.isin(segb, np.frombuffer(b"ACGTN", dtype=np.uint8))][0]
                raise RuntimeError(f"Invalid character '{chr(bad)}' in sequence. Only ACGTN allowed.")
            kmers, rid = orc.canonical_kmers_numpy(segb, offsets - offsets[0], self.k, return_read_id=True)
            lanes = ((self.next_read + rid) // 1000) % self.n_lanes
            for key, lane in zip(kmers.tolist(), lanes.tolist()):
                y = _mix(key, bits)
                gp = y >> r1
                o, sp = gp >> self.log_p1w, gp & ((1 << self.log_p1w) - 1)
                g = (lane << self.log_p1w) | sp
                j = int(cur[o * n_grp + g])
                if j < self.cap:
                    rec[o * seg + self._slot(g, n_grp, j)] = np.uint32(y & ((1 << r1) - 1)).view(np.int32)
                    cur[o * n_grp + g] = j + 1
                else:
                    self.spill.append((key, lane, 1))
            self.n_reads += n_seqs
            self.n_bases += len(segb)
            self.n_valid += int((segb != ord("N")).sum())
        lay = _Layout(n_owners=self.W, n_lanes=self.n_lanes, log_p1=self.log_p1, regions=n_grp,
                      region_cap=self.cap, segment_records=seg)
        return torch.from_numpy(rec), torch.from_numpy(cur), lay, len(self.spill)

    def _add(self, key, lane, cnt):
        v = self.counts.setdefault(key, np.zeros(self.n_lanes, dtype=np.int64))
        v[lane] = min(v[lane] + cnt, 0xFFFFFFFF)

    def xchg_absorb_tensors(self, rec_t, cur_t, lay):
        assert lay.n_owners == self.W and lay.regions == self.n_lanes << self.log_p1w
        rec, cur = rec_t.numpy().view(np.uint32), cur_t.numpy()
        bits, r1 = 2 * self.k, 2 * self.k - self.log_p1
        for g in range(lay.regions):
            lane, sp = g >> self.log_p1w, g & ((1 << self.log_p1w) - 1)
            for j in range(min(int(cur[g]), lay.region_cap)):
                y = ((((self.me << self.log_p1w) | sp) << r1) | int(rec[self._slot(g, lay.regions, j)]))
                self._add(_unmix(y, bits), lane, 1)

    def xchg_spill_tensors(self):
        a = np.array(self.spill, dtype=np.int64).reshape(-1, 3)
        return (torch.from_numpy(a[:, 0].copy()), torch.from_numpy(a[:, 1].astype(np.int32)),
                torch.from_numpy(a[:, 2].astype(np.int32)))

    def insert_tensors(self, k_t, l_t, c_t):
        bits = 2 * self.k
        for key, lane, cnt in zip(k_t.tolist(), l_t.tolist(), c_t.tolist() if c_t is not None else [1] * k_t.numel()):
            if (_mix(key, bits) >> (bits - self.lw) if self.lw else 0) == self.me:
                self._add(key, lane, cnt)

    def xchg_spill_clear(self):
        self.spill = []

    finalize = NumpyPagedEngine.finalize
    histograms = NumpyPagedEngine.histograms
    counters = NumpyPagedEngine.counters
    owned = None


class NumpyOwnerEngineTwoCalls(NumpyOwnerEngine):
    """The same stand-in with the scatter in two calls (include/shk.h: shk_xchg_scatter_begin / _end), behaving like the
    device engine: begin hands out the segments and reports nothing; what the scatter found — an invalid byte, foreign
    spills — comes out of end.  Keeps the order of the calls OwnerCounter makes."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.calls, self._held = [], None

    def xchg_scatter_begin_tensors(self, bases, offsets, n_seqs, n_bases, layout_bases=0):
        assert self._held is None, "begin before the last scatter was ended"
        self.calls.append("begin")
        try:
            rec, cur, lay, nf = self.xchg_scatter_tensors(bases, offsets, n_seqs, n_bases, layout_bases)
            self._held = (None, nf)
        except RuntimeError as e:  # (the device finds the byte while the host has already gone on: reported by end)
            rec, cur, lay, _ = self.xchg_scatter_tensors(0, 0, 0, 0, layout_bases)
            self._held = (e, 0)
        return rec, cur, lay

    def xchg_scatter_end(self):
        assert self._held is not None, "end without a begin"
        self.calls.append("end")
        (e, nf), self._held = self._held, None
        if e is not None:
            raise e
        return nf

    def xchg_absorb_tensors(self, rec_t, cur_t, lay):
        self.calls.append("absorb")
        super().xchg_absorb_tensors(rec_t, cur_t, lay)


def _owner_engine(*a):
    return (NumpyOwnerEngineTwoCalls if os.environ.get("SHK_TEST_TWO_CALL_ENGINE") else NumpyOwnerEngine)(*a)


def _owner_worker(rank, world, port, k, chunks, histo_max, n_reads, log_p1, cap, poly, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import sharkmer_amd as sa
    from sharkmer_amd.dist import OwnerCounter, shard_batches
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bases, offsets = _owner_input(sa, n_reads, poly)
    eng = _owner_engine(k, chunks, histo_max, world, rank, log_p1, cap)
    oc = OwnerCounter(eng, dist, round_bases=1000 * 160)
    mine = shard_batches(n_reads, rank, world)
    for r in range(len(shard_batches(n_reads, 0, world))):
        if r < len(mine):
            first, n = mine[r]
            oc.round((bases, offsets[first:first + n + 1], n, int(offsets[first + n] - offsets[first]), first))
        else:
            oc.round(None)
    hist = oc.finalize_histograms()
    if hasattr(eng, "calls"):
        open(os.path.join(out_dir, f"calls_{rank}.txt"), "w").write(" ".join(eng.calls))
    np.save(os.path.join(out_dir, f"hist_{rank}.npy"), hist)
    np.save(os.path.join(out_dir, f"tot_{rank}.npy"),
            np.array([oc.totals[x] for x in ("n_reads_ingested", "n_bases_read", "n_bases_ingested",
                                             "n_kmers_ingested", "n_unique_kmers")] + [oc.n_foreign_rounds], dtype=np.int64))
    dist.destroy_process_group()


def _owner_input(sa, n_reads, poly):
    spec = sa.SynthSpec(genome_len=3000, sub_per_64k=400, n_per_64k=100)
    bases, offsets = sa.synth_reads(spec, 0, n_reads)
    if poly:  # low-complexity reads: their one k-mer overflows its region → the foreign spill list
        bases = bases.copy()
        for i in range(0, n_reads, 3):
            bases[int(offsets[i]):int(offsets[i + 1])] = ord("A")
    return bases, offsets


@pytest.mark.parametrize("k,chunks,n_reads,log_p1,cap,poly,max_msg", [(21, 10, 3300, 10, 1024, False, 0), (15, 3, 2500, 4, 1024, True, 0),
                                                                     (9, 0, 1800, 1, 2048, False, 0), (21, 10, 3300, 10, 1024, False, 8192),
                                                                     (15, 3, 2500, 4, 1024, True, 1000),
                                                                     (31, 10, 3300, 10, 1024, False, 0), (25, 3, 2500, 10, 1024, True, 4096),
                                                                     (27, 0, 1800, 10, 1024, False, 0)])
def test_two_rank_owner_partitioned_ingest_matches_single_oracle(orc, tmp_path, monkeypatch, k, chunks, n_reads, log_p1, cap, poly, max_msg):
    """World 2, 10 chunk lanes (BASELINE configs[4]'s shape): every rank ingests its own 1000-read batches,
    records travel by owner every round, nothing is merged at finalize; a skewed case goes through the
    foreign spill list.  max_msg: segments above the (pinned) message limit travel in pieces and a rank's own segment
    is absorbed where the scatter left it.  k > 21 (2k − log_p1 > 32): the wide round — whole k-mers, as many per owner as
    there are, unequal parts through the same message limit."""
    import sharkmer_amd as sa
    if max_msg:
        monkeypatch.setenv("SHK_DIST_MAX_MESSAGE", str(max_msg))
    histo_max = 40
    port = _free_port()
    mp.spawn(_owner_worker, args=(2, port, k, chunks, histo_max, n_reads, log_p1, cap, poly, str(tmp_path)), nprocs=2, join=True)
    bases, offsets = _owner_input(sa, n_reads, poly)
    ref = orc.run_batch(bases, offsets, k, chunks, histo_max)
    h0 = np.load(tmp_path / "hist_0.npy")
    h1 = np.load(tmp_path / "hist_1.npy")
    assert np.array_equal(h0, h1)
    assert np.array_equal(h0, ref.histograms())
    t0 = np.load(tmp_path / "tot_0.npy")
    st = ref.stats
    assert list(t0[:5]) == [st["n_reads_ingested"], st["n_bases_read"], st["n_bases_ingested"],
                            st["n_kmers_ingested"], st["n_unique_kmers"]]
    if poly and 2 * k - log_p1 <= 32:
        assert t0[5] > 0, "the low-complexity reads were meant to overflow a region"


@pytest.mark.parametrize("k,chunks,log_p1", [(21, 3, 10), (27, 2, 10)])
def test_four_rank_owner_partitioned_ingest(orc, tmp_path, k, chunks, log_p1):
    """World 4 (two owner bits): the pipelined rounds — round r's segments exchanged while round r − 1 is absorbed —
    with ranks that run out of reads at different rounds, 4-byte records (k = 21) and the wide round (k = 27)."""
    import sharkmer_amd as sa
    histo_max, n_reads = 40, 4300
    port = _free_port()
    mp.spawn(_owner_worker, args=(4, port, k, chunks, histo_max, n_reads, log_p1, 1024, False, str(tmp_path)), nprocs=4, join=True)
    bases, offsets = _owner_input(sa, n_reads, False)
    ref = orc.run_batch(bases, offsets, k, chunks, histo_max)
    hs = [np.load(tmp_path / f"hist_{r}.npy") for r in range(4)]
    assert all(np.array_equal(hs[0], h) for h in hs[1:])
    assert np.array_equal(hs[0], ref.histograms())
    t0 = np.load(tmp_path / "tot_0.npy")
    st = ref.stats
    assert list(t0[:5]) == [st["n_reads_ingested"], st["n_bases_read"], st["n_bases_ingested"], st["n_kmers_ingested"], st["n_unique_kmers"]]


def _owner_error_worker(rank, world, port, k, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import sharkmer_amd as sa
    from sharkmer_amd.dist import OwnerCounter, shard_batches
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_reads = 6000
    bases, offsets = _owner_input(sa, n_reads, False)
    bases = bases.copy()
    bases[int(offsets[3100]) + 2] = ord("x")   # read 3100: rank 1's second batch (batches of 1000 go round robin)
    eng = _owner_engine(k, 2, 40, world, rank, 10 if k <= 21 else 10, 1024)
    oc = OwnerCounter(eng, dist, round_bases=1000 * 160)
    mine = shard_batches(n_reads, rank, world)
    outcome = "finished"
    try:
        for first, n in mine:
            oc.round((bases, offsets[first:first + n + 1], n, int(offsets[first + n] - offsets[first]), first))
        oc.finalize_histograms()
    except Exception as e:  # noqa: BLE001
        outcome = f"{type(e).__name__}: {e}"
    open(os.path.join(out_dir, f"outcome_{rank}.txt"), "w").write(outcome)
    dist.destroy_process_group()


@pytest.mark.parametrize("k", [21, 27])
def test_a_failing_rank_takes_every_rank_out_of_the_pipelined_rounds(tmp_path, k):
    """An invalid byte in ONE rank's batch of a later round — when an earlier round's segments are still waiting to be
    absorbed — ends the round for everybody: the failing rank with the reference's text, its peer with "a peer rank
    failed"; nobody is left waiting in a collective (the test would hang).  4-byte rounds and the wide ones."""
    port = _free_port()
    mp.spawn(_owner_error_worker, args=(2, port, k, str(tmp_path)), nprocs=2, join=True)
    o0 = (tmp_path / "outcome_0.txt").read_text()
    o1 = (tmp_path / "outcome_1.txt").read_text()
    assert "Invalid character 'x' in sequence. Only ACGTN allowed." in o1, o1
    assert "a peer rank failed" in o0, o0



@pytest.mark.parametrize("k,chunks,n_reads,log_p1,cap,poly", [(21, 10, 3300, 10, 1024, False), (15, 3, 2500, 4, 1024, True)])
def test_two_rank_owner_rounds_with_the_scatter_in_two_calls(orc, tmp_path, monkeypatch, k, chunks, n_reads, log_p1, cap, poly):
    """An engine that has shk_xchg_scatter_begin / _end (every device engine): OwnerCounter begins round r's scatter,
    launches the absorbs of round r − 1's segments — W of them — and only then asks for the scatter's outcome; the last
    round's segments are absorbed at finalize.  Same histograms, same totals, foreign spills (the skewed case) included."""
    import sharkmer_amd as sa
    monkeypatch.setenv("SHK_TEST_TWO_CALL_ENGINE", "1")
    port = _free_port()
    mp.spawn(_owner_worker, args=(2, port, k, chunks, 40, n_reads, log_p1, cap, poly, str(tmp_path)), nprocs=2, join=True)
    bases, offsets = _owner_input(sa, n_reads, poly)
    ref = orc.run_batch(bases, offsets, k, chunks, 40)
    h0, h1 = np.load(tmp_path / "hist_0.npy"), np.load(tmp_path / "hist_1.npy")
    assert np.array_equal(h0, h1) and np.array_equal(h0, ref.histograms())
    t0 = np.load(tmp_path / "tot_0.npy")
    assert list(t0[:5]) == [ref.stats[x] for x in ("n_reads_ingested", "n_bases_read", "n_bases_ingested", "n_kmers_ingested", "n_unique_kmers")]
    if poly:
        assert t0[5] > 0   # rounds with foreign spills
    n_rounds = -(-n_reads // 2000)
    for rank in (0, 1):
        calls = (tmp_path / f"calls_{rank}.txt").read_text().split()
        assert calls == ["begin", "end"] + ["begin", "absorb", "absorb", "end"] * (n_rounds - 1) + ["absorb", "absorb"], calls


def test_a_byte_found_at_the_scatter_s_end_takes_every_rank_out(tmp_path, monkeypatch):
    """The two-call scatter reports an invalid byte from its END, after the previous round's absorbs have been launched:
    the failing rank raises the reference's text, its peer "a peer rank failed", nobody hangs."""
    monkeypatch.setenv("SHK_TEST_TWO_CALL_ENGINE", "1")
    port = _free_port()
    mp.spawn(_owner_error_worker, args=(2, port, 21, str(tmp_path)), nprocs=2, join=True)
    assert "Invalid character 'x' in sequence. Only ACGTN allowed." in (tmp_path / "outcome_1.txt").read_text()
    assert "a peer rank failed" in (tmp_path / "outcome_0.txt").read_text()
