"""GPU tests of KEY-SPACE-PARTITIONED ingest (owner shares; SURVEY.md §8e's alternative, BASELINE
configs[4]): a context with shk_config.n_owners = W holds the k-mers of ONE owner in W.  Through the C ABI,
bit-exact against the oracle:

  * drop mode  — every share is handed all reads and keeps what it owns: the shares' histograms add up
                 to the oracle's, their tables are disjoint and their union is the oracle's table;
  * exchange   — W contexts on one card play W ranks, every rank ingests its own reads and the records are
                 exchanged by owner (sharkmer_amd.dist.OwnerCounter over an in-process transport);
  * one owner's share of configs[4] at full table size (2^30 slots × 48 B), foreign records dropped.
"""
import os
import threading

import numpy as np
import pytest
import torch

import sharkmer_amd as sa
from sharkmer_amd.dist import OwnerCounter, shard_batches

from test_gpu_dist import ThreadGroup

pytestmark = pytest.mark.gpu


def _owner_of(keys, k, W):
    """Owner of canonical k-mers = top log2(W) bits of the engine's bijective key mix (shk_device.hip.h,
    mix_key: one multiplication by an odd constant mod 2^2k)."""
    bits = 2 * k
    lw = W.bit_length() - 1
    if not lw or bits < lw:
        return np.zeros(len(keys), dtype=np.int64)
    M = 0xC2B2AE35 if bits <= 42 else 0x9E3779B97F4A7C15
    with np.errstate(over="ignore"):
        y = (np.asarray(keys, dtype=np.uint64) * np.uint64(M)) & np.uint64((1 << bits) - 1)
    return (y >> np.uint64(bits - lw)).astype(np.int64)


def _shares_against_oracle(orc, bases, offsets, k, chunks, histo_max, W, flags=0, hint=0, splits=None, timings=None):
    ref = orc.run_batch(bases, offsets, k, chunks, histo_max)
    rk, rc = ref.merged().export()
    hist_sum = None
    all_k, all_c = [], []
    tot = dict(n_kmers_ingested=0, n_unique_kmers=0, n_hashed_kmers=0)
    for o in range(W):
        with sa.KmerEngine(k, chunks, histo_max, capacity_hint=hint, flags=flags, n_owners=W, owner_id=o) as eng:
            n = len(offsets) - 1
            cuts = [0] + sorted(splits or []) + [n]
            for a, b in zip(cuts[:-1], cuts[1:]):
                eng.ingest_reads(bases, offsets[a:b + 1])
            eng.finalize()
            h = eng.histograms()
            c = eng.counters()
            if timings is not None:
                timings.append(eng.timings())
            gk, gc = eng.export_table()
            # point lookups: owned keys give their count, foreign keys 0
            probe = rk[:: max(len(rk) // 300, 1)]
            got = eng.lookup(probe)
            own = _owner_of(probe, k, W) == o
            want = np.where(own, rc[:: max(len(rk) // 300, 1)], 0)
            assert np.array_equal(got, want.astype(np.uint32))
        assert c["n_reads_ingested"] == ref.stats["n_reads_ingested"]
        assert c["n_bases_ingested"] == ref.stats["n_bases_ingested"]
        assert np.all(_owner_of(gk, k, W) == o), "a share holds a k-mer it does not own"
        hist_sum = h.astype(np.uint64) if hist_sum is None else hist_sum + h
        all_k.append(gk)
        all_c.append(gc)
        for f in tot:
            tot[f] += c[f]
    assert np.array_equal(hist_sum, ref.histograms())
    uk = np.concatenate(all_k)
    uc = np.concatenate(all_c)
    order = np.argsort(uk, kind="stable")
    assert np.array_equal(uk[order], rk) and np.array_equal(uc[order], rc)
    for f in tot:
        assert tot[f] == ref.stats[f], f


@pytest.mark.parametrize("W", [2, 8])
@pytest.mark.parametrize("k,chunks,flags,hint", [(21, 10, 0, 0), (21, 3, sa.FLAG_FORCE_DIRECT, 0), (31, 2, 0, 0),
                                                 (15, 0, 0, 0), (21, 1, 0, 4_200_000), (9, 4, 0, 0)])
def test_owner_shares_drop_mode(orc, k, chunks, flags, hint, W):
    """Every share sees every read and keeps its own k-mers (global atomics on small tables / long
    k-mers, the owner layout where 4-byte records fit)."""
    spec = sa.SynthSpec(genome_len=60_000, sub_per_64k=300, n_per_64k=60)
    bases, offsets = sa.synth_reads(spec, 0, 12_500)
    _shares_against_oracle(orc, bases, offsets, k, chunks, 200, W, flags=flags, hint=hint, splits=[3_100, 9_000])


@pytest.mark.parametrize("W,k,chunks", [(1, 31, 1), (2, 31, 3), (8, 27, 10), (4, 23, 0)])
def test_a_share_s_own_reads_at_k_over_21_take_the_8_byte_owner_layout(orc, W, k, chunks):
    """Drop mode at k > 21: the share's own ingest goes through k_scatter64's owner layout (its own records only, one
    segment), the level-2 pass and the page passes — no global atomics (they were the only way until round 4)."""
    spec = sa.SynthSpec(genome_len=60_000, sub_per_64k=300, n_per_64k=60)
    bases, offsets = sa.synth_reads(spec, 0, 12_500)
    tims = []
    _shares_against_oracle(orc, bases, offsets, k, chunks, 200, W, flags=sa.FLAG_TIMING, hint=4_200_000, splits=[3_100, 9_000], timings=tims)
    for t in tims:
        assert "direct" not in t and "scatter" in t and "pscan" in t and "pages" in t, t


@pytest.mark.parametrize("W,k,chunks,lvl1,hint,budget", [(2, 21, 10, 10, 4_200_000, 0), (4, 21, 3, 10, 2_000_000, 600_000),
                                                         (8, 19, 2, 6, 1_000_000, 0), (2, 17, 16, 9, 600_000, 300_000),
                                                         (4, 21, 1, 10, 16_000_000, 0)])
def test_owner_layout_route(orc, monkeypatch, W, k, chunks, lvl1, hint, budget):
    """The owner layout forced on small inputs: level 1 over [owner][lane][super-page] regions, level 2
    into the waiting (lane, page) regions, deferred page passes; several ingest calls per share."""
    monkeypatch.setenv("SHK_LEVEL1_LOG", str(lvl1))
    if budget:
        monkeypatch.setenv("SHK_DEFER_BUDGET", str(budget))
    spec = sa.SynthSpec(genome_len=150_000, sub_per_64k=200, n_per_64k=40)
    bases, offsets = sa.synth_reads(spec, 0, 21_000)
    with sa.KmerEngine(k, chunks, 100, capacity_hint=hint, flags=sa.FLAG_TIMING, n_owners=W, owner_id=W - 1) as eng:
        eng.ingest_reads(bases, offsets)
        t = eng.timings()
        assert "scatter" in t and "pscan" in t, t  # the route under test ran (level 1 + level 2)
    _shares_against_oracle(orc, bases, offsets, k, chunks, 100, W, hint=hint, splits=[2_000, 9_500, 15_250])


def test_owner_share_skewed_input_spills_exactly(orc, monkeypatch):
    """Low-complexity reads overflow their (lane, super-page) region: the excess takes the spill path."""
    monkeypatch.setenv("SHK_LEVEL1_LOG", "8")
    rng = np.random.default_rng(3)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [b"A" * 150 if i % 3 else lut[rng.integers(0, 4, size=150)].tobytes() for i in range(9_000)]
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offsets = np.arange(len(seqs) + 1, dtype=np.uint64) * 150
    _shares_against_oracle(orc, bases, offsets, 19, 4, 1000, 2, hint=2_000_000)


# ---- exchange mode: W contexts on one card as W ranks ------------------------------------------------

def _exchange_run(orc, bases, offsets, k, chunks, histo_max, W, hint, reserve_cus=0, timing=False):
    """Every rank ingests its own 1000-read batches (round-robin, shard_batches), one batch per round."""
    n_reads = len(offsets) - 1
    ref = orc.run_batch(bases, offsets, k, chunks, histo_max) if orc is not None else None
    shared = ThreadGroup.Shared(W)
    results, errors = [None] * W, []
    d_bases = torch.from_numpy(bases.copy()).cuda()
    n_rounds = len(shard_batches(n_reads, 0, W))  # rank 0 has the most batches

    def run(rank):
        try:
            eng = sa.KmerEngine(k, chunks, histo_max, capacity_hint=hint, n_owners=W, owner_id=rank, reserve_cus=reserve_cus,
                                flags=sa.FLAG_TIMING if timing else 0)
            oc = OwnerCounter(eng, ThreadGroup(shared, rank), device=0, round_bases=1000 * 160)
            mine = shard_batches(n_reads, rank, W)
            keep = []
            for r in range(n_rounds):
                if r >= len(mine):
                    oc.round(None)  # out of reads: still takes part in the round
                    continue
                first, n = mine[r]
                o0, o1 = int(offsets[first]), int(offsets[first + n])
                offs = torch.from_numpy((offsets[first:first + n + 1] - offsets[first]).astype(np.int64)).cuda()
                keep.append(offs)
                oc.round((d_bases[o0:o1].data_ptr(), offs.data_ptr(), n, o1 - o0, first))
            results[rank] = (oc.finalize_histograms(), oc.totals, eng.counters()["n_spilled"], oc.n_foreign_rounds, eng.timings() if timing else None)
            eng.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    st = ref.stats
    for hist, tot, _, _, _ in results:
        assert np.array_equal(hist, ref.histograms())
        for f in ("n_kmers_ingested", "n_unique_kmers", "n_reads_ingested", "n_bases_ingested", "n_hashed_kmers"):
            assert tot[f] == st[f], f
    return results


@pytest.mark.parametrize("W,k,chunks,lvl1", [(2, 21, 10, 10), (4, 21, 3, 10), (2, 17, 1, 5), (8, 19, 0, 6),
                                             (2, 31, 10, 10), (4, 27, 3, 10), (8, 23, 0, 10), (2, 22, 1, 10)])
def test_exchange_between_contexts_like_ranks(orc, monkeypatch, W, k, chunks, lvl1):
    """k > 21: 2k − 10 level-1 bits > 32 — no 4-byte record holds the rest of the key, shk_xchg_feasible says so and the
    rounds take the wide route (whole k-mers grouped by owner, shk_xchg_wide_scatter_device + shk_insert_device)."""
    monkeypatch.setenv("SHK_LEVEL1_LOG", str(lvl1))
    spec = sa.SynthSpec(genome_len=80_000, sub_per_64k=250, n_per_64k=50)
    bases, offsets = sa.synth_reads(spec, 0, 20_500)
    _exchange_run(orc, bases, offsets, k, chunks, 300, W, hint=4_200_000 if lvl1 == 10 else 1_000_000)  # (2k - lvl1 ≤ 32: 4-byte records)


@pytest.mark.parametrize("W,k,chunks", [(2, 31, 10), (4, 27, 3), (8, 23, 1), (1, 31, 2), (2, 22, 0)])
def test_wide_round_receiver_through_the_paged_passes(orc, monkeypatch, W, k, chunks):
    """The receiver of a wide round counts a list of a million k-mers and more through the partition + page passes
    (shk_insert_device's list route: level 1 by k_part_rescatter in its list mode, a pass per chunk lane, foreign owners'
    k-mers skipped) instead of k_insert's global atomics.  SHK_INSERT_PAGED_MIN=1 takes these small rounds that way;
    the result is the oracle's either way."""
    monkeypatch.setenv("SHK_LEVEL1_LOG", "10")
    monkeypatch.setenv("SHK_INSERT_PAGED_MIN", "1")
    monkeypatch.setenv("SHK_XL64", "0")   # (the wide round: without this, k > 21 takes the owner layout with 8-byte records)
    spec = sa.SynthSpec(genome_len=80_000, sub_per_64k=250, n_per_64k=50)
    bases, offsets = sa.synth_reads(spec, 0, 20_500)
    res = _exchange_run(orc, bases, offsets, k, chunks, 300, W, hint=4_200_000, timing=True)
    for r in res:
        assert "pcount" in r[4], r[4]                                  # the wide round's count-by-owner pass ran
        assert "insert" not in r[4] or r[4]["insert"][1] <= 2, r[4]   # (a repair insert at most: the rounds went the paged way)
        assert "pages" in r[4], r[4]
    monkeypatch.setenv("SHK_INSERT_PAGED", "0")
    res = _exchange_run(orc, bases, offsets, k, chunks, 300, W, hint=4_200_000, timing=True)
    assert all("insert" in r[4] for r in res)


@pytest.mark.parametrize("W,k,chunks", [(2, 31, 10), (4, 27, 3), (8, 23, 1), (1, 31, 2), (2, 22, 0), (2, 31, 32)])
def test_owner_layout_with_8_byte_records(orc, monkeypatch, W, k, chunks):
    """k > 21 between owner shares: the exchange rounds carry the same [owner][lane][super-page] segments as for k ≤ 21,
    with the canonical k-mer as the record (shk_xchg_layout.record_bytes = 8: k_scatter64's owner layout at the sender,
    k_part_rescatter + k_pages at the receiver) — no wide round, no insert by global atomics."""
    monkeypatch.setenv("SHK_LEVEL1_LOG", "10")
    spec = sa.SynthSpec(genome_len=80_000, sub_per_64k=250, n_per_64k=50)
    bases, offsets = sa.synth_reads(spec, 0, 20_500)
    with sa.KmerEngine(k, chunks, 300, capacity_hint=4_200_000, n_owners=W, owner_id=0) as eng:
        assert eng.xchg_feasible()
    res = _exchange_run(orc, bases, offsets, k, chunks, 300, W, hint=4_200_000, timing=True)
    for r in res:
        assert "pcount" not in r[4] and "scatter" in r[4] and "pscan" in r[4] and "pages" in r[4], r[4]
        assert "insert" not in r[4] or r[4]["insert"][1] <= 2, r[4]


def test_a_list_of_k_mers_with_lanes_either_way(orc, monkeypatch):
    """shk_insert_device on one context: k-mer occurrences with their chunk lanes, the list route against k_insert and
    against counting by hand; then more reads on top, and a grow in between (no capacity hint)."""
    rng = np.random.default_rng(3)
    k, chunks = 31, 3
    distinct = rng.integers(0, 1 << 62, size=40_000, dtype=np.uint64)
    kmers = distinct[rng.integers(0, len(distinct), size=300_000)]
    lanes = rng.integers(0, chunks, size=len(kmers)).astype(np.uint32)
    tables = []
    for paged in ("1", "0"):
        monkeypatch.setenv("SHK_INSERT_PAGED", paged)
        monkeypatch.setenv("SHK_INSERT_PAGED_MIN", "1")
        with sa.KmerEngine(k, chunks, 1000, capacity_hint=40_000_000, flags=sa.FLAG_TIMING) as eng:   # 2^27 slots: two levels
            eng.insert_tensors(torch.from_numpy(kmers.view(np.int64)).cuda(), torch.from_numpy(lanes.view(np.int32)).cuda(), None)
            eng.insert_tensors(torch.from_numpy(kmers[:1000].view(np.int64)).cuda(), torch.from_numpy(lanes[:1000].view(np.int32)).cuda(), None)
            eng.finalize()
            tim = eng.timings()
            assert ("insert" in tim) == (paged == "0"), tim
            tables.append((eng.export_table(), eng.histograms().copy()))
    (k1, c1), h1 = tables[0]
    (k0, c0), h0 = tables[1]
    assert np.array_equal(k1, k0) and np.array_equal(c1, c0) and np.array_equal(h1, h0)
    want_k, want_c = np.unique(np.concatenate([kmers, kmers[:1000]]), return_counts=True)
    assert np.array_equal(k1, want_k) and np.array_equal(c1, want_c.astype(np.uint32))


@pytest.mark.parametrize("reserve", [sa.RESERVE_NONE, 8, 100, 250])
def test_exchange_with_compute_units_left_to_the_collectives(orc, monkeypatch, reserve):
    """shk_config.reserve_cus: the persistent scatter of an owner share starts that many workgroups fewer (the default for
    a share of a multi-GPU job is 16; at most all but 64 of the card's).  The result does not depend on it."""
    monkeypatch.setenv("SHK_LEVEL1_LOG", "10")
    spec = sa.SynthSpec(genome_len=90_000, sub_per_64k=250, n_per_64k=50)
    bases, offsets = sa.synth_reads(spec, 0, 30_500)
    _exchange_run(orc, bases, offsets, 21, 3, 300, 2, hint=4_200_000, reserve_cus=reserve)


def test_exchange_with_skewed_input_goes_through_the_foreign_spill_list(orc, monkeypatch):
    monkeypatch.setenv("SHK_LEVEL1_LOG", "8")
    rng = np.random.default_rng(9)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [b"ACAC" * 37 + b"AC" if i % 2 else lut[rng.integers(0, 4, size=150)].tobytes() for i in range(8_000)]
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offsets = np.arange(len(seqs) + 1, dtype=np.uint64) * 150
    res = _exchange_run(orc, bases, offsets, 19, 2, 5000, 2, hint=1_000_000)
    assert any(r[3] > 0 for r in res), "the skew was meant to overflow a level-1 region"


@pytest.mark.parametrize("late,one_call", [(1, False), (0, False), (1, True)])
def test_exchange_with_a_hot_page_spills_at_the_absorb(orc, monkeypatch, late, one_call):
    """A k-mer that is a few per cent of every round fits its level-1 region round by round and overflows its PAGE's
    waiting region over the rounds of a window: the level-2 pass (absorb) spills, and what it spilled is looked at when
    the next round's scatter reads the statistics (SHK_XCHG_LATE_SETTLE, the default: the scatter is launched without
    waiting for the absorb in front of it, and the read-start kernel leaves the spill counter alone) or before that
    scatter is launched (= 0: round 3's order); with the scatter in two calls (shk_xchg_scatter_begin / _end, what
    OwnerCounter uses: the absorbs are launched between the two) or in one (SHK_DIST_ONE_CALL_SCATTER).  Exact every way,
    and the spill path was taken."""
    monkeypatch.setenv("SHK_LEVEL1_LOG", "5")         # (few, long level-1 regions: the hot k-mer fits them round by round)
    monkeypatch.setenv("SHK_ACC_MAX_MRECORDS", "1")   # windows of 2^20 records: page regions of a few thousand
    monkeypatch.setenv("SHK_XCHG_LATE_SETTLE", str(late))
    if one_call:
        monkeypatch.setenv("SHK_DIST_ONE_CALL_SCATTER", "1")
    rng = np.random.default_rng(11)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [b"ACAC" * 37 + b"AC" if i % 10 == 0 else lut[rng.integers(0, 4, size=150)].tobytes() for i in range(24_000)]
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offsets = np.arange(len(seqs) + 1, dtype=np.uint64) * 150
    res = _exchange_run(orc, bases, offsets, 19, 2, 5000, 2, hint=3_000_000)
    assert any(r[2] > 0 for r in res), "the hot page was meant to overflow its waiting region (n_spilled)"


def test_scatter_in_two_calls_keeps_its_order():
    """shk_xchg_scatter_end without a begin, and a second begin before the end: state errors, nothing launched."""
    eng = sa.KmerEngine(21, 1, 100, capacity_hint=4_200_000, n_owners=1, owner_id=0)
    bases, offsets = sa.synth_reads(sa.SynthSpec(genome_len=50_000), 0, 2_000)
    d_b = torch.from_numpy(bases.copy()).cuda()
    d_o = torch.from_numpy(offsets.astype(np.int64)).cuda()
    with pytest.raises(sa.ShkError, match="no exchange scatter is begun"):
        eng.xchg_scatter_end()
    rec, cur, lay = eng.xchg_scatter_begin_tensors(d_b.data_ptr(), d_o.data_ptr(), 2_000, len(bases))
    with pytest.raises(sa.ShkError, match="begun and not ended"):
        eng.xchg_scatter_begin_tensors(d_b.data_ptr(), d_o.data_ptr(), 2_000, len(bases))
    assert eng.xchg_scatter_end() == 0
    assert int(cur.sum().item()) == 2_000 * 130          # every k-mer is in a region of the one segment
    eng.xchg_absorb_tensors(rec, cur, lay)
    eng.finalize()
    assert eng.counters()["n_kmers_ingested"] == 2_000 * 130
    eng.close()


@pytest.mark.parametrize("k", [19, 31])
def test_exchange_invalid_byte_fails_every_rank(monkeypatch, k):
    monkeypatch.setenv("SHK_LEVEL1_LOG", "8")
    spec = sa.SynthSpec(genome_len=50_000)
    bases, offsets = sa.synth_reads(spec, 0, 4_000)
    bad = bases.copy()
    bad[int(offsets[2_500]) + 3] = ord("x")
    with pytest.raises(AssertionError, match="Invalid character 'x'"):
        _exchange_run(None, bad, offsets, k, 1, 100, 2, hint=1_000_000)


def test_exchange_feasibility_is_a_function_of_the_configuration(monkeypatch):
    for k, want in ((21, True), (22, True), (31, True), (15, True)):   # (k > 21: the 8-byte owner layout, round 4)
        with sa.KmerEngine(k, 3, 100, capacity_hint=4_200_000, n_owners=4, owner_id=1) as eng:
            assert eng.xchg_feasible() == want, k
    with sa.KmerEngine(31, 40, 100, capacity_hint=4_200_000, n_owners=4, owner_id=1) as eng:   # more than 32 lanes at k > 21: the wide round
        assert not eng.xchg_feasible()
    monkeypatch.setenv("SHK_XL64", "0")
    for k, want in ((21, True), (22, False), (31, False)):
        with sa.KmerEngine(k, 3, 100, capacity_hint=4_200_000, n_owners=4, owner_id=1) as eng:
            assert eng.xchg_feasible() == want, k
    with sa.KmerEngine(21, 3, 100) as eng:   # not a share at all
        assert not eng.xchg_feasible()
        with pytest.raises(sa.ShkError, match="not an owner share"):
            eng.xchg_wide_scatter_device(0, 0, 0, 0)


# ---- one owner's share of BASELINE configs[4] at its full table size ----------------------------------

def test_config5_owner_share_on_one_card(orc):
    """BASELINE.json configs[4] — 10 cumulative subsets (chunk lanes) of reads over a 3 Gb genome on 8 GPUs —
    as ONE owner's share on one card: owner 5 of 8, 10 lanes, a 2^30-slot table × (8 + 10·4) B = 51 GB (the
    whole key space would need 412 GB per rank).  Every read is offered, the seven other owners' records are
    dropped in the level-1 pass (the same kernels the 8-GPU exchange runs).  SHK_SHARE_READS reads: by default ALL
    of the config's 10^9 — every record this owner would receive in the 8-GPU run (80 s on the box, most of it the
    oracle's extractor over 150 Gbases on 16 CPUs); 125000000 = one rank's share, for a quick look.
    Checks: an EXACT probe set per chunk lane — 10^5 k-mers of sampled reads counted over all reads by the
    oracle's extractor; owned probes must come back with exactly that merged count, foreign ones with 0 —
    plus the size-independent properties of the incremental histograms."""
    from probe_util import ProbeChecker
    n = int(os.environ.get("SHK_SHARE_READS", "1000000000"))
    L, k, batch, W, owner, chunks = 150, 21, 1_700_000, 8, 5, 10
    n = n // batch * batch
    spec = sa.SynthSpec(genome_len=3_000_000_000, read_len=L)
    pc = ProbeChecker(orc, k, chunks, L, n_probes=120_000)
    with sa.KmerEngine(k, chunks, 1000, capacity_hint=3_000_000_000 // W, n_owners=W, owner_id=owner) as eng:
        n_pages, page_slots, n_lanes = eng.table_geometry()
        assert n_pages * page_slots == 1 << 30 and n_lanes == chunks
        d_bases = eng.alloc_device(batch * L)
        d_off = eng.alloc_device((batch + 1) * 8)
        try:
            for s_batch in (0, n // batch // 2, n // batch - 1):
                eng.synth_reads_device(spec, s_batch * batch + 4321, 320, d_bases, d_off)
                eng.sync()
                pc.add_sample(pc._fetch(eng, d_bases, 320), 320)
            probes = pc.freeze()
            for b in range(n // batch):
                eng.synth_reads_device(spec, b * batch, batch, d_bases, d_off)
                eng.ingest_reads_device(d_bases, d_off, batch, batch * L)
                pc.count_async(pc._fetch(eng, d_bases, batch), batch, b * batch)
            eng.finalize()
            h = eng.histograms()
            c = eng.counters()
            got = eng.lookup(probes)
        finally:
            eng.sync()
            eng.free_device(d_bases)
            eng.free_device(d_off)
    per_lane = pc.result()
    pc.close()
    own = _owner_of(probes, k, W) == owner
    assert 0.10 < own.mean() < 0.15                                   # owners are hash bits: 1/8 of the probes
    want = np.where(own, np.minimum(per_lane.sum(axis=0), 0xFFFFFFFF), 0).astype(np.uint32)
    assert np.array_equal(got, want)                                  # exact, k-mer by k-mer; foreign k-mers absent
    total = (L - k + 1) * n
    assert c["n_reads_ingested"] == n and c["n_bases_ingested"] == n * L
    assert abs(c["n_kmers_ingested"] - total / W) < 0.002 * total / W  # this owner's eighth of the occurrences
    assert c["n_hashed_kmers"] == c["n_kmers_ingested"] and c["n_grows"] == 0
    assert int(h[-1].sum()) == c["n_unique_kmers"]
    last = h[-1].astype(object)
    assert sum(int(f) * i for i, f in enumerate(last)) == c["n_kmers_ingested"]
    for j in range(1, chunks):                                        # cumulative columns only ever gain k-mers
        assert int(h[j].sum()) >= int(h[j - 1].sum())
    # column j = the histogram after lanes 0..j: the probes' per-lane counts say which bin each owned probe is in
    cum = np.minimum(np.cumsum(per_lane[:, own], axis=0), 0xFFFFFFFF)
    for j in (0, chunks // 2, chunks - 1):
        bins = np.bincount(np.minimum(cum[j][cum[j] > 0], 1001).astype(np.int64), minlength=1002)
        assert (h[j][:1002] >= bins[:1002].astype(np.uint64)).all()
    expect = 3e9 * (1 - np.exp(-total / 3e9)) / W
    assert abs(c["n_unique_kmers"] - expect) < 0.01 * expect


# ---- BASELINE configs[4] as a WHOLE JOB on one card: the eight owner shares one after the other ------------------------

def test_config5_whole_job_eight_shares_in_sequence(orc):
    """BASELINE.json configs[4] — 10 cumulative subsets of 10^9 reads over a 3 Gb genome on 8 GPUs (io.rs:340-361: read i
    belongs to subset (i / 1000) % 10; io.rs:1023-1028: column j of the histogram is taken after subsets 0..j) — as a JOB:
    the eight owner shares (10 lanes, 2^30 slots x 48 B = 51 GB each; all eight at once would be 412 GB) counted ONE AFTER
    THE OTHER on one card, each offered every read of the job with the other owners' records dropped in the level-1 pass
    (the kernels the 8-GPU exchange runs), and the eight 10-column histograms SUMMED — bins are additive over disjoint
    key sets (counting.rs:157-166), which is what the 8-GPU run's final all-reduce does.  SHK_JOB5_READS: the job's reads
    (default 10^9).  Exact checks on the job's result:
      * column j: sum of freq x count = the k-mer occurrences of subsets 0..j = (j + 1) x n / 10 x 130, for every j;
      * columns only ever gain k-mers; the last column's distinct count against the random-placement expectation;
      * a probe set over ALL owners — 10^5 k-mers of sampled reads counted per subset over all reads by the oracle's
        extractor in ONE sweep of the reads (during the first share's pass): every probe is found in exactly one share,
        with exactly the merged count, and the summed histogram's columns hold at least the probes' bins."""
    from probe_util import ProbeChecker
    n = int(os.environ.get("SHK_JOB5_READS", "1000000000"))
    L, k, batch, W, chunks, hmax = 150, 21, 1_700_000, 8, 10, 1000
    n = n // 10_000 * 10_000                     # (whole rounds of the ten subsets: every subset gets n / 10 reads)
    n_batches = -(-n // batch)
    spec = sa.SynthSpec(genome_len=3_000_000_000, read_len=L)
    pc = ProbeChecker(orc, k, chunks, L, n_probes=100_000)
    hist_sum = np.zeros((chunks, hmax + 2), dtype=object)
    tot = {"n_kmers_ingested": 0, "n_unique_kmers": 0, "n_hashed_kmers": 0}
    found = None
    probes = None
    for owner in range(W):
        with sa.KmerEngine(k, chunks, hmax, capacity_hint=3_000_000_000 // W, n_owners=W, owner_id=owner) as eng:
            n_pages, page_slots, n_lanes = eng.table_geometry()
            assert n_pages * page_slots == 1 << 30 and n_lanes == chunks
            d_bases = eng.alloc_device(batch * L)
            d_off = eng.alloc_device((batch + 1) * 8)
            try:
                if owner == 0:
                    for s_batch in (0, n_batches // 2, n_batches - 1):
                        eng.synth_reads_device(spec, min(s_batch * batch + 4321, n - 320), 320, d_bases, d_off)
                        eng.sync()
                        pc.add_sample(pc._fetch(eng, d_bases, 320), 320)
                    probes = pc.freeze()
                    found = np.zeros(len(probes), dtype=np.int64)
                    merged_got = np.zeros(len(probes), dtype=np.uint64)
                for b in range(n_batches):
                    nb = min(batch, n - b * batch)
                    eng.synth_reads_device(spec, b * batch, nb, d_bases, d_off)
                    eng.ingest_reads_device(d_bases, d_off, nb, nb * L)
                    if owner == 0:               # the oracle's one sweep of the reads
                        pc.count_async(pc._fetch(eng, d_bases, nb), nb, b * batch)
                eng.finalize()
                h = eng.histograms()
                c = eng.counters()
                got = eng.lookup(probes)
            finally:
                eng.sync()
                eng.free_device(d_bases)
                eng.free_device(d_off)
        assert c["n_reads_ingested"] == n and c["n_hashed_kmers"] == c["n_kmers_ingested"] and c["n_grows"] == 0
        assert int(h[-1].sum()) == c["n_unique_kmers"]
        hist_sum += h.astype(object)
        for f in tot:
            tot[f] += c[f]
        found += (got > 0)
        merged_got += got.astype(np.uint64)
        own = _owner_of(probes, k, W) == owner
        assert not (got[~own] > 0).any()                                   # a share holds nothing of the other owners
        sa.release_cached_memory()
    per_lane = pc.result()
    pc.close()
    # ---- the job's result --------------------------------------------------------------------------------------
    occ = (L - k + 1) * n
    assert tot["n_kmers_ingested"] == occ == tot["n_hashed_kmers"]         # every occurrence counted by exactly one share
    per_subset = (L - k + 1) * (n // chunks)
    for j in range(chunks):
        col = hist_sum[j]
        assert int(col[-1]) == 0                                           # (50x coverage: nothing near histo_max, so the sums below are exact)
        assert sum(int(f) * i for i, f in enumerate(col)) == (j + 1) * per_subset, j
        if j:
            assert int(col.sum()) >= int(hist_sum[j - 1].sum())
    assert int(hist_sum[-1].sum()) == tot["n_unique_kmers"]
    expect = 3e9 * (1 - np.exp(-occ / 3e9))
    assert abs(tot["n_unique_kmers"] - expect) < 0.01 * expect
    # every probe in exactly one share, with exactly the merged count
    assert (found == 1).all()
    want = np.minimum(per_lane.sum(axis=0), 0xFFFFFFFF).astype(np.uint64)
    assert np.array_equal(merged_got, want)
    cum = np.minimum(np.cumsum(per_lane, axis=0), 0xFFFFFFFF)
    for j in (0, chunks // 2, chunks - 1):
        bins = np.bincount(np.minimum(cum[j][cum[j] > 0], hmax + 1).astype(np.int64), minlength=hmax + 2)
        assert all(int(hist_sum[j][i]) >= int(bins[i]) for i in range(hmax + 2))


# ---- BASELINE configs[3] as a WHOLE JOB on one card: eight owner shares, the real exchange, all 10^9 reads ------------

def test_config4_whole_job_on_one_card(monkeypatch):
    """What the 8-GPU run of configs[3] does, rank for rank, with the eight ranks as eight contexts (and threads) on ONE
    card: every rank an owner share on a 2^30-slot table (8 × 12.9 GB), 125 M reads per rank generated round by round
    (1.7 M reads = one exchange round), OwnerCounter's pipelined rounds with the real segment sizes (137 MB per owner
    and round) over the in-process transport — only the links are missing.  SHK_JOB_READS: the job's reads (default
    10^9; the deferred windows are capped at 1 G records per context so that eight of them fit one card).  Exact:
    130 k-mers per read over all ranks, Σ freq·count = k-mer occurrences, every rank the same histogram; the distinct
    count against the random-placement expectation."""
    monkeypatch.setenv("SHK_ACC_MAX_MRECORDS", "1024")
    W, L, k, round_reads = 8, 150, 21, 1_700_000
    n_total = int(os.environ.get("SHK_JOB_READS", "1000000000"))
    per_rank = n_total // W
    n_rounds = -(-per_rank // round_reads)
    spec = sa.SynthSpec(genome_len=3_000_000_000, read_len=L)
    shared = ThreadGroup.Shared(W)
    results, errors = [None] * W, []

    def run(rank):
        try:
            eng = sa.KmerEngine(k, 1, 1000, capacity_hint=3_000_000_000 // W, n_owners=W, owner_id=rank)
            assert eng.table_geometry()[0] * eng.table_geometry()[1] == 1 << 30
            oc = OwnerCounter(eng, ThreadGroup(shared, rank), device=0, round_bases=round_reads * L)
            d_bases = eng.alloc_device(round_reads * L)
            d_off = eng.alloc_device((round_reads + 1) * 8)
            try:
                for r in range(n_rounds):
                    n = min(round_reads, per_rank - r * round_reads)
                    first = rank * per_rank + r * round_reads   # (any assignment of reads to ranks gives the same job)
                    eng.synth_reads_device(spec, first, n, d_bases, d_off)
                    lay = oc.round((d_bases, d_off, n, n * L, first))
                    assert lay.n_owners == W
                hist = oc.finalize_histograms()
                results[rank] = (hist, dict(oc.totals), oc.n_foreign_rounds, eng.counters()["n_grows"])
            finally:
                eng.sync()
                eng.free_device(d_bases)
                eng.free_device(d_off)
                eng.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    sa.release_cached_memory()
    assert not errors, errors
    n = per_rank * W
    h0, tot, _, _ = results[0]
    for hist, t, n_foreign, n_grows in results:
        assert np.array_equal(hist, h0) and t == tot
        assert n_grows == 0
    assert tot["n_reads_ingested"] == n and tot["n_bases_ingested"] == n * L
    assert tot["n_kmers_ingested"] == (L - k + 1) * n == tot["n_hashed_kmers"]
    col = h0[0].astype(object)
    assert sum(int(f) * i for i, f in enumerate(col)) == (L - k + 1) * n or int(col[-1]) > 0   # (the last bin clamps counts > histo_max)
    assert int(h0[0].sum()) == tot["n_unique_kmers"]
    expect = 3e9 * (1 - np.exp(-(L - k + 1) * n / 3e9))
    assert abs(tot["n_unique_kmers"] - expect) < 0.01 * expect


# ---- a randomized sweep of the exchange rounds ---------------------------------------------------------------------------

@pytest.mark.parametrize("seed", range(int(os.environ.get("SHK_FUZZ_SEEDS", "24"))))
def test_random_exchange_configuration_against_the_oracle(orc, monkeypatch, seed):
    """World size, k (4-byte rounds or — where a record does not fit — the wide ones), chunk lanes, level-1 fan-out,
    table size and input drawn from a seed: W contexts on one card as W ranks through OwnerCounter's pipelined rounds,
    every rank's histogram and totals against the oracle."""
    rng = np.random.default_rng(77_000 + seed)
    W = int(rng.choice([1, 2, 2, 4, 8]))
    k = int(rng.choice([13, 15, 17, 19, 21, 21, 22, 23, 27, 31]))
    chunks = int(rng.choice([0, 1, 2, 3, 10, 16]))
    lvl1 = int(rng.choice([6, 8, 10]))
    hint = int(rng.choice([600_000, 1_000_000, 4_200_000]))
    monkeypatch.setenv("SHK_LEVEL1_LOG", str(lvl1))
    if rng.random() < 0.3:
        monkeypatch.setenv("SHK_DEFER_BUDGET", str(int(rng.choice([30_000, 200_000]))))
    n_reads = int(rng.integers(1_500, 14_000))
    spec = sa.SynthSpec(genome_len=int(rng.choice([3_000, 60_000, 400_000])), sub_per_64k=int(rng.choice([0, 250, 2000])),
                        n_per_64k=int(rng.choice([0, 50, 600])))
    bases, offsets = sa.synth_reads(spec, int(rng.integers(0, 10_000)), n_reads)
    if rng.random() < 0.25:   # low-complexity reads: one k-mer's records overflow their region → the foreign spill list
        bases = bases.copy()
        for i in range(0, n_reads, 5):
            bases[int(offsets[i]):int(offsets[i + 1])] = ord("ACGT"[i % 4])
    _exchange_run(orc, bases, offsets, k, chunks, 200, W, hint=hint)
