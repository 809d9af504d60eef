"""The histogram a FRESH page pass leaves behind (k_pages32<true, true> + k_hist_reduce) against the scan of the table
(k_histo) and against the oracle: same columns (counting.rs:171-202, histogram.rs:51-85, 125-134), same totals —
whatever happens between the pass and the finalize.  SHK_FUSED_HIST is read at every launch: 0 = the scan only,
1 = the library's choice, 2 = the pass wherever it can."""
import numpy as np
import pytest

import sharkmer_amd as sa
from test_gpu_parity import check_against_oracle

pytestmark = pytest.mark.gpu


def genome_reads(rng, genome_len, n_reads, read_len, p_n=0.0):
    g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=genome_len)]
    starts = rng.integers(0, genome_len - read_len + 1, size=n_reads)
    bases = np.concatenate([g[s:s + read_len] for s in starts]).copy()
    if p_n:
        bases[rng.random(len(bases)) < p_n] = ord("N")
    offsets = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len))
    return bases, offsets


HINT = 3_000_000  # → 2^23 slots, 1024 pages: 4-byte records at k = 21 (the paged route a fresh pass belongs to)


def run(orc, bases, offsets, k, chunks, histo_max, hint=HINT):
    """One job through the paged route → (histograms, counters, which kernels made the histogram)."""
    ref = orc.run_batch(bases, offsets, k, chunks, histo_max)
    with sa.KmerEngine(k, chunks, histo_max, capacity_hint=hint, flags=sa.FLAG_FORCE_PAGED | sa.FLAG_TIMING) as eng:
        eng.ingest_reads(bases, offsets)
        eng.finalize()
        got, cnt, tim = eng.histograms(), eng.counters(), eng.timings()
    assert np.array_equal(got, ref.histograms())
    st = ref.stats
    for f in ("n_kmers_ingested", "n_unique_kmers", "n_hashed_kmers", "n_bases_ingested"):
        assert cnt[f] == st[f], f
    if chunks > 0:
        assert cnt["n_singleton_kmers"] == st["n_singleton_kmers"]
    return tim


@pytest.mark.parametrize("mode", ["0", "2"])
@pytest.mark.parametrize("k,chunks,histo_max,genome,n_reads", [
    (21, 1, 10000, 20000, 4000),     # 20x, one lane
    (21, 10, 10000, 20000, 12000),   # ten cumulative columns (blocks of 1000 reads go round the lanes)
    (21, 3, 10000, 200000, 1600000),  # 640x: sums far past the pass's LDS bins — the straight-to-histogram adds
    (21, 1, 7, 20000, 4000),         # histo_max below every sum: the fold bin (histogram.rs:125-134)
    (19, 4, 600, 2000, 9000),        # histo_max just past the LDS bins
    (21, 0, 10000, 20000, 3000),     # chunks = 0: totals only
])
def test_pass_and_scan_agree_with_the_oracle(orc, monkeypatch, mode, k, chunks, histo_max, genome, n_reads):
    monkeypatch.setenv("SHK_FUSED_HIST", mode)
    monkeypatch.setenv("SHK_SLICE_KB", str(1 << 20))  # (a host batch as ONE launch: a second slice's pass would overtake the first one's histogram)
    rng = np.random.default_rng(k * 1000 + chunks)
    bases, offsets = genome_reads(rng, genome, n_reads, 100, p_n=0.002)
    tim = run(orc, bases, offsets, k, chunks, histo_max)
    assert ("histo_rows" in tim) == (mode == "2"), tim   # the route that was asked for is the one that ran
    assert ("histo" in tim) == (mode == "0"), tim


def test_the_library_s_own_choice(orc):
    """One lane on a small table and every many-lane table: the pass; one lane on a large table: the scan."""
    rng = np.random.default_rng(2)
    bases, offsets = genome_reads(rng, 20000, 3000, 100)
    assert "histo_rows" in run(orc, bases, offsets, 21, 1, 10000)
    assert "histo_rows" in run(orc, bases, offsets, 21, 5, 10000)
    assert "histo" in run(orc, bases, offsets, 21, 1, 10000, hint=30_000_000)     # 2^26 slots, 8192 pages
    assert "histo_rows" in run(orc, bases, offsets, 21, 2, 10000, hint=30_000_000)


def test_whatever_comes_between_the_pass_and_the_finalize(orc, monkeypatch):
    """A second ingest, a lookup, an insert, a reset: each makes the pass's histogram stale (or must leave it alone)."""
    monkeypatch.setenv("SHK_FUSED_HIST", "2")
    rng = np.random.default_rng(5)
    k, chunks, hm = 21, 2, 10000
    bases, offsets = genome_reads(rng, 30000, 8000, 100)
    # two ingests: the first one's (fresh) pass has left a histogram that the second one overtakes
    check_against_oracle(orc, bases, offsets, k, chunks, hm, flags=sa.FLAG_FORCE_PAGED, hint=HINT, splits=[3000])
    ref = orc.run_batch(bases, offsets, k, chunks, hm)
    with sa.KmerEngine(k, chunks, hm, capacity_hint=HINT, flags=sa.FLAG_FORCE_PAGED | sa.FLAG_TIMING) as eng:
        eng.ingest_reads(bases, offsets)
        rk, rc = ref.merged().export()
        assert np.array_equal(eng.lookup(rk[:100]), rc[:100])  # a reader in between
        eng.finalize()
        assert np.array_equal(eng.histograms(), ref.histograms())
        eng.finalize()                                           # again: nothing is added twice
        assert np.array_equal(eng.histograms(), ref.histograms())
        for _ in range(3):                                       # job after job on one context
            eng.reset()
            eng.ingest_reads(bases, offsets)
            eng.finalize()
            assert np.array_equal(eng.histograms(), ref.histograms())
            assert eng.counters()["n_unique_kmers"] == ref.stats["n_unique_kmers"]
        # an insert behind the pass (KmerCounts::insert, counting.rs:152-154)
        eng.reset()
        eng.ingest_reads(bases, offsets)
        eng.insert(np.array([rk[0]], dtype=np.uint64), np.array([5], dtype=np.uint32), chunk_id=1)
        eng.finalize()
        h = eng.histograms()
    # the k-mer's merged count goes from c0 to c0 + 5 in column 1 (lanes 0 and 1 merged); column 0 is lane 0 alone
    want = ref.histograms().copy()
    c0 = int(rc[0])
    want[1][c0] -= 1
    want[1][c0 + 5] += 1
    assert np.array_equal(h, want)


def test_deep_coverage_switches_the_pass_off_for_later_jobs(orc, monkeypatch):
    """Sums past the LDS bins are same-line global adds: a job full of them (≥ 2^16) leaves the histogram to the scan
    from then on — exact either way."""
    monkeypatch.setenv("SHK_FUSED_HIST", "2")
    monkeypatch.setenv("SHK_SLICE_KB", str(1 << 20))
    rng = np.random.default_rng(9)
    k, hm = 21, 100000
    g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=70020)]
    # 70 000 distinct k-mers, every one 600 times: 600 copies of the same long read
    one = g.copy()
    bases = np.tile(one, 600)
    offsets = (np.arange(601, dtype=np.uint64) * np.uint64(len(one)))
    ref = orc.run_batch(bases, offsets, k, 1, hm)
    with sa.KmerEngine(k, 1, hm, capacity_hint=HINT, flags=sa.FLAG_FORCE_PAGED | sa.FLAG_TIMING) as eng:
        routes = []
        for job in range(3):
            eng.reset()
            eng.reset_timings()
            eng.ingest_reads(bases, offsets)
            eng.finalize()
            assert np.array_equal(eng.histograms(), ref.histograms()), job
            assert eng.counters()["n_hashed_kmers"] == ref.stats["n_hashed_kmers"]
            routes.append("histo_rows" in eng.timings())
        assert routes == [True, False, False], routes
