"""GPU test of the multi-GPU merge path on ONE card: two KmerEngine contexts play two ranks
and a thread-based stand-in for torch.distributed moves the tensors between them, so the real
HIP entry points (shk_table_reserve_pages, shk_owner_counts, shk_compact_owners, shk_merge_entries,
shk_table_device_ptrs, shk_merge_pages, shk_set_owned_pages, owned-range finalize) are checked against the oracle.  The RCCL transport
itself is exercised by bench.py --gpus N on the 8-GPU node."""
import os
import threading

import numpy as np
import pytest
import torch

import sharkmer_amd as sa
from sharkmer_amd.dist import DistCounter, shard_batches

pytestmark = pytest.mark.gpu


class _ReduceOp:
    SUM = "sum"
    MAX = "max"
    MIN = "min"


class ThreadGroup:
    """Minimal in-process torch.distributed look-alike: one instance per rank (thread)."""
    ReduceOp = _ReduceOp

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.s, self.rank = shared, rank

    def get_world_size(self):
        return self.s.world

    def get_rank(self):
        return self.rank

    def all_reduce(self, t, op=_ReduceOp.SUM):
        torch.cuda.synchronize()
        self.s.slots[self.rank] = t.clone()
        self.s.barrier.wait()
        stack = torch.stack(self.s.slots)
        res = stack.sum(0) if op == _ReduceOp.SUM else stack.max(0).values if op == _ReduceOp.MAX else stack.min(0).values
        self.s.barrier.wait()
        t.copy_(res)
        torch.cuda.synchronize()

    def all_gather(self, outs, t):
        torch.cuda.synchronize()
        self.s.slots[self.rank] = t.clone()
        self.s.barrier.wait()
        for src in range(self.s.world):
            outs[src].copy_(self.s.slots[src])
        torch.cuda.synchronize()
        self.s.barrier.wait()

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None):
        torch.cuda.synchronize()
        W = self.s.world
        n = inp.numel() // W
        ins = input_split_sizes if input_split_sizes is not None else [n] * W
        self.s.slots[self.rank] = (inp, ins)
        self.s.barrier.wait()
        at = 0
        for src in range(W):
            sinp, sins = self.s.slots[src]
            a = sum(sins[:self.rank])
            cnt = sins[self.rank]
            if output_split_sizes is not None:
                assert output_split_sizes[src] == cnt, (output_split_sizes, sins)
            out[at:at + cnt].copy_(sinp[a:a + cnt])
            at += cnt
        torch.cuda.synchronize()
        self.s.barrier.wait()


@pytest.mark.parametrize("dense", ["0", "1"])
@pytest.mark.parametrize("k,chunks,flags", [(21, 1, 0), (21, 3, 0), (31, 2, sa.FLAG_FORCE_DIRECT), (15, 0, 0)])
def test_two_contexts_merge_like_two_ranks(orc, monkeypatch, k, chunks, flags, dense):
    monkeypatch.setenv("SHK_DIST_DENSE", dense)  # "0": occupied entries only; "1": whole page ranges
    world, n_reads, histo_max = 2, 24_500, 300
    spec = sa.SynthSpec(genome_len=40_000, sub_per_64k=328, n_per_64k=66)
    bases, offsets = sa.synth_reads(spec, 0, n_reads)
    ref = orc.run_batch(bases, offsets, k, chunks, histo_max)
    shared = ThreadGroup.Shared(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            # different capacity hints: the ranks must first agree on a geometry
            eng = sa.KmerEngine(k, chunks, histo_max, capacity_hint=20_000 if rank else 0, flags=flags)
            for first, n in shard_batches(n_reads, rank, world):
                eng.set_read_index(first)
                eng.ingest_reads(bases, offsets[first:first + n + 1])
            dc = DistCounter(eng, ThreadGroup(shared, rank), device=0)
            results[rank] = (dc.finalize_histograms(), dc.totals)
            eng.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    st = ref.stats
    for hist, tot in results:
        assert np.array_equal(hist, ref.histograms())
        assert tot["n_kmers_ingested"] == st["n_kmers_ingested"]
        assert tot["n_unique_kmers"] == st["n_unique_kmers"]
        assert tot["n_reads_ingested"] == st["n_reads_ingested"]
        assert tot["n_bases_ingested"] == st["n_bases_ingested"]


@pytest.mark.parametrize("cap", [None, 1024, 1 << 20])
def test_fixed_capacity_pieces_from_the_second_job_on(orc, monkeypatch, cap):
    """The first finalize asks every peer for its entry counts; from then on pieces have a fixed capacity learnt
    from the job before (no host read-back in front of the all-to-all).  cap=1024 pins a capacity that is far too
    small: nobody may merge anything then, the exchange is repeated with exact counts, and the histogram must still
    be the oracle's.  cap=2^20 pins a capacity that is ample, on tables that start small: the exchange is queued
    behind a counting launch nobody has looked at, that launch spills records (the table has to grow), and the
    senders' poisoned headers must stop every merge just the same."""
    monkeypatch.setenv("SHK_DIST_DENSE", "0")
    if cap is not None:
        monkeypatch.setenv("SHK_DIST_CAP", str(cap))
    world, n_reads, histo_max, k, chunks = 2, 24_500, 300, 21, 2
    jobs = []
    big = cap == 1 << 20  # (nearly every k-mer distinct: the 2^20-slot table a context starts with cannot hold a shard)
    for seed, glen in ((0, 4_000_000), (5, 5_000_000)) if big else ((0, 40_000), (5, 55_000)):  # (job 2 has MORE distinct k-mers)
        spec = sa.SynthSpec(genome_len=glen, sub_per_64k=328, n_per_64k=66)
        bases, offsets = sa.synth_reads(spec, seed, n_reads)
        jobs.append((bases, offsets, orc.run_batch(bases, offsets, k, chunks, histo_max)))
    shared = ThreadGroup.Shared(world)
    results, errors = [[] for _ in range(world)], []

    def run(rank):
        try:
            eng = sa.KmerEngine(k, chunks, histo_max)
            dc = DistCounter(eng, ThreadGroup(shared, rank), device=0)
            keep = []
            for bases, offsets, _ in jobs + jobs[:1]:
                eng.reset()
                if big:  # one launch per rank, nobody looks at its outcome before the exchange is queued
                    first, n = rank * (n_reads // 2), n_reads // 2
                    lo, hi = int(offsets[first]), int(offsets[first + n])
                    d_b = torch.from_numpy(bases[lo:hi].copy()).cuda()
                    d_o = torch.from_numpy((offsets[first:first + n + 1] - offsets[first]).astype(np.int64)).cuda()
                    torch.cuda.synchronize()
                    eng.set_read_index(first)
                    eng.ingest_reads_device(d_b.data_ptr(), d_o.data_ptr(), n, hi - lo)
                    keep.append((d_b, d_o))
                else:
                    for first, n in shard_batches(n_reads, rank, world):
                        eng.set_read_index(first)
                        eng.ingest_reads(bases, offsets[first:first + n + 1])
                results[rank].append((dc.finalize_histograms(), dict(dc.totals)))
            results[rank].append((dc.n_fixed_exchanges, dc.n_redone_exchanges, eng.counters()["n_grows"]))
            eng.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for res in results:
        for (hist, tot), (_, _, ref) in zip(res[:3], jobs + jobs[:1]):
            assert np.array_equal(hist, ref.histograms())
            assert tot["n_unique_kmers"] == ref.stats["n_unique_kmers"]
            assert tot["n_kmers_ingested"] == ref.stats["n_kmers_ingested"]
            # the per-lane base counters live OUTSIDE the block the ranks reduce in place: a repeated exchange
            # (cap=1024: every job; cap=2^20: after the poisoned one) must not sum a sum
            assert tot["n_bases_ingested"] == ref.stats["n_bases_ingested"]
            assert tot["n_reads_ingested"] == ref.stats["n_reads_ingested"] == n_reads
            assert tot["n_bases_read"] == ref.stats["n_bases_read"]
        n_fixed, n_rem, n_grows = res[3]
        if cap == 1 << 20:
            assert n_fixed >= 4 and n_rem >= 1 and n_grows >= 1  # (a poisoned exchange is repeated with fixed pieces)
        elif cap is None:
            assert n_fixed == 2 and n_rem <= 1  # (job 2 may outgrow the capacity learnt from job 1 — by design)
        else:
            assert n_fixed == 3 and n_rem == 3


# ---- a randomized sweep of merge-at-finalize: what bench.py --gpus N runs, rank for rank -----------------------------------

@pytest.mark.parametrize("seed", range(int(os.environ.get("SHK_FUZZ_SEEDS", "16"))))
def test_random_merge_at_finalize_jobs_against_the_oracle(orc, monkeypatch, seed):
    """World size 2 … 8, k, chunk lanes, flags, the dense / packed piece format, and THREE jobs in a row on the same
    contexts (the first asks for exact counts, the others take fixed-capacity pieces learnt from the job before — job
    sizes drawn so that a later job may outgrow them), reads dealt to the ranks in 1000-read batches: every rank's
    histogram and totals of every job against the oracle."""
    rng = np.random.default_rng(55_000 + seed)
    world = int(rng.choice([2, 2, 4, 8]))
    k = int(rng.choice([11, 15, 19, 21, 21, 25, 31]))
    chunks = int(rng.choice([0, 1, 2, 3, 10]))
    flags = int(rng.choice([0, 0, sa.FLAG_FORCE_DIRECT, sa.FLAG_FORCE_PAGED]))
    histo_max = int(rng.choice([5, 300]))
    monkeypatch.setenv("SHK_DIST_DENSE", str(rng.choice(["0", "0", "1"])))
    hints = [int(rng.choice([0, 0, 30_000, 400_000])) for _ in range(world)]   # (the ranks must agree on a geometry first)
    jobs = []
    for j in range(3):
        n_reads = int(rng.integers(world * 1000, 16_000))
        spec = sa.SynthSpec(genome_len=int(rng.choice([8_000, 60_000, 900_000])), sub_per_64k=int(rng.choice([0, 300])),
                            n_per_64k=int(rng.choice([0, 60])))
        bases, offsets = sa.synth_reads(spec, int(rng.integers(0, 50_000)), n_reads)
        jobs.append((bases, offsets, n_reads, orc.run_batch(bases, offsets, k, chunks, histo_max)))
    shared = ThreadGroup.Shared(world)
    results, errors = [[] for _ in range(world)], []

    def run(rank):
        try:
            eng = sa.KmerEngine(k, chunks, histo_max, capacity_hint=hints[rank], flags=flags)
            dc = DistCounter(eng, ThreadGroup(shared, rank), device=0)
            for bases, offsets, n_reads, _ in jobs:
                eng.reset()
                for first, n in shard_batches(n_reads, rank, world):
                    eng.set_read_index(first)
                    eng.ingest_reads(bases, offsets[first:first + n + 1])
                results[rank].append((dc.finalize_histograms(), dict(dc.totals)))
            eng.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, (errors, dict(seed=seed, world=world, k=k, chunks=chunks, flags=flags))
    for res in results:
        for (hist, tot), (_, _, n_reads, ref) in zip(res, jobs):
            assert np.array_equal(hist, ref.histograms()), dict(seed=seed, world=world, k=k, chunks=chunks, flags=flags)
            for f in ("n_unique_kmers", "n_kmers_ingested", "n_bases_ingested", "n_reads_ingested", "n_bases_read"):
                assert tot[f] == ref.stats[f], (f, seed)
