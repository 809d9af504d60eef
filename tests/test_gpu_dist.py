"""GPU test of the multi-GPU merge path on ONE card: two KmerEngine contexts play two ranks
and a thread-based stand-in for torch.distributed moves the tensors between them, so the real
HIP entry points (shk_table_reserve_pages, shk_owner_counts, shk_compact_owners, shk_merge_entries,
shk_table_device_ptrs, shk_merge_pages, shk_set_owned_pages, owned-range finalize) are checked against the oracle.  The RCCL transport
itself is exercised by bench.py --gpus N on the 8-GPU node."""
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
        res = stack.sum(0) if op == _ReduceOp.SUM else stack.max(0).values
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
