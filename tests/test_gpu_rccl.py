"""The multi-process exchange path over a REAL RCCL communicator — a world of one, which is all a one-GPU box allows:
`OwnerCounter` (key-space-partitioned ingest, what BASELINE configs[3]/[4] run per rank) with its collectives queued
on the engine's own HIP stream through torch.cuda.ExternalStream — all_to_all_single on the uint32 record and cursor
tensors, the 2-word all_reduce(MAX), the histogram all_reduce(SUM) — against the oracle, bit for bit.  The engine is
created with n_owners = 1: a share that is the whole key space, so the scatter → exchange → absorb rounds are the
W-rank job's own code (shk_xchg_*), one segment wide.  `DistCounter` (tables merged at finalize) over the same
communicator rides along.  What a world of one cannot show — peer copies over xGMI, N ranks — is the driver's 8-GPU
run (bench.py --config 4 / 5)."""
import os

import numpy as np
import pytest
import torch

import sharkmer_amd as sa
from sharkmer_amd.dist import DistCounter, OwnerCounter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl_world_of_one():
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("k,chunks,lvl1,hint,round_reads,max_msg", [(21, 1, 10, 4_200_000, 5_000, 0), (21, 10, 10, 4_200_000, 2_300, 0),
                                                                    (17, 3, 6, 600_000, 1_000, 0), (19, 0, 8, 1_000_000, 7_777, 0),
                                                                    (21, 3, 10, 4_200_000, 5_000, 65536),
                                                                    (31, 3, 10, 4_200_000, 5_000, 0), (27, 10, 10, 4_200_000, 2_300, 65536),
                                                                    (31, 3, 10, 4_200_000, 5_000, -1), (27, 10, 10, 4_200_000, 2_300, -65536)])
def test_owner_counter_over_rccl_world_of_one(orc, rccl_world_of_one, monkeypatch, k, chunks, lvl1, hint, round_reads, max_msg):
    """max_msg: the message limit pinned below a segment's size — the segment then never goes through the communicator
    in one piece (RCCL was measured to deliver only half of a message above 2^30 bytes, tools/rccl_a2a_probe.py), and the
    rank's own segment is absorbed where the scatter left it.  k > 21: the same segments with 8-byte records (round 4);
    max_msg < 0: the wide round instead (SHK_XL64=0: whole k-mers, unequal parts), |max_msg| > 1 the limit."""
    monkeypatch.setenv("SHK_LEVEL1_LOG", str(lvl1))
    wide = max_msg < 0
    if wide:
        monkeypatch.setenv("SHK_XL64", "0")
        max_msg = -max_msg if max_msg < -1 else 0
    if max_msg:
        import sharkmer_amd.dist as sd
        monkeypatch.setattr(sd, "MAX_MESSAGE_BYTES", max_msg)
    spec = sa.SynthSpec(genome_len=90_000, sub_per_64k=250, n_per_64k=50)
    n_reads = 20_500
    bases, offsets = sa.synth_reads(spec, 0, n_reads)
    ref = orc.run_batch(bases, offsets, k, chunks, 300)
    d_bases = torch.from_numpy(bases.copy()).cuda()
    with sa.KmerEngine(k, chunks, 300, capacity_hint=hint, n_owners=1, owner_id=0) as eng:
        oc = OwnerCounter(eng, rccl_world_of_one, device=0, round_bases=round_reads * 160)
        keep = []
        for first in range(0, n_reads, round_reads):
            n = min(round_reads, n_reads - first)
            o0, o1 = int(offsets[first]), int(offsets[first + n])
            offs = torch.from_numpy((offsets[first:first + n + 1] - offsets[first]).astype(np.int64)).cuda()
            keep.append(offs)
            lay = oc.round((d_bases[o0:o1].data_ptr(), offs.data_ptr(), n, o1 - o0, first))
            if not wide:
                assert lay.n_owners == 1 and lay.n_lanes == max(chunks, 1) and (lay.record_bytes or 4) == (4 if k <= 21 else 8)
            else:   # the wide round: no layout to report
                assert lay is None and not eng.xchg_feasible()
        oc.round(None)   # a rank that has run out of reads still takes part
        hist = oc.finalize_histograms()
        tot = oc.totals
        gk, gc = eng.export_table()
    assert np.array_equal(hist, ref.histograms())
    for f in ("n_kmers_ingested", "n_unique_kmers", "n_reads_ingested", "n_bases_ingested", "n_hashed_kmers"):
        assert tot[f] == ref.stats[f], f
    rk, rc = ref.merged().export()
    order = np.argsort(gk, kind="stable")
    assert np.array_equal(gk[order], rk) and np.array_equal(gc[order], rc)
    assert oc.n_rounds == -(-n_reads // round_reads) + 1


def test_owner_counter_invalid_byte_over_rccl(rccl_world_of_one, monkeypatch):
    monkeypatch.setenv("SHK_LEVEL1_LOG", "8")
    bases, offsets = sa.synth_reads(sa.SynthSpec(genome_len=50_000), 0, 3_000)
    bad = bases.copy()
    bad[int(offsets[1_500]) + 7] = ord("x")
    d_bases = torch.from_numpy(bad).cuda()
    offs = torch.from_numpy(offsets.astype(np.int64)).cuda()
    with sa.KmerEngine(19, 1, 100, capacity_hint=1_000_000, n_owners=1) as eng:
        oc = OwnerCounter(eng, rccl_world_of_one, device=0, round_bases=3_000 * 160)
        with pytest.raises(sa.ShkError, match="Invalid character 'x' in sequence. Only ACGTN allowed."):
            oc.round((d_bases.data_ptr(), offs.data_ptr(), 3_000, int(offsets[-1]), 0))


@pytest.mark.parametrize("chunks", [1, 4])
def test_dist_counter_over_rccl_world_of_one(orc, rccl_world_of_one, chunks):
    """Merge-at-finalize over the same communicator: three jobs, so the second and third take the fixed-capacity
    pieces and the in-place all-reduce of the control block's sum region; read and base totals included (the live
    per-lane counters lie outside the reduced block)."""
    k, histo_max, n_reads = 21, 200, 30_000
    spec = sa.SynthSpec(genome_len=120_000, sub_per_64k=300, n_per_64k=60)
    with sa.KmerEngine(k, chunks, histo_max) as eng:
        dc = DistCounter(eng, rccl_world_of_one, device=0)
        for job in range(3):
            bases, offsets = sa.synth_reads(spec, job * n_reads, n_reads)
            ref = orc.run_batch(bases, offsets, k, chunks, histo_max)
            eng.reset()
            eng.ingest_reads(bases, offsets)
            hist = dc.finalize_histograms()
            assert np.array_equal(hist, ref.histograms())
            for f in ("n_kmers_ingested", "n_unique_kmers", "n_reads_ingested", "n_bases_ingested", "n_bases_read"):
                assert dc.totals[f] == ref.stats[f], (job, f)
