"""GPU tests of the packed input path (the reference's Read::from_str layout as the
device-buffer spec, src/kmer/encoding.rs:60-95; vectors src/kmer/mod.rs:61-156): the device-side packer is
asked for the stream of the reference's vectors, pack ∘ unpack is the identity on ragged / N input, and a
packed batch counts exactly like its ASCII twin (and like the oracle)."""
import numpy as np
import pytest
import torch

import sharkmer_amd as sa

from test_gpu_parity import ragged_reads
from test_packed_cpu import reads_from_packed, _TWO

pytestmark = pytest.mark.gpu


def _device_pack(eng, seq: bytes):
    n = len(seq)
    d_b = torch.from_numpy(np.frombuffer(seq, dtype=np.uint8).copy()).cuda() if n else torch.empty(0, dtype=torch.uint8, device="cuda")
    d_p = torch.zeros((n + 3) // 4 + 8, dtype=torch.uint8, device="cuda")
    d_m = torch.zeros((n + 31) // 32 + 2, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()  # (torch's copies and fills run on torch's stream, the engine's kernels on its own)
    eng.pack_reads_device(d_b.data_ptr(), n, d_p.data_ptr(), d_m.data_ptr())
    eng.sync()
    return sa.PackedReads(d_p.cpu().numpy()[:(n + 3) // 4], d_m.cpu().numpy().view(np.uint32)[:(n + 31) // 32],
                          np.array([0, n], dtype=np.uint64), n), d_p, d_m


@pytest.mark.parametrize("seq,packed", [
    ("CGTAATGCGGCGA", [0b01101100, 0b00111001, 0b10100110, 0b00000000]),   # kmer/mod.rs:61-111
    ("C", [0b01000000]),
    ("CGTAATGCGGCG", [0b01101100, 0b00111001, 0b10100110]),
])
def test_device_packer_gives_the_reference_vectors(seq, packed):
    with sa.KmerEngine(5, 1, 10) as eng:
        pk, _, _ = _device_pack(eng, seq.encode())
    assert list(pk.packed) == packed and not pk.nmask.any()


@pytest.mark.parametrize("seq,expected", [
    ("NCGTAATGCGGCG", [(bytes([0b01101100, 0b00111001, 0b10100110]), 12)]),   # kmer/mod.rs:113-156
    ("CGTANATGCGGCGA", _TWO), ("NCGTANATGCGGCGA", _TWO), ("NCGTANATGCGGCGANN", _TWO), ("NNCGTANATGCGGCGA", _TWO),
])
def test_device_packer_n_vectors(seq, expected):
    with sa.KmerEngine(5, 1, 10) as eng:
        pk, _, _ = _device_pack(eng, seq.encode())
    assert reads_from_packed(pk) == expected


def test_device_pack_equals_host_pack_and_unpack_inverts_it():
    rng = np.random.default_rng(8)
    bases, offsets = ragged_reads(rng, 3_000, max_len=300, p_n=0.03)
    host = sa.pack_reads(bases, offsets)
    with sa.KmerEngine(21, 1, 10) as eng:
        dev, d_p, d_m = _device_pack(eng, bases.tobytes())
        assert np.array_equal(dev.packed, host.packed) and np.array_equal(dev.nmask, host.nmask)
        d_out = torch.zeros(len(bases) + 16, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # (torch's fill runs on torch's stream, the engine's kernel on its own)
        eng.unpack_reads_device(d_p.data_ptr(), d_m.data_ptr(), len(bases), d_out.data_ptr())
        eng.sync()
        assert np.array_equal(d_out.cpu().numpy()[:len(bases)], bases)
        with pytest.raises(sa.ShkError, match="Invalid character 'Q' in sequence. Only ACGTN allowed."):
            _device_pack(eng, b"ACGTNNACGQACGT")
        eng.ingest_seq("ACGTACGTACGTACGTACGTACGTT")  # the context is still usable
        eng.finalize()


@pytest.mark.parametrize("k,chunks,slice_kb", [(21, 3, 0), (31, 2, 16), (15, 0, 7), (21, 10, 64)])
def test_packed_ingest_counts_like_ascii(orc, monkeypatch, k, chunks, slice_kb):
    """Host buffers: the packed batch crosses PCIe in slices that start at read boundaries (any bit offset
    into the streams) and is unpacked in HBM; split calls keep the read index."""
    if slice_kb:
        monkeypatch.setenv("SHK_SLICE_KB", str(slice_kb))
    rng = np.random.default_rng(k)
    bases, offsets = ragged_reads(rng, 7_000, max_len=260, p_n=0.02)
    ref = orc.run_batch(bases, offsets, k, chunks, 100)
    pk = sa.pack_reads(bases, offsets, pinned=True)
    with sa.KmerEngine(k, chunks, 100) as eng:
        cut = 2_345
        first = sa.PackedReads(pk.packed, pk.nmask, pk.offsets[:cut + 1], int(pk.offsets[cut]))
        rest = sa.PackedReads(pk.packed, pk.nmask, pk.offsets[cut:], pk.n_bases)   # offsets[0] > 0: mid-stream
        eng.ingest_packed(first)
        eng.ingest_packed(rest)
        eng.finalize()
        assert np.array_equal(eng.histograms(), ref.histograms())
        c = eng.counters()
        gk, gc = eng.export_table()
    for f in ("n_reads_ingested", "n_bases_read", "n_bases_ingested", "n_kmers_ingested", "n_unique_kmers"):
        assert c[f] == ref.stats[f], f
    rk, rc = ref.merged().export()
    assert np.array_equal(gk, rk) and np.array_equal(gc, rc)


def test_packed_device_resident_ingest(orc):
    spec = sa.SynthSpec(genome_len=90_000, sub_per_64k=300, n_per_64k=80)
    bases, offsets = sa.synth_reads(spec, 0, 30_000)
    ref = orc.run_batch(bases, offsets, 21, 4, 200)
    pk = sa.pack_reads(bases, offsets)
    with sa.KmerEngine(21, 4, 200) as eng:
        d_p = torch.from_numpy(np.concatenate([pk.packed, np.zeros(16, dtype=np.uint8)])).cuda()
        d_m = torch.from_numpy(np.concatenate([pk.nmask, np.zeros(2, dtype=np.uint32)]).view(np.int32)).cuda()
        d_o = torch.from_numpy(offsets.astype(np.int64)).cuda()
        eng.ingest_packed_device(d_p.data_ptr(), d_m.data_ptr(), d_o.data_ptr(), len(offsets) - 1, len(bases))
        eng.finalize()
        assert np.array_equal(eng.histograms(), ref.histograms())
        assert eng.counters()["n_bases_ingested"] == ref.stats["n_bases_ingested"]


@pytest.mark.parametrize("n_per_64k", [0, 6, 30000])
def test_packed_ingest_sparse_n_mask(orc, n_per_64k):
    """Slices of ≥ 2 M bases send an N mask that is nearly all zeros as the list of its non-zero words (none at
    all, a few, or — every other base an N — the mask as it is): the counts must not care."""
    spec = sa.SynthSpec(genome_len=400_000, sub_per_64k=100, n_per_64k=n_per_64k)
    bases, offsets = sa.synth_reads(spec, 0, 40_000)  # 6 M bases: one slice of 1.5 M (a mask as it is), one of 4.5 M
    ref = orc.run_batch(bases, offsets, 21, 2, 300)
    pk = sa.pack_reads(bases, offsets, pinned=True)
    with sa.KmerEngine(21, 2, 300) as eng:
        eng.ingest_packed(pk)
        eng.finalize()
        assert np.array_equal(eng.histograms(), ref.histograms())
        c = eng.counters()
    for f in ("n_reads_ingested", "n_bases_read", "n_bases_ingested", "n_kmers_ingested", "n_unique_kmers"):
        assert c[f] == ref.stats[f], f
