"""GPU tests of the MULTI-DEVICE context (shk_config.n_devices / device_ids): ONE context
over several devices behind the same C ABI — host batches dealt to per-device owner shares with their global
read index (io.rs:340-361), records exchanged by owner every round, histograms summed at finalize.  The box
has one card, so the device ids repeat (several shares on card 0): every entry point, thread and copy of the
multi-device path runs, bit-exact against the oracle and against the one-device path."""
import json
import os
import subprocess

import numpy as np
import pytest

import sharkmer_amd as sa

from test_gpu_parity import pack, ragged_reads

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "sharkmer_amd", "csrc", "shk_count")
FILES = [os.path.join(G, "reads_main.fastq.gz"), os.path.join(G, "reads_part2.fastq")]


def _check(orc, bases, offsets, k, chunks, histo_max, devs, splits=None, hint=0):
    ref = orc.run_batch(bases, offsets, k, chunks, histo_max)
    with sa.KmerEngine(k, chunks, histo_max, capacity_hint=hint, device_ids=devs) as eng:
        n = len(offsets) - 1
        cuts = [0] + sorted(splits or []) + [n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            eng.ingest_reads(bases, offsets[a:b + 1])
        eng.finalize()
        got = eng.histograms()
        c = eng.counters()
        gk, gc = eng.export_table()
        rk, rc = ref.merged().export()
        probe = np.concatenate([rk[::97], np.array([1, 2, 3], dtype=np.uint64)])
        look = eng.lookup(probe)
    assert np.array_equal(got, ref.histograms())
    for f in ("n_reads_ingested", "n_bases_read", "n_bases_ingested", "n_kmers_ingested", "n_unique_kmers", "n_hashed_kmers"):
        assert c[f] == ref.stats[f], f
    if chunks:
        assert c["n_singleton_kmers"] == ref.stats["n_singleton_kmers"]
    order = np.argsort(gk, kind="stable")
    assert np.array_equal(gk[order], rk) and np.array_equal(gc[order], rc)
    merged = ref.merged()
    assert [int(x) for x in look] == [merged.get_count(int(x)) for x in probe]
    return c


@pytest.mark.parametrize("devs,k,chunks", [([0, 0], 21, 10), ([0, 0, 0, 0], 19, 3), ([0, 0], 15, 0), ([0] * 8, 21, 1),
                                           ([0, 0], 31, 10), ([0, 0, 0, 0], 27, 3), ([0, 0], 21, 40), ([0, 0, 0, 0], 31, 40), ([0, 0], 23, 0)])
def test_multi_device_context_matches_oracle(orc, devs, k, chunks):
    """k > 21 (no 4-byte exchange record holds the rest of the key at the default fan-out): the context takes the WIDE
    round — whole k-mers grouped by owner, pulled and inserted by their owners — instead of refusing (round 4); 40 chunk
    lanes: the 4-byte rounds (the paged passes take up to 128 lanes since round 4).  The reference accepts 0 < k < 32 and
    any number of chunks (cli.rs:659-677)."""
    spec = sa.SynthSpec(genome_len=70_000, sub_per_64k=250, n_per_64k=50)
    bases, offsets = sa.synth_reads(spec, 0, 23_456)
    _check(orc, bases, offsets, k, chunks, 300, devs, splits=[1_700, 9_999, 17_000])


def test_multi_device_ragged_reads_and_small_rounds(orc, monkeypatch):
    """Empty reads, reads shorter than k, N runs; rounds of 64 KiB per device (many rounds per call)."""
    monkeypatch.setenv("SHK_GROUP_ROUND_KB", "64")
    rng = np.random.default_rng(11)
    bases, offsets = ragged_reads(rng, 9_000, max_len=260, p_n=0.02)
    _check(orc, bases, offsets, 17, 4, 50, [0, 0, 0, 0], splits=[4_321])


def test_multi_device_skewed_input(orc, monkeypatch):
    """Low-complexity reads overflow their level-1 region on the sending device: the foreign spill lists are
    pulled and inserted by the owners."""
    monkeypatch.setenv("SHK_LEVEL1_LOG", "8")
    rng = np.random.default_rng(4)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [b"AC" * 75 if i % 2 else lut[rng.integers(0, 4, size=150)].tobytes() for i in range(8_000)]
    bases, offsets = pack(seqs)
    c = _check(orc, bases, offsets, 19, 2, 5000, [0, 0], hint=1_000_000)
    assert c["n_spilled"] >= 0


def test_multi_device_explicit_chunks_insert_reset(orc):
    """drain_batch with an explicit chunk (io.rs:356-358), KmerCounts::insert (counting.rs:152-154), reset."""
    spec = sa.SynthSpec(genome_len=30_000, sub_per_64k=100)
    bases, offsets = sa.synth_reads(spec, 0, 6_000)
    with sa.KmerEngine(19, 3, 100, device_ids=[0, 0]) as eng:
        for rep in range(2):
            ref = orc.Run(19, 3, 100)
            # the oracle stripes 1000-read batches 0,1,2,0,…: hand the same batches over with explicit chunk ids
            for b in range(6):
                eng.ingest_batch(b % 3, bases[int(offsets[b * 1000]):int(offsets[(b + 1) * 1000])],
                                 offsets[b * 1000:(b + 1) * 1000 + 1] - offsets[b * 1000])
            ref.push_batch(bases, offsets)
            ref.finish()
            eng.finalize()
            assert np.array_equal(eng.histograms(), ref.histograms())
            eng.reset()
        eng.insert([5, 6, 7], [2, 0xFFFFFFFF, 9], chunk_id=1)
        eng.insert([6], [3], chunk_id=1)
        assert [int(x) for x in eng.lookup([5, 6, 7, 8])] == [2, 0xFFFFFFFF, 9, 0]


def test_multi_device_errors_carry_the_reference_messages():
    with sa.KmerEngine(21, 1, 100, device_ids=[0, 0]) as eng:
        with pytest.raises(sa.ShkError, match="No reads were ingested"):
            eng.finalize()
        with pytest.raises(sa.ShkError, match="Invalid character 'x' in sequence. Only ACGTN allowed."):
            eng.ingest_seqs(["ACGTACGTACGTACGTACGTACGTAAAA", "ACGTACGTACGTxACGTACGTACGTACGTACGT"])
    with sa.KmerEngine(31, 1, 100, device_ids=[0, 0]) as eng:   # (refused until round 4: k > 21 takes the wide round now)
        with pytest.raises(sa.ShkError, match="Invalid character 'x' in sequence. Only ACGTN allowed."):
            eng.ingest_seqs(["ACGTACGTACGTACGTACGTACGTAAAAACGTACGTACGT", "ACGTACGTACGTxACGTACGTACGTACGTACGTACGTACGTACGT"])
    with pytest.raises(sa.ShkError, match="power of two"):
        sa.KmerEngine(21, 1, 100, device_ids=[0, 0, 0])


def test_cli_over_two_device_shares_is_byte_exact(tmp_path):
    """shk_count --devices 0,0: the golden .histo / .final.histo byte for byte, the same stats."""
    r = subprocess.run([EXE, "-k", "21", "-s", "g", "-o", str(tmp_path), "--chunks", "3", "--histo-max", "50",
                        "--devices", "0,0"] + FILES, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "g.histo").read_bytes() == open(os.path.join(G, "golden_k21_c3.histo"), "rb").read()
    assert (tmp_path / "g.final.histo").read_bytes() == open(os.path.join(G, "golden_k21_c3.final.histo"), "rb").read()
    want = json.load(open(os.path.join(G, "golden_k21_c3.stats.json")))
    import yaml
    d = yaml.safe_load((tmp_path / "g.stats.yaml").read_text())
    assert d["n_kmers"] == want["n_kmers_ingested"] and d["n_subreads_ingested"] == want["n_reads_ingested"]
    assert d["n_bases_ingested"] == want["n_bases_ingested"] and d["n_singleton_kmers"] == want["n_singleton_kmers"]


def test_one_device_id_is_todays_path_byte_for_byte(tmp_path):
    """n_devices = 1 must be exactly the single-device run (VERDICT r1 #4a)."""
    a, b = tmp_path / "a", tmp_path / "b"
    sa.run_files(FILES, k=21, chunks=3, sample="s", outdir=str(a), histo_max=50)
    sa.run_files(FILES, k=21, chunks=3, sample="s", outdir=str(b), histo_max=50, device_ids=[0])
    for f in ("s.histo", "s.final.histo"):
        assert (a / f).read_bytes() == (b / f).read_bytes()
    import yaml
    ya, yb = yaml.safe_load((a / "s.stats.yaml").read_text()), yaml.safe_load((b / "s.stats.yaml").read_text())
    ya.pop("peak_memory_bytes"), yb.pop("peak_memory_bytes")
    assert ya == yb
