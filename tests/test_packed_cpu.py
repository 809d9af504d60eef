"""The 2-bit packed batch format (include/shk.h: shk_pack_reads) against the reference's own vectors for
its packed layout — Read::from_str, src/kmer/encoding.rs:60-95; vectors src/kmer/mod.rs:61-156 — and against
the oracle's restatement of from_str / seq_to_reads on random input.  Host code only: no GPU needed."""
import numpy as np
import pytest

import sharkmer_amd as sa


def _pack(seq: bytes):
    b = np.frombuffer(seq, dtype=np.uint8)
    return sa.pack_reads(b, np.array([0, len(b)], dtype=np.uint64))


def reads_from_packed(pk):
    """seq_to_reads (encoding.rs:284-298) read off the packed stream + N mask: the maximal N-free runs, each
    re-packed on its own like Read::from_str does (4 bases per byte, first base on top, tail left-aligned)."""
    n = pk.n_bases
    codes = np.array([(int(pk.packed[p >> 2]) >> (6 - 2 * (p & 3))) & 3 for p in range(n)], dtype=np.uint8)
    isn = np.array([(int(pk.nmask[p >> 5]) >> (p & 31)) & 1 for p in range(n)], dtype=bool)
    out, run = [], []
    for p in range(n + 1):
        if p < n and not isn[p]:
            run.append(int(codes[p]))
            continue
        if run:
            by = bytearray((len(run) + 3) // 4)
            for i, c in enumerate(run):
                by[i >> 2] |= c << (6 - 2 * (i & 3))
            out.append((bytes(by), len(run)))
            run = []
    return out


# --- kmer/mod.rs:61-111 test_from_str: the stream of an N-free read IS Read::from_str's bytes ---------------
@pytest.mark.parametrize("seq,packed,length", [
    ("CGTAATGCGGCGA", [0b01101100, 0b00111001, 0b10100110, 0b00000000], 13),
    ("C", [0b01000000], 1),
    ("CGTAATGCGGCG", [0b01101100, 0b00111001, 0b10100110], 12),
    ("", [], 0),
])
def test_pack_matches_from_str_vectors(seq, packed, length):
    pk = _pack(seq.encode())
    assert list(pk.packed) == packed and pk.n_bases == length
    assert not pk.nmask.any()


# --- kmer/mod.rs:113-156 test_seq_to_reads_n: N splits; the mask carries where ------------------------------
_TWO = [(bytes([0b01101100]), 4), (bytes([0b00111001, 0b10100110, 0b00000000]), 9)]


@pytest.mark.parametrize("seq,expected", [
    ("NCGTAATGCGGCG", [(bytes([0b01101100, 0b00111001, 0b10100110]), 12)]),
    ("CGTANATGCGGCGA", _TWO),
    ("NCGTANATGCGGCGA", _TWO),
    ("NCGTANATGCGGCGANN", _TWO),
    ("NNCGTANATGCGGCGA", _TWO),
])
def test_pack_n_vectors(seq, expected):
    pk = _pack(seq.encode())
    assert reads_from_packed(pk) == expected
    assert [p for p in range(len(seq)) if (int(pk.nmask[p >> 5]) >> (p & 31)) & 1] == [i for i, ch in enumerate(seq) if ch == "N"]


def test_packed_sizes_are_what_the_packer_writes():
    """shk_packed_sizes: (n + 3) / 4 bytes of stream, (n + 31) / 32 words of N mask — the arrays pack_reads allocates."""
    import ctypes as C
    from sharkmer_amd.engine import load_front_library
    L = load_front_library()
    for n in (0, 1, 3, 4, 5, 31, 32, 33, 1000, 12345, (1 << 32) + 7):
        pb, nw = C.c_uint64(99), C.c_uint64(99)
        L.shk_packed_sizes(n, C.byref(pb), C.byref(nw))
        assert (pb.value, nw.value) == ((n + 3) // 4, (n + 31) // 32), n
    pk = _pack(b"ACGTN" * 7)
    assert pk.packed.size == (35 + 3) // 4 and pk.nmask.size == (35 + 31) // 32


def test_pack_matches_oracle_on_random_reads(orc):
    rng = np.random.default_rng(2)
    for _ in range(40):
        n = int(rng.integers(1, 400))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=n, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        pk = _pack(seq)
        assert reads_from_packed(pk) == orc.seq_to_reads(seq.decode())
        if b"N" not in seq:
            got, ln = orc.read_from_str(seq.decode())
            assert bytes(pk.packed) == bytes(got) and ln == n


def test_pack_threads_agree_and_first_invalid_byte_wins():
    rng = np.random.default_rng(3)
    n = 3_000_000
    b = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=n)].copy()
    o = np.array([0, n], dtype=np.uint64)
    one, many = sa.pack_reads(b, o, threads=1), sa.pack_reads(b, o, threads=7)
    assert np.array_equal(one.packed, many.packed) and np.array_equal(one.nmask, many.nmask)
    # out=: the arrays of a batch packed before are written again (a streaming caller pins its buffers once)
    b2 = np.frombuffer(b"ACGTN", dtype=np.uint8)[rng.integers(0, 5, size=n)].copy()
    again = sa.pack_reads(b2, o, threads=5, out=many)
    fresh = sa.pack_reads(b2, o, threads=1)
    assert again is many and np.array_equal(many.packed, fresh.packed) and np.array_equal(many.nmask, fresh.nmask)
    with pytest.raises(AssertionError, match="another batch shape"):
        sa.pack_reads(b2[:1000], np.array([0, 1000], dtype=np.uint64), out=many)
    b[2_000_001] = ord("y")
    b[1_234_567] = ord("x")   # the first offender in input order is the one reported (encoding.rs:353-356)
    with pytest.raises(sa.ShkError, match="Invalid character 'x' in sequence. Only ACGTN allowed."):
        sa.pack_reads(b, o, threads=7)
    with pytest.raises(sa.ShkError, match="Invalid character 'a' in sequence"):
        _pack(b"ACGTacgt")


# ---- the reader's packed batches (shk_fastq_next_batch_packed) -------------------------------------------------------

def _fastq_file(tmp_path, name, seqs, gz=False):
    import zlib
    text = "".join(f"@r{i} x\n{s}\n+\n{'I' * len(s)}\n" for i, s in enumerate(seqs)).encode()
    p = tmp_path / name
    if gz:
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        text = co.compress(text) + co.flush()
    p.write_bytes(text)
    return str(p)


@pytest.mark.parametrize("window_kb,batch,gz", [(0, (1_000_000, 1 << 22), False), (3, (777, 50_000), False), (16, (5_000, 1 << 20), True),
                                                (64, (4_096, 100_000), False)])
def test_reader_packed_batches_equal_packed_ascii_batches(tmp_path, monkeypatch, window_kb, batch, gz):
    """The packed batch is what shk_pack_reads makes of the ASCII batch (the reference's Read::from_str layout,
    encoding.rs:60-95, over the batch's concatenated bases + the N mask): same calls, same batch boundaries, ragged
    lengths incl. empty reads, N runs, windows of a few KiB so that records straddle parse windows and a batch is put
    together from several chunks (a word of 32 bases is then begun by one chunk and finished by the next)."""
    if window_kb:
        monkeypatch.setenv("SHK_FASTQ_WINDOW_KB", str(window_kb))
    rng = np.random.default_rng(window_kb + batch[0])
    seqs = []
    for i in range(9_000):
        L = int(rng.integers(0, 220)) if i % 50 else int(rng.integers(0, 3))
        seqs.append("".join(rng.choice(list("ACGTN"), size=L, p=[0.24, 0.24, 0.24, 0.24, 0.04])))
    p = _fastq_file(tmp_path, "r.fastq" + (".gz" if gz else ""), seqs, gz)
    ra, rp = sa.FastqReader([p]), sa.FastqReader([p])
    n_tot = 0
    while True:
        b, o = ra.next_batch(max_seqs=batch[0], max_bases=batch[1])
        pk = rp.next_batch_packed(max_seqs=batch[0], max_bases=batch[1])
        assert np.array_equal(pk.offsets, o) and pk.n_bases == int(o[-1])
        want = sa.pack_reads(b, o, threads=1)
        assert np.array_equal(pk.packed, want.packed) and np.array_equal(pk.nmask, want.nmask)
        n_tot += len(o) - 1
        if ra.stats()["done"]:
            assert rp.stats()["done"] and rp.stats() == ra.stats()
            break
    assert n_tot == len(seqs)
    ra.close()
    rp.close()


def test_reader_packed_reports_the_first_invalid_byte(tmp_path, monkeypatch):
    """encoding.rs:353-356 on a packed hand-out: the first byte outside ACGTN in read order, lowercase included; and
    only among reads the reference would have drained (io.rs:340-343) — a bad byte behind the last full thousand in
    front of a reading error is never met."""
    monkeypatch.setenv("SHK_FASTQ_WINDOW_KB", "8")
    rng = np.random.default_rng(2)
    seqs = ["".join(rng.choice(list("ACGT"), size=100)) for _ in range(2_500)]
    bad = list(seqs)
    bad[1_800] = bad[1_800][:40] + "x" + bad[1_800][41:]
    bad[700] = bad[700][:99] + "R"
    p = _fastq_file(tmp_path, "bad.fastq", bad)
    r = sa.FastqReader([p])
    with pytest.raises(sa.ShkError, match="Invalid character 'R' in sequence. Only ACGTN allowed."):
        while not r.stats()["done"]:
            r.next_batch_packed(max_seqs=600, max_bases=1 << 20)
    r.close()
    # the file cut inside record 1901: reads 1000..1899 are never drained, so 'x' in read 1800 is never seen
    text = open(p, "rb").read().split(b"\n")
    only_x = list(seqs)
    only_x[1_800] = bad[1_800]
    q = _fastq_file(tmp_path, "cut.fastq", only_x)
    lines = open(q, "rb").read().split(b"\n")
    open(q, "wb").write(b"\n".join(lines[:4 * 1_900 + 2]) + b"\n")
    r = sa.FastqReader([q])
    with pytest.raises(sa.ShkError, match="Truncated FASTQ record at record 1901"):
        while not r.stats()["done"]:
            r.next_batch_packed(max_seqs=600, max_bases=1 << 20)
    assert r.stats()["n_reads_read"] == 1_000
    r.close()


def test_large_hand_outs_take_the_parallel_prefix_sum(tmp_path):
    """Hand-outs of ≥ 65536 reads: the offsets come from a prefix sum by the copy pool, and a batch that the base
    capacity cuts short is cut inside one thread's share — same reads, same order, same batches as the packed path."""
    rng = np.random.default_rng(11)
    lens = rng.integers(0, 40, size=300_000)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [letters[rng.integers(0, 4, size=int(L))].tobytes().decode() for L in lens]
    p = _fastq_file(tmp_path, "many.fastq", seqs)
    ra, rp = sa.FastqReader([p]), sa.FastqReader([p])
    got, n_batches = [], 0
    while not ra.stats()["done"]:
        b, o = ra.next_batch(max_seqs=1_000_000, max_bases=2_100_000)
        pk = rp.next_batch_packed(max_seqs=1_000_000, max_bases=2_100_000)
        assert np.array_equal(pk.offsets, o)
        want = sa.pack_reads(b, o, threads=1)
        assert np.array_equal(pk.packed, want.packed) and np.array_equal(pk.nmask, want.nmask)
        assert int(o[-1]) <= 2_100_000
        got += [b[int(o[i]):int(o[i + 1])].tobytes().decode() for i in range(len(o) - 1)]
        n_batches += 1
    assert got == seqs and n_batches >= 3
    ra.close()
    rp.close()
