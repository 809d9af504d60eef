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
    b[2_000_001] = ord("y")
    b[1_234_567] = ord("x")   # the first offender in input order is the one reported (encoding.rs:353-356)
    with pytest.raises(sa.ShkError, match="Invalid character 'x' in sequence. Only ACGTN allowed."):
        sa.pack_reads(b, o, threads=7)
    with pytest.raises(sa.ShkError, match="Invalid character 'a' in sequence"):
        _pack(b"ACGTacgt")
