"""Pins the CPU oracle against every known-answer vector the reference's own
unit tests hold for the hot path.  Values are data
transcribed from /root/reference/src/kmer/mod.rs:61-305 and
src/kmer/counting.rs:365-509; the test bodies are ours."""
import numpy as np
import pytest


# --- kmer/mod.rs:61-111 test_seq_to_reads / test_from_str ---------------------
@pytest.mark.parametrize("seq,packed,length", [
    ("CGTAATGCGGCGA", [0b01101100, 0b00111001, 0b10100110, 0b00000000], 13),
    ("C", [0b01000000], 1),
    ("CGTAATGCGGCG", [0b01101100, 0b00111001, 0b10100110], 12),
    ("", [], 0),
])
def test_from_str(orc, seq, packed, length):
    got, n = orc.read_from_str(seq)
    assert list(got) == packed and n == length
    if seq:
        assert orc.seq_to_reads(seq) == [(bytes(packed), length)]


# --- kmer/mod.rs:113-156 test_seq_to_reads_n -------------------------------------
_TWO = [(bytes([0b01101100]), 4), (bytes([0b00111001, 0b10100110, 0b00000000]), 9)]


@pytest.mark.parametrize("seq,expected", [
    ("NCGTAATGCGGCG", [(bytes([0b01101100, 0b00111001, 0b10100110]), 12)]),
    ("CGTANATGCGGCGA", _TWO),
    ("NCGTANATGCGGCGA", _TWO),
    ("NCGTANATGCGGCGANN", _TWO),
    ("NNCGTANATGCGGCGA", _TWO),
])
def test_seq_to_reads_n(orc, seq, expected):
    assert orc.seq_to_reads(seq) == expected


# --- kmer/mod.rs:33-42, 44-59 read validate / parsing (length bookkeeping) ------------
@pytest.mark.parametrize("seq", [
    "TANCACN", "NTANCACNAGAAAATC",
    "TATTAGCTCATCTANAACAATGAAAAATTGCATTGGCTNTAACTATGGATTTNTTAGAAATTAGTATTNATTTATCATTTTTAATTGGCATTATTNAACTCTTAAGAATAGATNGGAGTTCNCAATTAATTGAAGNTANCACNAGAAAATC",
])
def test_read_parsing_validates(orc, seq):
    for packed, length in orc.seq_to_reads(seq):
        assert len(packed) == (length + 3) // 4


# --- kmer/mod.rs:158-177 test_revcomp_kmer ----------------------------------------------
def test_revcomp_kmer(orc):
    assert orc.revcomp_kmer(0b0010_0110, 3) == 0b0001_1001
    assert orc.revcomp_kmer(orc.revcomp_kmer(0b0010_0110, 3), 3) == 0b0010_0110
    kmer = 0b0110_1100_0011_1001_1010_0110
    assert orc.revcomp_kmer(kmer, 12) == 0b0110_0101_1001_0011_1100_0110
    assert orc.revcomp_kmer(orc.revcomp_kmer(kmer, 12), 12) == kmer


# --- kmer/mod.rs:179-226 test_get_kmers ----------------------------------------------------
_INTS = bytes([0b01101100, 0b00111001, 0b10100110])
_K9 = [0b01_1001_0011_1100_0110, 0b01_0110_0100_1111_0001,
       0b10_0101_1001_0011_1100, 0b00_0011_1001_1010_0110]


@pytest.mark.parametrize("length,expected", [(12, _K9), (11, _K9[:3]), (10, _K9[:2]), (9, _K9[:1])])
def test_get_kmers(orc, length, expected):
    assert orc.read_get_kmers(_INTS, length, 9) == expected


def test_get_kmers_short(orc):
    assert orc.read_get_kmers(bytes([0b01101100, 0b00111001]), 8, 9) == []


def test_hot_path_matches_test_get_kmers_vector(orc):
    assert orc.kmers_from_ascii("CGTAATGCGGCG", 9) == _K9


# --- kmer/mod.rs:228-237 test_kmer_to_seq ---------------------------------------------------
def test_kmer_to_seq(orc):
    assert orc.kmer_to_seq(0b1001_1000, 4) == "GCGA"
    assert orc.kmer_to_seq(0b1001_1000_1001_1000, 8) == "GCGAGCGA"
    assert orc.seq_to_kmer("GCGAGCGA") == 0b1001_1000_1001_1000


# --- kmer/mod.rs:249-270 test_kmers_from_ascii_matches_read_pipeline -----------------------
_CASES = ["CGTAATGCGGCGA", "CGTANATGCGGCGA", "NCGTANATGCGGCGA", "NCGTANATGCGGCGANN",
          "NNCGTANATGCGGCGA", "TANCACN", "NTANCACNAGAAAATC", "AAAA", "ACGTACGTACGT"]


@pytest.mark.parametrize("k", [3, 5, 9, 11])
@pytest.mark.parametrize("seq", _CASES)
def test_kmers_from_ascii_matches_read_pipeline(orc, seq, k):
    expected = []
    for packed, length in orc.seq_to_reads(seq):
        expected += orc.read_get_kmers(packed, length, k)
    assert orc.kmers_from_ascii(seq, k) == expected
    # and the numpy windowing oracle agrees (as a multiset and in order)
    b = np.frombuffer(seq.encode(), dtype=np.uint8)
    assert list(orc.canonical_kmers_numpy(b, np.array([0, len(b)]), k)) == expected


# --- kmer/mod.rs:272-278 -----------------------------------------------------------------------
def test_kmers_from_ascii_short_sequences(orc):
    assert orc.kmers_from_ascii("ACGT", 9) == []
    assert len(orc.kmers_from_ascii("ACGTACGTA", 9)) == 1


# --- kmer/mod.rs:280-286 -----------------------------------------------------------------------
def test_count_valid_bases(orc):
    assert orc.count_valid_bases("ACGTACGT") == 8
    assert orc.count_valid_bases("ACNGT") == 4
    assert orc.count_valid_bases("NNN") == 0
    assert orc.count_valid_bases("") == 0


# --- kmer/mod.rs:288-305 test_histogram -------------------------------------------------------
def test_histogram(orc):
    kc = orc.KmerCounts(11)
    for kmer, c in [(1, 5), (20, 5), (2, 7), (11, 11), (12, 12)]:
        kc.insert(kmer, c)
    h = orc.Histogram.from_kmer_counts(kc, 10)
    v = h.get_vector()
    assert len(v) == 12
    assert list(v) == [0, 0, 0, 0, 0, 2, 0, 1, 0, 0, 0, 2]
    assert h.get_n_unique_kmers() == 5 and h.get_n_kmers() == 40


# --- encoding.rs:353-356: invalid character (lowercase included) is an error ---------------
@pytest.mark.parametrize("seq,bad", [("ACGTX", "X"), ("acgt", "a"), ("ACG T", " ")])
def test_invalid_character(orc, seq, bad):
    with pytest.raises(orc.OracleError) as e:
        orc.kmers_from_ascii(seq, 3)
    assert f"Invalid character '{bad}' in sequence. Only ACGTN allowed." in str(e.value)


def test_bad_k(orc):
    for k in (0, 32, 40):
        with pytest.raises(orc.OracleError):
            orc.kmers_from_ascii("ACGT", k)


# ================= counting.rs:365-509 ===============================================================
def test_new_and_basic_ops(orc):
    kc = orc.KmerCounts(5)
    assert kc.get_k() == 5 and kc.is_empty() and len(kc) == 0 and kc.get_n_kmers() == 0


def test_insert_and_get(orc):
    kc = orc.KmerCounts(5)
    kc.insert(42, 3)
    assert not kc.is_empty() and kc.get_count(42) == 3 and kc.contains(42) and not kc.contains(99)


def test_insert_accumulates(orc):
    kc = orc.KmerCounts(5)
    kc.insert(42, 3)
    kc.insert(42, 7)
    assert kc.get_count(42) == 10 and len(kc) == 1


def test_saturating_add(orc):
    kc = orc.KmerCounts(5)
    kc.insert(1, 0xFFFFFFFF)
    kc.insert(1, 1)
    assert kc.get_count(1) == 0xFFFFFFFF


def test_extend_merges(orc):
    a = orc.KmerCounts(5)
    a.insert(1, 10)
    a.insert(2, 20)
    b = orc.KmerCounts(5)
    b.insert(2, 5)
    b.insert(3, 15)
    a.extend(b)
    assert (a.get_count(1), a.get_count(2), a.get_count(3)) == (10, 25, 15)


def test_extend_different_k_fails(orc):
    with pytest.raises(orc.OracleError):
        orc.KmerCounts(5).extend(orc.KmerCounts(7))


def test_median_and_max(orc):
    kc = orc.KmerCounts(5)
    assert kc.get_median_count() == 0
    kc.insert(1, 10)
    kc.insert(2, 20)
    assert kc.get_median_count() == 15  # (10/2)+(20/2)
    kc.insert(3, 30)
    assert kc.get_median_count() == 20
    kc2 = orc.KmerCounts(5)
    for kmer, c in [(1, 5), (2, 100), (3, 50)]:
        kc2.insert(kmer, c)
    assert kc2.get_max_count() == 100


def test_remove_low_count_kmers(orc):
    kc = orc.KmerCounts(5)
    for kmer, c in [(1, 1), (2, 5), (3, 10)]:
        kc.insert(kmer, c)
    kc.remove_low_count_kmers(5)
    assert not kc.contains(1) and kc.contains(2) and kc.contains(3)


def test_ingest_seq(orc):
    kc = orc.KmerCounts(3)
    kc.ingest_seq("ACGT")  # ACG ≡ CGT canonical
    assert kc.get_n_unique_kmers() == 1 and kc.get_n_kmers() == 2


def test_extend_with_histogram(orc):
    a = orc.KmerCounts(5)
    a.insert(1, 3)
    b = orc.KmerCounts(5)
    b.insert(1, 2)
    b.insert(2, 5)
    h = orc.Histogram(100)
    h.move_count(0, 3)
    a.extend_with_histogram(b, h)
    assert a.get_count(1) == 5 and a.get_count(2) == 5
    v = h.get_vector()
    assert v[5] == 2 and v[3] == 0 and v.sum() == 2


def test_extend_with_histogram_saturation(orc):
    """counting.rs:183-200: the histogram follows the STORED (capped) count."""
    a = orc.KmerCounts(5)
    a.insert(7, 0xFFFFFFFE)
    b = orc.KmerCounts(5)
    b.insert(7, 5)
    h = orc.Histogram(10)
    h.move_count(0, 0xFFFFFFFE)
    assert a.extend_with_histogram(b, h) is True
    assert a.get_count(7) == 0xFFFFFFFF
    assert h.get(0xFFFFFFFF) == 1 and h.get(0xFFFFFFFE) == 0
    assert list(h.get_vector()) == [0] * 11 + [1]


def test_get_canonical(orc):
    kc = orc.KmerCounts(3)
    kc.ingest_seq("ACG")
    acg = orc.seq_to_kmer("ACG")
    cgt = orc.seq_to_kmer("CGT")
    assert kc.get_canonical(acg) == 1 and kc.get_canonical(cgt) == 1
    assert kc.get_canonical_count(cgt) == 1 and kc.get_canonical(orc.seq_to_kmer("AAA")) is None


# ---- pcr/primers.rs:593-695: find_oligos_in_kmers (the consumer-side scan of the merged table) ------------

_OLIGO_KATS = [
    # (sequence, k, oligo, min_count, expected k-mer strings or None for "non-empty")
    ("ACGTACGT", 5, "ACG", 1, None),          # test_find_oligos_exact_match_forward
    ("AAAAAAAAAA", 5, "GGG", 1, []),          # test_find_oligos_no_match
    ("AACCCAACC", 5, "AAC", 2, []),           # test_find_oligos_min_count_filter
    ("TTTTTTT", 5, "AAA", 1, ["AAAAA"]),      # test_find_oligos_rc_match
    ("ACGTACGT", 5, "ACGT", 1, None),         # test_find_oligos_oligo_equals_k_minus_1
]


@pytest.mark.parametrize("seq,k,oligo,min_count,expected", _OLIGO_KATS)
def test_find_oligos_in_kmers(orc, seq, k, oligo, min_count, expected):
    kc = orc.KmerCounts(k)
    kc.ingest_seq(seq)
    kmers, counts = kc.find_oligos([orc.seq_to_kmer(oligo)], len(oligo), min_count)
    got = [orc.kmer_to_seq(int(x), k) for x in kmers]
    if expected is None:
        assert got
        assert all(s.startswith(oligo) for s in got)   # reported in the oligo's orientation
    else:
        assert got == expected
    for x, c in zip(kmers, counts):
        assert kc.get_canonical_count(int(x)) == int(c) >= min_count


# ---- pcr/read_filter.rs:58-68 (+ the semantics of matches(), :43-49) ---------------------------------------

def test_filter_empty_set(orc):
    """test_empty_filter: no primer k-mers → nothing matches."""
    assert not orc.KmerCounts(3).filter_matches("ACGTACGT")


def test_filter_matches_semantics(orc):
    primers = orc.KmerCounts(5)
    primers.ingest_seq("ACGTAC")                       # ACGTA, CGTAC (canonical forms)
    assert primers.filter_matches("TTTTACGTACTTTT")    # contains a primer k-mer
    assert primers.filter_matches("AAAAGTACGTAAAA")    # … its reverse complement
    assert not primers.filter_matches("TTTTTTTTTTTT")
    assert not primers.filter_matches("ACGT")          # shorter than k
    assert not primers.filter_matches("ACGNTAC")       # N splits the only candidate window
    assert not primers.filter_matches("ACGTACXTTTT")   # invalid byte: kmers_from_ascii fails → false
    assert not primers.filter_matches("")
