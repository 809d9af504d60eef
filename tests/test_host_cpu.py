"""CPU tests of the host side either side of the path: libshk's C++
FASTQ front-end and writers against the oracle's restatement of read_fastq / the reference's
file formats, on the committed fixtures (tests/golden) and on malformed inputs
(tests/spcr_18s.rs:559-652).  Nothing here touches the GPU: the reader parses, the writers
format."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import sharkmer_amd as sa

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as g
    g.build()


def read_all(paths, **kw):
    r = sa.FastqReader(paths, **kw)
    bs, os_ = [], [np.zeros(1, dtype=np.uint64)]
    while True:
        b, o = r.next_batch(max_seqs=777, max_bases=1 << 16)  # odd sizes on purpose
        if len(o) > 1:
            bs.append(b)
            os_.append(o[1:] + os_[-1][-1])
        if r.stats()["done"]:
            break
    st = r.stats()
    r.close()
    return (np.concatenate(bs) if bs else np.zeros(0, np.uint8)), np.concatenate(os_), st


def oracle_read(orc, paths, k=21, chunks=1, **kw):
    run = orc.Run(k, chunks, 50)
    for p in paths:
        if run.read_fastq(p, **kw):
            break
    return run


def test_reader_matches_oracle_reader_on_fixtures(orc):
    paths = [os.path.join(G, "reads_main.fastq.gz"), os.path.join(G, "reads_part2.fastq")]
    bases, offs, st = read_all(paths)
    ref = oracle_read(orc, paths).finish()
    assert st["n_reads_read"] == ref.stats["n_reads_read"] == 2510
    assert st["n_bases_read"] == ref.stats["n_bases_read"]
    # same sequences in the same order ⇒ same histograms through the oracle
    again = orc.run_batch(bases, offs, 21, 3, 50)
    want = oracle_read(orc, paths, chunks=3).finish()
    assert np.array_equal(again.histograms(), want.histograms())


def test_reader_crlf_and_plain(orc):
    p = os.path.join(G, "reads_crlf.fastq")
    bases, offs, st = read_all([p])
    assert st["n_reads_read"] == 40 and b"\r" not in bases.tobytes()
    ref = oracle_read(orc, [p]).finish()
    assert st["n_bases_read"] == ref.stats["n_bases_read"]


@pytest.mark.parametrize("max_reads", [1, 999, 1000, 1001, 2300, 2400])
def test_max_reads_spans_files(orc, max_reads):
    """io.rs:345-348, 498-512."""
    paths = [os.path.join(G, "reads_main.fastq.gz"), os.path.join(G, "reads_part2.fastq")]
    _, offs, st = read_all(paths, max_reads=max_reads)
    assert st["n_reads_read"] == max_reads == len(offs) - 1 and st["reached_max"]


def _tmp(tmp_path, name, text, gz=False):
    p = tmp_path / name
    if gz:
        with gzip.open(p, "wb") as f:
            f.write(text.encode())
    else:
        p.write_bytes(text.encode())
    return str(p)


BAD = {
    "fasta": (">r1\nACGT\n>r2\nACGT\n", "Input appears to be FASTA format, not FASTQ (record 1 starts with '>')"),
    "qual": ("@r1\nACGT\n+\nIII\n", "FASTQ record 1 has mismatched sequence (4) and quality (3) lengths"),
    "header": ("r1\nACGT\n+\nIIII\n", "FASTQ record 1 has invalid header (expected '@', got 'r'): r1"),
    "sep": ("@r1\nACGT\n-\nIIII\n", "FASTQ record 1 has invalid separator line (expected '+', got '-'): -"),
    "trunc_seq": ("@r1\n", "Truncated FASTQ record at record 1 in {path}: missing sequence line"),
    "trunc_qual": ("@r1\nACGT\n+\nIIII\n@r2\nAC\n+\n", "Truncated FASTQ record at record 2 in {path}: missing quality line"),
}


@pytest.mark.parametrize("case", sorted(BAD))
@pytest.mark.parametrize("gz", [False, True])
def test_malformed_fastq_messages_match_oracle(orc, tmp_path, case, gz):
    """tests/spcr_18s.rs:588-652 + io.rs:161-198, 287-318: same error text as the restated reader."""
    text, msg = BAD[case]
    p = _tmp(tmp_path, f"{case}.fastq" + (".gz" if gz else ""), text, gz)
    with pytest.raises(sa.ShkError) as e:
        read_all([p])
    assert e.value.code == -7
    assert msg.format(path=p) in e.value.msg
    with pytest.raises(orc.OracleError) as eo:
        oracle_read(orc, [p])
    assert eo.value.msg == e.value.msg


def test_validate_every(orc, tmp_path):
    """io.rs:321-322: record 0 and every Nth are validated; others are not."""
    recs = ["@r%d\nACGT\n+\nIIII\n" % i for i in range(6)]
    recs[3] = "@r3\nACGT\n+\nII\n"  # bad quality length at record index 3
    p = _tmp(tmp_path, "ve.fastq", "".join(recs))
    _, offs, _ = read_all([p])  # validate_every = 0: only the first record is checked
    assert len(offs) - 1 == 6
    with pytest.raises(sa.ShkError) as e:
        read_all([p], validate_every=3)
    assert "FASTQ record 4 has mismatched sequence (4) and quality (2) lengths" in e.value.msg
    _, offs, _ = read_all([p], validate_every=2)  # 0,2,4 checked; 3 is not
    assert len(offs) - 1 == 6


def _ragged_fastq(rng, n, crlf_every=0, last_newline=True):
    recs, seqs = [], []
    for i in range(n):
        L = int(rng.integers(0, 300))
        seq = "".join(rng.choice(list("ACGTN"), size=L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        eol = "\r\n" if crlf_every and i % crlf_every == 0 else "\n"
        recs.append(f"@read{i} some description{eol}{seq}{eol}+{eol}{'I' * L}{eol}")
        seqs.append(seq)
    text = "".join(recs)
    if not last_newline:
        text = text[:-1]
    return text, seqs


@pytest.mark.parametrize("window_kb,last_newline", [(64, True), (3, True), (64, False), (0, True)])
def test_parallel_plain_parse_equals_sequential_semantics(tmp_path, monkeypatch, window_kb, last_newline):
    """Plain files are parsed in windows by a pool of threads (newline scan per share, prefix sum of line
    numbers, parallel copy-out): the sequences, their order and the counters must be what the line-by-line
    reader gives — windows of 64 KiB and 3 KiB here, so records straddle shares and windows; CRLF every
    7th record; an unterminated last line."""
    if window_kb:
        monkeypatch.setenv("SHK_FASTQ_WINDOW_KB", str(window_kb))
    rng = np.random.default_rng(window_kb + 1)
    text, seqs = _ragged_fastq(rng, 6_000, crlf_every=7, last_newline=last_newline)
    if not last_newline and seqs[-1] == "":   # (an empty unterminated last line is no line at all)
        pytest.skip("degenerate tail")
    p = _tmp(tmp_path, "ragged.fastq", text)
    bases, offs, st = read_all([p])
    got = [bases[int(offs[i]):int(offs[i + 1])].tobytes().decode() for i in range(len(offs) - 1)]
    assert got == seqs
    assert st["n_reads_read"] == len(seqs) and st["n_bases_read"] == sum(len(x) for x in seqs)


def test_later_streams_run_ahead_but_order_and_errors_stay_sequential(tmp_path, monkeypatch):
    """Five inputs — gzip, plain, gzip, gzip, plain — are produced concurrently and consumed in order; a flaw
    in the fourth file surfaces with the GLOBAL record number and only if the cadence reaches it."""
    monkeypatch.setenv("SHK_FASTQ_WINDOW_KB", "16")
    rng = np.random.default_rng(5)
    paths, all_seqs = [], []
    for i in range(5):
        text, seqs = _ragged_fastq(rng, 900 + 37 * i)
        paths.append(_tmp(tmp_path, f"f{i}.fastq" + ("" if i in (1, 4) else ".gz"), text, gz=i not in (1, 4)))
        all_seqs += seqs
    bases, offs, st = read_all(paths)
    got = [bases[int(offs[i]):int(offs[i + 1])].tobytes().decode() for i in range(len(offs) - 1)]
    assert got == all_seqs and st["n_reads_read"] == len(all_seqs)
    _, offs, st = read_all(paths, max_reads=2_000)   # stops inside the third file; the producers are cancelled
    assert st["n_reads_read"] == 2_000 and st["reached_max"]
    # a record with a short quality line in the fourth file: global index 900 + 937 + 974 + 5 = 2816
    text, _ = _ragged_fastq(np.random.default_rng(9), 300)
    lines = text.split("\n")
    lines[4 * 5 + 1] = "ACGTACGT"
    lines[4 * 5 + 3] = "III"
    bad = _tmp(tmp_path, "bad.fastq.gz", "\n".join(lines), gz=True)
    paths2 = paths[:3] + [bad] + paths[4:]
    read_all(paths2)                                   # cadence: record 0 only → not looked at
    with pytest.raises(sa.ShkError) as e:
        read_all(paths2, validate_every=2816)
    assert "FASTQ record 2817 has mismatched sequence (8) and quality (3) lengths" in e.value.msg
    read_all(paths2, validate_every=2815)              # 2815 and 5630 are fine records


def test_empty_file_yields_no_reads(tmp_path):
    p = _tmp(tmp_path, "empty.fastq", "")
    _, offs, st = read_all([p])
    assert len(offs) == 1 and st["n_reads_read"] == 0


def test_missing_file():
    with pytest.raises(sa.ShkError) as e:
        read_all(["/nonexistent/x.fastq"])
    assert "Failed to open file: /nonexistent/x.fastq" in e.value.msg


def test_long_sequence_carries_over_to_next_batch(tmp_path):
    seqs = ["ACGT" * 10, "A" * 3000, "ACG", "T" * 2999]
    p = _tmp(tmp_path, "long.fastq", "".join(f"@r\n{s}\n+\n{'I' * len(s)}\n" for s in seqs))
    r = sa.FastqReader([p])
    got = []
    while not r.stats()["done"]:
        b, o = r.next_batch(max_seqs=10, max_bases=3001)
        for i in range(len(o) - 1):
            got.append(b[int(o[i]):int(o[i + 1])].tobytes().decode())
    assert got == seqs


# ---- writers: byte-for-byte against the oracle's restatement of io.rs:1051-1094 / stats.rs ----------

def test_histo_writers_byte_exact(orc, tmp_path):
    paths = [os.path.join(G, "reads_main.fastq.gz"), os.path.join(G, "reads_part2.fastq")]
    ref = oracle_read(orc, paths, chunks=3).finish()
    h = ref.histograms()
    sa.write_histo(str(tmp_path / "a.histo"), h, 21, 50)
    sa.write_final_histo(str(tmp_path / "a.final.histo"), h, 21, 50)
    assert (tmp_path / "a.histo").read_bytes() == open(os.path.join(G, "golden_k21_c3.histo"), "rb").read()
    assert (tmp_path / "a.final.histo").read_bytes() == open(os.path.join(G, "golden_k21_c3.final.histo"), "rb").read()
    lines = (tmp_path / "a.histo").read_text().split("\n")
    assert lines[0] == "# sharkmer 3.1.0 k=21 chunks=3" and lines[1] == "count\tchunk_1\tchunk_2\tchunk_3"
    assert len(lines) == 2 + 51 + 1 and lines[-2].startswith("51\t")


def test_stats_yaml_matches_oracle_writer(orc, tmp_path):
    paths = [os.path.join(G, "reads_main.fastq.gz"), os.path.join(G, "reads_part2.fastq")]
    for chunks in (3, 0):
        ref = oracle_read(orc, paths, chunks=chunks).finish()
        st = ref.stats
        ref.write_stats_yaml(str(tmp_path / "o.yaml"), "sharkmer -k 21 -s x reads.fastq", "x", 12345)
        sa.write_stats_yaml(str(tmp_path / "s.yaml"), command="sharkmer -k 21 -s x reads.fastq", sample="x",
                            kmer_length=21, chunks=chunks, n_reads_read=st["n_reads_read"],
                            n_bases_read=st["n_bases_read"], n_subreads_ingested=st["n_reads_ingested"],
                            n_bases_ingested=st["n_bases_ingested"], n_kmers=st["n_kmers_ingested"],
                            n_multi_kmers=st["n_kmers_ingested"] - st["n_singleton_kmers"],
                            n_singleton_kmers=st["n_singleton_kmers"], peak_memory_bytes=12345)
        got = (tmp_path / "s.yaml").read_text()
        assert got == (tmp_path / "o.yaml").read_text()
        assert ("n_singleton_kmers:" in got) == (chunks > 0)  # Option fields skipped when None
    # the viewer's loader (sharkmer_viewer.py:136-148) reads n_bases_read with a YAML parser
    import yaml
    d = yaml.safe_load(got)
    assert d["n_bases_read"] == st["n_bases_read"] and d["sample"] == "x" and d["kmer_length"] == 21


def test_stats_yaml_quotes_awkward_strings(tmp_path):
    import yaml
    sa.write_stats_yaml(str(tmp_path / "q.yaml"), command="shk: -k 21 #x", sample="123", kmer_length=21,
                        chunks=0, n_reads_read=1, n_bases_read=2, n_subreads_ingested=1,
                        n_bases_ingested=2, n_kmers=0)
    d = yaml.safe_load((tmp_path / "q.yaml").read_text())
    assert d["command"] == "shk: -k 21 #x" and d["sample"] == "123"


# ---- argument validation (cli.rs:659-677, 645-652) ---------------------------------------------------

@pytest.mark.parametrize("k,hm,sample,msg", [
    (32, 10, "s", "k must be less than 32 due to use of 64 bit integers to encode kmers"),
    (0, 10, "s", "k must be greater than 0"),
    (20, 10, "s", "k must be odd"),
    (21, 0, "s", "histo_max must be greater than 0"),
    (21, 1_000_001, "s", "histo_max must not exceed 1000000, got 1000001"),
    (21, 10, None, "--sample is required."),
    (21, 10, "a/b", "Sample name 'a/b' contains characters that are unsafe for filenames."),
])
def test_validate_args(k, hm, sample, msg):
    with pytest.raises(sa.ShkError) as e:
        sa.validate_args(k, hm, sample)
    assert msg in e.value.msg
    sa.validate_args(21, 10, "ok_name-1.x")


def test_cli_rejects_bad_arguments_before_touching_the_gpu():
    exe = os.path.join(ROOT, "sharkmer_amd", "csrc", "shk_count")
    r = subprocess.run([exe, "-k", "20", "-s", "x", os.path.join(G, "reads_part2.fastq")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "Error: k must be odd" in r.stderr
    r = subprocess.run([exe, "-k", "21", os.path.join(G, "reads_part2.fastq")], capture_output=True, text=True)
    assert r.returncode == 1 and "--sample is required" in r.stderr
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 2


def test_record_longer_than_a_parse_window(tmp_path, monkeypatch):
    monkeypatch.setenv("SHK_FASTQ_WINDOW_KB", "1")
    seqs = ["ACGT" * 700, "TTGCA" * 900, "A" * 10]
    p = _tmp(tmp_path, "wide.fastq", "".join(f"@r\n{s}\n+\n{'I' * len(s)}\n" for s in seqs))
    bases, offs, _ = read_all([p])
    assert [bases[int(offs[i]):int(offs[i + 1])].tobytes().decode() for i in range(3)] == seqs


def test_histo_files_load_the_way_the_viewer_loads_them(orc, tmp_path):
    """The consumer of the output format: sharkmer_viewer reads a .histo with pandas' read_csv(sep="\\t", comment="#")
    (sharkmer_viewer.py:117-131: the `#` line is skipped, the header row names the columns, column 0 is the count) and
    genomescopemovie.sh slices the numeric rows column by column (:37-50)."""
    pd = pytest.importorskip("pandas")
    paths = [os.path.join(G, "reads_main.fastq.gz"), os.path.join(G, "reads_part2.fastq")]
    ref = oracle_read(orc, paths, chunks=3).finish()
    h = ref.histograms()
    sa.write_histo(str(tmp_path / "v.histo"), h, 21, 50)
    sa.write_final_histo(str(tmp_path / "v.final.histo"), h, 21, 50)
    df = pd.read_csv(tmp_path / "v.histo", sep="\t", comment="#")
    assert list(df.columns) == ["count", "chunk_1", "chunk_2", "chunk_3"] and len(df) == 51
    assert list(df["count"]) == list(range(1, 52))
    for j in range(3):
        assert np.array_equal(df[f"chunk_{j + 1}"].to_numpy(dtype=np.uint64), h[j, 1:])
    fin = pd.read_csv(tmp_path / "v.final.histo", sep="\t", comment="#")
    assert list(fin.columns) == ["count", "frequency"] and np.array_equal(fin["frequency"].to_numpy(dtype=np.uint64), h[2, 1:])
    # genomescopemovie.sh: the numeric rows, one column at a time, as "count<space>frequency" pairs
    rows = [ln.split("\t") for ln in (tmp_path / "v.histo").read_text().splitlines() if ln and ln[0].isdigit()]
    assert len(rows) == 51 and [int(r[0]) for r in rows] == list(range(1, 52)) and [int(r[2]) for r in rows] == list(h[1, 1:])


@pytest.mark.timeout(120)
def test_two_readers_on_large_gzip_files_in_lockstep(tmp_path, monkeypatch):
    """Two readers open at once, each on a gzip member that takes the many-thread decoder, consumed in lockstep by one
    thread — what read_fastq_paired does with R1 / R2 (io.rs:629-700).  The decoder's turn-taking is per READER (a
    reader's files take the decoder in file order): with one process-wide gate the second reader's first batch waited
    for the first reader to be drained, which a lockstep consumer never does.  Small windows and chunks so that the
    decoder of a ~1 MB member really waits on its consumer (three windows ahead at most)."""
    monkeypatch.setenv("SHK_FASTQ_WINDOW_KB", "32")
    monkeypatch.setenv("SHK_PGZ_MIN_KB", "1")
    monkeypatch.setenv("SHK_PGZ_CHUNK_KB", "16")
    monkeypatch.setenv("SHK_PGZ_THREADS", "3")
    rng = np.random.default_rng(11)
    ta, sa_ = _ragged_fastq(rng, 7_000)
    tb, sb = _ragged_fastq(rng, 7_000)
    pa, pb = _tmp(tmp_path, "r1.fastq.gz", ta, gz=True), _tmp(tmp_path, "r2.fastq.gz", tb, gz=True)
    ra, rb = sa.FastqReader([pa]), sa.FastqReader([pb])
    got_a, got_b = [], []

    def take(r, out):
        b, o = r.next_batch(max_seqs=100, max_bases=1 << 16)
        out += [b[int(o[i]):int(o[i + 1])].tobytes().decode() for i in range(len(o) - 1)]
        return r.stats()["done"]
    done_a = done_b = False
    while not (done_a and done_b):
        if not done_a:
            done_a = take(ra, got_a)
        if not done_b:
            done_b = take(rb, got_b)   # (used to block here until reader A had been drained)
    ra.close()
    rb.close()
    assert got_a == sa_ and got_b == sb
    # … and a reader left open and undrained does not stall a later one
    rc_ = sa.FastqReader([pa])
    take(rc_, [])
    bases, offs, st = read_all([pb])
    assert st["n_reads_read"] == len(sb)
    rc_.close()
