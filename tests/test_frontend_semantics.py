"""The FASTQ front-end's gzip and stream semantics against the reference's (CPU only).

What open_fastq_reader + read_fastq do with awkward inputs (src/io.rs:598-625, 271-352, 213-265), pinned three ways:
  * explicit cases with the reference's own message texts — a two-member .gz (flate2's GzDecoder reads ONE member), a
    plain file named .gz (the extension forces gzip), truncated streams (never a clean end of file), lines that are not
    UTF-8 (BufRead::lines fails), trailer and deflate damage, gzip by magic, stdin taken as it is;
  * the ORDER of errors: the reference drains 1000 reads at a time (io.rs:340-343), so a reading error at record R only
    ever follows reads [0, ⌊R/1000⌋·1000) — an invalid base among those wins, one among the rest is never seen;
  * a differential sweep: the product's reader (libshk: hand-written inflate, parallel window parse) feeding the
    oracle's counting, against the oracle's own line-by-line reader (zlib raw inflate) on randomly damaged inputs —
    same histograms and counters, or the same error text.
"""
import gzip
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

import sharkmer_amd as sa

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EM = "\u2014"


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as g
    g.build()


def product_run(orc, paths, k=11, chunks=2, histo_max=50, max_reads=0, validate_every=0, batch=(777, 1 << 16), all_members=False):
    """libshk's reader feeding the oracle's counting: what shk_run_files does, with the oracle in the engine's place."""
    run = orc.Run(k, chunks, histo_max)
    r = sa.FastqReader(paths, max_reads=max_reads, validate_every=validate_every, gzip_all_members=all_members)
    try:
        while True:
            b, o = r.next_batch(max_seqs=batch[0], max_bases=batch[1])
            if len(o) > 1:
                run.push_batch(b, o)
            if r.stats()["done"]:
                break
        st = r.stats()
    finally:
        r.close()
    res = run.finish()
    assert st["n_reads_read"] == res.stats["n_reads_read"] and st["n_bases_read"] == res.stats["n_bases_read"]
    return res


def oracle_run(orc, paths, k=11, chunks=2, histo_max=50, max_reads=0, validate_every=0, batch=None, all_members=False):
    run = orc.Run(k, chunks, histo_max)
    for p in paths:
        if run.read_fastq(p, max_reads=max_reads, validate_every=validate_every, gzip_all_members=all_members):
            break
    return run.finish()


def outcome(fn, orc, *a, **kw):
    try:
        res = fn(orc, *a, **kw)
    except (sa.ShkError, orc.OracleError) as e:
        return ("error", e.msg)
    st = res.stats
    return ("ok", res.histograms().tobytes(), tuple(st[x] for x in ("n_reads_read", "n_bases_read", "n_reads_ingested",
                                                                     "n_bases_ingested", "n_kmers_ingested")))


def both(orc, paths, **kw):
    a = outcome(product_run, orc, paths, **kw)
    b = outcome(oracle_run, orc, paths, **kw)
    assert a == b, (a[:2] if a[0] == "error" else a[0], b[:2] if b[0] == "error" else b[0])
    return a


def records(n, seed=0, length=(20, 60)):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        L = int(rng.integers(*length))
        seq = "".join(rng.choice(list("ACGTN"), size=L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        out.append([f"@r{i} d", seq, "+", "I" * L])
    return out


def text_of(recs, eol="\n"):
    return "".join(eol.join(r) + eol for r in recs).encode()


def gz_bytes(data, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    return co.compress(data) + co.flush()


def write(tmp_path, name, data):
    p = tmp_path / name
    p.write_bytes(data)
    return str(p)


# ---- first member only --------------------------------------------------------------------------------------------

def test_two_member_gzip_reads_the_first_member_only(orc, tmp_path):
    """flate2::read::GzDecoder (io.rs:618-621) decodes ONE member: 5 reads, not 12.  The fixture is committed."""
    p = os.path.join(G, "reads_two_members.fastq.gz")
    raw = open(p, "rb").read()
    second = raw.index(b"\x1f\x8b\x08", 10)
    assert gzip.decompress(raw).count(b"\n") == 4 * 12 and zlib.decompress(raw[:second], 31).count(b"\n") == 4 * 5
    res = both(orc, [p], k=5)
    assert res[0] == "ok" and res[2][0] == 5
    # the same bytes under a name without the extension: the magic decides (io.rs:611-616)
    q = write(tmp_path, "two_members.bin", raw)
    assert both(orc, [q], k=5) == res


def test_all_members_on_request(orc, tmp_path):
    """The opt-in that is NOT the reference's behaviour (shk_fastq_open_ex, SHK_FASTQ_GZIP_ALL_MEMBERS): every member, the
    way flate2::read::MultiGzDecoder reads — the fixture's 12 reads; a bgzip-like file of many small members with an
    empty one at the end; what is not a member behind a member is a header error; a bad CRC in a LATER member is that
    member's error, after everything before it has been read; a member cut short is the stream ending early."""
    p = os.path.join(G, "reads_two_members.fastq.gz")
    res = both(orc, [p], k=5, all_members=True)
    assert res[0] == "ok" and res[2][0] == 12
    recs = records(3000, 7)
    blocks = [text_of(recs[i:i + 37]) for i in range(0, len(recs), 37)]          # members end on record boundaries …
    text = text_of(recs)
    ragged = [text[i:i + 3001] for i in range(0, len(text), 3001)]               # … or anywhere
    for name, parts in (("blocks", blocks), ("ragged", ragged)):
        q = write(tmp_path, f"{name}.fastq.gz", b"".join(gz_bytes(x, 1) for x in parts) + gz_bytes(b""))
        res = both(orc, [q], all_members=True)
        assert res[0] == "ok" and res[2][0] == 3000, name
        first = both(orc, [q])                                                    # (the default: the first member — 37 reads, or a cut record)
        assert first[2][0] == 37 if name == "blocks" else first[0] == "error", first[:2]
    a, b = text_of(records(1200, 1)), text_of(records(900, 2))
    q = write(tmp_path, "junk.fastq.gz", gz_bytes(a) + b"not a gzip member at all")
    res = both(orc, [q], all_members=True)
    assert res == ("error", f"Failed to read header line of record 1201 in {q}: invalid gzip header (kind InvalidInput)")
    second = bytearray(gz_bytes(b))
    second[-6] ^= 0x40                                                            # the second member's CRC-32
    q = write(tmp_path, "badcrc.fastq.gz", gz_bytes(a) + bytes(second))
    res = both(orc, [q], all_members=True)
    assert res[0] == "error" and "record 2101" in res[1] and "matching checksum" in res[1], res
    q = write(tmp_path, "short.fastq.gz", gz_bytes(a) + gz_bytes(b)[:-300])
    res = both(orc, [q], all_members=True)
    assert res[0] == "error" and "Local read stream ended unexpectedly" in res[1], res


def test_a_member_has_no_history_but_its_own(orc, tmp_path):
    """All-members mode decodes member after member into one buffer.  A later member whose FIRST symbol is a match
    (fixed-Huffman block: length 3, distance 1, end of block) reaches back in front of its own first byte:
    flate2's MultiGzDecoder starts every member with empty history and reports "corrupt deflate stream" — and so must
    the reader, instead of copying the previous member's last byte and leaving it to the CRC."""
    bits = [1, 1, 0] + [0, 0, 0, 0, 0, 0, 1] + [0, 0, 0, 0, 0] + [0] * 7     # BFINAL, BTYPE = 01; code 257; distance code 0; code 256
    raw = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        raw[i >> 3] |= b << (i & 7)
    a = text_of(records(1200, 1))
    ghost = a[-1:] * 3                                                        # what a decoder without the check would produce
    member2 = (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + bytes(raw) +
               zlib.crc32(ghost).to_bytes(4, "little") + len(ghost).to_bytes(4, "little"))
    q = write(tmp_path, "reach_back.fastq.gz", gz_bytes(a) + member2)
    res = both(orc, [q], all_members=True)
    assert res[0] == "error" and "corrupt deflate stream" in res[1] and "kind InvalidInput" in res[1], res
    assert both(orc, [q])[2][0] == 1200                                       # (the default never looks at a second member)


def test_member_followed_by_garbage_and_members_across_files(orc, tmp_path):
    a = text_of(records(1200, 1))
    b = text_of(records(900, 2))
    p1 = write(tmp_path, "a.fastq.gz", gz_bytes(a) + b"not a gzip member at all")
    p2 = write(tmp_path, "b.fq.gzip", gz_bytes(b) + gz_bytes(a))
    res = both(orc, [p1, p2])
    assert res[0] == "ok" and res[2][0] == 2100


# ---- the extension forces gzip ------------------------------------------------------------------------------------

def test_plain_file_named_gz_is_a_header_error(orc, tmp_path):
    p = write(tmp_path, "plain.fastq.gz", text_of(records(5)))
    res = both(orc, [p])
    assert res == ("error", f"Failed to read header line of record 1 in {p}: invalid gzip header (kind InvalidInput)")
    # behind a good file the record number is the GLOBAL one (state persists across files, io.rs:498-512)
    good = write(tmp_path, "good.fastq", text_of(records(3)))
    res = both(orc, [good, p])
    assert res == ("error", f"Failed to read header line of record 4 in {p}: invalid gzip header (kind InvalidInput)")


def test_gzip_by_magic_without_extension(orc, tmp_path):
    p = write(tmp_path, "reads.dat", gz_bytes(text_of(records(1500, 3))))
    res = both(orc, [p])
    assert res[0] == "ok" and res[2][0] == 1500


def test_empty_file_named_gz(orc, tmp_path):
    p = write(tmp_path, "empty.fastq.gz", b"")
    res = both(orc, [p])
    assert res == ("error", f"Local read stream ended unexpectedly while reading header line of record 1 in {p} "
                            f"(I/O error: unexpected end of file {EM} kind UnexpectedEof). The file may be truncated or corrupted.")
    # an empty PLAIN file is no error of the reader's (the run then fails with "No reads were ingested")
    q = write(tmp_path, "empty.fastq", b"")
    assert both(orc, [q]) == ("error", "No reads were ingested. Check that input files contain valid FASTQ records.")


# ---- truncated streams ---------------------------------------------------------------------------------------------

def _stream_ended(role, rec, p):
    return (f"Local read stream ended unexpectedly while reading {role} line of record {rec} in {p} "
            f"(I/O error: unexpected end of file {EM} kind UnexpectedEof). The file may be truncated or corrupted.")


def test_truncated_gzip_is_never_a_clean_end(orc, tmp_path):
    recs = records(40, 4)
    data = text_of(recs)
    z = gz_bytes(data)
    # cut the COMPRESSED stream at many places: whatever still decodes is read, then the stream-ended error for the line
    # the reader was on (io.rs:226-250) — never "Truncated FASTQ record", never fewer reads and no error
    seen_roles = set()
    for cut in range(11, len(z)):
        p = write(tmp_path, f"cut{cut}.fastq.gz", z[:cut])
        got = zlib.decompressobj(-15).decompress(z[10:cut])
        n_lines = got.count(b"\n")
        role = ["header", "sequence", "separator", "quality"][n_lines % 4]
        seen_roles.add(role)
        res = both(orc, [p])
        assert res == ("error", _stream_ended(role, n_lines // 4 + 1, p)), cut
    assert len(seen_roles) >= 3
    # the whole member minus its trailer decodes every record — and is still an error (the 8 trailer bytes are not there)
    p = write(tmp_path, "notrailer.fastq.gz", z[:-8])
    assert both(orc, [p]) == ("error", _stream_ended("header", 41, p))
    # cut inside the gzip header
    p = write(tmp_path, "hdr.fastq.gz", z[:6])
    assert both(orc, [p]) == ("error", _stream_ended("header", 1, p))


def test_truncated_plain_file_keeps_the_reference_messages(orc, tmp_path):
    data = text_of(records(3, 5))
    lines = data.split(b"\n")
    p = write(tmp_path, "t.fastq", b"\n".join(lines[:9]) + b"\n")   # record 3: header only
    assert both(orc, [p]) == ("error", f"Truncated FASTQ record at record 3 in {p}: missing sequence line")
    # the same text, complete as a gzip member: the record is short, the stream is not
    q = write(tmp_path, "t.fastq.gz", gz_bytes(b"\n".join(lines[:11]) + b"\n"))
    assert both(orc, [q]) == ("error", f"Truncated FASTQ record at record 3 in {q}: missing quality line")


def test_trailer_and_deflate_damage(orc, tmp_path):
    data = text_of(records(30, 6))
    z = bytearray(gz_bytes(data))
    z[-6] ^= 0x40   # CRC-32
    p = write(tmp_path, "crc.fastq.gz", bytes(z))
    msg = "corrupt gzip stream does not have a matching checksum"
    assert both(orc, [p]) == ("error", f"Failed to read header line of record 31 in {p}: {msg} (kind InvalidInput)")
    z = bytearray(gz_bytes(data))
    z[-1] ^= 1      # ISIZE
    p = write(tmp_path, "isize.fastq.gz", bytes(z))
    assert both(orc, [p]) == ("error", f"Failed to read header line of record 31 in {p}: {msg} (kind InvalidInput)")
    # the last line has no newline: the failed read takes it along — it is the quality line of record 30 that fails
    z = bytearray(gz_bytes(data[:-1]))
    z[-6] ^= 0x40
    p = write(tmp_path, "crc2.fastq.gz", bytes(z))
    assert both(orc, [p]) == ("error", f"Failed to read quality line of record 30 in {p}: {msg} (kind InvalidInput)")
    # a stored block whose length check fails: a decoder error (zio::read's "corrupt deflate stream")
    raw = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + b"\x01\x05\x00\x00\x00" + b"@r\nAC"
    p = write(tmp_path, "bad.fastq.gz", raw)
    assert both(orc, [p]) == ("error", f"Failed to read header line of record 1 in {p}: corrupt deflate stream (kind InvalidInput)")
    # reserved block type 3
    p = write(tmp_path, "bad3.fastq.gz", b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + b"\x07" + bytes(20))
    assert both(orc, [p]) == ("error", f"Failed to read header line of record 1 in {p}: corrupt deflate stream (kind InvalidInput)")


def test_gzip_header_fields(orc, tmp_path):
    data = text_of(records(7, 7))
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(data) + co.flush()
    tr = struct.pack("<II", zlib.crc32(data), len(data))
    hdr = bytes([0x1f, 0x8b, 8, 4 | 8 | 16 | 2, 1, 2, 3, 4, 0, 3]) + struct.pack("<H", 6) + b"extra!" + b"name.fq\0" + b"a comment\0"
    hc = struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    p = write(tmp_path, "fields.gz", hdr + hc + body + tr)
    res = both(orc, [p])
    assert res[0] == "ok" and res[2][0] == 7
    p = write(tmp_path, "fields_badcrc.gz", hdr + b"\x00\x00" + body + tr)
    assert both(orc, [p])[1].startswith(f"Failed to read header line of record 1 in {p}: corrupt gzip stream does not have a matching checksum")
    p = write(tmp_path, "reserved.gz", bytes([0x1f, 0x8b, 8, 0x20]) + bytes(6) + body + tr)
    assert both(orc, [p])[1] == f"Failed to read header line of record 1 in {p}: invalid gzip header (kind InvalidInput)"
    p = write(tmp_path, "method.gz", bytes([0x1f, 0x8b, 7, 0]) + bytes(6) + body + tr)
    assert both(orc, [p])[1] == f"Failed to read header line of record 1 in {p}: invalid gzip header (kind InvalidInput)"
    long_name = bytes([0x1f, 0x8b, 8, 8]) + bytes(6) + b"n" * 65536 + b"\0" + body + tr
    p = write(tmp_path, "longname.gz", long_name)
    assert both(orc, [p])[1] == f"Failed to read header line of record 1 in {p}: gzip header field too long (kind InvalidInput)"
    ok_name = bytes([0x1f, 0x8b, 8, 8]) + bytes(6) + b"n" * 65535 + b"\0" + body + tr
    p = write(tmp_path, "okname.gz", ok_name)
    assert both(orc, [p])[0] == "ok"


# ---- lines that are not UTF-8 --------------------------------------------------------------------------------------

@pytest.mark.parametrize("gz", [False, True])
@pytest.mark.parametrize("line,role", [(0, "header"), (1, "sequence"), (2, "separator"), (3, "quality")])
def test_line_that_is_not_utf8(orc, tmp_path, line, role, gz):
    """BufRead::lines yields InvalidData for such a line (io.rs:282-318 → stream_io_error), whatever the validation
    cadence — record 1201 here is never validated."""
    recs = records(1500, 8)
    data = text_of(recs).split(b"\n")
    i = 4 * 1200 + line
    data[i] = data[i][:1] + b"\xff" + data[i][1:]
    blob = b"\n".join(data)
    p = write(tmp_path, f"u{line}.fastq" + (".gz" if gz else ""), gz_bytes(blob) if gz else blob)
    res = both(orc, [p])
    assert res == ("error", f"Failed to read {role} line of record 1201 in {p}: stream did not contain valid UTF-8 (kind InvalidData)")


def test_utf8_edge_forms(orc, tmp_path):
    ok = ["é", "€", "😀", "\u07ff", "\ud7ff", "\ue000", "\U0010ffff"]
    bad = [b"\xc0\xaf", b"\xc1\xbf", b"\xe0\x80\xaf", b"\xed\xa0\x80", b"\xf0\x8f\xbf\xbf", b"\xf4\x90\x80\x80", b"\xf5\x80\x80\x80",
           b"\x80", b"\xe2\x82", b"\xc3"]
    for j, s in enumerate(ok):   # valid multi-byte text in a header is just a header
        p = write(tmp_path, f"ok{j}.fastq", f"@r {s}\nACGTACGTACGTA\n+\nIIIIIIIIIIIII\n".encode())
        assert both(orc, [p])[0] == "ok", s
    for j, s in enumerate(bad):
        p = write(tmp_path, f"bad{j}.fastq", b"@r " + s + b"\nACGTACGTACGTA\n+\nIIIIIIIIIIIII\n")
        assert both(orc, [p])[1] == f"Failed to read header line of record 1 in {p}: stream did not contain valid UTF-8 (kind InvalidData)", s
    # a header that starts with a multi-byte character: '{}' of chars().next() prints the whole character
    p = write(tmp_path, "hdr.fastq", "é1\nACGT\n+\nIIII\n".encode())
    assert both(orc, [p])[1] == "FASTQ record 1 has invalid header (expected '@', got 'é'): é1"
    # valid UTF-8 in the SEQUENCE reaches kmers_from_ascii, which prints `b as char` of the first byte (0xC3 → 'Ã')
    p = write(tmp_path, "seq.fastq", "@r\nACGTéACGT\n+\nIIIIIIIIII\n".encode())
    assert both(orc, [p])[1] == "Invalid character 'Ã' in sequence. Only ACGTN allowed."
    # the last line without a newline, not UTF-8: still a failed read of that line
    p = write(tmp_path, "tail.fastq", b"@r\nACGT\n+\nII\xffI")
    assert both(orc, [p])[1] == f"Failed to read quality line of record 1 in {p}: stream did not contain valid UTF-8 (kind InvalidData)"
    # in a record that is short of lines the bad line is met before the end of the file is
    p = write(tmp_path, "short.fastq", b"@r\nACGT\n+\nIIII\n@r2\nAC\xffGT\n")
    assert both(orc, [p])[1] == f"Failed to read sequence line of record 2 in {p}: stream did not contain valid UTF-8 (kind InvalidData)"


# ---- which error comes first ------------------------------------------------------------------------------------------

def test_an_error_only_follows_what_the_reference_had_drained(orc, tmp_path):
    recs = records(3500, 9)
    recs[1500][1] = "ACGTXACGT"     # an invalid base in the thousand 1000..1999
    recs[1500][3] = "I" * 9

    def variant(name, bad_rec=None, cut_after=None, gz=False, extra=None):
        r2 = [list(r) for r in recs]
        if bad_rec is not None:
            r2[bad_rec][3] = r2[bad_rec][3] + "I"   # quality longer than the sequence
        data = text_of(r2[:cut_after] if cut_after else r2)
        if extra:
            data += extra
        return write(tmp_path, name, gz_bytes(data) if gz else data)

    bad_base = ("error", "Invalid character 'X' in sequence. Only ACGTN allowed.")
    # a flawed record at 1999 (validate_every = 1999): reads 1000..1998 were never drained → the record's error
    p = variant("a.fastq", bad_rec=1999)
    assert both(orc, [p], validate_every=1999)[1] == "FASTQ record 2000 has mismatched sequence (%d) and quality (%d) lengths" % (
        len(recs[1999][1]), len(recs[1999][1]) + 1)
    # the same flaw one record later: the thousand with the bad base was drained first
    p = variant("b.fastq", bad_rec=2000)
    assert both(orc, [p], validate_every=2000) == bad_base
    assert both(orc, [p], validate_every=2000, batch=(5000, 1 << 20)) == bad_base
    # a file that ends inside record 1800 / 2100
    p = variant("c.fastq", cut_after=1799, extra=b"@x\nACGT\n")
    assert both(orc, [p])[1] == f"Truncated FASTQ record at record 1800 in {p}: missing separator line"
    p = variant("d.fastq", cut_after=2099, extra=b"@x\nACGT\n")
    assert both(orc, [p]) == bad_base
    # a second file that cannot be opened, met at record 1700 resp. 2300
    p = variant("e.fastq", cut_after=1700)
    assert both(orc, [p, "/nonexistent/x.fastq"])[1] == "Failed to open file: /nonexistent/x.fastq"
    p = variant("f.fastq", cut_after=2300)
    assert both(orc, [p, "/nonexistent/x.fastq"]) == bad_base
    # --max-reads stops in front of an error: nothing behind the last read is looked at (io.rs:345-348)
    p = variant("g.fastq", bad_rec=1200)
    assert both(orc, [p], validate_every=1, max_reads=1200)[0] == "ok"
    assert both(orc, [p], validate_every=1, max_reads=1201)[1].startswith("FASTQ record 1201 has mismatched")
    assert both(orc, [p, "/nonexistent/x.fastq"], max_reads=1400)[0] == "ok"


def test_stdin_is_read_as_it_is(tmp_path):
    """io.rs:517-537: stdin is never decompressed — gzip bytes on it are lines that are not UTF-8."""
    data = text_of(records(1200, 10))
    plain = write(tmp_path, "in.fastq", data)
    gz = write(tmp_path, "in.fastq.gz", gz_bytes(data))
    code = ("import sys; sys.path.insert(0, %r); import sharkmer_amd as sa\n"
            "r = sa.FastqReader([])\n"
            "n = 0\n"
            "try:\n"
            "    while not r.stats()['done']:\n"
            "        b, o = r.next_batch(); n += len(o) - 1\n"
            "    print('ok', n)\n"
            "except sa.ShkError as e:\n"
            "    print('error', e.msg)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], stdin=open(plain, "rb"), capture_output=True, text=True)
    assert out.stdout.strip() == "ok 1200", out.stderr
    out = subprocess.run([sys.executable, "-c", code], stdin=open(gz, "rb"), capture_output=True, text=True)
    assert out.stdout.strip() == "error Failed to read header line of record 1 in stdin: stream did not contain valid UTF-8 (kind InvalidData)"


def test_named_pipe_plain_and_gzip(orc, tmp_path):
    """A source that cannot be mapped (a FIFO): sniffed for the magic like a file, then streamed / inflated."""
    import threading
    data = text_of(records(2500, 11))
    for blob, name in [(data, "p.fastq"), (gz_bytes(data), "p2.fastq"), (gz_bytes(data), "p3.fq.gz")]:
        fifo = str(tmp_path / name)
        os.mkfifo(fifo)
        outs = []
        for fn in (product_run, oracle_run):
            t = threading.Thread(target=lambda: open(fifo, "wb").write(blob))
            t.start()
            outs.append(outcome(fn, orc, [fifo]))
            t.join()
        assert outs[0] == outs[1] and outs[0][0] == "ok" and outs[0][2][0] == 2500


# ---- differential sweep -------------------------------------------------------------------------------------------------

def _damage(rng, recs):
    """0-2 random flaws in a list of records; returns bytes."""
    recs = [list(r) for r in recs]
    n = len(recs)
    crlf = rng.random() < 0.15
    for _ in range(int(rng.integers(0, 3))):
        if n == 0:
            break
        i = int(rng.integers(0, n))
        kind = int(rng.integers(0, 9))
        if len(recs[i]) < 4:
            continue
        if kind == 0:
            s = recs[i][1]
            if s:
                j = int(rng.integers(0, len(s)))
                recs[i][1] = s[:j] + str(rng.choice(list("XacgtRn.-*"))) + s[j + 1:]
        elif kind == 1:
            recs[i][3] = recs[i][3] + "I"
        elif kind == 2:
            recs[i][0] = "r" + recs[i][0][1:]
        elif kind == 3:
            recs[i][2] = "-"
        elif kind == 4:
            recs[i][0] = ">" + recs[i][0][1:]
        elif kind == 5:
            recs[i][int(rng.integers(0, 4))] += "\udcff"   # becomes the byte 0xff below
        elif kind == 6:
            recs[i][int(rng.integers(0, 4))] += "é"
        elif kind == 7:
            del recs[i][int(rng.integers(0, 4))]
        else:
            recs[i][1] = ""
            recs[i][3] = ""
    eol = "\r\n" if crlf else "\n"
    data = "".join(eol.join(r) + eol for r in recs).encode("utf-8", "surrogateescape")
    r = rng.random()
    if r < 0.15 and data:
        data = data[:int(rng.integers(0, len(data)))]
    elif r < 0.3 and data.endswith(b"\n"):
        data = data[:-1]
    return data


def _wrap(rng, data, tmp_path, idx):
    """plain or some flavour of gzip"""
    r = rng.random()
    if r < 0.35:
        return write(tmp_path, f"f{idx}.fastq", data)
    z = gz_bytes(data, int(rng.choice([1, 6, 9])))
    r = rng.random()
    name = f"f{idx}.fastq.gz" if rng.random() < 0.7 else f"f{idx}.bin"
    if r < 0.2 and len(z) > 20:
        z = z[:int(rng.integers(1, len(z)))]
    elif r < 0.3:
        z = z + gz_bytes(b"@second\nACGT\n+\nIIII\n")
    elif r < 0.4 and len(z) > 30:
        zz = bytearray(z)
        zz[int(rng.integers(10, len(z)))] ^= 1 << int(rng.integers(0, 8))
        z = bytes(zz)
    return write(tmp_path, name, z)


@pytest.mark.parametrize("seed", range(int(os.environ.get("SHK_FUZZ_SEEDS", "6"))))
def test_differential_sweep_against_the_oracle_reader(orc, tmp_path, monkeypatch, seed):
    rng = np.random.default_rng(1000 + seed)
    monkeypatch.setenv("SHK_FASTQ_WINDOW_KB", str(int(rng.choice([1, 3, 16, 64]))))
    n_err = n_ok = 0
    for case in range(40):
        files = []
        for f in range(int(rng.integers(1, 4))):
            n = int(rng.choice([0, 1, 3, 999, 1000, 1001, 2500])) if rng.random() < 0.5 else int(rng.integers(0, 2600))
            data = _damage(rng, records(n, int(rng.integers(1 << 30)), length=(0, 40)))
            files.append(_wrap(rng, data, tmp_path, f"{case}_{f}"))
        kw = dict(validate_every=int(rng.choice([0, 1, 7, 1000])), max_reads=int(rng.choice([0, 0, 1, 1000, 1500, 4000])),
                  batch=(int(rng.choice([1, 777, 1000, 5000])), 1 << 18))
        res = both(orc, files, **kw)
        n_err += res[0] == "error"
        n_ok += res[0] == "ok"
    assert n_err > 5 and n_ok > 5


# ---- one gzip member, many threads ------------------------------------------------------------------------------------

def _reader_outcome(paths, **kw):
    r = sa.FastqReader(paths, **kw)
    seqs = []
    try:
        while not r.stats()["done"]:
            b, o = r.next_batch(max_seqs=5_000, max_bases=1 << 20)
            seqs += [b[int(o[i]):int(o[i + 1])].tobytes() for i in range(len(o) - 1)]
        return ("ok", seqs)
    except sa.ShkError as e:
        return ("error", e.msg, len(seqs))
    finally:
        r.close()


def _gz_variants(rng, data):
    """The same bytes compressed in ways that put every kind of block and boundary into the stream."""
    out = {}
    for level in (1, 6, 9):
        out[f"level{level}"] = gz_bytes(data, level)
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
    out["fixed"] = co.compress(data) + co.flush()
    co = zlib.compressobj(0, zlib.DEFLATED, 31)
    out["stored"] = co.compress(data) + co.flush()
    # a stream of many short blocks of changing kinds: sync and full flushes, level changes, stored stretches
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    parts, at = [], 0
    while at < len(data):
        n = int(rng.integers(500, 60_000))
        parts.append(co.compress(data[at:at + n]))
        parts.append(co.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH, zlib.Z_BLOCK]))))
        at += n
    parts.append(co.flush())
    out["flushes"] = b"".join(parts)
    return out


@pytest.mark.parametrize("chunk_kb,threads", [(1, 3), (4, 5), (64, 2)])
def test_parallel_member_decode_is_the_sequential_decode(orc, tmp_path, monkeypatch, chunk_kb, threads):
    """A large gzip member is decoded by several threads (speculative symbol decoding from block boundaries found inside
    the stream, verified and resolved in order).  With the threshold forced to zero and chunks of a few KiB — so that the
    stream is cut hundreds of times, inside stored and fixed blocks too — everything must come out exactly as from the
    one-thread decoder: the reads, and on damaged streams the error text and how many reads came before it."""
    rng = np.random.default_rng(chunk_kb)
    data = text_of(records(6_000, 77 + chunk_kb, length=(30, 160)))
    variants = _gz_variants(rng, data)
    # damaged ones: cut short, a flipped bit, a second member behind, a wrong CRC
    z = variants["level6"]
    variants["cut"] = z[:len(z) * 2 // 3]
    zz = bytearray(z)
    zz[len(z) // 2] ^= 0x10
    variants["flipped"] = bytes(zz)
    variants["two_members"] = z + gz_bytes(b"@x\nACGT\n+\nIIII\n")
    zz = bytearray(z)
    zz[-7] ^= 1
    variants["bad_crc"] = bytes(zz)
    for name, blob in variants.items():
        p = write(tmp_path, f"{name}.fastq.gz", blob)
        monkeypatch.setenv("SHK_PGZ_THREADS", "1")
        want = _reader_outcome([p])
        monkeypatch.setenv("SHK_PGZ_THREADS", str(threads))
        monkeypatch.setenv("SHK_PGZ_MIN_KB", "0")
        monkeypatch.setenv("SHK_PGZ_CHUNK_KB", str(chunk_kb))
        got = _reader_outcome([p])
        monkeypatch.delenv("SHK_PGZ_MIN_KB")
        assert got == want, (name, got[:2] if got[0] == "error" else got[0], want[:2] if want[0] == "error" else want[0])
        if name in ("level1", "level6", "level9", "fixed", "stored", "flushes", "two_members"):
            assert got[0] == "ok" and len(got[1]) == 6_000, name
        assert both(orc, [p])[0] == got[0], name   # (and the oracle's own reader agrees on the kind of outcome)


def test_member_checksum_at_every_small_size(orc, tmp_path):
    """The member's CRC-32 is checked over everything written (flate2: "corrupt gzip stream does not have a matching
    checksum" otherwise); the product computes it by carry-less multiplication from 256 bytes on, 16 bytes at a time
    with a table-driven tail — every output size from 207 to 1007 bytes, against trailers zlib wrote; and one flipped
    trailer bit is still caught."""
    for L in range(100, 501):
        seq = "ACGT" * (L // 4) + "ACGT"[:L % 4]
        data = f"@r\n{seq}\n+\n{'I' * L}\n".encode()
        p = write(tmp_path, "one.fastq.gz", gz_bytes(data, 1))
        r = sa.FastqReader([p])
        try:
            b, o = r.next_batch(max_seqs=10, max_bases=1 << 16)
            assert len(o) == 2 and bytes(b) == seq.encode(), L
            assert r.stats()["done"]
        finally:
            r.close()
    raw = bytearray(gz_bytes(data, 1))
    raw[-7] ^= 0x10
    p = write(tmp_path, "bad.fastq.gz", bytes(raw))
    res = both(orc, [p])
    assert res[0] == "error" and "matching checksum" in res[1]


def test_many_gzip_files_take_the_parallel_decoder_in_turn(orc, tmp_path, monkeypatch):
    """Eight inputs whose producers all start at once — large members (the many-thread decoder, forced here by
    SHK_PGZ_MIN_KB=0), a plain file, a member too damaged to start, stdin-like small ones — decode one after the other
    in FILE order (a ticket per file, drawn by the reader): the outcome is the oracle's, also when --max-reads ends
    the run while later files are still waiting for their turn, and when an early file fails."""
    monkeypatch.setenv("SHK_PGZ_MIN_KB", "0")
    monkeypatch.setenv("SHK_PGZ_CHUNK_KB", "16")
    monkeypatch.setenv("SHK_PGZ_THREADS", "4")
    paths = []
    for i in range(8):
        data = text_of(records(1500 + 100 * i, 40 + i, length=(60, 120)))
        if i == 3:
            paths.append(write(tmp_path, f"f{i}.fastq", data))
        else:
            paths.append(write(tmp_path, f"f{i}.fastq.gz", gz_bytes(data, 1 + i % 9)))
    res = both(orc, paths)
    assert res[0] == "ok" and res[2][0] == sum(1500 + 100 * i for i in range(8))
    cut = both(orc, paths, max_reads=3300)
    assert cut[0] == "ok" and cut[2][0] == 3300
    bad = bytearray(open(paths[2], "rb").read())
    bad[len(bad) // 2] ^= 0x55
    paths[2] = write(tmp_path, "f2bad.fastq.gz", bytes(bad))
    res = both(orc, paths)
    assert res[0] == "error"
