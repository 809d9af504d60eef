"""ctypes binding of the CPU oracle (oracle/liborc.so) + a second, code-independent
sort-based oracle in numpy.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from sharkmer_amd/ (the product path).

References restated (all under /root/reference, caseywdunn/sharkmer v3.1.0):
  src/kmer/encoding.rs:332-376, src/kmer/counting.rs:82-92,144-202,254-260,
  src/kmer/histogram.rs:19-134, src/kmer/chunk.rs:25-30,
  src/io.rs:15,161-198,271-361,378,542-552,977-1161.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")

ORC_OK = 0
ORC_ERR_INVALID_CHAR = -1
ORC_ERR_BAD_K = -2
ORC_ERR_K_MISMATCH = -3
ORC_ERR_NOMEM = -4
ORC_ERR_NO_READS = -5
ORC_ERR_INVARIANT = -6
ORC_ERR_FASTQ = -7
ORC_ERR_IO = -8


class OracleError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code
        self.msg = msg


def build(force: bool = False) -> str:
    """Compile oracle/liborc.so with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "shk_oracle.c")
    hdr = os.path.join(_HERE, "shk_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _RunStats(C.Structure):
    _fields_ = [
        ("n_reads_read", C.c_uint64),
        ("n_bases_read", C.c_uint64),
        ("n_reads_ingested", C.c_uint64),
        ("n_bases_ingested", C.c_uint64),
        ("n_kmers_ingested", C.c_uint64),
        ("n_unique_kmers", C.c_uint64),
        ("n_hashed_kmers", C.c_uint64),
        ("n_singleton_kmers", C.c_uint64),
        ("has_histogram", C.c_int),
        ("any_saturated", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    u8p = C.POINTER(C.c_uint8)
    u64p = C.POINTER(C.c_uint64)
    u32p = C.POINTER(C.c_uint32)
    L.orc_kmers_from_ascii.argtypes = [C.c_char_p, C.c_size_t, C.c_int, u64p, C.POINTER(C.c_size_t), u8p]
    L.orc_kmers_from_ascii.restype = C.c_int
    L.orc_count_valid_bases.argtypes = [C.c_char_p, C.c_size_t]
    L.orc_count_valid_bases.restype = C.c_uint64
    L.orc_revcomp_kmer.argtypes = [C.c_uint64, C.c_int]
    L.orc_revcomp_kmer.restype = C.c_uint64
    L.orc_read_from_str.argtypes = [C.c_char_p, C.c_size_t, u8p]
    L.orc_read_from_str.restype = C.c_long
    L.orc_read_get_kmers.argtypes = [u8p, C.c_size_t, C.c_size_t, C.c_int, u64p, C.POINTER(C.c_size_t)]
    L.orc_read_get_kmers.restype = C.c_int
    L.orc_seq_to_kmer.argtypes = [C.c_char_p, C.c_size_t, u64p]
    L.orc_seq_to_kmer.restype = C.c_int
    L.orc_kmer_to_seq.argtypes = [C.c_uint64, C.c_int, C.c_char_p]
    L.orc_kmer_to_seq.restype = None

    L.orc_counts_new.argtypes = [C.c_int]
    L.orc_counts_new.restype = C.c_void_p
    L.orc_counts_new_with_capacity.argtypes = [C.c_int, C.c_size_t]
    L.orc_counts_new_with_capacity.restype = C.c_void_p
    L.orc_counts_free.argtypes = [C.c_void_p]
    L.orc_counts_free.restype = None
    L.orc_counts_k.argtypes = [C.c_void_p]
    L.orc_counts_k.restype = C.c_int
    L.orc_counts_ingest_seq.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, u8p]
    L.orc_counts_ingest_seq.restype = C.c_int
    L.orc_counts_insert.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
    L.orc_counts_insert.restype = C.c_int
    L.orc_counts_extend.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_counts_extend.restype = C.c_int
    L.orc_counts_get_count.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_counts_get_count.restype = C.c_uint32
    L.orc_counts_contains.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_counts_contains.restype = C.c_int
    L.orc_counts_get_canonical_count.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_counts_get_canonical_count.restype = C.c_uint32
    L.orc_counts_get_canonical.argtypes = [C.c_void_p, C.c_uint64, u32p]
    L.orc_counts_get_canonical.restype = C.c_int
    L.orc_counts_n_kmers.argtypes = [C.c_void_p]
    L.orc_counts_n_kmers.restype = C.c_uint64
    L.orc_counts_n_unique.argtypes = [C.c_void_p]
    L.orc_counts_n_unique.restype = C.c_uint64
    L.orc_counts_max_count.argtypes = [C.c_void_p]
    L.orc_counts_max_count.restype = C.c_uint32
    L.orc_counts_median_count.argtypes = [C.c_void_p]
    L.orc_counts_median_count.restype = C.c_uint32
    L.orc_counts_remove_low.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_counts_remove_low.restype = None
    L.orc_counts_export.argtypes = [C.c_void_p, u64p, u32p]
    L.orc_counts_export.restype = C.c_size_t
    L.orc_find_oligos.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_int, C.c_uint32, u64p, u32p]
    L.orc_find_oligos.restype = C.c_size_t
    L.orc_filter_matches.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.orc_filter_matches.restype = C.c_int
    L.orc_probe_count.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint32,
                                  C.c_void_p, C.c_uint64, C.c_void_p]
    L.orc_probe_count.restype = C.c_int

    L.orc_histo_new.argtypes = [C.c_uint64]
    L.orc_histo_new.restype = C.c_void_p
    L.orc_histo_free.argtypes = [C.c_void_p]
    L.orc_histo_free.restype = None
    L.orc_histo_move_count.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.orc_histo_move_count.restype = None
    L.orc_histo_ingest_counts.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_histo_ingest_counts.restype = None
    L.orc_histo_get.argtypes = [C.c_void_p, C.c_uint64]
    L.orc_histo_get.restype = C.c_uint64
    L.orc_histo_n_kmers.argtypes = [C.c_void_p]
    L.orc_histo_n_kmers.restype = C.c_uint64
    L.orc_histo_n_unique.argtypes = [C.c_void_p]
    L.orc_histo_n_unique.restype = C.c_uint64
    L.orc_histo_get_vector.argtypes = [C.c_void_p, u64p]
    L.orc_histo_get_vector.restype = None
    L.orc_counts_extend_with_histogram.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    L.orc_counts_extend_with_histogram.restype = C.c_int

    L.orc_run_new.argtypes = [C.c_int, C.c_uint32, C.c_uint64]
    L.orc_run_new.restype = C.c_void_p
    L.orc_run_free.argtypes = [C.c_void_p]
    L.orc_run_free.restype = None
    L.orc_run_push_seq.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.orc_run_push_seq.restype = C.c_int
    L.orc_run_push_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_run_push_batch.restype = C.c_int
    L.orc_run_finish.argtypes = [C.c_void_p]
    L.orc_run_finish.restype = C.c_int
    L.orc_run_get_stats.argtypes = [C.c_void_p]
    L.orc_run_get_stats.restype = C.POINTER(_RunStats)
    L.orc_run_histograms.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.orc_run_histograms.restype = u64p
    L.orc_run_merged.argtypes = [C.c_void_p]
    L.orc_run_merged.restype = C.c_void_p
    L.orc_run_error.argtypes = [C.c_void_p]
    L.orc_run_error.restype = C.c_char_p
    L.orc_run_bad_char.argtypes = [C.c_void_p]
    L.orc_run_bad_char.restype = C.c_uint8
    L.orc_run_read_fastq.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]
    L.orc_run_read_fastq.restype = C.c_int
    L.orc_set_gzip_all_members.argtypes = [C.c_int]
    L.orc_set_gzip_all_members.restype = None
    L.orc_run_write_histo.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.orc_run_write_histo.restype = C.c_int
    L.orc_run_write_final_histo.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.orc_run_write_final_histo.restype = C.c_int
    L.orc_run_write_stats_yaml.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64]
    L.orc_run_write_stats_yaml.restype = C.c_int
    _lib = L
    return L


def _b(seq) -> bytes:
    return seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)


# ---- encoding.rs ----------------------------------------------------------

def kmers_from_ascii(seq, k: int) -> list[int]:
    """encoding.rs:332-371."""
    s = _b(seq)
    out = (C.c_uint64 * max(len(s), 1))()
    n = C.c_size_t(0)
    bad = C.c_uint8(0)
    rc = lib().orc_kmers_from_ascii(s, len(s), k, out, C.byref(n), C.byref(bad))
    if rc == ORC_ERR_INVALID_CHAR:
        raise OracleError(rc, f"Invalid character '{chr(bad.value)}' in sequence. Only ACGTN allowed.")
    if rc != ORC_OK:
        raise OracleError(rc, "k must be between 1 and 31")
    return [int(out[i]) for i in range(n.value)]


def count_valid_bases(seq) -> int:
    s = _b(seq)
    return int(lib().orc_count_valid_bases(s, len(s)))


def revcomp_kmer(kmer: int, k: int) -> int:
    return int(lib().orc_revcomp_kmer(kmer, k))


def read_from_str(seq):
    """Read::from_str, encoding.rs:60-95 → (bytes, length)."""
    s = _b(seq)
    out = (C.c_uint8 * (len(s) // 4 + 1))()
    n = lib().orc_read_from_str(s, len(s), out)
    if n < 0:
        raise OracleError(n, "Invalid character in sequence. Only ACGT allowed.")
    return bytes(out[:n]), len(s)


def seq_to_reads(seq):
    """seq_to_reads, encoding.rs:284-298: split on N, drop empty pieces."""
    return [read_from_str(p) for p in _b(seq).split(b"N") if p]


def read_get_kmers(packed: bytes, length: int, k: int) -> list[int]:
    """Read::get_kmers, encoding.rs:130-190."""
    buf = (C.c_uint8 * max(len(packed), 1)).from_buffer_copy(packed or b"\0")
    out = (C.c_uint64 * (len(packed) * 4 + 4))()
    n = C.c_size_t(0)
    rc = lib().orc_read_get_kmers(buf, len(packed), length, k, out, C.byref(n))
    if rc != ORC_OK:
        raise OracleError(rc, "get_kmers failed")
    return [int(out[i]) for i in range(n.value)]


def seq_to_kmer(seq) -> int:
    s = _b(seq)
    out = C.c_uint64(0)
    rc = lib().orc_seq_to_kmer(s, len(s), C.byref(out))
    if rc != ORC_OK:
        raise OracleError(rc, "Invalid base")
    return int(out.value)


def kmer_to_seq(kmer: int, k: int) -> str:
    buf = C.create_string_buffer(k + 1)
    lib().orc_kmer_to_seq(kmer, k, buf)
    return buf.value.decode()


# ---- counting.rs / histogram.rs -------------------------------------------

class KmerCounts:
    """Mirror of reference `KmerCounts` (counting.rs:113-312)."""

    def __init__(self, k: int, capacity: int = 0, _ptr=None, _owned=True):
        self._owned = _owned
        self._p = _ptr if _ptr is not None else lib().orc_counts_new_with_capacity(k, capacity)

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "_p", None):
            lib().orc_counts_free(self._p)
            self._p = None

    def get_k(self):
        return lib().orc_counts_k(self._p)

    def ingest_seq(self, seq):
        s = _b(seq)
        bad = C.c_uint8(0)
        rc = lib().orc_counts_ingest_seq(self._p, s, len(s), C.byref(bad))
        if rc == ORC_ERR_INVALID_CHAR:
            raise OracleError(rc, f"Invalid character '{chr(bad.value)}' in sequence. Only ACGTN allowed.")
        if rc != ORC_OK:
            raise OracleError(rc)

    def insert(self, kmer: int, count: int):
        lib().orc_counts_insert(self._p, kmer, count)

    def extend(self, other: "KmerCounts"):
        rc = lib().orc_counts_extend(self._p, other._p)
        if rc != ORC_OK:
            raise OracleError(rc, "Cannot extend KmerCounts with different k")

    def extend_with_histogram(self, other: "KmerCounts", histo: "Histogram") -> bool:
        sat = C.c_int(0)
        rc = lib().orc_counts_extend_with_histogram(self._p, other._p, histo._p, C.byref(sat))
        if rc != ORC_OK:
            raise OracleError(rc, "Cannot extend KmerCounts with different k")
        return bool(sat.value)

    def get_count(self, kmer):
        return int(lib().orc_counts_get_count(self._p, kmer))

    def contains(self, kmer):
        return bool(lib().orc_counts_contains(self._p, kmer))

    def get_canonical_count(self, kmer):
        return int(lib().orc_counts_get_canonical_count(self._p, kmer))

    def get_canonical(self, kmer):
        c = C.c_uint32(0)
        return int(c.value) if lib().orc_counts_get_canonical(self._p, kmer, C.byref(c)) else None

    def get_n_kmers(self):
        return int(lib().orc_counts_n_kmers(self._p))

    def get_n_unique_kmers(self):
        return int(lib().orc_counts_n_unique(self._p))

    def __len__(self):
        return self.get_n_unique_kmers()

    def is_empty(self):
        return len(self) == 0

    def get_max_count(self):
        return int(lib().orc_counts_max_count(self._p))

    def get_median_count(self):
        return int(lib().orc_counts_median_count(self._p))

    def remove_low_count_kmers(self, min_count):
        lib().orc_counts_remove_low(self._p, min_count)

    def find_oligos(self, oligos, oligo_len: int, min_count: int = 1):
        """find_oligos_in_kmers (pcr/primers.rs:163-226) → (kmers, counts) sorted by k-mer."""
        oligos = np.ascontiguousarray(np.atleast_1d(oligos), dtype=np.uint64)
        n = len(self)
        keys = np.empty(max(n, 1), dtype=np.uint64)
        cnts = np.empty(max(n, 1), dtype=np.uint32)
        m = lib().orc_find_oligos(self._p, oligos.ctypes.data_as(C.POINTER(C.c_uint64)), len(oligos),
                                  oligo_len, min_count, keys.ctypes.data_as(C.POINTER(C.c_uint64)),
                                  cnts.ctypes.data_as(C.POINTER(C.c_uint32)))
        keys, cnts = keys[:m], cnts[:m]
        o = np.argsort(keys, kind="stable")
        return keys[o], cnts[o]

    def filter_matches(self, seq) -> bool:
        """PrimerReadFilter::matches (pcr/read_filter.rs:43-49) with this table as the primer set."""
        b = _b(seq)
        return bool(lib().orc_filter_matches(self._p, b, len(b)))

    def export(self):
        """iter(): (keys u64[n], counts u32[n]) sorted by key for comparison."""
        n = len(self)
        keys = np.empty(n, dtype=np.uint64)
        cnts = np.empty(n, dtype=np.uint32)
        if n:
            lib().orc_counts_export(self._p, keys.ctypes.data_as(C.POINTER(C.c_uint64)),
                                    cnts.ctypes.data_as(C.POINTER(C.c_uint32)))
        o = np.argsort(keys, kind="stable")
        return keys[o], cnts[o]


class Histogram:
    """Mirror of reference `Histogram` (histogram.rs:12-135)."""

    def __init__(self, histo_max: int):
        self.histo_max = histo_max
        self._p = lib().orc_histo_new(histo_max)

    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_histo_free(self._p)
            self._p = None

    @classmethod
    def from_kmer_counts(cls, kc: KmerCounts, histo_max: int):
        h = cls(histo_max)
        lib().orc_histo_ingest_counts(h._p, kc._p)
        return h

    def move_count(self, old, new):
        lib().orc_histo_move_count(self._p, old, new)

    def get(self, count):
        return int(lib().orc_histo_get(self._p, count))

    def get_n_kmers(self):
        return int(lib().orc_histo_n_kmers(self._p))

    def get_n_unique_kmers(self):
        return int(lib().orc_histo_n_unique(self._p))

    def get_vector(self):
        out = np.zeros(self.histo_max + 2, dtype=np.uint64)
        lib().orc_histo_get_vector(self._p, out.ctypes.data_as(C.POINTER(C.c_uint64)))
        return out


# ---- io.rs driver -----------------------------------------------------------

class Run:
    """ingest_reads + consolidate_and_histogram (io.rs:366-595, 977-1161)."""

    def __init__(self, k: int, chunks: int, histo_max: int = 10000):
        self.k, self.chunks, self.histo_max = k, chunks, histo_max
        self._p = lib().orc_run_new(k, chunks, histo_max)

    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_run_free(self._p)
            self._p = None

    def _check(self, rc):
        if rc != ORC_OK:
            raise OracleError(rc, (lib().orc_run_error(self._p) or b"").decode("utf-8", "replace"))

    def push_seq(self, seq):
        s = _b(seq)
        self._check(lib().orc_run_push_seq(self._p, s, len(s)))

    def push_batch(self, bases: np.ndarray, offsets: np.ndarray):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(lib().orc_run_push_batch(self._p, bases.ctypes.data, offsets.ctypes.data,
                                             len(offsets) - 1))

    def read_fastq(self, path: str, max_reads: int = 0, validate_every: int = 0, gzip_all_members: bool = False) -> bool:
        """gzip_all_members: NOT the reference (which reads a gzip file's first member only) — the product's opt-in."""
        reached = C.c_int(0)
        lib().orc_set_gzip_all_members(1 if gzip_all_members else 0)
        try:
            self._check(lib().orc_run_read_fastq(self._p, path.encode(), max_reads, validate_every,
                                                 C.byref(reached)))
        finally:
            lib().orc_set_gzip_all_members(0)
        return bool(reached.value)

    def finish(self):
        self._check(lib().orc_run_finish(self._p))
        return self

    @property
    def stats(self) -> dict:
        s = lib().orc_run_get_stats(self._p).contents
        return {f: int(getattr(s, f)) for f, _ in _RunStats._fields_}

    def histograms(self) -> np.ndarray:
        """histo_vecs (io.rs:1020-1028): shape (chunks, histo_max+2), u64."""
        nc, ln = C.c_size_t(0), C.c_size_t(0)
        p = lib().orc_run_histograms(self._p, C.byref(nc), C.byref(ln))
        if not p or nc.value == 0:
            return np.zeros((0, self.histo_max + 2), dtype=np.uint64)
        a = np.ctypeslib.as_array(p, shape=(nc.value, ln.value))
        return a.copy()

    def merged(self) -> KmerCounts:
        return KmerCounts(self.k, _ptr=lib().orc_run_merged(self._p), _owned=False)

    def write_histo(self, path, version="3.1.0"):
        self._check(lib().orc_run_write_histo(self._p, path.encode(), version.encode()))

    def write_final_histo(self, path, version="3.1.0"):
        self._check(lib().orc_run_write_final_histo(self._p, path.encode(), version.encode()))

    def write_stats_yaml(self, path, command, sample, peak_memory_bytes=0, version="3.1.0"):
        self._check(lib().orc_run_write_stats_yaml(self._p, path.encode(), version.encode(),
                                                   command.encode(), sample.encode(), peak_memory_bytes))


def run_batch(bases: np.ndarray, offsets: np.ndarray, k: int, chunks: int, histo_max: int = 10000):
    """One-shot: histograms + stats for concatenated sequences."""
    r = Run(k, chunks, histo_max)
    r.push_batch(bases, offsets)
    r.finish()
    return r


def probe_counts(bases: np.ndarray, offsets: np.ndarray, k: int, probes: np.ndarray, n_lanes: int = 1,
                 first_read_index: int = 0, threads: int = 0, out: np.ndarray | None = None) -> np.ndarray:
    """Occurrences of every probe k-mer (sorted, distinct u64) over the reads, per chunk lane:
    (n_lanes, len(probes)) u64, extraction by the oracle's kmers_from_ascii, lane of read i =
    (first_read_index + i) // 1000 % n_lanes.  The reads are sharded over `threads` host threads
    (ctypes releases the GIL); `out` accumulates across calls."""
    import concurrent.futures as cf
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    probes = np.ascontiguousarray(probes, dtype=np.uint64)
    n = len(offsets) - 1
    T = threads or min(os.cpu_count() or 1, 32)
    T = max(1, min(T, (n + 999) // 1000))
    cuts = [n * t // T for t in range(T + 1)]
    parts = [np.zeros((n_lanes, len(probes)), dtype=np.uint64) for _ in range(T)]
    L = lib()

    def work(t):
        a, b = cuts[t], cuts[t + 1]
        if b <= a:
            return ORC_OK
        return L.orc_probe_count(bases.ctypes.data, offsets[a:b + 1].ctypes.data, b - a, k, first_read_index + a,
                                 n_lanes, probes.ctypes.data, len(probes), parts[t].ctypes.data)
    with cf.ThreadPoolExecutor(T) as ex:
        for rc in ex.map(work, range(T)):
            if rc != ORC_OK:
                raise OracleError(rc, "orc_probe_count")
    tot = out if out is not None else np.zeros((n_lanes, len(probes)), dtype=np.uint64)
    for p_ in parts:
        tot += p_
    return tot


# ---- second, code-independent oracle (numpy; small inputs) -------------------

_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
_CODE[ord("N")] = 4


def canonical_kmers_numpy(bases: np.ndarray, offsets: np.ndarray, k: int, return_read_id: bool = False):
    """All canonical k-mers of all sequences, by explicit windowing (no rolling
    state, no hash table): a window is valid iff it lies inside one sequence and
    holds no N.  Shares no code with shk_oracle.c."""
    bases = np.asarray(bases, dtype=np.uint8)
    offsets = np.asarray(offsets, dtype=np.int64)
    code = _CODE[bases]
    if (code == 255).any():
        raise OracleError(ORC_ERR_INVALID_CHAR, "invalid character")
    n = len(bases)
    if n < k:
        z = np.zeros(0, dtype=np.uint64)
        return (z, np.zeros(0, dtype=np.int64)) if return_read_id else z
    # window start positions p in [0, n-k]; valid iff same read and no N
    read_id = np.searchsorted(offsets, np.arange(n), side="right") - 1
    is_n = (code == 4).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(is_n)])
    starts = np.arange(n - k + 1)
    ok = (read_id[starts] == read_id[starts + k - 1]) & ((csum[starts + k] - csum[starts]) == 0)
    starts = starts[ok]
    fwd = np.zeros(len(starts), dtype=np.uint64)
    rev = np.zeros(len(starts), dtype=np.uint64)
    c64 = code.astype(np.uint64)
    for j in range(k):
        b = c64[starts + j]
        fwd = (fwd << np.uint64(2)) | b
        rev = rev | ((np.uint64(3) - b) << np.uint64(2 * j))
    canon = np.minimum(fwd, rev)
    return (canon, read_id[starts]) if return_read_id else canon


def sort_count_histogram(bases, offsets, k, chunks, histo_max):
    """Sort-based oracle: chunk of read i = (i // 1000) % n_chunks (io.rs:15,
    340-361); column j = spectrum of the union of chunks 0..j; bins
    1..histo_max exact, last bin = everything above (histogram.rs:125-134)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    n_chunks = max(chunks, 1)
    n_reads = len(offsets) - 1
    out = np.zeros((n_chunks, histo_max + 2), dtype=np.uint64)
    chunk_of_read = (np.arange(n_reads) // 1000) % n_chunks
    canon, rid = canonical_kmers_numpy(bases, offsets, k, return_read_id=True)
    kchunk = chunk_of_read[rid] if len(rid) else np.zeros(0, dtype=np.int64)
    per_chunk = [canon[kchunk == c] for c in range(n_chunks)]
    for j in range(n_chunks):
        allk = np.concatenate(per_chunk[: j + 1])
        if len(allk) == 0:
            continue
        _, cnt = np.unique(allk, return_counts=True)
        cnt = np.minimum(cnt, 0xFFFFFFFF)
        binned = np.minimum(cnt, histo_max + 1)
        out[j] = np.bincount(binned, minlength=histo_max + 2).astype(np.uint64)
    return out
