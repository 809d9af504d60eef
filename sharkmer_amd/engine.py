"""ctypes host mirror of the libshk C ABI (include/shk.h).

`KmerEngine` plays the role the reference's `FastqReadState` + `Chunk`s +
`consolidate_and_histogram` play in src/io.rs: sequences go in (explicit chunk
like drain_batch, io.rs:355-361, or striped by read index like read_fastq,
io.rs:335-343), histograms and totals come out (io.rs:1020-1028, 545-552).
Errors surface as `ShkError` carrying the reference's message text.

The library is loaded from sharkmer_amd/csrc/libshk.so (in-tree).  If it is
missing the import of this module still works, but any use raises — there is no
fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

N_READS_PER_BATCH = 1000  # io.rs:15

FLAG_TIMING = 1
FLAG_FORCE_DIRECT = 2
FLAG_FORCE_PAGED = 4
FLAG_TIMING_SAMPLED = 16  # with FLAG_TIMING: only every 4th job's launches are bracketed (include/shk.h)
RESERVE_NONE = 0xFFFFFFFF  # shk_config.reserve_cus: every compute unit (include/shk.h)
FASTQ_GZIP_ALL_MEMBERS = 1  # shk_fastq_open_ex / shk_run_config.fastq_flags (include/shk.h)
FLAG_DEFER_ERRORS = 8  # host-buffer ingests return once queued; errors surface at the next call (include/shk.h)

KERNEL_NAMES = ["mark", "scan", "direct", "scatter", "pages", "histo", "grow", "insert",
                "lookup", "export", "synth", "merge", "pcount", "pscan", "histo_rows"]

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path() -> str:
    # SHK_LIB_PATH: experiment hook to A/B two builds of the same HIP library in one session
    return os.environ.get("SHK_LIB_PATH") or os.path.join(_HERE, "csrc", "libshk.so")


class ShkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[shk {code}] {msg}")
        self.code = code
        self.msg = msg


class _Config(C.Structure):
    _fields_ = [("k", C.c_uint32), ("chunks", C.c_uint32), ("histo_max", C.c_uint64),
                ("device", C.c_int32), ("flags", C.c_uint32),
                ("table_capacity_hint", C.c_uint64), ("n_owners", C.c_uint32), ("owner_id", C.c_uint32),
                ("n_devices", C.c_uint32), ("reserve_cus", C.c_uint32), ("device_ids", C.POINTER(C.c_int32)),
                ("reserved", C.c_uint64 * 1)]


class XchgLayout(C.Structure):
    """shk_xchg_layout: how one exchange round's owner segments are laid out (include/shk.h)."""
    _fields_ = [("n_owners", C.c_uint32), ("n_lanes", C.c_uint32), ("log_p1", C.c_uint32),
                ("regions", C.c_uint32), ("region_cap", C.c_uint32), ("record_bytes", C.c_uint32),
                ("segment_records", C.c_uint64)]


class _Counters(C.Structure):
    _fields_ = [("n_reads_ingested", C.c_uint64), ("n_bases_read", C.c_uint64),
                ("n_bases_ingested", C.c_uint64), ("n_kmers_ingested", C.c_uint64),
                ("n_unique_kmers", C.c_uint64), ("n_hashed_kmers", C.c_uint64),
                ("n_singleton_kmers", C.c_uint64), ("any_saturated", C.c_uint32),
                ("n_chunks", C.c_uint32), ("table_capacity", C.c_uint64),
                ("n_grows", C.c_uint64), ("n_spilled", C.c_uint64)]


class _Timings(C.Structure):
    _fields_ = [("ms", C.c_double * 16), ("launches", C.c_uint64 * 16)]


class _RunStats(C.Structure):
    _fields_ = [("sharkmer_version", C.c_char_p), ("command", C.c_char_p), ("sample", C.c_char_p),
                ("kmer_length", C.c_uint32), ("chunks", C.c_uint32), ("n_reads_read", C.c_uint64),
                ("n_bases_read", C.c_uint64), ("n_subreads_ingested", C.c_uint64),
                ("n_bases_ingested", C.c_uint64), ("n_kmers", C.c_uint64), ("n_multi_kmers", C.c_uint64),
                ("n_singleton_kmers", C.c_uint64), ("peak_memory_bytes", C.c_uint64),
                ("has_histogram", C.c_uint32), ("reserved", C.c_uint32)]


class _RunConfig(C.Structure):
    _fields_ = [("inputs", C.POINTER(C.c_char_p)), ("n_inputs", C.c_uint32), ("k", C.c_uint32),
                ("chunks", C.c_uint32), ("device", C.c_int32), ("histo_max", C.c_uint64),
                ("max_reads", C.c_uint64), ("validate_every", C.c_uint64), ("sample", C.c_char_p),
                ("outdir", C.c_char_p), ("command", C.c_char_p), ("version", C.c_char_p),
                ("table_capacity_hint", C.c_uint64), ("batch_reads", C.c_uint64),
                ("batch_bases", C.c_uint64), ("n_devices", C.c_uint32), ("fastq_flags", C.c_uint32),
                ("device_ids", C.POINTER(C.c_int32))]


class _Synth(C.Structure):
    _fields_ = [("seed_genome", C.c_uint64), ("seed_reads", C.c_uint64), ("genome_len", C.c_uint64),
                ("read_len", C.c_uint32), ("sub_per_64k", C.c_uint32), ("n_per_64k", C.c_uint32),
                ("reserved", C.c_uint32)]


# every symbol include/shk.h declares (tests check the .so exports them all)
ABI_SYMBOLS = [
    "shk_abi_version", "shk_create", "shk_destroy", "shk_reset", "shk_last_error", "shk_ingest_batch",
    "shk_ingest_reads", "shk_set_read_index", "shk_ingest_reads_device", "shk_insert_counts", "shk_sync", "shk_finalize",
    "shk_histograms", "shk_get_counters", "shk_get_timings", "shk_reset_timings",
    "shk_export_table", "shk_lookup", "shk_find_oligos", "shk_filter_reads", "shk_kmers_from_reads", "shk_table_geometry", "shk_table_reserve_pages", "shk_owner_counts", "shk_compact_owners",
    "shk_merge_entries",
    "shk_table_device_ptrs", "shk_merge_pages", "shk_set_owned_pages", "shk_alloc_pinned",
    "shk_free_pinned", "shk_alloc_device", "shk_free_device", "shk_release_cached_memory", "shk_synth_reads_device",
    "shk_fastq_open", "shk_fastq_open_ex", "shk_fastq_close", "shk_fastq_error", "shk_fastq_next_batch", "shk_fastq_next_batch_packed", "shk_fastq_stats",
    "shk_write_histo", "shk_write_final_histo", "shk_write_stats_yaml", "shk_validate_args",
    "shk_run_error", "shk_run_files",
    "shk_xchg_scatter_device", "shk_xchg_absorb", "shk_xchg_spill", "shk_xchg_spill_clear", "shk_insert_device",
    "shk_xchg_wide_scatter_device", "shk_xchg_feasible", "shk_xchg_scatter_begin", "shk_xchg_scatter_end",
    "shk_stream", "shk_compact_owners_packed", "shk_compact_owners_fixed", "shk_merge_pieces_max", "shk_merge_pieces", "shk_set_owner_share", "shk_finalize_begin", "shk_finalize_end",
    "shk_packed_sizes", "shk_pack_reads", "shk_ingest_packed", "shk_ingest_packed_device", "shk_pack_reads_device",
    "shk_unpack_reads_device",
]

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, the same
    as /opt/rocm's).  Two HIP/HSA runtimes in one process cannot both open the GPU, so when
    torch is installed we load ITS copy first (by path, without importing torch); libshk's
    NEEDED libamdhip64.so.7 then binds to it, and a later `import torch` reuses the same
    file.  Without torch, libshk uses the system ROCm runtime."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


FRONT_SYMBOLS = [
    "shk_fastq_open", "shk_fastq_open_ex", "shk_fastq_close", "shk_fastq_error", "shk_fastq_next_batch", "shk_fastq_next_batch_packed", "shk_fastq_stats",
    "shk_write_histo", "shk_write_final_histo", "shk_write_stats_yaml", "shk_validate_args", "shk_run_error",
    "shk_packed_sizes", "shk_pack_reads",
]


def _type_front(L):
    """The entry points of the plain-C++ host side (csrc/shk_front.cpp): FASTQ front-end, packer, writers."""
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.shk_packed_sizes.argtypes = [u64, C.POINTER(u64), C.POINTER(u64)]
    L.shk_packed_sizes.restype = None
    L.shk_pack_reads.argtypes = [vp, u64, vp, vp, u32]
    L.shk_fastq_open.argtypes = [C.POINTER(C.c_char_p), u32, u64, u64, C.POINTER(vp)]
    L.shk_fastq_open_ex.argtypes = [C.POINTER(C.c_char_p), u32, u64, u64, u32, C.POINTER(vp)]
    L.shk_fastq_close.argtypes = [vp]
    L.shk_fastq_close.restype = None
    L.shk_fastq_error.argtypes = [vp]
    L.shk_fastq_error.restype = C.c_char_p
    L.shk_fastq_next_batch.argtypes = [vp, vp, u64, vp, u64, C.POINTER(u64)]
    L.shk_fastq_next_batch_packed.argtypes = [vp, vp, vp, u64, vp, u64, C.POINTER(u64)]
    L.shk_fastq_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.shk_write_histo.argtypes = [C.c_char_p, C.c_char_p, u32, u32, u64, vp]
    L.shk_write_final_histo.argtypes = [C.c_char_p, C.c_char_p, u32, u32, u64, vp]
    L.shk_write_stats_yaml.argtypes = [C.c_char_p, C.POINTER(_RunStats)]
    L.shk_validate_args.argtypes = [u32, u64, C.c_char_p]
    L.shk_run_error.argtypes = []
    L.shk_run_error.restype = C.c_char_p


_front = None


def load_front_library():
    """The library the host-side entry points are called in: libshk.so — or, with SHK_FRONT_LIB set, a build of
    csrc/shk_front.cpp + shk_inflate.cpp alone (the sanitizer builds of `make -C sharkmer_amd/csrc san`, which
    tests/test_host_san.py runs the CPU tests against)."""
    global _front
    if _front is not None:
        return _front
    alt = os.environ.get("SHK_FRONT_LIB")
    if not alt:
        _front = load_library()
        return _front
    L = C.CDLL(alt)
    _type_front(L)
    for name in FRONT_SYMBOLS:
        getattr(L, name)
    _front = L
    return L


def load_library():
    """dlopen libshk.so and type its entry points.  Raises if the HIP build is missing."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise ShkError(-9, f"{p} not found: build it with __graft_entry__.build() "
                           f"(hipcc --offload-arch=gfx950); libshk has no CPU fallback")
    _share_hip_runtime_with_torch()
    L = C.CDLL(p)
    vp, u64, u32 = C.c_void_p, C.c_uint64, C.c_uint32
    L.shk_abi_version.restype = C.c_int
    L.shk_create.argtypes = [C.POINTER(_Config), C.POINTER(vp)]
    L.shk_create.restype = C.c_int
    L.shk_destroy.argtypes = [vp]
    L.shk_destroy.restype = None
    L.shk_reset.argtypes = [vp]
    L.shk_last_error.argtypes = [vp]
    L.shk_last_error.restype = C.c_char_p
    L.shk_ingest_batch.argtypes = [vp, u32, vp, vp, u64]
    L.shk_ingest_reads.argtypes = [vp, vp, vp, u64]
    L.shk_ingest_reads_device.argtypes = [vp, vp, vp, u64, u64]
    L.shk_insert_counts.argtypes = [vp, u32, vp, vp, u64]
    L.shk_sync.argtypes = [vp]
    L.shk_finalize.argtypes = [vp]
    L.shk_histograms.argtypes = [vp, vp]
    L.shk_get_counters.argtypes = [vp, C.POINTER(_Counters)]
    L.shk_get_timings.argtypes = [vp, C.POINTER(_Timings)]
    L.shk_reset_timings.argtypes = [vp]
    L.shk_export_table.argtypes = [vp, vp, vp, u64, C.POINTER(u64)]
    L.shk_lookup.argtypes = [vp, vp, vp, u64, C.c_int]
    L.shk_find_oligos.argtypes = [vp, vp, u32, u32, u32, vp, vp, u64, C.POINTER(u64)]
    L.shk_filter_reads.argtypes = [vp, vp, vp, u64, vp, u64, vp]
    L.shk_kmers_from_reads.argtypes = [vp, vp, vp, u64, vp, u64, vp, vp]
    L.shk_owner_counts.argtypes = [vp, u32, vp]
    L.shk_compact_owners.argtypes = [vp, u32, vp, vp, vp, u64, C.c_int32]
    L.shk_merge_entries.argtypes = [vp, vp, vp, u64, u64]
    L.shk_compact_owners_packed.argtypes = [vp, u32, vp, vp, C.c_int32]
    L.shk_compact_owners_fixed.argtypes = [vp, u32, u64, vp, C.c_int32]
    L.shk_merge_pieces_max.argtypes = [vp, vp]
    L.shk_merge_pieces.argtypes = [vp, vp, u32, u64, C.c_int32]
    L.shk_set_owner_share.argtypes = [vp, u32, u32]
    L.shk_finalize_begin.argtypes = [vp, u64, C.POINTER(vp), C.POINTER(u64)]
    L.shk_finalize_end.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(u64)]
    L.shk_table_geometry.argtypes = [vp, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32)]
    L.shk_table_reserve_pages.argtypes = [vp, u64]
    L.shk_table_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.shk_merge_pages.argtypes = [vp, u64, u64, vp, vp, u64]
    L.shk_set_read_index.argtypes = [vp, u64]
    L.shk_xchg_scatter_device.argtypes = [vp, vp, vp, u64, u64, u64, C.POINTER(vp), C.POINTER(vp),
                                          C.POINTER(XchgLayout), C.POINTER(u64)]
    L.shk_xchg_scatter_begin.argtypes = [vp, vp, vp, u64, u64, u64, C.POINTER(vp), C.POINTER(vp), C.POINTER(XchgLayout)]
    L.shk_xchg_scatter_end.argtypes = [vp, C.POINTER(u64)]
    L.shk_xchg_absorb.argtypes = [vp, vp, vp, C.POINTER(XchgLayout)]
    L.shk_xchg_spill.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    L.shk_xchg_spill_clear.argtypes = [vp]
    L.shk_insert_device.argtypes = [vp, vp, vp, vp, u64]
    L.shk_release_cached_memory.argtypes = []
    L.shk_release_cached_memory.restype = None
    L.shk_xchg_wide_scatter_device.argtypes = [vp, vp, vp, u64, u64, C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)]
    L.shk_xchg_feasible.argtypes = [vp]
    L.shk_stream.argtypes = [vp]
    L.shk_stream.restype = vp
    L.shk_ingest_packed.argtypes = [vp, vp, vp, vp, u64]
    L.shk_ingest_packed_device.argtypes = [vp, vp, vp, vp, u64, u64]
    L.shk_pack_reads_device.argtypes = [vp, vp, u64, vp, vp]
    L.shk_unpack_reads_device.argtypes = [vp, vp, vp, u64, vp]
    L.shk_set_owned_pages.argtypes = [vp, u64, u64]
    L.shk_alloc_pinned.argtypes = [C.c_size_t]
    L.shk_alloc_pinned.restype = vp
    L.shk_free_pinned.argtypes = [vp]
    L.shk_free_pinned.restype = None
    L.shk_alloc_device.argtypes = [vp, C.c_size_t]
    L.shk_alloc_device.restype = vp
    L.shk_free_device.argtypes = [vp, vp]
    L.shk_free_device.restype = None
    L.shk_synth_reads_device.argtypes = [vp, C.POINTER(_Synth), u64, u64, vp, vp]
    _type_front(L)
    L.shk_run_files.argtypes = [C.POINTER(_RunConfig), C.POINTER(_RunStats)]
    for name in ABI_SYMBOLS:
        getattr(L, name)  # AttributeError here = the .so is stale
    _lib = L
    return L


class KmerEngine:
    """One counting context on one GPU.

    k, chunks, histo_max: the reference's -k / --chunks / --histo-max
    (cli.rs:207-224; validated like cli.rs:659-677 except "k odd", which is the
    CLI's rule).  capacity_hint: expected number of distinct k-mers."""

    def __init__(self, k: int, chunks: int = 0, histo_max: int = 10000, device: int = 0,
                 capacity_hint: int = 0, flags: int = 0, n_owners: int = 0, owner_id: int = 0,
                 device_ids=None, reserve_cus: int = 0):
        """n_owners/owner_id: an OWNER SHARE — the context holds 1/n_owners of the key space
        (shk_config.n_owners).  device_ids: a multi-device context (shk_config.n_devices).
        reserve_cus: compute units the context's kernels leave free (shk_config.reserve_cus; 0 = the
        library's default, RESERVE_NONE = none)."""
        self._L = load_library()
        self.k, self.chunks, self.histo_max = k, chunks, histo_max
        self.n_owners, self.owner_id = max(n_owners, 1), owner_id
        self._tdev = f"cuda:{device}"  # torch tensors over this context's memory live on ITS device, whatever torch's current one is
        cfg = _Config(k=k, chunks=chunks, histo_max=histo_max, device=device, flags=flags,
                      table_capacity_hint=capacity_hint, n_owners=n_owners, owner_id=owner_id, reserve_cus=reserve_cus)
        if device_ids is not None:
            ids = (C.c_int32 * len(device_ids))(*device_ids)
            cfg.n_devices = len(device_ids)
            cfg.device_ids = C.cast(ids, C.POINTER(C.c_int32))
        h = C.c_void_p()
        rc = self._L.shk_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise ShkError(rc, (self._L.shk_last_error(None) or b"").decode("utf-8", "replace"))
        self._h = h

    # -- plumbing ------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            raise ShkError(rc, (self._L.shk_last_error(self._h) or b"").decode("utf-8", "replace"))

    def close(self):
        if getattr(self, "_h", None):
            self._L.shk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @staticmethod
    def _pack(seqs):
        bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
        offsets = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            offsets[1:] = np.cumsum([len(b) for b in bs])
        bases = np.frombuffer(b"".join(bs), dtype=np.uint8) if bs else np.zeros(0, dtype=np.uint8)
        return bases, offsets

    # -- ingest --------------------------------------------------------------------------
    def ingest_batch(self, chunk_id: int, bases: np.ndarray, offsets: np.ndarray):
        """drain_batch body (io.rs:356-358): all sequences to `chunk_id`."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self._L.shk_ingest_batch(self._h, chunk_id, bases.ctypes.data, offsets.ctypes.data,
                                             len(offsets) - 1))

    def ingest_reads(self, bases: np.ndarray, offsets: np.ndarray):
        """read_fastq cadence (io.rs:335-343,355-361): read i → chunk (i//1000) % n_chunks."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self._L.shk_ingest_reads(self._h, bases.ctypes.data, offsets.ctypes.data,
                                             len(offsets) - 1))

    def ingest_packed(self, pk: "PackedReads"):
        """shk_ingest_packed: a batch packed by pack_reads (2-bit stream + N mask + base offsets)."""
        self._check(self._L.shk_ingest_packed(self._h, pk.packed.ctypes.data, pk.nmask.ctypes.data,
                                              pk.offsets.ctypes.data, len(pk.offsets) - 1))

    def ingest_packed_slice(self, pk: "PackedReads", first_seq: int, n_seqs: int):
        """shk_ingest_packed over reads [first_seq, first_seq + n_seqs) of a packed batch: the offsets are positions in
        the batch's streams, so a slice is the same two streams and a window of the offsets."""
        assert 0 <= first_seq and first_seq + n_seqs <= len(pk.offsets) - 1
        self._check(self._L.shk_ingest_packed(self._h, pk.packed.ctypes.data, pk.nmask.ctypes.data,
                                              pk.offsets.ctypes.data + 8 * first_seq, n_seqs))

    def ingest_packed_device(self, d_packed: int, d_nmask: int, d_offsets: int, n_seqs: int, n_bases: int):
        self._check(self._L.shk_ingest_packed_device(self._h, d_packed, d_nmask, d_offsets, n_seqs, n_bases))

    def pack_reads_device(self, d_bases: int, n_bases: int, d_packed: int, d_nmask: int):
        self._check(self._L.shk_pack_reads_device(self._h, d_bases, n_bases, d_packed, d_nmask))

    def unpack_reads_device(self, d_packed: int, d_nmask: int, n_bases: int, d_bases: int):
        self._check(self._L.shk_unpack_reads_device(self._h, d_packed, d_nmask, n_bases, d_bases))

    def ingest_seqs(self, seqs):
        """Convenience: a list of str/bytes sequences in input order."""
        self.ingest_reads(*self._pack(seqs))

    def ingest_seq(self, seq, chunk_id: int = 0):
        """Chunk::ingest_seq (chunk.rs:25-30) for one sequence (launch-bound; tests only)."""
        self.ingest_batch(chunk_id, *self._pack([seq]))

    def ingest_reads_device(self, d_bases: int, d_offsets: int, n_seqs: int, n_bases: int):
        self._check(self._L.shk_ingest_reads_device(self._h, d_bases, d_offsets, n_seqs, n_bases))

    def insert(self, kmers, counts, chunk_id: int = 0):
        """KmerCounts::insert (counting.rs:152-154)."""
        kmers = np.ascontiguousarray(np.atleast_1d(kmers), dtype=np.uint64)
        counts = np.ascontiguousarray(np.atleast_1d(counts), dtype=np.uint32)
        assert len(kmers) == len(counts)
        self._check(self._L.shk_insert_counts(self._h, chunk_id, kmers.ctypes.data, counts.ctypes.data,
                                              len(kmers)))

    def sync(self):
        self._check(self._L.shk_sync(self._h))

    def reset(self):
        """Empty chunks and zero counters, keeping allocations (a fresh FastqReadState)."""
        self._check(self._L.shk_reset(self._h))

    # -- consolidate -------------------------------------------------------------------
    def finalize(self):
        self._check(self._L.shk_finalize(self._h))
        return self

    def finalize_begin(self, user_word: int = 0):
        """shk_finalize_begin: queues the scan; returns the int64 CUDA tensor (a view of the engine's control block)
        a reduction over the ranks may sum in place — on the engine's stream."""
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._L.shk_finalize_begin(self._h, user_word, C.byref(p), C.byref(n)))
        return self._raw_tensor(p.value, int(n.value), "<i8", self._tdev)

    def finalize_end(self):
        """shk_finalize_end → (again, Σ user_word)."""
        again, usum = C.c_int(0), C.c_uint64(0)
        self._check(self._L.shk_finalize_end(self._h, C.byref(again), C.byref(usum)))
        return bool(again.value), int(usum.value)

    def histograms(self, out: np.ndarray | None = None) -> np.ndarray:
        """histo_vecs (io.rs:1020-1028): (chunks, histo_max+2) u64.  out: a caller-owned array of that shape to
        fill (a job of a few hundred µs notices the allocation of a fresh one)."""
        if out is None:
            out = np.empty((self.chunks, self.histo_max + 2), dtype=np.uint64)
        else:
            assert out.shape == (self.chunks, self.histo_max + 2) and out.dtype == np.uint64 and out.flags.c_contiguous
        self._check(self._L.shk_histograms(self._h, out.ctypes.data))
        return out

    def counters(self) -> dict:
        c = _Counters()
        self._check(self._L.shk_get_counters(self._h, C.byref(c)))
        return {f: int(getattr(c, f)) for f, _ in _Counters._fields_}

    def timings(self) -> dict:
        t = _Timings()
        self._check(self._L.shk_get_timings(self._h, C.byref(t)))
        return {KERNEL_NAMES[i]: (float(t.ms[i]), int(t.launches[i]))
                for i in range(len(KERNEL_NAMES)) if t.launches[i]}

    def reset_timings(self):
        self._check(self._L.shk_reset_timings(self._h))

    # -- merged-table read API ------------------------------------------------------
    def export_table(self):
        """KmerCounts::iter (counting.rs:239-241), sorted by k-mer for comparison."""
        n = C.c_uint64(0)
        self._check(self._L.shk_export_table(self._h, None, None, 0, C.byref(n)))
        cap = int(n.value)
        keys = np.zeros(cap, dtype=np.uint64)
        cnts = np.zeros(cap, dtype=np.uint32)
        if cap:
            self._check(self._L.shk_export_table(self._h, keys.ctypes.data, cnts.ctypes.data, cap,
                                                 C.byref(n)))
        o = np.argsort(keys, kind="stable")
        return keys[o], cnts[o]

    def lookup(self, kmers, canonical: bool = False) -> np.ndarray:
        """get_count (counting.rs:224-226) / get_canonical_count (:205-209)."""
        kmers = np.ascontiguousarray(np.atleast_1d(kmers), dtype=np.uint64)
        out = np.zeros(len(kmers), dtype=np.uint32)
        self._check(self._L.shk_lookup(self._h, kmers.ctypes.data, out.ctypes.data, len(kmers),
                                       1 if canonical else 0))
        return out

    def find_oligos(self, oligos, oligo_len: int, min_count: int = 1):
        """find_oligos_in_kmers (pcr/primers.rs:163-226): (kmers, counts) sorted by k-mer."""
        oligos = np.ascontiguousarray(np.atleast_1d(oligos), dtype=np.uint64)
        n = C.c_uint64(0)
        self._check(self._L.shk_find_oligos(self._h, oligos.ctypes.data, len(oligos), oligo_len, min_count,
                                            None, None, 0, C.byref(n)))
        cap = int(n.value)
        keys = np.zeros(cap, dtype=np.uint64)
        cnts = np.zeros(cap, dtype=np.uint32)
        if cap:
            self._check(self._L.shk_find_oligos(self._h, oligos.ctypes.data, len(oligos), oligo_len,
                                                min_count, keys.ctypes.data, cnts.ctypes.data, cap, C.byref(n)))
        o = np.argsort(keys, kind="stable")
        return keys[o], cnts[o]

    def filter_reads(self, bases: np.ndarray, offsets: np.ndarray, primer_kmers) -> np.ndarray:
        """PrimerReadFilter::matches per read (pcr/read_filter.rs:43-55) → bool array."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        pk = np.ascontiguousarray(np.atleast_1d(primer_kmers), dtype=np.uint64)
        n = len(offsets) - 1
        out = np.zeros(max(n, 1), dtype=np.uint8)
        self._check(self._L.shk_filter_reads(self._h, bases.ctypes.data, offsets.ctypes.data, n,
                                             pk.ctypes.data, len(pk), out.ctypes.data))
        return out[:n].astype(bool)

    def kmers_from_reads(self, bases: np.ndarray, offsets: np.ndarray):
        """Batched kmers_from_ascii (kmer/encoding.rs:332-371) in the form thread_reads uses it
        (pcr/threading.rs:97-101) → (list of per-read uint64 arrays, bad_byte uint8 array).  A read with
        a byte outside ACGTN has bad_byte != 0 and no k-mers."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        lens = np.diff(offsets.astype(np.int64))
        room = np.maximum(lens - self.k + 1, 0)
        koff = np.concatenate([[0], np.cumsum(room)]).astype(np.int64)
        kmers = np.empty(max(int(koff[-1]), 1), dtype=np.uint64)
        n_k = np.zeros(max(n, 1), dtype=np.uint32)
        bad = np.zeros(max(n, 1), dtype=np.uint8)
        self._check(self._L.shk_kmers_from_reads(self._h, bases.ctypes.data, offsets.ctypes.data, n,
                                                 kmers.ctypes.data, int(koff[-1]), n_k.ctypes.data,
                                                 bad.ctypes.data))
        return [kmers[koff[i]:koff[i] + int(n_k[i])] for i in range(n)], bad[:n]

    def kmers_from_ascii(self, seq) -> np.ndarray:
        """kmers_from_ascii(seq, k) (kmer/encoding.rs:332-371) for one sequence; raises with the
        reference's message on a byte outside ACGTN (encoding.rs:353-356)."""
        b = np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
        out, bad = self.kmers_from_reads(b, np.array([0, len(b)], dtype=np.uint64))
        if bad[0]:
            raise ShkError(-1, "Invalid character '%s' in sequence. Only ACGTN allowed." % chr(int(bad[0])))
        return out[0]

    # -- multi-GPU hooks ---------------------------------------------------------------
    def table_geometry(self):
        p, s, l = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
        self._check(self._L.shk_table_geometry(self._h, C.byref(p), C.byref(s), C.byref(l)))
        return int(p.value), int(s.value), int(l.value)

    def reserve_pages(self, n_pages: int):
        self._check(self._L.shk_table_reserve_pages(self._h, n_pages))

    def table_device_ptrs(self):
        k, v = C.c_void_p(), C.c_void_p()
        self._check(self._L.shk_table_device_ptrs(self._h, C.byref(k), C.byref(v)))
        return int(k.value), int(v.value)

    def merge_pages(self, p0: int, p1: int, d_keys: int, d_vals: int, lane_stride: int):
        self._check(self._L.shk_merge_pages(self._h, p0, p1, d_keys, d_vals, lane_stride))

    # -- exchange rounds between owner shares (include/shk.h, shk_xchg_*) ---------------------
    def stream(self) -> int:
        return int(self._L.shk_stream(self._h) or 0)

    def xchg_scatter_device(self, d_bases: int, d_offsets: int, n_seqs: int, n_bases: int, layout_bases: int = 0):
        """→ (d_records, d_cursors, XchgLayout, n_foreign_spilled)."""
        rec, cur, lay, nf = C.c_void_p(), C.c_void_p(), XchgLayout(), C.c_uint64(0)
        self._check(self._L.shk_xchg_scatter_device(self._h, d_bases, d_offsets, n_seqs, n_bases, layout_bases,
                                                    C.byref(rec), C.byref(cur), C.byref(lay), C.byref(nf)))
        return int(rec.value or 0), int(cur.value or 0), lay, int(nf.value)

    def xchg_scatter_begin(self, d_bases: int, d_offsets: int, n_seqs: int, n_bases: int, layout_bases: int = 0):
        """The scatter launched, not waited for → (d_records, d_cursors, XchgLayout); xchg_scatter_end() → n_foreign_spilled."""
        rec, cur, lay = C.c_void_p(), C.c_void_p(), XchgLayout()
        self._check(self._L.shk_xchg_scatter_begin(self._h, d_bases, d_offsets, n_seqs, n_bases, layout_bases,
                                                   C.byref(rec), C.byref(cur), C.byref(lay)))
        return int(rec.value or 0), int(cur.value or 0), lay

    def xchg_scatter_end(self) -> int:
        nf = C.c_uint64(0)
        self._check(self._L.shk_xchg_scatter_end(self._h, C.byref(nf)))
        return int(nf.value)

    def xchg_absorb(self, d_records: int, d_cursors: int, lay):
        self._check(self._L.shk_xchg_absorb(self._h, d_records, d_cursors, C.byref(lay)))

    def xchg_spill(self):
        """→ (d_kmers, d_lanes, d_counts, n): the foreign spill list."""
        k, l, c, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64(0)
        self._check(self._L.shk_xchg_spill(self._h, C.byref(k), C.byref(l), C.byref(c), C.byref(n)))
        return int(k.value or 0), int(l.value or 0), int(c.value or 0), int(n.value)

    def xchg_spill_clear(self):
        self._check(self._L.shk_xchg_spill_clear(self._h))

    def xchg_feasible(self) -> bool:
        """Does this share's key width fit the owner layout's 4-byte records (shk_xchg_scatter_device)?  If not
        (k > 21 at the default fan-out), rounds go through xchg_wide_scatter_device + insert_device."""
        return bool(self._L.shk_xchg_feasible(self._h))

    def xchg_wide_scatter_device(self, d_bases: int, d_offsets: int, n_seqs: int, n_bases: int):
        """→ (d_kmers, d_lanes, counts[n_owners]): the batch's k-mers as whole 64-bit values grouped by owner."""
        km, ln = C.c_void_p(), C.c_void_p()
        counts = (C.c_uint64 * 64)()
        self._check(self._L.shk_xchg_wide_scatter_device(self._h, d_bases, d_offsets, n_seqs, n_bases, C.byref(km), C.byref(ln), counts))
        return int(km.value or 0), int(ln.value or 0), [int(counts[i]) for i in range(max(self.n_owners, 1))]

    def insert_device(self, d_kmers: int, d_lanes: int, d_counts: int, n: int):
        self._check(self._L.shk_insert_device(self._h, d_kmers, d_lanes, d_counts, n))

    def set_read_index(self, next_read_index: int):
        """Global index of the next read (chunk striping of a sharded stream)."""
        self._check(self._L.shk_set_read_index(self._h, next_read_index))

    # -- torch views for the multi-GPU driver (sharkmer_amd/dist.py) ---------------------
    def table_tensors(self):
        """(keys int64[capacity], vals int32[n_lanes, capacity]) as zero-copy CUDA tensors over
        the live table (valid until the table grows)."""
        import torch
        n_pages, page_slots, n_lanes = self.table_geometry()
        cap = n_pages * page_slots
        dk, dv = self.table_device_ptrs()

        class _Raw:  # minimal __cuda_array_interface__ carrier
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr,
                                                 "data": (ptr, False), "version": 2}

        keys = torch.as_tensor(_Raw(dk, (cap,), "<i8"), device=self._tdev)
        vals = torch.as_tensor(_Raw(dv, (n_lanes, cap), "<i4"), device=self._tdev)
        return keys, vals

    def merge_page_tensors(self, p0: int, p1: int, keys_t, vals_t):
        """KmerCounts::extend of a peer's page range held in torch tensors
        (keys_t int64[(p1-p0)*page_slots], vals_t int32[n_lanes, same])."""
        assert keys_t.is_contiguous() and vals_t.stride(-1) == 1
        self.merge_pages(p0, p1, keys_t.data_ptr(), vals_t.data_ptr(),
                         vals_t.stride(0) if vals_t.dim() == 2 else keys_t.numel())

    def set_owned_pages(self, p0: int, p1: int):
        self._check(self._L.shk_set_owned_pages(self._h, p0, p1))

    def set_owner_share(self, n_owners: int, owner: int):
        self._check(self._L.shk_set_owner_share(self._h, n_owners, owner))

    def owner_counts(self, n_owners: int) -> np.ndarray:
        """Occupied slots in each of n_owners equal page ranges."""
        out = np.zeros(n_owners, dtype=np.uint64)
        self._check(self._L.shk_owner_counts(self._h, n_owners, out.ctypes.data))
        return out

    def compact_owner_tensors(self, counts, skip_owner: int = -1):
        """The occupied entries, owner after owner: (keys int64[n], vals int32[n_lanes, n]) as new
        CUDA tensors, n = sum(counts) (counts from owner_counts with the same number of owners;
        skip_owner's count must be 0: its range is left out)."""
        import torch
        counts = np.asarray(counts, dtype=np.uint64)
        n = int(counts.sum())
        _, _, n_lanes = self.table_geometry()
        keys = torch.empty(max(n, 1), dtype=torch.int64, device=self._tdev)
        vals = torch.empty((n_lanes, max(n, 1)), dtype=torch.int32, device=self._tdev)
        off = np.zeros(len(counts), dtype=np.uint64)
        off[1:] = np.cumsum(counts)[:-1]
        self._check(self._L.shk_compact_owners(self._h, len(counts), off.ctypes.data, keys.data_ptr(),
                                               vals.data_ptr(), vals.stride(0), skip_owner))
        return keys[:n], vals[:, :n]

    def compact_owner_packed(self, counts, skip_owner: int = -1):
        """Every owner's occupied entries as one self-contained piece of ONE int32 CUDA tensor (include/shk.h,
        shk_compact_owners_packed): piece o = [k-mers 2·c ints][lane 0 counts c ints]…, c = counts[o].
        Queued on the engine's stream."""
        import torch
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        _, _, n_lanes = self.table_geometry()
        n = int(counts.sum())
        buf = torch.empty(max(n * (2 + n_lanes), 1), dtype=torch.int32, device=self._tdev)
        self._check(self._L.shk_compact_owners_packed(self._h, len(counts), counts.ctypes.data, buf.data_ptr(), skip_owner))
        return buf[:n * (2 + n_lanes)], n_lanes

    def compact_owner_fixed(self, n_owners: int, capacity: int, skip_owner: int = -1):
        """The same with pieces of a fixed capacity at fixed places (include/shk.h, shk_compact_owners_fixed):
        piece o = [header 2 ints][k-mers 2·capacity][lane counts capacity each], unused places EMPTY."""
        import torch
        n_lanes = max(1, self.chunks)  # (not table_geometry(): that waits for the counting launches)
        buf = torch.empty(n_owners * (2 + capacity * (2 + n_lanes)), dtype=torch.int32, device=self._tdev)
        self._check(self._L.shk_compact_owners_fixed(self._h, n_owners, capacity, buf.data_ptr(), skip_owner))
        return buf, n_lanes

    def merge_fixed_pieces(self, buf_t, n_pieces: int, capacity: int, skip_piece: int = -1):
        """KmerCounts::extend of all received fixed-capacity pieces in one launch — or of none, if any sender had
        more entries than a piece holds (shk_merge_pieces)."""
        assert buf_t.is_contiguous()
        self._check(self._L.shk_merge_pieces(self._h, buf_t.data_ptr(), n_pieces, capacity, skip_piece))

    def merge_pieces_max(self) -> int:
        """Largest piece header the last merge_fixed_pieces saw (valid after finalize): > capacity ⇒ not merged."""
        v = C.c_uint64(0)
        self._check(self._L.shk_merge_pieces_max(self._h, C.byref(v)))
        return int(v.value)

    def merge_packed_piece(self, piece_t, c: int, n_lanes: int):
        """KmerCounts::extend of one received piece (c entries: k-mers, then each lane's counts)."""
        if c:
            assert piece_t.is_contiguous() and piece_t.numel() == c * (2 + n_lanes)
            p = piece_t.data_ptr()
            self._check(self._L.shk_merge_entries(self._h, p, p + 8 * c, c, c))

    def merge_entry_tensors(self, keys_t, vals_t):
        """KmerCounts::extend of received entries (keys_t int64[n], vals_t int32[n_lanes, n], unit stride)."""
        n = keys_t.numel()
        if n == 0:
            return
        assert keys_t.is_contiguous() and vals_t.stride(-1) == 1
        self._check(self._L.shk_merge_entries(self._h, keys_t.data_ptr(), vals_t.data_ptr(), n, vals_t.stride(0)))

    # -- exchange rounds as torch tensors (sharkmer_amd/dist.py: OwnerCounter) -----------------------
    @staticmethod
    def _raw_tensor(ptr: int, n: int, typestr: str, device: str = "cuda"):
        import torch

        class _Raw:  # minimal __cuda_array_interface__ carrier
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr,
                                                 "data": (ptr, False), "version": 2}
        if n == 0 or not ptr:
            return torch.empty(0, dtype={"<i4": torch.int32, "<i8": torch.int64, "|u1": torch.uint8}[typestr], device=device)
        return torch.as_tensor(_Raw(ptr, (n,), typestr), device=device)

    def xchg_scatter_tensors(self, d_bases: int, d_offsets: int, n_seqs: int, n_bases: int, layout_bases: int = 0):
        """One round's level-1 scatter → (records int32 or int64 [W·segment_records], cursors int32[W·regions],
        layout, n_foreign_spilled) as zero-copy views of the context's exchange buffer."""
        rec, cur, lay, nf = self.xchg_scatter_device(d_bases, d_offsets, n_seqs, n_bases, layout_bases)
        W = lay.n_owners
        # (a tensor element is one record: int32 for the 4-byte layout, int64 for the 8-byte one — layout.record_bytes)
        return (self._raw_tensor(rec, W * lay.segment_records, "<i8" if lay.record_bytes == 8 else "<i4", self._tdev),
                self._raw_tensor(cur, W * lay.regions, "<i4", self._tdev), lay, nf)

    def xchg_scatter_begin_tensors(self, d_bases: int, d_offsets: int, n_seqs: int, n_bases: int, layout_bases: int = 0):
        """xchg_scatter_tensors without the wait: (records, cursors, layout) — views that are complete once
        xchg_scatter_end() has returned (stream-ordered users on the engine's stream need not wait for that)."""
        rec, cur, lay = self.xchg_scatter_begin(d_bases, d_offsets, n_seqs, n_bases, layout_bases)
        W = lay.n_owners
        return (self._raw_tensor(rec, W * lay.segment_records, "<i8" if lay.record_bytes == 8 else "<i4", self._tdev),
                self._raw_tensor(cur, W * lay.regions, "<i4", self._tdev), lay)

    def xchg_absorb_tensors(self, rec_t, cur_t, lay):
        assert rec_t.is_contiguous() and cur_t.is_contiguous()
        assert rec_t.numel() == lay.segment_records and cur_t.numel() == lay.regions
        self.xchg_absorb(rec_t.data_ptr(), cur_t.data_ptr(), lay)

    def xchg_spill_tensors(self):
        """(kmers int64[n], lanes int32[n], counts int32[n]) views of the foreign spill list."""
        k, l, c, n = self.xchg_spill()
        return self._raw_tensor(k, n, "<i8", self._tdev), self._raw_tensor(l, n, "<i4", self._tdev), self._raw_tensor(c, n, "<i4", self._tdev)

    def insert_tensors(self, kmers_t, lanes_t, counts_t=None):
        """counts_t None: every record counts once."""
        n = kmers_t.numel()
        if n:
            assert kmers_t.is_contiguous() and lanes_t.is_contiguous() and (counts_t is None or counts_t.is_contiguous())
            self.insert_device(kmers_t.data_ptr(), lanes_t.data_ptr(), counts_t.data_ptr() if counts_t is not None else None, n)

    def xchg_wide_scatter_tensors(self, d_bases: int, d_offsets: int, n_seqs: int, n_bases: int):
        """→ (kmers int64[n], lanes int32[n], counts list[n_owners]) — views of the context's buffers, valid until the
        next scatter."""
        km, ln, counts = self.xchg_wide_scatter_device(d_bases, d_offsets, n_seqs, n_bases)
        n = sum(counts)
        return self._raw_tensor(km, n, "<i8", self._tdev), self._raw_tensor(ln, n, "<i4", self._tdev), counts

    # -- device memory + synthetic input ---------------------------------------------
    def alloc_device(self, nbytes: int) -> int:
        p = self._L.shk_alloc_device(self._h, nbytes)
        if not p:
            raise ShkError(-4, f"device allocation of {nbytes} bytes failed")
        return int(p)

    def free_device(self, p: int):
        self._L.shk_free_device(self._h, p)

    def synth_reads_device(self, spec, first_read: int, n_reads: int, d_bases: int, d_offsets: int):
        s = _Synth(seed_genome=spec.seed_genome, seed_reads=spec.seed_reads,
                   genome_len=spec.genome_len, read_len=spec.read_len,
                   sub_per_64k=spec.sub_per_64k, n_per_64k=spec.n_per_64k)
        self._check(self._L.shk_synth_reads_device(self._h, C.byref(s), first_read, n_reads,
                                                   d_bases, d_offsets))


# ---- 2-bit packed batches (include/shk.h: the reference's Read::from_str layout + an N mask) --------------

class PackedReads:
    """A batch as shk_pack_reads leaves it: packed u8[(n+3)//4] (4 bases per byte, first base on top),
    nmask u32[(n+31)//32] (bit p%32 of word p//32 ⇔ base p is N), offsets u64[n_seqs+1] in bases."""

    def __init__(self, packed, nmask, offsets, n_bases, _pinned=None):
        self.packed, self.nmask, self.offsets, self.n_bases = packed, nmask, offsets, n_bases
        self._pinned = _pinned  # (library, [pointers]) when the arrays are views over shk_alloc_pinned memory

    def close(self):
        """Give pinned arrays back (shk_free_pinned); the numpy views must not be used afterwards."""
        if self._pinned:
            L, ptrs = self._pinned
            self._pinned = None
            self.packed = self.nmask = self.offsets = None
            for q in ptrs:
                L.shk_free_pinned(q)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def nbytes(self):
        return self.packed.nbytes + self.nmask.nbytes + self.offsets.nbytes

    def wire_bytes(self) -> int:
        """What shk_ingest_packed moves over PCIe for this batch: the 2-bit stream, the offsets, and the N mask —
        as (index, word) pairs of its non-zero words when those are at most 1/16 of it (per slice in the engine;
        estimated here over the whole batch)."""
        nz = int(np.count_nonzero(self.nmask))
        mask = 8 * nz if nz <= self.nmask.size // 16 else self.nmask.nbytes
        return self.packed.nbytes + self.offsets.nbytes + mask


def release_cached_memory():
    """shk_release_cached_memory: device and pinned blocks the process keeps for the next context go back to the driver."""
    load_library().shk_release_cached_memory()


def pack_reads(bases: np.ndarray, offsets: np.ndarray, threads: int = 0, pinned: bool = False, out: "PackedReads" = None) -> PackedReads:
    """shk_pack_reads over a batch of concatenated ASCII reads (host, multi-threaded).  pinned: the
    output arrays live in pinned host memory (shk_alloc_pinned), owned by the PackedReads: close() — or dropping the
    object — gives them back.  out: a PackedReads of the same shape (a batch packed before) whose arrays are written
    again — a caller that streams batches pins its buffers once."""
    L = load_library() if (pinned or (out is not None and out._pinned)) else load_front_library()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = int(offsets[-1]) if len(offsets) else 0
    assert len(offsets) == 0 or int(offsets[0]) == 0
    nb, nw = (n + 3) // 4, (n + 31) // 32
    if out is not None:
        assert out.packed.size == nb and out.nmask.size == nw and out.offsets.size == len(offsets), "out: another batch shape"
        out.offsets[:] = offsets
        rc = L.shk_pack_reads(bases.ctypes.data, n, out.packed.ctypes.data, out.nmask.ctypes.data, threads)
        if rc != 0:
            raise ShkError(rc, (L.shk_run_error() or b"").decode("utf-8", "replace"))
        out.n_bases = n
        return out
    if pinned:
        pp = L.shk_alloc_pinned(max(nb, 1) + 16)
        pm = L.shk_alloc_pinned(max(nw, 1) * 4 + 16)
        po = L.shk_alloc_pinned(offsets.nbytes + 16)
        packed = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_uint8)), shape=(nb,))
        nmask = np.ctypeslib.as_array(C.cast(pm, C.POINTER(C.c_uint32)), shape=(nw,))
        offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(len(offsets),))
        offs[:] = offsets
    else:
        packed, nmask, offs = np.zeros(nb, dtype=np.uint8), np.zeros(nw, dtype=np.uint32), offsets
    rc = L.shk_pack_reads(bases.ctypes.data, n, packed.ctypes.data, nmask.ctypes.data, threads)
    if rc != 0:
        raise ShkError(rc, (L.shk_run_error() or b"").decode("utf-8", "replace"))
    return PackedReads(packed, nmask, offs, n, _pinned=(L, [pp, pm, po]) if pinned else None)


# ---- host side either side of the path: FASTQ front-end, writers, whole-run driver -----------------

class FastqReader:
    """read_fastq / open_fastq_reader (io.rs:271-352, 598-625) through libshk's C++ host.
    Parses only (no GPU needed)."""

    def __init__(self, paths, max_reads: int = 0, validate_every: int = 0, gzip_all_members: bool = False):
        """gzip_all_members: NOT the reference (a gzip file is read up to the end of its first member, io.rs:606-617) —
        every member (bgzip, concatenated files), shk_fastq_open_ex's SHK_FASTQ_GZIP_ALL_MEMBERS."""
        self._L = load_front_library()
        arr = (C.c_char_p * max(len(paths), 1))(*[os.fsencode(p) for p in paths])
        h = C.c_void_p()
        rc = self._L.shk_fastq_open_ex(arr, len(paths), max_reads, validate_every, FASTQ_GZIP_ALL_MEMBERS if gzip_all_members else 0, C.byref(h))
        if rc != 0:
            raise ShkError(rc, "shk_fastq_open failed")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.shk_fastq_close(self._h)
            self._h = None

    __del__ = close

    def next_batch(self, max_seqs: int = 1_000_000, max_bases: int = 64 << 20):
        bases = np.empty(max_bases, dtype=np.uint8)
        offsets = np.zeros(max_seqs + 1, dtype=np.uint64)
        n = C.c_uint64(0)
        rc = self._L.shk_fastq_next_batch(self._h, bases.ctypes.data, max_bases, offsets.ctypes.data,
                                          max_seqs, C.byref(n))
        if rc != 0:
            raise ShkError(rc, (self._L.shk_fastq_error(self._h) or b"").decode("utf-8", "replace"))
        n = int(n.value)
        return bases[:int(offsets[n])].copy(), offsets[:n + 1].copy()

    def next_batch_packed(self, max_seqs: int = 1_000_000, max_bases: int = 64 << 20) -> "PackedReads":
        """The next batch in the 2-bit packed input format (shk_fastq_next_batch_packed): what pack_reads would make
        of next_batch's output, without the ASCII batch in between."""
        packed = np.zeros(((max_bases + 3) // 4 + 7) // 8 * 8, dtype=np.uint8)
        nmask = np.zeros((max_bases + 31) // 32, dtype=np.uint32)
        offsets = np.zeros(max_seqs + 1, dtype=np.uint64)
        n = C.c_uint64(0)
        rc = self._L.shk_fastq_next_batch_packed(self._h, packed.ctypes.data, nmask.ctypes.data, max_bases,
                                                 offsets.ctypes.data, max_seqs, C.byref(n))
        if rc != 0:
            raise ShkError(rc, (self._L.shk_fastq_error(self._h) or b"").decode("utf-8", "replace"))
        n = int(n.value)
        nb = int(offsets[n])
        return PackedReads(packed[:(nb + 3) // 4].copy(), nmask[:(nb + 31) // 32].copy(), offsets[:n + 1].copy(), nb)

    def stats(self) -> dict:
        a, b, m, d = C.c_uint64(0), C.c_uint64(0), C.c_int(0), C.c_int(0)
        self._L.shk_fastq_stats(self._h, C.byref(a), C.byref(b), C.byref(m), C.byref(d))
        return {"n_reads_read": int(a.value), "n_bases_read": int(b.value),
                "reached_max": bool(m.value), "done": bool(d.value)}


def write_histo(path: str, histo: np.ndarray, k: int, histo_max: int, version: str = "3.1.0"):
    h = np.ascontiguousarray(histo, dtype=np.uint64)
    rc = load_front_library().shk_write_histo(os.fsencode(path), version.encode(), k, h.shape[0], histo_max,
                                        h.ctypes.data)
    if rc != 0:
        raise ShkError(rc, "shk_write_histo failed")


def write_final_histo(path: str, histo: np.ndarray, k: int, histo_max: int, version: str = "3.1.0"):
    h = np.ascontiguousarray(histo, dtype=np.uint64)
    rc = load_front_library().shk_write_final_histo(os.fsencode(path), version.encode(), k, h.shape[0],
                                              histo_max, h.ctypes.data)
    if rc != 0:
        raise ShkError(rc, "shk_write_final_histo failed")


def write_stats_yaml(path: str, **f):
    st = _RunStats(sharkmer_version=f.get("sharkmer_version", "3.1.0").encode(),
                   command=f.get("command", "").encode(), sample=f.get("sample", "").encode(),
                   kmer_length=f["kmer_length"], chunks=f["chunks"], n_reads_read=f["n_reads_read"],
                   n_bases_read=f["n_bases_read"], n_subreads_ingested=f["n_subreads_ingested"],
                   n_bases_ingested=f["n_bases_ingested"], n_kmers=f["n_kmers"],
                   n_multi_kmers=f.get("n_multi_kmers", 0), n_singleton_kmers=f.get("n_singleton_kmers", 0),
                   peak_memory_bytes=f.get("peak_memory_bytes", 0),
                   has_histogram=1 if f.get("has_histogram", f["chunks"] > 0) else 0)
    rc = load_front_library().shk_write_stats_yaml(os.fsencode(path), C.byref(st))
    if rc != 0:
        raise ShkError(rc, "shk_write_stats_yaml failed")


def validate_args(k: int, histo_max: int, sample):
    """cli.rs:659-673 + 645-652; raises ShkError with the reference's message."""
    L = load_front_library()
    rc = L.shk_validate_args(k, histo_max, None if sample is None else sample.encode())
    if rc != 0:
        raise ShkError(rc, (L.shk_run_error() or b"").decode("utf-8", "replace"))


def run_files(inputs, k: int, chunks: int, sample: str, outdir: str = "./", histo_max: int = 10000,
              max_reads: int = 0, validate_every: int = 0, device: int = 0, capacity_hint: int = 0,
              command: str = "", batch_reads: int = 0, batch_bases: int = 0, device_ids=None, gzip_all_members: bool = False) -> dict:
    """main.rs:112-197 without sPCR: FASTQ files → counts → .histo/.final.histo/.stats.yaml.
    device_ids: run on one multi-device context (shk_run_config.n_devices).  gzip_all_members: see FastqReader."""
    L = load_library()
    arr = (C.c_char_p * max(len(inputs), 1))(*[os.fsencode(p) for p in inputs])
    cfg = _RunConfig(inputs=arr, n_inputs=len(inputs), k=k, chunks=chunks, device=device,
                     histo_max=histo_max, max_reads=max_reads, validate_every=validate_every,
                     sample=None if sample is None else sample.encode(), outdir=os.fsencode(outdir),
                     command=command.encode(), version=None, table_capacity_hint=capacity_hint,
                     batch_reads=batch_reads, batch_bases=batch_bases, fastq_flags=FASTQ_GZIP_ALL_MEMBERS if gzip_all_members else 0)
    if device_ids is not None:
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        cfg.n_devices = len(device_ids)
        cfg.device_ids = C.cast(ids, C.POINTER(C.c_int32))
    st = _RunStats()
    rc = L.shk_run_files(C.byref(cfg), C.byref(st))
    if rc != 0:
        raise ShkError(rc, (L.shk_run_error() or b"").decode("utf-8", "replace"))
    return {f: int(getattr(st, f)) for f, t in _RunStats._fields_ if t in (C.c_uint32, C.c_uint64)}
