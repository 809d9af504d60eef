"""sharkmer_amd — MI355X-native incremental k-mer counting engine.

Drop-in for the counting hot path of caseywdunn/sharkmer (src/kmer/* and the
ingest/consolidate loop of src/io.rs).  The product is the C-ABI shared library
built from sharkmer_amd/csrc (include/shk.h); this package is the thin ctypes
host mirror used by tests, bench.py and the multi-GPU driver.  There is no CPU
fallback: importing works anywhere, creating an engine needs a gfx950 GPU.
"""
from .engine import (  # noqa: F401
    KmerEngine, ShkError, N_READS_PER_BATCH, lib_path, load_library,
    FLAG_TIMING, FLAG_FORCE_DIRECT, FLAG_FORCE_PAGED, FLAG_DEFER_ERRORS, FLAG_TIMING_SAMPLED, KERNEL_NAMES, RESERVE_NONE,
    PackedReads, pack_reads, release_cached_memory, FastqReader, write_histo, write_final_histo, write_stats_yaml, validate_args, run_files,
)
from .synth import SynthSpec, synth_reads  # noqa: F401

__version__ = "0.1.0"
SHARKMER_VERSION = "3.1.0"  # reference version whose output formats we emit
