#!/usr/bin/env python3
"""Experiment: k_histo time only (finalize repeated on a populated table; invariants unchecked)."""
import ctypes as C, os, sys
import torch
import sharkmer_amd as sa
n_reads, L = 1_000_000, 150
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
eng = sa.KmerEngine(21, 1, 10000, device=0, capacity_hint=3_000_000, flags=sa.FLAG_TIMING)
d_bases = torch.empty(n_reads * L, dtype=torch.uint8, device="cuda:0")
d_offsets = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda:0")
eng.synth_reads_device(spec, 0, n_reads, d_bases.data_ptr(), d_offsets.data_ptr())
eng.ingest_reads_device(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_reads * L)
eng.sync()
eng.reset_timings()
for _ in range(10):
    try:
        eng.finalize()
    except Exception as e:
        pass
    # force a re-scan next time: a zero-read ingest clears the finalized flag
    eng.ingest_reads_device(d_bases.data_ptr(), d_offsets.data_ptr(), 0, 0)
t = eng.timings()
print(os.environ.get("SHK_LIB_PATH", "base"), os.environ.get("SHK_HISTO_G", ""), {k: (round(v[0] / max(v[1], 1), 4), v[1]) for k, v in t.items()}, flush=True)
