#!/usr/bin/env python3
"""Experiment: k_pages time on an empty table (every first occurrence misses) vs on a table that
already holds every key (second pass over the same reads, no reset).  Shows what miss handling costs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

n_reads, L = 1_000_000, 150
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
eng = sa.KmerEngine(21, 1, 10000, device=0, capacity_hint=3_000_000, flags=sa.FLAG_TIMING)
d_bases = torch.empty(n_reads * L, dtype=torch.uint8, device="cuda:0")
d_offsets = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda:0")
eng.synth_reads_device(spec, 0, n_reads, d_bases.data_ptr(), d_offsets.data_ptr())
for tag, do_reset in (("cold (reset each pass)", True), ("warm (keys present)", False)):
    eng.reset()
    eng.ingest_reads_device(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_reads * L)
    eng.sync()
    eng.reset_timings()
    for _ in range(10):
        if do_reset:
            eng.reset()
        eng.ingest_reads_device(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_reads * L)
    eng.sync()
    t = eng.timings()
    print(tag, {k: round(v[0] / 10, 4) for k, v in t.items()}, flush=True)
