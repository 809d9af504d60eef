#!/usr/bin/env python3
"""Does the scatter kernel's fast / slow mode (±4 % from process to process) follow the PROCESS or the
ALLOCATIONS?  One process, several engines created and destroyed in turn, same input buffer."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

n_reads, L = 1_000_000, 150
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
realloc_input = len(sys.argv) > 1 and sys.argv[1] == "input"
d_bases = torch.empty(n_reads * L, dtype=torch.uint8, device="cuda:0")
d_offsets = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda:0")
keep = []
for rep in range(8):
    eng = sa.KmerEngine(21, 1, 10000, device=0, capacity_hint=3_000_000, flags=sa.FLAG_TIMING)
    if realloc_input:
        keep.append((d_bases, d_offsets))  # hold the old ones so that the new ones land elsewhere
        d_bases = torch.empty(n_reads * L, dtype=torch.uint8, device="cuda:0")
        d_offsets = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda:0")
    eng.synth_reads_device(spec, 0, n_reads, d_bases.data_ptr(), d_offsets.data_ptr())
    for _ in range(3):
        eng.reset()
        eng.ingest_reads_device(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_reads * L)
        eng.finalize()
    eng.reset_timings()
    for _ in range(10):
        eng.reset()
        eng.ingest_reads_device(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_reads * L)
        eng.finalize()
    t = eng.timings()
    print(rep, {k: round(v[0] / 10, 4) for k, v in t.items()}, hex(d_bases.data_ptr()), flush=True)
    eng.close()
