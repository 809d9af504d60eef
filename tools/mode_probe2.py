#!/usr/bin/env python3
"""Which warm-up makes the FIRST engine of a process run in the fast mode?  arg: none | torchbig |
engine_idle | engine_run | hipbig"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "none"
n_reads, L = 1_000_000, 150
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
d_bases = torch.empty(n_reads * L, dtype=torch.uint8, device="cuda:0")
d_offsets = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda:0")


def run(eng, steps):
    for _ in range(steps):
        eng.reset()
        eng.ingest_reads_device(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_reads * L)
        eng.finalize()


if mode == "torchbig":
    x = torch.empty(6 << 30, dtype=torch.uint8, device="cuda:0")
    x.fill_(1)
    torch.cuda.synchronize()
    del x
    torch.cuda.empty_cache()
if mode in ("engine_idle", "engine_run", "hipbig"):
    e0 = sa.KmerEngine(21, 1, 10000, device=0, capacity_hint=3_000_000)
    if mode == "engine_run":
        e0.synth_reads_device(spec, 0, n_reads, d_bases.data_ptr(), d_offsets.data_ptr())
        run(e0, 2)
    if mode == "hipbig":
        p = e0.alloc_device(6 << 30)
        e0.free_device(p)
    e0.close()
eng = sa.KmerEngine(21, 1, 10000, device=0, capacity_hint=3_000_000, flags=sa.FLAG_TIMING)
eng.synth_reads_device(spec, 0, n_reads, d_bases.data_ptr(), d_offsets.data_ptr())
run(eng, 3)
eng.reset_timings()
run(eng, 10)
t = eng.timings()
print(mode, {k: round(v[0] / 10, 4) for k, v in t.items()}, flush=True)
eng.close()
