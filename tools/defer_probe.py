#!/usr/bin/env python3
"""Small batches streamed into a large table (the usual situation: reads arrive in slices, the
table holds a genome): Gbases/s with deferred page passes (default) and with SHK_DEFER=0
(every launch on the direct-atomics path).  usage: defer_probe.py [genome] [batch_reads] [batches] [k]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

genome = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
n_batches = int(sys.argv[3]) if len(sys.argv) > 3 else 40
K = int(sys.argv[4]) if len(sys.argv) > 4 else 21
L = 150
spec = sa.SynthSpec(genome_len=genome, read_len=L)
eng = sa.KmerEngine(K, 1, 10000, capacity_hint=genome, flags=sa.FLAG_TIMING)
d_bases = torch.empty(batch * L, dtype=torch.uint8, device="cuda:0")
d_offsets = torch.empty(batch + 1, dtype=torch.int64, device="cuda:0")
bufs = []
for i in range(4):  # a few distinct batches, cycled (generation is not what is timed)
    b = torch.empty_like(d_bases)
    o = torch.empty_like(d_offsets)
    eng.synth_reads_device(spec, i * batch, batch, b.data_ptr(), o.data_ptr())
    bufs.append((b, o))
for rep in range(2):
    eng.reset()
    eng.reset_timings()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_batches):
        b, o = bufs[i % 4]
        eng.ingest_reads_device(b.data_ptr(), o.data_ptr(), batch, batch * L)
    eng.finalize()
    dt = time.perf_counter() - t0
t = eng.timings()
print(f"defer={os.environ.get('SHK_DEFER', '1')} k={K} genome={genome} batch={batch} x{n_batches}: "
      f"{n_batches * batch * L / dt / 1e9:.1f} Gbases/s  ({dt * 1e3:.1f} ms)  "
      f"{ {k: (round(v[0], 2), v[1]) for k, v in t.items()} }", flush=True)
