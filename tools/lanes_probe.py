#!/usr/bin/env python3
"""Chunk-lane counts beyond 16 on one GPU (experiment helper): config 2's batch with 40 / 64 / 100 lanes, the paged
path's lane limit as set by SHK_PAGED_MAX_LANES (default 16: more lanes take the direct path), checked against the
oracle at a small size first."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa
from oracle import oracle as orc
import torch
L = 150
spec = sa.SynthSpec(genome_len=200_000, sub_per_64k=200, n_per_64k=40)
hb, ho = sa.synth_reads(spec, 0, 130_000)
for lanes in (40, 64, 100):
    ref = orc.run_batch(hb, ho, 21, lanes, 200)
    with sa.KmerEngine(21, lanes, 200, capacity_hint=200_000) as eng:
        eng.ingest_reads(hb, ho)
        eng.finalize()
        ok = np.array_equal(eng.histograms(), ref.histograms())
        print(f"parity lanes={lanes} SHK_PAGED_MAX_LANES={os.environ.get('SHK_PAGED_MAX_LANES','16')}: {'equal' if ok else 'DIFFERENT'}", flush=True)
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
n = 1_000_000
for lanes in (10, 40, 64, 100):
    with sa.KmerEngine(21, lanes, 10000, capacity_hint=3_000_000, flags=sa.FLAG_TIMING) as eng:
        db = torch.empty(n * L, dtype=torch.uint8, device="cuda:0"); do = torch.empty(n + 1, dtype=torch.int64, device="cuda:0")
        eng.synth_reads_device(spec, 0, n, db.data_ptr(), do.data_ptr())
        def one():
            eng.reset(); eng.ingest_reads_device(db.data_ptr(), do.data_ptr(), n, n * L); eng.finalize()
        for _ in range(2): one()
        eng.reset_timings(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): one()
        dt = (time.perf_counter() - t0) / 5
        print(f"lanes={lanes}: {n*L/dt/1e9:.1f} Gbases/s, {dt*1e3:.2f} ms/step, kernels {({k: round(v[0]/5,3) for k,v in eng.timings().items() if v[0]>0})}", flush=True)
