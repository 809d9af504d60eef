#!/usr/bin/env python3
"""Probe: does the chip have idle issue capacity while one counting job runs?  N independent
contexts (own streams, own tables) count their own batches from N host threads on ONE GPU;
compare the aggregate Gbases/s with a single context doing all the work."""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
import sharkmer_amd as sa  # noqa: E402

L, reads_total, steps = 150, 2_000_000, 6


def worker(n_reads, first, out, i, bar):
    spec = sa.SynthSpec(genome_len=3_000_000)
    eng = sa.KmerEngine(21, 1, 10000, capacity_hint=3_000_000)
    db = torch.empty(n_reads * L, dtype=torch.uint8, device="cuda")
    do = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
    eng.synth_reads_device(spec, first, n_reads, db.data_ptr(), do.data_ptr())
    for _ in range(2):
        eng.reset(); eng.ingest_reads_device(db.data_ptr(), do.data_ptr(), n_reads, n_reads * L); eng.finalize()
    bar.wait()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.reset(); eng.ingest_reads_device(db.data_ptr(), do.data_ptr(), n_reads, n_reads * L); eng.finalize()
    out[i] = time.perf_counter() - t0
    eng.close()


for n_ctx in (1, 2, 4):
    per = reads_total // n_ctx
    out = [0.0] * n_ctx
    bar = threading.Barrier(n_ctx)
    th = [threading.Thread(target=worker, args=(per, i * per, out, i, bar)) for i in range(n_ctx)]
    [t.start() for t in th]
    [t.join() for t in th]
    print(f"contexts={n_ctx} reads/ctx={per}: aggregate {reads_total * L * steps / max(out) / 1e9:.1f} Gbases/s", flush=True)
