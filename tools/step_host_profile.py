"""Host-side time of each call of the bench step (config 2): where the GPU waits for the host."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

n, L = 1_000_000, 150
eng = sa.KmerEngine(21, 1, 10000, capacity_hint=3_000_000)
db = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
do = torch.empty(n + 1, dtype=torch.int64, device="cuda:0")
eng.synth_reads_device(sa.SynthSpec(genome_len=3_000_000, read_len=L), 0, n, db.data_ptr(), do.data_ptr())
eng.sync()
out = np.empty((1, 10002), dtype=np.uint64)
pb, po = db.data_ptr(), do.data_ptr()
acc = {k: 0.0 for k in ("reset", "set_index", "ingest", "finalize", "histograms")}
pc = time.perf_counter
for it in range(600):
    t0 = pc(); eng.reset()
    t1 = pc(); eng.set_read_index(0)
    t2 = pc(); eng.ingest_reads_device(pb, po, n, n * L)
    t3 = pc(); eng.finalize()
    t4 = pc(); eng.histograms(out)
    t5 = pc()
    if it >= 100:
        for k, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[k] += d
tot = sum(acc.values())
print({k: round(v / 500 * 1e6, 1) for k, v in acc.items()}, "us per call; step", round(tot / 500 * 1e6, 1), "us")
