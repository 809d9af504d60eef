#!/usr/bin/env python3
"""The chunk-lane / larger-table shapes of bench.py's extras, one at a time (for rocprofv3 and quick A/B runs):
    python tools/shapes_run.py --genome 30000000 --reads 1700000 --chunks 10 --steps 10
step = reset + count + histogram emit, input resident in HBM.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--genome", type=int, default=30_000_000)
ap.add_argument("--reads", type=int, default=1_700_000)
ap.add_argument("--chunks", type=int, default=10)
ap.add_argument("--k", type=int, default=21)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
L = 150
spec = sa.SynthSpec(genome_len=a.genome, read_len=L)
eng = sa.KmerEngine(a.k, a.chunks, 10000, device=0, capacity_hint=a.genome, flags=sa.FLAG_TIMING)
db = torch.empty(a.reads * L, dtype=torch.uint8, device="cuda:0")
do = torch.empty(a.reads + 1, dtype=torch.int64, device="cuda:0")
eng.synth_reads_device(spec, 0, a.reads, db.data_ptr(), do.data_ptr())
eng.sync()


def one():
    eng.reset()
    eng.ingest_reads_device(db.data_ptr(), do.data_ptr(), a.reads, a.reads * L)
    if os.environ.get("SHK_EXP_IGNORE_FINALIZE"):  # kernel experiments that break the histogram on purpose: timing only
        try:
            eng.finalize()
        except Exception:
            pass
    else:
        eng.finalize()


for _ in range(3):
    one()
eng.reset_timings()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    one()
dt = time.perf_counter() - t0
tim = eng.timings()
c = eng.counters()
print(json.dumps({"workload": f"{a.reads} reads, {a.genome} bp genome, {a.chunks} chunk lane(s), k={a.k}",
                  "Gbases_per_s": round(a.reads * L * a.steps / dt / 1e9, 2), "ms_per_step": round(dt / a.steps * 1e3, 4),
                  "kernels_ms_per_step": {k: round(ms / a.steps, 4) for k, (ms, n) in tim.items() if n},
                  "n_unique": c["n_unique_kmers"], "table_capacity": c["table_capacity"], "n_spilled": c["n_spilled"]}))
