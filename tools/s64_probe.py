#!/usr/bin/env python3
"""The k = 31 path's kernels over repeated passes of the same batches (config 3's geometry: 300 Mb genome, 2^30 slots):
per-kernel milliseconds per pass, minimum and median over the repetitions.  For same-box A/Bs of libshk builds
(SHK_LIB_PATH) — single passes of tools/config3_run.py differ by ±10 % in the scatter's time."""
import argparse, json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=24_000_000)
ap.add_argument("--batch", type=int, default=4_000_000)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--genome", type=int, default=300_000_000)
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--tag", default="")
a = ap.parse_args()
L = 150
spec = sa.SynthSpec(genome_len=a.genome, read_len=L)
nb = a.reads // a.batch
d_all = torch.empty(a.reads * L, dtype=torch.uint8, device="cuda:0")
d_off = torch.empty(a.batch + 1, dtype=torch.int64, device="cuda:0")
eng = sa.KmerEngine(a.k, 1, 10000, device=0, capacity_hint=a.genome, flags=sa.FLAG_TIMING)
for b in range(nb):
    eng.synth_reads_device(spec, b * a.batch, a.batch, d_all.data_ptr() + b * a.batch * L, d_off.data_ptr())
eng.sync()
rows = []
for rep in range(a.reps + 1):
    eng.reset()
    eng.reset_timings()
    for b in range(nb):
        eng.ingest_reads_device(d_all.data_ptr() + b * a.batch * L, d_off.data_ptr(), a.batch, a.batch * L)
    eng.finalize()
    eng.sync()
    if rep:  # (the first pass allocates)
        rows.append({k: v[0] for k, v in eng.timings().items() if v[0] > 0})
out = {"tag": a.tag, "reads": a.reads, "k": a.k}
for key in rows[0]:
    vals = [r[key] for r in rows]
    out[key] = {"min": round(min(vals), 2), "median": round(statistics.median(vals), 2)}
print(json.dumps(out), flush=True)
eng.close()
