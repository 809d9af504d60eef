#!/usr/bin/env python3
"""One owner's share of BASELINE.json configs[4] on one card, timed: 10 chunk lanes, a 3 Gb genome, owner 5 of 8
(2^30 slots × 48 B = 51 GB).  Every read is offered; the other owners' records are dropped in the level-1 pass
(the kernels of the 8-GPU exchange).  Prints one JSON line; tests/test_gpu_owner.py holds the exact checks."""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=125_000_000)
ap.add_argument("--batch", type=int, default=1_700_000)
ap.add_argument("--owners", type=int, default=8)
ap.add_argument("--chunks", type=int, default=10)
a = ap.parse_args()
L, k = 150, 21
spec = sa.SynthSpec(genome_len=3_000_000_000, read_len=L)
eng = sa.KmerEngine(k, a.chunks, 1000, capacity_hint=3_000_000_000 // a.owners, n_owners=a.owners, owner_id=5 % a.owners,
                    flags=sa.FLAG_TIMING)
nb = a.reads // a.batch
d = [torch.empty(a.batch * L, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
d_off = torch.empty(a.batch + 1, dtype=torch.int64, device="cuda:0")
for warm in (True, False):
    eng.reset()
    eng.reset_timings()
    torch.cuda.synchronize()
    t0 = time.time()
    for b in range(2 if warm else nb):
        eng.synth_reads_device(spec, b * a.batch, a.batch, d[b & 1].data_ptr(), d_off.data_ptr())
        eng.ingest_reads_device(d[b & 1].data_ptr(), d_off.data_ptr(), a.batch, a.batch * L)
    eng.finalize()
    eng.sync()
    dt = time.time() - t0
c = eng.counters()
tm = eng.timings()
print(json.dumps({"workload": f"owner share 5/{a.owners} of configs[4]: {nb * a.batch} reads offered (synthesised on the device inside the timed loop), "
                              f"{a.chunks} lanes, 2^30-slot table", "seconds": round(dt, 3),
                  "offered_gbases_per_s": round(nb * a.batch * L / dt / 1e9, 1),
                  "owned_kmers": c["n_kmers_ingested"], "owned_Gkmers_per_s": round(c["n_kmers_ingested"] / dt / 1e9, 2),
                  "n_unique": c["n_unique_kmers"], "n_spilled": c["n_spilled"],
                  "kernel_ms": {k_: round(v[0], 1) for k_, v in tm.items() if v[0] > 0}}))
eng.close()
