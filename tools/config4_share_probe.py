"""One GPU's share of BASELINE configs[3] WITHOUT the exchange (a whole-key-space context on a 2^33-slot table):
reads resident in HBM, batches of 1.7 M reads, per-kernel times.  SHK_RS32_BIG=0/1 pins the level-2 tile shape.
Usage: python tools/config4_share_probe.py [n_reads] [hooks: NAME=V,NAME=V;NAME=V ...] [capacity_hint]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000_000
variants = sys.argv[2].split(";") if len(sys.argv) > 2 else [""]
hint = int(sys.argv[3]) if len(sys.argv) > 3 else 3_000_000_000
L, k, batch = 150, 21, 1_700_000
spec = sa.SynthSpec(genome_len=3_000_000_000, read_len=L)
d_all = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
d_off = torch.empty(batch + 1, dtype=torch.int64, device="cuda:0")
nb = -(-n // batch)
for v in variants:
    for kv in filter(None, v.split(",")):
        a, b = kv.split("=")
        os.environ[a] = b
    with sa.KmerEngine(k, 1, 10000, device=0, capacity_hint=hint, flags=sa.FLAG_TIMING) as eng:
        for b in range(nb):
            m = min(batch, n - b * batch)
            eng.synth_reads_device(spec, b * batch, m, d_all.data_ptr() + b * batch * L, d_off.data_ptr())
        eng.sync()
        for rep in range(2):
            eng.reset()
            eng.reset_timings()
            t0 = time.perf_counter()
            for b in range(nb):
                m = min(batch, n - b * batch)
                eng.ingest_reads_device(d_all.data_ptr() + b * batch * L, d_off.data_ptr(), m, m * L)
            eng.finalize()
            dt = time.perf_counter() - t0
        ms = {k_: round(v[0], 2) for k_, v in eng.timings().items()}
        c = eng.counters()
        print(v or "default", "hint", hint, round(n * L / dt / 1e9, 2), "Gbases/s", round(dt * 1e3, 1), "ms", ms, "unique", c["n_unique_kmers"], flush=True)
    for kv in filter(None, v.split(",")):
        os.environ.pop(kv.split("=")[0], None)
