"""shk_ingest_reads at the reference's own cadence — 1000 reads a call (io.rs:340-343) — and larger: Gbases/s."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa

spec = sa.SynthSpec(genome_len=3_000_000)
n = 2_000_000
bases, offsets = sa.synth_reads(spec, 0, n)
for per_call in (1000, 10_000, 100_000, 1_000_000):
    with sa.KmerEngine(21, 10, 10000, capacity_hint=3_000_000) as eng:
        eng.ingest_reads(bases, offsets[:per_call + 1])   # (warm)
        eng.finalize()
        eng.reset()
        t0 = time.perf_counter()
        for a in range(0, n, per_call):
            eng.ingest_reads(bases, offsets[a:a + per_call + 1])
        eng.finalize()
        dt = time.perf_counter() - t0
    print(per_call, "reads per call:", round(n * 150 / dt / 1e9, 2), "Gbases/s", round(dt / (n / per_call) * 1e6, 1), "us per call")
