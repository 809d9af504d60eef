#!/usr/bin/env python3
"""One-off assurance run (not part of the suite): full config-2-size batches (1 M reads, with
substitutions and N) at other k / chunk counts, histograms + counters + exported table against the
CPU oracle bit for bit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402
from oracle import oracle as orc  # noqa: E402

fails = 0
for k, chunks, hint, genome in ((31, 3, 3_000_000, 3_000_000), (21, 10, 3_000_000, 3_000_000),
                                (27, 2, 0, 3_000_000), (31, 1, 30_000_000, 30_000_000), (17, 4, 0, 500_000)):
    spec = sa.SynthSpec(genome_len=genome, sub_per_64k=300, n_per_64k=60)
    n = 1_000_000
    bases, offsets = sa.synth_reads(spec, 0, n)
    t0 = time.time()
    ref = orc.run_batch(bases, offsets, k, chunks, 10000)
    t1 = time.time()
    with sa.KmerEngine(k, chunks, 10000, capacity_hint=hint) as eng:
        for a in range(0, n, 250_000):   # four host batches
            b = a + 250_000
            eng.ingest_reads(bases[int(offsets[a]):int(offsets[b])], offsets[a:b + 1] - offsets[a])
        eng.finalize()
        ok = np.array_equal(eng.histograms(), ref.histograms())
        c = eng.counters()
        st = ref.stats
        ok &= all(c[x] == st[x] for x in ("n_kmers_ingested", "n_unique_kmers", "n_bases_ingested", "n_reads_ingested"))
        gk, gc = eng.export_table()
        rk, rc = ref.merged().export()
        ok &= np.array_equal(gk, rk) and np.array_equal(gc, rc)
    print(f"k={k} chunks={chunks} hint={hint} genome={genome}: {'bit-exact' if ok else 'MISMATCH'} "
          f"(oracle {t1 - t0:.1f} s, {c['n_unique_kmers']} distinct, grows {c['n_grows']}, spilled {c['n_spilled']})", flush=True)
    fails += not ok
sys.exit(1 if fails else 0)
