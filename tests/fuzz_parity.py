#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite): random k, chunk count, read shapes,
N / substitution rates, table hints, path flags and env hooks, every case checked bit for bit
against the CPU oracle (histograms, counters, exported table).  usage: fuzz_parity.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
for case in range(n_cases):
    k = int(rng.integers(1, 32))
    chunks = int(rng.choice([0, 1, 1, 2, 3, 7, 16]))
    genome = int(rng.choice([300, 5_000, 80_000, 1_000_000]))
    n_reads = int(rng.choice([1, 37, 999, 1000, 1001, 5_000, 20_000, 60_000]))
    read_len = int(rng.choice([max(k - 1, 1), k, k + 1, 50, 150, 301]))
    read_len = min(read_len, genome)
    spec = sa.SynthSpec(genome_len=genome, read_len=read_len, sub_per_64k=int(rng.choice([0, 100, 2000])),
                        n_per_64k=int(rng.choice([0, 50, 3000])))
    bases, offsets = sa.synth_reads(spec, int(rng.integers(0, 1000)), n_reads)
    flags = int(rng.choice([0, 0, sa.FLAG_FORCE_DIRECT, sa.FLAG_FORCE_PAGED, sa.FLAG_FORCE_PAGED]))
    hint = int(rng.choice([0, 0, 1_000, 100_000, 3_000_000]))
    env = {}
    if rng.random() < 0.25:
        env["SHK_REC32"] = "0"
    if rng.random() < 0.25:
        env["SHK_SCATTER32_LDS"] = "0"
    if rng.random() < 0.3:
        env["SHK_TWO_LEVEL_MIN_PAGES"] = "4"
        env["SHK_LEVEL1_LOG"] = str(int(rng.integers(0, 4)))
    if rng.random() < 0.2:
        env["SHK_FLUSH_GROUP_PAGES"] = str(int(rng.integers(1, 9)))  # deferred flushes in groups of a few pages
    if rng.random() < 0.2:
        env["SHK_DEFER_BUDGET"] = str(int(rng.choice([20_000, 300_000])))
    for key in ("SHK_REC32", "SHK_SCATTER32_LDS", "SHK_TWO_LEVEL_MIN_PAGES", "SHK_LEVEL1_LOG", "SHK_FLUSH_GROUP_PAGES", "SHK_DEFER_BUDGET"):
        os.environ.pop(key, None)
    os.environ.update(env)
    histo_max = int(rng.choice([1, 10, 255, 10_000]))
    n_split = int(rng.choice([1, 1, 2, 5]))
    cuts = sorted(set([0, n_reads] + [int(x) for x in rng.integers(0, n_reads + 1, size=n_split - 1)]))
    desc = f"case {case}: k={k} chunks={chunks} genome={genome} reads={n_reads}x{read_len} flags={flags} hint={hint} " \
           f"histo_max={histo_max} cuts={cuts} env={env}"
    try:
        ref = orc.run_batch(bases, offsets, k, chunks, histo_max)
        with sa.KmerEngine(k, chunks, histo_max, capacity_hint=hint, flags=flags) as eng:
            if rng.random() < 0.5:  # a job before this one: the table's memory holds its leftovers after the reset
                ob, oo = sa.synth_reads(sa.SynthSpec(genome_len=max(genome // 2, read_len), read_len=read_len, sub_per_64k=500),
                                        7, max(n_reads // 2, 1))
                eng.ingest_reads(ob, oo)
                if rng.random() < 0.5:
                    eng.finalize()
                eng.reset()
                desc += " after-reset"
            for a, b in zip(cuts[:-1], cuts[1:]):
                eng.ingest_reads(bases[int(offsets[a]):int(offsets[b])], offsets[a:b + 1] - offsets[a])
            eng.finalize()
            ok = np.array_equal(eng.histograms(), ref.histograms()) if chunks else True
            c = eng.counters()
            st = ref.stats
            ok &= all(c[x] == st[x] for x in ("n_kmers_ingested", "n_unique_kmers", "n_bases_ingested", "n_reads_ingested"))
            gk, gc = eng.export_table()
            rk, rc = ref.merged().export()
            ok &= np.array_equal(gk, rk) and np.array_equal(gc, rc)
    except Exception as e:  # noqa: BLE001
        ok = False
        desc += f" EXC {e!r}"
    if not ok:
        fails += 1
        print("FAIL", desc, flush=True)
    elif case % 10 == 0:
        print("ok  ", desc, flush=True)
print(f"{n_cases - fails}/{n_cases} cases bit-exact")
sys.exit(1 if fails else 0)
