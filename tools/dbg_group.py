import sys, numpy as np
sys.path.insert(0, '.')
import sharkmer_amd as sa
from oracle import oracle as orc
orc.build()
spec = sa.SynthSpec(genome_len=70_000, sub_per_64k=250, n_per_64k=50)
for n, splits, hint, chunks, devs in [(23456, [1700, 9999, 17000], 0, 10, None), (23456, [9999], 0, 10, None), (23456, [9999], 0, 10, [0,0]), (23456, [9999, 17000], 0, 10, [0,0]),(23456, [1700, 9999], 0, 10, [0,0]),(23456, [1700, 9999, 17000], 0, 10, [0,0]), (23456, [1700], 0, 10, [0,0]), (23456, [2000], 0, 10, [0,0]), (23456, [1700], 0, 2, [0,0]), (5000, [1700], 0, 2, [0,0]),(23456, [], 0, 10, [0,0]), (23456, [], 4_200_000, 10, [0,0]), ]:
    bases, offsets = sa.synth_reads(spec, 0, n)
    ref = orc.run_batch(bases, offsets, 21, chunks, 300)
    with sa.KmerEngine(21, chunks, 300, capacity_hint=hint, device_ids=devs) as eng:
        cuts = [0] + splits + [n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            eng.ingest_reads(bases, offsets[a:b + 1])
        eng.finalize()
        got = eng.histograms(); c = eng.counters()
    want = ref.histograms()
    print(n, splits, hint, chunks, "equal" if np.array_equal(got, want) else "DIFF", c["n_kmers_ingested"], ref.stats["n_kmers_ingested"], c["n_unique_kmers"], ref.stats["n_unique_kmers"], c["n_spilled"])
    if not np.array_equal(got, want):
        d = np.argwhere(got != want)
        print("  first diffs", d[:5].tolist(), [(int(got[i,j]), int(want[i,j])) for i,j in d[:5]])
