import sys, os, numpy as np
sys.path.insert(0, '.')
import torch
import sharkmer_amd as sa
L=150; n=1_000_000
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
for chunks in (1, 2, 10):
    eng = sa.KmerEngine(21, chunks, 10000, capacity_hint=3_000_000, flags=sa.FLAG_TIMING)
    d_b = torch.empty(n*L, dtype=torch.uint8, device="cuda:0"); d_o = torch.empty(n+1, dtype=torch.int64, device="cuda:0")
    eng.synth_reads_device(spec, 0, n, d_b.data_ptr(), d_o.data_ptr())
    for rep in range(3):
        eng.reset()
        eng.ingest_reads_device(d_b.data_ptr(), d_o.data_ptr(), n, n*L)
        eng.sync()
        c1 = eng.counters()["n_spilled"]
        eng.finalize()
        c2 = eng.counters()["n_spilled"]
        print(chunks, rep, "spilled after ingest+sync", c1, "after finalize", c2, os.environ.get("SHK_ALL_LANES"))
    eng.close()
