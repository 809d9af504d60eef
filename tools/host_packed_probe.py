"""Timeline probe of the packed host path (SHK_HOST_TRACE=1 prints per-slice host timestamps)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

L = 150
n = 4_000_000
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
with sa.KmerEngine(21, 1, 10000, capacity_hint=3_000_000, flags=sa.FLAG_TIMING if os.environ.get("TIMING") else 0) as eng:
    db = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    eng.synth_reads_device(spec, 0, n, db.data_ptr(), do.data_ptr())
    eng.sync()
    hb = torch.empty(n * L, dtype=torch.uint8, pin_memory=True)
    hb.copy_(db)
    ho_t = torch.empty(n + 1, dtype=torch.int64, pin_memory=True)
    ho_t.copy_(torch.arange(n + 1, dtype=torch.int64) * L)
    ho = ho_t.numpy().view(np.uint64)
    pk = sa.pack_reads(hb.numpy(), ho, pinned=True)
    for name, fn in (("pinned", lambda: eng.ingest_reads(hb.numpy(), ho)), ("packed", lambda: eng.ingest_packed(pk))):
        eng.reset()
        fn()
        eng.finalize()
        best = None
        for _ in range(3):
            eng.reset()
            eng.reset_timings()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            eng.finalize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            print(f"  {name}: ingest {1e3 * (t1 - t0):.3f} ms, finalize {1e3 * (dt - (t1 - t0)):.3f} ms", file=sys.stderr)
        print(name, round(n * L / best / 1e9, 2), "Gbases/s", round(best * 1e3, 3), "ms", eng.timings() if os.environ.get("TIMING") else "")
