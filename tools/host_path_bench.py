#!/usr/bin/env python3
"""PCIe-inclusive rate of the counting path: the same config-2 batch, but handed over as HOST
buffers (pinned and pageable) through shk_ingest_reads — the number DESIGN.md §5 quotes beside
the HBM-resident headline.  Not the bench metric."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402
from sharkmer_amd.engine import load_library  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
spec = sa.SynthSpec(genome_len=3_000_000)
bases, offsets = sa.synth_reads(spec, 0, n_reads)
L = load_library()
out = {}
for kind in ("pageable", "pinned"):
    if kind == "pinned":
        pb = L.shk_alloc_pinned(len(bases))
        po = L.shk_alloc_pinned(len(offsets) * 8)
        C.memmove(pb, bases.ctypes.data, len(bases))
        C.memmove(po, offsets.ctypes.data, len(offsets) * 8)
        b = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), shape=(len(bases),))
        o = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(len(offsets),))
    else:
        b, o = bases, offsets
    with sa.KmerEngine(21, 1, 10000, capacity_hint=3_000_000) as eng:
        eng.ingest_reads(b, o)  # warm-up (allocations)
        eng.finalize()
        ts = []
        for _ in range(3):
            eng.reset()
            t0 = time.perf_counter()
            eng.ingest_reads(b, o)
            eng.finalize()
            ts.append(time.perf_counter() - t0)
        out[kind] = round(len(bases) / min(ts) / 1e9, 2)
print(json.dumps({"reads": n_reads, "host_to_histogram_Gbases_per_s": out}))
