"""BASELINE configs[2] streamed from pinned HOST memory (the first 24 M of its 100 M reads; k = 31, 300 Mb genome):
ASCII and 2-bit packed, with and without SHK_FLAG_DEFER_ERRORS.  What bench.py's extras.config3 reports, alone, so
that SHK_HOST_TRACE=1 shows where the link waits.  Usage: python tools/config3_host_probe.py [n_reads] [batch]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

L, k = 150, 31
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
spec = sa.SynthSpec(genome_len=300_000_000, read_len=L)
out = {}
for flags, tag in ((0, ""), (sa.FLAG_DEFER_ERRORS, "_defer")):
    with sa.KmerEngine(k, 1, 10000, device=0, capacity_hint=300_000_000, flags=flags) as eng:
        d_all = torch.empty(nh * L, dtype=torch.uint8, device="cuda:0")
        d_off = torch.empty(batch + 1, dtype=torch.int64, device="cuda:0")
        for b in range(nh // batch):
            eng.synth_reads_device(spec, b * batch, batch, d_all.data_ptr() + b * batch * L, d_off.data_ptr())
        eng.sync()
        hb = torch.empty(nh * L, dtype=torch.uint8, pin_memory=True)
        hb.copy_(d_all)
        del d_all
        torch.cuda.empty_cache()
        ho = np.arange(batch + 1, dtype=np.uint64) * np.uint64(L)
        # from HBM would be: see bench.py; here the host paths only
        for rep in range(2):
            eng.reset()
            marks = []
            t0 = time.perf_counter()
            for b in range(nh // batch):
                eng.ingest_reads(hb.numpy()[b * batch * L:(b + 1) * batch * L], ho)
                marks.append(time.perf_counter())
            eng.finalize()
            dt = time.perf_counter() - t0
        out["ascii_calls_ms" + tag] = [round((m - a) * 1e3, 1) for a, m in zip([t0] + marks, marks + [t0 + dt])]
        c_ascii = eng.counters()
        out["ascii" + tag] = round(nh * L / dt / 1e9, 2)
        ho_all = np.arange(nh + 1, dtype=np.uint64) * np.uint64(L)
        t_p = time.perf_counter()
        pk = sa.pack_reads(hb.numpy(), ho_all, pinned=True)
        t_p = time.perf_counter() - t_p
        out["host_pack_Gbases_per_s"] = round(nh * L / t_p / 1e9, 2)
        for rep in range(2):
            eng.reset()
            marks = []
            t0 = time.perf_counter()
            for b in range(nh // batch):
                eng.ingest_packed_slice(pk, b * batch, batch)
                marks.append(time.perf_counter())
            eng.finalize()
            dt = time.perf_counter() - t0
        out["packed_calls_ms" + tag] = [round((m - a) * 1e3, 1) for a, m in zip([t0] + marks, marks + [t0 + dt])]
        c_pk = eng.counters()
        assert c_pk["n_unique_kmers"] == c_ascii["n_unique_kmers"] and c_pk["n_kmers_ingested"] == c_ascii["n_kmers_ingested"]
        out["packed" + tag] = round(nh * L / dt / 1e9, 2)
        pk.close()
        del hb
print(out)
