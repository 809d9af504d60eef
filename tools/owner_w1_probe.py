#!/usr/bin/env python3
"""Probe: a share that is the whole key space (n_owners = 1) through OwnerCounter's exchange rounds over a one-rank RCCL
communicator, against the same reads through the ordinary ingest of a whole-key-space context (n_owners = 0): kernel
times, spills, grows.  usage: python3 tools/owner_w1_probe.py [reads] [genome] [lanes]"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402
from sharkmer_amd.dist import OwnerCounter  # noqa: E402
import torch.distributed as dist  # noqa: E402

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
genome = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000_000
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
L, k = 150, (int(sys.argv[4]) if len(sys.argv) > 4 else 21)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
spec = sa.SynthSpec(genome_len=genome, read_len=L)
rr = 1_700_000
n_rounds = -(-reads // rr)
for mode in ("share_w1", "plain"):
    eng = sa.KmerEngine(k, lanes, 1000, capacity_hint=genome, flags=sa.FLAG_TIMING, n_owners=1 if mode == "share_w1" else 0)
    d_all = torch.empty(reads * L, dtype=torch.uint8, device="cuda:0")
    d_off = torch.empty(rr + 1, dtype=torch.int64, device="cuda:0")
    for r in range(n_rounds):
        n = min(rr, reads - r * rr)
        eng.synth_reads_device(spec, r * rr, n, d_all.data_ptr() + r * rr * L, d_off.data_ptr())
    eng.sync()
    oc = OwnerCounter(eng, dist, device=0, round_bases=rr * L) if mode == "share_w1" else None
    for rep in range(2):
        eng.reset()
        eng.reset_timings()
        torch.cuda.synchronize()
        t0 = time.time()
        for r in range(n_rounds):
            n = min(rr, reads - r * rr)
            ptr = d_all.data_ptr() + r * rr * L
            if oc:
                oc.round((ptr, d_off.data_ptr(), n, n * L, r * rr))
            else:
                eng.set_read_index(r * rr)
                eng.ingest_reads_device(ptr, d_off.data_ptr(), n, n * L)
        if oc:
            oc.finalize_histograms()
        else:
            eng.finalize()
        torch.cuda.synchronize()
        dt = time.time() - t0
        c = eng.counters()
        print(json.dumps({"mode": mode, "rep": rep, "seconds": round(dt, 3), "gbases_per_s": round(reads * L / dt / 1e9, 1),
                          "n_spilled": c["n_spilled"], "n_grows": c["n_grows"], "cap": c["table_capacity"], "n_unique": c["n_unique_kmers"],
                          "kernel_ms": {k_: (round(v[0], 1), v[1]) for k_, v in eng.timings().items() if v[0] > 0}}), flush=True)
    eng.close()
    del d_all, d_off
    torch.cuda.empty_cache()
dist.destroy_process_group()
