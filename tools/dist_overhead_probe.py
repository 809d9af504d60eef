#!/usr/bin/env python3
"""Where the fixed cost of the multi-GPU finalize goes: world of one over RCCL, each phase timed
with a device sync after it."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402
from sharkmer_amd.dist import DistCounter  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n_reads, L = 1_000_000, 150
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
eng = sa.KmerEngine(21, 1, 10000, device=0, capacity_hint=3_000_000)
d_bases = torch.empty(n_reads * L, dtype=torch.uint8, device="cuda:0")
d_offsets = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda:0")
eng.synth_reads_device(spec, 0, n_reads, d_bases.data_ptr(), d_offsets.data_ptr())
dc = DistCounter(eng, dist, device=0)
acc = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return r


for it in range(12):
    if it == 2:
        acc.clear()
    eng.reset()
    eng.ingest_reads_device(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_reads * L)
    timed("sync_after_ingest", lambda: eng.sync())
    P, _, n_lanes = timed("agree_on_pages", dc._agree_on_pages)
    counts = timed("owner_counts", lambda: np.asarray(eng.owner_counts(1), dtype=np.int64))
    counts[0] = 0
    send_n = torch.from_numpy(counts.copy()).cuda()
    recv_n = torch.empty_like(send_n)
    timed("a2a_counts", lambda: (dist.all_to_all_single(recv_n, send_n), recv_n.cpu()))
    keys, vals = timed("compact", lambda: eng.compact_owner_tensors(counts, 0))
    rk = keys.new_empty(0)
    timed("a2a_keys_vals", lambda: (dist.all_to_all_single(rk, keys, output_split_sizes=[0], input_split_sizes=[0]),
                                    dist.all_to_all_single(rk.to(torch.int32), vals[0].contiguous(), output_split_sizes=[0], input_split_sizes=[0])))
    timed("own", lambda: dc._own(P))
    timed("finalize", eng.finalize)
    h = timed("histograms+counters", lambda: (eng.histograms(), eng.counters()))
    pt = torch.zeros(10009, dtype=torch.int64, device="cuda")
    timed("allreduce_hist", lambda: (dist.all_reduce(pt), pt.cpu()))
print({k: round(v / 10 * 1e3, 3) for k, v in acc.items()}, "ms per step")
dist.destroy_process_group()
