"""Host-side profile of the forced single-rank DistCounter step (where do the fixed 0.2 ms go?).
Run on the GPU box: python tools/dist_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402
from sharkmer_amd.dist import DistCounter  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
n_reads, L = 1_000_000, 150
eng = sa.KmerEngine(21, 1, 10000, capacity_hint=3_000_000)
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
d_bases = torch.empty(n_reads * L, dtype=torch.uint8, device="cuda:0")
d_offsets = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda:0")
eng.synth_reads_device(spec, 0, n_reads, d_bases.data_ptr(), d_offsets.data_ptr())
dc = DistCounter(eng, dist, device=0)


def step():
    eng.reset()
    eng.set_read_index(0)
    eng.ingest_reads_device(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_reads * L)
    return dc.finalize_histograms()


for _ in range(300):
    step()
t0 = time.perf_counter()
for _ in range(200):
    step()
print("ms/step", (time.perf_counter() - t0) / 200 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
dist.destroy_process_group()
