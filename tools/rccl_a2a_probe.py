#!/usr/bin/env python3
"""Probe: does a world-of-one all_to_all_single move all of a large tensor?  (It did not: see DESIGN.md.)"""
import os, sys
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for n in (1 << 20, (1 << 27) - 3, 1 << 27, (1 << 27) + 5, 3 << 26, 1 << 28, 273_678_336):
    src = torch.arange(n, dtype=torch.int32, device="cuda") ^ 0x5A5A5A5
    dst = torch.zeros_like(src)
    dist.all_to_all_single(dst, src)
    torch.cuda.synchronize()
    bad = (dst != src)
    nb = int(bad.sum().item())
    first = int(bad.nonzero()[0].item()) if nb else -1
    dst2 = torch.zeros_like(src)
    dist.all_to_all([dst2], [src])
    torch.cuda.synchronize()
    nb2 = int((dst2 != src).sum().item())
    dst3 = torch.zeros_like(src)
    ops = [dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst3, 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    nb3 = int((dst3 != src).sum().item())
    print(f"n={n} ({n*4/2**20:.1f} MiB): all_to_all_single wrong={nb} first_wrong={first} | all_to_all(list) wrong={nb2} | p2p self wrong={nb3}", flush=True)
dist.destroy_process_group()
