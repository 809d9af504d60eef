#!/usr/bin/env python3
"""bench.py — Gbases/s of k-mer counting (k=21, 150 bp reads), histograms bit-exact vs CPU.

One "step" = the whole BASELINE.json config-2 job on one GPU: an empty table, one batch of
synthetic reads already resident in HBM → validate/scan → count → histogram emit.
With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank counts its own
shard of reads (weak scaling: per-GPU work fixed); the per-rank tables are exchanged by owner
page range over RCCL (all_to_all), merged, scanned, and the histograms all-reduced, so the
emitted histogram is that of the union of all reads.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed with HIP events
on the engine's own stream inside libshk (SHK_FLAG_TIMING); `cpu_baseline` times the CPU
oracle (a single-threaded C restatement of the reference algorithm — "port") on the same
reads and checks the GPU histogram against it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_all_cores(orc, hb, ho, k, histo_max, n_bases):
    """The CPU oracle on every host core (at most 16, a one-GPU box's share): contiguous shards of
    the reads counted in threads (ctypes releases the GIL), then merged sequentially."""
    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(os.cpu_count() or 1, 16))
    n = len(ho) - 1
    cuts = [n * i // T for i in range(T + 1)]

    def one(i):
        a, b = cuts[i], cuts[i + 1]
        return orc.run_batch(hb[int(ho[a]):int(ho[b])], ho[a:b + 1] - ho[a], k, 0, histo_max)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(T) as ex:
        runs = list(ex.map(one, range(T)))
    dst = runs[0].merged()
    for r in runs[1:]:
        dst.extend(r.merged())
    h = orc.Histogram.from_kmer_counts(dst, histo_max)
    n_unique = h.get_n_unique_kmers()
    dt = time.perf_counter() - t0
    del runs
    return {"value": round(n_bases / dt / 1e9, 5), "unit": "Gbases/s", "cores": T, "kind": "port",
            "sample": f"the same batch sharded over {T} threads, tables merged sequentially, one histogram "
                      f"({n_unique} distinct k-mers)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 50 warm-up steps (≈40 ms) because the first ≈20 ms of work after the card has been idle run
    # ≈4 % slower (measured: scatter 0.48 ms with 3 warm-up steps, 0.46 ms with 50 or 300), then 100 timed steps
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step")
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--genome", type=int, default=3_000_000)
    ap.add_argument("--chunks", type=int, default=1)
    ap.add_argument("--histo-max", type=int, default=10000)
    ap.add_argument("--path", choices=["auto", "direct", "paged"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=0,
                    help="reads for the CPU baseline (0 = the whole step batch of rank 0)")
    args = ap.parse_args()

    import torch
    import sharkmer_amd as sa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # SHK_BENCH_FORCE_DIST=1: take the multi-GPU code path (RCCL collectives, owner exchange, merged
    # histogram) with a world of one — a rehearsal of the N > 1 run on a single card
    force_dist = world == 1 and bool(os.environ.get("SHK_BENCH_FORCE_DIST"))
    if world > 1 or force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)
    L = 150
    n_reads = args.reads
    # weak scaling: every GPU counts its own 1 M reads of the SAME 3 Mb genome, so the per-GPU
    # work (k-mer occurrences, distinct load, table size) is exactly config 2's at every N; the
    # merged histogram is that of N × 50× coverage
    genome = args.genome
    spec = sa.SynthSpec(genome_len=genome, read_len=L)
    flags = 0 if os.environ.get("SHK_BENCH_NO_TIMING") else sa.FLAG_TIMING  # (experiment hook)
    if args.path == "direct":
        flags |= sa.FLAG_FORCE_DIRECT
    elif args.path == "paged":
        flags |= sa.FLAG_FORCE_PAGED
    eng = sa.KmerEngine(args.k, args.chunks, args.histo_max, device=dev,
                        capacity_hint=genome, flags=flags)

    d_bases = torch.empty(n_reads * L, dtype=torch.uint8, device=f"cuda:{dev}")
    d_offsets = torch.empty(n_reads + 1, dtype=torch.int64, device=f"cuda:{dev}")
    eng.synth_reads_device(spec, rank * n_reads, n_reads, d_bases.data_ptr(), d_offsets.data_ptr())
    n_bases = n_reads * L

    if dist is not None:
        from sharkmer_amd.dist import DistCounter
        dc = DistCounter(eng, dist, device=dev)

    def step():
        eng.reset()
        eng.set_read_index(rank * n_reads)
        eng.ingest_reads_device(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_bases)
        if dist is not None:
            return dc.finalize_histograms()
        eng.finalize()
        return eng.histograms()

    for _ in range(args.warmup):
        step()
    eng.reset_timings()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hist = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    tim = eng.timings()
    cnt = eng.counters()
    total_bases = n_bases * n_gpus * args.steps
    value = total_bases / dt / 1e9

    if rank == 0:
        # ---- roofline (HBM-bound path; algorithmic bytes per launch) --------------
        kpr = L - args.k + 1
        n_kmers = n_reads * kpr
        nd = cnt["n_unique_kmers"] if world == 1 else min(genome, n_kmers)
        cap = cnt["table_capacity"]
        lanes = max(args.chunks, 1)
        # per-kernel algorithmic bytes of ONE launch (SURVEY.md §8d's per-unit figures: 1 B per base,
        # 8 B per k-mer record written and read, 12 B per table slot in and out); a step has several
        # launches of a kernel when the batch spans chunk lanes or exceeds 2^28 bases, and then each
        # launch moves its share of the batch (but every page of the table)
        def alg_bytes(name, lps):
            return {
                "scatter": (n_bases * 1 + n_kmers * 8) / lps,      # bases read + one record per k-mer written
                "pages": n_kmers * 8 / lps + cap * 12 * 2,         # records read + every page in and out
                "histo": cap * (8 + 4 * lanes),                    # one table scan per histogram emit
                "direct": (n_bases * 1 + n_kmers * 16 + nd * 8) / lps,
                "scan": n_bases * 1 / lps,
            }.get(name)
        per_kernel = {}
        for name, (ms, launches) in tim.items():
            lps = launches / args.steps
            if launches and alg_bytes(name, lps) is not None:
                avg = ms / launches
                ab = alg_bytes(name, lps)
                per_kernel[name] = {"avg_launch_ms": round(avg, 4), "launches_per_step": lps,
                                    "alg_bytes_per_launch": int(ab),
                                    "achieved_GBps": round(ab / (avg * 1e-3) / 1e9, 1)}
        hot = [k_ for k_ in ("direct", "scatter", "pages") if k_ in per_kernel]
        dom = max(hot, key=lambda k_: per_kernel[k_]["avg_launch_ms"] * per_kernel[k_]["launches_per_step"]) if hot else None
        roof = None
        if dom:
            d = per_kernel[dom]
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
            if os.path.exists(tpath):  # PMC bytes of the same command, collected by tools/profile_round.sh
                tj = json.load(open(tpath))
                if tj.get("reads") == n_reads and tj.get("k") == args.k:
                    traffic = tj.get("kernels", {}).get(dom, {}).get("hbm_bytes_per_launch")
            roof = {"bound": "hbm", "kernel": dom, "achieved": d["achieved_GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(d["achieved_GBps"] / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "alg_bytes_per_launch": d["alg_bytes_per_launch"],
                    "avg_launch_ms": d["avg_launch_ms"]}
        # the whole counting path against SURVEY.md §8d's B_alg (1 B/base + 16 B/k-mer + 8 B/distinct
        # + one table scan per emit), over the summed device time of its kernels
        path_ms = sum(ms for name, (ms, _) in tim.items()
                      if name in ("mark", "scan", "direct", "pcount", "pscan", "scatter", "pages", "histo")) / args.steps
        b_alg = n_bases * 1 + n_kmers * 16 + nd * 8 + cap * (8 + 4 * lanes)
        path_roof = {"alg_bytes_per_step": int(b_alg), "device_ms_per_step": round(path_ms, 4),
                     "achieved_GBps": round(b_alg / (path_ms * 1e-3) / 1e9, 1) if path_ms else None,
                     "frac": round(b_alg / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if path_ms else None}
        cpu = cpu_all = None
        if not args.no_cpu_baseline:
            from oracle import oracle as orc
            ns = args.cpu_sample_reads or n_reads
            hb, ho = sa.synth_reads(spec, 0, ns)
            t1 = time.perf_counter()
            ref = orc.run_batch(hb, ho, args.k, args.chunks, args.histo_max)
            cdt = time.perf_counter() - t1
            cpu = {"value": round(ns * L / cdt / 1e9, 5), "unit": "Gbases/s", "cores": 1,
                   "kind": "port",
                   "sample": f"{ns} reads x {L} bp of the step batch (rank 0 shard), "
                             f"single-threaded C restatement of sharkmer's counting path"}
            cpu["cpu_model"] = cpu_model()
            # context only: the same restatement on all host cores — reads sharded over
            # threads, the per-thread tables merged one after the other (KmerCounts::extend), one
            # histogram of the merged table.  The reference itself counts on one thread.
            cpu_all = cpu_all_cores(orc, hb, ho, args.k, args.histo_max, ns * L)
            if world == 1 and ns == n_reads:
                exact = bool(np.array_equal(hist, ref.histograms()))
                cpu["histogram_bit_exact"] = exact
                if not exact:
                    print("ERROR: GPU histogram differs from the CPU oracle", file=sys.stderr)
                    sys.exit(2)
        out = {
            "metric": "Gbases/sec k-mer counted (k=21, 150bp reads); histogram bit-exact vs CPU",
            "value": round(value, 4), "unit": "Gbases/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[1]: {n_reads} synthetic {L}bp reads per GPU, "
                                   f"k={args.k}, {n_gpus}xMI355X, single hash-table shard per GPU; "
                                   f"step = reset + count + histogram emit, input resident in HBM",
                       "reads_per_gpu": n_reads, "k": args.k, "chunks": args.chunks,
                       "genome": genome, "path": args.path},
            "roofline": roof, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
            "roofline_path": path_roof, "kernels": per_kernel,
            "kernels_ms_per_step": {k_: round(v[0] / args.steps, 4) for k_, v in tim.items()},
            "table": {"capacity": cnt["table_capacity"], "n_unique": cnt["n_unique_kmers"],
                      "n_grows": cnt["n_grows"], "n_spilled": cnt["n_spilled"]},
        }
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
