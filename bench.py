#!/usr/bin/env python3
"""bench.py — Gbases/s of k-mer counting (k=21, 150 bp reads), histograms bit-exact vs CPU.

--config 2 (default; BASELINE.json configs[1], the configuration the metric is quoted on): one "step" = the whole
job on one GPU: an empty table, one batch of synthetic reads already resident in HBM → validate/scan → count →
histogram emit.  The timed loop ROTATES over `--batches` (4) distinct resident batches — consecutive reads of the same
genome — so no input byte is re-read within 600 MB of input traffic and nothing of it can sit in the 256 MiB Infinity
Cache between steps.  With --gpus N (launched by torch.distributed.run, one rank per GPU) every rank counts its own
shard of reads (weak scaling: per-GPU work fixed); the per-rank tables are exchanged by owner page range over RCCL
(all_to_all), merged, scanned, and the histograms all-reduced, so the emitted histogram is that of the union of all
reads.

--config 4 / --config 5 (BASELINE.json configs[3] / configs[4]: 1 B reads over a 3 Gb genome on 8×MI355X, one chunk
lane resp. 10 cumulative subsets): KEY-SPACE-PARTITIONED ingest (sharkmer_amd.dist.OwnerCounter) — every rank holds
1/N of the key space, scatters its reads' k-mer records by owner, the records cross the links in one all_to_all per
round, every rank absorbs what it owns; histograms are summed at the end.  A rank's input is ITS SHARE OF THE CONFIG
AS STATED: 10^9 / 8 = 125 M reads, resident in HBM (18.75 GB), whatever N is — at N = 8 the run IS the config; at
smaller N it is the same per-rank load on a proportionally smaller job (weak scaling).  At N = 1: config 4 runs the
exchange path over a one-rank RCCL communicator into the whole 2^33-slot table; config 5's whole key space (48 B ×
2^33 slots) does not fit one card, so the one rank plays owner 0 of 8 and drops what the seven others would have
received (the same kernels; no exchange) — `config.workload` says which.

Before the W warm-up steps an untimed RAMP of steps (≈0.25 s of device work) lets the card reach its
steady clocks — the first ≈20 ms after idling run ≈4 % slower; `ramp_steps` is reported.  The timed region
is exactly K steps.  After it, outside the timed region and at N = 1 only, `extras` puts the other timing
points of SURVEY.md §8d on record: pinned host buffers → histogram (ii), FASTQ file → histogram (iii),
BASELINE configs[2] (k = 31, 100 M reads) from HBM and streamed from pinned host memory, and the
chunk-lane / large-table shapes.  --no-extras skips them.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, timed with HIP events on the engine's own
stream inside libshk (SHK_FLAG_TIMING): `achieved` = ALGORITHMIC bytes per launch ÷ its duration; every
kernel also carries `physical_GBps` = the HBM bytes per launch the committed PMC passes of this round measured
(profiles/rNN_traffic.json) ÷ the same duration.  `roofline_path` is the whole job against SURVEY.md §8d's B_alg, over
summed kernel time (`frac`) and over the driver's own clock (`frac_driver_clock` = B_alg ÷ ms_per_step).
`cpu_baseline` times the CPU oracle (a single-threaded C restatement of the reference algorithm — "port") on the
reads of the last timed step and checks the GPU histogram against it.
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


def _timings_ms(eng, reps):
    return {k_: round(v[0] / reps, 4) for k_, v in eng.timings().items()}


def extras(sa, torch, dev):
    """The other timing points (SURVEY.md §8d ii / iii, BASELINE configs[2], chunk lanes, large tables), each
    OUTSIDE the headline's timed region; wall clock around whole jobs, best of a few repetitions, inputs
    generated on the device (untimed).  A failure in one of them is recorded, never raised."""
    import tempfile
    out = {}
    L = 150

    def device_reads(eng, spec, first, n):
        db = torch.empty(n * L, dtype=torch.uint8, device=f"cuda:{dev}")
        do = torch.empty(n + 1, dtype=torch.int64, device=f"cuda:{dev}")
        eng.synth_reads_device(spec, first, n, db.data_ptr(), do.data_ptr())
        eng.sync()
        return db, do

    only = [x for x in os.environ.get("SHK_BENCH_EXTRAS", "").split(",") if x]   # (a probe's shortcut: just these)

    def guarded(name, fn):
        if only and name not in only:
            return
        try:
            t0 = time.perf_counter()
            out[name] = fn()
            out[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- (ii) pinned host buffers → histogram: config-2 reads in batches of 4 M, PCIe inclusive --------
    def host_pinned():
        n = 4_000_000
        spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
        res = {"workload": f"{n} reads x {L} bp (config 2's genome) handed over as HOST buffers: shk_ingest_reads + finalize"}
        with sa.KmerEngine(21, 1, 10000, device=dev, capacity_hint=3_000_000) as eng:
            db, do = device_reads(eng, spec, 0, n)
            hb = torch.empty(n * L, dtype=torch.uint8, pin_memory=True)
            hb.copy_(db)
            ho_t = torch.empty(n + 1, dtype=torch.int64, pin_memory=True)
            ho_t.copy_(torch.arange(n + 1, dtype=torch.int64) * L)
            ho = ho_t.numpy().view(np.uint64)
            del db, do
            pageable = hb.numpy().copy()
            for kind, arr, ho_, hp in (("pinned", hb.numpy(), ho, None), ("pageable", pageable, ho.copy(), None),
                                       ("pinned_packed_by_the_library", hb.numpy(), ho, "1"), ("pageable_ascii_on_the_wire", pageable, ho.copy(), "0")):
                if hp is not None:
                    os.environ["SHK_HOST_PACK"] = hp   # (read at every call)
                eng.reset()
                eng.ingest_reads(arr, ho_)
                eng.finalize()
                best = None
                for _ in range(3):
                    eng.reset()
                    t0 = time.perf_counter()
                    eng.ingest_reads(arr, ho_)
                    eng.finalize()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                os.environ.pop("SHK_HOST_PACK", None)
                ascii_wire = hp == "0" or (hp is None and kind == "pinned")
                res[kind] = {"Gbases_per_s": round(n * L / best / 1e9, 2),
                             "bound": "PCIe H2D, 1 B/base ASCII" if ascii_wire else
                                      "the library's packer thread: every slice 2-bit on the host's cores (a pageable batch on 12 or more cores: the default), 0.3 B/base on the link"}
                if ascii_wire:
                    res[kind]["pcie_GB_per_s"] = round(n * L / best / 1e9, 2)
            del pageable
            if hasattr(eng, "ingest_packed"):
                t_p = time.perf_counter()
                pk = sa.pack_reads(hb.numpy(), ho, pinned=True)
                t_p = time.perf_counter() - t_p
                eng.reset()
                eng.ingest_packed(pk)
                eng.finalize()
                best = None
                for _ in range(3):
                    eng.reset()
                    t0 = time.perf_counter()
                    eng.ingest_packed(pk)
                    eng.finalize()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                res["pinned_packed"] = {"Gbases_per_s": round(n * L / best / 1e9, 2),
                                        "bound": "2-bit packed stream + N mask (as the list of its non-zero words when they are few) + 8-B offsets over PCIe (shk_pack_reads untimed)",
                                        "pcie_GB_per_s": round(pk.wire_bytes() / best / 1e9, 2), "wire_bytes_per_base": round(pk.wire_bytes() / (n * L), 3),
                                        "host_pack_Gbases_per_s": round(n * L / t_p / 1e9, 2)}
                # … and with the packing inside the clock: ASCII in pinned memory → shk_pack_reads → shk_ingest_packed →
                # histogram, one after the other (the FASTQ front-end packs while it parses instead: extras.file_path)
                best = None
                for _ in range(3):
                    eng.reset()
                    t0 = time.perf_counter()
                    sa.pack_reads(hb.numpy(), ho, out=pk)   # (into the pinned arrays of the batch packed above)
                    eng.ingest_packed(pk)
                    eng.finalize()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                res["pinned_packed_incl_pack"] = {"Gbases_per_s": round(n * L / best / 1e9, 2),
                                                  "bound": "host: shk_pack_reads on the box's CPU quota into pinned arrays, then the packed ingest"}
                pk.close()
        return res
    guarded("host_pinned", host_pinned)

    # ---- (iii) FASTQ(.gz) file → histogram + output files (shk_run_files), host parse inclusive --------
    def file_path():
        import gzip
        n = 8_000_000
        spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
        with sa.KmerEngine(21, 1, 10000, device=dev, capacity_hint=3_000_000) as eng:
            db, _ = device_reads(eng, spec, 0, n)
            bases = db.cpu().numpy()
            del db
        tmp = tempfile.mkdtemp(prefix="shk_bench_")
        rec = np.empty((n, 2 * L + 7), dtype=np.uint8)  # "@r\n" + seq + "\n+\n" + qual + "\n"
        rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
        rec[:, 3:3 + L] = bases.reshape(n, L)
        rec[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, 6 + L:6 + 2 * L] = ord("I")
        rec[:, 6 + 2 * L] = ord("\n")
        plain = os.path.join(tmp, "reads.fastq")
        rec.tofile(plain)
        parts = []
        NP = 8
        for i in range(NP):  # eight gzip files: the multi-file case (one inflate thread per stream, later streams run ahead)
            a, b = n * i // NP * (2 * L + 7), n * (i + 1) // NP * (2 * L + 7)
            pth = os.path.join(tmp, f"part{i}.fastq.gz")
            with gzip.open(pth, "wb", compresslevel=1) as g:
                g.write(rec.reshape(-1)[a:b].tobytes())
            parts.append(pth)
        res = {"workload": f"{n} reads x {L} bp as FASTQ in the page cache → shk_run_files (parse + count + .histo/.stats.yaml), best of 3 (the first job of a process also maps the device and pinned memory that later ones find in the library's cache)"}
        for name, paths in (("plain", [plain]), ("gzip_8_files", parts), ("gzip_1_file", parts[:1])):
            best, first = None, None
            for _ in range(3):
                t0 = time.perf_counter()
                sa.run_files(paths, k=21, chunks=1, histo_max=10000, sample="s", outdir=tmp, capacity_hint=3_000_000)
                dt = time.perf_counter() - t0
                first = dt if first is None else first
                best = dt if best is None else min(best, dt)
            nn = n if name != "gzip_1_file" else n // NP
            res[name] = {"Gbases_per_s": round(nn * L / best / 1e9, 3), "first_job_Gbases_per_s": round(nn * L / first / 1e9, 3),
                         "file_MB": round(sum(os.path.getsize(p_) for p_ in paths) / 1e6, 1), "bound": "host: read + inflate + line split"}
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
        return res
    guarded("file_path", file_path)

    # ---- BASELINE configs[2]: 100 M reads, k = 31, a 300 Mb genome; from HBM, and streamed from pinned host memory
    def config3():
        n, k, batch = 100_000_000, 31, 4_000_000
        spec = sa.SynthSpec(genome_len=300_000_000, read_len=L)
        res = {"workload": f"BASELINE.json configs[2]: {n} synthetic {L}bp reads, k={k}, 300 Mb genome (2^30-slot table), batches of {batch} reads"}
        with sa.KmerEngine(k, 1, 10000, device=dev, capacity_hint=300_000_000, flags=sa.FLAG_TIMING) as eng:
            d_all = torch.empty(n * L, dtype=torch.uint8, device=f"cuda:{dev}")
            d_off = torch.empty(batch + 1, dtype=torch.int64, device=f"cuda:{dev}")
            for b in range(n // batch):
                eng.synth_reads_device(spec, b * batch, batch, d_all.data_ptr() + b * batch * L, d_off.data_ptr())
            eng.sync()
            for rep in range(2):
                eng.reset()
                eng.reset_timings()
                t0 = time.perf_counter()
                for b in range(n // batch):
                    eng.ingest_reads_device(d_all.data_ptr() + b * batch * L, d_off.data_ptr(), batch, batch * L)
                eng.finalize()
                dt = time.perf_counter() - t0
            c = eng.counters()
            res["from_hbm"] = {"Gbases_per_s": round(n * L / dt / 1e9, 2), "seconds": round(dt, 4), "kernels_ms": _timings_ms(eng, 1),
                               "n_kmers": c["n_kmers_ingested"], "kmers_as_expected": c["n_kmers_ingested"] == (L - k + 1) * n,
                               "n_unique": c["n_unique_kmers"]}
            # host-streamed: the first 24 M reads from pinned memory (the same engine, the table already sized)
            nh = 24_000_000
            hb = torch.empty(nh * L, dtype=torch.uint8, pin_memory=True)
            hb.copy_(d_all[:nh * L])
            del d_all
            torch.cuda.empty_cache()
            ho = np.arange(batch + 1, dtype=np.uint64) * np.uint64(L)
            eng.reset()
            eng.ingest_reads(hb.numpy()[:batch * L], ho)  # (untimed: the staging buffers of the host path are allocated here)
            dt = None
            for rep in range(2):   # (best of two: a single pass has come out at 24 once, behind the file jobs' teardown)
                eng.reset()
                t0 = time.perf_counter()
                for b in range(nh // batch):
                    eng.ingest_reads(hb.numpy()[b * batch * L:(b + 1) * batch * L], ho)
                eng.finalize()
                dt = min(dt or 1e9, time.perf_counter() - t0)
            res["host_streamed"] = {"reads": nh, "Gbases_per_s": round(nh * L / dt / 1e9, 2), "pcie_GB_per_s": round(nh * L / dt / 1e9, 2),
                                    "bound": "PCIe H2D, 1 B/base ASCII; the copies of the next slices are queued under the counting of slice i"}
            # … and the same reads as a 2-bit packed stream + N mask (packed once, untimed; its rate is reported)
            ho_all = np.arange(nh + 1, dtype=np.uint64) * np.uint64(L)
            t_p = time.perf_counter()
            pk = sa.pack_reads(hb.numpy(), ho_all, pinned=True)
            t_p = time.perf_counter() - t_p
            del hb
            for rep in range(2):
                eng.reset()
                t0 = time.perf_counter()
                for b in range(nh // batch):
                    eng.ingest_packed_slice(pk, b * batch, batch)
                eng.finalize()
                dt = time.perf_counter() - t0
            res["host_streamed_packed"] = {"reads": nh, "Gbases_per_s": round(nh * L / dt / 1e9, 2), "pcie_GB_per_s": round(pk.wire_bytes() / dt / 1e9, 2),
                                           "host_pack_Gbases_per_s": round(nh * L / t_p / 1e9, 2),
                                           "bound": "the counting itself (0.30 B/base on the link); shk_pack_reads untimed"}
            pk.close()
        return res
    guarded("config3", config3)

    # ---- chunk lanes and large tables (the shapes BASELINE configs[3]/[4] put on every GPU) -----------------
    def shapes():
        res = {}
        for name, genome, n, chunks, steps in (("config2_10_lanes", 3_000_000, 1_000_000, 10, 30),
                                               ("config2_40_lanes", 3_000_000, 1_000_000, 40, 10),
                                               ("config2_100_lanes", 3_000_000, 1_000_000, 100, 5),   # the reference's own historical runs: n = 100 (sharkmer_viewer/tests/data/Cordagalma.stats)
                                               ("genome_30Mb_1_lane", 30_000_000, 1_700_000, 1, 10),
                                               ("genome_30Mb_10_lanes", 30_000_000, 1_700_000, 10, 10)):
            spec = sa.SynthSpec(genome_len=genome, read_len=L)
            with sa.KmerEngine(21, chunks, 10000, device=dev, capacity_hint=genome, flags=sa.FLAG_TIMING) as eng:
                db, do = device_reads(eng, spec, 0, n)

                def one():
                    eng.reset()
                    eng.ingest_reads_device(db.data_ptr(), do.data_ptr(), n, n * L)
                    eng.finalize()
                for _ in range(3):
                    one()
                eng.reset_timings()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    one()
                dt = time.perf_counter() - t0
                res[name] = {"workload": f"{n} reads, {genome} bp genome, {chunks} chunk lane(s); step = reset + count + histogram emit, input in HBM",
                             "Gbases_per_s": round(n * L * steps / dt / 1e9, 2), "ms_per_step": round(dt / steps * 1e3, 4),
                             "kernels_ms_per_step": _timings_ms(eng, steps)}
        return res
    guarded("shapes", shapes)

    # ---- the multi-GPU finalize's fixed cost: this very bench with SHK_BENCH_FORCE_DIST=1 (the table exchange and
    # the device-side histogram reduction over RCCL in a world of ONE: what every rank of an N-GPU run adds per
    # job) and without, as child processes one after the other -------------------------------------------------
    def dist_fixed_cost():
        import subprocess
        res = {}
        for name, force in (("plain", False), ("merged", True)):
            env = dict(os.environ)
            env.pop("SHK_BENCH_FORCE_DIST", None)
            if force:
                env["SHK_BENCH_FORCE_DIST"] = "1"
                env["MASTER_PORT"] = "29543"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-extras", "--no-cpu-baseline"], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=True)
            res[name] = json.loads(r.stdout.decode().strip().splitlines()[-1])["ms_per_step"]
        return {"workload": "config 2's step (bench.py defaults) with SHK_BENCH_FORCE_DIST=1 — fixed-capacity pieces + in-place histogram "
                            "all_reduce over RCCL, world of one — and without",
                "ms_per_step_plain": res["plain"], "ms_per_step_merged": res["merged"],
                "fixed_cost_ms": round(res["merged"] - res["plain"], 4)}
    guarded("dist_fixed_cost", dist_fixed_cost)

    # k > 21 between owner shares, a world of one: a share of configs[2]'s shape (k = 31, 300 Mb genome) through
    # OwnerCounter's rounds — the owner layout with 8-byte records (round 4: the same segment exchange as for k ≤ 21), and
    # the WIDE round it replaces for ≤ 32 chunk lanes (SHK_XL64=0: whole k-mers grouped by owner, 12 B each on the links,
    # counted at the receiver through the partition + page passes since round 4, by global atomics before) — beside the
    # same reads through the plain ingest
    def exchange_k31():
        import subprocess
        res = {"workload": "12.5 M reads x 150 bp, k=31, 300 Mb genome (2^30-slot table): OwnerCounter's rounds over RCCL, a world of one (n_owners = 1), "
                           "against the plain ingest of the same reads; best of 2 passes each"}
        for name, env in (("owner_layout_8_byte_records", {}), ("wide_round", {"SHK_XL64": "0"})):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "owner_w1_probe.py"), "12500000", "300000000", "1", "31"],
                               env={**os.environ, "MASTER_PORT": "29547", **env}, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=True)
            rows = [json.loads(ln) for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
            best = {m: max((x for x in rows if x["mode"] == m), key=lambda x: x["gbases_per_s"]) for m in ("share_w1", "plain")}
            res[name] = {"Gbases_per_s": best["share_w1"]["gbases_per_s"], "kernels_ms": {k_: v[0] for k_, v in best["share_w1"]["kernel_ms"].items()}}
            res["plain_ingest"] = {"Gbases_per_s": best["plain"]["gbases_per_s"], "kernels_ms": {k_: v[0] for k_, v in best["plain"]["kernel_ms"].items()}}
        return res
    guarded("exchange_k31", exchange_k31)
    return out


def owner_config(args, real_stdout):
    """--config 4 / 5: one rank's share of BASELINE.json configs[3] / configs[4] per GPU, key-space-partitioned ingest
    (reference semantics: io.rs:340-361 striping by global read index, io.rs:1023-1028 per-chunk histograms).

    --config 5 on ONE card (the whole key space — 48 B x 2^33 slots — does not fit it) has two ways to load the card:
      * exchange (the line's `value`): a world of one that is loaded the way a RANK OF THE 8-GPU JOB is — its own 125 M
        reads scattered, as many records absorbed as it scatters (in the job: an eighth of eight ranks' records), into
        the rank's real table: 2^30 slots x 48 B, filled by the 375 Mb eighth of the genome its share of the key space
        amounts to.  Everything a rank does per step except waiting for the links.
      * drop (`extras.drop_mode`): owner 0 of 8 of the 3 Gb genome — every read offered, 7/8 of the records dropped in
        the level-1 pass, 1/8 absorbed.  The scatter sees the job's real key distribution; the passes behind it see an
        eighth of a rank's load — an upper bound, not a configs[4] rate."""
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    mode = args.mode
    if mode == "auto":
        mode = "exchange"
    if args.config == 5 and world in (2, 3) and mode != "drop":
        raise SystemExit("--config 5 needs 1 GPU (one rank's load, or --mode drop) or >= 4 GPUs (the key space must fit)")
    out = owner_run(args, args.config, mode, dist_ready=False)
    if args.config == 5 and world == 1 and mode == "exchange" and not args.no_extras:
        try:
            d = owner_run(args, 5, "drop", dist_ready=True)
            out["extras"] = {"drop_mode": {key: d[key] for key in ("value", "ms_per_step", "config", "kernels_ms_per_step", "table", "roofline")}}
        except Exception as e:  # noqa: BLE001 — an extra must not take the line down
            out["extras"] = {"drop_mode": {"error": f"{type(e).__name__}: {e}"[:300]}}
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def owner_run(args, config, mode, dist_ready):
    """One measurement of --config 4 / 5 (see owner_config); returns the line as a dict on rank 0 (None elsewhere)."""
    import torch
    import sharkmer_amd as sa
    from sharkmer_amd.dist import OwnerCounter

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    L, k, histo_max = 150, 21, args.histo_max
    lanes = 1 if config == 4 else 10
    JOB_GPUS = 8                                  # BASELINE.json: "8×MI355X"
    drop_mode = mode == "drop"
    # the genome a rank's table sees: the job's 3 Gb — or, where ONE rank stands in for a rank of the 8-GPU job (config
    # 5, a world of one, exchange mode), the eighth of it that a rank's share of the key space amounts to
    rank_load = config == 5 and world == 1 and not drop_mode
    genome = 3_000_000_000 // JOB_GPUS if rank_load else 3_000_000_000
    rpr = args.reads or 10**9 // JOB_GPUS         # reads per rank: its share of the job as stated
    # (two warm-up jobs: besides the first job's allocations, the runtime stands 35-50 ms ONCE — inside one call of the
    # second job's 46th round or thereabouts, with or without the timing events, never again afterwards:
    # tools/absorb_time_probe.py — which a single warm-up job left in the first timed step)
    steps, warmup = args.steps if args.steps != 100 else 3, args.warmup if args.warmup != 50 else 2
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if not drop_mode and not dist.is_initialized():
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29532")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    n_owners = JOB_GPUS if drop_mode else world
    flags = sa.FLAG_TIMING
    eng = sa.KmerEngine(k, lanes, histo_max, device=dev, capacity_hint=genome // n_owners, flags=flags,
                        n_owners=n_owners, owner_id=0 if drop_mode else rank, reserve_cus=args.reserve_cus)
    spec = sa.SynthSpec(genome_len=genome, read_len=L)
    # the rank's reads, resident in HBM: rounds of ≤ 2^28 bases
    round_reads = 1_700_000
    n_rounds = -(-rpr // round_reads)
    d_all = torch.empty(rpr * L, dtype=torch.uint8, device=f"cuda:{dev}")
    d_off = torch.empty(round_reads + 1, dtype=torch.int64, device=f"cuda:{dev}")
    first_read = rank * rpr
    for r in range(n_rounds):
        n = min(round_reads, rpr - r * round_reads)
        eng.synth_reads_device(spec, first_read + r * round_reads, n, d_all.data_ptr() + r * round_reads * L, d_off.data_ptr())
    # (d_off: offsets 0, L, 2L, … — a shorter last round writes the same values over a prefix of them)
    eng.sync()
    oc = None if drop_mode else OwnerCounter(eng, dist, device=dev, round_bases=round_reads * L)
    lay_box = [None]

    def step():
        eng.reset()
        if oc is not None:
            oc.n_rounds = oc.n_foreign_rounds = oc.wire_bytes = 0
        for r in range(n_rounds):
            n = min(round_reads, rpr - r * round_reads)
            ptr = d_all.data_ptr() + r * round_reads * L
            if oc is None:
                eng.set_read_index(first_read + r * round_reads)
                eng.ingest_reads_device(ptr, d_off.data_ptr(), n, n * L)
            else:
                lay_box[0] = oc.round((ptr, d_off.data_ptr(), n, n * L, first_read + r * round_reads))
        if oc is None:
            eng.finalize()
            return eng.histograms(), eng.counters()
        h = oc.finalize_histograms()
        return h, oc.totals

    def barrier():
        if not drop_mode and world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    eng.reset_timings()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        hist, tot = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tim = eng.timings()
    cnt = eng.counters()
    if rank == 0:
        n_bases = rpr * L
        n_kmers_rank = rpr * (L - k + 1)
        total_bases = n_bases * world * steps
        value = total_bases / dt / 1e9
        cap = cnt["table_capacity"]
        own_frac = 1.0 / n_owners if drop_mode else 1.0   # the part of a rank's records that ends up in its own page passes
        # algorithmic bytes of a step on ONE rank: its bases read, a 4-byte record per k-mer written by the level-1 pass,
        # read and written by the level-2 pass and read by the page pass (what this rank absorbs: 1/N of N ranks' records —
        # or 1/8 of its own in drop mode), every page in and out once, one table scan
        rec = 4
        stage_bytes = {"scatter": n_bases + n_kmers_rank * rec,
                       "pscan": 2 * n_kmers_rank * rec * own_frac,
                       "pages": n_kmers_rank * rec * own_frac + 2 * cap * (8 + 4 * lanes),
                       "histo": cap * (8 + 4 * lanes)}
        per_kernel = {}
        for name, (ms, launches) in tim.items():
            if launches and name in stage_bytes:
                per_step_ms = ms / steps
                per_kernel[name] = {"ms_per_step": round(per_step_ms, 3), "launches_per_step": launches / steps,
                                    "alg_bytes_per_step": int(stage_bytes[name]),
                                    "achieved_GBps": round(stage_bytes[name] / (per_step_ms * 1e-3) / 1e9, 1)}
        dom = max(per_kernel, key=lambda n_: per_kernel[n_]["ms_per_step"]) if per_kernel else None
        roof = None
        if dom:
            d = per_kernel[dom]
            lps = d["launches_per_step"]
            # HBM bytes per launch from the PMC passes of this round's build over the same command on one card
            # (tools/profile_owner.sh → profiles/rNN_config{4,5}_traffic.json); only for the mode it was taken in
            import glob
            traffic, tsrc = None, None
            for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_config{config}_traffic.json")), reverse=True):
                tj = json.load(open(tpath))
                if world == 1 and not drop_mode and not args.reads:
                    traffic = tj.get("kernels", {}).get(dom, {}).get("hbm_bytes_per_launch")
                    tsrc = os.path.relpath(tpath, ROOT)
                break
            roof = {"bound": "hbm", "kernel": dom, "achieved": d["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(d["achieved_GBps"] / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc if traffic else None,
                    "alg_bytes_per_launch": int(d["alg_bytes_per_step"] / lps), "avg_launch_ms": round(d["ms_per_step"] / lps, 4)}
        b_alg = n_bases + n_kmers_rank * 16 * own_frac + cnt["n_unique_kmers"] / (1 if drop_mode else world) * 8 + cap * (8 + 4 * lanes)
        path_roof = {"alg_bytes_per_step_per_rank": int(b_alg), "achieved_GBps": round(b_alg / (dt / steps) / 1e9, 1),
                     "frac_driver_clock": round(b_alg / (dt / steps) / 1e9 / HBM_PEAK_GBS, 5)}
        kmers_expected = (L - k + 1) * rpr * world
        exchange = None
        if oc is not None and lay_box[0] is not None:
            lay = lay_box[0]
            exchange = {"rounds": oc.n_rounds, "n_foreign_rounds": oc.n_foreign_rounds,
                        "segment_bytes": int((lay.segment_records + lay.regions) * 4),
                        "link_bytes_per_round": int((lay.segment_records + lay.regions) * 4) if world > 1 else 0,
                        "wire_bytes_per_rank_per_step": int(oc.wire_bytes), "region_cap": int(lay.region_cap), "regions": int(lay.regions)}
        cpu = None
        if not args.no_cpu_baseline:
            from oracle import oracle as orc
            ns = args.cpu_sample_reads or 200_000
            hb, ho = sa.synth_reads(spec, 0, ns)
            t1 = time.perf_counter()
            orc.run_batch(hb, ho, k, lanes, histo_max)
            cdt = time.perf_counter() - t1
            cpu = {"value": round(ns * L / cdt / 1e9, 5), "unit": "Gbases/s", "cores": 1, "kind": "port",
                   "sample": f"the first {ns} reads of the job ({lanes} chunk lane(s)), single-threaded C restatement of sharkmer's counting path",
                   "cpu_model": cpu_model()}
        which = "configs[3]" if config == 4 else "configs[4]"
        how = (f"owner 0 of {n_owners}: every read of the share offered, the other owners' records dropped in the level-1 pass (no exchange)"
               if drop_mode else f"exchange rounds among {world} owner share(s) over RCCL (OwnerCounter)")
        if rank_load:
            how += (f" — ONE card loaded like a rank of the {JOB_GPUS}-GPU job: its {rpr} reads scattered, as many records absorbed as scattered, "
                    f"the rank's table ({genome} bp of distinct k-mers, {lanes} lanes); no link time")
        out = {
            "metric": "Gbases/sec k-mer counted (k=21, 150bp reads); histogram bit-exact vs CPU",
            "value": round(value, 4), "unit": "Gbases/s", "n_gpus": (dist.get_world_size() if dist.is_initialized() else 1), "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"BASELINE.json {which}: one rank's share per GPU — {rpr} synthetic {L}bp reads of a {genome} bp genome, k={k}, "
                                   f"{lanes} chunk lane(s), {world}xMI355X; {how}; step = reset + all rounds + histogram emit, input resident in HBM",
                       "reads_per_gpu": rpr, "k": k, "chunks": lanes, "genome": genome, "n_owners": n_owners,
                       "mode": "drop" if drop_mode else "exchange", "round_reads": round_reads,
                       "reserve_cus": args.reserve_cus if args.reserve_cus else ("library default: 16 for n_owners > 1 (SHK_RESERVE_CUS)" if n_owners > 1 else 0)},
            "roofline": roof, "cpu_baseline": cpu, "roofline_path": path_roof, "kernels": per_kernel,
            "kernels_ms_per_step": {k_: round(v[0] / steps, 3) for k_, v in tim.items() if v[0] > 0},
            "exchange": exchange,
            "totals": {f: int(tot[f]) for f in ("n_reads_ingested", "n_kmers_ingested", "n_unique_kmers") if f in tot},
            "kmers_as_expected": (int(tot["n_kmers_ingested"]) == kmers_expected) if not drop_mode else None,
            "histogram_rows_sum_to_distinct": bool(int(hist[-1, 1:].sum()) == int(tot["n_unique_kmers"])),
            "table": {"capacity": cap, "n_grows": cnt["n_grows"], "n_spilled": cnt["n_spilled"]},
        }
    else:
        out = None
    eng.close()
    del d_all, d_off
    torch.cuda.empty_cache()
    return out


def launch_ranks(n: int, argv: list[str]) -> int:
    """`python3 bench.py --gpus N` without a launcher in front: start N fresh rank processes — one per GPU, RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* set the way torch.distributed.run sets them — wait for them and hand rank 0's
    ONE JSON line on.  Called before this process has imported torch or touched HIP (a process that holds the GPU must
    not exec or fork into another GPU program on this pool); the children are ordinary child processes, never an exec
    of this one.  A rank that fails takes the others down with it and the parent exits non-zero.
    SHK_BENCH_CHILD: the child's command (a test hook: a stub that records its environment)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    child = os.environ.get("SHK_BENCH_CHILD")
    cmd = child.split() if child else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC: what RCCL needs between processes on this pool)
        procs.append(subprocess.Popen(cmd + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = b""
    rc = 0
    try:
        out0, _ = procs[0].communicate()
        pending = list(range(n))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.remove(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with {code}; stopping the others", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    lines = [ln for ln in out0.decode("utf-8", "replace").splitlines() if ln.startswith("{")]
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: rank 0 printed {len(lines)} JSON lines", file=sys.stderr)
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    return rc


def main():
    # `--gpus N` with no launcher in front (WORLD_SIZE unset): this process becomes the launcher — BEFORE torch or HIP
    # are touched — and the N ranks are its children
    if "WORLD_SIZE" not in os.environ:
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument("--gpus", type=int, default=1)
        n_req = pre.parse_known_args()[0].gpus
        if n_req > 1:
            sys.exit(launch_ranks(n_req, sys.argv[1:]))
    # ONE line on stdout, whatever libraries print on theirs (RCCL announces its version there when a communicator
    # is created): the process's stdout goes to stderr until the JSON line is written to the real one
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 50 warm-up steps (≈40 ms) because the first ≈20 ms of work after the card has been idle run
    # ≈4 % slower (measured: scatter 0.48 ms with 3 warm-up steps, 0.46 ms with 50 or 300), then 100 timed steps
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--config", type=int, choices=[2, 4, 5], default=2,
                    help="2: BASELINE configs[1] (the metric's configuration); 4 / 5: configs[3] / configs[4], a rank's share per GPU")
    ap.add_argument("--mode", choices=["auto", "exchange", "drop"], default="auto",
                    help="--config 4 / 5: exchange rounds over RCCL (auto) or one owner's share with the others' records dropped")
    ap.add_argument("--reserve-cus", type=int, default=0,
                    help="--config 4 / 5: compute units the counting leaves to the collectives (shk_config.reserve_cus; 0 = the library's default)")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU per step (default: 1 M for --config 2, 125 M for 4 / 5)")
    ap.add_argument("--batches", type=int, default=4, help="--config 2: distinct resident batches the steps rotate over")
    ap.add_argument("--k", type=int, default=21)
    ap.add_argument("--genome", type=int, default=3_000_000)
    ap.add_argument("--chunks", type=int, default=1)
    ap.add_argument("--histo-max", type=int, default=10000)
    ap.add_argument("--path", choices=["auto", "direct", "paged"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra timing points (host / file / config 3 …)")
    ap.add_argument("--ramp-ms", type=float, default=250.0, help="untimed clock ramp before the warm-up steps")
    ap.add_argument("--cpu-sample-reads", type=int, default=0,
                    help="reads for the CPU baseline (0 = the whole step batch of rank 0)")
    args = ap.parse_args()
    if args.config != 2:
        return owner_config(args, real_stdout)
    args.reads = args.reads or 1_000_000

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
    n_gpus = dist.get_world_size() if dist is not None else 1   # (as the communicator saw it)
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
    # kernel durations: HIP events recorded by libshk on its own stream around every launch of every 4th step of
    # the timed region (SHK_FLAG_TIMING_SAMPLED: bracketing every launch of every step costs ≈3 % of the step)
    flags = 0 if os.environ.get("SHK_BENCH_NO_TIMING") else sa.FLAG_TIMING | sa.FLAG_TIMING_SAMPLED  # (experiment hook)
    timed_steps = -(-args.steps // 4)  # steps 1, 5, 9, … after reset_timings
    if args.path == "direct":
        flags |= sa.FLAG_FORCE_DIRECT
    elif args.path == "paged":
        flags |= sa.FLAG_FORCE_PAGED
    eng = sa.KmerEngine(args.k, args.chunks, args.histo_max, device=dev,
                        capacity_hint=genome, flags=flags)

    # NB distinct batches resident in HBM: batch b of rank r is reads [(r·NB + b)·n, (r·NB + b + 1)·n) of the generator
    NB = max(1, args.batches)
    d_bases = [torch.empty(n_reads * L, dtype=torch.uint8, device=f"cuda:{dev}") for _ in range(NB)]
    d_offsets = torch.empty(n_reads + 1, dtype=torch.int64, device=f"cuda:{dev}")
    batch_first = [(rank * NB + b) * n_reads for b in range(NB)]
    for b in range(NB):
        eng.synth_reads_device(spec, batch_first[b], n_reads, d_bases[b].data_ptr(), d_offsets.data_ptr())
    n_bases = n_reads * L
    step_no = [0]

    if dist is not None:
        from sharkmer_amd.dist import DistCounter
        dc = DistCounter(eng, dist, device=dev)

    def step():
        b = step_no[0] % NB
        step_no[0] += 1
        eng.reset()
        eng.set_read_index(batch_first[b])
        eng.ingest_reads_device(d_bases[b].data_ptr(), d_offsets.data_ptr(), n_reads, n_bases)
        if dist is not None:
            return dc.finalize_histograms()
        eng.finalize()
        return eng.histograms(h_out)

    h_out = np.empty((args.chunks, args.histo_max + 2), dtype=np.uint64)  # the caller's histogram array, filled every step
    ramp_steps = 0
    if dist is not None:
        # every step ends in collectives: all ranks must run the SAME number of ramp steps
        for _ in range(max(int(args.ramp_ms / 1.2), 0)):
            step()
            ramp_steps += 1
    else:
        t_r = time.perf_counter()
        while (time.perf_counter() - t_r) * 1e3 < args.ramp_ms and ramp_steps < 2000:  # untimed: clocks up
            step()
            ramp_steps += 1
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
        # the record a k-mer is between the partition and the page pass: 4 bytes when the mixed key's bits below the
        # level-1 fan-out (1024) fit a word — 2k − 10 ≤ 32, k ≤ 21 — else 8.  The page pass of a job that counts one
        # batch into an empty table is the FRESH one: it reads no page, it writes every page once
        rec_bytes = 4 if 2 * args.k - 10 <= 32 else 8
        def alg_bytes(name, lps):
            return {
                "scatter": (n_bases * 1 + n_kmers * rec_bytes) / lps,   # bases read + one record per k-mer written, at the width in use (what the kernel must move by design; §8d's 8-byte figure: achieved_sec8d_GBps)
                "pages": n_kmers * rec_bytes / lps + cap * (8 + 4 * lanes),   # records read at the width in use + every page written out once
                "histo": cap * (8 + 4 * lanes),                    # one table scan per histogram emit
                "direct": (n_bases * 1 + n_kmers * 16 + nd * 8) / lps,
                "scan": n_bases * 1 / lps,
            }.get(name)
        # HBM bytes per launch from the PMC passes of THIS round's build (tools/profile_round.sh →
        # tools/collect_profile.py → profiles/<round>_traffic.json, FETCH_SIZE doubled per the gfx950 note of
        # MI355X_MICROARCH.md): a committed measurement of the same command, named in the line; only the newest
        # round's file counts (an older one would be a stale constant), and only for the workload it was taken on
        import glob
        traffic_kernels, traffic_src = {}, None
        for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")), reverse=True):
            tj = json.load(open(tpath))
            if tj.get("reads") == n_reads and tj.get("k") == args.k and tj.get("batches", 1) == NB:
                traffic_kernels = tj.get("kernels", {})
                traffic_src = os.path.relpath(tpath, ROOT)
            break
        per_kernel = {}
        for name, (ms, launches) in tim.items():
            lps = launches / timed_steps
            if launches and alg_bytes(name, lps) is not None:
                avg = ms / launches
                ab = alg_bytes(name, lps)
                phys = traffic_kernels.get(name, {}).get("hbm_bytes_per_launch")
                # (SURVEY.md §8d prices a k-mer record at 8 bytes whatever travels: the same time against that figure)
                ab8 = (n_bases * 1 + n_kmers * 8) / lps if name == "scatter" else (n_kmers * 8 / lps + cap * (8 + 4 * lanes) if name == "pages" else ab)
                per_kernel[name] = {"avg_launch_ms": round(avg, 4), "launches_per_step": lps,
                                    "alg_bytes_per_launch": int(ab),
                                    "achieved_GBps": round(ab / (avg * 1e-3) / 1e9, 1),           # bytes the kernel must move by design ÷ time
                                    "achieved_sec8d_GBps": round(ab8 / (avg * 1e-3) / 1e9, 1),    # §8d's 8-byte-record bytes ÷ the same time
                                    "hbm_bytes_per_launch": phys,
                                    "physical_GBps": round(phys / (avg * 1e-3) / 1e9, 1) if phys else None}  # PMC bytes ÷ time
        hot = [k_ for k_ in ("direct", "scatter", "pages") if k_ in per_kernel]
        dom = max(hot, key=lambda k_: per_kernel[k_]["avg_launch_ms"] * per_kernel[k_]["launches_per_step"]) if hot else None
        roof = None
        if dom:
            d = per_kernel[dom]
            traffic = d["hbm_bytes_per_launch"]
            roof = {"bound": "hbm", "kernel": dom, "achieved": d["achieved_GBps"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(d["achieved_GBps"] / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "traffic_source": traffic_src if traffic else None, "alg_bytes_per_launch": d["alg_bytes_per_launch"],
                    "avg_launch_ms": d["avg_launch_ms"], "physical_GBps": d["physical_GBps"], "achieved_sec8d_GBps": d["achieved_sec8d_GBps"],
                    "what": "bytes the dominant kernel must move by design (records at the width in use) ÷ its average launch time; "
                            "achieved_sec8d_GBps prices a record at SURVEY §8d's 8 bytes; roofline_path is the whole step against §8d"}
        # the whole counting path against SURVEY.md §8d's B_alg (1 B/base + 16 B/k-mer + 8 B/distinct
        # + one table scan per emit), over the summed device time of its kernels
        path_ms = sum(ms for name, (ms, _) in tim.items()
                      if name in ("mark", "scan", "direct", "pcount", "pscan", "scatter", "pages", "histo")) / timed_steps
        b_alg = n_bases * 1 + n_kmers * 16 + nd * 8 + cap * (8 + 4 * lanes)
        path_roof = {"alg_bytes_per_step": int(b_alg), "device_ms_per_step": round(path_ms, 4),
                     "achieved_GBps": round(b_alg / (path_ms * 1e-3) / 1e9, 1) if path_ms else None,
                     "frac": round(b_alg / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if path_ms else None,
                     # the same bytes over the DRIVER-style clock (launch gaps and the host round trip included)
                     "frac_driver_clock": round(b_alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 5)}
        cpu = cpu_all = None
        if not args.no_cpu_baseline:
            from oracle import oracle as orc
            ns = args.cpu_sample_reads or n_reads
            last_first = batch_first[(step_no[0] - 1) % NB]   # the batch the last timed step counted (rank 0)
            hb, ho = sa.synth_reads(spec, last_first, ns)
            t1 = time.perf_counter()
            ref = orc.run_batch(hb, ho, args.k, args.chunks, args.histo_max)
            cdt = time.perf_counter() - t1
            cpu = {"value": round(ns * L / cdt / 1e9, 5), "unit": "Gbases/s", "cores": 1,
                   "kind": "port",
                   "sample": f"{ns} reads x {L} bp: the batch of the last timed step (rank 0 shard, reads {last_first}..), "
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
                                   f"step = reset + count + histogram emit, input resident in HBM, "
                                   f"steps rotate over {NB} distinct batches",
                       "reads_per_gpu": n_reads, "batches": NB, "k": args.k, "chunks": args.chunks,
                       "genome": genome, "path": args.path},
            "roofline": roof, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
            "roofline_path": path_roof, "kernels": per_kernel,
            "kernels_ms_per_step": {k_: round(v[0] / timed_steps, 4) for k_, v in tim.items()},
            "kernel_timing": f"HIP events (libshk, its own stream) around every launch of {timed_steps} of the {args.steps} timed steps (every 4th)",
            "table": {"capacity": cnt["table_capacity"], "n_unique": cnt["n_unique_kmers"],
                      "n_grows": cnt["n_grows"], "n_spilled": cnt["n_spilled"]},
            "ramp_steps": ramp_steps,
        }
        if world == 1 and dist is None and not args.no_extras:
            eng.close()
            d_bases.clear()
            del d_offsets
            torch.cuda.empty_cache()
            out["extras"] = extras(sa, torch, dev)
    else:
        out = None
    if world > 1 and not args.no_extras:
        # N > 1: the same ranks then run one rank's share each of BASELINE configs[3] (1 B reads of a 3 Gb genome over
        # 8 GPUs; key-space-partitioned ingest, the records exchanged by owner every round) — the multi-GPU
        # configuration the north star names, on record beside the metric's own configuration
        eng.close()
        d_bases.clear()
        torch.cuda.empty_cache()
        import copy
        a4 = copy.copy(args)
        a4.steps, a4.warmup, a4.reads, a4.no_cpu_baseline = min(args.steps, 3), 2, 0, True   # (two warm-up jobs: owner_run's note)
        c4 = owner_run(a4, 4, "exchange", dist_ready=True)
        if rank == 0:
            out["extras"] = {"config4": {key: c4[key] for key in ("value", "ms_per_step", "n_gpus", "config", "kernels_ms_per_step", "exchange",
                                                                     "kmers_as_expected", "histogram_rows_sum_to_distinct", "table")}}
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
