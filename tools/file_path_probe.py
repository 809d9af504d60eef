#!/usr/bin/env python3
"""FASTQ file → histogram + output files through shk_run_files (§8d timing point iii), best of N, on a file made once:
usage: python3 tools/file_path_probe.py [reads] [reps]"""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402
import torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = 150
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
with sa.KmerEngine(21, 1, 10000, capacity_hint=3_000_000) as eng:
    db = torch.empty(n * L, dtype=torch.uint8, device="cuda:0")
    do = torch.empty(n + 1, dtype=torch.int64, device="cuda:0")
    eng.synth_reads_device(spec, 0, n, db.data_ptr(), do.data_ptr())
    eng.sync()
    bases = db.cpu().numpy()
    del db, do
tmp = tempfile.mkdtemp(prefix="shk_fp_")
rec = np.empty((n, 2 * L + 7), dtype=np.uint8)
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
rec[:, 3:3 + L] = bases.reshape(n, L)
rec[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 6 + L:6 + 2 * L] = ord("I")
rec[:, 6 + 2 * L] = ord("\n")
plain = os.path.join(tmp, "reads.fastq")
rec.tofile(plain)
out = {}
for mode in ("packed", "ascii"):
    if mode == "ascii":
        os.environ["SHK_RUN_ASCII"] = "1"
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        sa.run_files([plain], k=21, chunks=1, histo_max=10000, sample="s", outdir=tmp, capacity_hint=3_000_000)
        ts.append(time.perf_counter() - t0)
        if os.environ.get("SHK_TRACE"):
            print(f"[py] run_files wall {ts[-1] * 1e3:.1f} ms", file=sys.stderr)
    out[mode] = {"Gbases_per_s_best": round(n * L / min(ts) / 1e9, 2), "all_ms": [round(t * 1e3, 1) for t in ts]}
print(json.dumps(out))
