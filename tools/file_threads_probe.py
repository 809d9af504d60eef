"""FASTQ file path (plain + 8 gzip parts) under different reader pool sizes (SHK_FASTQ_THREADS), one box, interleaved
repetitions: what the container's CPU quota makes of them."""
import gzip
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "child":
    paths = sys.argv[3:]
    n = int(sys.argv[2])
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        sa.run_files(paths, k=21, chunks=1, histo_max=10000, sample="s", outdir=os.path.dirname(paths[0]), capacity_hint=3_000_000)
        ts.append(time.perf_counter() - t0)
    print(round(n * 150 / min(ts) / 1e9, 2))
    sys.exit(0)

n, L = 8_000_000, 150
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
bases, _ = sa.synth_reads(spec, 0, n)
tmp = tempfile.mkdtemp(prefix="shk_thr_")
rec = np.empty((n, 2 * L + 7), dtype=np.uint8)
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
rec[:, 3:3 + L] = bases.reshape(n, L)
rec[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 6 + L:6 + 2 * L] = ord("I")
rec[:, 6 + 2 * L] = ord("\n")
plain = os.path.join(tmp, "reads.fastq")
rec.tofile(plain)
parts = []
for i in range(8):
    a, b = n * i // 8 * (2 * L + 7), n * (i + 1) // 8 * (2 * L + 7)
    pth = os.path.join(tmp, f"part{i}.fastq.gz")
    with gzip.open(pth, "wb", compresslevel=1) as g:
        g.write(rec.reshape(-1)[a:b].tobytes())
    parts.append(pth)
del rec, bases
for rep in range(3):
    for thr in ("16/4", "24/4", "32/4", "24/8"):
        env = dict(os.environ)
        if thr:
            env["SHK_FASTQ_THREADS"], env["SHK_FASTQ_COPY_THREADS"] = thr.split("/")
        out = []
        for paths in ([plain],):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n)] + paths, env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            out.append(r.stdout.decode().strip())
        print(f"threads={thr or 'default'}: plain {out[0]}", flush=True)
