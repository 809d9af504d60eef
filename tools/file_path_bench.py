#!/usr/bin/env python3
"""Timing point (iii) of SURVEY.md §8d: FASTQ(.gz) file → histogram through shk_run_files (the C++
reader restating read_fastq + the counting path + the writers), host parse inclusive."""
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sharkmer_amd as sa  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = 150
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
bases, offsets = sa.synth_reads(spec, 0, n_reads)
tmp = tempfile.mkdtemp(prefix="shk_file_")
plain = os.path.join(tmp, "reads.fastq")
rec = np.empty((n_reads, 2 * L + 10), dtype=np.uint8)   # "@r\n" + seq + "\n+\n" + qual + "\n"
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
rec[:, 3:3 + L] = bases.reshape(n_reads, L)
rec[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 6 + L:6 + 2 * L] = ord("I")
rec[:, 6 + 2 * L] = ord("\n")
rec = rec[:, :7 + 2 * L]
rec.tofile(plain)
gz = plain + ".gz"
with open(plain, "rb") as f, gzip.open(gz, "wb", compresslevel=1) as g:
    g.write(f.read())
out = {"reads": n_reads}
for name, path in (("plain", plain), ("gzip", gz)):
    t0 = time.perf_counter()
    st = sa.run_files([path], k=21, chunks=1, histo_max=10000, sample="s", outdir=tmp)
    dt = time.perf_counter() - t0
    out[name + "_Gbases_per_s"] = round(n_reads * L / dt / 1e9, 3)
    out[name + "_file_MB"] = round(os.path.getsize(path) / 1e6, 1)
print(json.dumps(out))
