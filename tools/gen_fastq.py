#!/usr/bin/env python3
"""Writes a synthetic FASTQ of N 150-bp reads (and, optionally, P gzip parts of it) for front-end timing."""
import gzip, sys, numpy as np
n = int(sys.argv[1]); out = sys.argv[2]; parts = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L = 150
rng = np.random.default_rng(1)
rec = np.empty((n, 2 * L + 7), dtype=np.uint8)
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
rec[:, 3:3 + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
rec[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 6 + L:6 + 2 * L] = ord("I")
rec[:, 6 + 2 * L] = ord("\n")
rec.tofile(out)
for i in range(parts):
    a, b = n * i // parts, n * (i + 1) // parts
    with gzip.open(f"{out}.part{i}.gz", "wb", compresslevel=1) as g:
        g.write(rec[a:b].tobytes())
