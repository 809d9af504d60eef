#!/usr/bin/env python3
"""Generates tests/golden/reads_two_members.fastq.gz (data only): TWO gzip members back to back — 5 reads in the
first, 7 in the second — the shape `cat a.fastq.gz b.fastq.gz` or bgzip produces.  The reference opens a .gz with
flate2::read::GzDecoder (src/io.rs:618-621), which decodes the FIRST member and then reports end of stream, so the
expected read count of this file is 5, not 12 (tests/test_frontend_semantics.py).
Run from the repo root:  python tests/golden/make_two_members.py
"""
import os
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def member(text: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, 31)   # gzip wrapper, mtime 0: byte-stable
    return co.compress(text) + co.flush()


def main():
    rng = np.random.default_rng(20261004)
    genome = "".join(rng.choice(list("ACGT"), size=400))

    def reads(n, tag):
        out = []
        for i in range(n):
            L = int(rng.integers(30, 61))
            s = int(rng.integers(0, len(genome) - L))
            out.append(f"@{tag}.{i}\n{genome[s:s + L]}\n+\n{'I' * L}\n")
        return "".join(out).encode()

    blob = member(reads(5, "first")) + member(reads(7, "second"))
    with open(os.path.join(HERE, "reads_two_members.fastq.gz"), "wb") as f:
        f.write(blob)
    print("wrote reads_two_members.fastq.gz:", len(blob), "bytes")


if __name__ == "__main__":
    main()
