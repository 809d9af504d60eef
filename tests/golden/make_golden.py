#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (data only):

  reads_main.fastq.gz   2300 reads, ragged lengths 0..90 incl. N, reads < k, one empty read
  reads_part2.fastq     210 reads (a second input file: batches of 1000 span the file boundary)
  reads_crlf.fastq      40 reads with CRLF line ends
  golden_k21_c3.histo / .final.histo / .stats.json   expected outputs for
        -k 21 --chunks 3 --histo-max 50 reads_main.fastq.gz reads_part2.fastq
  golden_k15_c0.stats.json                              expected totals for -k 15 --chunks 0

The reference (Rust) cannot run in this image, so the expected outputs come from the CPU
oracle (oracle/shk_oracle.c), itself pinned by the reference's known-answer tests, and are
cross-checked here against the independent numpy sort-based oracle before being written.
Run from the repo root:  python tests/golden/make_golden.py
"""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402


def make_reads(rng, n, max_len, genome):
    out = []
    for i in range(n):
        L = int(rng.integers(0, max_len))
        if i == 7:
            L = 0
        s = int(rng.integers(0, len(genome) - max_len))
        seq = bytearray(genome[s:s + L])
        for j in range(L):
            u = rng.random()
            if u < 0.004:
                seq[j] = ord("N")
            elif u < 0.012:
                seq[j] = b"ACGT"[int(rng.integers(0, 4))]
        out.append(bytes(seq))
    return out


def fastq_text(seqs, tag, eol="\n"):
    parts = []
    for i, s in enumerate(seqs):
        parts.append(f"@{tag}.{i} len={len(s)}{eol}{s.decode()}{eol}+{eol}{'I' * len(s)}{eol}")
    return "".join(parts).encode()


def main():
    rng = np.random.default_rng(20261003)
    genome = bytes(b"ACGT"[c] for c in rng.integers(0, 4, size=6000))
    main_reads = make_reads(rng, 2300, 90, genome)
    part2 = make_reads(rng, 210, 90, genome)
    crlf = make_reads(rng, 40, 60, genome)
    with gzip.GzipFile(os.path.join(HERE, "reads_main.fastq.gz"), "wb", mtime=0) as f:
        f.write(fastq_text(main_reads, "main"))
    open(os.path.join(HERE, "reads_part2.fastq"), "wb").write(fastq_text(part2, "p2"))
    open(os.path.join(HERE, "reads_crlf.fastq"), "wb").write(fastq_text(crlf, "crlf", eol="\r\n"))

    def run(k, chunks, histo_max, files):
        r = orc.Run(k, chunks, histo_max)
        for fn in files:
            r.read_fastq(os.path.join(HERE, fn))
        r.finish()
        return r

    files = ["reads_main.fastq.gz", "reads_part2.fastq"]
    r = run(21, 3, 50, files)
    # cross-check with the code-independent oracle
    allseq = main_reads + part2
    bases = np.frombuffer(b"".join(allseq), dtype=np.uint8)
    offs = np.cumsum([0] + [len(s) for s in allseq])
    assert np.array_equal(r.histograms(), orc.sort_count_histogram(bases, offs, 21, 3, 50))
    r.write_histo(os.path.join(HERE, "golden_k21_c3.histo"))
    r.write_final_histo(os.path.join(HERE, "golden_k21_c3.final.histo"))
    json.dump(r.stats, open(os.path.join(HERE, "golden_k21_c3.stats.json"), "w"), indent=1, sort_keys=True)
    r0 = run(15, 0, 50, files)
    json.dump(r0.stats, open(os.path.join(HERE, "golden_k15_c0.stats.json"), "w"), indent=1, sort_keys=True)
    rc = run(11, 2, 20, ["reads_crlf.fastq"])
    rc.write_histo(os.path.join(HERE, "golden_crlf_k11_c2.histo"))
    print("wrote fixtures:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
