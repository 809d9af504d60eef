"""Synthetic reads: implicit uniform random genome
base(j) = splitmix64(seed_g + j) & 3; read i starts at splitmix64(seed_r + 2i) %
(G - L + 1) on the strand given by bit 63 of splitmix64(seed_r + 2i + 1);
optional substitutions / N for the parity variants.  The device generator
(k_synth in csrc/shk_device.hip.h) produces the same bytes; tests check that."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

SEED_GENOME = 0x5EED0001
SEED_READS = 0x5EED0002
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


@dataclass
class SynthSpec:
    genome_len: int
    read_len: int = 150
    seed_genome: int = SEED_GENOME
    seed_reads: int = SEED_READS
    sub_per_64k: int = 0   # substitution errors per 65536 bases
    n_per_64k: int = 0     # N per 65536 bases


_LUT = np.frombuffer(b"ACGT", dtype=np.uint8)


def synth_reads(spec: SynthSpec, first_read: int, n_reads: int):
    """Returns (bases u8[n_reads*L], offsets u64[n_reads+1])."""
    L = spec.read_len
    with np.errstate(over="ignore"):
        i = np.arange(first_read, first_read + n_reads, dtype=np.uint64)
        h1 = splitmix64(np.uint64(spec.seed_reads) + np.uint64(2) * i)
        h2 = splitmix64(np.uint64(spec.seed_reads) + np.uint64(2) * i + np.uint64(1))
        start = h1 % np.uint64(spec.genome_len - L + 1)
        rc = (h2 >> np.uint64(63)).astype(bool)
        j = np.arange(L, dtype=np.uint64)
        gp = np.where(rc[:, None], start[:, None] + (np.uint64(L - 1) - j)[None, :],
                      start[:, None] + j[None, :])
        b = (splitmix64(np.uint64(spec.seed_genome) + gp) & np.uint64(3)).astype(np.uint8)
        b = np.where(rc[:, None], 3 - b, b).astype(np.uint8)
        ch = _LUT[b]
        if spec.sub_per_64k or spec.n_per_64k:
            e = splitmix64(np.uint64(spec.seed_reads ^ 0xE44044) + i[:, None] * np.uint64(L) + j[None, :])
            u = (e & np.uint64(0xFFFF)).astype(np.int64)
            d = (1 + ((e >> np.uint64(16)) % np.uint64(3))).astype(np.uint8)
            sub = _LUT[(b + d) & 3]
            is_n = u < spec.n_per_64k
            is_sub = (~is_n) & (u < spec.n_per_64k + spec.sub_per_64k)
            ch = np.where(is_n, np.uint8(ord("N")), np.where(is_sub, sub, ch)).astype(np.uint8)
    bases = np.ascontiguousarray(ch.reshape(-1))
    offsets = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L))
    return bases, offsets
