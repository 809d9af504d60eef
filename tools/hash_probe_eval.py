#!/usr/bin/env python3
"""Linear-probing displacement of the k-mers of a random genome in the engine's paged table (page = top 10
bits, home bucket = next 11 bits, 4 slots per bucket) under candidate key mixes: a tag holds a key within 7
buckets of home; keys further out take the spill path.  Consecutive k-mers of a genome are related by
x' = 4x + b, which a purely multiplicative mix carries over (y' = 4y + bM): occupancy of neighbouring buckets
is then correlated along the genome — this counts what that costs."""
import numpy as np
M32 = np.uint64(0xC2B2AE35)
def canon(codes, k):
    n = len(codes) - k + 1
    f = np.zeros(n, dtype=np.uint64); r = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        b = codes[j:j + n].astype(np.uint64)
        f = (f << np.uint64(2)) | b
        r |= (np.uint64(3) - b) << np.uint64(2 * j)
    return np.unique(np.minimum(f, r))
def mixes(bits):
    mask = np.uint64((1 << bits) - 1); s = np.uint64(bits // 2)
    return {
        "mul": lambda x: (x * M32) & mask,
        "fold_mul": lambda x: ((x ^ (x >> s)) * M32) & mask,
        "mul_fold_mul": lambda x: ((((x * np.uint64(0x9E3779B1)) & mask) ^ (((x * np.uint64(0x9E3779B1)) & mask) >> np.uint64((bits + 1) // 2))) * np.uint64(0x85EBCA6B)) & mask,
    }
def displacement(y, bits, lp=10):
    page = (y >> np.uint64(bits - lp)).astype(np.int64)
    home = ((y >> np.uint64(bits - lp - 11)) & np.uint64(2047)).astype(np.int64) * 4
    order = np.lexsort((home, page))
    page, home = page[order], home[order]
    # positions under linear probing in home order: p_i = max(home_i, p_{i-1} + 1) within a page
    d = np.zeros(len(y), dtype=np.int64)
    start = 0
    bounds = np.flatnonzero(np.diff(page)) + 1
    for a, b in zip(np.r_[0, bounds], np.r_[bounds, len(y)]):
        h = home[a:b]
        # p_i - i is the running max of (h_i - i)
        idx = np.arange(b - a)
        p = np.maximum.accumulate(h - idx) + idx
        d[a:b] = p - h
    return d
rng = np.random.default_rng(1)
for G in (3_000_000, 6_000_000):
    keys = canon(rng.integers(0, 4, size=G), 21)
    for name, f in mixes(42).items():
        d = displacement(f(keys), 42)
        print(f"G={G} {name:13s} load {len(keys)/(1024*8192):.2f}  mean displacement {d.mean():.3f} slots  beyond home bucket {np.mean(d >= 4):.4f}  >= 7 buckets {int((d >= 28).sum())}  max {d.max()}")
