#!/usr/bin/env python3
"""Quality check of the table hash (shk::hash64 in sharkmer_amd/csrc/shk_device.hip.h): page
occupancy spread and (page, home-slot) collisions against a Poisson process, on random,
AT-rich, tandem-repeat and sequential keys.  CPU only (numpy restatement of the device formula)."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)
A, B, C = (np.uint64(0x9E3779B1 & 0xFFFFFF), np.uint64(0x85EBCA77 & 0xFFFFFF), np.uint64(0xC2B2AE3D & 0xFFFFFF))


def hash64(key, fin=True):
    c0 = key & np.uint64(0xFFFFFF)
    c1 = (key >> np.uint64(24)) & np.uint64(0xFFFFFF)
    c2 = key >> np.uint64(48)
    h = (c0 * A + c1 * B + c2 * C) & M32
    if fin:
        h ^= h >> np.uint64(15)
        h = (h * np.uint64(0x2C1B3C6D)) & M32
        h ^= h >> np.uint64(12)
    return h


def canon(codes, k):
    n = len(codes) - k + 1
    f = np.zeros(n, dtype=np.uint64)
    r = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        b = codes[j:j + n].astype(np.uint64)
        f = (f << np.uint64(2)) | b
        r |= (np.uint64(3) - b) << np.uint64(2 * j)
    return np.unique(np.minimum(f, r))


def stats(keys, name, log_pages=11):
    for nm, fin in (("final", True), ("no-finaliser", False)):
        h = hash64(keys, fin)
        page = (h >> np.uint64(32 - log_pages)).astype(np.int64)
        cnt = np.bincount(page, minlength=1 << log_pages)
        slot = ((h >> np.uint64(20 - log_pages)) & np.uint64(4095)).astype(np.int64)
        u = len(np.unique(page * 4096 + slot))
        lam = len(keys) / (1 << log_pages)
        print(f"{name:12s} {nm:13s} n={len(keys):8d} mean/page={lam:7.0f} max={cnt.max():6d} "
              f"std={cnt.std():6.1f} (poisson {lam ** 0.5:5.1f}) distinct(page,slot)/n={u / len(keys):.4f}")


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    stats(canon(rng.integers(0, 4, size=3_000_000), 21), "random k21")
    unit = rng.integers(0, 4, size=7)
    g2 = np.tile(unit, 300000)
    mut = rng.random(len(g2)) < 0.02
    g2[mut] = rng.integers(0, 4, size=mut.sum())
    stats(canon(g2, 21), "tandem7")
    stats(canon(rng.choice(4, size=2_000_000, p=[0.45, 0.05, 0.05, 0.45]), 21), "AT-rich")
    stats(np.arange(3_000_000, dtype=np.uint64) * np.uint64(4), "sequential")
    stats(canon(rng.integers(0, 4, size=2_000_000), 31), "random k31")
