"""A randomized sweep of the engine against the oracle: configuration (k, chunk lanes, histo_max, flags, capacity hint),
the test hooks that pick the engine's routes (partition levels, deferral budget, record width, slice size, host
packing, …), the shape of the input (ragged reads, empty and all-N reads, long reads, skew) and the way it is handed
over (host ASCII in several calls, host packed, device buffers) are all drawn from a seed —
what the parametrized tests of test_gpu_parity.py fix by hand, in combinations nobody wrote down.  Every case:
histograms, counters and the exported table, bit for bit.  SHK_FUZZ_SEEDS: how many (default 60); SHK_FUZZ_FIRST: the
first seed (a failing case prints its seed and its draw)."""
import os

import numpy as np
import pytest
import torch

import sharkmer_amd as sa

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("SHK_FUZZ_SEEDS", "60"))
FIRST = int(os.environ.get("SHK_FUZZ_FIRST", "0"))

HOOKS = {   # name → values a case may pin (None: leave the engine's own choice)
    "SHK_LEVEL1_LOG": [None, None, "0", "2", "3", "5", "8", "10"],
    "SHK_TWO_LEVEL_MIN_PAGES": [None, None, "4", "16"],
    "SHK_DEFER_BUDGET": [None, None, "20000", "150000", "1000000"],
    "SHK_REC32": [None, None, None, "0"],
    "SHK_SCATTER32_LDS": [None, None, None, "0"],
    "SHK_ALL_LANES": [None, None, "0", "1"],
    "SHK_NO_FRESH": [None, None, None, "1"],
    "SHK_DEFER": [None, None, None, "0"],
    "SHK_XL": [None, None, None, "0"],
    "SHK_SLICE_KB": [None, "16", "64", "300"],
    "SHK_HOST_PACK": [None, "0", "1"],
    "SHK_NMASK_SPARSE": [None, None, "0"],
    "SHK_FLUSH_GROUP_PAGES": [None, None, None, "5"],
    "SHK_WIDE_WINDOW": [None, None, "0"],
}


def draw_reads(rng):
    shape = rng.choice(["ragged", "uniform", "long", "skew", "tiny"])
    p_n = float(rng.choice([0.0, 0.002, 0.03]))
    if shape == "ragged":
        n = int(rng.integers(200, 9000))
        lens = rng.integers(0, int(rng.choice([40, 160, 400])), size=n)
        lens[rng.random(n) < 0.05] = 0
    elif shape == "uniform":
        n = int(rng.integers(500, 12000))
        lens = np.full(n, int(rng.choice([36, 100, 150, 151])))
    elif shape == "long":
        n = int(rng.integers(3, 40))
        lens = rng.integers(5_000, 90_000, size=n)
    elif shape == "skew":
        n = int(rng.integers(500, 5000))
        lens = np.full(n, 150)
    else:
        n = int(rng.integers(1, 30))
        lens = rng.integers(0, 50, size=n)
    total = int(lens.sum())
    genome = rng.integers(0, 4, size=int(rng.choice([2_000, 30_000, 400_000])) + 400)
    bases = np.empty(total, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    at = 0
    for L in lens.tolist():
        if L:
            if shape == "skew" and rng.random() < 0.4:
                unit = lut[rng.integers(0, 4, size=int(rng.integers(1, 4)))]
                bases[at:at + L] = np.resize(unit, L)
            elif L <= len(genome) - 1 and rng.random() < 0.8:
                s = int(rng.integers(0, len(genome) - L))
                seg = genome[s:s + L]
                if rng.random() < 0.5:
                    seg = (3 - seg)[::-1]
                bases[at:at + L] = lut[seg]
            else:
                bases[at:at + L] = lut[rng.integers(0, 4, size=L)]
        at += L
    if p_n and total:
        bases[rng.random(total) < p_n] = ord("N")
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    return shape, bases, offsets


@pytest.mark.parametrize("seed", range(FIRST, FIRST + N_SEEDS))
def test_random_configuration_against_the_oracle(orc, monkeypatch, seed):
    rng = np.random.default_rng(10_000 + seed)
    k = int(rng.choice([1, 2, 5, 9, 11, 13, 15, 16, 17, 19, 21, 21, 21, 22, 23, 25, 27, 29, 31, 31]))
    chunks = int(rng.choice([0, 1, 1, 2, 3, 7, 10, 16, 17, 40, 100, 129]))   # (up to 128 lanes: the paged passes; beyond: the atomics)
    histo_max = int(rng.choice([1, 5, 50, 300]))
    flags = int(rng.choice([0, 0, 0, sa.FLAG_FORCE_DIRECT, sa.FLAG_FORCE_PAGED, sa.FLAG_DEFER_ERRORS, sa.FLAG_TIMING]))
    hint = int(rng.choice([0, 0, 20_000, 400_000, 1_100_000, 4_200_000]))
    hooks = {}
    for name, values in HOOKS.items():
        v = values[int(rng.integers(0, len(values)))]
        if v is not None:
            hooks[name] = v
            monkeypatch.setenv(name, v)
    shape, bases, offsets = draw_reads(rng)
    n = len(offsets) - 1
    route = str(rng.choice(["host", "host", "packed", "device"]))
    n_cuts = int(rng.integers(0, 5))
    cuts = [0] + sorted(int(x) for x in rng.integers(0, n + 1, size=n_cuts)) + [n]
    draw = dict(seed=seed, k=k, chunks=chunks, histo_max=histo_max, flags=flags, hint=hint, hooks=hooks, shape=shape,
                n_reads=n, n_bases=int(offsets[-1]), route=route, cuts=cuts)
    ref = orc.run_batch(bases, offsets, k, chunks, histo_max)
    with sa.KmerEngine(k, chunks, histo_max, capacity_hint=hint, flags=flags) as eng:
        keep = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            o = offsets[a:b + 1]
            if route == "host":
                eng.ingest_reads(bases, o)
            elif route == "packed":
                lo, hi = int(o[0]), int(o[-1])
                pk = sa.pack_reads(bases[lo:hi], o - o[0])
                eng.ingest_packed(pk)
            elif route == "device":
                lo, hi = int(o[0]), int(o[-1])
                db = torch.from_numpy(bases[lo:hi].copy()).cuda() if hi > lo else torch.zeros(1, dtype=torch.uint8).cuda()
                do = torch.from_numpy((o - o[0]).astype(np.int64)).cuda()
                keep += [db, do]
                eng.ingest_reads_device(db.data_ptr(), do.data_ptr(), b - a, hi - lo)
        eng.finalize()
        got = eng.histograms()
        cnt = eng.counters()
        keys, cnts = eng.export_table()
    assert np.array_equal(got, ref.histograms()), draw
    st = ref.stats
    for f in ("n_reads_ingested", "n_bases_read", "n_bases_ingested", "n_kmers_ingested", "n_unique_kmers", "n_hashed_kmers"):
        assert cnt[f] == st[f], (f, draw)
    rk, rc = ref.merged().export()
    assert np.array_equal(keys, rk) and np.array_equal(cnts, rc), draw
