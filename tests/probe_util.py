"""Exact probe-set checks at sizes the table-based oracle cannot run (tests/test_gpu_*.py): ~10^5 k-mers
taken from sampled reads are counted over ALL reads with the oracle's extractor (orc_probe_count:
kmers_from_ascii, encoding.rs:332-371; chunk lane of read i = (i / 1000) % n_lanes, io.rs:340-361) on all
host cores, and the engine's point lookups must equal those counts exactly — a wrong count that
preserves the histogram sums (two k-mers swapped across pages) cannot pass."""
import concurrent.futures as cf
import os

import numpy as np


class ProbeChecker:
    def __init__(self, orc, k, n_lanes, read_len, n_probes=100_000, threads=None):
        self.orc, self.k, self.n_lanes, self.L = orc, k, max(n_lanes, 1), read_len
        self.n_probes = n_probes
        self.probes = None
        self._sample = []
        self.threads = threads or max(1, min((os.cpu_count() or 2) - 1, 30))
        self._pool = cf.ThreadPoolExecutor(self.threads)
        self._jobs = []
        self._parts = []

    def _fetch(self, eng, d_bases, n_reads):
        """Device batch → host copy (pinned staging through torch)."""
        import torch
        t = eng._raw_tensor(d_bases, n_reads * self.L, "|u1")
        return t.cpu().numpy()

    def add_sample(self, host_bases, n_reads):
        """k-mers of these reads become probe candidates (call before the first count)."""
        kms = []
        for i in range(n_reads):
            kms += self.orc.kmers_from_ascii(host_bases[i * self.L:(i + 1) * self.L].tobytes(), self.k)
        self._sample.append(np.array(kms, dtype=np.uint64))

    def freeze(self, extra=None):
        allk = np.unique(np.concatenate(self._sample + ([np.asarray(extra, dtype=np.uint64)] if extra is not None else [])))
        if len(allk) > self.n_probes:
            allk = allk[np.sort(np.random.default_rng(7).choice(len(allk), self.n_probes, replace=False))]
        self.probes = np.ascontiguousarray(allk, dtype=np.uint64)
        return self.probes

    def count_async(self, host_bases, n_reads, first_read_index):
        """Count the probes over one host batch on the pool (the batch array must stay alive: kept here)."""
        T = self.threads
        cuts = [n_reads * t // T for t in range(T + 1)]
        L = self.orc.lib()
        offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(self.L)

        def work(t):
            a, b = cuts[t], cuts[t + 1]
            part = np.zeros((self.n_lanes, len(self.probes)), dtype=np.uint64)
            if b > a:
                rc = L.orc_probe_count(host_bases.ctypes.data, offsets[a:b + 1].ctypes.data, b - a, self.k,
                                       first_read_index + a, self.n_lanes, self.probes.ctypes.data, len(self.probes),
                                       part.ctypes.data)
                assert rc == 0, rc
            return part
        self._jobs.append((host_bases, offsets, [self._pool.submit(work, t) for t in range(T)]))
        # bound the host memory held by batches in flight
        while len(self._jobs) > 3:
            self._drain_one()

    def _drain_one(self):
        _, _, futs = self._jobs.pop(0)
        for f in futs:
            self._parts.append(f.result())
            if len(self._parts) > 1:
                self._parts = [self._parts[0] + self._parts[1]]

    def result(self):
        """(n_lanes, n_probes) u64 occurrence counts over everything counted so far."""
        while self._jobs:
            self._drain_one()
        return self._parts[0] if self._parts else np.zeros((self.n_lanes, len(self.probes)), dtype=np.uint64)

    def merged(self):
        """What get_count of the merged table must return: sequential saturating adds of the lanes
        (counting.rs:82-92) = the clamped sum."""
        return np.minimum(self.result().sum(axis=0), np.uint64(0xFFFFFFFF)).astype(np.uint32)

    def close(self):
        self._pool.shutdown(wait=True)
