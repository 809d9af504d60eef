"""Multi-GPU counting: one process per GPU, reads sharded across ranks, per-rank count tables
merged by owner over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests), histograms all-reduced.

What it reproduces: the reference merges per-chunk tables into one with
`KmerCounts::extend{,_with_histogram}` (counting.rs:157-202) and reads the histogram off the
merged table (io.rs:1020-1028).  Here the "tables to merge" are the per-rank tables:

  1. all ranks agree on a table geometry (all_reduce MAX of the page count; smaller tables grow)
  2. rank r OWNS pages [r·P/W, (r+1)·P/W).  Every rank compacts the OCCUPIED entries of each
     owner's range (at load ≤ 1/2 the EMPTY slots would be most of the bytes), the per-owner
     entry counts are exchanged, and then "send every peer its entries" is ONE all_to_all_single
     per array with those split sizes — the fully connected xGMI mesh carries all W-1 transfers
     of a rank at once.  (SHK_DIST_DENSE=1: ship the ranges as they lie in the table instead.)
  3. every rank folds the received entries into its own range (saturating per lane)
  4. histogram scan restricted to the owned slice; bins are additive across disjoint key
     shards, so a dense all_reduce(SUM) of the (chunks × (histo_max+2)) u64 histogram and of
     the scalar totals finishes the job

The engine object is duck-typed (`table_geometry, reserve_pages, owner_counts,
compact_owner_tensors, merge_entry_tensors, table_tensors, merge_page_tensors, set_owned_pages,
finalize, histograms, counters`): the product passes
`KmerEngine` (HIP); the CPU tests pass a numpy stand-in to exercise the collective logic under
gloo with world_size 2.
"""
from __future__ import annotations

import os

import numpy as np
import torch


# No single message above this many bytes.  Measured on MI355X (RCCL 2.26.6 as shipped with torch 2.10 / ROCm 7.0,
# tools/rccl_a2a_probe.py): a send/recv pair of more than 2^30 bytes delivers only the first half of the buffer and
# reports success — all_to_all_single, all_to_all and batch_isend_irecv alike (a 1044 MiB message: the second 522 MiB
# never arrive; 1024 MiB and below are whole).  Everything here that can grow with the job goes in pieces well below.
MAX_MESSAGE_BYTES = int(os.environ.get("SHK_DIST_MAX_MESSAGE", 1 << 28))   # (the variable: a test hook)


def exchange_parts(dist, out: torch.Tensor, inp: torch.Tensor, world: int, rank: int, in_splits=None, out_splits=None,
                   skip_self: bool = False, in_pieces=None):
    """all_to_all_single — part s of `inp` (in_splits[s] elements; equal parts when None) goes to rank s, part s of
    `out` comes from rank s — that never hands the communicator a message above MAX_MESSAGE_BYTES: when every part is
    below it, ONE all_to_all_single; otherwise grouped send/recv pairs over slices of the parts (the fully connected
    mesh still carries a rank's W−1 transfers at once), the rank's own part by a device copy (skip_self: not at all —
    the caller uses it where it lies).  in_pieces: with UNEQUAL parts every rank must take the same route — the caller
    then says which (True: send/recv pairs), from a maximum it has reduced over the ranks."""
    if in_splits is None:
        per = inp.numel() // world
        assert inp.numel() == per * world and out.numel() == inp.numel()
        in_splits = out_splits = [per] * world
    esz = inp.element_size()
    if in_pieces is None:
        largest = max(max(in_splits), max(out_splits)) * esz
        if len(set(in_splits)) != 1 or list(in_splits) != list(out_splits):
            # UNEQUAL parts and nobody has agreed on the route: a rank only sees its own row and column of the W x W
            # part sizes, and a rank whose parts are all small must not call all_to_all_single while a pair with a
            # large part waits in send/recv pairs (a hang on gloo, undefined pairing on RCCL).  One small
            # all_reduce(MAX) makes the choice the same everywhere; callers that already hold a job-wide maximum
            # pass in_pieces and skip it.
            t = torch.tensor([largest], dtype=torch.int64, device=inp.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            largest = int(t.item())
        in_pieces = largest > MAX_MESSAGE_BYTES
    if not in_pieces or not hasattr(dist, "batch_isend_irecv"):
        if in_splits[0] * world == inp.numel() and len(set(in_splits)) == 1 and list(in_splits) == list(out_splits):
            dist.all_to_all_single(out, inp)   # (in-process test transports have nothing else)
        else:
            dist.all_to_all_single(out, inp, output_split_sizes=list(out_splits), input_split_sizes=list(in_splits))
        return
    step = MAX_MESSAGE_BYTES // esz
    in_at = [sum(in_splits[:i]) for i in range(world)]
    out_at = [sum(out_splits[:i]) for i in range(world)]
    assert in_splits[rank] == out_splits[rank]
    if not skip_self and in_splits[rank]:
        out[out_at[rank]:out_at[rank] + out_splits[rank]].copy_(inp[in_at[rank]:in_at[rank] + in_splits[rank]])
    ops = []
    for d in range(1, world):
        to, frm = (rank + d) % world, (rank - d) % world
        for c0 in range(0, in_splits[to], step):   # (both ends cut a part the same way: at multiples of `step`)
            ops.append(dist.P2POp(dist.isend, inp[in_at[to] + c0:in_at[to] + min(in_splits[to], c0 + step)], to))
        for c0 in range(0, out_splits[frm], step):
            ops.append(dist.P2POp(dist.irecv, out[out_at[frm] + c0:out_at[frm] + min(out_splits[frm], c0 + step)], frm))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def shard_batches(n_reads: int, rank: int, world: int, batch: int = 1000):
    """Round-robin assignment of whole 1000-read batches to ranks: returns
    [(first_read, n_reads_in_batch), ...] for `rank`.  Chunk membership is a function of the
    GLOBAL read index (io.rs:340-361), so sharding never changes it."""
    out = []
    n_batches = (n_reads + batch - 1) // batch
    for b in range(rank, n_batches, world):
        first = b * batch
        out.append((first, min(batch, n_reads - first)))
    return out


class OwnerCounter:
    """KEY-SPACE-PARTITIONED ingest ("alternative when local tables do not fit"; what
    BASELINE configs[4] — 10 cumulative lanes × a 3 Gb-genome load — needs: 48 B × 2^33 slots do not fit
    one GPU, 1/8 of them does).  Rank r's engine is an OWNER SHARE (n_owners = world, owner_id = r): it
    holds the k-mers whose owner bits (the top log2 W bits of the engine's key mix) equal r and nothing
    else, so no table is ever exchanged; what crosses the links instead, during ingest, is every batch's
    4-byte k-mer records grouped by owner.

    One ROUND (collective: every rank calls `round` the same number of times, with an empty batch when it
    has run out of reads):
      1. engine.xchg_scatter: validate + extract + level-1 partition of the rank's batch; owner o's
         records and their fill levels are one contiguous segment each;
      2. ONE all_to_all_single per array with equal splits (fully connected xGMI: the W-1 transfers of a
         rank run concurrently);
      3. engine.xchg_absorb on each of the W received segments (level-2 partition into the waiting
         (lane, page) regions; the page pass runs when enough has accumulated);
      4. a 2-word all_reduce(MAX) carries (somebody hit an invalid byte, somebody has foreign spills);
         foreign spills — records that overflowed a region on skewed input — are all-gathered and every
         rank inserts what it owns: exact for any input.
    Bytes on a link per round: one segment = regions × region_cap × 4 B ≈ 1.25 (× 1.5 with several
    lanes) × 4 B × 0.87 × batch bases / W — at configs[3]'s shape 137 MB per link and round against ≈ 1.3 ms of
    counting per round: the links are the bottleneck of a round taken step by step.  So the rounds are PIPELINED: with a
    CUDA engine the collectives run on a stream of their own, step 3 of round r is deferred until round r + 1 has
    scattered (the engine hands its exchange buffers out in turn; the receive buffers here come in two sets), and
    events order the two streams where they meet — round r's segments cross the links while the engine absorbs
    round r − 1 and scatters round r + 1.  The host waits twice per round: for the scatter's outcome — after it has
    launched the previous round's absorbs behind the scatter (the engine's two-call scatter) — and for the 2-word flags.

    finalize_histograms: every rank's histogram covers its share; bins are additive across disjoint key
    sets (KmerCounts::extend, counting.rs:157-166), so one all_reduce(SUM) of histogram + totals
    finishes the job (io.rs:1023-1028 semantics on the union of all reads).

    The engine is duck-typed (`xchg_scatter_tensors, xchg_absorb_tensors, xchg_spill_tensors,
    insert_tensors, xchg_spill_clear, set_read_index, finalize, histograms, counters, stream`)."""

    def __init__(self, engine, dist, device=None, round_bases: int = 1 << 28):
        self.eng, self.dist = engine, dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.device = device
        self.round_bases = int(round_bases)
        self._recv = [None, None]   # receive buffers of the rounds' exchanges, two sets taken in turn
        self._absorbed = [None, None]   # … and the events behind the absorbs that read them last
        self._pending = None   # the round whose exchange is in flight: absorbed at the next round or at finalize
        self._comm = None
        self._wide = None   # rounds take the wide route (engine.xchg_feasible() is False)
        self._two_calls = hasattr(engine, "xchg_scatter_begin_tensors") and not os.environ.get("SHK_DIST_ONE_CALL_SCATTER")
        self._trace = [0.0, 0.0, 0.0, 0] if os.environ.get("SHK_DIST_TRACE") else None
        self.n_rounds = 0
        self.n_foreign_rounds = 0   # rounds in which somebody had foreign spills to hand on
        self.wire_bytes = 0

    def _dev(self, t):
        return t if self.device is None else t.to(f"cuda:{self.device}")

    def _engine_stream(self):
        if self.device is None:
            return None
        return torch.cuda.ExternalStream(self.eng.stream(), device=f"cuda:{self.device}")

    def _on_engine_stream(self):
        import contextlib
        ext = self._engine_stream()
        return contextlib.nullcontext() if ext is None else torch.cuda.stream(ext)

    # The exchange's own stream: round r's collectives run under the scatter of round r + 1 and the absorbs of round
    # r − 1 on the engine's stream; events order the two where they meet (receive buffers, two sets taken in turn).
    def _on_comm_stream(self):
        import contextlib
        if self.device is None:
            return contextlib.nullcontext()
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=f"cuda:{self.device}")
        return torch.cuda.stream(self._comm)

    def _wait_on_comm(self, ev):
        if ev is not None and self._comm is not None:
            self._comm.wait_event(ev)

    def _record_on_comm(self):
        if self._comm is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self._comm)
        return ev

    def round(self, batch=None):
        """batch = (bases, offsets, n_seqs, n_bases, first_read_index) — device pointers for the HIP
        engine — or None.  n_bases ≤ round_bases."""
        dist, W = self.dist, self.world
        if self._wide is None:
            # decided ONCE, by all ranks together (feasibility looks at the share's table geometry, which a rank's
            # capacity hint sets): the 4-byte owner layout if every rank can take it, the wide round otherwise
            ok = self._dev(torch.tensor([1 if not hasattr(self.eng, "xchg_feasible") or self.eng.xchg_feasible() else 0], dtype=torch.int64))
            with self._on_engine_stream():
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                self._wide = int(ok.item()) == 0
        if self._wide:
            return self._round_wide(batch)
        err = None
        tr = self._trace
        if tr is not None:
            import time as _t
            t_in = _t.perf_counter()
        # (an engine with the two-call scatter — shk_xchg_scatter_begin / _end — is not waited for here: the absorbs below
        # go onto its stream behind the scatter, and the outcome is fetched after they have been launched)
        begun = False
        try:
            args = (batch[0], batch[1], batch[2], batch[3]) if batch is not None else (0, 0, 0, 0)
            if batch is not None:
                self.eng.set_read_index(batch[4])
            if self._two_calls:
                rec, cur, lay = self.eng.xchg_scatter_begin_tensors(*args, self.round_bases)
                begun, n_foreign = True, 0
            else:
                rec, cur, lay, n_foreign = self.eng.xchg_scatter_tensors(*args, self.round_bases)
        except Exception as e:  # noqa: BLE001 — reported to every rank below
            err, n_foreign = e, 0
        if tr is not None:
            t_sc = _t.perf_counter()
        # What the engine counts while this round's flags and segments travel: the round BEFORE — its exchange ran
        # under this round's scatter — and then the next round's scatter (the engine hands its exchange buffers out
        # in turn: include/shk.h, shk_xchg_scatter_device).  Launched first, so the host's wait for the flags below
        # costs the GPU nothing.  (A rank whose scatter — or this absorb — failed still takes part in the flags below:
        # the others must not be left waiting in a collective.)
        if err is None:
            try:
                self._absorb_pending()
            except Exception as e:  # noqa: BLE001
                err = e
        if begun:
            try:
                n_foreign = self.eng.xchg_scatter_end()
            except Exception as e:  # noqa: BLE001
                err = err if err is not None else e
        r2 = self.n_rounds & 1
        with self._on_comm_stream():
            st = self._dev(torch.tensor([1 if err is not None else 0, n_foreign], dtype=torch.int64))
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
            any_err, any_foreign = (int(x) for x in st.cpu())   # (in stream order behind the previous round's exchange: it has arrived)
            if tr is not None:
                t_ar = _t.perf_counter()
            if any_err:
                raise err if err is not None else RuntimeError("a peer rank failed in this exchange round")
            if self._recv[r2] is None or self._recv[r2][0].numel() != rec.numel() or self._recv[r2][1].numel() != cur.numel():
                self._recv[r2] = (torch.empty_like(rec), torch.empty_like(cur))
            rrec, rcur = self._recv[r2]
            S = lay.segment_records
            # a segment above the message limit travels in pieces, and the rank's own segment then does not travel at
            # all: it is absorbed where the scatter left it (shk.h: "rank r's own segment needs no copy")
            big = S * rec.element_size() > MAX_MESSAGE_BYTES and hasattr(dist, "batch_isend_irecv")
            self._wait_on_comm(self._absorbed[r2])   # these receive buffers were read by the absorbs of round r − 2
            exchange_parts(dist, rrec, rec, W, self.rank, skip_self=True)
            dist.all_to_all_single(rcur, cur)
            arrived = self._record_on_comm()
        self._pending = (rec, cur, rrec, rcur, lay, big, arrived, r2)   # (on the links now; absorbed by the next round)
        if any_foreign:
            self.n_foreign_rounds += 1
            with self._on_comm_stream():
                self._exchange_spills()
        if tr is not None:   # host-side phases of the round (SHK_DIST_TRACE): scatter incl. its sync, flags, exchange + absorb launches
            t_out = _t.perf_counter()
            tr[0] += t_sc - t_in
            tr[1] += t_ar - t_sc
            tr[2] += t_out - t_ar
            tr[3] += 1
        self.n_rounds += 1
        self.wire_bytes += (W - 1) * (lay.segment_records * rec.element_size() + lay.regions * 4)
        return lay

    def _absorb_pending(self):
        """The level-2 pass over every segment of the round whose exchange was queued last (engine stream, behind the
        exchange's completion)."""
        if self._pending is None:
            return
        rec, cur, rrec, rcur, lay, big, arrived, r2 = self._pending
        self._pending = None
        S, G = lay.segment_records, lay.regions
        ext = self._engine_stream()
        if ext is not None and arrived is not None:
            ext.wait_event(arrived)
        for s in range(self.world):
            src = rec if (big and s == self.rank) else rrec
            self.eng.xchg_absorb_tensors(src[s * S:(s + 1) * S], rcur[s * G:(s + 1) * G], lay)
        if ext is not None:
            ev = torch.cuda.Event()
            ev.record(ext)
            self._absorbed[r2] = ev

    def _round_wide(self, batch):
        """The round for k-mers that do not fit the owner layout's 4-byte records (k > 21 at the default fan-out;
        engine.xchg_feasible() is False — a function of the configuration, so every rank answers alike): the batch's
        k-mers travel whole — 8 B k-mer + 4 B lane, grouped by owner, as many as there are — and the receiver inserts them
        through the table's general path (shk_insert_device).  Exact for every k ≤ 32; three times the bytes on the
        links and the slow side of the table, which is what the 4-byte layout exists to avoid."""
        dist, W = self.dist, self.world
        err, km, ln, counts = None, None, None, [0] * W
        try:
            if batch is not None:
                self.eng.set_read_index(batch[4])
                km, ln, counts = self.eng.xchg_wide_scatter_tensors(batch[0], batch[1], batch[2], batch[3])
        except Exception as e:  # noqa: BLE001 — reported to every rank below
            err = e
        with self._on_engine_stream():
            st = self._dev(torch.tensor([1 if err is not None else 0, max(counts)], dtype=torch.int64))
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
            any_err, max_part = (int(x) for x in st.cpu())
            if any_err:
                raise err if err is not None else RuntimeError("a peer rank failed in this exchange round")
            mine = self._dev(torch.tensor(counts, dtype=torch.int64))
            theirs = torch.empty_like(mine)
            dist.all_to_all_single(theirs, mine)
            got = [int(x) for x in theirs.cpu()]
            n_in = sum(got)
            if km is None:
                km = self._dev(torch.empty(0, dtype=torch.int64))
                ln = self._dev(torch.empty(0, dtype=torch.int32))
            rk = torch.empty(n_in, dtype=torch.int64, device=km.device)
            rl = torch.empty(n_in, dtype=torch.int32, device=km.device)
            pieces = max_part * 8 > MAX_MESSAGE_BYTES
            if W > 1 or n_in:
                exchange_parts(dist, rk, km, W, self.rank, counts, got, in_pieces=pieces)
                exchange_parts(dist, rl, ln, W, self.rank, counts, got, in_pieces=pieces)
            if km.is_cuda:
                torch.cuda.current_stream().synchronize()
            self.eng.insert_tensors(rk, rl, None)
        self.n_rounds += 1
        self.wire_bytes += (sum(counts) - counts[self.rank]) * 12
        return None

    def _exchange_spills(self):
        """All-gather the foreign spill lists; every rank inserts what it owns (insert drops the rest)."""
        dist, W = self.dist, self.world
        k, l, c = self.eng.xchg_spill_tensors()
        n = self._dev(torch.tensor([k.numel()], dtype=torch.int64))
        ns = [torch.zeros_like(n) for _ in range(W)]
        dist.all_gather(ns, n)
        ns = [int(x.item()) for x in ns]
        m = max(ns)
        if m == 0:
            return

        def padded(t, dtype):
            out = torch.zeros(m, dtype=dtype, device=t.device)
            out[:t.numel()] = t
            return out
        for src_t, dtype, which in ((k, torch.int64, 0), (l, torch.int32, 1), (c, torch.int32, 2)):
            parts = [torch.empty(m, dtype=dtype, device=src_t.device) for _ in range(W)]
            dist.all_gather(parts, padded(src_t, dtype))
            if which == 0:
                gk = parts
            elif which == 1:
                gl = parts
            else:
                gc = parts
        if k.is_cuda:
            torch.cuda.current_stream().synchronize()
        for s in range(W):
            if ns[s]:
                self.eng.insert_tensors(gk[s][:ns[s]].contiguous(), gl[s][:ns[s]].contiguous(), gc[s][:ns[s]].contiguous())
        self.eng.xchg_spill_clear()

    def finalize_histograms(self):
        dist = self.dist
        if self._trace is not None and self._trace[3]:
            import sys
            a, b, c, n = self._trace
            print(f"[shk] OwnerCounter rank {self.rank}: {n} rounds, host ms per round: scatter+sync {a / n * 1e3:.3f}, flags {b / n * 1e3:.3f}, "
                  f"exchange+absorb launches {c / n * 1e3:.3f}", file=sys.stderr)
            self._trace = [0.0, 0.0, 0.0, 0]
        err = None
        try:
            self._absorb_pending()   # (the last round's segments)
            self.eng.finalize()
            h = self.eng.histograms()
            c = self.eng.counters()
        except Exception as e:  # noqa: BLE001
            err = e
        names = ["n_reads_ingested", "n_bases_read", "n_bases_ingested", "n_kmers_ingested",
                 "n_unique_kmers", "n_hashed_kmers", "any_saturated"]
        if err is not None:
            packed = np.zeros(1, dtype=np.int64)
        else:
            packed = np.concatenate([np.array([0] + [c[k] for k in names], dtype=np.int64), h.astype(np.int64).reshape(-1)])
        flag = self._dev(torch.tensor([1 if err is not None else 0], dtype=torch.int64))
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            raise err if err is not None else RuntimeError("a peer rank failed in finalize")
        pt = self._dev(torch.from_numpy(packed))
        dist.all_reduce(pt, op=dist.ReduceOp.SUM)
        red = pt.cpu().numpy()[1:]
        self.totals = {k: int(v) for k, v in zip(names, red[:len(names)])}
        self.totals["any_saturated"] = int(self.totals["any_saturated"] > 0)
        hist = red[len(names):].astype(np.uint64).reshape(h.shape)
        if self.totals["n_reads_ingested"] == 0:  # io.rs:578-580 on the whole job
            raise RuntimeError("No reads were ingested. Check that input files contain valid FASTQ records.")
        if hist.shape[0] > 0:
            self.totals["n_singleton_kmers"] = int(hist[-1, 1])
            if self.totals["n_hashed_kmers"] != self.totals["n_kmers_ingested"]:  # io.rs:1042-1047
                raise RuntimeError(
                    f"The total count of hashed kmers ({self.totals['n_hashed_kmers']}) does not equal "
                    f"the number of ingested kmers ({self.totals['n_kmers_ingested']})")
            if int(hist[-1, 1:].sum()) != self.totals["n_unique_kmers"]:  # io.rs:1127-1132
                raise RuntimeError("The total count of unique kmers in the histogram does not equal "
                                   "the total count of hashed kmers")
        return hist


class DistCounter:
    def __init__(self, engine, dist, device=None):
        self.eng = engine
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.device = device
        self.dense = bool(int(os.environ.get("SHK_DIST_DENSE", "0")))
        # Entries a fixed-capacity piece holds (None: the first exchange asks for exact counts and sets it).
        # SHK_DIST_CAP pins it (tests: a capacity that is too small must still give the exact result).
        self._cap = int(os.environ["SHK_DIST_CAP"]) if os.environ.get("SHK_DIST_CAP") else None
        self._cap_pinned = self._cap is not None
        self._fixed_used = 0      # capacity of the pieces the last exchange handed over (0: exact protocol)
        self._last_max = 0
        self.n_fixed_exchanges = 0
        self.n_redone_exchanges = 0

    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        return t if self.device is None else t.to(f"cuda:{self.device}")

    def _agree_on_pages(self):
        """Step 1: every rank ends up with the same page count P, a multiple of the world size."""
        dist, W = self.dist, self.world
        n_pages, _, _ = self.eng.table_geometry()
        t = self._dev(torch.tensor([max(n_pages, W)], dtype=torch.int64))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        P = int(t.item())
        self.eng.reserve_pages(P)
        n_pages, page_slots, n_lanes = self.eng.table_geometry()
        assert n_pages == P and P % W == 0, (n_pages, P, W)
        return P, page_slots, n_lanes

    def _own(self, P_before):
        """Owned page range after the merges (a merge that had to grow the table splits every page
        into consecutive child pages, so the range scales with the page count)."""
        P_now, _, _ = self.eng.table_geometry()
        f = P_now // P_before
        per = P_before // self.world
        p0, p1 = self.rank * per * f, (self.rank + 1) * per * f
        self.eng.set_owned_pages(p0, p1)
        return p0, p1

    def exchange_and_merge(self, exact=False):
        """Steps 1-3.  After it, this rank's owned page range holds the merged counts.
        Only the occupied (key, counts) entries cross the links: per-owner entry counts first
        (one small all_to_all), then one all_to_all with those split sizes per array."""
        if self.dense:
            return self.exchange_and_merge_dense()
        dist, W = self.dist, self.world
        # Geometry and entry counts in one small all_to_all: every rank sends every peer (its page
        # count, what it holds of that peer's range).  Page counts that differ (a table grew on one
        # rank only) show up on every rank alike: all grow to the largest and go round once more.
        self._fixed_used = 0
        if self._cap is not None and not exact and hasattr(self.eng, "compact_owner_fixed") and self.device is not None:
            # (not even the table geometry is asked for — that would wait for the counting launches: an earlier
            # exact exchange has made sure there are at least W pages, and tables never shrink)
            # Nothing to ask anybody first: pieces of a fixed capacity at fixed places, unused places EMPTY (the
            # merge skips them).  An owner is a range of the mixed key's TOP bits, so pieces are meaningful to the
            # receiver whatever page count either table has grown to.  Every piece's header says how many entries
            # its sender's fullest range had; the merge is all or nothing on every rank alike, and whether it was
            # nothing comes back with finalize's own read-back (finalize_histograms then repeats the exchange
            # the exact way — the tables are as they were).
            ext = torch.cuda.ExternalStream(self.eng.stream(), device=f"cuda:{self.device}")
            with torch.cuda.stream(ext):
                buf, L = self.eng.compact_owner_fixed(W, self._cap, self.rank)
                rbuf = torch.empty_like(buf)
                exchange_parts(dist, rbuf, buf, W, self.rank)
                self.eng.merge_fixed_pieces(rbuf, W, self._cap, self.rank)
                self._keep = (buf, rbuf)
            self._fixed_used = self._cap
            self.n_fixed_exchanges += 1
            self.eng.set_owner_share(W, self.rank)
            return None
        n_pages, _, n_lanes = self.eng.table_geometry()
        if n_pages < W or n_pages % W:
            self.eng.reserve_pages(max(n_pages, W))
            n_pages, _, n_lanes = self.eng.table_geometry()
        while True:
            counts = np.asarray(self.eng.owner_counts(W), dtype=np.int64)   # what I hold of every owner's range
            counts[self.rank] = 0                                           # … my own range stays where it is
            # (third word: the largest part this rank SENDS — every part is sent by somebody, so the maximum over what a
            # rank receives is the largest part of the whole exchange, the same number on every rank: it picks the
            # route of the exchange below, which all ranks must take alike)
            msg = np.empty((W, 3), dtype=np.int64)
            msg[:, 0] = n_pages
            msg[:, 1] = counts
            msg[:, 2] = int(counts.max()) if W else 0
            send_n = self._dev(torch.from_numpy(msg.reshape(-1)))
            recv_n = torch.empty_like(send_n)
            dist.all_to_all_single(recv_n, send_n)
            got = recv_n.cpu().numpy().reshape(W, 3)
            P = int(got[:, 0].max())
            if P == n_pages and int(got[:, 0].min()) == n_pages:
                break
            self.eng.reserve_pages(P)
            n_pages, _, n_lanes = self.eng.table_geometry()
            assert n_pages == P, (n_pages, P)
        recv = [int(x) for x in got[:, 1]]                              # what every peer holds of MY range
        send = [int(x) for x in counts]
        largest_part = int(got[:, 2].max())                             # entries; the same on every rank
        self._last_max = max(max(send), max(recv))  # (finalize_histograms turns the job-wide maximum into the next capacity)
        if hasattr(self.eng, "compact_owner_packed") and self.device is not None:
            # k-mers and ALL lanes' counts of a peer in ONE collective: every owner's entries are one
            # self-contained piece [k-mers][lane 0]…[lane L-1]; compaction, the all_to_all and the merges are
            # queued on the engine's own HIP stream (no host-side synchronisation in between)
            ext = torch.cuda.ExternalStream(self.eng.stream(), device=f"cuda:{self.device}")
            with torch.cuda.stream(ext):
                buf, L = self.eng.compact_owner_packed(counts, self.rank)
                w = 2 + L
                rbuf = torch.empty(max(sum(recv) * w, 1), dtype=torch.int32, device=buf.device)[:sum(recv) * w]
                exchange_parts(dist, rbuf, buf, W, self.rank, in_splits=[x * w for x in send], out_splits=[r * w for r in recv],
                               in_pieces=largest_part * w * buf.element_size() > MAX_MESSAGE_BYTES)
                at = 0
                for src in range(W):
                    self.eng.merge_packed_piece(rbuf[at:at + recv[src] * w], recv[src], L)
                    at += recv[src] * w
                self._keep = (buf, rbuf)  # (alive until the stream has passed the merges: the next sync is finalize's)
            return self._own(P)
        keys, vals = self.eng.compact_owner_tensors(counts, self.rank)  # [sum(send)], [L, sum(send)]
        n_recv = sum(recv)
        rk = keys.new_empty(n_recv)
        rv = vals.new_empty((n_lanes, n_recv))
        exchange_parts(dist, rk, keys, W, self.rank, in_splits=send, out_splits=recv,
                       in_pieces=largest_part * keys.element_size() > MAX_MESSAGE_BYTES)
        for l in range(n_lanes):
            exchange_parts(dist, rv[l], vals[l].contiguous(), W, self.rank, in_splits=send, out_splits=recv,
                           in_pieces=largest_part * vals.element_size() > MAX_MESSAGE_BYTES)
        if rk.is_cuda:
            # RCCL enqueues on torch's stream; the merge below runs on the engine's own HIP
            # stream, so the received entries must have landed before it is launched
            torch.cuda.synchronize()
        self.eng.merge_entry_tensors(rk, rv)
        return self._own(P)

    def exchange_and_merge_dense(self):
        """The same with the owner ranges shipped as they lie in the table (EMPTY slots included):
        no compaction pass, one all_to_all per array with equal splits."""
        dist, W = self.dist, self.world
        P, page_slots, n_lanes = self._agree_on_pages()
        per = P // W                      # pages per owner
        n = per * page_slots              # slots per owner slice
        keys, vals = self.eng.table_tensors()          # [P*S], [L, P*S]
        rk = torch.empty_like(keys)                     # rk[s*n:(s+1)*n] = rank s's slice of MY range
        rv = torch.empty_like(vals)
        exchange_parts(dist, rk, keys, W, self.rank)
        for l in range(n_lanes):
            exchange_parts(dist, rv[l], vals[l], W, self.rank)
        if rk.is_cuda:
            torch.cuda.synchronize()
        p0, p1 = self.rank * per, (self.rank + 1) * per
        for s in range(W):
            if s == self.rank:
                continue
            self.eng.merge_page_tensors(p0, p1, rk[s * n:(s + 1) * n], rv[:, s * n:(s + 1) * n])
        return self._own(P)

    def finalize_histograms(self):
        """Steps 1-4.  Returns the (chunks, histo_max+2) uint64 histogram of the union of all
        ranks' reads — identical on every rank — and stores the reduced totals in self.totals."""
        dist, W = self.dist, self.world
        names = ["n_reads_ingested", "n_bases_read", "n_bases_ingested", "n_kmers_ingested",
                 "n_unique_kmers", "n_hashed_kmers", "any_saturated"]
        self.exchange_and_merge()
        if hasattr(self.eng, "finalize_begin") and self.device is not None:
            # The histogram never visits the host before it is summed: the scan is queued behind the merges, the
            # all_reduce runs in place on the engine's control block on the engine's stream, and ONE read-back
            # brings the whole job's histogram and totals.  Three things can ask for another go, each known to
            # every rank alike: somebody's scan ran over a table that had to be repaired first (`again`), the
            # pieces of the exchange were poisoned (a sender's table was incomplete when they were cut), or a
            # piece was too small (then with exact counts).
            ext = torch.cuda.ExternalStream(self.eng.stream(), device=f"cuda:{self.device}")
            while True:
                with torch.cuda.stream(ext):
                    blk = self.eng.finalize_begin(self._last_max)
                    dist.all_reduce(blk, op=dist.ReduceOp.SUM)
                again, max_sum = self.eng.finalize_end()
                if self._fixed_used:
                    most = self.eng.merge_pieces_max()   # (the same number on every rank: see shk_merge_pieces)
                    if most > self._fixed_used:
                        self.n_redone_exchanges += 1
                        self.exchange_and_merge(exact=most != (1 << 64) - 1)
                        continue
                else:
                    most = -(-max_sum // W)  # (exact protocol: mean over the ranks of their fullest range — a guess; a
                    #                           piece that turns out too small costs one repeated exchange, once)
                if not again:
                    break
            if not self._cap_pinned and not self.dense:
                self._cap = -(-(most + most // 8 + 4096) // 1024) * 1024
            hist = self.eng.histograms()
            c = self.eng.counters()
            red = np.array([c[k] for k in names], dtype=np.int64)
        else:
            self.eng.finalize()
            h = self.eng.histograms()
            c = self.eng.counters()
            # one all_reduce for the histogram and the scalar totals together (both are sums)
            packed = np.concatenate([np.array([c[k] for k in names], dtype=np.int64),
                                     h.astype(np.int64).reshape(-1)])
            pt = self._dev(torch.from_numpy(packed))
            dist.all_reduce(pt, op=dist.ReduceOp.SUM)
            red = pt.cpu().numpy()
            hist = red[len(names):].astype(np.uint64).reshape(h.shape)
        self.totals = {k: int(v) for k, v in zip(names, red[:len(names)])}
        if hist.shape[0] > 0:
            self.totals["n_singleton_kmers"] = int(hist[-1, 1])
            # io.rs:1042-1047 / 1120-1132 on the merged whole
            if self.totals["n_hashed_kmers"] != self.totals["n_kmers_ingested"]:
                raise RuntimeError(
                    f"The total count of hashed kmers ({self.totals['n_hashed_kmers']}) does not equal "
                    f"the number of ingested kmers ({self.totals['n_kmers_ingested']})")
            if int(hist[-1, 1:].sum()) != self.totals["n_unique_kmers"]:
                raise RuntimeError("The total count of unique kmers in the histogram does not equal "
                                   "the total count of hashed kmers")
        return hist
