/*
 * shk.h — C ABI of libshk, the MI355X-native k-mer counting engine.
 *
 * Drop-in boundary for the hot path of caseywdunn/sharkmer v3.1.0 (pure Rust,
 * no FFI seam of its own): the seam sits where the reference's src/io.rs calls
 * into src/kmer/ — drain_batch → Chunk::ingest_seq (io.rs:355-361,
 * kmer/chunk.rs:25-30) and consolidate_and_histogram →
 * KmerCounts::extend_with_histogram / Histogram::get_vector (io.rs:1021-1028).
 * Each entry point below names the reference interface it replaces.
 *
 * Conventions (mirroring the reference's anyhow::Result-to-main behaviour,
 * SURVEY.md §8b): every function returns SHK_OK (0) or a negative SHK_ERR_*;
 * the message a reference run would have printed is available from
 * shk_last_error().  No exceptions, no unwinding, no torch types: plain
 * pointers and sizes only.  One caller thread per context (the reference's
 * counting is single-threaded); the library uses HIP streams internally.
 *
 * There is NO CPU fallback: shk_create fails with SHK_ERR_NO_DEVICE when no
 * gfx950 device is usable.
 */
#ifndef SHK_H
#define SHK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHK_ABI_VERSION 2

#define SHK_OK 0
#define SHK_ERR_INVALID_CHAR (-1) /* kmer/encoding.rs:353-356 */
#define SHK_ERR_BAD_ARG (-2)      /* cli.rs:659-677; encoding.rs:333 */
#define SHK_ERR_K_MISMATCH (-3)   /* counting.rs:158-160 */
#define SHK_ERR_NOMEM (-4)
#define SHK_ERR_NO_READS (-5)     /* io.rs:578-580 */
#define SHK_ERR_INVARIANT (-6)    /* io.rs:1042-1047, 1120-1132 */
#define SHK_ERR_FASTQ (-7)        /* io.rs:161-198, 287-318 */
#define SHK_ERR_IO (-8)
#define SHK_ERR_NO_DEVICE (-9)
#define SHK_ERR_HIP (-10)
#define SHK_ERR_STATE (-11)       /* call out of order / poisoned context */

#define SHK_READS_PER_BATCH 1000u /* N_READS_PER_BATCH, io.rs:15 */

/* flags */
#define SHK_FLAG_TIMING 1u        /* bracket every kernel with HIP events */
#define SHK_FLAG_FORCE_DIRECT 2u  /* count with the global-atomic kernel only */
#define SHK_FLAG_FORCE_PAGED 4u   /* count with the LDS-page kernels only */
#define SHK_FLAG_TIMING_SAMPLED 16u /* with SHK_FLAG_TIMING: only the launches of every 4th job are bracketed (a job =
                                    * what lies between two shk_reset calls; the first after shk_reset_timings is) — the
                                    * event records themselves cost ≈3 % of a 0.6 ms job */
#define SHK_RESERVE_NONE 0xFFFFFFFFu /* shk_config.reserve_cus: every compute unit, whatever SHK_RESERVE_CUS says */
#define SHK_FLAG_DEFER_ERRORS 8u  /* host-buffer ingests (shk_ingest_reads/_batch/_packed) return once their last
                                   * slice is queued, like the device-buffer ones: an invalid byte is reported by
                                   * the NEXT call on the context (ingest, sync, finalize), and that call's first
                                   * host-to-device copies overlap this one's counting */

typedef struct shk_ctx shk_ctx;

typedef struct shk_config {
  uint32_t k;          /* 0 < k < 32 (encoding.rs:333; CLI adds "odd", cli.rs:667) */
  uint32_t chunks;     /* CLI --chunks: 0 ⇒ one internal chunk and no histogram (io.rs:378) */
  uint64_t histo_max;  /* 0 < histo_max ≤ 1_000_000 (cli.rs:668-673) */
  int32_t device;      /* HIP device ordinal */
  uint32_t flags;      /* SHK_FLAG_* */
  uint64_t table_capacity_hint; /* expected distinct k-mers this context will hold; 0 = start small and grow.
                                   A multi-device context (n_devices > 1) splits it over its devices */
  /* OWNER SHARE ("alternative when local tables do not fit"): with n_owners = W > 1 (a
   * power of two ≤ 64) the context holds only the k-mers whose owner — the top log2(W) bits of the
   * engine's bijective key mix — is owner_id: 1/W of the key space, so W contexts (one per GPU) together
   * hold every k-mer exactly once and no table is ever exchanged.  shk_ingest_* on such a context counts
   * the owned k-mers of the reads it is given and drops the rest; the shk_xchg_* entry points below
   * partition a batch's records by owner for an exchange between the W contexts instead.  Read/base
   * counters count every read handed over; k-mer counters and histograms cover the owned share
   * (histogram bins are additive across shares: KmerCounts::extend over disjoint key sets,
   * counting.rs:157-166, io.rs:1023-1028).  0: the whole key space.  1: the whole key space AS A SHARE — the
   * context takes the shk_xchg_* rounds like any owner share (one segment), so that the exchange path of a
   * W-process job runs unchanged in a world of one. */
  uint32_t n_owners;
  uint32_t owner_id;
  /* MULTI-DEVICE CONTEXT (SURVEY.md §8b `n_devices, device_ids`): n_devices > 1 makes ONE context that
   * spans device_ids[0..n_devices) (a power of two ≤ 64 devices; an id may repeat — the test rigs put
   * several shares on one card): one owner share per device, host batches of 1000 reads dealt round-robin
   * to the devices with their global read index (io.rs:340-361), records exchanged by owner between the
   * devices during ingest, one histogram sum at finalize.  `device` is ignored then.  0 or 1: one device. */
  uint32_t n_devices;
  /* COMPUTE UNITS LEFT FREE (round 4): the context's PERSISTENT kernel — the 4-byte-record scatter, one 1024-thread
   * workgroup with 144 of 160 KiB of LDS on every CU for the whole launch — starts this many workgroups fewer (rounded
   * up to a multiple of 8), so that whoever else works on the card while it counts finds a CU: the collectives of a
   * multi-GPU job, whose channels are workgroups that need CUs and LDS while a round's segments travel under the next
   * round's scatter.  (The other kernels are many short workgroups; a collective's get in as those retire.)
   * 0: the default — none for a whole-key-space context, SHK_RESERVE_CUS (default 16) for an owner share of a
   * multi-GPU job (n_owners > 1); SHK_RESERVE_NONE (~0): none, whatever the environment says.  What it costs the
   * counting on one card is in DESIGN.md §6. */
  uint32_t reserve_cus;
  const int32_t *device_ids;
  uint64_t reserved[1];
} shk_config;

/* io.rs:545-552 and counting.rs:254-260 totals, plus device-side facts */
typedef struct shk_counters {
  uint64_t n_reads_ingested;  /* Σ Chunk::n_reads (chunk.rs:27) */
  uint64_t n_bases_read;      /* Σ sequence lengths, N included (io.rs:335) */
  uint64_t n_bases_ingested;  /* Σ Chunk::n_bases = non-N bytes (chunk.rs:28) */
  uint64_t n_kmers_ingested;  /* Σ over chunks of Σ table counts (io.rs:550) */
  uint64_t n_unique_kmers;    /* merged table len (counting.rs:258) — valid after finalize */
  uint64_t n_hashed_kmers;    /* Σ merged counts (io.rs:1035) — valid after finalize */
  uint64_t n_singleton_kmers; /* last column [1] (io.rs:1096-1099) — chunks>0 only */
  uint32_t any_saturated;     /* counting.rs:190-200 */
  uint32_t n_chunks;          /* internal chunk count (≥1) */
  uint64_t table_capacity;    /* slots */
  uint64_t n_grows;           /* table doublings so far */
  uint64_t n_spilled;         /* k-mers that took the spill path */
} shk_counters;

/* Accumulated per-kernel device time (HIP events on the engine's stream);
 * filled only when SHK_FLAG_TIMING is set. */
#define SHK_N_KERNELS 16
typedef struct shk_timings {
  double ms[SHK_N_KERNELS];
  uint64_t launches[SHK_N_KERNELS];
} shk_timings;
/* kernel ids for shk_timings */
enum {
  SHK_K_MARK = 0,      /* read-start bitmap + tile list */
  SHK_K_SCAN = 1,      /* validate / count bases / partition histogram */
  SHK_K_DIRECT = 2,    /* extract + global-atomic insert */
  SHK_K_SCATTER = 3,   /* extract + LDS-sorted partition scatter (k_part_scatter_sorted) */
  SHK_K_PAGES = 4,     /* LDS page count */
  SHK_K_HISTO = 5,     /* table scan → histograms + totals */
  SHK_K_GROW = 6,      /* rehash into a bigger table */
  SHK_K_INSERT = 7,    /* (kmer,count) insert / spill re-insert */
  SHK_K_LOOKUP = 8,
  SHK_K_EXPORT = 9,
  SHK_K_SYNTH = 10,
  SHK_K_MERGE = 11,
  SHK_K_PCOUNT = 12,   /* validate + count k-mers per partition (k_part_count) */
  SHK_K_PSCAN = 13,    /* the two small exclusive scans between count and scatter */
  SHK_K_HISTO_ROWS = 14 /* histograms + totals from the rows the (one) fresh page pass of a job left behind, instead of SHK_K_HISTO's scan */
};

/* ---- lifecycle ------------------------------------------------------------ */

/* Replaces Chunk::new ×n_chunks (io.rs:378-379) + Histogram::new (io.rs:1021).
 * Validates the cli.rs:659-677 ranges except "k odd" (a CLI rule; the kmer
 * module itself accepts any 0<k<32, encoding.rs:333). */
int shk_create(const shk_config *cfg, shk_ctx **out);
/* Drop of FastqReadState / KmerCounts. */
void shk_destroy(shk_ctx *ctx);
/* Back to the state right after shk_create (empty chunks, zero counters), keeping the
 * allocated table and scratch: a fresh FastqReadState (io.rs:381-387) without re-allocating. */
int shk_reset(shk_ctx *ctx);
/* The anyhow message of the last failure on this context ("" if none).
 * ctx == NULL returns the message of the last failed shk_create on this thread. */
const char *shk_last_error(const shk_ctx *ctx);
int shk_abi_version(void);

/* ---- ingest ----------------------------------------------------------------- */

/* Replaces the body of drain_batch (io.rs:356-358): every sequence of the
 * batch goes to chunk `chunk_id` (Chunk::ingest_seq, chunk.rs:25-30).
 * bases: concatenated ASCII sequences; offsets[n_seqs+1] byte offsets into
 * bases.  Host buffers, caller-owned, may be freed after return.
 * Any byte outside ACGTN ⇒ SHK_ERR_INVALID_CHAR, message identical to
 * encoding.rs:353-356, context poisoned (the reference aborts the run). */
int shk_ingest_batch(shk_ctx *ctx, uint32_t chunk_id, const uint8_t *bases,
                     const uint64_t *offsets, uint64_t n_seqs);

/* Replaces read_fastq's push + drain_batch cadence (io.rs:335-343, 355-361)
 * for a run of sequences in input order: the engine keeps the running read
 * index i and sends read i to chunk (i / 1000) % n_chunks. */
int shk_ingest_reads(shk_ctx *ctx, const uint8_t *bases, const uint64_t *offsets,
                     uint64_t n_seqs);

/* ---- 2-bit packed input ---------------------------------------------------------------------------------
 * The layout of the reference's Read::from_str (src/kmer/encoding.rs:60-95; known-answer vectors
 * src/kmer/mod.rs:61-156) applied to a batch's concatenated bases as ONE sequence: 4 bases per byte, the first
 * base in the two most significant bits (A 00, C 01, G 10, T 11), the tail left-aligned in its byte; plus an
 * N mask — bit p % 32 of word p / 32 set where base p is N (code 00 in the stream) — because
 * kmers_from_ascii, unlike from_str, takes N (encoding.rs:346-352).  0.28 B/base instead of 1 over PCIe.
 * shk_pack_reads (host, n_threads = 0: all cores) validates like encoding.rs:353-356: the first byte outside
 * ACGTN in input order ⇒ SHK_ERR_INVALID_CHAR, message from shk_run_error().  packed: (n_bases+3)/4 bytes,
 * nmask: (n_bases+31)/32 words (shk_packed_sizes). */
void shk_packed_sizes(uint64_t n_bases, uint64_t *packed_bytes, uint64_t *nmask_words);
int shk_pack_reads(const uint8_t *bases, uint64_t n_bases, uint8_t *packed, uint32_t *nmask, uint32_t n_threads);
/* shk_ingest_reads for a packed batch: offsets[n_seqs+1] are BASE indices into the stream (offsets[0] is
 * where the batch starts in it).  Host buffers. */
int shk_ingest_packed(shk_ctx *ctx, const uint8_t *packed, const uint32_t *nmask, const uint64_t *offsets,
                      uint64_t n_seqs);
/* The same with everything resident in device memory (the stream starts at base 0; d_offsets as in
 * shk_ingest_reads_device).  Asynchronous like shk_ingest_reads_device. */
int shk_ingest_packed_device(shk_ctx *ctx, const void *d_packed, const void *d_nmask, const void *d_offsets,
                             uint64_t n_seqs, uint64_t n_bases);
/* The packer and its inverse on the device (ASCII bytes ↔ stream + mask, all device pointers): what the
 * device-side buffer of a batch looks like, pinned by the reference's vectors in tests/test_gpu_packed.py.
 * shk_pack_reads_device reports an invalid byte like shk_pack_reads (message from shk_last_error) and leaves
 * the context usable. */
int shk_pack_reads_device(shk_ctx *ctx, const void *d_bases, uint64_t n_bases, void *d_packed, void *d_nmask);
int shk_unpack_reads_device(shk_ctx *ctx, const void *d_packed, const void *d_nmask, uint64_t n_bases, void *d_bases);

/* Set the running read index used by shk_ingest_reads* for chunk striping (default 0).  A
 * host that shards one input stream over several contexts/GPUs gives each shard the global
 * index of its first read, so that read i still lands in chunk (i / 1000) % n_chunks. */
int shk_set_read_index(shk_ctx *ctx, uint64_t next_read_index);

/* Same as shk_ingest_reads with input already resident in device memory
 * (d_bases: n_bases bytes, d_offsets: n_seqs+1 u64).  Asynchronous on the
 * engine stream; errors surface at the next synchronising call. */
int shk_ingest_reads_device(shk_ctx *ctx, const void *d_bases, const void *d_offsets,
                            uint64_t n_seqs, uint64_t n_bases);

/* Replaces KmerCounts::insert (counting.rs:152-154) on chunk `chunk_id`:
 * saturating add of counts[i] to kmers[i]. Host buffers. */
int shk_insert_counts(shk_ctx *ctx, uint32_t chunk_id, const uint64_t *kmers,
                      const uint32_t *counts, uint64_t n);

/* Wait for all queued device work; report deferred errors. */
int shk_sync(shk_ctx *ctx);

/* ---- consolidate ----------------------------------------------------------- */

/* Replaces the merge loop of consolidate_and_histogram (io.rs:1023-1028 for
 * chunks>0, io.rs:1135-1138 for chunks==0) and its invariants
 * (io.rs:1042-1047, 1120-1132).  SHK_ERR_NO_READS mirrors io.rs:578-580.
 * Idempotent; ingest after finalize re-opens the context. */
int shk_finalize(shk_ctx *ctx);
/* shk_finalize in two halves — histo_vecs (io.rs:1020-1028) of the UNION of several contexts' reads, bins being
 * additive over disjoint key sets (KmerCounts::extend, counting.rs:157-166) — for a caller that sums histograms
 * over several contexts ON THE DEVICE (one process
 * per GPU: an in-place all-reduce of *d_sum over the ranks, queued on shk_stream(), instead of a host round trip
 * through shk_histograms).  _begin queues the scan and hands out the block a reduction may SUM — n_words u64:
 * the totals, the per-lane base counts, four bookkeeping words (reads ingested, bases read, "my scan ran over a
 * table that still had to be repaired", user_word), the histogram.  _end brings the block back (one host sync for
 * everything queued so far) and repairs this context's table if its last launch had spilled records; *again = 1
 * when ANY context's had — the same sum on every rank — in which case nothing is final and every rank repeats
 * _begin / reduction / _end.  Afterwards shk_histograms and shk_get_counters (n_reads_ingested, n_bases_*,
 * n_kmers_ingested, n_unique_kmers, n_hashed_kmers) report what the reduction left: the whole job's.  The
 * io.rs:1042-1047 invariants are the caller's to check on the summed numbers.  *user_sum: Σ user_word. */
int shk_finalize_begin(shk_ctx *ctx, uint64_t user_word, void **d_sum, uint64_t *n_words);
int shk_finalize_end(shk_ctx *ctx, int *again, uint64_t *user_sum);

/* histo_vecs of io.rs:1020-1028: chunks × (histo_max+2) u64, row-major by
 * chunk; column j = Histogram::get_vector() after merging chunks 0..j
 * (histogram.rs:125-134).  chunks==0 ⇒ nothing is written, returns SHK_OK. */
int shk_histograms(shk_ctx *ctx, uint64_t *out);

int shk_get_counters(shk_ctx *ctx, shk_counters *out);
int shk_get_timings(shk_ctx *ctx, shk_timings *out);
int shk_reset_timings(shk_ctx *ctx);

/* ---- merged-table read API (counting.rs:205-260) --------------------------------- */

/* KmerCounts::iter (counting.rs:239-241) over the merged table: writes up to
 * cap (kmer,count) pairs in unspecified order; *n_out = number of entries. */
int shk_export_table(shk_ctx *ctx, uint64_t *kmers, uint32_t *counts, uint64_t cap,
                     uint64_t *n_out);
/* KmerCounts::get_count (canonical=0, counting.rs:224-226) or
 * get_canonical_count (canonical=1: min(kmer, revcomp), counting.rs:205-209). */
int shk_lookup(shk_ctx *ctx, const uint64_t *kmers, uint32_t *counts, uint64_t n,
               int canonical);

/* find_oligos_in_kmers (src/pcr/primers.rs:163-226), sPCR's primer seed scan over the merged
 * table: oligos[] are 2-bit encoded values of oligo_len bases (1 ≤ oligo_len < k).  A k-mer
 * with merged count ≥ min_count whose FIRST oligo_len bases equal an oligo is reported as is;
 * one whose LAST oligo_len bases equal an oligo's reverse complement is reported
 * reverse-complemented (primers.rs:212-223).  Same output convention as shk_export_table. */
int shk_find_oligos(shk_ctx *ctx, const uint64_t *oligos, uint32_t n_oligos, uint32_t oligo_len,
                    uint32_t min_count, uint64_t *kmers, uint32_t *counts, uint64_t cap, uint64_t *n_out);

/* PrimerReadFilter::matches over a batch (src/pcr/read_filter.rs:24-55), the read selection in front
 * of sPCR's read threading: out_matches[i] = 1 when read i has no byte outside ACGTN (otherwise
 * kmers_from_ascii fails and the reference returns false for the read) and at least one of its
 * canonical k-mers (k of the context) is among primer_kmers[] (canonical 2-bit k-mers, the union
 * of the forward and reverse primer tables), else 0.  Host buffers; the context's table is not
 * touched. */
int shk_filter_reads(shk_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_seqs,
                     const uint64_t *primer_kmers, uint64_t n_kmers, uint8_t *out_matches);

/* Batched kmers_from_ascii (src/kmer/encoding.rs:332-371) as sPCR's read threading calls it
 * (src/pcr/threading.rs:97-101, also pcr/read_filter.rs:44): read i's canonical k-mers, in read
 * order, are kmers[koff(i) .. koff(i) + n_kmers[i]) with koff(i) = Σ_{r<i} max(0, len_r − k + 1),
 * the most a read can yield (an N shortens it: n_kmers[i] ≤ max(0, len_i − k + 1); the rest of the
 * read's span is left untouched).  bad_byte[i] = 0, or the first byte outside ACGTN: such a read
 * is the reference's Err (message of encoding.rs:353-356 with that byte) and yields no k-mers
 * (n_kmers[i] = 0), which is how thread_reads treats it (threading.rs:99-101: skipped).
 * kmers_cap must be ≥ koff(n_seqs).  Host buffers; the context's table is not touched. */
int shk_kmers_from_reads(shk_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_seqs,
                         uint64_t *kmers, uint64_t kmers_cap, uint32_t *n_kmers, uint8_t *bad_byte);

/* ---- multi-GPU hooks (device pointers; exchanged by the caller over RCCL) ---- */

/* Table geometry needed to shard by owner: n_pages (power of two),
 * page_slots, n_lanes.  Pages [p0,p1) are contiguous in every array. */
int shk_table_geometry(shk_ctx *ctx, uint64_t *n_pages, uint32_t *page_slots,
                       uint32_t *n_lanes);
/* Grow the table until it has at least n_pages pages (all ranks must agree
 * before exchanging page ranges). */
int shk_table_reserve_pages(shk_ctx *ctx, uint64_t n_pages);
/* Device pointers to the live table arrays: keys u64[n_pages*page_slots],
 * vals u32[n_lanes][n_pages*page_slots]. Valid until the next growing call. */
int shk_table_device_ptrs(shk_ctx *ctx, void **d_keys, void **d_vals);
/* KmerCounts::extend across devices (counting.rs:157-166): merge a peer's
 * page range [p0,p1) (same geometry; device pointers to its keys and to its
 * lane-major vals for that range, lane l at d_vals + l*vals_lane_stride u32
 * elements) into this table, saturating per lane. */
int shk_merge_pages(shk_ctx *ctx, uint64_t p0, uint64_t p1, const void *d_keys,
                    const void *d_vals, uint64_t vals_lane_stride);
/* The same merge with only the OCCUPIED entries on the wire.  The table's pages split evenly over
 * n_owners contiguous ranges (owner o = pages [o·P/n, (o+1)·P/n)).
 *   shk_owner_counts   : counts[o] = occupied slots of owner o's range
 *   shk_compact_owners : writes owner o's entries to d_keys[seg_offsets[o] ..] (and lane l of their
 *                        counts to d_vals + l·vals_lane_stride + the same index), owner after owner;
 *                        skip_owner ≥ 0: that owner's range is left out (a rank keeps its own)
 *   shk_merge_entries  : KmerCounts::extend (counting.rs:157-166) of n received entries */
int shk_owner_counts(shk_ctx *ctx, uint32_t n_owners, uint64_t *counts);
int shk_compact_owners(shk_ctx *ctx, uint32_t n_owners, const uint64_t *seg_offsets, void *d_keys, void *d_vals,
                       uint64_t vals_lane_stride, int32_t skip_owner);
int shk_merge_entries(shk_ctx *ctx, const void *d_keys, const void *d_vals, uint64_t n, uint64_t vals_lane_stride);
/* shk_compact_owners with every owner's entries as ONE self-contained piece of d_buf — at byte offset
 * (Σ_{o'<o} counts[o'])·(8 + 4·n_lanes): [k-mers 8·c][lane 0 counts 4·c] … [lane L−1 counts], c = counts[o]
 * (from shk_owner_counts; counts[skip_owner] must be 0) — so that k-mers and all lanes' counts cross the links
 * in one all-to-all.  A received piece goes to shk_merge_entries(piece, piece + 8·c, c, c).  Asynchronous on
 * the context's stream (shk_stream). */
int shk_compact_owners_packed(shk_ctx *ctx, uint32_t n_owners, const uint64_t *counts, void *d_buf, int32_t skip_owner);
/* The same (the sender's half of KmerCounts::extend across devices, counting.rs:157-166) with pieces of a FIXED
 * capacity (entries) at fixed places, so that nobody has to know anybody's counts
 * before the all-to-all (no host read-back in front of it).  Piece o lies at byte offset o·(8 + capacity·(8 +
 * 4·n_lanes)): an 8-byte header, then [k-mers][lane 0]…[lane L-1] with unused places holding EMPTY k-mers (all
 * ones).  Entries beyond the capacity are NOT written; the header of every piece holds how many entries the
 * fullest owner range of this sender had.  Asynchronous on the context's stream. */
int shk_compact_owners_fixed(shk_ctx *ctx, uint32_t n_owners, uint64_t capacity, void *d_buf, int32_t skip_owner);
/* KmerCounts::extend (counting.rs:157-166, saturating per lane as counting.rs:144-149) of the n_pieces received
 * fixed-capacity pieces lying back to back (what the equal-split
 * all-to-all of shk_compact_owners_fixed buffers delivers, this rank's own piece included at skip_piece) in ONE
 * launch.  All or nothing: if ANY header says a range had more entries than the capacity, nothing is merged —
 * every rank receives a piece from every sender, so every rank decides alike — and the caller repeats the exchange
 * with exact counts (shk_owner_counts / shk_compact_owners_packed).  Asynchronous like shk_merge_entries;
 * shk_merge_pieces_max reports the largest header seen (valid after the next shk_finalize or shk_sync):
 * > capacity ⇒ nothing was merged. */
int shk_merge_pieces(shk_ctx *ctx, const void *d_buf, uint32_t n_pieces, uint64_t capacity, int32_t skip_piece);
int shk_merge_pieces_max(shk_ctx *ctx, uint64_t *max_count);
/* The owned range of shk_set_owned_pages — the part of the merged table (io.rs:1023-1028) whose histogram this
 * context contributes — as share `owner` of `n_owners` (a power of two) equal parts of the pages,
 * worked out when the histogram scan is launched — so it needs no look at the table geometry (which would have to
 * wait for the counting launches) and follows the table when a merge grows it. */
int shk_set_owner_share(shk_ctx *ctx, uint32_t n_owners, uint32_t owner);
/* Restrict finalize's histogram scan to pages [p0,p1) (owner shard). */
int shk_set_owned_pages(shk_ctx *ctx, uint64_t p0, uint64_t p1);

/* ---- key-space-partitioned ingest: the exchange between owner shares (device pointers) -------------
 *
 * One round, on every one of the W contexts (cfg.n_owners = W, cfg.owner_id = its rank):
 *   1. shk_xchg_scatter_device: the rank's batch (≤ SHK_XCHG_MAX_BASES bases, resident in HBM) is
 *      validated, its canonical k-mers (kmers_from_ascii, encoding.rs:332-371) extracted and their 4-byte
 *      records written, grouped by owner, into the context's exchange buffer: owner o's SEGMENT is
 *      d_records + o·layout.segment_records (u32 records) with its fill levels at d_cursors + o·layout.regions
 *      (u32 words) — one contiguous piece each, so "send every peer its segment" is one all-to-all with
 *      equal splits (or W−1 peer copies).  The context has TWO exchange buffers and takes them in turn: what a
 *      call hands out stays valid until the next call BUT ONE, so round r's segments can be on the links while
 *      round r + 1 is scattered.  The call returns when the kernel is done: *n_foreign_spilled
 *      is the fill of the context's foreign spill list (records that overflowed a region on skewed input;
 *      shk_xchg_spill), an invalid byte is reported here (SHK_ERR_INVALID_CHAR) and poisons the context.
 *   2. the caller moves segment o to rank o (RCCL all_to_all / hipMemcpyPeer), rank r's own segment
 *      needs no copy;
 *   3. shk_xchg_absorb on every received segment (and the own one): level-2 partition of its records into
 *      the context's (lane, page) regions, where they wait for the page pass like any deferred batch.
 *      Asynchronous on the context's stream (shk_stream): the segment must stay untouched until the
 *      stream has passed the call.
 * Foreign spills: shk_xchg_spill exposes the list (k-mer, chunk lane, count triples of any owner);
 * every rank hands every other rank's list to shk_insert_device (which drops what the context does
 * not own) and then calls shk_xchg_spill_clear.  All of it is exact for any input.
 * Needs 4-byte records at the exchange geometry: 2k − layout.log_p1 ≤ 32 (k ≤ 21 at the default
 * fan-out of 1024); otherwise SHK_ERR_STATE — take the wide round (shk_xchg_wide_scatter_device below) or merge
 * the tables at finalize instead (shk_merge_*). */
#define SHK_XCHG_MAX_BASES (1ull << 28)
typedef struct shk_xchg_layout {
  uint32_t n_owners;        /* W */
  uint32_t n_lanes;         /* chunk lanes (max(chunks, 1)) */
  uint32_t log_p1;          /* level-1 fan-out bits, owner bits included */
  uint32_t regions;         /* regions per owner segment: n_lanes << (log_p1 − log2 W), ordered [lane][super-page] */
  uint32_t region_cap;      /* records per region (a multiple of 1024); 4-byte records: the regions of a segment are
                               block-interleaved — record j of region g at ((j>>10)·regions + g)<<10 | (j & 1023); 8-byte
                               records: region g at g·region_cap, linear */
  uint32_t record_bytes;    /* 4: the low bits of the mixed key below the level-1 fan-out (2k − log_p1 ≤ 32: k ≤ 21);
                               8 (round 4): the canonical k-mer itself (k ≥ 18 on a two-level table) — runs are padded to
                               even length with the all-ones word; 0 reads as 4 */
  uint64_t segment_records; /* regions · region_cap */
} shk_xchg_layout;
/* layout_bases: what the segments are sized for — every rank of a round passes the same number
 * (≥ its n_bases; e.g. the round's batch size), so that all come out with the same layout; 0 = n_bases. */
int shk_xchg_scatter_device(shk_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_seqs,
                            uint64_t n_bases, uint64_t layout_bases, void **d_records, void **d_cursors,
                            shk_xchg_layout *layout, uint64_t *n_foreign_spilled);
/* The same scatter in two calls (round 4), for a caller that has something to put on the context's stream while the
 * scatter runs — the absorbs of the previous round's segments: _begin launches and returns at once, with the
 * segments' addresses and layout (functions of the configuration, not of the data), _end waits for the scatter
 * and reports what shk_xchg_scatter_device reports at its return (SHK_ERR_INVALID_CHAR and the poisoned context,
 * *n_foreign_spilled).  Between the two: shk_xchg_absorb only; the segments are complete when _end has returned.
 * With the one-call form the GPU stood idle every round from the end of the scatter until the host had woken up
 * and launched the absorbs (io.rs has no counterpart: it is the counting loop's batch boundary, io.rs:340-361). */
int shk_xchg_scatter_begin(shk_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_seqs,
                           uint64_t n_bases, uint64_t layout_bases, void **d_records, void **d_cursors,
                           shk_xchg_layout *layout);
int shk_xchg_scatter_end(shk_ctx *ctx, uint64_t *n_foreign_spilled);
int shk_xchg_absorb(shk_ctx *ctx, const void *d_records, const void *d_cursors, const shk_xchg_layout *layout);
int shk_xchg_spill(shk_ctx *ctx, void **d_kmers, void **d_lanes, void **d_counts, uint64_t *n);
int shk_xchg_spill_clear(shk_ctx *ctx);
/* The exchange round for k-mers the owner layout cannot take (4-byte records need 2k − layout.log_p1 ≤ 32: k ≤ 21 at
 * the default fan-out; ≤ 128 chunk lanes) — any k, any number of lanes, slower: the batch is validated, its canonical
 * k-mers (kmers_from_ascii, encoding.rs:332-371) are extracted as whole 64-bit values and grouped by owner:
 *   *d_kmers u64[Σ counts], owner o's k-mers at [Σ_{o'<o} counts[o'], …) in no particular order;
 *   *d_lanes u32[Σ counts], each k-mer's chunk lane (read i → lane (i / 1000) % n_chunks, io.rs:340-361);
 *   counts   u64[n_owners], on the host.
 * The caller moves owner o's piece of both arrays to rank o (all-to-all with these split sizes) and every rank hands
 * what it receives to shk_insert_device with d_counts = NULL (one occurrence each: KmerCounts::ingest_seq's
 * saturating add, counting.rs:82-85, by device-scope atomics).  The call returns when the arrays are complete; an
 * invalid byte is reported here (SHK_ERR_INVALID_CHAR) and poisons the context.  The arrays stay valid until the next
 * scatter call on the context.  shk_xchg_feasible: 1 when the owner-layout rounds (shk_xchg_scatter_device) can take
 * this context's batches at its present table geometry, 0 when only the wide round can. */
int shk_xchg_wide_scatter_device(shk_ctx *ctx, const void *d_bases, const void *d_offsets, uint64_t n_seqs,
                                 uint64_t n_bases, void **d_kmers, void **d_lanes, uint64_t *counts);
int shk_xchg_feasible(shk_ctx *ctx);
/* KmerCounts::insert (counting.rs:152-154) from device memory with a chunk lane per record:
 * saturating add of d_counts[i] to d_kmers[i] in lane d_lanes[i]; k-mers the context does not own
 * are dropped; d_counts = NULL: one occurrence each.  Synchronous. */
int shk_insert_device(shk_ctx *ctx, const void *d_kmers, const void *d_lanes, const void *d_counts, uint64_t n);
/* The context's HIP stream (hipStream_t): device work queued by the calls above is ordered on it, so
 * a host that runs its collectives on the same stream needs no host-side synchronisation. */
void *shk_stream(shk_ctx *ctx);

/* ---- pinned staging buffers for streaming hosts -------------------------------- */
void *shk_alloc_pinned(size_t bytes);
void shk_free_pinned(void *p);
void *shk_alloc_device(shk_ctx *ctx, size_t bytes);
void shk_free_device(shk_ctx *ctx, void *p);
/* Blocks of device and pinned host memory that a context (or shk_free_pinned) gives back are kept by the process —
 * a bounded amount: 4 GiB per process on the devices, 512 MiB pinned, no block above 1 GiB / 160 MiB — and handed to
 * the next request of about their size: mapping and unmapping memory is what setting a context up and taking it down
 * costs (17-19 ms of an 80 ms FASTQ job).  This call gives everything back to the driver; the environment variable
 * SHK_NO_MEM_CACHE=1 turns the keeping off.  (The reference has no counterpart: its allocator is the process's.) */
void shk_release_cached_memory(void);

/* ---- synthetic reads, generated on device ------------------------ */

typedef struct shk_synth {
  uint64_t seed_genome;   /* 0x5EED0001 */
  uint64_t seed_reads;    /* 0x5EED0002 */
  uint64_t genome_len;    /* G */
  uint32_t read_len;      /* 150 */
  uint32_t sub_per_64k;   /* substitution errors per 65536 bases (0 = error-free) */
  uint32_t n_per_64k;     /* N per 65536 bases */
  uint32_t reserved;
} shk_synth;
/* Writes reads [first_read, first_read+n_reads) as ASCII to d_bases
 * (n_reads*read_len bytes) and offsets to d_offsets (n_reads+1 u64). */
int shk_synth_reads_device(shk_ctx *ctx, const shk_synth *spec, uint64_t first_read,
                           uint64_t n_reads, void *d_bases, void *d_offsets);

/* ---- host side either side of the path --------------------------------- */

/* FASTQ(.gz) front-end: read_fastq / open_fastq_reader / validate_fastq_record
 * (io.rs:161-198, 271-352, 598-625).  Parses only; state (read count, validation cadence,
 * --max-reads) persists across files like FastqReadState (io.rs:498-512).  n_paths == 0 reads
 * stdin (io.rs:517-537).  Errors carry the reference's texts (shk_fastq_error). */
typedef struct shk_fastq shk_fastq;
int shk_fastq_open(const char *const *paths, uint32_t n_paths, uint64_t max_reads,
                   uint64_t validate_every, shk_fastq **out);
/* The same with flags.  SHK_FASTQ_GZIP_ALL_MEMBERS: NOT what the reference does — its flate2::read::GzDecoder
 * (io.rs:606-617) reads a gzip file's FIRST member and stops, which is what shk_fastq_open does too; a bgzip'd FASTQ
 * or `cat a.fastq.gz b.fastq.gz` then counts as its first member (64 KB of a bgzip file).  With the flag every member
 * is read, the way flate2::read::MultiGzDecoder would: what follows a member's trailer is the next member's header,
 * every member's CRC-32 and length are checked, an error in any of them is the stream's (same texts). */
#define SHK_FASTQ_GZIP_ALL_MEMBERS 1u
int shk_fastq_open_ex(const char *const *paths, uint32_t n_paths, uint64_t max_reads,
                      uint64_t validate_every, uint32_t flags, shk_fastq **out);
void shk_fastq_close(shk_fastq *r);
const char *shk_fastq_error(const shk_fastq *r);
/* Up to max_seqs sequences / bases_cap bytes, in input order; offsets[0] = 0.  A sequence that
 * does not fit is delivered first by the next call. */
int shk_fastq_next_batch(shk_fastq *r, uint8_t *bases, uint64_t bases_cap, uint64_t *offsets,
                         uint64_t max_seqs, uint64_t *n_seqs);
/* The same batch in the 2-bit packed input format (above: Read::from_str's layout over the batch's concatenated
 * bases + the N mask), packed while the sequences are copied out of the parsed file — what shk_pack_reads would make
 * of shk_fastq_next_batch's output, without the ASCII batch in between: packed takes (bases_cap+3)/4 bytes rounded up
 * to 8, nmask (bases_cap+31)/32 words; offsets are base indices from 0.  A byte outside ACGTN among the reads handed
 * out is SHK_ERR_INVALID_CHAR with the text of encoding.rs:353-356 (those reads are ones the reference would have
 * drained, io.rs:340-343, so it would have met the byte too). */
int shk_fastq_next_batch_packed(shk_fastq *r, uint8_t *packed, uint32_t *nmask, uint64_t bases_cap, uint64_t *offsets,
                                uint64_t max_seqs, uint64_t *n_seqs);
int shk_fastq_stats(const shk_fastq *r, uint64_t *n_reads_read, uint64_t *n_bases_read,
                    int *reached_max, int *done);

/* {sample}.histo / {sample}.final.histo, io.rs:1009-1014, 1051-1094 (byte for byte). */
int shk_write_histo(const char *path, const char *version, uint32_t k, uint32_t chunks,
                    uint64_t histo_max, const uint64_t *histo);
int shk_write_final_histo(const char *path, const char *version, uint32_t k, uint32_t chunks,
                          uint64_t histo_max, const uint64_t *histo);

/* RunStats (stats.rs:27-45) as filled at main.rs:182-197; pcr_results is always empty here. */
typedef struct shk_run_stats {
  const char *sharkmer_version;
  const char *command;
  const char *sample;
  uint32_t kmer_length;
  uint32_t chunks;
  uint64_t n_reads_read;
  uint64_t n_bases_read;
  uint64_t n_subreads_ingested;
  uint64_t n_bases_ingested;
  uint64_t n_kmers;
  uint64_t n_multi_kmers;      /* written only when has_histogram (chunks > 0) */
  uint64_t n_singleton_kmers;  /* idem */
  uint64_t peak_memory_bytes;  /* device memory in use at the end of the run */
  uint32_t has_histogram;
  uint32_t reserved;
} shk_run_stats;
int shk_write_stats_yaml(const char *path, const shk_run_stats *st); /* stats.rs:186-193 */

/* cli.rs:659-673 + 645-652: 0<k<32, k odd, 0<histo_max≤1e6, sample-name charset; the message is
 * available from shk_run_error(). */
int shk_validate_args(uint32_t k, uint64_t histo_max, const char *sample);
const char *shk_run_error(void);

/* ingest_reads + consolidate_and_histogram + write_stats over local files (main.rs:74-78,
 * 112-197 without sPCR): the flags -k --chunks --histo-max -m -s -o --validate-every. */
typedef struct shk_run_config {
  const char *const *inputs; /* FASTQ(.gz) paths; n_inputs == 0 ⇒ stdin */
  uint32_t n_inputs;
  uint32_t k;
  uint32_t chunks;
  int32_t device;
  uint64_t histo_max;
  uint64_t max_reads;       /* 0 = all */
  uint64_t validate_every;  /* 0 = first record only */
  const char *sample;
  const char *outdir;       /* NULL = "./" */
  const char *command;      /* recorded in stats.yaml */
  const char *version;      /* NULL = "3.1.0" */
  uint64_t table_capacity_hint;
  uint64_t batch_reads;     /* reads per device super-batch; 0 = 1,000,000 */
  uint64_t batch_bases;     /* pinned buffer bytes; 0 = 256 MiB */
  uint32_t n_devices;       /* > 1: one multi-device context over device_ids (shk_config.n_devices); else `device` */
  uint32_t fastq_flags;     /* SHK_FASTQ_* (shk_fastq_open_ex); 0 = the reference's reader */
  const int32_t *device_ids;
} shk_run_config;
int shk_run_files(const shk_run_config *cfg, shk_run_stats *out_stats);

#ifdef __cplusplus
}
#endif
#endif
