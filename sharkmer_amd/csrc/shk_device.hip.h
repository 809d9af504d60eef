// shk_device.hip.h — gfx950 device code of libshk (hand-written HIP, wave64).
//
// Data layout in HBM:
//   bases      u8 [n_bases]             concatenated ASCII reads (as handed over)
//   startbits  u32[n_bases/32+2]        bit p set ⇔ position p is the first base of a read
//   tiles      TileDesc[n_tiles]        ≤TILE_T-base spans, never crossing a 1000-read block
//   keys       u64[n_pages*PAGE_SLOTS]  canonical k-mer or EMPTY; page = top hash bits,
//                                       linear probing confined to the page
//   vals       u32[n_lanes][capacity]   per-chunk-lane counts, lane-major
//
// Reference semantics reproduced (file:line under /root/reference):
//   extraction   src/kmer/encoding.rs:332-371 (kmers_from_ascii), :374-376
//   counting     src/kmer/counting.rs:82-85 (entry().or_insert(0); saturating_add)
//   histogram    src/kmer/counting.rs:171-202 + src/kmer/histogram.rs:51-85,125-134
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stddef.h>
#include <stdint.h>

namespace shk {

constexpr uint64_t EMPTY = 0xFFFFFFFFFFFFFFFFull;  // never a k-mer: k<32 ⇒ key < 2^62
constexpr int PAGE_LOG = 13;
constexpr uint32_t PAGE_SLOTS = 1u << PAGE_LOG;     // 8192 slots: 64 KiB keys + 32 KiB counts in LDS
constexpr int WG = 256;                             // 4 waves
constexpr int TILE_T = 16384;                       // k-mer end positions per tile
constexpr int HALO = 32;                            // ≥ k-1 bases before the tile (k ≤ 31)
constexpr int TILE_LDS = TILE_T + HALO;               // code bytes of one staged tile
constexpr int TILE_GROUPS = TILE_LDS / 16;            // 16-base groups
constexpr int STAGE_BYTES = TILE_LDS + 4 * TILE_GROUPS;  // + one (N|start) mask word per group

struct TileDesc {
  uint64_t begin;  // first k-mer END position owned by the tile
  uint64_t end;    // one past the last
  uint32_t lane;   // chunk lane of the enclosing 1000-read block
  uint32_t pad;
};

struct DevStats {
  unsigned long long bad;          // min over (position<<8 | byte) of invalid bytes; ~0 = none
  unsigned long long bad_pad;      // (bad, bad_pad) = the control block's first 16-B word: reset fills it with ~0
  unsigned long long n_distinct;   // keys inserted so far (all launches)
  unsigned long long spill_count;  // entries in the spill list (this launch)
  unsigned long long n_tiles;      // written by k_build_tiles
  unsigned long long scratch[3];
};
static_assert(sizeof(DevStats) == 64 && offsetof(DevStats, bad) == 0, "control block layout");

struct TableRef {
  uint64_t *keys;
  uint32_t *vals;
  uint64_t cap;      // n_pages * PAGE_SLOTS
  uint32_t log_pages;
  uint32_t n_lanes;
  uint32_t key_bits;  // 2k
  // OWNER SHARE (key-space-partitioned ingest, SURVEY.md §8e "alternative when local tables do not
  // fit"): the table is the slice of owner `owner_id` — the pages whose top `owner_bits` bits of the
  // GLOBAL page index (top owner_bits + log_pages bits of the mixed key) equal owner_id — of a virtual
  // table of 2^(owner_bits + log_pages) pages.  owner_bits = 0: the whole key space (owner_id = 0).
  uint32_t owner_bits;
  uint32_t owner_id;
  uint32_t pad_;
};

struct BatchRef {
  const uint8_t *bases;
  const uint32_t *startbits;
  uint64_t n_bases;
  const TileDesc *tiles;  // nullptr ⇒ one block: tile t = [t*TILE_T, …) ∩ [0,n_bases), lane = lane0
  const DevStats *stats;  // n_tiles lives here when tiles != nullptr
  uint64_t n_tiles_single;
  uint64_t tile_first;  // this launch covers tiles [tile_first, tile_first + tile_count)
  uint64_t tile_count;
  uint32_t lane0;
  int k;
};

struct SpillRef {
  uint64_t *keys;
  uint32_t *lanes;
  uint32_t *counts;
  uint64_t cap;
  unsigned long long *count;  // the list's fill counter; nullptr = DevStats::spill_count (k_scatter32 only looks at this)
};

// ---- hashing.  Results never depend on it — only speed does, and on gfx950
// 32-bit integer multiplies issue at quarter rate while 24-bit ones (v_mad_u32_u24) are full
// rate.  So: multilinear hash of the key's three 24-bit chunks (full rate) + one xorshift-
// multiply finaliser.  Page = top log_pages bits, in-page home slot = the next 12.
// tools/hash_eval.py: page occupancy and slot collisions match a Poisson process on random,
// AT-rich, tandem-repeat and sequential keys (without the finaliser sequential keys collide).
constexpr uint32_t MAX_LOG_PAGES = 20;  // page bits + the 11 home-bucket bits ≤ 32 hash bits; 2^33 slots ≈ 103 GB
// mix_key: a BIJECTION of the 2k-bit key space — ONE multiplication by an odd constant mod 2^2k — so the
// position of a key in the table (page = top log_pages bits of the product, home bucket = the next 11
// bits) together with the remaining low bits identifies the key.  The 4-byte-record path of the paged
// counter ships only those low bits and rebuilds a key with unmix_key (the inverse multiplication) when
// it has to.  Everything the engine reads off the mixed key it reads off its TOP bits (owner, partition,
// page, bucket), and the top bits of a product depend on every bit of the key; the low bits, which depend
// on the key's low bits only, are just carried along as the fingerprint.  Up to 42 key bits (k ≤ 21: the
// 4-byte-record path, k_scatter32, which is bound by VALU issue) the multiplier is a 32-bit constant — a
// 64-bit product by it is one v_mad_u64_u32 + one v_mul_lo_u32 — beyond that a 64-bit one (longer keys need
// the wider constant for keys that differ in their low bits only to reach the top).  tools/hash_eval.py:
// page occupancy and bucket overflow match a Poisson process on random, AT-rich, tandem-repeat and
// sequential keys for k = 9…31 (round 1 used multiply–fold–multiply; one multiply measures the same
// spread and takes five instructions fewer per k-mer in the scatter's walk).
#ifndef SHK_MIX_M32
#define SHK_MIX_M32 0xC2B2AE35ull
#define SHK_MIX_M32_INV 0xD16E308E7ED1B41Dull
#endif
constexpr uint64_t MIX_M32 = SHK_MIX_M32, MIX_M64 = 0x9E3779B97F4A7C15ull;
constexpr uint64_t MIX_M32_INV = SHK_MIX_M32_INV, MIX_M64_INV = 0xF1DE83E19937733Dull;  // mod 2^64
static_assert((MIX_M32 * MIX_M32_INV) == 1ull && (MIX_M64 * MIX_M64_INV) == 1ull, "inverses mod 2^64");
#ifndef SHK_MIX_NARROW_BITS
#define SHK_MIX_NARROW_BITS 42
#endif
constexpr uint32_t MIX_NARROW_BITS = SHK_MIX_NARROW_BITS;
__host__ __device__ __forceinline__ uint64_t mix_key(uint64_t x, uint32_t bits) {
  const uint64_t mask = ~0ull >> (64 - bits);
  return (x * (bits <= MIX_NARROW_BITS ? MIX_M32 : MIX_M64)) & mask;
}
__host__ __device__ __forceinline__ uint64_t unmix_key(uint64_t y, uint32_t bits) {
  const uint64_t mask = ~0ull >> (64 - bits);
  return (y * (bits <= MIX_NARROW_BITS ? MIX_M32_INV : MIX_M64_INV)) & mask;
}
// the 32 hash bits the table geometry is read from: the top of the mixed key
__device__ __forceinline__ uint32_t hash64(uint64_t key, uint32_t bits) {
  return (uint32_t)((mix_key(key, bits) << (64 - bits)) >> 32);
}
// a full-avalanche hash for small open-addressing sets that index by LOW bits (k_filter_reads' primer set)
__host__ __device__ __forceinline__ uint32_t set_hash(uint64_t key) {
  key ^= key >> 31;
  return (uint32_t)((key * MIX_M64) >> 32);
}
__device__ __forceinline__ uint64_t page_of(uint32_t h, uint32_t log_pages) {
  return log_pages ? (uint64_t)(h >> (32 - log_pages)) : 0ull;
}
// Home slots are bucket-aligned (multiples of 4): a key is almost always found inside the 32-B
// bucket its probe sequence starts in, which k_pages reads with two 16-B LDS loads.
__device__ __forceinline__ uint32_t slot_of(uint32_t h, uint32_t log_pages) {
  return ((h >> (32 - (PAGE_LOG - 2) - log_pages)) & (PAGE_SLOTS / 4 - 1)) << 2;  // the 11 bits below the page bits
}
static_assert(MAX_LOG_PAGES + PAGE_LOG - 2 <= 32, "page + home-bucket bits come out of 32 hash bits");
// Where a key lives in a table that may be an owner share: page (as a slot base in the LOCAL arrays),
// home slot, and whether this table owns the key at all.
struct Home {
  uint64_t base;
  uint32_t slot;
  bool owned;
};
__device__ __forceinline__ Home home_of(const TableRef &tb, uint64_t key) {
  const uint32_t h = hash64(key, tb.key_bits);
  const uint32_t lpg = tb.log_pages + tb.owner_bits;  // page bits of the virtual global table
  const uint64_t gp = page_of(h, lpg);
  Home r;
  r.owned = (uint32_t)(gp >> tb.log_pages) == tb.owner_id;
  r.base = (gp & ((1ull << tb.log_pages) - 1ull)) << PAGE_LOG;
  r.slot = slot_of(h, lpg);
  return r;
}

__device__ __forceinline__ uint32_t sat_add_u32(uint32_t a, uint32_t b) {
  uint32_t s = a + b;
  return s < a ? 0xFFFFFFFFu : s;
}

// ---- tile geometry -------------------------------------------------------------------
__device__ __forceinline__ bool tile_get(const BatchRef &b, uint64_t t, uint64_t &t0, uint64_t &t1,
                                         uint32_t &lane) {
  if (t >= b.tile_count) return false;
  t += b.tile_first;
  if (b.tiles) {
    if (t >= b.stats->n_tiles) return false;
    TileDesc d = b.tiles[t];
    t0 = d.begin;
    t1 = d.end;
    lane = d.lane;
  } else {
    if (t >= b.n_tiles_single) return false;
    t0 = t * (uint64_t)TILE_T;
    t1 = t0 + TILE_T < b.n_bases ? t0 + TILE_T : b.n_bases;
    lane = b.lane0;
  }
  return true;
}

// Advance t (stride gridDim.x) to this workgroup's next tile, optionally only tiles of one
// chunk lane.  Uniform across the workgroup.
__device__ __forceinline__ bool next_tile(const BatchRef &b, uint64_t &t, bool use_filter,
                                          uint32_t lane_filter, uint64_t &t0, uint64_t &t1,
                                          uint32_t &lane) {
  while (tile_get(b, t, t0, t1, lane)) {
    if (!use_filter || lane == lane_filter) return true;
    t += gridDim.x;
  }
  return false;
}

// ASCII → 2-bit base for 4 bytes at once: A,C,G,T → 0..3 via ((c>>1)^(c>>2))&3 ('N' → 0);
// *nbytes gets 0x80 in every byte that is 'N' — for VALID input: among A,C,G,T,N only N has
// bit 3 set.  (Input with any other byte is rejected as a whole, encoding.rs:353-356: nothing
// derived from it is ever used.)
__device__ __forceinline__ uint32_t codes4(uint32_t w, uint32_t *nbytes) {
  *nbytes = (w << 4) & 0x80808080u;
  return ((w >> 1) ^ (w >> 2)) & 0x03030303u;
}
// the four byte-MSBs of z gathered into bits 0-3, byte order kept
__device__ __forceinline__ uint32_t msb_gather4(uint32_t z) { return (z * 0x00204081u) >> 28; }
// Bytes of w that are none of A,C,G,T,N (non-zero result ⇔ some byte is invalid,
// encoding.rs:341-356), given w's codes c: a byte is valid exactly when it equals the letter
// that its own (bit 3, 2-bit code) stands for, and that letter comes out of an 8-entry byte table
// in one v_perm_b32 — selectors 0-3 → A C G T, 4 → N (bit 3 set, code 0), 5-7 → 0x00, which no
// byte with bit 3 set equals.
__device__ __forceinline__ uint32_t invalid_bytes4(uint32_t w, uint32_t c) {
  const uint32_t sel = c | ((w >> 1) & 0x04040404u);
  return w ^ __builtin_amdgcn_perm(0x0000004Eu /* . . . N */, 0x54474341u /* T G C A */, sel);
}

// Is byte c one of A,C,G,T,N?  (A=0x41 C=0x43 G=0x47 N=0x4E T=0x54 → bits 1,3,7,14,20 of
// the 0x40..0x5F word.)  encoding.rs:341-356.
__device__ __forceinline__ bool byte_is_acgtn(uint32_t c) {
  return (c >> 5) == 2 && ((0x0010408Au >> (c & 31)) & 1u);
}
// 2-bit packed copies of a staged tile (16 bases per u32, first base on top): PACK_WORDS words
// each — the forward stream, and the mirrored complement stream in which base j of the tile is
// base TILE_LDS-1-j, so that the reverse complement of a window is a window again.
constexpr int PACK_WORDS = TILE_GROUPS + 2;  // + 2 pad words (a window read touches 3 words)
static_assert(TILE_T <= (1 << 14) && TILE_LDS % 16 == 0, "sorted entries: 14-bit position + strand bit");
__device__ __forceinline__ uint32_t pack4(uint32_t c) {  // 4 code bytes (0..3 each) → 8 bits, first base on top
  return (c * 0x40100401u) >> 24;                        // the four 2-bit fields land in bits 31..24, no carries
}
__device__ __forceinline__ uint32_t pack16(uint4 c4) {
  return (pack4(c4.x) << 24) | (pack4(c4.y) << 16) | (pack4(c4.z) << 8) | pack4(c4.w);
}
// reverse the order of the sixteen 2-bit fields of x
__device__ __forceinline__ uint32_t rev2(uint32_t x) {
  x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
  return __builtin_bswap32(x);
}

// Stage positions [t0-HALO, t0+TILE_T) of the batch into LDS as one code byte per base:
// bits 0-1 = base, bit 2 = "a valid k-mer ENDS here", i.e. the k bases up to and including
// this one lie in one read and none is N (encoding.rs:346-352,363: n_valid ≥ k).  The bit is
// computed bit-parallel for 16 positions at a time from per-group masks of N positions and
// read-start positions (smeared over k resp. k-1 positions), so the walks below carry no
// per-base validity state at all.
// VALIDATE: additionally check every byte of [t0,t1) against ACGTN (encoding.rs:353-356;
// the first offender in input order is reported through stats->bad) and return this
// thread's count of non-N bytes in [t0,t1) (count_valid_bases, encoding.rs:374-376).
// Contains one __syncthreads(); the caller adds another before reading the codes.
// Bytewise staging of a 16-base group that touches the start or the end of the batch (a few
// groups per launch).  Kept out of line: inlined into the unrolled staging loop it costs the
// callers > 100 VGPRs.  Returns x = the 16 base codes, 2 bits each (base i at bits 2i),
// y = the group's N / outside-the-batch mask, z = its non-N count (VALIDATE only).
__device__ __noinline__ uint4 stage_edge_group(const uint8_t *bases, uint64_t n_bases, int64_t p,
                                               uint64_t t0, uint64_t t1, bool validate,
                                               DevStats *stats) {
  uint32_t codes = 0, nmask = 0, n_non_n = 0;
  for (int r = 0; r < 16; ++r) {
    int64_t pp = p + r;
    uint32_t c = 0, isn = 1;  // outside the batch: behaves like N
    if (pp >= 0 && (uint64_t)pp < n_bases) {
      uint32_t a = bases[pp];
      isn = a == 'N';
      c = isn ? 0u : (((a >> 1) ^ (a >> 2)) & 3u);
      if (validate && (uint64_t)pp >= t0 && (uint64_t)pp < t1) {
        if (!byte_is_acgtn(a)) atomicMin(&stats->bad, ((unsigned long long)pp << 8) | a);
        n_non_n += (a != 'N');
      }
    }
    codes |= c << (2 * r);
    nmask |= isn << r;
  }
  return make_uint4(codes, nmask, n_non_n, 0u);
}

// Raw bytes + read-start words of one thread's share of a tile, loaded ahead of time so that
// the HBM latency of tile i+1 hides behind the processing of tile i.
template <int NT, int TT = TILE_T>
struct StageRegs {
  static constexpr int GROUPS = (TT + HALO) / 16;  // 16-base groups of one staged tile of TT end positions
  static constexpr int R = (GROUPS + NT - 1) / NT;
  uint32_t raw[R][4];
  uint32_t sb0[R], sb1[R];
};
template <int NT, int TT = TILE_T>
__device__ __forceinline__ void stage_prefetch(const BatchRef &b, uint64_t t0, StageRegs<NT, TT> &pre) {
  constexpr int TILE_GROUPS = StageRegs<NT, TT>::GROUPS;
  const int64_t p0 = (int64_t)t0 - HALO;
#pragma unroll
  for (int r = 0; r < StageRegs<NT, TT>::R; ++r) {
    const int m = threadIdx.x + r * NT;
    const int64_t p = p0 + (int64_t)m * 16;
    if (m < TILE_GROUPS && p >= 0 && (uint64_t)p + 16 <= b.n_bases) {
      __builtin_memcpy(pre.raw[r], b.bases + p, 16);  // unaligned 16-B global load (one dwordx4)
      pre.sb0[r] = b.startbits[(uint64_t)p >> 5];
      pre.sb1[r] = b.startbits[((uint64_t)p >> 5) + 1];
    }
  }
}

// NOBYTES (k_scatter32): no code byte per base at all — the first pass writes the 2-bit packed stream
// straight from its registers, the second one 16-bit word of "k-mer ends here" bits per group
// (okbits); `lds` then only holds the group masks.  The walk reads one word + one half-word per thread.
template <bool VALIDATE, int NT = WG, bool PACK = false, int TT = TILE_T, bool NOBYTES = false>
__device__ __forceinline__ uint32_t stage_tile(const BatchRef &b, uint64_t t0, uint64_t t1,
                                               uint8_t *lds, DevStats *stats,
                                               const StageRegs<NT, TT> &pre, uint32_t *packed = nullptr,
                                               uint32_t *rcpacked = nullptr, uint16_t *okbits = nullptr) {
  constexpr int TILE_GROUPS = StageRegs<NT, TT>::GROUPS;  // (shadow the file-scope constants: this
  constexpr int TILE_LDS = TT + HALO;                      // function stages tiles of TT positions)
  uint32_t *gmask = reinterpret_cast<uint32_t *>(NOBYTES ? lds : lds + TILE_LDS);  // nmask16 | smask16<<16
  const int64_t p0 = (int64_t)t0 - HALO;
  uint32_t n_non_n = 0;
#pragma unroll
  for (int r = 0; r < StageRegs<NT, TT>::R; ++r) {
    const int m = threadIdx.x + r * NT;
    if (m >= TILE_GROUPS) continue;  // (no break: keeps r a compile-time index into `pre`)
    int64_t p = p0 + (int64_t)m * 16;
    uint32_t w[4];
    uint32_t nmask = 0;
    const bool fast = p >= 0 && (uint64_t)p + 16 <= b.n_bases;  // same test as stage_prefetch
    if (fast) {
      w[0] = pre.raw[r][0];
      w[1] = pre.raw[r][1];
      w[2] = pre.raw[r][2];
      w[3] = pre.raw[r][3];
      if (VALIDATE && (uint64_t)p >= t0 && (uint64_t)p < t1) {
        if ((uint64_t)p + 16 <= t1) {
          uint32_t bad = 0, nn = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint32_t z;
            const uint32_t c = codes4(w[q], &z);  // (the conversion below recomputes it: CSE)
            bad |= invalid_bytes4(w[q], c);
            nn += __builtin_popcount(z);
          }
          n_non_n += 16 - nn;
          if (bad) {
            for (int r = 0; r < 16; ++r) {
              uint32_t c = (w[r >> 2] >> (8 * (r & 3))) & 0xFF;
              if (!byte_is_acgtn(c)) {
                atomicMin(&stats->bad, ((unsigned long long)(p + r) << 8) | c);
                break;
              }
            }
          }
        } else {
          for (int r = 0; r < 16 && (uint64_t)p + r < t1; ++r) {
            uint32_t c = (w[r >> 2] >> (8 * (r & 3))) & 0xFF;
            if (!byte_is_acgtn(c)) atomicMin(&stats->bad, ((unsigned long long)(p + r) << 8) | c);
            n_non_n += (c != 'N');
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t z;
        w[q] = codes4(w[q], &z);
        nmask |= msb_gather4(z) << (4 * q);
      }
    } else {
      uint4 e = stage_edge_group(b.bases, b.n_bases, p, t0, t1, VALIDATE, stats);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t c8 = (e.x >> (8 * q)) & 0xFFu;  // four 2-bit codes
        w[q] = (c8 & 3u) | (((c8 >> 2) & 3u) << 8) | (((c8 >> 4) & 3u) << 16) | (((c8 >> 6) & 3u) << 24);
      }
      nmask = e.y;
      n_non_n += e.z;
    }
    // read-start flags for these 16 positions: bits [p, p+16) of startbits
    uint32_t f = 0;
    if (fast) {
      uint64_t two = (uint64_t)pre.sb0[r] | ((uint64_t)pre.sb1[r] << 32);
      f = (uint32_t)(two >> ((uint32_t)p & 31)) & 0xFFFFu;
    } else if (p >= 0 && (uint64_t)p < b.n_bases) {
      uint64_t wi = (uint64_t)p >> 5;
      uint32_t sh = (uint32_t)p & 31;
      uint64_t two = (uint64_t)b.startbits[wi] | ((uint64_t)b.startbits[wi + 1] << 32);
      f = (uint32_t)(two >> sh) & 0xFFFFu;
    } else if (p < 0 && p + 16 > 0) {  // straddles position 0
      uint32_t lo = b.startbits[0];
      f = (lo << (uint32_t)(-p)) & 0xFFFFu;
    }
    if (NOBYTES) packed[m] = pack16(make_uint4(w[0], w[1], w[2], w[3]));
    else *reinterpret_cast<uint4 *>(lds + m * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    gmask[m] = nmask | (f << 16);
    __builtin_amdgcn_sched_barrier(0);  // one group at a time: interleaving the unrolled
  }                                     // iterations costs > 100 VGPRs
  __syncthreads();
  // "k-mer ends here" bits: position j is bad if an N lies in [j-k+1, j] or a read starts in
  // [j-k+2, j].  48 mask bits (this group and the two before it) are smeared upwards.  Positions
  // at or beyond t1 (a tile that ends at a 1000-read block boundary) never get the bit, so the
  // walks need no end-of-tile test.  PACK: the 2-bit packed streams are written on the way.
  const int k = b.k;
  const int n_lds = (int)(t1 - t0) + HALO;  // LDS positions below this may end a k-mer
  if (NOBYTES) {
    if (threadIdx.x < 2) {  // the stream's pad words; halo groups never end a k-mer
      packed[TILE_GROUPS + threadIdx.x] = 0;
      okbits[threadIdx.x] = 0;
    }
  } else if (PACK) {
    for (int m = threadIdx.x; m < 2 + 2; m += NT) {  // halo groups 0,1 and the streams' pad words
      if (m < 2) {
        const uint32_t v = pack16(*reinterpret_cast<const uint4 *>(lds + m * 16));
        packed[m] = v;
        if (rcpacked) rcpacked[TILE_GROUPS - 1 - m] = rev2(~v);
      } else {
        packed[TILE_GROUPS + m - 2] = 0;
        if (rcpacked) rcpacked[TILE_GROUPS + m - 2] = 0;
      }
    }
  }
  for (int m = 2 + threadIdx.x; m < TILE_GROUPS; m += NT) {
    const uint32_t g0 = gmask[m - 2], g1 = gmask[m - 1], g2 = gmask[m];
    uint64_t N = (uint64_t)(g0 & 0xFFFFu) | ((uint64_t)(g1 & 0xFFFFu) << 16) |
                 ((uint64_t)(g2 & 0xFFFFu) << 32);
    uint64_t S = (uint64_t)(g0 >> 16) | ((uint64_t)(g1 >> 16) << 16) | ((uint64_t)(g2 >> 16) << 32);
    int s = 1;
    while (2 * s <= k) {
      N |= N << s;
      s *= 2;
    }
    if (k > s) N |= N << (k - s);
    if (k >= 2) {
      s = 1;
      while (2 * s <= k - 1) {
        S |= S << s;
        s *= 2;
      }
      if (k - 1 > s) S |= S << (k - 1 - s);
    } else {
      S = 0;
    }
    uint32_t ok = ~(uint32_t)((N | S) >> 32) & 0xFFFFu;  // m ≥ 2: ≥ 32 ≥ k-1 bases of history
    const int lim = n_lds - m * 16;                       // positions of this group below t1
    if (lim < 16) ok = lim > 0 ? ok & ((1u << lim) - 1u) : 0u;
    if (NOBYTES) {
      okbits[m] = (uint16_t)ok;
    } else if (PACK || ok) {
      uint4 c4 = *reinterpret_cast<const uint4 *>(lds + m * 16);
      if (PACK) {
        const uint32_t v = pack16(c4);
        packed[m] = v;
        if (rcpacked) rcpacked[TILE_GROUPS - 1 - m] = rev2(~v);
      }
      if (ok) {
        uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t fq = (ok >> (4 * q)) & 0xF;
          cw[q] |= ((fq & 1) << 2) | ((fq & 2) << 9) | ((fq & 4) << 16) | ((fq & 8) << 23);
        }
        *reinterpret_cast<uint4 *>(lds + m * 16) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
      }
    }
  }
  return n_non_n;
}

// ---- k_scatter32's staging: one 16-base group per thread, kept in REGISTERS -------------------------------
// Thread t owns group t + 2 of the staged tile — exactly the sixteen end positions it walks — so the
// group's packed bases and its "a k-mer ends here" bits never leave its registers; only what its
// NEIGHBOURS need (the packed word for the walk's warm-up window, the N / read-start masks for the
// validity smear) goes through LDS, as one (packed, masks) pair per group, behind ONE barrier.
// Threads 0 and 1 stage the two halo groups as well.  (stage_tile's generic form — code bytes or packed
// streams in LDS, a second pass for the validity bits behind a second barrier — serves the other kernels.)
template <int NT, int TT>
struct StageRegs32 {
  uint32_t raw[2][4];
  uint32_t sb0[2], sb1[2];
};
// group index of (thread, slot): slot 0 = the thread's own group, slot 1 = a halo group (threads 0, 1)
template <int NT>
__device__ __forceinline__ int stage32_group(int slot) { return slot == 0 ? (int)threadIdx.x + 2 : (int)threadIdx.x; }
template <int NT, int TT>
__device__ __forceinline__ void stage32_prefetch(const BatchRef &b, uint64_t t0, StageRegs32<NT, TT> &pre) {
  const int64_t p0 = (int64_t)t0 - HALO;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r == 1 && threadIdx.x >= 2) break;
    const int m = stage32_group<NT>(r);
    const int64_t p = p0 + (int64_t)m * 16;
    if (p >= 0 && (uint64_t)p + 16 <= b.n_bases) {
      __builtin_memcpy(pre.raw[r], b.bases + p, 16);  // unaligned 16-B global load (one dwordx4)
      pre.sb0[r] = b.startbits[(uint64_t)p >> 5];
      pre.sb1[r] = b.startbits[((uint64_t)p >> 5) + 1];
    }
  }
}
// One group: validate + count non-N bases (positions in [t0, t1) only), 2-bit pack, N / read-start masks.
// Returns the packed word (first base on top); *gm = nmask16 | startmask16 << 16; *nn += non-N bases.
__device__ __forceinline__ uint32_t stage32_group_regs(const BatchRef &b, uint64_t t0, uint64_t t1, int m, const uint32_t (&raw)[4],
                                                       uint32_t sb0, uint32_t sb1, DevStats *stats, uint32_t *gm, uint32_t *nn) {
  const int64_t p = (int64_t)t0 - HALO + (int64_t)m * 16;
  uint32_t w[4];
  uint32_t nmask = 0;
  const bool fast = p >= 0 && (uint64_t)p + 16 <= b.n_bases;  // same test as stage32_prefetch
  if (fast) {
    w[0] = raw[0], w[1] = raw[1], w[2] = raw[2], w[3] = raw[3];
    if ((uint64_t)p >= t0 && (uint64_t)p < t1) {
      if ((uint64_t)p + 16 <= t1) {
        uint32_t bad = 0, n_n = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t z;
          const uint32_t c = codes4(w[q], &z);  // (the conversion below recomputes it: CSE)
          bad |= invalid_bytes4(w[q], c);
          n_n += __builtin_popcount(z);
        }
        *nn += 16 - n_n;
        if (bad) {
          for (int r = 0; r < 16; ++r) {
            const uint32_t c = (w[r >> 2] >> (8 * (r & 3))) & 0xFF;
            if (!byte_is_acgtn(c)) {
              atomicMin(&stats->bad, ((unsigned long long)(p + r) << 8) | c);
              break;
            }
          }
        }
      } else {
        for (int r = 0; r < 16 && (uint64_t)p + r < t1; ++r) {
          const uint32_t c = (w[r >> 2] >> (8 * (r & 3))) & 0xFF;
          if (!byte_is_acgtn(c)) atomicMin(&stats->bad, ((unsigned long long)(p + r) << 8) | c);
          *nn += (c != 'N');
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t z;
      w[q] = codes4(w[q], &z);
      nmask |= msb_gather4(z) << (4 * q);
    }
  } else {
    const uint4 e = stage_edge_group(b.bases, b.n_bases, p, t0, t1, true, stats);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t c8 = (e.x >> (8 * q)) & 0xFFu;  // four 2-bit codes
      w[q] = (c8 & 3u) | (((c8 >> 2) & 3u) << 8) | (((c8 >> 4) & 3u) << 16) | (((c8 >> 6) & 3u) << 24);
    }
    nmask = e.y;
    *nn += e.z;
  }
  uint32_t f = 0;  // read-start flags for these 16 positions: bits [p, p+16) of startbits
  if (fast) {
    const uint64_t two = (uint64_t)sb0 | ((uint64_t)sb1 << 32);
    f = (uint32_t)(two >> ((uint32_t)p & 31)) & 0xFFFFu;
  } else if (p >= 0 && (uint64_t)p < b.n_bases) {
    const uint64_t wi = (uint64_t)p >> 5;
    const uint64_t two = (uint64_t)b.startbits[wi] | ((uint64_t)b.startbits[wi + 1] << 32);
    f = (uint32_t)(two >> ((uint32_t)p & 31)) & 0xFFFFu;
  } else if (p < 0 && p + 16 > 0) {  // straddles position 0
    f = (b.startbits[0] << (uint32_t)(-p)) & 0xFFFFu;
  }
  *gm = nmask | (f << 16);
  return pack16(make_uint4(w[0], w[1], w[2], w[3]));
}
// "a k-mer ends here" bits of a group from its own masks g2 and those of the two groups before it
// (position j is bad if an N lies in [j-k+1, j] or a read starts in [j-k+2, j]); positions at or beyond
// the tile's end never get the bit.
__device__ __forceinline__ uint32_t stage32_okbits(uint32_t g0, uint32_t g1, uint32_t g2, int k, int lim) {
  uint64_t N = (uint64_t)(g0 & 0xFFFFu) | ((uint64_t)(g1 & 0xFFFFu) << 16) | ((uint64_t)(g2 & 0xFFFFu) << 32);
  uint64_t S = (uint64_t)(g0 >> 16) | ((uint64_t)(g1 >> 16) << 16) | ((uint64_t)(g2 >> 16) << 32);
  // an N at p spoils the end positions [p, p+k-1], a read start at p the end positions [p, p+k-2]: ONE
  // smear of N | S over k-1 positions, plus the N's themselves k-1 further on
  uint64_t T = k >= 2 ? N | S : N;
  if (k >= 2) {
    int s = 1;
    while (2 * s <= k - 1) {
      T |= T << s;
      s *= 2;
    }
    if (k - 1 > s) T |= T << (k - 1 - s);
    T |= N << (k - 1);
  }
  uint32_t ok = ~(uint32_t)(T >> 32) & 0xFFFFu;  // ≥ 32 ≥ k-1 bases of history
  if (lim < 16) ok = lim > 0 ? ok & ((1u << lim) - 1u) : 0u;
  return ok;
}

// workgroup sum of a per-thread u32, result valid in thread 0 (red: WG/64 words of LDS)
template <int NT = WG>
__device__ __forceinline__ uint32_t wg_sum(uint32_t v, uint32_t *red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t s = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < NT / 64; ++w) s += red[w];
  return s;
}

// Rolling frames of encoding.rs:359-367, kept as 32-bit halves so that every step is a
// handful of full-rate 32-bit ops: the forward frame right-aligned and masked to 2k bits, the
// reverse-complement frame LEFT-aligned in 64 bits (its new base always enters at bit 62).
struct Roll {
  uint32_t f_lo, f_hi, r_lo, r_hi;
};
__device__ __forceinline__ void roll_step(Roll &x, uint32_t base, uint32_t mask_lo, uint32_t mask_hi) {
  x.f_hi = __builtin_amdgcn_alignbit(x.f_hi, x.f_lo, 30) & mask_hi;
  x.f_lo = ((x.f_lo << 2) | base) & mask_lo;
  x.r_lo = __builtin_amdgcn_alignbit(x.r_hi, x.r_lo, 2);
  x.r_hi = (x.r_hi >> 2) | ((base ^ 3u) << 30);
}
__device__ __forceinline__ uint64_t roll_canonical(const Roll &x, int k) {
  uint64_t fwd = ((uint64_t)x.f_hi << 32) | x.f_lo;
  uint64_t rev = (((uint64_t)x.r_hi << 32) | x.r_lo) >> (64 - 2 * k);
  return fwd < rev ? fwd : rev;
}

// Walk this thread's SPAN end positions of the staged tile and hand every canonical k-mer to
// emit(kmer).  The walk starts k-1 bases early (rounded down to a ds_read_b64 boundary) to
// warm the frames up; bit 2 of the code byte says where a k-mer may be emitted.
template <int NT = WG, class Emit>
__device__ __forceinline__ void walk_tile(const uint8_t *lds, uint64_t t0, uint64_t t1, int k,
                                          Emit &&emit) {
  constexpr int SPAN = TILE_T / NT;
  const uint64_t mask = (1ull << (2 * k)) - 1;
  const uint32_t mask_lo = (uint32_t)mask, mask_hi = (uint32_t)(mask >> 32);
  const int e0 = threadIdx.x * SPAN;  // first end position (tile-relative)
  const int n_end = (int)(t1 - t0);   // valid end positions in this tile
  if (e0 >= n_end) return;
  const int jemit = HALO + e0;
  const int jend = HALO + (e0 + SPAN < n_end ? e0 + SPAN : n_end);
  Roll x{0, 0, 0, 0};
  for (int j = (jemit - (k - 1)) & ~7; j < jemit; j += 8) {  // warm-up: no emission
    uint64_t w = *reinterpret_cast<const uint64_t *>(lds + j);
#pragma unroll
    for (int r = 0; r < 8; ++r) roll_step(x, (uint32_t)(w >> (8 * r)) & 3u, mask_lo, mask_hi);
  }
  for (int j = jemit; j < jend; j += 8) {
    uint64_t w = *reinterpret_cast<const uint64_t *>(lds + j);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint32_t c = (uint32_t)(w >> (8 * r)) & 0xFF;
      roll_step(x, c & 3u, mask_lo, mask_hi);
      if ((c & 4u) && j + r < jend) emit(roll_canonical(x, k));
    }
  }
}

// ---- open-addressing probe inside one page ----------------------------------------------
// Keys only ever go EMPTY → key, so a stale read can only show EMPTY where a key already
// sits; the device-scope CAS is the arbiter.  Returns the slot index or -1 (page full).
__device__ __forceinline__ int64_t find_or_insert(uint64_t *keys, uint64_t base, uint32_t s0,
                                                  uint64_t key, bool &inserted) {
  uint32_t s = s0;
  for (uint32_t i = 0; i < PAGE_SLOTS; ++i) {
    uint64_t cur = __hip_atomic_load(&keys[base + s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == key) return (int64_t)(base + s);
    if (cur == EMPTY) {
      uint64_t prev = atomicCAS((unsigned long long *)&keys[base + s], (unsigned long long)EMPTY,
                                (unsigned long long)key);
      if (prev == EMPTY) {
        inserted = true;
        return (int64_t)(base + s);
      }
      if (prev == key) return (int64_t)(base + s);
    }
    s = (s + 1) & (PAGE_SLOTS - 1);
  }
  return -1;
}

__device__ __forceinline__ int64_t find_slot(const uint64_t *keys, uint64_t base, uint32_t s0,
                                             uint64_t key) {
  uint32_t s = s0;
  for (uint32_t i = 0; i < PAGE_SLOTS; ++i) {
    uint64_t cur = keys[base + s];
    if (cur == key) return (int64_t)(base + s);
    if (cur == EMPTY) return -1;
    s = (s + 1) & (PAGE_SLOTS - 1);
  }
  return -1;
}

// Saturating add of an arbitrary delta (counting.rs:82-92 semantics) without a CAS loop (a CAS
// loop livelocks for seconds when millions of records hit one k-mer, e.g. poly-A input): a
// returning add, and if that add wrapped the slot (old > MAX - delta) a repair with
// atomicMax(MAX).  Every wrapping add is followed by its own repair and an add that lands after
// a repair wraps again and repairs again, so the last operation on a saturated slot leaves MAX.
__device__ __forceinline__ void sat_add_atomic(uint32_t *p, uint32_t delta) {
  if (!delta) return;
  uint32_t old = atomicAdd(p, delta);
  if (old > 0xFFFFFFFFu - delta) atomicMax(p, 0xFFFFFFFFu);
}

// ==========================================================================================
// K_FILL: up to four buffers set to a 64-bit pattern in ONE launch (table reset: keys = EMPTY,
// counts = 0, control block = initial state).  Every segment is 16-B aligned and a multiple of
// 16 B long; the segments are walked as one concatenated range of 16-B words.
// ==========================================================================================
struct FillSegs {
  void *ptr[4];
  uint64_t n16[4];  // length in 16-B words
  uint64_t val[4];  // 64-bit pattern
};
__global__ void __launch_bounds__(WG) k_fill(FillSegs f) {
  const uint64_t total = f.n16[0] + f.n16[1] + f.n16[2] + f.n16[3];
  for (uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x; i < total; i += (uint64_t)gridDim.x * WG) {
    uint64_t j = i;
    int sgm = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (sgm == q && j >= f.n16[q]) {
        j -= f.n16[q];
        sgm = q + 1;
      }
    ulonglong2 v;
    v.x = v.y = f.val[sgm];
    reinterpret_cast<ulonglong2 *>(f.ptr[sgm])[j] = v;
  }
}

// ==========================================================================================
// K_PACK / K_UNPACK: the 2-bit packed input format.  A batch's concatenated bases as ONE sequence in the
// layout of the reference's Read::from_str (src/kmer/encoding.rs:60-95): 4 bases per byte, first base in
// the two most significant bits (A 00, C 01, G 10, T 11), the tail left-aligned in its byte — plus what
// from_str has no room for and kmers_from_ascii needs (encoding.rs:346-352): an N mask, bit p % 32 of word
// p / 32 set where base p is N (its 2-bit code is then 00).  0.28 B/base instead of 1: what a host sends
// over PCIe when it packs (shk_pack_reads) before it hands a batch over (shk_ingest_packed).
// An N mask that is nearly all zeros crosses the link as the list of its non-zero words (ingest_host): the staged
// mask is cleared on the device and the listed words are written into it.
__global__ void __launch_bounds__(WG) k_nmask_sparse(uint32_t *__restrict__ nm, const uint2 *__restrict__ list, uint32_t n) {
  const uint32_t i = blockIdx.x * WG + threadIdx.x;
  if (i < n) nm[list[i].x] = list[i].y;
}

// k_unpack restores the ASCII batch in HBM for the counting kernels; k_pack is the device-side packer
// (validation as in encoding.rs:353-356: the first offender in input order through stats->bad).
// ==========================================================================================
__global__ void __launch_bounds__(WG) k_pack(const uint8_t *__restrict__ bases, uint64_t n_bases,
                                             uint8_t *__restrict__ packed, uint32_t *__restrict__ nmask,
                                             DevStats *__restrict__ stats) {
  const uint64_t g = (uint64_t)blockIdx.x * WG + threadIdx.x;  // one 32-base group = 8 packed bytes + 1 mask word
  const uint64_t p0 = g * 32;
  if (p0 >= n_bases) return;
  const uint32_t n = n_bases - p0 < 32 ? (uint32_t)(n_bases - p0) : 32u;
  uint32_t nm = 0;
  uint64_t bits = 0;  // 32 bases, first base on top
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t c = bases[p0 + i];
    if (!byte_is_acgtn(c)) atomicMin(&stats->bad, ((unsigned long long)(p0 + i) << 8) | c);
    const uint32_t isn = c == 'N';
    nm |= isn << i;
    bits |= (uint64_t)(isn ? 0u : (((c >> 1) ^ (c >> 2)) & 3u)) << (62 - 2 * i);
  }
  nmask[g] = nm;
  for (uint32_t j = 0; j < (n + 3) / 4; ++j) packed[g * 8 + j] = (uint8_t)(bits >> (56 - 8 * j));
}

// (a slice of a batch starts at a read boundary, not at a byte or word boundary of the streams: pk_base0, nm_base0)
__global__ void __launch_bounds__(WG) k_unpack(const uint8_t *__restrict__ packed, uint32_t pk_base0,
                                               const uint32_t *__restrict__ nmask, uint32_t nm_base0,
                                               uint64_t n_bases, uint8_t *__restrict__ out) {
  const uint64_t q = ((uint64_t)blockIdx.x * WG + threadIdx.x) * 16;  // 16 bases → one 16-byte store
  if (q >= n_bases) return;
  // output position 0 is base pk_base0 (0-3) of packed[0] and bit nm_base0 (0-31) of nmask[0]
  const uint64_t sp = pk_base0 + q;
  const uint64_t by = sp >> 2;
  const uint32_t r = (uint32_t)sp & 3u;
  // 16 bases = 32 bits from bit 2r of byte `by` on: five bytes, first base on top
  uint64_t five = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) five = (five << 8) | packed[by + j];  // (the staging buffers are padded: no read past the allocation)
  const uint32_t w = (uint32_t)(five >> (8 - 2 * r));
  const uint64_t sn = nm_base0 + q;
  const uint64_t nw = sn >> 5;
  const uint32_t nsh = (uint32_t)sn & 31u;
  const uint64_t two = (uint64_t)nmask[nw] | ((uint64_t)nmask[nw + 1] << 32);
  const uint32_t nm = (uint32_t)(two >> nsh) & 0xFFFFu;
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t c8 = (w >> (24 - 8 * j)) & 0xFFu, n4 = (nm >> (4 * j)) & 0xFu;
    // selector byte = 2-bit code | N bit << 2: 0-3 → A C G T, 4-7 → N
    const uint32_t sel = ((c8 >> 6) & 3u) | (((c8 >> 4) & 3u) << 8) | (((c8 >> 2) & 3u) << 16) | ((c8 & 3u) << 24) |
                         ((n4 & 1u) << 2) | ((n4 & 2u) << 9) | ((n4 & 4u) << 16) | ((n4 & 8u) << 23);
    o[j] = __builtin_amdgcn_perm(0x4E4E4E4Eu /* N N N N */, 0x54474341u /* T G C A */, sel);
  }
  if (q + 16 <= n_bases) {
    *reinterpret_cast<uint4 *>(out + q) = make_uint4(o[0], o[1], o[2], o[3]);
  } else {
    for (uint32_t i = 0; q + i < n_bases; ++i) out[q + i] = (uint8_t)(o[i >> 2] >> (8 * (i & 3)));
  }
}

// ==========================================================================================
// K_MARK: read-start bitmap.  One thread per read, NO atomics and no zeroed bitmap to start from:
// the reads are in offset order, so the first read of every 32-position word (the "leader")
// gathers the start bits of all reads that begin in that word, stores the word, and zero-fills
// the words up to the next leader's — together the leaders write every word of the bitmap
// exactly once.  (A memset + one atomicOr per read cost 27 µs per million reads; this costs 8.)
// ==========================================================================================
// It also clears the small per-launch state of the counting pass that follows it on the stream
// (partition cursors, spill counter), which saves two fill launches per batch.
__global__ void __launch_bounds__(WG) k_mark_starts(const uint64_t *__restrict__ offsets,
                                                    uint64_t n_seqs, uint64_t n_bases,
                                                    uint32_t *__restrict__ startbits, uint64_t sb_words,
                                                    unsigned int *__restrict__ zero_u32, uint32_t n_zero,
                                                    unsigned long long *__restrict__ zero_u64, uint64_t off_bias,
                                                    uint4 *__restrict__ zero_a16 = nullptr, uint32_t n_a16 = 0,
                                                    uint4 *__restrict__ zero_b16 = nullptr, uint32_t n_b16 = 0) {
  // off_bias: what the offsets are ahead of the batch's first byte by (a slice of a host batch is handed on with
  // the caller's own offsets: no re-based copy is made on the host)
  const uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
  for (uint64_t j = i; j < n_zero; j += (uint64_t)gridDim.x * WG) zero_u32[j] = 0;
  if (i == 0 && zero_u64) *zero_u64 = 0;
  // (the histogram and totals a finalize has read back: cleared here, on the way, instead of by a launch of their own)
  for (uint64_t j = i; j < n_a16; j += (uint64_t)gridDim.x * WG) zero_a16[j] = make_uint4(0u, 0u, 0u, 0u);
  for (uint64_t j = i; j < n_b16; j += (uint64_t)gridDim.x * WG) zero_b16[j] = make_uint4(0u, 0u, 0u, 0u);
  if (i >= n_seqs) return;
  const uint64_t o = offsets[i] - off_bias;
  const uint64_t w = o >> 5;
  if (i > 0 && ((offsets[i - 1] - off_bias) >> 5) == w) return;  // not the first read of its word
  uint32_t bits = 0;
  uint64_t j = i, oj = o;
  while (j < n_seqs && (oj >> 5) == w) {  // usually one iteration: reads are longer than a word
    const uint64_t ej = offsets[j + 1] - off_bias;
    if (ej > oj && oj < n_bases) bits |= 1u << (oj & 31);  // empty reads start nothing
    oj = ej;
    ++j;
  }
  uint64_t w_next = j < n_seqs ? oj >> 5 : sb_words;
  if (w_next > sb_words) w_next = sb_words;
  if (i == 0)
    for (uint64_t x = 0; x < w && x < sb_words; ++x) startbits[x] = 0;  // (offsets[0] is 0 in practice)
  if (w < sb_words) startbits[w] = bits;
  for (uint64_t x = w + 1; x < w_next; ++x) startbits[x] = 0;
}

// ==========================================================================================
// K_MARK (part 2): tile list for multi-chunk batches.  Blocks are the runs of reads between
// global read indices that are multiples of 1000 (io.rs:340-343, 355-361); every tile lies
// inside one block so it has one chunk lane.  Single workgroup; exclusive scan over blocks.
// ==========================================================================================
constexpr int TB_WG = 1024;
__global__ void __launch_bounds__(TB_WG) k_build_tiles(const uint64_t *__restrict__ offsets,
                                                       uint64_t n_seqs, uint64_t g0,
                                                       uint32_t n_chunks, uint64_t n_blocks,
                                                       TileDesc *__restrict__ tiles,
                                                       DevStats *__restrict__ stats, uint64_t off_bias) {
  __shared__ uint64_t sc[TB_WG];
  __shared__ uint64_t running;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  const uint64_t first = (g0 % 1000 == 0) ? 1000 : 1000 - g0 % 1000;
  for (uint64_t jb = 0; jb < n_blocks; jb += TB_WG) {
    uint64_t j = jb + threadIdx.x;
    uint64_t nt = 0, b0 = 0, b1 = 0;
    uint32_t lane = 0;
    if (j < n_blocks) {
      uint64_t r0 = j == 0 ? 0 : first + (j - 1) * 1000;
      uint64_t r1 = first + j * 1000;
      if (r1 > n_seqs) r1 = n_seqs;
      b0 = offsets[r0] - off_bias;
      b1 = offsets[r1] - off_bias;
      nt = (b1 - b0 + TILE_T - 1) / TILE_T;
      lane = (uint32_t)(((g0 + r0) / 1000) % n_chunks);
    }
    sc[threadIdx.x] = nt;
    __syncthreads();
    for (int d = 1; d < TB_WG; d <<= 1) {  // inclusive Hillis–Steele scan
      uint64_t v = threadIdx.x >= (unsigned)d ? sc[threadIdx.x - d] : 0;
      __syncthreads();
      sc[threadIdx.x] += v;
      __syncthreads();
    }
    uint64_t excl = running + sc[threadIdx.x] - nt;
    for (uint64_t t = 0; t < nt; ++t) {
      TileDesc d;
      d.begin = b0 + t * TILE_T;
      d.end = d.begin + TILE_T < b1 ? d.begin + TILE_T : b1;
      d.lane = lane;
      d.pad = 0;
      tiles[excl + t] = d;
    }
    __syncthreads();
    if (threadIdx.x == TB_WG - 1) running += sc[TB_WG - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) stats->n_tiles = running;
}

// ==========================================================================================
// K_SCAN: validate bytes (encoding.rs:353-356) and count non-N bases per chunk lane
// (encoding.rs:374-376 via chunk.rs:28).  Runs before any counting kernel, so an invalid
// byte aborts the batch with the table untouched.
// ==========================================================================================
__global__ void __launch_bounds__(WG) k_scan(BatchRef b, DevStats *__restrict__ stats,
                                             unsigned long long *__restrict__ lane_bases) {
  __shared__ uint32_t red[WG / 64];
  for (uint64_t t = blockIdx.x;; t += gridDim.x) {
    uint64_t t0, t1;
    uint32_t lane;
    if (!tile_get(b, t, t0, t1, lane)) break;
    uint32_t n_valid = 0;
    for (uint64_t p = t0 + (uint64_t)threadIdx.x * 16; p < t1; p += (uint64_t)WG * 16) {
      uint8_t buf[16];
      int n = (int)(t1 - p < 16 ? t1 - p : 16);
      if (n == 16) {
        __builtin_memcpy(buf, b.bases + p, 16);
      } else {
        for (int r = 0; r < n; ++r) buf[r] = b.bases[p + r];
      }
      for (int r = 0; r < n; ++r) {
        uint32_t c = buf[r];
        if (!byte_is_acgtn(c)) {
          atomicMin(&stats->bad, ((unsigned long long)(p + r) << 8) | c);
        }
        n_valid += (c != 'N');
      }
    }
    // wave reduce, then one atomic per workgroup
    for (int off = 32; off > 0; off >>= 1) n_valid += __shfl_down(n_valid, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n_valid;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t s = 0;
      for (int w = 0; w < WG / 64; ++w) s += red[w];
      if (s) atomicAdd(&lane_bases[lane], (unsigned long long)s);
    }
    __syncthreads();
  }
}

// ==========================================================================================
// K_DIRECT: extract + insert with global device-scope atomics.  General path: any table
// size, any batch size; also the re-insert path for spilled k-mers.
// counting.rs:82-85: entry(kmer).or_insert(0); *c = c.saturating_add(1).
// Saturation stays exact with a returning add: an add that observes old == u32::MAX has
// wrapped the slot and repairs it with atomicMax(MAX); every wrap is followed by its own
// repair, so the last operation on a saturated slot always leaves MAX.
// ==========================================================================================
__device__ __forceinline__ void count_one(const TableRef &tb, uint64_t key, uint32_t lane,
                                          DevStats *stats, const SpillRef &sp,
                                          uint32_t &n_new) {
  const Home hm = home_of(tb, key);
  if (!hm.owned) return;  // another owner's k-mer (owner share: foreign records are dropped)
  bool inserted = false;
  int64_t s = find_or_insert(tb.keys, hm.base, hm.slot, key, inserted);
  if (s < 0) {
    unsigned long long i = atomicAdd(&stats->spill_count, 1ull);
    if (i < sp.cap) {
      sp.keys[i] = key;
      sp.lanes[i] = lane;
      sp.counts[i] = 1u;
    }
    return;
  }
  n_new += inserted;
  uint32_t *v = tb.vals + (uint64_t)lane * tb.cap + (uint64_t)s;
  uint32_t old = atomicAdd(v, 1u);
  if (old == 0xFFFFFFFFu) atomicMax(v, 0xFFFFFFFFu);
}

__global__ void __launch_bounds__(WG) k_direct(BatchRef b, TableRef tb, DevStats *__restrict__ stats,
                                               SpillRef sp) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[STAGE_BYTES];
  uint32_t n_new = 0;
  if (stats->bad != ~0ull) return;  // k_scan found an invalid byte: leave the table untouched
  uint64_t t = blockIdx.x, t0, t1;
  uint32_t lane;
  bool have = next_tile(b, t, false, 0, t0, t1, lane);
  StageRegs<WG> pre;
  if (have) stage_prefetch<WG>(b, t0, pre);
  while (have) {
    __syncthreads();
    stage_tile<false, WG>(b, t0, t1, lds, stats, pre);
    uint64_t tn = t + gridDim.x, n0, n1;
    uint32_t nl;
    const bool hn = next_tile(b, tn, false, 0, n0, n1, nl);
    if (hn) stage_prefetch<WG>(b, n0, pre);
    __syncthreads();
    walk_tile(lds, t0, t1, b.k, [&](uint64_t kmer) { count_one(tb, kmer, lane, stats, sp, n_new); });
    t = tn;
    t0 = n0;
    t1 = n1;
    lane = nl;
    have = hn;
  }
  for (int off = 32; off > 0; off >>= 1) n_new += __shfl_down(n_new, off, 64);
  if ((threadIdx.x & 63) == 0 && n_new) atomicAdd(&stats->n_distinct, (unsigned long long)n_new);
}

// ---- the WIDE exchange: k-mers too long for the owner layout's 4-byte records (k > 21 at a fan-out of 1024) ----------
// The batch's canonical k-mers (kmers_from_ascii, encoding.rs:332-371) as whole 64-bit values, grouped by OWNER — the
// top owner_bits of the mixed key, what home_of reads — with their chunk lanes beside them.  Two launches of this
// kernel: PLACE = false counts every owner's k-mers (owner_count[o] += …), the host turns the counts into segment
// bases, PLACE = true writes them (owner_count[o] then is owner o's running cursor, starting at its segment's base).
// Per tile: the k-mers are counted by owner in LDS, one device-scope add per (tile, owner) reserves a run, and a second
// walk of the same tile (walk_tile is deterministic) puts every k-mer at run base + its rank.
__device__ __forceinline__ uint32_t owner_of(uint64_t key, uint32_t key_bits, uint32_t owner_bits) {
  return owner_bits ? hash64(key, key_bits) >> (32 - owner_bits) : 0u;
}
template <bool PLACE>
__global__ void __launch_bounds__(WG) k_xw_scatter(BatchRef b, uint32_t key_bits, uint32_t owner_bits, DevStats *__restrict__ stats,
                                                   unsigned long long *__restrict__ owner_count, uint64_t *__restrict__ out_kmers,
                                                   uint32_t *__restrict__ out_lanes) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[STAGE_BYTES];
  __shared__ uint32_t cnt[64];
  __shared__ unsigned long long run_base[64];
  if (stats->bad != ~0ull) return;  // k_scan found an invalid byte: nothing is handed on
  const uint32_t W = 1u << owner_bits;
  uint64_t t = blockIdx.x, t0, t1;
  uint32_t lane;
  bool have = next_tile(b, t, false, 0, t0, t1, lane);
  StageRegs<WG> pre;
  if (have) stage_prefetch<WG>(b, t0, pre);
  while (have) {
    __syncthreads();
    stage_tile<false, WG>(b, t0, t1, lds, stats, pre);
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    uint64_t tn = t + gridDim.x, n0, n1;
    uint32_t nl;
    const bool hn = next_tile(b, tn, false, 0, n0, n1, nl);
    if (hn) stage_prefetch<WG>(b, n0, pre);
    __syncthreads();
    walk_tile(lds, t0, t1, b.k, [&](uint64_t kmer) { atomicAdd(&cnt[owner_of(kmer, key_bits, owner_bits)], 1u); });
    __syncthreads();
    if (threadIdx.x < W) {
      const uint32_t c1 = cnt[threadIdx.x];
      const unsigned long long at = c1 ? atomicAdd(&owner_count[threadIdx.x], (unsigned long long)c1) : 0ull;
      if (PLACE) {
        run_base[threadIdx.x] = at;
        cnt[threadIdx.x] = 0;
      }
    }
    if (PLACE) {
      __syncthreads();
      walk_tile(lds, t0, t1, b.k, [&](uint64_t kmer) {
        const uint32_t o = owner_of(kmer, key_bits, owner_bits);
        const unsigned long long at = run_base[o] + atomicAdd(&cnt[o], 1u);
        out_kmers[at] = kmer;
        out_lanes[at] = lane;
      });
    }
    t = tn;
    t0 = n0;
    t1 = n1;
    lane = nl;
    have = hn;
  }
}

// ==========================================================================================
// K_INSERT: (kmer, lane, count) records — KmerCounts::insert (counting.rs:152-154) and the
// re-insert of spilled k-mers after a grow.  lanes == nullptr ⇒ all records use lane0;
// counts == nullptr ⇒ count 1.
// ==========================================================================================
__global__ void __launch_bounds__(WG) k_insert(const uint64_t *__restrict__ kmers,
                                               const uint32_t *__restrict__ lanes,
                                               const uint32_t *__restrict__ counts, uint64_t n,
                                               uint32_t lane0, TableRef tb,
                                               DevStats *__restrict__ stats, SpillRef sp) {
  uint32_t n_new = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x; i < n; i += (uint64_t)gridDim.x * WG) {
    uint64_t key = kmers[i];
    uint32_t lane = lanes ? lanes[i] : lane0;
    uint32_t cnt = counts ? counts[i] : 1u;
    const Home hm = home_of(tb, key);
    if (!hm.owned) continue;
    bool inserted = false;
    int64_t s = find_or_insert(tb.keys, hm.base, hm.slot, key, inserted);
    if (s < 0) {
      unsigned long long j = atomicAdd(&stats->spill_count, 1ull);
      if (j < sp.cap) {
        sp.keys[j] = key;
        sp.lanes[j] = lane;
        sp.counts[j] = cnt;
      }
      continue;
    }
    n_new += inserted;
    sat_add_atomic(tb.vals + (uint64_t)lane * tb.cap + (uint64_t)s, cnt);
  }
  for (int off = 32; off > 0; off >>= 1) n_new += __shfl_down(n_new, off, 64);
  if ((threadIdx.x & 63) == 0 && n_new) atomicAdd(&stats->n_distinct, (unsigned long long)n_new);
}

// ==========================================================================================
// K_GROW: rehash every occupied slot of the old table into a table with more pages.  A
// page's entries land in its 2^d child pages, each of which has room for all of them.
// ==========================================================================================
__global__ void __launch_bounds__(WG) k_grow(TableRef oldt, TableRef newt) {
  for (uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x; i < oldt.cap;
       i += (uint64_t)gridDim.x * WG) {
    uint64_t key = oldt.keys[i];
    if (key == EMPTY) continue;
    const Home hm = home_of(newt, key);
    bool inserted = false;
    int64_t s = find_or_insert(newt.keys, hm.base, hm.slot, key, inserted);
    // s >= 0 by construction (child pages cannot overflow)
    for (uint32_t l = 0; l < oldt.n_lanes; ++l)
      newt.vals[(uint64_t)l * newt.cap + (uint64_t)s] = oldt.vals[(uint64_t)l * oldt.cap + i];
  }
}

// ==========================================================================================
// K_HISTO: one pass over pages [page0,page1) → every histogram column + totals.
// Column j of the reference is the histogram of the table after merging chunks 0..j with
// saturating adds (counting.rs:171-202; histogram.rs:51-85) — sequential saturating adds of
// non-negative lane counts equal the clamped prefix sum, so column j is the histogram of
// min(Σ_{c≤j} lane_c, u32::MAX); counts > histo_max fold into bin histo_max+1
// (histogram.rs:125-134); zero prefix sums (k-mer not seen yet) are not counted.
// Low bins are privatised in LDS (u32), the rest go straight to global u64 atomics.  The grid is
// ≤ 256 workgroups of 1024 threads: enough loads in flight to stream the table, and few enough
// workgroups that the final flush (every workgroup adds its non-zero bins to the same few cache
// lines of `hist`) stays short — measured: 1024 × 256-thread workgroups spend 25 µs in that flush.
// ==========================================================================================
struct HistoTotals {
  unsigned long long n_unique;       // occupied slots
  unsigned long long n_hashed;       // Σ merged (clamped) counts
  unsigned long long n_lane_sum;     // Σ over lanes of Σ counts  (= n_kmers_ingested)
  unsigned long long any_saturated;
};

constexpr int HISTO_WG = 1024;  // few, large workgroups: the flush of a workgroup's bins is what contends
// KEYS = false: occupancy is read off the counts — a slot holds a key exactly when its counts are not
// all zero, unless a key was inserted with count 0 (shk_insert_counts can; the host then asks for
// KEYS = true) — so the scan reads 4 B per slot and lane instead of 8 + 4.
template <bool KEYS>
__global__ void __launch_bounds__(HISTO_WG) k_histo(TableRef tb, uint64_t slot0, uint64_t slot1,
                                              uint64_t histo_max, uint32_t n_cols,
                                              uint32_t lds_bins,
                                              unsigned long long *__restrict__ hist,
                                              HistoTotals *__restrict__ tot) {
  extern __shared__ uint32_t lh[];  // n_cols * lds_bins
  __shared__ unsigned long long wtot[HISTO_WG / 64][4];
  const uint32_t n_l = n_cols * lds_bins;
  for (uint32_t i = threadIdx.x; i < n_l; i += HISTO_WG) lh[i] = 0;
  __syncthreads();
  const uint64_t hlen = histo_max + 2;
  unsigned long long n_unique = 0, n_hashed = 0, n_lane = 0, sat = 0;
  // four consecutive slots per thread per step: keys as 2×16 B, each lane's counts as 16 B,
  // all loads issued before any is consumed (slot0/slot1 are multiples of PAGE_SLOTS)
  const uint64_t stride = (uint64_t)gridDim.x * HISTO_WG * 4;
  uint64_t s = slot0 + ((uint64_t)blockIdx.x * HISTO_WG + threadIdx.x) * 4;
  if (!KEYS && tb.n_lanes == 1 && n_cols <= 1) {
    // One chunk lane, counts only: a step is a single 16-B load, far too little in flight for a scan
    // — four steps' loads are issued together.
    auto one = [&](const uint4 &v4) {
      const uint32_t v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (v[q] == 0) continue;
        n_lane += v[q];
        n_unique++;
        n_hashed += v[q];
        sat |= (v[q] == 0xFFFFFFFFu);
        if (n_cols) {
          const uint64_t bin = v[q] <= histo_max ? v[q] : histo_max + 1;
          if (bin < lds_bins)
            atomicAdd(&lh[(uint32_t)bin], 1u);
          else
            atomicAdd(&hist[bin], 1ull);
        }
      }
    };
    for (; s + 3 * stride < slot1; s += 4 * stride) {
      const uint4 a0 = *reinterpret_cast<const uint4 *>(tb.vals + s);
      const uint4 a1 = *reinterpret_cast<const uint4 *>(tb.vals + s + stride);
      const uint4 a2 = *reinterpret_cast<const uint4 *>(tb.vals + s + 2 * stride);
      const uint4 a3 = *reinterpret_cast<const uint4 *>(tb.vals + s + 3 * stride);
      one(a0);
      one(a1);
      one(a2);
      one(a3);
    }
  }
  for (; s < slot1; s += stride) {
    uint32_t cum[4] = {0, 0, 0, 0};
    bool occ[4] = {true, true, true, true};
    if (KEYS) {
      const ulonglong2 ka = *reinterpret_cast<const ulonglong2 *>(tb.keys + s);
      const ulonglong2 kb = *reinterpret_cast<const ulonglong2 *>(tb.keys + s + 2);
      occ[0] = ka.x != EMPTY, occ[1] = ka.y != EMPTY, occ[2] = kb.x != EMPTY, occ[3] = kb.y != EMPTY;
      if (!(occ[0] | occ[1] | occ[2] | occ[3])) continue;
    }
    for (uint32_t l0 = 0; l0 < tb.n_lanes; l0 += 4) {  // four lanes' counts in flight together
      uint4 vv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (l0 + j < tb.n_lanes) vv[j] = *reinterpret_cast<const uint4 *>(tb.vals + (uint64_t)(l0 + j) * tb.cap + s);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t l = l0 + j;
        if (l >= tb.n_lanes) break;
        const uint32_t v[4] = {vv[j].x, vv[j].y, vv[j].z, vv[j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (!occ[q]) continue;
          n_lane += v[q];
          cum[q] = sat_add_u32(cum[q], v[q]);
          if (l < n_cols && cum[q] > 0) {
            uint64_t bin = cum[q] <= histo_max ? cum[q] : histo_max + 1;
            if (bin < lds_bins)
              atomicAdd(&lh[l * lds_bins + (uint32_t)bin], 1u);
            else
              atomicAdd(&hist[(uint64_t)l * hlen + bin], 1ull);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (KEYS ? !occ[q] : cum[q] == 0) continue;
      n_unique++;
      n_hashed += cum[q];
      sat |= (cum[q] == 0xFFFFFFFFu);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < n_l; i += HISTO_WG) {
    uint32_t v = lh[i];
    if (v) {
      uint32_t l = i / lds_bins, bin = i % lds_bins;
      if ((uint64_t)bin < hlen) atomicAdd(&hist[(uint64_t)l * hlen + bin], (unsigned long long)v);
    }
  }
  // totals: wave reduce, then one workgroup reduce, then ONE set of atomics per workgroup —
  // same-line atomics retire one after the other (≈2 ns each on MI355X), so their number, not
  // their payload, is what costs
  for (int off = 32; off > 0; off >>= 1) {
    n_unique += __shfl_down(n_unique, off, 64);
    n_hashed += __shfl_down(n_hashed, off, 64);
    n_lane += __shfl_down(n_lane, off, 64);
    sat |= __shfl_down(sat, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    unsigned long long *w = wtot[threadIdx.x >> 6];
    w[0] = n_unique;
    w[1] = n_hashed;
    w[2] = n_lane;
    w[3] = sat;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    unsigned long long v = 0;
    for (int w = 0; w < HISTO_WG / 64; ++w) v = threadIdx.x == 3 ? (v | wtot[w][3]) : v + wtot[w][threadIdx.x];
    if (v) {
      if (threadIdx.x == 0) atomicAdd(&tot->n_unique, v);
      if (threadIdx.x == 1) atomicAdd(&tot->n_hashed, v);
      if (threadIdx.x == 2) atomicAdd(&tot->n_lane_sum, v);
      if (threadIdx.x == 3) atomicOr(&tot->any_saturated, 1ull);
    }
  }
}

// The control block (launch outcome, totals, lane sums, histogram) straight into the caller's pinned, device-visible
// copy of it: the first `head8` words whatever they hold, the histogram words only where they are not zero (the host
// cleared its copy before the launch; nearly all of them are, and stores over the link are slow).  Takes the place of
// a copy queued behind k_histo — the barrier between a kernel and a blit costs 12-14 µs, a fortieth of a 1 M-read job.
__global__ void __launch_bounds__(WG) k_ctl_out(const unsigned long long *__restrict__ src, unsigned long long *__restrict__ dst,
                                                uint32_t n8, uint32_t head8) {
  const uint32_t i = blockIdx.x * WG + threadIdx.x;
  if (i >= n8) return;
  const unsigned long long v = src[i];
  if (i < head8 || v) dst[i] = v;
}

// ==========================================================================================
// K_LOOKUP / K_EXPORT: merged-table read API (counting.rs:205-241).
// ==========================================================================================
__device__ __forceinline__ uint64_t revcomp(uint64_t kmer, int k) {
  // complement = bitwise NOT of each 2-bit code; reverse 2-bit groups of the 64-bit word
  uint64_t x = ~kmer;
  x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
  x = __builtin_bswap64(x);
  return x >> (64 - 2 * k);
}

__device__ __forceinline__ uint32_t merged_count(const TableRef &tb, uint64_t key) {
  const Home hm = home_of(tb, key);
  if (!hm.owned) return 0u;
  int64_t s = find_slot(tb.keys, hm.base, hm.slot, key);
  if (s < 0) return 0;
  uint32_t cum = 0;
  for (uint32_t l = 0; l < tb.n_lanes; ++l) cum = sat_add_u32(cum, tb.vals[(uint64_t)l * tb.cap + s]);
  return cum;
}

__global__ void __launch_bounds__(WG) k_lookup(TableRef tb, const uint64_t *__restrict__ kmers,
                                               uint32_t *__restrict__ out, uint64_t n, int canonical,
                                               int k) {
  uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
  if (i >= n) return;
  uint64_t key = kmers[i];
  if (canonical) {
    uint64_t rc = revcomp(key, k);
    key = key < rc ? key : rc;
  }
  out[i] = merged_count(tb, key);
}

__global__ void __launch_bounds__(WG) k_export(TableRef tb, uint64_t slot0, uint64_t slot1,
                                               uint64_t *__restrict__ kmers,
                                               uint32_t *__restrict__ counts, uint64_t cap,
                                               unsigned long long *__restrict__ n_out) {
  for (uint64_t s = slot0 + (uint64_t)blockIdx.x * WG + threadIdx.x; s < slot1;
       s += (uint64_t)gridDim.x * WG) {
    uint64_t key = tb.keys[s];
    if (key == EMPTY) continue;
    uint32_t cum = 0;
    for (uint32_t l = 0; l < tb.n_lanes; ++l) cum = sat_add_u32(cum, tb.vals[(uint64_t)l * tb.cap + s]);
    unsigned long long i = atomicAdd(n_out, 1ull);
    if (i < cap) {
      kmers[i] = key;
      counts[i] = cum;
    }
  }
}

// ==========================================================================================
// K_OLIGO: the primer seed scan of sPCR, find_oligos_in_kmers (src/pcr/primers.rs:163-226): one
// bandwidth-bound pass over the merged table.  A k-mer with count ≥ min_count is reported as
// is when its first oligo_len bases equal one of the oligos, or reverse-complemented when its
// LAST oligo_len bases equal the reverse complement of one (primers.rs:212-223).  The two
// sorted oligo sets sit in LDS; membership = binary search.
// ==========================================================================================
__device__ __forceinline__ bool in_sorted(const uint64_t *a, uint32_t n, uint64_t x) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1;
    else hi = mid;
  }
  return lo < n && a[lo] == x;
}

__global__ void __launch_bounds__(WG) k_find_oligos(TableRef tb, uint64_t slot0, uint64_t slot1, int k,
                                                    int oligo_len, uint32_t min_count,
                                                    const uint64_t *__restrict__ fwd_set /* shifted, sorted */,
                                                    const uint64_t *__restrict__ rc_set /* sorted */,
                                                    uint32_t n_oligos, uint64_t *__restrict__ out_kmers,
                                                    uint32_t *__restrict__ out_counts, uint64_t cap,
                                                    unsigned long long *__restrict__ n_out) {
  extern __shared__ uint64_t sets[];  // fwd then rc
  for (uint32_t i = threadIdx.x; i < n_oligos; i += WG) {
    sets[i] = fwd_set[i];
    sets[n_oligos + i] = rc_set[i];
  }
  __syncthreads();
  const uint64_t mask = ((1ull << (2 * oligo_len)) - 1) << (2 * (k - oligo_len));
  const uint64_t rc_mask = (1ull << (2 * oligo_len)) - 1;
  for (uint64_t s = slot0 + (uint64_t)blockIdx.x * WG + threadIdx.x; s < slot1;
       s += (uint64_t)gridDim.x * WG) {
    const uint64_t key = tb.keys[s];
    if (key == EMPTY) continue;
    uint32_t cum = 0;
    for (uint32_t l = 0; l < tb.n_lanes; ++l) cum = sat_add_u32(cum, tb.vals[(uint64_t)l * tb.cap + s]);
    if (cum < min_count) continue;
    uint64_t hit = EMPTY;
    if (in_sorted(sets, n_oligos, key & mask)) hit = key;
    else if (in_sorted(sets + n_oligos, n_oligos, key & rc_mask)) hit = revcomp(key, k);
    if (hit != EMPTY) {
      unsigned long long i = atomicAdd(n_out, 1ull);
      if (i < cap) {
        out_kmers[i] = hit;
        out_counts[i] = cum;
      }
    }
  }
}

// ==========================================================================================
// K_FILTER: PrimerReadFilter::matches (src/pcr/read_filter.rs:43-49) for a batch of reads: a read
// matches when kmers_from_ascii accepts it (no byte outside ACGTN — otherwise the reference
// returns false for the whole read) and at least one of its canonical k-mers is in the set.
// One thread per read (reads are a few hundred bases; this is the step before sPCR's read
// threading, not the counting hot path); the set is a small open-addressing table of mixed keys.
// ==========================================================================================
__global__ void __launch_bounds__(WG) k_filter_reads(const uint8_t *__restrict__ bases,
                                                     const uint64_t *__restrict__ offsets, uint64_t n_seqs,
                                                     int k, const uint64_t *__restrict__ set_keys,
                                                     uint32_t set_mask, uint8_t *__restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
  if (i >= n_seqs) return;
  const uint64_t mask = (1ull << (2 * k)) - 1;
  uint64_t fwd = 0, rev = 0;
  int n_valid = 0;
  bool hit = false, ok = true;
  for (uint64_t p = offsets[i], e = offsets[i + 1]; p < e; ++p) {
    const uint32_t c = bases[p];
    if (c == 'N') {  // encoding.rs:346-352
      n_valid = 0;
      continue;
    }
    if (!byte_is_acgtn(c)) {  // encoding.rs:353-356 → Err → matches() is false
      ok = false;
      break;
    }
    const uint64_t b2 = ((c >> 1) ^ (c >> 2)) & 3u;
    fwd = ((fwd << 2) | b2) & mask;
    rev = (rev >> 2) | ((3 - b2) << (2 * (k - 1)));
    if (++n_valid >= k && !hit) {
      const uint64_t key = fwd < rev ? fwd : rev;
      for (uint32_t s = set_hash(key) & set_mask;; s = (s + 1) & set_mask) {
        const uint64_t cur = set_keys[s];
        if (cur == key) hit = true;
        if (cur == key || cur == EMPTY) break;
      }
    }
  }
  out[i] = ok && hit;
}

// ==========================================================================================
// K_KMERS: kmers_from_ascii (src/kmer/encoding.rs:332-371) for a batch of reads, k-mers returned in
// read order (what thread_reads consumes, pcr/threading.rs:97-101).  Same one-thread-per-read
// rolling form as K_FILTER; read i writes at koff[i] (host prefix of max(0, len − k + 1)).
// ==========================================================================================
__global__ void __launch_bounds__(WG) k_kmers_from_reads(const uint8_t *__restrict__ bases,
                                                         const uint64_t *__restrict__ offsets,
                                                         const uint64_t *__restrict__ koff, uint64_t n_seqs,
                                                         int k, uint64_t *__restrict__ out,
                                                         uint32_t *__restrict__ n_out,
                                                         uint8_t *__restrict__ bad_byte) {
  const uint64_t i = (uint64_t)blockIdx.x * WG + threadIdx.x;
  if (i >= n_seqs) return;
  const uint64_t mask = (1ull << (2 * k)) - 1;
  uint64_t fwd = 0, rev = 0;
  int n_valid = 0;
  uint32_t n = 0, bad = 0;
  uint64_t *dst = out + koff[i];
  for (uint64_t p = offsets[i], e = offsets[i + 1]; p < e; ++p) {
    const uint32_t c = bases[p];
    if (c == 'N') {  // encoding.rs:346-352
      n_valid = 0;
      continue;
    }
    if (!byte_is_acgtn(c)) {  // encoding.rs:353-356 → Err for the whole read
      bad = c;
      break;
    }
    const uint64_t b2 = ((c >> 1) ^ (c >> 2)) & 3u;
    fwd = ((fwd << 2) | b2) & mask;
    rev = (rev >> 2) | ((3 - b2) << (2 * (k - 1)));
    if (++n_valid >= k) dst[n++] = fwd < rev ? fwd : rev;
  }
  n_out[i] = bad ? 0u : n;
  bad_byte[i] = (uint8_t)bad;
}

// ==========================================================================================
// K_OWNER_COUNTS / K_COMPACT_OWNERS: the sender side of the multi-GPU merge.  Owner o of W owns the
// slots [o·spo, (o+1)·spo) (a contiguous page range).  Instead of shipping its range as it lies
// in the table — EMPTY slots included, ≥ half of it — a rank ships only the occupied (key, counts)
// entries: first the per-owner entry counts, then the entries, owner after owner.  The grid is
// W × blocks_per_owner; a block owns a fixed slice of one owner's range, counts it, reserves its
// share of the owner's segment with ONE atomic, and writes its entries in slot order.
// ==========================================================================================
__global__ void __launch_bounds__(WG) k_owner_counts(TableRef tb, uint64_t spo, uint32_t OWNER_BLOCKS,
                                                     unsigned long long *__restrict__ counts) {
  __shared__ uint32_t red[WG / 64];
  const uint32_t o = blockIdx.x / OWNER_BLOCKS, j = blockIdx.x % OWNER_BLOCKS;
  const uint64_t per = (spo + OWNER_BLOCKS - 1) / OWNER_BLOCKS;
  const uint64_t s0 = (uint64_t)o * spo + (uint64_t)j * per;
  const uint64_t s1 = s0 + per < (uint64_t)(o + 1) * spo ? s0 + per : (uint64_t)(o + 1) * spo;
  uint32_t n = 0;
  for (uint64_t s = s0 + threadIdx.x; s < s1; s += WG) n += tb.keys[s] != EMPTY;
  const uint32_t tot = wg_sum<WG>(n, red);
  if (threadIdx.x == 0 && tot) atomicAdd(&counts[o], (unsigned long long)tot);
}

__global__ void __launch_bounds__(WG) k_compact_owners(TableRef tb, uint64_t spo,
                                                       const unsigned long long *__restrict__ seg_offset,
                                                       unsigned long long *__restrict__ seg_cursor,
                                                       uint64_t *__restrict__ out_keys,
                                                       uint32_t *__restrict__ out_vals, uint64_t lane_stride,
                                                       uint32_t skip_owner, uint32_t OWNER_BLOCKS,
                                                       const unsigned long long *__restrict__ packed,
                                                       unsigned long long hdr_bytes, unsigned long long *__restrict__ max_count) {
  __shared__ uint32_t red[WG / 64];
  __shared__ uint32_t wbase[WG / 64];
  __shared__ unsigned long long blk_base;
  const uint32_t o = blockIdx.x / OWNER_BLOCKS, j = blockIdx.x % OWNER_BLOCKS;
  if (o == skip_owner) return;  // (a rank does not ship its own range to itself)
  const uint64_t per = (spo + OWNER_BLOCKS - 1) / OWNER_BLOCKS;
  const uint64_t s0 = (uint64_t)o * spo + (uint64_t)j * per;
  const uint64_t s1 = s0 + per < (uint64_t)(o + 1) * spo ? s0 + per : (uint64_t)(o + 1) * spo;
  uint32_t n = 0;
  for (uint64_t s = s0 + threadIdx.x; s < s1; s += WG) n += tb.keys[s] != EMPTY;
  const uint32_t tot = wg_sum<WG>(n, red);  // valid in thread 0
  if (threadIdx.x == 0) {
    const unsigned long long at0 = tot ? atomicAdd(&seg_cursor[o], (unsigned long long)tot) : 0ull;
    blk_base = seg_offset[o] + at0;
    if (max_count && tot) atomicMax(max_count, at0 + tot);  // (fixed-capacity pieces: the sender learns how many entries an owner really had)
  }
  __syncthreads();
  unsigned long long at = blk_base;
  const uint32_t wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
  for (uint64_t sb = s0; sb < s1; sb += WG) {  // second pass: the slice is L2-resident by now
    const uint64_t s = sb + threadIdx.x;
    const uint64_t key = s < s1 ? tb.keys[s] : EMPTY;
    const unsigned long long mm = __ballot(key != EMPTY);
    if (lane_id == 0) wbase[wave] = (uint32_t)__popcll(mm);
    __syncthreads();
    uint32_t off = 0, chunk = 0;
    for (uint32_t w = 0; w < WG / 64; ++w) {
      if (w < wave) off += wbase[w];
      chunk += wbase[w];
    }
    if (key != EMPTY) {
      const unsigned long long i = at + off + __popcll(mm & ((1ull << lane_id) - 1ull));
      if (packed) {
        // PACKED: owner o's entries are one self-contained piece — [keys 8·c][lane 0 counts 4·c]…[lane L-1] at
        // byte offset seg_offset[o]·(8 + 4L), c = packed[o] — so keys and every lane's counts cross the
        // links in ONE collective; i counts from seg_offset[o]
        // (fixed-capacity pieces: hdr_bytes in front of every piece, entries beyond the capacity are not written
        // — max_count tells the receivers, who then leave ALL pieces alone: shk_compact_owners_fixed)
        const unsigned long long so = seg_offset[o], c_o = packed[o], j = i - so;
        if (j < c_o) {
          char *const seg = reinterpret_cast<char *>(out_keys) + so * (8ull + 4ull * tb.n_lanes) + (o + 1ull) * hdr_bytes;
          __builtin_memcpy(seg + j * 8ull, &key, 8);  // (4-byte aligned when L and the offset are odd)
          for (uint32_t l = 0; l < tb.n_lanes; ++l)
            *reinterpret_cast<uint32_t *>(seg + c_o * 8ull + ((unsigned long long)l * c_o + j) * 4ull) = tb.vals[(uint64_t)l * tb.cap + s];
        }
      } else {
        out_keys[i] = key;
        for (uint32_t l = 0; l < tb.n_lanes; ++l) out_vals[(uint64_t)l * lane_stride + i] = tb.vals[(uint64_t)l * tb.cap + s];
      }
    }
    at += chunk;
    __syncthreads();
  }
}

// ==========================================================================================
// K_MERGE: KmerCounts::extend across devices (counting.rs:157-166): fold a peer's page range
// (same geometry) into this table, lane by lane, saturating.
// ==========================================================================================
__global__ void __launch_bounds__(WG) k_merge(TableRef tb, uint64_t n_slots, uint64_t lane_stride,
                                              const uint64_t *__restrict__ pkeys,
                                              const uint32_t *__restrict__ pvals_,
                                              DevStats *__restrict__ stats, SpillRef sp,
                                              uint64_t piece_cap, uint32_t skip_piece) {
  if (piece_cap) {
    // Every piece's header holds how many entries its sender's fullest owner range had.  Every rank gets one
    // piece from every sender (its own included), so every rank sees the same maximum here: beyond the capacity
    // NOBODY merges anything (the tables stay as they were; the host repeats the exchange with exact counts).
    const uint64_t n_pieces = n_slots / piece_cap;
    unsigned long long most = 0;
    for (uint64_t p = 0; p < n_pieces; ++p) {
      unsigned long long h;
      __builtin_memcpy(&h, reinterpret_cast<const uint32_t *>(pkeys) + p * (2ull + piece_cap * (2ull + tb.n_lanes)), 8);
      most = h > most ? h : most;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(&stats->scratch[2], most);
    if (most > piece_cap) return;
  }
  uint32_t n_new = 0;
  for (uint64_t i0 = (uint64_t)blockIdx.x * WG + threadIdx.x; i0 < n_slots;
       i0 += (uint64_t)gridDim.x * WG) {
    uint64_t i = i0, key;
    const uint32_t *pvals = pvals_;
    if (piece_cap) {
      // several self-contained pieces of piece_cap entries each ([header][k-mers][lane 0]…[lane L-1]) in one launch
      const uint64_t piece = i0 / piece_cap;
      if (piece == skip_piece) continue;
      i = i0 - piece * piece_cap;
      const uint32_t *pb = reinterpret_cast<const uint32_t *>(pkeys) + piece * (2ull + piece_cap * (2ull + tb.n_lanes)) + 2;
      __builtin_memcpy(&key, pb + 2 * i, 8);
      pvals = pb + 2 * piece_cap;
    } else {
      key = pkeys[i];
    }
    if (key == EMPTY) continue;
    const Home hm = home_of(tb, key);
    if (!hm.owned) continue;
    bool inserted = false;
    int64_t s = find_or_insert(tb.keys, hm.base, hm.slot, key, inserted);
    if (s < 0) {
      // page full: one spill record per non-zero lane
      for (uint32_t l = 0; l < tb.n_lanes; ++l) {
        uint32_t v = pvals[(uint64_t)l * lane_stride + i];
        if (!v) continue;
        unsigned long long j = atomicAdd(&stats->spill_count, 1ull);
        if (j < sp.cap) {
          sp.keys[j] = key;
          sp.lanes[j] = l;
          sp.counts[j] = v;
        }
      }
      continue;
    }
    n_new += inserted;
    for (uint32_t l = 0; l < tb.n_lanes; ++l) {
      uint32_t v = pvals[(uint64_t)l * lane_stride + i];
      if (v) sat_add_atomic(tb.vals + (uint64_t)l * tb.cap + (uint64_t)s, v);
    }
  }
  for (int off = 32; off > 0; off >>= 1) n_new += __shfl_down(n_new, off, 64);
  if ((threadIdx.x & 63) == 0 && n_new) atomicAdd(&stats->n_distinct, (unsigned long long)n_new);
}

// shk_finalize_begin: the words a reduction over the ranks sums besides the totals and the histogram.
__global__ void k_fin_extras(unsigned long long *__restrict__ extra, unsigned long long n_reads,
                             unsigned long long n_bases_read, unsigned long long user_word,
                             const DevStats *__restrict__ stats, uint32_t was_unsettled,
                             const unsigned long long *__restrict__ lane_bases, unsigned long long *__restrict__ lane_sum,
                             uint32_t n_lanes) {
  // the live per-lane base counters go into the summed block afresh on every finalize (they are never reduced in place)
  for (uint32_t l = threadIdx.x; l < n_lanes; l += blockDim.x) lane_sum[l] = lane_bases[l];
  if (threadIdx.x == 0) {
    extra[0] = n_reads;
    extra[1] = n_bases_read;
    extra[2] = (was_unsettled && (stats->spill_count || stats->bad != ~0ull)) ? 1ull : 0ull;
    extra[3] = user_word;
  }
}

__global__ void k_piece_headers(uint32_t *__restrict__ buf, uint32_t n_pieces, unsigned long long piece_ints,
                                const DevStats *__restrict__ stats) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_pieces) {
    // (scratch[1]: entries of the fullest owner range.  A launch before the compaction that spilled records or
    // met an invalid byte left a table that is not what the pieces should say: nobody may merge them.)
    const unsigned long long m = (stats->spill_count || stats->bad != ~0ull) ? ~0ull : stats->scratch[1];
    __builtin_memcpy(buf + p * piece_ints, &m, 8);
  }
}

// ==========================================================================================
// K_SYNTH: synthetic reads: implicit uniform genome, fixed-length reads,
// random strand, optional substitutions and N.  Must match sharkmer_amd/synth.py bit for bit.
// ==========================================================================================
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct SynthSpec {
  uint64_t seed_genome, seed_reads, genome_len;
  uint32_t read_len, sub_per_64k, n_per_64k, pad;
};

__global__ void __launch_bounds__(WG) k_synth(SynthSpec sp, uint64_t first_read, uint64_t n_reads,
                                              uint8_t *__restrict__ bases,
                                              uint64_t *__restrict__ offsets) {
  // one thread per 8 bases: 8-byte stores, coalesced
  const uint64_t L = sp.read_len;
  const uint64_t total = n_reads * L;
  uint64_t g = ((uint64_t)blockIdx.x * WG + threadIdx.x) * 8;
  if (g <= n_reads && g + 8 > 0) {
    for (int r = 0; r < 8; ++r)
      if (g + r <= n_reads) offsets[g + r] = (g + r) * L;
  }
  if (g >= total) return;
  uint8_t out[8];
  const char LUT[4] = {'A', 'C', 'G', 'T'};
  for (int r = 0; r < 8; ++r) {
    uint64_t p = g + r;
    if (p >= total) { out[r] = 0; continue; }
    uint64_t i = first_read + p / L, j = p % L;
    uint64_t h1 = splitmix64(sp.seed_reads + 2 * i);
    uint64_t h2 = splitmix64(sp.seed_reads + 2 * i + 1);
    uint64_t start = h1 % (sp.genome_len - L + 1);
    bool rc = (h2 >> 63) != 0;
    uint64_t gp = rc ? start + (L - 1 - j) : start + j;
    uint32_t b = (uint32_t)(splitmix64(sp.seed_genome + gp) & 3);
    if (rc) b = 3 - b;
    uint8_t ch = (uint8_t)LUT[b];
    if (sp.sub_per_64k | sp.n_per_64k) {
      uint64_t e = splitmix64((sp.seed_reads ^ 0xE44044ull) + i * L + j);
      uint32_t u = (uint32_t)(e & 0xFFFF);
      if (u < sp.n_per_64k) {
        ch = 'N';
      } else if (u < sp.n_per_64k + sp.sub_per_64k) {
        uint32_t d = 1 + (uint32_t)((e >> 16) % 3);
        ch = (uint8_t)LUT[(b + d) & 3];
      }
    }
    out[r] = ch;
  }
  if (g + 8 <= total) {
    __builtin_memcpy(bases + g, out, 8);
  } else {
    for (int r = 0; r < 8 && g + r < total; ++r) bases[g + r] = out[r];
  }
}

}  // namespace shk

// ##########################################################################################
// Paged path: the table is a sequence of PAGE_SLOTS-slot pages, each an independent
// open-addressing table that fits in LDS (64 KiB keys + 32 KiB counts).  One counting pass =
//   K_SCATTER  k_part_scatter_sorted  validate + extract k-mers tile by tile, counting-sort each
//                                     tile by page in LDS, append every page's run to that page's
//                                     region of part_buf (reserved with one returning atomic per
//                                     (tile, page)) as aligned 16-B record pairs
//   K_PAGES    k_pages                one workgroup per page: page → LDS, stream the page's
//                                     region through LDS atomics, page → HBM
// HBM traffic per k-mer occurrence: 8 B written + 8 B read, all of it streaming; the table
// is read and written once per pass.  counting.rs:82-85 semantics, exact incl. saturation.
// ##########################################################################################
namespace shk {

constexpr int MAX_PARTS = 4096;   // partitions one LDS sort fans out to; more pages ⇒ two levels
constexpr int PG_WG = 512;        // k_pages workgroup: 8 waves; two workgroups (80 KiB LDS each) per CU
constexpr uint32_t PAGE_FILL_CAP = PAGE_SLOTS - PAGE_SLOTS / 8;  // new keys spill beyond this

// ------------------------------------------------------------------------------------------
// k_part_scatter_sorted: the same job as k_part_scatter, but every tile is counting-sorted by
// partition in LDS first, so that a partition's k-mers leave the CU as runs of consecutive
// 8-B records (coalesced 64..512-B requests) instead of one 8-B request per k-mer.
// Measured on MI355X (tools/wbench.hip): 8-B scattered stores 0.6 TB/s, 64-B chunks 3.2 TB/s,
// ≥128-B chunks 4.9 TB/s.
//   walk    : rolling extraction; per k-mer a returning LDS add gives its rank inside its
//             partition; (partition, rank) stays in a register
//   scan    : exclusive scan of the tile's partition counts
//   place   : sorted[tstart[p] + rank] = end position (u16), aliasing the dead code bytes
//   write   : entry i → k-mer re-read from the 2-bit packed copy of the tile → its place in
//             the partition's output run
// ------------------------------------------------------------------------------------------
#ifndef SORTED_WAVES_PER_SIMD
#define SORTED_WAVES_PER_SIMD 4
#endif
// LDS bytes of the sorted-entry region: TILE_T entries + one possible pad per page (u16), and it
// doubles as the staging area (code bytes + group masks) before the sort
__host__ __device__ inline uint32_t sort_region_bytes(uint32_t P) {
  uint32_t a = 2u * ((uint32_t)TILE_T + P), b = (uint32_t)STAGE_BYTES;
  return ((a > b ? a : b) + 15u) & ~15u;
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / broadcasts (VALU only;
// the __shfl_up formulation costs six ds_bpermute round trips).
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
  // within each row of 16 lanes: shifts by 1, 2, 4, 8 (bound_ctrl: lanes shifted in read 0)
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
  // across rows: lane 15 of a row into the next row (rows 1 and 3), then lane 31 into rows 2 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31
  return v;
}

// the k-base window that starts at base index q of an MSB-first packed stream
__device__ __forceinline__ uint64_t window_at(const uint32_t *stream, int q, int k) {
  const int s = 2 * q;
  const int wi = s >> 5;
  const uint32_t off = (uint32_t)s & 31u;
  // 64 stream bits from bit s on, as two funnel shifts (v_alignbit_b32 shifts by its count mod 32,
  // hence the selects for off = 0); the third word is a pad word at worst, never out of bounds
  const uint32_t w0 = stream[wi], w1 = stream[wi + 1], w2 = stream[wi + 2];
  const uint32_t hi = off ? __builtin_amdgcn_alignbit(w0, w1, 32u - off) : w0;
  const uint32_t lo = off ? __builtin_amdgcn_alignbit(w1, w2, 32u - off) : w1;
  return (((uint64_t)hi << 32) | lo) >> (64 - 2 * k);
}
// Canonical k-mer whose LAST base sits at LDS position j (halo included).  The walk has already
// decided the strand: rc = 1 ⇔ the reverse complement is the smaller one, and that is a plain
// window of the mirrored complement stream (base j of the tile is its base TILE_LDS-1-j).
__device__ __forceinline__ uint64_t kmer_at(const uint32_t *packed, const uint32_t *rcpacked, int j,
                                            uint32_t rc, int k) {
  return window_at(rc ? rcpacked : packed, rc ? TILE_LDS - 1 - j : j - k + 1, k);  // one window read, no branch
}

// Layout of the 4-byte-record buffers: BLOCK-INTERLEAVED.  Record j of region r (a page, or a
// super-page at level 1) lives at
//     ((j >> RB_LOG) · n_regions + r) << RB_LOG  |  (j & (2^RB_LOG - 1))
// i.e. the regions' 4-KiB blocks alternate.  All regions fill at nearly the same rate (the hash
// spreads k-mers evenly), so the write fronts of all ≤ 4096 regions stay within a few MiB of one
// another instead of one per region-sized stride, and the scatter's stores — every store
// instruction touches a dozen regions — keep hitting the same few address translations.
constexpr uint32_t RB_LOG = 10;  // records per block: 4 KiB
__device__ __forceinline__ uint32_t rec_slot(uint32_t region, uint32_t n_regions, uint32_t j) {
  return (((j >> RB_LOG) * n_regions + region) << RB_LOG) | (j & ((1u << RB_LOG) - 1u));
}
__device__ __forceinline__ uint64_t rec_slot64(uint64_t region, uint64_t n_regions, uint32_t j) {
  return ((((uint64_t)(j >> RB_LOG)) * n_regions + region) << RB_LOG) | (j & ((1u << RB_LOG) - 1u));
}

// REC32: the 4-byte-record variant (one level, 2k - log_parts ≤ 32): a record is the low
// 2k - log_parts bits of the MIXED key (the page is implied by the region, mix_key is a bijection),
// runs are packed without padding, one 4-B store per record.
template <int NT, bool REC32>
__global__ void __launch_bounds__(NT, SORTED_WAVES_PER_SIMD) k_part_scatter_sorted(
    BatchRef b, uint32_t log_parts, uint32_t lane_filter, unsigned int *__restrict__ cursor,
    uint32_t cap_p, void *__restrict__ part_buf_, DevStats *__restrict__ stats,
    unsigned long long *__restrict__ lane_bases, SpillRef sp, unsigned long long *__restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
  __shared__ uint32_t wsum[NT / 64];
  __shared__ uint32_t red[NT / 64];
  constexpr int SPAN = TILE_T / NT;
  const uint32_t P = 1u << log_parts;
  const uint32_t sort_bytes = sort_region_bytes(P);
  uint8_t *codes = reinterpret_cast<uint8_t *>(sh);                 // STAGE_BYTES ≤ sort_bytes
  uint16_t *sorted = reinterpret_cast<uint16_t *>(sh);              // TILE_T + P entries; aliases codes
  uint32_t *packed = sh + sort_bytes / 4;                           // PACK_WORDS
  uint32_t *rcpacked = packed + PACK_WORDS;                         // PACK_WORDS: mirrored complement
  uint32_t *cnt = rcpacked + PACK_WORDS;                            // P
  uint32_t *tstart = cnt + P;                                       // P
  uint32_t *gbase = tstart + P;                                     // P: this tile's reservation per page
#ifdef SHK_PHASE_TIMING
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
#define STAMP(i)                                        \
  do {                                                  \
    unsigned long long tn = __builtin_readcyclecounter(); \
    ph[i] += tn - tprev;                                \
    tprev = tn;                                         \
  } while (0)
#else
#define STAMP(i)
#endif
  // This kernel is also the validation pass (encoding.rs:353-356) and the non-N base count
  // (chunk.rs:28): both ride on the staging loads.  An invalid byte found by ANY workgroup
  // keeps k_pages from running (it tests stats->bad), so the table stays untouched.
  uint32_t n_non_n = 0;
  const int k = b.k;
  const uint64_t mask = (1ull << (2 * k)) - 1;
  const uint32_t mask_lo = (uint32_t)mask, mask_hi = (uint32_t)(mask >> 32);
  const uint32_t per = P / NT ? P / NT : 1;  // partitions per thread in the scan (P ≥ WG or P < WG)
  uint64_t *part_buf = reinterpret_cast<uint64_t *>(part_buf_);
  uint32_t *part_buf32 = reinterpret_cast<uint32_t *>(part_buf_);
  const uint32_t pad1 = REC32 ? 0u : 1u;  // runs are padded to even length unless REC32

  uint64_t t = blockIdx.x, t0, t1;
  uint32_t lane;
  bool have = next_tile(b, t, true, lane_filter, t0, t1, lane);
  StageRegs<NT> pre;
  if (have) stage_prefetch<NT>(b, t0, pre);
  while (have) {
    __syncthreads();  // previous tile's write phase is done with sorted/cnt/tstart
#ifdef SHK_PHASE_TIMING
    tprev = __builtin_readcyclecounter();
#endif
    for (uint32_t i = threadIdx.x; i < P; i += NT) cnt[i] = 0;
    n_non_n += stage_tile<true, NT, true>(b, t0, t1, codes, stats, pre, packed, rcpacked);
    uint64_t tn = t + gridDim.x, n0, n1;
    uint32_t nl;
    const bool hn = next_tile(b, tn, true, lane_filter, n0, n1, nl);
    if (hn) stage_prefetch<NT>(b, n0, pre);  // in flight during the rest of this tile
    __syncthreads();
    STAMP(0);
    STAMP(1);
    // ---- walk: (partition, rank) per end position, kept in registers -------------------
    uint32_t pr[SPAN];
    {
      const int e0 = threadIdx.x * SPAN;
      const int n_end = (int)(t1 - t0);
      const int jemit = HALO + e0;
      const int jend = HALO + (e0 + SPAN < n_end ? e0 + SPAN : n_end);
      Roll x{0, 0, 0, 0};
      if (e0 < n_end) {
        // the frames as they stand just before the first end position: the k bases up to
        // jemit-1, read as one window of either packed stream (N / foreign bases pack as A; the
        // "k-mer ends here" bits keep windows that contain them from being emitted)
        const uint64_t f0 = window_at(packed, jemit - k, k);
        const uint64_t r0 = window_at(rcpacked, TILE_LDS - jemit, k) << (64 - 2 * k);
        x.f_lo = (uint32_t)f0;
        x.f_hi = (uint32_t)(f0 >> 32);
        x.r_lo = (uint32_t)r0;
        x.r_hi = (uint32_t)(r0 >> 32);
      }
#pragma unroll
      for (int q = 0; q < SPAN / 8; ++q) {
        uint64_t w = 0;
        if (jemit + q * 8 < jend) w = *reinterpret_cast<const uint64_t *>(codes + jemit + q * 8);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          uint32_t c = (uint32_t)(w >> (8 * r)) & 0xFF;
          roll_step(x, c & 3u, mask_lo, mask_hi);
          uint32_t v = 0xFFFFFFFFu;
          if (c & 4u) {  // (never set at or beyond the tile's end: stage_tile)
            const uint64_t fwd = ((uint64_t)x.f_hi << 32) | x.f_lo;
            const uint64_t rev = (((uint64_t)x.r_hi << 32) | x.r_lo) >> (64 - 2 * k);
            const bool rc = rev < fwd;
            const uint32_t pc = (uint32_t)page_of(hash64(rc ? rev : fwd, 2 * k), log_parts);
            v = (pc << 16) | (rc ? 0x8000u : 0u) | atomicAdd(&cnt[pc], 1u);  // rank < 2^14
          }
          pr[q * 8 + r] = v;
        }
      }
    }
    __syncthreads();
    STAMP(2);
    // ---- exclusive scan of the even-padded counts → tstart (every run starts on a pair) ------
    {
      uint32_t lo = threadIdx.x * per, s = 0;
      if (lo < P)
        for (uint32_t i = 0; i < per; ++i) s += (cnt[lo + i] + pad1) & ~pad1;
      const uint32_t inc = wave_scan_incl(s);
      if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
      __syncthreads();
      const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
      const uint32_t part = (ln < wv && ln < (uint32_t)(NT / 64)) ? wsum[ln] : 0u;
      const uint32_t woff = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(part), 63);
      uint32_t run = woff + inc - s;
      if (lo < P)
        for (uint32_t i = 0; i < per; ++i) {
          tstart[lo + i] = run;
          run += (cnt[lo + i] + pad1) & ~pad1;
        }
    }
    // reserve this tile's (even-padded) run in every page's output region: one returning
    // device-scope add per non-empty (tile, page); consecutive lanes hit consecutive cursors.
    // The results are first needed by the write phase, a barrier and the place phase later, so
    // they are parked in registers and only stored to LDS (where the wait for them sits) then.
    uint32_t gres[MAX_PARTS / NT];  // stay in registers until the place phase is over
#pragma unroll
    for (int r = 0; r < MAX_PARTS / NT; ++r) {
      const uint32_t i = threadIdx.x + r * NT;
      gres[r] = 0;
      if (i < P) {
        const uint32_t c2 = (cnt[i] + pad1) & ~pad1;
        if (c2) gres[r] = atomicAdd(&cursor[i], c2);
      }
    }
    __syncthreads();  // codes are dead from here: `sorted` may overwrite them
    STAMP(3);
    // ---- place (the odd runs' padding slot gets the sentinel 0xFFFF) ---------------------------
#pragma unroll
    for (int i = 0; i < SPAN; ++i) {
      uint32_t v = pr[i];
      if (v != 0xFFFFFFFFu)  // entry = end position | strand << 14
        sorted[tstart[v >> 16] + (v & 0x7FFFu)] = (uint16_t)((threadIdx.x * SPAN + i) | ((v & 0x8000u) >> 1));
    }
    if (!REC32)
      for (uint32_t i = threadIdx.x; i < P; i += NT)
        if (cnt[i] & 1u) sorted[tstart[i] + cnt[i]] = 0xFFFFu;
#pragma unroll
    for (int r = 0; r < MAX_PARTS / NT; ++r)  // the reservations have had the place phase to come back
      if (threadIdx.x + r * NT < P)  // gbase - tstart: what the write phase adds to an entry's index
        gbase[threadIdx.x + r * NT] = gres[r] - tstart[threadIdx.x + r * NT];
    __syncthreads();
    STAMP(4);
    // ---- write: one PAIR of records per lane per store (16 B, aligned: runs start on even
    // record indices in LDS and in HBM); pairs never straddle partitions ----------------------
    if (REC32) {
      const uint32_t n_rec = tstart[P - 1] + cnt[P - 1];
      const uint32_t rbits = 2u * (uint32_t)k - log_parts;  // ≤ 32
      // two consecutive entries per lane: they nearly always belong to the same page, and then
      // leave as one 8-B store (4-B aligned is enough for global memory)
      const uint32_t *sorted2r = reinterpret_cast<const uint32_t *>(sorted);
      const uint32_t rmask = (uint32_t)(0xFFFFFFFFull >> (32 - rbits));
      auto spill_km = [&](uint64_t km) {  // the page's region is full (skewed input)
        const unsigned long long j = atomicAdd(&stats->spill_count, 1ull);
        if (j < sp.cap) {
          sp.keys[j] = km;
          sp.lanes[j] = lane;
          sp.counts[j] = 1u;
        }
      };
      for (uint32_t i = threadIdx.x; 2 * i < n_rec; i += NT) {
        const uint32_t ee = sorted2r[i];
        const uint32_t ea = ee & 0xFFFFu, eb = ee >> 16;
        const bool two = 2 * i + 1 < n_rec;
        const uint64_t km0 = kmer_at(packed, rcpacked, HALO + (int)(ea & 0x3FFFu), ea >> 14, k);
        const uint64_t km1 = two ? kmer_at(packed, rcpacked, HALO + (int)(eb & 0x3FFFu), (eb >> 14) & 1u, k) : km0;
        const uint64_t y0 = mix_key(km0, 2 * k), y1 = mix_key(km1, 2 * k);
        const uint32_t pc0 = (uint32_t)(y0 >> rbits), pc1 = (uint32_t)(y1 >> rbits);  // = page_of(hash64(km))
        const uint32_t r0 = (uint32_t)y0 & rmask, r1 = (uint32_t)y1 & rmask;
        const uint32_t at0 = gbase[pc0] + 2 * i;  // record index inside the page's region
        // (a launch covers ≤ 2^28 k-mers: byte offsets into part_buf fit 32 bits)
        char *const base = reinterpret_cast<char *>(part_buf32);
        if (two && pc1 == pc0 && at0 + 2 <= cap_p && (at0 & ((1u << RB_LOG) - 1u)) != (1u << RB_LOG) - 1u) {
          uint2 rec2 = make_uint2(r0, r1);  // (both records in one block)
          __builtin_memcpy(base + rec_slot(pc0, P, at0) * 4u, &rec2, 8);
        } else {
          if (at0 < cap_p) *reinterpret_cast<uint32_t *>(base + rec_slot(pc0, P, at0) * 4u) = r0;
          else spill_km(km0);
          if (two) {
            const uint32_t at1 = gbase[pc1] + 2 * i + 1;
            if (at1 < cap_p) *reinterpret_cast<uint32_t *>(base + rec_slot(pc1, P, at1) * 4u) = r1;
            else spill_km(km1);
          }
        }
      }
    }
    const uint32_t n_pairs = REC32 ? 0u : (tstart[P - 1] + ((cnt[P - 1] + 1u) & ~1u)) >> 1;
    const uint32_t *sorted2 = reinterpret_cast<const uint32_t *>(sorted);
    for (uint32_t i = threadIdx.x; i < n_pairs; i += NT) {
      const uint32_t ee = sorted2[i];
      const uint32_t e0 = ee & 0xFFFFu, e1 = ee >> 16;
      const uint64_t km0 = kmer_at(packed, rcpacked, HALO + (int)(e0 & 0x3FFFu), e0 >> 14, k);
      const uint64_t km1 =
          e1 == 0xFFFFu ? EMPTY : kmer_at(packed, rcpacked, HALO + (int)(e1 & 0x3FFFu), e1 >> 14, k);
      const uint32_t pc = (uint32_t)page_of(hash64(km0, 2 * k), log_parts);
      const uint32_t at = gbase[pc] + 2 * i;  // record index inside page pc's region
      if (at + 2 <= cap_p) {
        ulonglong2 rec;
        rec.x = km0;
        rec.y = km1;
        const uint32_t byte_off = (pc * cap_p + at) * 8u;  // < 2^32: ≤ 2^28 k-mers per launch
        *reinterpret_cast<ulonglong2 *>(reinterpret_cast<char *>(part_buf) + byte_off) = rec;
      } else {  // the page's region is full (skewed input): these records take the spill path
        const unsigned long long j = atomicAdd(&stats->spill_count, km1 == EMPTY ? 1ull : 2ull);
        if (j < sp.cap) {
          sp.keys[j] = km0;
          sp.lanes[j] = lane;
          sp.counts[j] = 1u;
        }
        if (km1 != EMPTY && j + 1 < sp.cap) {
          sp.keys[j + 1] = km1;
          sp.lanes[j + 1] = lane;
          sp.counts[j + 1] = 1u;
        }
      }
    }
    __syncthreads();
    STAMP(5);
    t = tn;
    t0 = n0;
    t1 = n1;
    lane = nl;
    have = hn;
  }
  __syncthreads();
  {
    uint32_t tot = wg_sum<NT>(n_non_n, red);
    if (threadIdx.x == 0 && tot) atomicAdd(&lane_bases[lane_filter], (unsigned long long)tot);
  }
#ifdef SHK_PHASE_TIMING
  if (dbg && threadIdx.x == 0)
    for (int i = 0; i < 8; ++i) dbg[(uint64_t)blockIdx.x * 8 + i] = ph[i];
#endif
}

// ------------------------------------------------------------------------------------------
// k_scatter32: the 4-byte-record scatter with the records kept in LDS.  Same job and same output
// as k_part_scatter_sorted<.., true>, but a (macro) tile is processed as TILE_T/TT sub-tiles of
// TT end positions, small enough that the RECORD of every k-mer (32 bits) stays in LDS from the
// walk to the write-out:
//   walk   : rolling extraction → mix_key → page (top bits) and record (low bits); the record goes
//            to recs[], (page, rank) stays in a register
//   scan, reserve : as before
//   place  : sorted[tstart[page] + rank] = page << 14 | index into recs    (32-bit entries)
//   write  : entry → record from recs[], page from the entry: no k-mer is rebuilt, nothing is
//            hashed a second time; two entries per lane, one 8-B store when they share a page
// LDS: 4·TT (entries; aliases the staged code bytes) + 4·TT (records) + one packed stream
// (walk warm-up) + 12·P.
// ------------------------------------------------------------------------------------------
// place phase of k_scatter32.  The __restrict__ parameters are the point: `sorted` and `tstart`
// are carved out of the same LDS block, and without the promise that they do not overlap every
// tstart read has to wait behind the previous entry's write (one LDS round trip per entry).
template <int NT, int SPAN>
__device__ __forceinline__ void place_entries(uint32_t *__restrict__ sorted, const uint32_t *__restrict__ tstart,
                                              const uint32_t (&pr)[SPAN], uint32_t P) {
  const char *const ts_b = reinterpret_cast<const char *>(tstart);
#pragma unroll
  for (int i = 0; i < SPAN; ++i) {
    const uint32_t v = pr[i];  // partition · 2^18 | rank; partitions ≥ P are the walk's spare counters (no record)
    // (v >> 16 = partition · 4: tstart's byte offset; the entry keeps the partition field and gets the record's byte offset)
    if (v < (P << 18))
      sorted[*reinterpret_cast<const uint32_t *>(ts_b + (v >> 16)) + (v & 0x3FFFFu)] = (v & 0xFFFC0000u) | ((i * NT + threadIdx.x) << 2);
  }
}

// LOGP: the fan-out as a compile-time constant (0 = log_parts_ at run time).  1024 partitions — one per
// thread — is what every table from 8 M slots up gets (one level up to 4096 pages aside), so the
// scan, reserve and place phases are specialised for it.
// ALL: ALL-LANES mode (lane_filter = ~0) as a compile-time constant: with one lane and LOGP the number
// of regions is an immediate and rec_slot a shift and an or.
// OWN (with ALL): the OWNER layout of key-space-partitioned ingest.  The fan-out 2^log_parts covers the
// virtual global table: the top log_w bits of a partition index are the record's OWNER, the rest its
// super-page inside the owner's share.  Regions and cursors are ordered [owner][lane][super-page], every
// owner's regions block-interleave among themselves only, so that what one owner gets — its seg_recs
// records' worth of regions and their cursor words — is ONE contiguous piece of each array (what crosses
// the link in a multi-GPU run).  keep = an owner id: records of every other owner are dropped in the walk
// and there is a single segment (seg_recs = 0); keep = ~0: all owners' records are kept.
constexpr uint32_t SC32_MAX_LANES = 128;  // chunk lanes one all-lanes pass of k_scatter32 takes (its per-lane base counts: 1 KiB of static LDS)
struct OwnerCfg {
  uint32_t log_w;     // owner bits (≤ log_parts)
  uint32_t keep;      // owner id to keep, or ~0 = all
  uint32_t seg_recs;  // records per owner segment (0: one segment only)
  uint32_t pad;
};
template <int NT, int TT, bool WIDE, int LOGP = 0, bool ALL = false, bool OWN = false>
__global__ void __launch_bounds__(NT, 4) k_scatter32(
    BatchRef b, uint32_t log_parts_, uint32_t lane_filter, unsigned int *__restrict__ cursor,
    uint32_t cap_p, uint32_t *__restrict__ part_buf32, DevStats *__restrict__ stats,
    unsigned long long *__restrict__ lane_bases, SpillRef sp, unsigned long long *__restrict__ dbg,
    uint32_t n_region_lanes, OwnerCfg own) {
  static_assert(!OWN || (ALL && !WIDE), "the owner layout is an all-lanes, per-launch layout");
  extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
  __shared__ uint32_t wsum[NT / 64];
  __shared__ uint32_t red[NT / 64];
  __shared__ unsigned long long lane_nn[SC32_MAX_LANES];  // ALL-LANES mode: non-N bases per chunk lane
  // lane_filter = ~0: ALL-LANES mode — one pass over the tiles of every chunk lane; region and cursor
  // index = lane · P + page (n_region_lanes · P regions, block-interleaved together)
  constexpr bool all_lanes = ALL;  // (the host picks the variant from lane_filter == ~0)
  if (all_lanes && threadIdx.x < SC32_MAX_LANES) lane_nn[threadIdx.x] = 0;
  static_assert(TILE_T % TT == 0 && TT % (8 * NT) == 0 && 4 * TT >= TT + HALO + 4 * ((TT + HALO) / 16), "tile shape");
  constexpr int SPAN = TT / NT;
  constexpr int GROUPS = (TT + HALO) / 16;
  const uint32_t log_parts = LOGP ? (uint32_t)LOGP : log_parts_;
  const uint32_t P = 1u << log_parts;
  uint2 *pg = reinterpret_cast<uint2 *>(sh);         // the stage's (packed word, N | read-start masks) per group, for the neighbours; dead once the walk starts
  static_assert(SPAN == 16 && GROUPS == NT + 2 && 2 * GROUPS <= TT, "a thread walks exactly the group it staged; the staged pairs fit the entries they alias");
  uint32_t *sorted = sh;                             // TT + P entries (every partition's run starts at an EVEN entry: up to P holes); aliases pg
  uint32_t *recs = sh + TT + P;                      // TT records: thread t's i-th end position at i·NT + t
  uint32_t *cnt = recs + TT;                         // P
  uint32_t *tstart = cnt + P;                        // P
  uint32_t *gbase = cnt;                             // P: ((this tile's reservation) - tstart) · 4 — takes cnt's place once the runs are placed
  __shared__ uint32_t n_ent_sh;                      // entries of this tile, holes included
  // (cnt[P..P+7] = tstart[0..7] double as the spare counters of the walk: tstart is written after it)
#ifdef SHK_PHASE_TIMING
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
#endif
  uint32_t n_non_n = 0;
  const int k = b.k;
  const uint64_t mask = (1ull << (2 * k)) - 1;
  const uint32_t mask_lo = (uint32_t)mask, mask_hi = (uint32_t)(mask >> 32);
  const uint32_t per = P / NT ? P / NT : 1;
  const uint32_t rbits = 2u * (uint32_t)k - log_parts;  // ≤ 32: bits of a record
  const uint32_t rmask = (uint32_t)(0xFFFFFFFFull >> (32 - rbits));
  const uint32_t n_regions = (all_lanes ? n_region_lanes : 1u) * P;
  // owner layout
  const uint32_t log_p1w = OWN ? log_parts - own.log_w : 0u;  // super-page bits inside an owner's share
  const uint32_t p1w_mask = (1u << log_p1w) - 1u;
  const uint32_t n_grp = n_region_lanes << log_p1w;           // regions of one owner segment
  const uint32_t keep_mask = own.keep == 0xFFFFFFFFu ? 0u : 0xFFFFFFFFu;
  // cursor word / region-in-segment of partition i for this tile's lane; owner segment of partition i
  auto own_grp = [&](uint32_t i, uint32_t ln) -> uint32_t { return (ln << log_p1w) | (i & p1w_mask); };
  auto own_seg = [&](uint32_t i) -> uint32_t { return own.seg_recs ? i >> log_p1w : 0u; };

  // iteration over sub-tiles: (macro tile t = [t0,t1) of chunk lane `lane`, sub-tile index sub)
  uint64_t t = blockIdx.x, t0, t1;
  uint32_t lane;
  bool have = next_tile(b, t, !all_lanes, lane_filter, t0, t1, lane);
  uint32_t sub = 0;
  StageRegs32<NT, TT> pre;
  if (have) stage32_prefetch<NT, TT>(b, t0, pre);
  // ALL-LANES mode walks a tile LIST (k_build_tiles): a tile's descriptor is two dependent loads away (how many tiles
  // there are, then the descriptor), and the prefetch of its bases a third — asked for when the tile before it is
  // staged, that chain stood in the stage phase of every tile (10 lanes: 169 k cycles per workgroup against 63 k with
  // one lane).  So the descriptor of the tile AFTER the next one is asked for a whole tile ahead.
  // (… and through the VECTOR memory path: a scalar load shares its counter with the LDS operations, so the first LDS
  // wait after it would wait for it as well)
  const uint64_t n_tiles_all = all_lanes && b.tiles ? (uint64_t)b.stats->n_tiles : 0;
  auto desc_ahead = [&](uint64_t tt, uint64_t &d0, uint64_t &d1, uint32_t &dl) -> bool {
    if (!b.tiles) return tile_get(b, tt, d0, d1, dl);
    if (tt >= b.tile_count || tt + b.tile_first >= n_tiles_all) return false;
    const TileDesc *p = b.tiles + tt + b.tile_first;
    asm volatile("" : "+v"(p));  // (the address in vector registers: a global_load, counted by vmcnt)
    const TileDesc d = *p;
    d0 = d.begin;
    d1 = d.end;
    dl = d.lane;
    return true;
  };
  uint64_t at = t + gridDim.x, a0 = 0, a1 = 0;
  uint32_t al = 0;
  bool a_have = false;
  if (all_lanes && have) a_have = desc_ahead(at, a0, a1, al);
  while (have) {
    const uint64_t s0 = t0 + (uint64_t)sub * TT;
    const uint64_t s1 = s0 + TT < t1 ? s0 + TT : t1;
    __syncthreads();  // previous sub-tile's write phase is done with sorted/recs/cnt/tstart/gbase
#ifdef SHK_PHASE_TIMING
    tprev = __builtin_readcyclecounter();
#endif
    for (uint32_t i = threadIdx.x; i < P; i += NT) cnt[i] = 0;
    // ---- stage: this thread's group (and, threads 0 and 1, a halo group) → registers + one LDS pair ----
    uint32_t my_pw, my_gm;
    {
      uint32_t nn = 0;
      my_pw = stage32_group_regs(b, s0, s1, (int)threadIdx.x + 2, pre.raw[0], pre.sb0[0], pre.sb1[0], stats, &my_gm, &nn);
      pg[threadIdx.x + 2] = make_uint2(my_pw, my_gm);
      if (threadIdx.x < 2) {
        uint32_t gm1;
        const uint32_t pw1 = stage32_group_regs(b, s0, s1, (int)threadIdx.x, pre.raw[1], pre.sb0[1], pre.sb1[1], stats, &gm1, &nn);
        pg[threadIdx.x] = make_uint2(pw1, gm1);
      }
      if (all_lanes) {  // this tile's lane: wave sum, one LDS add per wave
        uint32_t v = nn;
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&lane_nn[lane], (unsigned long long)v);
      } else {
        n_non_n += nn;
      }
    }
    // the next sub-tile: of this macro tile, or the first one of this workgroup's next macro tile
    uint64_t nt = t, n0 = t0, n1 = t1;
    uint32_t nl = lane, nsub = sub + 1;
    bool hn = true;
    if (t0 + (uint64_t)nsub * TT >= t1) {
      nsub = 0;
      if (all_lanes) {  // the descriptor asked for a tile ago; the one after it is asked for now
        nt = at, hn = a_have, n0 = a0, n1 = a1, nl = al;
        at = nt + gridDim.x;
        if (hn) a_have = desc_ahead(at, a0, a1, al);
      } else {
        nt = t + gridDim.x;
        hn = next_tile(b, nt, !all_lanes, lane_filter, n0, n1, nl);
      }
    }
    if (hn) stage32_prefetch<NT, TT>(b, n0 + (uint64_t)nsub * TT, pre);  // in flight during the rest of this one
    __syncthreads();
    STAMP(0);
    // ---- walk ---------------------------------------------------------------------------------
    uint32_t pr[SPAN];
    {
      const int e0 = threadIdx.x * SPAN;
      const int n_end = (int)(s1 - s0);
      const int jemit = HALO + e0;
      const int jend = HALO + (e0 + SPAN < n_end ? e0 + SPAN : n_end);
      Roll x{0, 0, 0, 0};
      // the two groups before mine: their packed words are the walk's warm-up window, their masks the
      // history of my validity bits
      const uint2 gA = pg[threadIdx.x], gB = pg[threadIdx.x + 1];
      const uint32_t my_ok = stage32_okbits(gA.y, gB.y, my_gm, k, n_end + HALO - ((int)threadIdx.x + 2) * 16);
      {
        // frames just before the first end position: the k bases up to jemit-1 are the low 2k bits of the
        // 32 bases of the two groups before mine
        const uint64_t f0 = (((uint64_t)gA.x << 32) | gB.x) & mask;
        const uint64_t r0 = revcomp(f0, k) << (64 - 2 * k);
        x.f_lo = (uint32_t)f0;
        x.f_hi = (uint32_t)(f0 >> 32);
        x.r_lo = (uint32_t)r0;
        x.r_hi = (uint32_t)(r0 >> 32);
      }
      // Copies of the loop: with 32-bit records — k = 21 on 1024 pages — page and record are simply
      // the two words of the mixed key; and for the usual k (KC = 17, 19, 21; KC = 0: any k at run
      // time) the masks, shifts and the fold of mix_key are compile-time constants (scatter
      // 0.461 → 0.450 → 0.431 ms at k = 21).
      auto walk = [&](auto rb32_t, auto k_t) {
      constexpr bool RB32 = decltype(rb32_t)::value;
      constexpr int KC = decltype(k_t)::value;
      constexpr uint64_t MKC = KC ? (1ull << (2 * KC)) - 1 : 0;
      const int kk = KC ? KC : k;
      const uint32_t mlo = KC ? (uint32_t)MKC : mask_lo, mhi = KC ? (uint32_t)(MKC >> 32) : mask_hi;
      // this thread's sixteen end positions are exactly group 2 + threadIdx.x of the staged tile: their
      // bases are one word of the packed stream (first base on top), their "a k-mer ends here" bits one
      // half-word (never set at or beyond the tile's end)
      const uint32_t pw = my_pw;
      const uint32_t okw = my_ok;
      const uint32_t spare_pc = P + (threadIdx.x & 7u);
      (void)jend;
      (void)jemit;
      const uint32_t rbits_c = RB32 ? 32u : rbits;
      // the window copies (KC ≥ 17): the 48 bases in sight and their reverse complement, as packed words
      const uint32_t w0 = gA.x, w1 = gB.x, w2 = pw;
      const uint32_t rw0 = rev2(~w2), rw1 = rev2(~w1), rw2 = rev2(~w0);
      (void)w0, (void)w1, (void)w2, (void)rw0, (void)rw1, (void)rw2;
      (void)rbits_c;
#pragma unroll
      for (int q = 0; q < SPAN / 8; ++q) {
        // Straight-line on purpose: every lane mixes and issues its LDS add (end positions without
        // a k-mer — bit 2 clear; never set at or beyond the tile's end — count into a spare
        // counter behind cnt[P-1]), and the eight returned ranks are only looked at after the
        // eighth add has been issued.  With a branch per k-mer the wave sits out a full LDS round
        // trip for every single rank.
        uint32_t pcs[8], rks[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const uint32_t c = ((pw >> (30 - 2 * (q * 8 + r))) & 3u) | (((okw >> (q * 8 + r)) & 1u) << 2);
          uint64_t y;
          uint32_t page;
#ifdef SHK_EXP_GENERIC_WALK
          if (false) {
#else
          if (KC >= 17) {
#endif
            // Hand-shaped for instruction COUNT (every integer VALU op of this loop — 64-bit shifts, compares
            // and v_mad_u64_u32 included — issues at the same 4 cycles per wave on gfx950, tools/ibench2.hip,
            // and the kernel is 70 % VALU-busy).  No rolling state at all: the 48 bases this thread can see
            // — the two groups before its own and its own, three packed words, first base on top — hold
            // every k-mer it emits as a WINDOW, and so do the three words of their reverse complement
            // (rw0..rw2, built once per tile).  A frame LEFT-aligned in 64 bits is two funnel shifts by
            // compile-time counts; the bits below the k-mer are neighbouring bases, cleared once, after the
            // select: min(fwd, rev) is one 64-bit compare on the frames as they are (k odd: a k-mer is never
            // its own reverse complement, so those bits cannot decide it; k even: a tie is a palindrome and
            // both arms are the same k-mer).  The product by the 32-bit multiplier is one v_mul_lo_u32 + one
            // v_mad_u64_u32: mod 2^64 on a left-aligned key IS mod 2^2k on the key.
            constexpr int KK = KC ? KC : 1;
            constexpr uint32_t LS = 64u - 2u * KK;                 // left shift of a frame (22 at k = 21)
            constexpr int j = 0;
            (void)j;
            const int e = q * 8 + r;                               // this step's end position in my group (compile-time: unrolled)
            const int of = 2 * (32 + e - KK + 1);                  // bit offset of the forward window in w0:w1:w2
            const int orv = 2 * (15 - e);                          // … of the reverse-complement window in rw0:rw1:rw2
            uint32_t f_hi, f_lo, r_hi, r_lo;
            if (of < 32) {
              f_hi = of ? __builtin_amdgcn_alignbit(w0, w1, (32 - of) & 31) : w0;
              f_lo = of ? __builtin_amdgcn_alignbit(w1, w2, (32 - of) & 31) : w1;
            } else {
              f_hi = of > 32 ? __builtin_amdgcn_alignbit(w1, w2, (64 - of) & 31) : w1;
              f_lo = of > 32 ? w2 << ((of - 32) & 31) : w2;
            }
            r_hi = orv ? __builtin_amdgcn_alignbit(rw0, rw1, (32 - orv) & 31) : rw0;
            r_lo = orv ? __builtin_amdgcn_alignbit(rw1, rw2, (32 - orv) & 31) : rw1;
            // min(fwd, rev) on the 32-bit halves: the borrow of rev - fwd (two carry-chained subtracts) selects —
            // a 64-bit compare wants its operands in register PAIRS, which costs two or three v_mov per k-mer
            uint32_t c_hi, c_lo, scratch;
            asm("v_sub_co_u32 %2, vcc, %5, %3\n\t"
                "v_subb_co_u32 %2, vcc, %6, %4, vcc\n\t"
                "v_cndmask_b32 %0, %3, %5, vcc\n\t"
                "v_cndmask_b32 %1, %4, %6, vcc"
                : "=&v"(c_lo), "=&v"(c_hi), "=&v"(scratch)
                : "v"(f_lo), "v"(f_hi), "v"(r_lo), "v"(r_hi)
                : "vcc");
            const uint64_t cL = (((uint64_t)c_hi << 32) | c_lo) & ~((1ull << LS) - 1ull);
            const uint64_t yL = (uint64_t)(uint32_t)cL * (uint32_t)MIX_M32 + ((uint64_t)((uint32_t)(cL >> 32) * (uint32_t)MIX_M32) << 32);
            y = yL >> LS;                                          // (the record below takes its low 32 bits)
            page = (uint32_t)(yL >> ((LS + rbits_c) & 63));
          } else {
            roll_step(x, c & 3u, mlo, mhi);
            const uint64_t fwd = ((uint64_t)x.f_hi << 32) | x.f_lo;
            const uint64_t rev = (((uint64_t)x.r_hi << 32) | x.r_lo) >> (64 - 2 * kk);
            y = mix_key(rev < fwd ? rev : fwd, 2 * kk);
            page = RB32 ? (uint32_t)(y >> 32) : (uint32_t)(y >> rbits);
          }
          // emit mask as arithmetic (all ones / zero) and a bit-select instead of ?: — the compiler turns a
          // select whose one arm is expensive into an exec-mask branch around that arm, per k-mer
          uint32_t em;
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(em) : "v"(okw), "n"(q * 8 + r));  // all ones where a k-mer ends here
          if (OWN) em &= 0u - (uint32_t)((((page >> log_p1w) ^ own.keep) & keep_mask) == 0u);  // a foreign owner's record: dropped
          asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(pcs[r]) : "v"(em), "v"(page), "v"(spare_pc));  // (em & page) | (~em & spare)
          recs[(q * 8 + r) * NT + threadIdx.x] = RB32 ? (uint32_t)y : (uint32_t)y & rmask;  // transposed: no bank conflicts
          rks[r] = atomicAdd(&cnt[pcs[r]], 1u);  // rank < 2^14
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) pr[q * 8 + r] = (pcs[r] << 18) | rks[r];  // partition · 2^18 | rank (a spare counter, ≥ P: no record — place_entries looks)
      }
      };
      auto walk_k = [&](auto rb32_t) {
        switch (k) {
          case 21: walk(rb32_t, std::integral_constant<int, 21>{}); break;
          case 19: walk(rb32_t, std::integral_constant<int, 19>{}); break;
          case 17: walk(rb32_t, std::integral_constant<int, 17>{}); break;
          default: walk(rb32_t, std::integral_constant<int, 0>{}); break;
        }
      };
      if (rbits == 32) walk_k(std::true_type{});
      else walk_k(std::false_type{});
    }
    __syncthreads();
    STAMP(2);
    // ---- exclusive scan of the counts → tstart ---------------------------------------------------
    {
      // (a run takes an even number of entries: the write phase works on aligned pairs, and a pair never
      // straddles two partitions; an odd run's last pair ends in a hole)
      uint32_t lo = threadIdx.x * per, sacc = 0;
      if (lo < P)
        for (uint32_t i = 0; i < per; ++i) sacc += (cnt[lo + i] + 1u) & ~1u;
      const uint32_t inc = wave_scan_incl(sacc);
      if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
      __syncthreads();
      // offset of this wave = sum of the totals of the waves before it: one masked read per lane
      // and a wave reduction instead of a serial loop over up to NT/64 LDS reads
      const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
      const uint32_t part = (ln < wv && ln < (uint32_t)(NT / 64)) ? wsum[ln] : 0u;
      const uint32_t woff = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(part), 63);
      uint32_t run = woff + inc - sacc;
      if (lo < P)
        for (uint32_t i = 0; i < per; ++i) {
          tstart[lo + i] = run;
          run += (cnt[lo + i] + 1u) & ~1u;
        }
    }
    const uint32_t rbase = all_lanes ? lane * P : 0u;  // first region of this tile's lane
    // reserve this sub-tile's run in every page's region: one returning device-scope add per
    // non-empty (sub-tile, page); the results are parked in registers over the place phase
    uint32_t gres[MAX_PARTS / NT > 0 ? MAX_PARTS / NT : 1];
#pragma unroll
    for (int r = 0; r < (int)(sizeof(gres) / 4); ++r) {
      const uint32_t i = threadIdx.x + r * NT;
      gres[r] = 0;
      if (i < P) {
        const uint32_t c1 = cnt[i];
        if (c1) gres[r] = atomicAdd(&cursor[OWN ? own_seg(i) * n_grp + own_grp(i, lane) : rbase + i], c1);
      }
    }
    __syncthreads();  // codes are dead from here: `sorted` may overwrite them; tstart is complete
    STAMP(3);
    // ---- place ------------------------------------------------------------------------------------
    place_entries<NT, SPAN>(sorted, tstart, pr, P);
#pragma unroll
    for (int r = 0; r < (int)(sizeof(gres) / 4); ++r) {
      const uint32_t i = threadIdx.x + r * NT;
      if (i < P) {
        const uint32_t ts = tstart[i], c1 = cnt[i];
        gbase[i] = (gres[r] - ts) << 2;              // (overwrites cnt[i]: nobody else looks at it any more)
        if (c1 & 1u) sorted[ts + c1] = 0xFFFFFFFFu;  // the hole behind an odd run
        if (i == P - 1) n_ent_sh = ts + ((c1 + 1u) & ~1u);
      }
    }
    __syncthreads();
    STAMP(4);
    // ---- write: one aligned PAIR of entries per lane and step, both of one partition -------------------
    {
      const uint32_t n_ent = n_ent_sh;  // entries, holes included (even)
      const uint2 *sorted2 = reinterpret_cast<const uint2 *>(sorted);
      char *const base = reinterpret_cast<char *>(part_buf32);
      const char *const recs_b = reinterpret_cast<const char *>(recs);
      const char *const gbase_b = reinterpret_cast<const char *>(gbase);
      auto spill_rec = [&](uint32_t pc, uint32_t rec) {  // the page's region is full (skewed input)
        const uint64_t km = unmix_key(((uint64_t)pc << rbits) | rec, 2 * k);
        const unsigned long long j = atomicAdd(OWN && sp.count ? sp.count : &stats->spill_count, 1ull);
        if (j < sp.cap) {
          sp.keys[j] = km;
          sp.lanes[j] = lane;
          sp.counts[j] = 1u;
        }
      };
      // record index `at` inside partition pc's region → where it lives (block-interleaved regions; WIDE:
      // the buffer accumulates the records of many launches and needs 64-bit offsets)
      auto slot_ptr = [&](uint32_t pc, uint32_t at) -> uint32_t * {
        if (OWN)  // segment of pc's owner, then the block-interleave among that owner's regions
          return reinterpret_cast<uint32_t *>(base + (own_seg(pc) * own.seg_recs + rec_slot(own_grp(pc, lane), n_grp, at)) * 4u);
        if (WIDE) return part_buf32 + rec_slot64(rbase + pc, n_regions, at);
        return reinterpret_cast<uint32_t *>(base + rec_slot(rbase + pc, n_regions, at) * 4u);
      };
      const uint32_t cap4 = cap_p << 2;
#pragma unroll 2
      for (uint32_t i = threadIdx.x; 2 * i < n_ent; i += NT) {
        const uint2 ee = sorted2[i];
        const uint32_t pcb = ee.x >> 16;  // partition · 4 (a byte offset into gbase)
        const uint32_t r0 = *reinterpret_cast<const uint32_t *>(recs_b + (ee.x & 0xFFFCu));
        const uint32_t r1 = *reinterpret_cast<const uint32_t *>(recs_b + (ee.y & 0xFFFCu));  // (a hole reads some record: unused)
        const uint32_t at4 = *reinterpret_cast<const uint32_t *>(gbase_b + pcb) + 8u * i;  // record index inside the region, · 4
        const bool hole = ee.y == 0xFFFFFFFFu;
        const bool edge = (at4 & (((1u << RB_LOG) - 1u) << 2)) == (((1u << RB_LOG) - 1u) << 2);  // last record of a block
        if (!hole && !edge && at4 + 8u <= cap4) {
          const uint2 rec2 = make_uint2(r0, r1);  // (both records in one block: one 8-byte store)
          if (!OWN && !WIDE && !ALL && LOGP) {
            // 2^LOGP regions of one lane: byte offset = at·4 + (at >> 10)·(regions - 1)·4096 + region·4096
            const uint32_t off = __umul24(at4 >> (RB_LOG + 2), (uint32_t)(((1u << LOGP) - 1u) << (RB_LOG + 2))) + at4 + (pcb << RB_LOG);
            __builtin_memcpy(base + off, &rec2, 8);
          } else if (!OWN && !WIDE && ALL) {
            // the same with the regions of all lanes block-interleaved together (their number is a wave-uniform
            // run-time value; a lane's buffer stays below 4 GiB or the launch is a WIDE one)
            const uint32_t off = (at4 >> (RB_LOG + 2)) * ((n_regions - 1u) << (RB_LOG + 2)) + at4 + ((rbase << (RB_LOG + 2)) + (pcb << RB_LOG));
            __builtin_memcpy(base + off, &rec2, 8);
          } else {
            __builtin_memcpy(slot_ptr(pcb >> 2, at4 >> 2), &rec2, 8);
          }
        } else {
          const uint32_t pc = pcb >> 2, at0 = at4 >> 2;
          if (at0 < cap_p) *slot_ptr(pc, at0) = r0;
          else spill_rec(pc, r0);
          if (!hole) {
            if (at0 + 1 < cap_p) *slot_ptr(pc, at0 + 1) = r1;
            else spill_rec(pc, r1);
          }
        }
      }
    }
    STAMP(5);
    t = nt;
    t0 = n0;
    t1 = n1;
    lane = nl;
    sub = nsub;
    have = hn;
  }
  __syncthreads();
  if (all_lanes) {
    if (threadIdx.x < SC32_MAX_LANES && threadIdx.x < n_region_lanes && lane_nn[threadIdx.x])
      atomicAdd(&lane_bases[threadIdx.x], lane_nn[threadIdx.x]);
  } else {
    uint32_t tot = wg_sum<NT>(n_non_n, red);
    if (threadIdx.x == 0 && tot) atomicAdd(&lane_bases[lane_filter], (unsigned long long)tot);
  }
#ifdef SHK_PHASE_TIMING
  if (dbg && threadIdx.x == 0)
    for (int i = 0; i < 8; ++i) dbg[(uint64_t)blockIdx.x * 8 + i] = ph[i];
#endif
}

// ------------------------------------------------------------------------------------------
// k_scatter64: the 8-byte-record scatter (k-mers whose mixed-key remainder does not fit 32 bits: every
// table at k ≥ 22) with k_scatter32's machine — a persistent 1024-thread workgroup per CU, 16 Ki-position
// tiles, the thread's group staged in registers, the window walk — and the RECORDS (canonical k-mers, what
// k_part_rescatter and k_pages read) carried in registers from the walk to the place phase:
//   walk   : windows of the 48 bases in sight → canonical k-mer (two registers per end position, sixteen
//            positions per thread) + partition off the top of the mixed key + rank (returning LDS add)
//   scan, reserve : as k_scatter32 (runs are even: the write-out moves aligned PAIRS)
//   place  : sorted[tstart[partition] + rank] = k-mer — the 8-byte record itself, not an index
//   write  : one 16-byte LDS read and one 16-byte store per lane and step; the partition of a pair is
//            re-read off its first k-mer (one 64-bit multiply per PAIR)
// k_part_scatter_sorted<.., false>, which it replaces for ≤ 1024 partitions and k ≥ 18, kept 16-bit
// position entries and rebuilt every k-mer from the packed tile in its write phase (three LDS reads, funnel
// shifts and the multiply per RECORD: 37 % of that kernel on BASELINE configs[2]).
// LDS: 8·(TT + P) (the sorted records) + 12·P + 8·(NT + 2) (the staged pairs, a buffer of their own: the next
// tile is staged while this one's records wait to be written); same output, same cursors, same padding (EMPTY
// behind an odd run) as the kernel it replaces.
// ------------------------------------------------------------------------------------------
// IL: the output regions BLOCK-INTERLEAVED in blocks of RS_TILE records (record j of region r at
// ((j / RS_TILE) · P + r) · RS_TILE + j mod RS_TILE) — the level-1 buffer k_part_rescatter reads tile by tile: the P
// write fronts stay within P · 32 KiB of one another instead of one per region-sized stride (a dozen address
// translations per store instruction otherwise).  !IL: region r at r · cap_p (page regions k_pages reads).
constexpr int S64_IL_LOG = 12;  // = log2(RS_TILE), asserted below
// OWN (layout 2): the OWNER layout of the exchange between owner shares, as k_scatter32's — the fan-out covers the
// virtual table (the top own.log_w bits of a partition are the record's owner), regions and cursors are ordered
// [owner][lane][super-page], every owner's own.seg_recs records' worth of regions one contiguous piece (what crosses the
// link), linear inside; records of all owners are kept, a full region spills to the list sp.count counts.
constexpr int S64_LINEAR = 0, S64_INTERLEAVED = 1, S64_OWNER = 2;
template <int NT, int TT, int LAYOUT>
__global__ void __launch_bounds__(NT, 4) k_scatter64(
    BatchRef b, uint32_t log_parts, uint32_t lane_filter, unsigned int *__restrict__ cursor,
    uint32_t cap_p, uint64_t *__restrict__ part_buf, DevStats *__restrict__ stats,
    unsigned long long *__restrict__ lane_bases, SpillRef sp, unsigned long long *__restrict__ dbg,
    OwnerCfg own = OwnerCfg{}, uint32_t n_region_lanes = 1) {
  constexpr bool IL = LAYOUT == S64_INTERLEAVED, OWN = LAYOUT == S64_OWNER;
  const uint32_t log_p1w = OWN ? log_parts - own.log_w : 0u;  // super-page bits inside an owner's share
  const uint32_t p1w_mask = (1u << log_p1w) - 1u;
  const uint32_t n_grp = n_region_lanes << log_p1w;           // regions of one owner segment
  // region of partition i for a tile of chunk lane ln, among ALL regions of the launch ([owner][lane][super-page]);
  // own.keep = an owner id: only that owner's records are kept (dropped in the walk) and there is ONE segment
  // (own.seg_recs = 0) — a share's own ingest; own.keep = ~0: every owner's records, a segment each (the exchange)
  const uint32_t keep_mask = own.keep == 0xFFFFFFFFu ? 0u : 0xFFFFFFFFu;
  auto own_region = [&](uint32_t i, uint32_t ln) -> uint32_t {
    return (own.seg_recs ? i >> log_p1w : 0u) * n_grp + ((ln << log_p1w) | (i & p1w_mask));
  };
  extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
  __shared__ uint32_t wsum[NT / 64];
  __shared__ uint32_t red[NT / 64];
  __shared__ uint32_t n_ent_sh;
  static_assert(TILE_T % TT == 0 && TT % (8 * NT) == 0, "tile shape");
  constexpr int SPAN = TT / NT;
  constexpr int GROUPS = (TT + HALO) / 16;
  static_assert(SPAN == 16 && GROUPS == NT + 2, "a thread walks exactly the group it staged");
  const uint32_t P = 1u << log_parts;                  // ≤ NT
  uint2 *sorted = reinterpret_cast<uint2 *>(sh);       // TT + P records (every run starts at an EVEN index)
  uint32_t *cnt = sh + 2 * (TT + P);                   // P
  uint32_t *tstart = cnt + P;                          // P  (tstart[0..7] double as the walk's spare counters: rewritten by every scan)
  uint32_t *gbase = tstart + P;                        // P: (this tile's reservation) − tstart, in records
  uint2 *pg = reinterpret_cast<uint2 *>(gbase + P);    // GROUPS staged (packed word, masks) pairs — of the NEXT tile while this one is written out
#ifdef SHK_PHASE_TIMING
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
#endif
  uint32_t n_non_n = 0;
  const int k = b.k;                                   // 18 ≤ k ≤ 31 (the host's choice)
  const uint32_t LS = 64u - 2u * (uint32_t)k;          // left shift of a frame: 2 … 28
  const uint32_t of0 = 2u * (33u - (uint32_t)k);       // bit offset of the first end position's window in w0:w1:w2 (4 … 30)
  const uint32_t lo_keep = 0xFFFFFFFFu << LS;          // the k-mer's bits of a left-aligned frame's low word
  const uint64_t M = 2u * (uint32_t)k <= MIX_NARROW_BITS ? MIX_M32 : MIX_M64;
  const uint32_t M_lo = (uint32_t)M, M_hi = (uint32_t)(M >> 32);
  const uint32_t psh = 32u - log_parts;                // log_parts ≥ 1

  // The tiles of this workgroup, one after another; a tile's sub-tiles of TT end positions.  Three in flight: the one
  // being walked (c_*), the one staged next (n_*: its bytes are in `pre`), the one after it (asked for when n is staged).
  uint64_t c_t = blockIdx.x, c_t0 = 0, c_t1 = 0, n_t = 0, n_t0 = 0, n_t1 = 0;
  uint32_t c_lane = 0, c_sub = 0, n_lane = 0, n_sub = 0;
  bool c_have = next_tile(b, c_t, true, lane_filter, c_t0, c_t1, c_lane), n_have = false;
#define SHK_S64_BOUNDS(t0_, t1_, sub_, s0_, s1_) \
  const uint64_t s0_ = (t0_) + (uint64_t)(sub_) * TT; \
  const uint64_t s1_ = s0_ + TT < (t1_) ? s0_ + TT : (t1_)
  // (XX_*) ← the sub-tile after (YY_*)
#define SHK_S64_ADVANCE(XX, YY)                                                        \
  do {                                                                               \
    XX##_t = YY##_t, XX##_t0 = YY##_t0, XX##_t1 = YY##_t1, XX##_lane = YY##_lane;            \
    XX##_sub = YY##_sub + 1;                                                           \
    XX##_have = true;                                                                 \
    if (YY##_t0 + (uint64_t)XX##_sub * TT >= YY##_t1) {                                 \
      XX##_sub = 0;                                                                   \
      XX##_t = YY##_t + gridDim.x;                                                     \
      XX##_have = next_tile(b, XX##_t, true, lane_filter, XX##_t0, XX##_t1, XX##_lane);   \
    }                                                                                \
  } while (0)
  StageRegs32<NT, TT> pre;
  uint32_t my_pw = 0, my_gm = 0;  // this thread's group of the tile about to be walked: packed bases, N | read-start masks
  // stage: the prefetched bytes of a sub-tile → this thread's registers + one LDS pair for the neighbours
  auto stage = [&](uint64_t s0, uint64_t s1) {
    uint32_t nn = 0;
    my_pw = stage32_group_regs(b, s0, s1, (int)threadIdx.x + 2, pre.raw[0], pre.sb0[0], pre.sb1[0], stats, &my_gm, &nn);
    pg[threadIdx.x + 2] = make_uint2(my_pw, my_gm);
    if (threadIdx.x < 2) {
      uint32_t gm1;
      const uint32_t pw1 = stage32_group_regs(b, s0, s1, (int)threadIdx.x, pre.raw[1], pre.sb0[1], pre.sb1[1], stats, &gm1, &nn);
      pg[threadIdx.x] = make_uint2(pw1, gm1);
    }
    n_non_n += nn;
  };

  if (threadIdx.x < P) cnt[threadIdx.x] = 0;
  if (c_have) {
    SHK_S64_BOUNDS(c_t0, c_t1, c_sub, s0, s1);
    stage32_prefetch<NT, TT>(b, s0, pre);
    stage(s0, s1);
    SHK_S64_ADVANCE(n, c);
    if (n_have) {
      SHK_S64_BOUNDS(n_t0, n_t1, n_sub, ns0, ns1);
      (void)ns1;
      stage32_prefetch<NT, TT>(b, ns0, pre);
    }
  }
  // Order of a tile's phases: walk, scan + reserve, place, THE NEXT TILE'S STAGE, write.  The stores of the write
  // phase are the last thing a tile issues: they drain under the next tile's walk, and nothing that waits on the
  // vector-memory counter (the prefetched bytes are consumed before them, the reservations' returns a walk later)
  // stands behind them.
  while (c_have) {
    SHK_S64_BOUNDS(c_t0, c_t1, c_sub, s0, s1);
    const uint32_t lane = c_lane;
    __syncthreads();  // the staged pairs and the cleared counts are there; the previous write phase is done with sorted / gbase
#ifdef SHK_PHASE_TIMING
    tprev = __builtin_readcyclecounter();
#endif
    // ---- walk ---------------------------------------------------------------------------------
    uint32_t pr[SPAN], km_lo[SPAN], km_hi[SPAN];
    {
      const int n_end = (int)(s1 - s0);
      const uint2 gA = pg[threadIdx.x], gB = pg[threadIdx.x + 1];
      const uint32_t okw = stage32_okbits(gA.y, gB.y, my_gm, k, n_end + HALO - ((int)threadIdx.x + 2) * 16);
      const uint32_t spare_pc = P + (threadIdx.x & 7u);
      // the 48 bases in sight (the two groups before mine and mine, first base on top), shifted up so that
      // the window of end position e starts 2e bits in — a compile-time funnel shift for ANY k — and the
      // three words of their reverse complement, where that window starts 2·(15 − e) bits in
      const uint32_t w0 = gA.x, w1 = gB.x, w2 = my_pw;
      const uint32_t v0 = __builtin_amdgcn_alignbit(w0, w1, 32u - of0), v1 = __builtin_amdgcn_alignbit(w1, w2, 32u - of0), v2 = w2 << of0;
      const uint32_t rw0 = rev2(~w2), rw1 = rev2(~w1), rw2 = rev2(~w0);
#pragma unroll
      for (int q = 0; q < SPAN / 8; ++q) {
        uint32_t pcs[8], rks[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int e = q * 8 + r;
          const int orv = 2 * (15 - e);
          const uint32_t f_hi = e ? __builtin_amdgcn_alignbit(v0, v1, (32 - 2 * e) & 31) : v0;
          const uint32_t f_lo = e ? __builtin_amdgcn_alignbit(v1, v2, (32 - 2 * e) & 31) : v1;
          const uint32_t r_hi = orv ? __builtin_amdgcn_alignbit(rw0, rw1, (32 - orv) & 31) : rw0;
          const uint32_t r_lo = orv ? __builtin_amdgcn_alignbit(rw1, rw2, (32 - orv) & 31) : rw1;
          // min(fwd, rev) on the frames as they are (see k_scatter32): the borrow of rev − fwd selects
          uint32_t c_hi, c_lo, scratch;
          asm("v_sub_co_u32 %2, vcc, %5, %3\n\t"
              "v_subb_co_u32 %2, vcc, %6, %4, vcc\n\t"
              "v_cndmask_b32 %0, %3, %5, vcc\n\t"
              "v_cndmask_b32 %1, %4, %6, vcc"
              : "=&v"(c_lo), "=&v"(c_hi), "=&v"(scratch)
              : "v"(f_lo), "v"(f_hi), "v"(r_lo), "v"(r_hi)
              : "vcc");
          km_lo[e] = __builtin_amdgcn_alignbit(c_hi, c_lo, LS);  // the k-mer, right-aligned (LS < 32)
          km_hi[e] = c_hi >> LS;
          // top word of (left-aligned key · M) mod 2^64 = the top of the mixed key
          const uint32_t cl = c_lo & lo_keep;
          const uint32_t yhi = __umulhi(cl, M_lo) + cl * M_hi + c_hi * M_lo;
          const uint32_t page = yhi >> psh;
          uint32_t em;
          asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(em) : "v"(okw), "n"(q * 8 + r));  // all ones where a k-mer ends here
          if (OWN) em &= 0u - (uint32_t)((((page >> log_p1w) ^ own.keep) & keep_mask) == 0u);  // a foreign owner's record: dropped
          asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(pcs[r]) : "v"(em), "v"(page), "v"(spare_pc));
          rks[r] = atomicAdd(&cnt[pcs[r]], 1u);  // rank < 2^14
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) pr[q * 8 + r] = (pcs[r] << 18) | rks[r];
      }
    }
    __syncthreads();
    STAMP(2);
    // ---- exclusive scan of the even-padded counts → tstart ---------------------------------------
    uint32_t gres = 0, my_c1 = 0;
    {
      const bool mine = threadIdx.x < P;
      my_c1 = mine ? cnt[threadIdx.x] : 0u;
      const uint32_t sacc = (my_c1 + 1u) & ~1u;
      // reserve this sub-tile's (even-padded) run in the partition's region: the return is looked at in the place phase
      if (my_c1) gres = atomicAdd(&cursor[OWN ? own_region(threadIdx.x, lane) : threadIdx.x], sacc);
      const uint32_t inc = wave_scan_incl(sacc);
      if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
      __syncthreads();
      const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
      const uint32_t part = (ln < wv && ln < (uint32_t)(NT / 64)) ? wsum[ln] : 0u;
      const uint32_t woff = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(part), 63);
      if (mine) tstart[threadIdx.x] = woff + inc - sacc;
    }
    __syncthreads();  // tstart is complete; nobody looks at the counts any more
    STAMP(3);
    // ---- place: the records themselves ---------------------------------------------------------
    {
      const char *const ts_b = reinterpret_cast<const char *>(tstart);
      uint2 *__restrict__ const dst = sorted;
#pragma unroll
      for (int i = 0; i < SPAN; ++i) {
        const uint32_t v = pr[i];  // partition · 2^18 | rank; partitions ≥ P are the spare counters (no record)
        if (v < (P << 18)) dst[*reinterpret_cast<const uint32_t *>(ts_b + (v >> 16)) + (v & 0x3FFFFu)] = make_uint2(km_lo[i], km_hi[i]);
      }
      if (threadIdx.x < P) {
        const uint32_t ts = tstart[threadIdx.x];
        gbase[threadIdx.x] = gres - ts;
        if (my_c1 & 1u) dst[ts + my_c1] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);  // EMPTY behind an odd run
        if (threadIdx.x == P - 1) n_ent_sh = ts + ((my_c1 + 1u) & ~1u);
        cnt[threadIdx.x] = 0;  // for the next tile's walk
      }
    }
    STAMP(4);
    // ---- the next tile's stage (its bytes were asked for a tile ago), and the one after it asked for ----
    uint64_t a_t = 0, a_t0 = 0, a_t1 = 0;
    uint32_t a_lane = 0, a_sub = 0;
    bool a_have = false;
#ifdef SHK_PHASE_TIMING
    asm volatile("s_waitcnt vmcnt(0)");  // (phase build only: what the stage would wait for, booked apart)
    STAMP(6);
#endif
    if (n_have) {
      SHK_S64_BOUNDS(n_t0, n_t1, n_sub, ns0, ns1);
      stage(ns0, ns1);
      SHK_S64_ADVANCE(a, n);
    }
    STAMP(1);
    __syncthreads();  // the sorted records are complete
    STAMP(0);
    // ---- write: one aligned PAIR of records per lane and step, both of one partition; four steps' LDS reads
    // (the pair, then its partition's base) are in flight together — one step at a time the loop is a chain of two
    // LDS round trips per store ------------------------------------------------------------------------------
    {
      const uint32_t n_pairs = n_ent_sh >> 1;  // (records, padding included: even)
      const ulonglong2 *sorted2 = reinterpret_cast<const ulonglong2 *>(sorted);
      char *const base = reinterpret_cast<char *>(part_buf);
#ifndef SHK_S64_WB
#define SHK_S64_WB 1
#endif
      constexpr int WB = SHK_S64_WB;
      for (uint32_t i0 = threadIdx.x; i0 < n_pairs; i0 += WB * NT) {
        ulonglong2 rec[WB];
        uint32_t pc[WB], at[WB];
#pragma unroll
        for (int j = 0; j < WB; ++j) {
          const uint32_t i = i0 + j * NT;
          rec[j] = sorted2[i < n_pairs ? i : n_pairs - 1u];
        }
#pragma unroll
        for (int j = 0; j < WB; ++j) pc[j] = (uint32_t)(((rec[j].x * M) << LS) >> 32) >> psh;
#pragma unroll
        for (int j = 0; j < WB; ++j) at[j] = gbase[pc[j]] + 2 * (i0 + j * NT);  // record index inside partition pc's region
#pragma unroll
        for (int j = 0; j < WB; ++j) {
          if (i0 + j * NT >= n_pairs) continue;
          if (at[j] + 2 <= cap_p) {
            const uint32_t rec_off = IL ? ((((at[j] >> S64_IL_LOG) << log_parts) + pc[j]) << S64_IL_LOG) | (at[j] & ((1u << S64_IL_LOG) - 1u))
                                        : (OWN ? own_region(pc[j], lane) : pc[j]) * cap_p + at[j];
            const uint32_t byte_off = rec_off * 8u;  // < 2^32 (the host sizes a launch's regions so)
            *reinterpret_cast<ulonglong2 *>(base + byte_off) = rec[j];
          } else {  // the region is full (skewed input): these records take the spill path
            const unsigned long long jj = atomicAdd(OWN && sp.count ? sp.count : &stats->spill_count, rec[j].y == EMPTY ? 1ull : 2ull);
            if (jj < sp.cap) {
              sp.keys[jj] = rec[j].x;
              sp.lanes[jj] = lane;
              sp.counts[jj] = 1u;
            }
            if (rec[j].y != EMPTY && jj + 1 < sp.cap) {
              sp.keys[jj + 1] = rec[j].y;
              sp.lanes[jj + 1] = lane;
              sp.counts[jj + 1] = 1u;
            }
          }
        }
      }
    }
    // the bytes of the tile after the next one: asked for BEHIND this tile's stores (in front of them the loads' misses
    // hold up the stores of every wave of the CU in the L1's queue: write phase 2.4× as long), consumed a tile later
    if (a_have) {
      SHK_S64_BOUNDS(a_t0, a_t1, a_sub, as0, as1);
      (void)as1;
      stage32_prefetch<NT, TT>(b, as0, pre);
    }
    STAMP(5);
    c_t = n_t, c_t0 = n_t0, c_t1 = n_t1, c_lane = n_lane, c_sub = n_sub, c_have = n_have;
    n_t = a_t, n_t0 = a_t0, n_t1 = a_t1, n_lane = a_lane, n_sub = a_sub, n_have = a_have;
  }
#undef SHK_S64_BOUNDS
#undef SHK_S64_ADVANCE
  __syncthreads();
  {
    uint32_t tot = wg_sum<NT>(n_non_n, red);
    if (threadIdx.x == 0 && tot) atomicAdd(&lane_bases[lane_filter], (unsigned long long)tot);
  }
#ifdef SHK_PHASE_TIMING
  if (dbg && threadIdx.x == 0)
    for (int i = 0; i < 8; ++i) dbg[(uint64_t)blockIdx.x * 8 + i] = ph[i];
#endif
}

// ------------------------------------------------------------------------------------------
// k_part_rescatter: second level of the partition for tables with more pages than one LDS sort
// can fan out to (> MAX_PARTS).  Level 1 (k_part_scatter_sorted with log_parts < log_pages) has
// grouped the records by SUPER-PAGE (2^log_sub consecutive pages); this kernel takes one
// 16 Ki-record tile of one super-page's region, counting-sorts it by page inside the super-page
// in LDS, reserves each page's share of the final page regions with one returning atomic per
// (tile, page) and writes the records there as aligned pairs — the same machine as level 1 with
// "re-read the record from the LDS copy of the tile" in place of "re-read the k-mer from the
// packed bases" (the fan-out is small here, so 4 Ki-record tiles already give long runs).
// Costs one more 8-B read + 8-B write per k-mer occurrence.
// ------------------------------------------------------------------------------------------
#ifndef SHK_RS_NT
#define SHK_RS_NT 512  // (8 records per thread; with 256 threads the 8-byte re-scatter of config 3 took 58 ms instead of 50)
#endif
constexpr int RS_NT = SHK_RS_NT;
constexpr int RS_TILE = 4096;              // records per tile: they stay in LDS (32 KiB) for the write-out
constexpr int RS_SPAN = RS_TILE / RS_NT;   // records per thread
// LIST mode of k_part_rescatter (below): a flat list of n k-mers (+ lanes) as the source; sub_shift / owner bits: which
// bits of a k-mer's page this pass sorts by (0 / none for the ordinary second level)
struct RescatterList {
  const uint32_t *lanes;  // chunk lane per k-mer, or nullptr: all of them are `lane`'s
  uint64_t n;             // 0: not a list
  uint32_t sub_shift, owner_bits, owner_id, pad;
};
__global__ void __launch_bounds__(RS_NT) k_part_rescatter(
    const uint64_t *__restrict__ src_buf, const unsigned int *__restrict__ src_cursor, uint32_t src_cap,
    uint32_t tiles_per_region, uint32_t log_pages, uint32_t log_sub, uint32_t key_bits,
    unsigned int *__restrict__ dst_cursor,
    uint32_t dst_cap, uint64_t *__restrict__ dst_buf, uint32_t lane, DevStats *__restrict__ stats,
    SpillRef sp, uint32_t src_interleaved, RescatterList ls = RescatterList{}) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
  __shared__ uint32_t wsum[RS_NT / 64];
  static_assert(RS_TILE == 1 << S64_IL_LOG, "k_scatter64's interleave block is this kernel's tile");
  if (stats->bad != ~0ull) return;
  const uint32_t S = 1u << log_sub;  // pages per super-page
  // (the tiles of one source region go to ONE XCD: see k_part_rescatter32)
  uint32_t bid = blockIdx.x;
  if ((gridDim.x & 7u) == 0) bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  // LIST mode (ls.n > 0): the source is a flat list of k-mers (+ their chunk lanes) instead of a region — the first
  // level of the partition for k-mers that arrive as k-mers (shk_insert_device: the wide exchange round's receiver).
  // One "region" 0, tile = workgroup; k-mers of other lanes and — on an owner share — of other owners are skipped.
  const bool list = ls.n != 0;
  const uint32_t region = list ? 0u : bid / tiles_per_region, tile = list ? bid : bid % tiles_per_region;
  const uint64_t filled = list ? ls.n : (uint64_t)(src_cursor[region] < src_cap ? src_cursor[region] : src_cap);
  const uint64_t r0 = (uint64_t)tile * RS_TILE;
  if (r0 >= filled) return;
  const uint32_t n = filled - r0 < (uint64_t)RS_TILE ? (uint32_t)(filled - r0) : (uint32_t)RS_TILE;  // (even, except a list's last tile)
  const uint32_t lp_mask = log_pages >= 32 ? 0xFFFFFFFFu : (1u << log_pages) - 1u;
  // the page of a k-mer inside this table (an owner share: the low log_pages bits of its page in the virtual table;
  // *own = the top bits name this share) and from it the partition this pass sorts by
  auto sub_of = [&](uint64_t km, bool *own) -> uint32_t {
    const uint32_t gp = (uint32_t)page_of(hash64(km, key_bits), log_pages + ls.owner_bits);
    if (own) *own = ls.owner_bits == 0 || (gp >> log_pages) == ls.owner_id;
    return ((gp & lp_mask) >> ls.sub_shift) & (S - 1u);
  };
  // (src_interleaved: k_scatter64<.., true>'s layout — tile t of region r is block t · n_regions + r)
  const uint64_t *src = list ? src_buf + r0
                        : src_interleaved ? src_buf + (((uint64_t)tile * (gridDim.x / tiles_per_region) + region) << S64_IL_LOG)
                                          : src_buf + (uint64_t)region * src_cap + r0;
  uint64_t *recs = reinterpret_cast<uint64_t *>(sh);                       // RS_TILE records
  uint16_t *sorted = reinterpret_cast<uint16_t *>(sh + 2 * RS_TILE);       // RS_TILE + S entries
  uint32_t *cnt = sh + 2 * RS_TILE + (((uint32_t)RS_TILE + S) * 2 + 15) / 16 * 4;  // S
  uint32_t *tstart = cnt + S;                                              // S
  uint32_t *gbase = tstart + S;                                            // S
  for (uint32_t i = threadIdx.x; i < S; i += RS_NT) cnt[i] = 0;
  __syncthreads();
  // ---- rank: (page-in-super-page, rank) per record, in registers -------------------------------
  uint32_t pr[RS_SPAN];
#pragma unroll
  for (int q = 0; q < RS_SPAN / 2; ++q) {
    const uint32_t i = (uint32_t)(q * RS_NT + threadIdx.x) * 2;  // record pair, coalesced 16-B loads
    uint32_t va = 0xFFFFFFFFu, vb = 0xFFFFFFFFu;
    if (i < n) {
      ulonglong2 rec;
      if (list) {  // single k-mers, the lane's and the share's own only
        rec.x = !ls.lanes || ls.lanes[r0 + i] == lane ? src[i] : EMPTY;
        rec.y = i + 1 < n && (!ls.lanes || ls.lanes[r0 + i + 1] == lane) ? src[i + 1] : EMPTY;
      } else {
        rec = *reinterpret_cast<const ulonglong2 *>(src + i);
      }
      uint32_t sa = 0, sb = 0;
      if (rec.x != EMPTY) {
        bool own;
        sa = sub_of(rec.x, &own);
        if (!own) rec.x = EMPTY;
      }
      if (rec.y != EMPTY) {
        bool own;
        sb = sub_of(rec.y, &own);
        if (!own) rec.y = EMPTY;
      }
      *reinterpret_cast<ulonglong2 *>(recs + i) = rec;
      if (rec.x != EMPTY) va = (sa << 16) | atomicAdd(&cnt[sa], 1u);
      if (rec.y != EMPTY) vb = (sb << 16) | atomicAdd(&cnt[sb], 1u);
    }
    pr[2 * q] = va;
    pr[2 * q + 1] = vb;
  }
  __syncthreads();
  // ---- exclusive scan of the even-padded counts; reservation in the final page regions ---------
  {
    const uint32_t per = S / RS_NT ? S / RS_NT : 1;
    uint32_t lo = threadIdx.x * per, sacc = 0;
    if (lo < S)
      for (uint32_t i = 0; i < per; ++i) sacc += (cnt[lo + i] + 1u) & ~1u;
    const uint32_t inc = wave_scan_incl(sacc);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const uint32_t part = (ln < wv && ln < (uint32_t)(RS_NT / 64)) ? wsum[ln] : 0u;
    const uint32_t woff = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(part), 63);
    uint32_t run = woff + inc - sacc;
    if (lo < S)
      for (uint32_t i = 0; i < per; ++i) {
        tstart[lo + i] = run;
        run += (cnt[lo + i] + 1u) & ~1u;
      }
  }
  for (uint32_t i = threadIdx.x; i < S; i += RS_NT) {
    const uint32_t c2 = (cnt[i] + 1u) & ~1u;
    gbase[i] = c2 ? atomicAdd(&dst_cursor[((uint64_t)region << log_sub) + i], c2) : 0u;
  }
  __syncthreads();
  // ---- place: entry = record index inside the tile (0xFFFF = pad) ------------------------------------
#pragma unroll
  for (int q = 0; q < RS_SPAN / 2; ++q) {
    const uint32_t i = (uint32_t)(q * RS_NT + threadIdx.x) * 2;
    const uint32_t va = pr[2 * q], vb = pr[2 * q + 1];
    if (va != 0xFFFFFFFFu) sorted[tstart[va >> 16] + (va & 0xFFFFu)] = (uint16_t)i;
    if (vb != 0xFFFFFFFFu) sorted[tstart[vb >> 16] + (vb & 0xFFFFu)] = (uint16_t)(i + 1);
  }
  for (uint32_t i = threadIdx.x; i < S; i += RS_NT)
    if (cnt[i] & 1u) sorted[tstart[i] + cnt[i]] = 0xFFFFu;
  __syncthreads();
  // ---- write: pairs of records, re-read from the LDS copy of the tile ---------------------------
  const uint32_t n_pairs = (tstart[S - 1] + ((cnt[S - 1] + 1u) & ~1u)) >> 1;
  const uint32_t *sorted2 = reinterpret_cast<const uint32_t *>(sorted);
  for (uint32_t i = threadIdx.x; i < n_pairs; i += RS_NT) {
    const uint32_t ee = sorted2[i];
    const uint32_t ea = ee & 0xFFFFu, eb = ee >> 16;
    const uint64_t km0 = recs[ea];
    const uint64_t km1 = eb == 0xFFFFu ? EMPTY : recs[eb];
    const uint32_t sub = sub_of(km0, nullptr);
    const uint32_t at = gbase[sub] + (2 * i - tstart[sub]);
    const uint64_t page = ((uint64_t)region << log_sub) + sub;
    if (at + 2 <= dst_cap) {
      ulonglong2 rec;
      rec.x = km0;
      rec.y = km1;
      *reinterpret_cast<ulonglong2 *>(dst_buf + page * dst_cap + at) = rec;
    } else {
      const unsigned long long j = atomicAdd(&stats->spill_count, km1 == EMPTY ? 1ull : 2ull);
      if (j < sp.cap) {
        sp.keys[j] = km0;
        sp.lanes[j] = lane;
        sp.counts[j] = 1u;
      }
      if (km1 != EMPTY && j + 1 < sp.cap) {
        sp.keys[j + 1] = km1;
        sp.lanes[j + 1] = lane;
        sp.counts[j + 1] = 1u;
      }
    }
  }
}

// k_part_rescatter32: the same second level for 4-byte records.  A level-1 record is the low
// R1 = 2k - log_p1 bits of the mixed key (its super-page is the region it sits in); the page inside
// the super-page is its top log_sub bits, and what goes on to k_pages32 is the rest.  Runs are packed
// (no padding), two entries per lane leave as one 8-B store when they share a page.
#ifndef SHK_RS32_TILE
#define SHK_RS32_TILE 8192
#endif
#ifndef SHK_RS32_NT
#define SHK_RS32_NT 512  // (16 records per thread; 256 threads × 32 records: re-scatter 0.64 → 0.43 ms on a 30 Mb genome)
#endif
constexpr int RS32_TILE = SHK_RS32_TILE;       // records per tile (4 B each in LDS)
constexpr int RS32_NT = SHK_RS32_NT;
constexpr int RS32_SPAN = RS32_TILE / RS32_NT;   // 32 records per thread
static_assert(RS32_TILE <= 65536 && RS32_SPAN % 4 == 0, "16-bit entries; four records per load");
__global__ void __launch_bounds__(RS32_NT) k_part_rescatter32(
    const uint32_t *__restrict__ src_buf, const unsigned int *__restrict__ src_cursor, uint32_t src_cap,
    uint32_t tiles_per_region, uint32_t log_sub, uint32_t r1_bits, uint32_t key_bits,
    unsigned int *__restrict__ dst_cursor, uint32_t dst_cap, uint32_t *__restrict__ dst_buf, uint32_t lane_,
    DevStats *__restrict__ stats, SpillRef sp, uint64_t dst_region_base_, uint64_t n_dst_total,
    uint32_t log_src_lane, uint32_t region_hi, uint64_t dst_lane_stride) {
  extern __shared__ __attribute__((aligned(16))) uint32_t sh[];
  __shared__ uint32_t wsum[RS32_NT / 64];
  if (stats->bad != ~0ull) return;
  const uint32_t S = 1u << log_sub;  // pages per super-page
  // Source regions: one per super-page, or — the owner layout of k_scatter32 — [lane][super-page] with
  // 2^log_src_lane super-pages per lane (log_src_lane = 31: a single lane, lane_).  `region` below is the
  // super-page inside this table's share; region_hi carries the owner bits above it for a key rebuild.
  // Workgroups are dealt to the 8 XCDs round robin; every XCD has its own L2.  All tiles of a source region
  // append to the same 2^log_sub destination regions — runs that lie next to each other in HBM, and cursors
  // that are bumped once per tile — so the tiles of one region are given to ONE XCD (logical block = the XCD's
  // share of the grid, taken in order): its L2 merges the neighbouring partial-line writes and keeps the cursors.
  uint32_t bid = blockIdx.x;
  if ((gridDim.x & 7u) == 0) bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint32_t src_region = bid / tiles_per_region, tile = bid % tiles_per_region;
  const uint32_t lane = lane_ + (src_region >> log_src_lane);
  const uint32_t region = src_region & ((1u << log_src_lane) - 1u);
  const uint64_t dst_region_base = dst_region_base_ + (uint64_t)(src_region >> log_src_lane) * dst_lane_stride;
  const uint32_t filled = src_cursor[src_region] < src_cap ? src_cursor[src_region] : src_cap;
  const uint32_t r0 = tile * RS32_TILE;
  if (r0 >= filled) return;
  const uint32_t n = filled - r0 < (uint32_t)RS32_TILE ? filled - r0 : (uint32_t)RS32_TILE;
  const uint32_t n_src_regions = gridDim.x / tiles_per_region;  // level-1 regions of the source buffer
  uint32_t *recs = sh;                                                      // RS32_TILE records
  uint16_t *sorted = reinterpret_cast<uint16_t *>(sh + RS32_TILE);          // RS32_TILE entries
  uint32_t *cnt = sh + RS32_TILE + RS32_TILE / 2;                           // S
  uint32_t *tstart = cnt + S;                                               // S
  uint32_t *gbase = tstart + S;                                             // S
  for (uint32_t i = threadIdx.x; i < S; i += RS32_NT) cnt[i] = 0;
  __syncthreads();
  // ---- rank: (page-in-super-page, rank) per record, in registers; four records per 16-B load ----
  const uint32_t rbits2 = r1_bits - log_sub;  // record bits that go on to the page workgroup
  uint32_t pr[RS32_SPAN];
  uint4 pre[RS32_SPAN / 4];  // this thread's full quads, all loads issued before the first is used
#pragma unroll
  for (int q = 0; q < RS32_SPAN / 4; ++q) {
    const uint32_t i = (uint32_t)(q * RS32_NT + threadIdx.x) * 4;
    if (i + 4 <= n) pre[q] = *reinterpret_cast<const uint4 *>(src_buf + rec_slot64(src_region, n_src_regions, r0 + i));
  }
#pragma unroll
  for (int q = 0; q < RS32_SPAN / 4; ++q) {
    const uint32_t i = (uint32_t)(q * RS32_NT + threadIdx.x) * 4;
    uint32_t rr[4] = {0, 0, 0, 0};
    if (i + 4 <= n) {  // (four records at a multiple of four never straddle a block)
      const uint4 v = pre[q];
      rr[0] = v.x, rr[1] = v.y, rr[2] = v.z, rr[3] = v.w;
      *reinterpret_cast<uint4 *>(recs + i) = v;
    } else {
      for (int r = 0; r < 4; ++r)
        if (i + r < n) recs[i + r] = rr[r] = src_buf[rec_slot64(src_region, n_src_regions, r0 + i + r)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint32_t v = 0xFFFFFFFFu;
      if (i + r < n) {
        const uint32_t sub = log_sub ? rr[r] >> rbits2 : 0u;
        v = (sub << 16) | atomicAdd(&cnt[sub], 1u);
      }
      pr[4 * q + r] = v;
    }
  }
  __syncthreads();
  // ---- exclusive scan of the counts; reservation in the final page regions ---------------------
  {
    const uint32_t per = S / RS32_NT ? S / RS32_NT : 1;
    uint32_t lo = threadIdx.x * per, sacc = 0;
    if (lo < S)
      for (uint32_t i = 0; i < per; ++i) sacc += cnt[lo + i];
    const uint32_t inc = wave_scan_incl(sacc);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const uint32_t part = (ln < wv && ln < (uint32_t)(RS32_NT / 64)) ? wsum[ln] : 0u;
    const uint32_t woff = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(part), 63);
    uint32_t run = woff + inc - sacc;
    if (lo < S)
      for (uint32_t i = 0; i < per; ++i) {
        tstart[lo + i] = run;
        run += cnt[lo + i];
      }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < S; i += RS32_NT) {
    const uint32_t c1 = cnt[i];
    gbase[i] = (c1 ? atomicAdd(&dst_cursor[dst_region_base + ((uint64_t)region << log_sub) + i], c1) : 0u) - tstart[i];
  }
  // ---- place: entry = record index inside the tile -----------------------------------------------
#pragma unroll
  for (int q = 0; q < RS32_SPAN / 4; ++q) {
    const uint32_t i = (uint32_t)(q * RS32_NT + threadIdx.x) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t v = pr[4 * q + r];
      if (v != 0xFFFFFFFFu) sorted[tstart[v >> 16] + (v & 0xFFFFu)] = (uint16_t)(i + r);
    }
  }
  __syncthreads();
  // ---- write: two entries per lane, re-read from the LDS copy of the tile ------------------------
  const uint32_t *sorted2 = reinterpret_cast<const uint32_t *>(sorted);
  const uint32_t rmask2 = rbits2 >= 32 ? 0xFFFFFFFFu : ((1u << rbits2) - 1u);
  auto spill_rec = [&](uint32_t rec) {  // the page's region is full: the whole key takes the spill path
    const uint64_t key = unmix_key(((uint64_t)(region_hi | region) << r1_bits) | rec, key_bits);
    const unsigned long long j = atomicAdd(&stats->spill_count, 1ull);
    if (j < sp.cap) {
      sp.keys[j] = key;
      sp.lanes[j] = lane;
      sp.counts[j] = 1u;
    }
  };
  for (uint32_t i = threadIdx.x; 2 * i < n; i += RS32_NT) {
    const uint32_t ee = sorted2[i];
    const bool two = 2 * i + 1 < n;
    const uint32_t ra = recs[ee & 0xFFFFu], rb = two ? recs[ee >> 16] : 0u;
    const uint32_t sa = log_sub ? ra >> rbits2 : 0u, sb = two ? (log_sub ? rb >> rbits2 : 0u) : sa;
    const uint32_t at0 = gbase[sa] + 2 * i;
    // destination regions: (this lane's first region) + page; n_dst regions interleave together
    const uint64_t pa = dst_region_base + ((uint64_t)region << log_sub) + sa, pb = dst_region_base + ((uint64_t)region << log_sub) + sb;
    const uint64_t n_dst = n_dst_total;
    if (two && sb == sa && at0 + 2 <= dst_cap && (at0 & ((1u << RB_LOG) - 1u)) != (1u << RB_LOG) - 1u) {
      const uint2 rec2 = make_uint2(ra & rmask2, rb & rmask2);
      __builtin_memcpy(dst_buf + rec_slot64(pa, n_dst, at0), &rec2, 8);
    } else {
      if (at0 < dst_cap) dst_buf[rec_slot64(pa, n_dst, at0)] = ra & rmask2;
      else spill_rec(ra);
      if (two) {
        const uint32_t at1 = gbase[sb] + 2 * i + 1;
        if (at1 < dst_cap) dst_buf[rec_slot64(pb, n_dst, at1)] = rb & rmask2;
        else spill_rec(rb);
      }
    }
  }
}

// In LDS a page is 8192 keys (64 KiB) + 8192 sixteen-bit DELTAS of this pass (16 KiB, two per
// word) = 80 KiB, so two page workgroups share a CU.  The deltas are folded into the page's
// 32-bit counts in HBM when the workgroup leaves (saturating, counting.rs:82-85) — and earlier
// for any slot whose delta reaches 2^15, which a sweep checks every SWEEP_EVERY steps: between two
// sweeps a slot gains at most SWEEP_EVERY·4·PG_WG = 24576 on top of < 2^15, so a delta never wraps.
__device__ __forceinline__ void delta_add(uint32_t *dl, uint32_t slot) {
  atomicAdd(&dl[slot >> 1], 1u << (16 * (slot & 1)));
}

// Workgroup reduction through a few words of scratch LDS (sum), result to every thread.
__device__ __forceinline__ uint32_t pg_wg_sum(uint32_t v, uint32_t *scratch) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t s = 0;
  for (int w = 0; w < PG_WG / 64; ++w) s += scratch[w];
  __syncthreads();
  return s;
}

// LDS budget: 64 KiB keys + 16 KiB deltas = exactly half a CU's 160 KiB, so there is no room
// for even one shared counter: occupancy and new-key counts are reduced through the (then
// idle) delta words, and every wave keeps its own miss queue (a slice of the page's miss_buf
// region) with the fill level in a wave-uniform register — ballot/popcount, no atomics.
constexpr uint32_t MISS_SLACK = 4096;  // extra records per page region of miss_buf (8 slices)
#ifndef SHK_PG_DRAIN_EVERY
#define SHK_PG_DRAIN_EVERY 4
#endif
constexpr uint32_t PG_DRAIN_EVERY = SHK_PG_DRAIN_EVERY; // k_pages: steps (of 4 records per thread) between miss-queue drains
// A wave's queue is emptied at every drain, so it never holds more than the records the wave saw
// since the last one: a page's slice of miss_buf is capped at that, however large its region is.
constexpr uint32_t MISS_WAVE_MAX = PG_DRAIN_EVERY * 4 * 64;
constexpr uint32_t MISS_PAGE_MAX = (PG_WG / 64) * MISS_WAVE_MAX;
// FK / FV: the table holds nothing yet and its memory has not been cleared (the first page pass after a reset, over
// every page and lane: k_pages32's FRESH, for 8-byte records) — FK: the page's keys are not read (they start EMPTY)
// and are written out whole at the end; FV: this lane's counts of the page are zeroed here, first, instead of by a
// fill of the whole table that the pass would then read back.  Lane 0's launch of the pass is <true, true>, the other
// lanes' <false, true>.
template <bool FK, bool FV>
__global__ void __launch_bounds__(PG_WG) k_pages(TableRef tb, uint32_t lane_lo, uint32_t lane_hi, uint32_t lane_stride,
                                                 const unsigned int *__restrict__ cursor, uint32_t cap_p,
                                                 const uint64_t *__restrict__ part_buf,
                                                 uint64_t *__restrict__ miss_buf,
                                                 DevStats *__restrict__ stats, SpillRef sp, uint32_t page0 = 0) {
  // Chunk lanes [lane_lo, lane_hi) one after the other on the same LDS copy of the page's keys (as k_pages32): lane l's
  // records are region l · lane_stride + page of part_buf, its fill level cursor[l · lane_stride + page] (lane_stride
  // = 0: the arrays are one lane's).  The keys are read once and written once however many lanes there are — a launch
  // per lane read and wrote them per lane (ten lanes at k = 31: 215 GB through the page passes of 12.5 M reads, 49 ms).
  __shared__ __attribute__((aligned(16))) uint64_t keys[PAGE_SLOTS];
  __shared__ __attribute__((aligned(16))) uint32_t dl[PAGE_SLOTS / 2];
  if (stats->bad != ~0ull) return;
  const uint32_t page = blockIdx.x + page0;  // (page0: a launch over a range of pages — grouped flush, see flush_acc)
  uint64_t *gk = tb.keys + ((uint64_t)page << PAGE_LOG);
  if (!FK && !FV) {
    uint32_t any = 0;
    for (uint32_t l = lane_lo; l < lane_hi; ++l) any |= cursor[(uint64_t)l * lane_stride + page];
    if (any == 0) return;  // nothing for this page in any lane: leave it untouched in HBM
  }
  // page keys → LDS (16-B vectors), counting occupied slots on the way
  uint32_t my_occ = 0;
  for (uint32_t i = threadIdx.x; i < PAGE_SLOTS / 2; i += PG_WG) {
    ulonglong2 v;
    if (FK) v.x = v.y = EMPTY;
    else v = reinterpret_cast<const ulonglong2 *>(gk)[i];
    reinterpret_cast<ulonglong2 *>(keys)[i] = v;
    my_occ += (v.x != EMPTY) + (v.y != EMPTY);
  }
  const uint32_t occ0 = pg_wg_sum(my_occ, dl);
  // every wave may add its share of what is left below the fill cap; beyond it new keys spill
  const uint32_t room = occ0 < PAGE_FILL_CAP ? (PAGE_FILL_CAP - occ0) / (PG_WG / 64) : 0u;
  uint32_t n_new = 0;   // per thread, over all lanes
  uint32_t lane = lane_lo;
  for (; lane < lane_hi; ++lane) {
  const uint32_t filled0 = cursor[(uint64_t)lane * lane_stride + page];
  const uint32_t filled = filled0 < cap_p ? filled0 : cap_p;  // beyond cap_p: spilled
  uint32_t *gv = tb.vals + (uint64_t)lane * tb.cap + ((uint64_t)page << PAGE_LOG);
  if (FV) {
    for (uint32_t j = threadIdx.x; j < PAGE_SLOTS / 4; j += PG_WG) reinterpret_cast<uint4 *>(gv)[j] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (filled == 0) continue;  // (uniform) nothing of this lane for the page
  for (uint32_t i = threadIdx.x; i < PAGE_SLOTS / 2; i += PG_WG) dl[i] = 0;
  __syncthreads();
  const uint64_t n = filled;
  const uint64_t *src = part_buf + ((uint64_t)lane * lane_stride + page) * cap_p;
  const uint32_t wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
  const uint32_t slice_n = (uint32_t)(n / (PG_WG / 64)) + MISS_SLACK / (PG_WG / 64);
  const uint32_t slice = slice_n < MISS_WAVE_MAX ? slice_n : MISS_WAVE_MAX;
  const uint32_t mq_stride = cap_p + MISS_SLACK < MISS_PAGE_MAX ? cap_p + MISS_SLACK : MISS_PAGE_MAX;
  uint64_t *mq = miss_buf + (uint64_t)page * mq_stride + (uint64_t)wave * slice;
  uint32_t n_miss = 0;  // wave-uniform
  // The fill cap is soft: whether this wave may still insert new keys is decided once per
  // drain from its running total, so a wave can overshoot its share by one drain's misses; a
  // page that fills up completely still ends in the spill path (bounded probe).
  bool may_insert = room > 0;
  auto update_may_insert = [&]() {
    uint32_t w = n_new;
    for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
    may_insert = w < room;
  };
  // general probe of one record into the LDS page: find the key or insert it, add one.  Bucket by bucket (two
  // 16-B LDS reads bring a bucket's four keys; the first of them, in probe order, that is this key or EMPTY ends
  // the search) instead of one LDS round trip per slot — see k_pages32.
  auto spill = [&](uint64_t key) {
    unsigned long long i = atomicAdd(&stats->spill_count, 1ull);
    if (i < sp.cap) {
      sp.keys[i] = key;
      sp.lanes[i] = lane;
      sp.counts[i] = 1u;
    }
  };
  auto insert = [&](uint64_t key) {
    if (key == EMPTY) return;  // padding record of an odd (tile, page) run
    const uint32_t b0 = slot_of(hash64(key, tb.key_bits), tb.log_pages + tb.owner_bits) >> 2;
    for (uint32_t nb = 0; nb < PAGE_SLOTS / 4;) {
      const uint32_t s0 = ((b0 + nb) & (PAGE_SLOTS / 4 - 1)) << 2;
      const ulonglong2 ka = *reinterpret_cast<const ulonglong2 *>(&keys[s0]);
      const ulonglong2 kb = *reinterpret_cast<const ulonglong2 *>(&keys[s0 + 2]);
      const uint32_t e0 = (ka.x == key) | (ka.x == EMPTY), e1 = (ka.y == key) | (ka.y == EMPTY),
                     e2 = (kb.x == key) | (kb.x == EMPTY), e3 = (kb.y == key) | (kb.y == EMPTY);
      if (!(e0 | e1 | e2 | e3)) {
        ++nb;
        continue;
      }
      const uint32_t q = e0 ? 0u : e1 ? 1u : e2 ? 2u : 3u;
      const uint64_t cur = e0 ? ka.x : e1 ? ka.y : e2 ? kb.x : kb.y;
      if (cur == key) {
        delta_add(dl, s0 + q);
        return;
      }
      if (!may_insert) break;  // this wave's share of the page is used up → spill
      const uint64_t prev = atomicCAS((unsigned long long *)&keys[s0 + q], (unsigned long long)EMPTY,
                                      (unsigned long long)key);
      if (prev == EMPTY) {
        n_new++;
        delta_add(dl, s0 + q);
        return;
      }
      // somebody else took the slot: look at the bucket again (it may be this very key)
    }
    spill(key);
  };
  // Four records per thread per step.  The global loads of the NEXT step, then the four 32-B
  // home buckets (two ds_read_b128 each), are in flight together; a record whose key sits in
  // its home bucket (≈99 % at load ≤ 1/2) costs one non-returning LDS add.  The others (first
  // occurrences, displaced keys) are only QUEUED here and handled densely by the general probe
  // afterwards: a divergent in-line slow path would be executed by nearly every wave for one or
  // two lanes each.  Regions are sequences of aligned record PAIRS (padding = EMPTY): one 16-B
  // load per lane fetches two records; two such loads per step.
  const ulonglong2 *src2 = reinterpret_cast<const ulonglong2 *>(src);
  const uint32_t n_quads = (uint32_t)(n / (4 * PG_WG));  // steps of four records per thread
  constexpr uint32_t DRAIN_EVERY = PG_DRAIN_EVERY;        // steps between miss-queue drains
  // 16-bit deltas: a sweep moves every delta ≥ 2^15 to the 32-bit counts in HBM, so a slot starts an
  // interval below 2^15 and gains at most SWEEP_EVERY·4·PG_WG in it (+ the tail of < 4·PG_WG records)
  constexpr uint32_t SWEEP_EVERY = 12;
  static_assert(0x7FFF + SWEEP_EVERY * 4 * PG_WG + 4 * PG_WG <= 0xFFFF, "a 16-bit delta must not wrap between sweeps");
  ulonglong2 nxt[2];
  if (n_quads) {
    nxt[0] = src2[threadIdx.x];
    nxt[1] = src2[threadIdx.x + PG_WG];
  }
  for (uint32_t quad = 0; quad < n_quads; ++quad) {
    uint64_t kk[4] = {nxt[0].x, nxt[0].y, nxt[1].x, nxt[1].y};
    uint32_t ss[4];
    ulonglong2 ba[4], bb[4];
    if (quad + 1 < n_quads) {  // next step's loads are in flight while this one is processed
      const uint64_t ib = (uint64_t)(quad + 1) * 2 * PG_WG + threadIdx.x;
      nxt[0] = src2[ib];
      nxt[1] = src2[ib + PG_WG];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) ss[q] = slot_of(hash64(kk[q], tb.key_bits), tb.log_pages + tb.owner_bits);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ba[q] = *reinterpret_cast<const ulonglong2 *>(&keys[ss[q]]);
      bb[q] = *reinterpret_cast<const ulonglong2 *>(&keys[ss[q] + 2]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint64_t kq = kk[q];
      // straight-line (a ?: chain comes out as nested exec-mask branches): at most one of the four
      // keys can match, so the matching index is a sum of the compare bits; a record that missed —
      // or a padding record — adds 0 to the bucket's first delta instead of skipping the add
      const uint32_t e0 = ba[q].x == kq, e1 = ba[q].y == kq, e2 = bb[q].x == kq, e3 = bb[q].y == kq;
      const uint32_t idx = e1 + 2u * e2 + 3u * e3;
      const uint32_t found = e0 | e1 | e2 | e3;       // (EMPTY = padding "finds" a free slot)
      const uint32_t slot = ss[q] + idx;
      atomicAdd(&dl[slot >> 1], (found & (uint32_t)(kq != EMPTY)) << (16u * (slot & 1u)));
      const bool missed = !found;
      const unsigned long long mm = __ballot(missed);
      if (missed) mq[n_miss + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u))] = kq;
      n_miss += (uint32_t)__popcll(mm);
    }
    // drain after each of the first DRAIN_EVERY steps (an empty page misses on every first
    // occurrence, and on its repeats until it is inserted), then every DRAIN_EVERY steps
    // (no barrier: the queue is the wave's own, and a key another wave is inserting right now is at
    // worst missed once more and found by the general probe)
    if (quad < DRAIN_EVERY || (quad % DRAIN_EVERY) == DRAIN_EVERY - 1 || quad + 1 == n_quads) {
      for (uint32_t j = lane_id; j < n_miss; j += 64) insert(mq[j]);  // this wave's own queue
      n_miss = 0;
      update_may_insert();
    }
    if ((quad % SWEEP_EVERY) == SWEEP_EVERY - 1) {
      __syncthreads();
      // deltas >= 2^15 go to the 32-bit counts in HBM now (a slot gains < 2^15 between checks)
      for (uint32_t j = threadIdx.x; j < PAGE_SLOTS / 2; j += PG_WG) {
        const uint32_t w = dl[j];
        if (w & 0x80008000u) {
          if (w & 0x8000u) gv[2 * j] = sat_add_u32(gv[2 * j], w & 0xFFFFu);
          if (w & 0x80000000u) gv[2 * j + 1] = sat_add_u32(gv[2 * j + 1], w >> 16);
          dl[j] = (w & 0x8000u ? 0u : (w & 0xFFFFu)) | (w & 0x80000000u ? 0u : (w & 0xFFFF0000u));
        }
      }
      __syncthreads();
    }
  }
  // tail (< 4*PG_WG records): straight through the general probe
  for (uint64_t i = (uint64_t)n_quads * 4 * PG_WG + threadIdx.x; i < n; i += PG_WG) insert(src[i]);
  __syncthreads();
  // LDS → page: counts += deltas (saturating), four slots per lane; then the keys if any is new
  for (uint32_t j = threadIdx.x; j < PAGE_SLOTS / 4; j += PG_WG) {
    const uint2 d = reinterpret_cast<const uint2 *>(dl)[j];
    if (d.x | d.y) {
      uint4 v = reinterpret_cast<const uint4 *>(gv)[j];
      v.x = sat_add_u32(v.x, d.x & 0xFFFFu);
      v.y = sat_add_u32(v.y, d.x >> 16);
      v.z = sat_add_u32(v.z, d.y & 0xFFFFu);
      v.w = sat_add_u32(v.w, d.y >> 16);
      reinterpret_cast<uint4 *>(gv)[j] = v;
    }
  }
  __syncthreads();  // (the next lane clears the deltas)
  }  // lanes
  const uint32_t nnew = pg_wg_sum(n_new, dl);
  if (nnew || FK) {
    for (uint32_t j = threadIdx.x; j < PAGE_SLOTS / 2; j += PG_WG)
      reinterpret_cast<ulonglong2 *>(gk)[j] = reinterpret_cast<const ulonglong2 *>(keys)[j];
    if (threadIdx.x == 0 && nnew) atomicAdd(&stats->n_distinct, (unsigned long long)nnew);
  }
}

// ------------------------------------------------------------------------------------------
// k_pages32: the page workgroup for 4-byte records (k_part_scatter_sorted<.., true>).  A record
// is rec = mix_key(key) mod 2^R, R = 2k - log_pages ≤ 32: home bucket = its top 11 bits,
// fingerprint fp = the R-11 bits below.  The page sits in LDS as 32-bit TAGS instead of keys:
//     tag(slot) = fp(key) << 4 | d ,  d = (bucket(slot) - home bucket(key)) mod 2048, d ≤ 14
// which — mix_key being a bijection and the page known — identifies the key.  A record hits when
// one of the four tags of its home bucket (ONE ds_read_b128) equals fp << 4; no hash is computed
// here at all.  Everything else (first occurrences, displaced keys) goes through a per-wave miss
// queue IN LDS (the tags leave room for it) and the general probe, which compares (fp, d) slot
// by slot, inserts with a CAS on the tag and then writes the rebuilt key (unmix_key) to the
// page in HBM.  What a tag cannot express — a key ≥ 15 buckets from home (d = 15, only ever created
// by the direct path) met on a probe, or a probe that would have to insert that far out —
// sends the record to the spill list, i.e. through the exact global-memory path.
// Counts: 16-bit deltas as in k_pages.  Keys are written when inserted, so the page's keys are
// never written back wholesale.
// ------------------------------------------------------------------------------------------
// A workgroup barrier that orders LDS traffic only: global stores still in flight are not waited for.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr uint32_t TAG_EMPTY = 0xFFFFFFFFu;
// A tag's low TAG_DB bits: how many buckets past its home bucket the key sits (0 … TAG_FAR - 1; TAG_FAR = all ones:
// farther than a tag can say — only ever created by the direct path — or, with all other bits set too, EMPTY).
// Four bits since round 2 (the fingerprint has at most 21): with three, ONE k-mer of a 3 M-k-mer input that came to
// lie 7 buckets from home sent its ≈1000 records through the spill path in every job.
constexpr uint32_t TAG_DB = 4, TAG_FAR = (1u << TAG_DB) - 1u;
constexpr uint32_t MQ32 = 512;              // miss-queue entries per wave (LDS)
constexpr uint32_t P32_RPS = 4;             // records per thread per step (one 16-B load)
constexpr uint32_t P32_EARLY = 4;           // steps after which the miss queue is drained regardless of its fill
static_assert(PAGE_SLOTS * 4 + PAGE_SLOTS * 4 + (PG_WG / 64) * MQ32 * 4 <= 81920, "two page workgroups per CU");
static_assert(MQ32 >= 2 * 64 * P32_RPS, "the queue must take a whole step of misses on top of the drain threshold");
// Chunk lanes [lane_lo, lane_hi) are processed one after the other on the same LDS copy of the
// page (tags stay, the per-pass counts are written back and cleared in between); lane l's records
// are region l·lane_stride + page of the n_regions block-interleaved regions (lane_stride = 0: the
// buffer holds one lane, region = page).
//
// FRESH: the table holds nothing yet and its memory has not even been cleared (shk_reset leaves that to the first
// page pass when that pass covers every page and every lane).  The page is then not READ at all — tags start
// empty, counts start at zero — and written out whole at the end: every lane's counts, and the keys rebuilt from
// the tags (a tag and the page identify the key), EMPTY where no tag is.  What a reset would have written is
// written once, with the result in it, and a page in costs nothing.
//
// HIST (with FRESH): the page's share of EVERY histogram column and of the totals on the way out.  A fresh pass has
// each slot's counts of every chunk lane in its registers as it writes them — what k_histo would read back a moment
// later (4 B per slot and lane: a fifth of a ten-lane job on a 30 Mb genome).  Column l is the histogram of the
// clamped prefix sums over lanes 0..l (k_histo's comment; counting.rs:171-202, histogram.rs:51-85): a thread keeps
// the running sum of its sixteen slots across the lane loop, bins below FH_BINS go through LDS histograms (every
// wave's own, then idle, miss queue) into this page's row of `partial` — plain stores, k_hist_reduce sums the rows —
// and the few beyond straight into `hist`.  The host uses the result only if nothing else touches the table before
// the histogram is asked for (shk_ctx::fused_valid); otherwise k_histo runs as before.
constexpr uint32_t FH_BINS = 512;
struct FusedHist {
  uint32_t *partial;             // [page][col][FH_BINS]
  unsigned long long *ptot;      // [page][4]: occupied slots, Σ clamped sums, Σ lane counts, any saturated
  unsigned long long *hist;      // the context's histogram (bins ≥ FH_BINS)
  unsigned long long histo_max;
  uint32_t n_cols;
  uint32_t pad;
};
// (HIST = 2: the launch covers ONE chunk lane — no running sums to carry across a lane loop: the fold's sixteen
// registers and what they cost the record loop's code are not there)
template <bool FRESH, int HIST = 0>
__global__ void __launch_bounds__(PG_WG, 4) k_pages32(TableRef tb, uint32_t lane_lo, uint32_t lane_hi,
                                                      uint32_t lane_stride, uint32_t n_regions_,
                                                      const unsigned int *__restrict__ cursor, uint32_t cap_p,
                                                      const uint32_t *__restrict__ part_buf,
                                                      DevStats *__restrict__ stats, SpillRef sp, uint32_t page0 = 0,
                                                      FusedHist fh = FusedHist{}) {
  static_assert(!HIST || FRESH, "the fused histogram is a fresh pass's");
  static_assert(FH_BINS == MQ32, "a wave bins into its own miss queue");
  __shared__ __attribute__((aligned(16))) uint32_t tags[PAGE_SLOTS];
  __shared__ __attribute__((aligned(16))) uint32_t dl[PAGE_SLOTS];  // this pass's count per slot
  __shared__ __attribute__((aligned(16))) uint32_t mqs[(PG_WG / 64) * MQ32];
  if (stats->bad != ~0ull) return;
  const uint32_t page = blockIdx.x + page0;
  if (!FRESH) {
    uint32_t any = 0;
    for (uint32_t l = lane_lo; l < lane_hi; ++l) any |= cursor[l * lane_stride + page];
    if (any == 0) return;  // nothing for this page in any lane: leave it untouched in HBM
  }
  uint32_t lane = lane_lo;
  const uint32_t bits = tb.key_bits, R = bits - (tb.log_pages + tb.owner_bits), fpb = R - 11;
  const uint32_t fpmask = (1u << fpb) - 1u;
  const uint64_t gpage = ((uint64_t)tb.owner_id << tb.log_pages) | page;  // page index in the virtual global table (owner share)
  uint64_t *gk = tb.keys + ((uint64_t)page << PAGE_LOG);
  uint32_t *gv = nullptr;  // this lane's counts of the page (set per lane below)
  // page keys → tags, counting occupied slots on the way
  uint32_t my_occ = 0;
  constexpr int LD = PAGE_SLOTS / 2 / PG_WG;  // 16-B loads per thread: all issued before the first is used
  if (FRESH) {
#pragma unroll
    for (int u = 0; u < LD; ++u) reinterpret_cast<uint2 *>(tags)[threadIdx.x + u * PG_WG] = make_uint2(TAG_EMPTY, TAG_EMPTY);
  } else {
  ulonglong2 pv[LD];
#pragma unroll
  for (int u = 0; u < LD; ++u) pv[u] = reinterpret_cast<const ulonglong2 *>(gk)[threadIdx.x + u * PG_WG];
#pragma unroll
  for (int u = 0; u < LD; ++u) {
    const uint32_t i = threadIdx.x + u * PG_WG;
    const ulonglong2 v = pv[u];
    const uint64_t kv[2] = {v.x, v.y};
    uint32_t tg[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      tg[q] = TAG_EMPTY;
      if (kv[q] != EMPTY) {
        const uint32_t rec = (uint32_t)mix_key(kv[q], bits) & (uint32_t)(0xFFFFFFFFull >> (32 - R));
        const uint32_t d = (((2 * i + q) >> 2) - (rec >> fpb)) & (PAGE_SLOTS / 4 - 1);
        tg[q] = ((rec & fpmask) << TAG_DB) | (d < TAG_FAR ? d : TAG_FAR);
        my_occ++;
      }
    }
    reinterpret_cast<uint2 *>(tags)[i] = make_uint2(tg[0], tg[1]);
  }
  }
  const uint32_t occ0 = FRESH ? 0u : pg_wg_sum(my_occ, dl);
  const uint32_t room = occ0 < PAGE_FILL_CAP ? (PAGE_FILL_CAP - occ0) / (PG_WG / 64) : 0u;
  for (uint32_t i = threadIdx.x; i < PAGE_SLOTS; i += PG_WG) dl[i] = 0;
  __syncthreads();
  uint32_t n = 0;        // records of the current lane's region (set per lane below)
  uint64_t region = 0;   // … and its index among the block-interleaved regions (rec_slot)
  const uint64_t n_regions = n_regions_;
  const uint32_t wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
  uint32_t *mq = mqs + wave * MQ32;
  uint32_t n_miss = 0;  // wave-uniform
  uint32_t n_new = 0;   // per thread
  bool may_insert = room > 0;
  auto update_may_insert = [&]() {
    uint32_t w = n_new;
    for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
    may_insert = w < room;
  };
  auto spill = [&](uint32_t rec) {
    const uint64_t key = unmix_key((gpage << R) | rec, bits);
    unsigned long long i = atomicAdd(&stats->spill_count, 1ull);
    if (i < sp.cap) {
      sp.keys[i] = key;
      sp.lanes[i] = lane;
      sp.counts[i] = 1u;
    }
  };
  // general probe of one record: find its (fp, d) tag or insert it, add one.  Bucket by bucket — ONE 16-B LDS read
  // brings the four tags of a bucket, which are then looked at in probe order in registers (a tag is written once,
  // from EMPTY: what was read as occupied stays what it is; what was read as EMPTY is claimed with a CAS, whose
  // answer is the slot's tag if somebody else was faster).  Slot by slot this was a chain of LDS round trips per
  // record: most of a many-lane page pass (10 lanes, 30 Mb genome: 10 k cycles per lane and page in the drains).
  auto insert = [&](uint32_t rec) {
    const uint32_t home = rec >> fpb, fp3 = (rec & fpmask) << TAG_DB;
    for (uint32_t d = 0; d < TAG_FAR;) {
      const uint32_t sl0 = ((home + d) & (PAGE_SLOTS / 4 - 1)) << 2, want = fp3 | d;
      const uint4 t4 = *reinterpret_cast<const uint4 *>(&tags[sl0]);
      // the first slot of the bucket, in probe order, that ends the search: this key's tag, an EMPTY slot, or a far
      // entry (low bits 7 — EMPTY has them too); one decision per bucket, the same code for every lane
      const uint32_t s0 = (t4.x == want) | ((t4.x & TAG_FAR) == TAG_FAR), s1 = (t4.y == want) | ((t4.y & TAG_FAR) == TAG_FAR),
                     s2 = (t4.z == want) | ((t4.z & TAG_FAR) == TAG_FAR), s3 = (t4.w == want) | ((t4.w & TAG_FAR) == TAG_FAR);
      if (!(s0 | s1 | s2 | s3)) {
        ++d;
        continue;
      }
      const uint32_t q = s0 ? 0u : s1 ? 1u : s2 ? 2u : 3u;
      const uint32_t cur = s0 ? t4.x : s1 ? t4.y : s2 ? t4.z : t4.w;
      if (cur == want) {
        atomicAdd(&dl[sl0 + q], 1u);
        return;
      }
      if (cur != TAG_EMPTY || !may_insert) break;  // a far entry (cannot tell whether it is this key), or this wave's share of the page's room is used up → spill
      const uint32_t prev = atomicCAS(&tags[sl0 + q], TAG_EMPTY, want);
      if (prev == TAG_EMPTY) {
        n_new++;
        if (!FRESH) gk[sl0 + q] = unmix_key((gpage << R) | rec, bits);
        atomicAdd(&dl[sl0 + q], 1u);
        return;
      }
      // somebody else took the slot: look at the bucket again (it may be this very key)
    }
    spill(rec);  // (or too far out for a tag)
  };
  auto drain = [&]() {
    for (uint32_t j = lane_id; j < n_miss; j += 64) insert(mq[j]);  // this wave's own queue
    n_miss = 0;
    update_may_insert();
  };
  // HIST: the running (clamped) sums of this thread's sixteen slots over the lanes so far, and its share of the totals
  constexpr int WBH = PAGE_SLOTS / 4 / PG_WG;
  uint32_t cum[HIST == 1 ? WBH * 4 : 1];
  uint32_t one_unique = 0, one_sat = 0;  // HIST = 2: the totals as the (only) lane's counts go by
  unsigned long long fh_lane = 0;
  uint32_t n_high = 0;  // sums this thread sent past the LDS bins (same-line global adds: a job full of them should leave the histogram to k_histo — the host looks)
  if (HIST == 1) {
#pragma unroll
    for (int i = 0; i < WBH * 4; ++i) cum[i] = 0;
  }
  // one lane's counts d (this thread's quads) folded into the sums; column `lane` of the page → partial.  Called by
  // every thread of the workgroup, the miss queues idle (between a lane's drain and the next lane's first record).
  auto fold_hist = [&](const uint4 (&d)[WBH]) {
    if (!HIST) return;
    // every wave bins into its OWN miss queue (MQ32 = FH_BINS words, idle since the wave's last drain): nothing to
    // agree on before, one barrier before the rows are summed, one before the queues are queues again
    const bool col = lane < fh.n_cols;
    if (col) {
      for (uint32_t i = lane_id; i < FH_BINS; i += 64) mq[i] = 0;
    }
    const unsigned long long hlen = fh.histo_max + 2;
    // bin of a sum cc: min(cc, histo_max + 1) — in 32 bits (histo_max ≥ 2^32 - 1: every sum is its own bin)
    const uint32_t top32 = fh.histo_max >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)fh.histo_max + 1u;
    uint32_t lsum = 0, lcarry = 0;  // Σ of this lane's sixteen counts
#pragma unroll
    for (int u = 0; u < WBH; ++u) {
      const uint32_t v[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t s1 = lsum + v[q];
        lcarry += s1 < lsum;
        lsum = s1;
        uint32_t cc;
        if (HIST == 1) {
          cc = sat_add_u32(cum[u * 4 + q], v[q]);
          cum[u * 4 + q] = cc;
        } else {
          cc = v[q];
          one_unique += cc != 0;
          one_sat |= cc == 0xFFFFFFFFu;
        }
        const uint32_t bin = cc < top32 ? cc : top32;
        if (col && cc) {
          if (bin < FH_BINS) atomicAdd(&mq[bin], 1u);
          else atomicAdd(&fh.hist[(unsigned long long)lane * hlen + bin], 1ull), ++n_high;
        }
      }
    }
    fh_lane += ((unsigned long long)lcarry << 32) | lsum;
    if (col) {
      lds_barrier();
      uint32_t *row = fh.partial + ((size_t)page * fh.n_cols + lane) * FH_BINS;
      for (uint32_t i = threadIdx.x; i < FH_BINS; i += PG_WG) {
        uint32_t sum = 0;
#pragma unroll
        for (int w = 0; w < PG_WG / 64; ++w) sum += mqs[w * MQ32 + i];
        row[i] = sum;
      }
      lds_barrier();  // (the next lane's misses go where the bins were)
    }
  };
  for (lane = lane_lo; lane < lane_hi; ++lane) {
    {
      const uint32_t filled = cursor[lane * lane_stride + page];
      n = filled < cap_p ? filled : cap_p;  // beyond cap_p: spilled
    }
    gv = tb.vals + (uint64_t)lane * tb.cap + ((uint64_t)page << PAGE_LOG);
    if (n == 0) {  // (uniform across the workgroup)
      if (FRESH) {
#pragma unroll
        for (int u = 0; u < PAGE_SLOTS / 4 / PG_WG; ++u)
          reinterpret_cast<uint4 *>(gv)[threadIdx.x + u * PG_WG] = make_uint4(0u, 0u, 0u, 0u);
      }
      if (HIST) {  // (the column still counts every slot whose sum so far is not zero)
        uint4 z[WBH];
#pragma unroll
        for (int u = 0; u < WBH; ++u) z[u] = make_uint4(0u, 0u, 0u, 0u);
        fold_hist(z);
      }
      continue;
    }
    region = (uint64_t)lane * lane_stride + page;
    // this lane's counts of the page: asked for NOW, needed when the lane's records have been counted — the
    // HBM round trip hides behind the record loop instead of standing between two lanes.  (A deeper pipeline —
    // every lane's fill level asked for up front, the next lane's first records in flight over this lane's drain
    // and write-out — was measured twice: ten lanes on a 30 Mb genome 3.04 → 2.91 ms before the drains were
    // made cheap, 1.81 → 1.86 ms after.  Not kept.)
    constexpr int WB = PAGE_SLOTS / 4 / PG_WG;
    uint4 gvv[WB];
    const bool prefetch = !FRESH && lane_hi - lane_lo > 1;  // (one lane: nothing stands between two lanes; the quads that did not change are then not read at all)
    if (prefetch) {
#pragma unroll
      for (int u = 0; u < WB; ++u) gvv[u] = reinterpret_cast<const uint4 *>(gv)[threadIdx.x + u * PG_WG];
    }
    // Main loop: one 16-B load = four records per thread per step, the next step's load in flight.
    // No barrier in here: queues are per wave, and a slot's count of this pass is a full 32-bit word
    // (a page sees < 2^31 records), so nothing has to be folded away mid-pass.
    const uint32_t n_steps = n / (P32_RPS * PG_WG);
    // (a region of at most two steps — a chunk lane's share of a page on a many-lane table — fits the miss queue
    // whole: one drain at the end instead of three, each of which is a chain of LDS round trips)
    const bool early = n > 2 * P32_RPS * PG_WG;
    auto load4 = [&](uint32_t step) {  // this lane's four records of a step (they share a block)
      return *reinterpret_cast<const uint4 *>(part_buf + rec_slot64(region, n_regions, (step * PG_WG + threadIdx.x) * 4u));
    };
    // `nv` = how many of the lane's four records exist (4 in every step but a page's last, partial one)
    auto body = [&](uint32_t step, const uint4 &cur, uint32_t nv, auto partial) {
      const uint32_t rr[4] = {cur.x, cur.y, cur.z, cur.w};
      uint4 bk[4];
  #pragma unroll
      for (int q = 0; q < 4; ++q) bk[q] = *reinterpret_cast<const uint4 *>(&tags[(rr[q] >> fpb) << 2]);
      // Straight-line code on purpose (no ?: chains, which come out as nested exec-mask branches):
      // at most one of the four tags can match, so the matching index is a sum of the compare bits,
      // and a record that missed adds 0 to its bucket's first slot instead of skipping the add.
      bool missed[4];
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t want = (rr[q] & fpmask) << TAG_DB;
        const uint32_t e0 = bk[q].x == want, e1 = bk[q].y == want, e2 = bk[q].z == want, e3 = bk[q].w == want;
        const uint32_t idx = e1 + 2u * e2 + 3u * e3;
        uint32_t found = e0 | e1 | e2 | e3;
        bool exists = true;
        if (decltype(partial)::value) {
          exists = (uint32_t)q < nv;
          found &= (uint32_t)exists;
        }
        // (a lane past the end of the region adds its 0 to a slot of its own: left at record 0's bucket, the up
        // to 2047 of them in a page's last step queue up on ONE LDS address, and same-address atomics go one
        // by one — it shows on many-lane tables, where every lane of every page has such a step)
        uint32_t at = ((rr[q] >> fpb) << 2) + idx;
        if (decltype(partial)::value) at = exists ? at : threadIdx.x * 4u + (uint32_t)q;
        atomicAdd(&dl[at], found);
        missed[q] = exists && !found;
      }
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned long long mm = __ballot(missed[q]);
        // (lanes below this one that missed: v_mbcnt_lo/hi take the mask as it is)
        if (missed[q]) mq[n_miss + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u))] = rr[q];
        n_miss += (uint32_t)__popcll(mm);
      }
      // drain when the next step might not fit (worst case: every record of it misses), and after
      // each of the first few steps: an empty page misses on every first occurrence and on its
      // repeats until it is inserted (measured: 4 early drains 0.232 ms, 0 → 0.240, 16 → 0.257)
      if (n_miss > MQ32 - 64 * P32_RPS || (step < P32_EARLY && early)) drain();
    };
    uint4 nxt;
    if (n_steps) nxt = load4(0);
    // the last, partial step's records are asked for now as well (a short region is two loads: both in flight)
    const uint32_t tail_j = (n_steps * PG_WG + threadIdx.x) * 4u;
    const uint32_t tail_nv = tail_j < n ? (n - tail_j < 4u ? n - tail_j : 4u) : 0u;
    uint4 tailv = make_uint4(0u, 0u, 0u, 0u);
    if (tail_nv) tailv = load4(n_steps);  // (a quad never straddles a block; its tail past n is ignored)
    for (uint32_t step = 0; step < n_steps; ++step) {
      const uint4 cur = nxt;
      if (step + 1 < n_steps) nxt = load4(step + 1);
      body(step, cur, 4u, std::false_type{});
    }
    if (n_steps * P32_RPS * PG_WG < n)  // the page's last, partial step: lanes past the end sit it out
      body(n_steps, tailv, tail_nv, std::true_type{});
    drain();
    lds_barrier();
    // this pass's counts → the page's counts (saturating), four slots per lane and step; all of a
    // thread's loads are issued before the first is used, and the LDS counts go back to zero on the
    // way (the next chunk lane counts from zero)
    {
      uint4 d[WB];
#pragma unroll
      for (int u = 0; u < WB; ++u) d[u] = reinterpret_cast<const uint4 *>(dl)[threadIdx.x + u * PG_WG];
      if (FRESH) {  // counts start at zero: every quad is written, none is read
#pragma unroll
        for (int u = 0; u < WB; ++u) {
          reinterpret_cast<uint4 *>(gv)[threadIdx.x + u * PG_WG] = d[u];
          if (d[u].x | d[u].y | d[u].z | d[u].w)
            reinterpret_cast<uint4 *>(dl)[threadIdx.x + u * PG_WG] = make_uint4(0u, 0u, 0u, 0u);
        }
        fold_hist(d);
      }
      if (!FRESH && !prefetch) {
#pragma unroll
        for (int u = 0; u < WB; ++u)
          if (d[u].x | d[u].y | d[u].z | d[u].w) gvv[u] = reinterpret_cast<const uint4 *>(gv)[threadIdx.x + u * PG_WG];
      }
#pragma unroll
      for (int u = 0; u < WB; ++u)
        if (!FRESH && (d[u].x | d[u].y | d[u].z | d[u].w)) {
          uint4 v = gvv[u];
          v.x = sat_add_u32(v.x, d[u].x);
          v.y = sat_add_u32(v.y, d[u].y);
          v.z = sat_add_u32(v.z, d[u].z);
          v.w = sat_add_u32(v.w, d[u].w);
          reinterpret_cast<uint4 *>(gv)[threadIdx.x + u * PG_WG] = v;
          reinterpret_cast<uint4 *>(dl)[threadIdx.x + u * PG_WG] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    lds_barrier();  // (the counts on their way to HBM are nobody's business in here: the next lane's are elsewhere)
  }
  __syncthreads();
  if (FRESH) {  // the page's keys, rebuilt from the tags: home bucket = bucket - d, fingerprint = the tag's upper bits
#pragma unroll
    for (int u = 0; u < LD; ++u) {
      const uint32_t i = threadIdx.x + u * PG_WG;
      const uint2 tg = reinterpret_cast<const uint2 *>(tags)[i];
      const uint32_t t2[2] = {tg.x, tg.y};
      ulonglong2 kv;
      unsigned long long ko[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        ko[q] = EMPTY;
        if (t2[q] != TAG_EMPTY) {
          const uint32_t home = (((2 * i + q) >> 2) - (t2[q] & TAG_FAR)) & (PAGE_SLOTS / 4 - 1);
          const uint32_t rec = (home << fpb) | (t2[q] >> TAG_DB);
          ko[q] = unmix_key((gpage << R) | rec, bits);
        }
      }
      kv.x = ko[0];
      kv.y = ko[1];
      reinterpret_cast<ulonglong2 *>(gk)[i] = kv;
    }
    __syncthreads();
  }
  const uint32_t nnew = pg_wg_sum(n_new, dl);
  if (nnew && threadIdx.x == 0) atomicAdd(&stats->n_distinct, (unsigned long long)nnew);
  if (HIST) {  // the page's share of the totals (k_histo's HistoTotals), one row of ptot
    unsigned long long t4[4] = {0, 0, fh_lane, 0};
    if (HIST == 1) {
#pragma unroll
      for (int i = 0; i < WBH * 4; ++i) {
        t4[0] += cum[i] != 0;
        t4[1] += cum[i];
        t4[3] += cum[i] == 0xFFFFFFFFu;
      }
    } else {  // (one lane: a slot's sum is its count)
      t4[0] = one_unique;
      t4[1] = fh_lane;
      t4[3] = one_sat;
    }
    t4[3] += (unsigned long long)n_high << 32;  // word 3: low half = saturated sums (a count here, a flag to the host), high half = adds past the LDS bins
    unsigned long long *red = reinterpret_cast<unsigned long long *>(dl);  // [wave][4]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned long long v = t4[j];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * 4 + j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      unsigned long long v = 0;
      for (int w = 0; w < PG_WG / 64; ++w) v += red[w * 4 + threadIdx.x];
      fh.ptot[(size_t)page * 4 + threadIdx.x] = v;
    }
  }
}

// The rows of a fused page pass (k_pages32<true, true>) summed into the context's histogram and totals — what k_histo
// leaves there.  grid.x covers the (column, bin) pairs, grid.y slices the pages.
__global__ void __launch_bounds__(WG) k_hist_reduce(const uint32_t *__restrict__ partial, const unsigned long long *__restrict__ ptot,
                                                    uint32_t n_pages, uint32_t n_cols, unsigned long long hlen,
                                                    unsigned long long *__restrict__ hist, HistoTotals *__restrict__ tot) {
  const uint32_t per = (n_pages + gridDim.y - 1) / gridDim.y;
  const uint32_t p0 = blockIdx.y * per, p1 = p0 + per < n_pages ? p0 + per : n_pages;
  const uint32_t idx = blockIdx.x * WG + threadIdx.x, n_idx = n_cols * FH_BINS;
  if (idx < n_idx && p0 < p1) {
    const size_t stride = (size_t)n_idx;
    unsigned long long s = 0;
    uint32_t p = p0;
    for (; p + 8 <= p1; p += 8) {  // eight rows' loads in flight together
      uint32_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = partial[(size_t)(p + j) * stride + idx];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; p < p1; ++p) s += partial[(size_t)p * stride + idx];
    const uint32_t col = idx / FH_BINS, bin = idx % FH_BINS;
    if (s && bin < hlen) atomicAdd(&hist[(unsigned long long)col * hlen + bin], s);
  }
  if (blockIdx.x == 0 && threadIdx.x < 4 && p0 < p1) {
    unsigned long long v = 0;
    for (uint32_t p = p0; p < p1; ++p) v += ptot[(size_t)p * 4 + threadIdx.x];
    if (v) {
      if (threadIdx.x == 0) atomicAdd(&tot->n_unique, v);
      if (threadIdx.x == 1) atomicAdd(&tot->n_hashed, v);
      if (threadIdx.x == 2) atomicAdd(&tot->n_lane_sum, v);
      if (threadIdx.x == 3) atomicAdd(&tot->any_saturated, v);  // (low half: saturated sums — non-zero is what counts; high half: adds past the LDS bins)
    }
  }
}

}  // namespace shk
