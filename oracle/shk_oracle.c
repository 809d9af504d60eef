/*
 * shk_oracle.c — CPU ORACLE (test infrastructure only; see shk_oracle.h).
 *
 * Plain-C single-threaded restatement of sharkmer v3.1.0's k-mer counting
 * path.  Citations are to /root/reference (relative paths).  Nothing here is
 * reachable from the product library; tests/, smoke() and bench.py's
 * cpu_baseline leg are the only callers.
 */
#define _GNU_SOURCE
#include "shk_oracle.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ===================================================================== */
/* encoding.rs                                                           */
/* ===================================================================== */

/* src/kmer/encoding.rs:332-371 — single pass over ASCII bytes; A,C,G,T map to
 * 0..3 (:341-345); N resets frame/revframe/n_valid (:346-352); any other byte
 * is an error (:353-356); frame shifts left, revframe shifts right and takes
 * the complement at the top (:359-360); canonical = min(fwd,rev) once k valid
 * bases have been seen (:363-367). */
int orc_kmers_from_ascii(const uint8_t *seq, size_t len, int k, uint64_t *out,
                         size_t *n_out, uint8_t *bad_char) {
  if (!(k > 0 && k < 32)) return ORC_ERR_BAD_K; /* :333 */
  const uint64_t mask = (1ull << (2 * k)) - 1;
  uint64_t frame = 0, revframe = 0;
  size_t n_valid = 0, n = 0;
  for (size_t i = 0; i < len; i++) {
    uint64_t base;
    switch (seq[i]) {
      case 'A': base = 0; break;
      case 'C': base = 1; break;
      case 'G': base = 2; break;
      case 'T': base = 3; break;
      case 'N':
        n_valid = 0;
        frame = 0;
        revframe = 0;
        continue;
      default:
        if (bad_char) *bad_char = seq[i];
        if (n_out) *n_out = 0;
        return ORC_ERR_INVALID_CHAR;
    }
    frame = (frame << 2) | base;
    revframe = (revframe >> 2) | ((3 - base) << (2 * (k - 1)));
    n_valid++;
    if (n_valid >= (size_t)k) {
      uint64_t f = frame & mask, r = revframe & mask;
      out[n++] = f < r ? f : r;
    }
  }
  if (n_out) *n_out = n;
  return ORC_OK;
}

/* src/kmer/encoding.rs:374-376 */
uint64_t orc_count_valid_bases(const uint8_t *seq, size_t len) {
  uint64_t n = 0;
  for (size_t i = 0; i < len; i++) n += (seq[i] != 'N');
  return n;
}

/* src/kmer/encoding.rs:205-262 — 4 bases at a time through a byte table whose
 * entry is the reversed, complemented byte; then the 1–3 leftover bases. */
static uint8_t g_rc_lut[256];
static int g_rc_lut_ready = 0;
static void rc_lut_init(void) {
  for (int i = 0; i < 256; i++) {
    int b0 = i & 3, b1 = (i >> 2) & 3, b2 = (i >> 4) & 3, b3 = (i >> 6) & 3;
    g_rc_lut[i] = (uint8_t)(((3 - b0) << 6) | ((3 - b1) << 4) | ((3 - b2) << 2) | (3 - b3));
  }
  g_rc_lut_ready = 1;
}
uint64_t orc_revcomp_kmer(uint64_t kmer, int k) {
  if (!g_rc_lut_ready) rc_lut_init();
  int total_bits = 2 * k, remaining = k, shift = 0;
  uint64_t rc = 0;
  while (remaining >= 4) {
    rc = (rc << 8) | g_rc_lut[(kmer >> shift) & 0xFF];
    shift += 8;
    remaining -= 4;
  }
  for (int i = 0; i < remaining; i++) {
    uint64_t base = (kmer >> (shift + 2 * i)) & 3;
    rc = (rc << 2) | (3 - base);
  }
  return total_bits < 64 ? (rc & ((1ull << total_bits) - 1)) : rc;
}

/* src/kmer/encoding.rs:60-95 — 4 bases per byte, MSB first, tail left-aligned;
 * only ACGT accepted. */
long orc_read_from_str(const uint8_t *seq, size_t len, uint8_t *out) {
  uint8_t frame = 0;
  size_t length = 0, nb = 0;
  for (size_t i = 0; i < len; i++) {
    uint8_t base;
    length++;
    switch (seq[i]) {
      case 'A': base = 0; break;
      case 'C': base = 1; break;
      case 'G': base = 2; break;
      case 'T': base = 3; break;
      default: return ORC_ERR_INVALID_CHAR;
    }
    frame = (uint8_t)((frame << 2) | base);
    if (length % 4 == 0) {
      out[nb++] = frame;
      frame = 0;
    }
  }
  size_t modulo = length % 4;
  if (modulo != 0) {
    frame = (uint8_t)(frame << (2 * (4 - modulo)));
    out[nb++] = frame;
  }
  return (long)nb;
}

/* src/kmer/encoding.rs:130-190 — the reference's test-only extractor: walk
 * every packed base (including the padding of the last byte), then truncate
 * the (4 - length%4) k-mers the padding produced (:171-176). */
int orc_read_get_kmers(const uint8_t *packed, size_t n_bytes, size_t length,
                       int k, uint64_t *out, size_t *n_out) {
  if (!(k > 0 && k < 32)) return ORC_ERR_BAD_K;
  const uint64_t mask = (1ull << (2 * k)) - 1;
  uint64_t frame = 0, revframe = 0;
  size_t n_valid = 0, n = 0;
  *n_out = 0;
  if (length < (size_t)k) return ORC_OK; /* :141-143 */
  for (size_t b = 0; b < n_bytes; b++) {
    for (int j = 0; j < 4; j++) {
      uint64_t base = (packed[b] >> ((3 - j) * 2)) & 3;
      frame = (frame << 2) | base;
      revframe = (revframe >> 2) | ((3 - base) << (2 * (k - 1)));
      n_valid++;
      if (n_valid >= (size_t)k) {
        uint64_t f = frame & mask, r = revframe & mask;
        out[n++] = f < r ? f : r;
      }
    }
  }
  size_t modulo = length % 4;
  if (modulo != 0) {
    size_t extra = 4 - modulo;
    n = n > extra ? n - extra : 0;
  }
  if (n != length - (size_t)k + 1) return ORC_ERR_INVARIANT; /* :178-187 */
  *n_out = n;
  return ORC_OK;
}

int orc_seq_to_kmer(const uint8_t *seq, size_t len, uint64_t *out) {
  uint64_t kmer = 0;
  for (size_t i = 0; i < len; i++) {
    uint64_t base;
    switch (seq[i]) {
      case 'A': base = 0; break;
      case 'C': base = 1; break;
      case 'G': base = 2; break;
      case 'T': base = 3; break;
      default: return ORC_ERR_INVALID_CHAR;
    }
    kmer = (kmer << 2) | base;
  }
  *out = kmer;
  return ORC_OK;
}

void orc_kmer_to_seq(uint64_t kmer, int k, char *out) {
  static const char L[4] = {'A', 'C', 'G', 'T'};
  for (int i = 0; i < k; i++) out[i] = L[(kmer >> (2 * (k - i - 1))) & 3];
  out[k] = 0;
}

/* ===================================================================== */
/* u64 -> u32 / u64 open-addressing maps (stand-in for std HashMap+ahash; */
/* results are independent of hash function and iteration order —         */
/* SURVEY.md §8c "Third-party arithmetic")                                */
/* ===================================================================== */

#define MAP_EMPTY (~0ull) /* never a valid k-mer (k<32 ⇒ key < 2^62) nor a
                             valid count key in histo_large (count ≤ u32::MAX) */

static inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

typedef struct {
  uint64_t *keys;
  uint32_t *vals;
  size_t cap; /* power of two */
  size_t len;
} map32;

static int map32_init(map32 *m, size_t want) {
  size_t cap = 16;
  while (cap * 7 / 10 < want) cap <<= 1;
  m->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
  m->vals = (uint32_t *)calloc(cap, sizeof(uint32_t));
  if (!m->keys || !m->vals) return ORC_ERR_NOMEM;
  memset(m->keys, 0xFF, cap * sizeof(uint64_t));
  m->cap = cap;
  m->len = 0;
  return ORC_OK;
}
static void map32_release(map32 *m) {
  free(m->keys);
  free(m->vals);
  m->keys = NULL;
  m->vals = NULL;
  m->cap = m->len = 0;
}
static int map32_grow(map32 *m) {
  map32 n;
  size_t cap = m->cap << 1;
  n.keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
  n.vals = (uint32_t *)calloc(cap, sizeof(uint32_t));
  if (!n.keys || !n.vals) return ORC_ERR_NOMEM;
  memset(n.keys, 0xFF, cap * sizeof(uint64_t));
  n.cap = cap;
  n.len = m->len;
  for (size_t i = 0; i < m->cap; i++) {
    if (m->keys[i] == MAP_EMPTY) continue;
    size_t s = mix64(m->keys[i]) & (cap - 1);
    while (n.keys[s] != MAP_EMPTY) s = (s + 1) & (cap - 1);
    n.keys[s] = m->keys[i];
    n.vals[s] = m->vals[i];
  }
  free(m->keys);
  free(m->vals);
  *m = n;
  return ORC_OK;
}
/* entry(key).or_insert(0): returns pointer to the value slot */
static inline uint32_t *map32_entry(map32 *m, uint64_t key) {
  if ((m->len + 1) * 10 > m->cap * 7) {
    if (map32_grow(m) != ORC_OK) return NULL;
  }
  size_t s = mix64(key) & (m->cap - 1);
  for (;;) {
    uint64_t kk = m->keys[s];
    if (kk == key) return &m->vals[s];
    if (kk == MAP_EMPTY) {
      m->keys[s] = key;
      m->vals[s] = 0;
      m->len++;
      return &m->vals[s];
    }
    s = (s + 1) & (m->cap - 1);
  }
}
static inline const uint32_t *map32_find(const map32 *m, uint64_t key) {
  size_t s = mix64(key) & (m->cap - 1);
  for (;;) {
    uint64_t kk = m->keys[s];
    if (kk == key) return &m->vals[s];
    if (kk == MAP_EMPTY) return NULL;
    s = (s + 1) & (m->cap - 1);
  }
}

static inline uint32_t sat_add_u32(uint32_t a, uint32_t b) {
  uint32_t s = a + b;
  return s < a ? 0xFFFFFFFFu : s; /* u32::saturating_add */
}

/* small u64->u64 map for Histogram.histo_large (histogram.rs:15) */
typedef struct {
  uint64_t *keys;
  uint64_t *vals;
  size_t cap, len;
} map64;
static void map64_init(map64 *m) {
  m->cap = 16;
  m->len = 0;
  m->keys = (uint64_t *)malloc(m->cap * sizeof(uint64_t));
  m->vals = (uint64_t *)calloc(m->cap, sizeof(uint64_t));
  memset(m->keys, 0xFF, m->cap * sizeof(uint64_t));
}
static void map64_release(map64 *m) {
  free(m->keys);
  free(m->vals);
}
static void map64_rebuild(map64 *m, size_t cap) {
  map64 n;
  n.cap = cap;
  n.len = 0;
  n.keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
  n.vals = (uint64_t *)calloc(cap, sizeof(uint64_t));
  memset(n.keys, 0xFF, cap * sizeof(uint64_t));
  for (size_t i = 0; i < m->cap; i++) {
    if (m->keys[i] == MAP_EMPTY) continue;
    size_t s = mix64(m->keys[i]) & (cap - 1);
    while (n.keys[s] != MAP_EMPTY) s = (s + 1) & (cap - 1);
    n.keys[s] = m->keys[i];
    n.vals[s] = m->vals[i];
    n.len++;
  }
  free(m->keys);
  free(m->vals);
  *m = n;
}
static uint64_t *map64_entry(map64 *m, uint64_t key) {
  if ((m->len + 1) * 10 > m->cap * 7) map64_rebuild(m, m->cap << 1);
  size_t s = mix64(key) & (m->cap - 1);
  for (;;) {
    if (m->keys[s] == key) return &m->vals[s];
    if (m->keys[s] == MAP_EMPTY) {
      m->keys[s] = key;
      m->vals[s] = 0;
      m->len++;
      return &m->vals[s];
    }
    s = (s + 1) & (m->cap - 1);
  }
}
static uint64_t *map64_find(const map64 *m, uint64_t key) {
  size_t s = mix64(key) & (m->cap - 1);
  for (;;) {
    if (m->keys[s] == key) return &m->vals[s];
    if (m->keys[s] == MAP_EMPTY) return NULL;
    s = (s + 1) & (m->cap - 1);
  }
}
/* remove by rebuilding without the key (the tail is tiny; simplicity wins) */
static void map64_remove(map64 *m, uint64_t key) {
  uint64_t *v = map64_find(m, key);
  if (!v) return;
  size_t idx = (size_t)(v - m->vals);
  m->keys[idx] = MAP_EMPTY;
  m->vals[idx] = 0;
  m->len--;
  map64_rebuild(m, m->cap);
}

/* ===================================================================== */
/* counting.rs: KmerCounts                                               */
/* ===================================================================== */

struct orc_counts {
  map32 kmers; /* counting.rs:113-116 */
  int k;
  uint64_t *scratch; /* k-mer buffer for ingest_seq */
  size_t scratch_cap;
};

orc_counts *orc_counts_new_with_capacity(int k, size_t cap) {
  orc_counts *c = (orc_counts *)calloc(1, sizeof(*c));
  if (!c) return NULL;
  c->k = k;
  if (map32_init(&c->kmers, cap) != ORC_OK) {
    free(c);
    return NULL;
  }
  return c;
}
orc_counts *orc_counts_new(int k) { return orc_counts_new_with_capacity(k, 0); }
void orc_counts_free(orc_counts *c) {
  if (!c) return;
  map32_release(&c->kmers);
  free(c->scratch);
  free(c);
}
int orc_counts_k(const orc_counts *c) { return c->k; }

/* counting.rs:144-149 with insert_or_add(kmer,1) (:82-85):
 * kmers_from_ascii first (so an invalid character leaves the table untouched
 * for this sequence), then entry().or_insert(0), saturating_add(1). */
int orc_counts_ingest_seq(orc_counts *c, const uint8_t *seq, size_t len,
                          uint8_t *bad_char) {
  if (c->scratch_cap < len + 1) {
    size_t nc = len + 256;
    uint64_t *p = (uint64_t *)realloc(c->scratch, nc * sizeof(uint64_t));
    if (!p) return ORC_ERR_NOMEM;
    c->scratch = p;
    c->scratch_cap = nc;
  }
  size_t n = 0;
  int rc = orc_kmers_from_ascii(seq, len, c->k, c->scratch, &n, bad_char);
  if (rc != ORC_OK) return rc;
  for (size_t i = 0; i < n; i++) {
    uint32_t *v = map32_entry(&c->kmers, c->scratch[i]);
    if (!v) return ORC_ERR_NOMEM;
    *v = sat_add_u32(*v, 1);
  }
  return ORC_OK;
}

int orc_counts_insert(orc_counts *c, uint64_t kmer, uint32_t count) {
  uint32_t *v = map32_entry(&c->kmers, kmer);
  if (!v) return ORC_ERR_NOMEM;
  *v = sat_add_u32(*v, count);
  return ORC_OK;
}

/* counting.rs:157-166 */
int orc_counts_extend(orc_counts *dst, const orc_counts *src) {
  if (dst->k != src->k) return ORC_ERR_K_MISMATCH;
  for (size_t i = 0; i < src->kmers.cap; i++) {
    if (src->kmers.keys[i] == MAP_EMPTY) continue;
    uint32_t *v = map32_entry(&dst->kmers, src->kmers.keys[i]);
    if (!v) return ORC_ERR_NOMEM;
    *v = sat_add_u32(*v, src->kmers.vals[i]);
  }
  return ORC_OK;
}

uint32_t orc_counts_get_count(const orc_counts *c, uint64_t kmer) {
  const uint32_t *v = map32_find(&c->kmers, kmer);
  return v ? *v : 0;
}
int orc_counts_contains(const orc_counts *c, uint64_t kmer) {
  return map32_find(&c->kmers, kmer) != NULL;
}
/* counting.rs:205-209 */
uint32_t orc_counts_get_canonical_count(const orc_counts *c, uint64_t kmer) {
  uint64_t rc = orc_revcomp_kmer(kmer, c->k);
  return orc_counts_get_count(c, kmer < rc ? kmer : rc);
}
/* counting.rs:218-222 */
int orc_counts_get_canonical(const orc_counts *c, uint64_t kmer, uint32_t *count) {
  const uint32_t *v = map32_find(&c->kmers, kmer);
  if (!v) v = map32_find(&c->kmers, orc_revcomp_kmer(kmer, c->k));
  if (!v) return 0;
  if (count) *count = *v;
  return 1;
}
uint64_t orc_counts_n_kmers(const orc_counts *c) {
  uint64_t s = 0;
  for (size_t i = 0; i < c->kmers.cap; i++)
    if (c->kmers.keys[i] != MAP_EMPTY) s += c->kmers.vals[i];
  return s;
}
uint64_t orc_counts_n_unique(const orc_counts *c) { return c->kmers.len; }
uint32_t orc_counts_max_count(const orc_counts *c) {
  uint32_t m = 0;
  for (size_t i = 0; i < c->kmers.cap; i++)
    if (c->kmers.keys[i] != MAP_EMPTY && c->kmers.vals[i] > m) m = c->kmers.vals[i];
  return m;
}
static int cmp_u32(const void *a, const void *b) {
  uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return x < y ? -1 : x > y;
}
/* counting.rs:275-298 — even length: (lower_mid/2)+(upper_mid/2) */
uint32_t orc_counts_median_count(const orc_counts *c) {
  size_t n = c->kmers.len;
  if (n == 0) return 0;
  uint32_t *v = (uint32_t *)malloc(n * sizeof(uint32_t));
  size_t j = 0;
  for (size_t i = 0; i < c->kmers.cap; i++)
    if (c->kmers.keys[i] != MAP_EMPTY) v[j++] = c->kmers.vals[i];
  qsort(v, n, sizeof(uint32_t), cmp_u32);
  size_t mid = n / 2;
  uint32_t r = (n % 2 == 0) ? (v[mid - 1] / 2) + (v[mid] / 2) : v[mid];
  free(v);
  return r;
}
/* counting.rs:234-237 → retain_above (:66-68) */
void orc_counts_remove_low(orc_counts *c, uint32_t min_count) {
  map32 n;
  map32_init(&n, c->kmers.len);
  for (size_t i = 0; i < c->kmers.cap; i++) {
    if (c->kmers.keys[i] == MAP_EMPTY || c->kmers.vals[i] < min_count) continue;
    *map32_entry(&n, c->kmers.keys[i]) = c->kmers.vals[i];
  }
  map32_release(&c->kmers);
  c->kmers = n;
}
size_t orc_counts_export(const orc_counts *c, uint64_t *keys, uint32_t *counts) {
  size_t j = 0;
  for (size_t i = 0; i < c->kmers.cap; i++) {
    if (c->kmers.keys[i] == MAP_EMPTY) continue;
    if (keys) keys[j] = c->kmers.keys[i];
    if (counts) counts[j] = c->kmers.vals[i];
    j++;
  }
  return j;
}

/* ===================================================================== */
/* histogram.rs: Histogram                                               */
/* ===================================================================== */

struct orc_histo {
  uint64_t *histo; /* len histo_max+2, histogram.rs:13,19-21 */
  map64 large;     /* histogram.rs:14 */
  uint64_t histo_max;
};

orc_histo *orc_histo_new(uint64_t histo_max) {
  orc_histo *h = (orc_histo *)calloc(1, sizeof(*h));
  if (!h) return NULL;
  h->histo = (uint64_t *)calloc(histo_max + 2, sizeof(uint64_t));
  if (!h->histo) {
    free(h);
    return NULL;
  }
  map64_init(&h->large);
  h->histo_max = histo_max;
  return h;
}
void orc_histo_free(orc_histo *h) {
  if (!h) return;
  free(h->histo);
  map64_release(&h->large);
  free(h);
}

/* histogram.rs:51-85 */
void orc_histo_move_count(orc_histo *h, uint64_t old_count, uint64_t new_count) {
  if (old_count == new_count) return;
  if (old_count > 0) {
    if (old_count <= h->histo_max) {
      if (h->histo[old_count] > 0) h->histo[old_count]--; /* saturating_sub */
    } else {
      uint64_t *n = map64_find(&h->large, old_count);
      if (n) {
        if (*n > 0) (*n)--;
        if (*n == 0) map64_remove(&h->large, old_count);
      }
    }
  }
  if (new_count <= h->histo_max)
    h->histo[new_count]++;
  else
    (*map64_entry(&h->large, new_count))++;
}

/* histogram.rs:31-42 */
void orc_histo_ingest_counts(orc_histo *h, const orc_counts *c) {
  for (size_t i = 0; i < c->kmers.cap; i++) {
    if (c->kmers.keys[i] == MAP_EMPTY) continue;
    uint64_t count = c->kmers.vals[i];
    if (count <= h->histo_max)
      h->histo[count]++;
    else
      (*map64_entry(&h->large, count))++;
  }
}
uint64_t orc_histo_get(const orc_histo *h, uint64_t count) {
  if (count <= h->histo_max) return h->histo[count];
  uint64_t *v = map64_find(&h->large, count);
  return v ? *v : 0;
}
/* histogram.rs:103-117 */
uint64_t orc_histo_n_kmers(const orc_histo *h) {
  uint64_t s = 0;
  for (uint64_t i = 1; i < h->histo_max + 2; i++) s += h->histo[i] * i;
  for (size_t i = 0; i < h->large.cap; i++)
    if (h->large.keys[i] != MAP_EMPTY) s += h->large.keys[i] * h->large.vals[i];
  return s;
}
/* histogram.rs:119-123 */
uint64_t orc_histo_n_unique(const orc_histo *h) {
  uint64_t s = 0;
  for (uint64_t i = 1; i < h->histo_max + 2; i++) s += h->histo[i];
  for (size_t i = 0; i < h->large.cap; i++)
    if (h->large.keys[i] != MAP_EMPTY) s += h->large.vals[i];
  return s;
}
/* histogram.rs:125-134 — dense copy, tail folded into the LAST element */
void orc_histo_get_vector(const orc_histo *h, uint64_t *out) {
  memcpy(out, h->histo, (h->histo_max + 2) * sizeof(uint64_t));
  for (size_t i = 0; i < h->large.cap; i++)
    if (h->large.keys[i] != MAP_EMPTY) out[h->histo_max + 1] += h->large.vals[i];
}

/* counting.rs:171-202 with insert_or_add_get_counts (:86-92) */
int orc_counts_extend_with_histogram(orc_counts *dst, const orc_counts *src,
                                     orc_histo *h, int *any_saturated) {
  if (dst->k != src->k) return ORC_ERR_K_MISMATCH;
  int sat = 0;
  for (size_t i = 0; i < src->kmers.cap; i++) {
    if (src->kmers.keys[i] == MAP_EMPTY) continue;
    uint32_t *v = map32_entry(&dst->kmers, src->kmers.keys[i]);
    if (!v) return ORC_ERR_NOMEM;
    uint32_t old = *v;
    uint32_t stored = sat_add_u32(old, src->kmers.vals[i]);
    *v = stored;
    orc_histo_move_count(h, old, stored);
    if (stored == 0xFFFFFFFFu && old < 0xFFFFFFFFu) sat = 1;
  }
  if (any_saturated && sat) *any_saturated = 1;
  return ORC_OK;
}

/* ===================================================================== */
/* chunk.rs + io.rs driver                                               */
/* ===================================================================== */

#define N_READS_PER_BATCH 1000 /* io.rs:15 */

typedef struct {
  orc_counts *kmer_counts; /* chunk.rs:10 */
  uint64_t n_reads;        /* chunk.rs:11 */
  uint64_t n_bases;        /* chunk.rs:12 */
} orc_chunk;

struct orc_run {
  int k;
  uint32_t chunks_arg; /* args.chunks */
  uint32_t n_chunks;   /* io.rs:378 */
  uint64_t histo_max;
  orc_chunk *chunks;   /* FastqReadState.chunks, io.rs:202 */
  size_t chunk_index;  /* io.rs:203 */
  /* FastqReadState.seqs (io.rs:204): pending batch, concatenated */
  uint8_t *pend;
  size_t pend_len, pend_cap;
  size_t *pend_off; /* n_pend+1 offsets */
  size_t n_pend, pend_off_cap;
  orc_run_stats st;
  uint64_t *histo_vecs; /* chunks × (histo_max+2) */
  orc_counts *merged;
  int finished;
  uint8_t bad_char;
  char err[512];
};

orc_run *orc_run_new(int k, uint32_t chunks, uint64_t histo_max) {
  orc_run *r = (orc_run *)calloc(1, sizeof(*r));
  if (!r) return NULL;
  r->k = k;
  r->chunks_arg = chunks;
  r->n_chunks = chunks == 0 ? 1 : chunks; /* io.rs:378 */
  r->histo_max = histo_max;
  r->chunks = (orc_chunk *)calloc(r->n_chunks, sizeof(orc_chunk));
  for (uint32_t i = 0; i < r->n_chunks; i++) r->chunks[i].kmer_counts = orc_counts_new(k);
  r->pend_off_cap = N_READS_PER_BATCH + 1;
  r->pend_off = (size_t *)calloc(r->pend_off_cap, sizeof(size_t));
  return r;
}

void orc_run_free(orc_run *r) {
  if (!r) return;
  if (r->chunks) {
    for (uint32_t i = 0; i < r->n_chunks; i++) orc_counts_free(r->chunks[i].kmer_counts);
    free(r->chunks);
  }
  orc_counts_free(r->merged);
  free(r->pend);
  free(r->pend_off);
  free(r->histo_vecs);
  free(r);
}

/* chunk.rs:25-30 */
static int chunk_ingest_seq(orc_run *r, orc_chunk *c, const uint8_t *seq, size_t len) {
  int rc = orc_counts_ingest_seq(c->kmer_counts, seq, len, &r->bad_char);
  if (rc != ORC_OK) {
    if (rc == ORC_ERR_INVALID_CHAR)
    { /* `b as char`: the byte is the scalar U+00b — two UTF-8 bytes from 0x80 on */
      char ch[3] = {0, 0, 0};
      if (r->bad_char < 0x80) ch[0] = (char)r->bad_char;
      else { ch[0] = (char)(0xC0 | (r->bad_char >> 6)); ch[1] = (char)(0x80 | (r->bad_char & 0x3F)); }
      snprintf(r->err, sizeof r->err,
               "Invalid character '%s' in sequence. Only ACGTN allowed.", ch);
    }
    return rc;
  }
  c->n_reads += 1;
  c->n_bases += orc_count_valid_bases(seq, len);
  return ORC_OK;
}

/* io.rs:355-361 */
static int drain_batch(orc_run *r) {
  for (size_t i = 0; i < r->n_pend; i++) {
    int rc = chunk_ingest_seq(r, &r->chunks[r->chunk_index], r->pend + r->pend_off[i],
                              r->pend_off[i + 1] - r->pend_off[i]);
    if (rc != ORC_OK) {
      /* the reference aborts the whole run here ('?' to main) */
      r->n_pend = 0;
      r->pend_len = 0;
      return rc;
    }
  }
  r->n_pend = 0;
  r->pend_len = 0;
  r->chunk_index = (r->chunk_index + 1) % r->n_chunks;
  return ORC_OK;
}

/* io.rs:335-343 */
int orc_run_push_seq(orc_run *r, const uint8_t *seq, size_t len) {
  if (r->pend_len + len > r->pend_cap) {
    size_t nc = (r->pend_cap ? r->pend_cap * 2 : 1 << 18);
    while (nc < r->pend_len + len) nc *= 2;
    uint8_t *p = (uint8_t *)realloc(r->pend, nc);
    if (!p) return ORC_ERR_NOMEM;
    r->pend = p;
    r->pend_cap = nc;
  }
  if (len) memcpy(r->pend + r->pend_len, seq, len);
  r->pend_off[r->n_pend] = r->pend_len;
  r->pend_len += len;
  r->n_pend++;
  r->pend_off[r->n_pend] = r->pend_len;
  r->st.n_bases_read += len;
  r->st.n_reads_read += 1;
  if (r->st.n_reads_read % N_READS_PER_BATCH == 0) return drain_batch(r);
  return ORC_OK;
}

int orc_run_push_batch(orc_run *r, const uint8_t *bases, const uint64_t *offsets,
                       size_t n_seqs) {
  for (size_t i = 0; i < n_seqs; i++) {
    int rc = orc_run_push_seq(r, bases + offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
    if (rc != ORC_OK) return rc;
  }
  return ORC_OK;
}

/* io.rs:542-552, 578-580, then io.rs:977-1161 */
int orc_run_finish(orc_run *r) {
  if (r->finished) return ORC_OK;
  int rc = drain_batch(r); /* io.rs:542-543 (advances chunk_index too; harmless) */
  if (rc != ORC_OK) return rc;
  for (uint32_t i = 0; i < r->n_chunks; i++) { /* io.rs:545-552 */
    r->st.n_reads_ingested += r->chunks[i].n_reads;
    r->st.n_bases_ingested += r->chunks[i].n_bases;
    r->st.n_kmers_ingested += orc_counts_n_kmers(r->chunks[i].kmer_counts);
  }
  if (r->st.n_reads_ingested == 0) { /* io.rs:578-580 */
    snprintf(r->err, sizeof r->err,
             "No reads were ingested. Check that input files contain valid FASTQ records.");
    return ORC_ERR_NO_READS;
  }
  size_t est = 0; /* io.rs:1005-1006 */
  for (uint32_t i = 0; i < r->n_chunks; i++) est += orc_counts_n_unique(r->chunks[i].kmer_counts);
  r->merged = orc_counts_new_with_capacity(r->k, est);
  if (!r->merged) return ORC_ERR_NOMEM;
  r->st.has_histogram = r->chunks_arg > 0;
  if (r->chunks_arg > 0) {
    size_t len = r->histo_max + 2;
    r->histo_vecs = (uint64_t *)calloc((size_t)r->n_chunks * len, sizeof(uint64_t));
    orc_histo *running = orc_histo_new(r->histo_max); /* io.rs:1021 */
    if (!r->histo_vecs || !running) return ORC_ERR_NOMEM;
    for (uint32_t i = 0; i < r->n_chunks; i++) { /* io.rs:1023-1028 */
      rc = orc_counts_extend_with_histogram(r->merged, r->chunks[i].kmer_counts, running,
                                            &r->st.any_saturated);
      if (rc != ORC_OK) return rc;
      orc_counts_free(r->chunks[i].kmer_counts); /* drop(chunk) */
      r->chunks[i].kmer_counts = NULL;
      orc_histo_get_vector(running, r->histo_vecs + (size_t)i * len);
    }
    r->st.n_hashed_kmers = orc_counts_n_kmers(r->merged); /* io.rs:1035 */
    r->st.n_unique_kmers = orc_counts_n_unique(r->merged);
    if (r->st.n_hashed_kmers != r->st.n_kmers_ingested) { /* io.rs:1042-1047 */
      snprintf(r->err, sizeof r->err,
               "The total count of hashed kmers (%llu) does not equal the number of ingested kmers (%llu)",
               (unsigned long long)r->st.n_hashed_kmers,
               (unsigned long long)r->st.n_kmers_ingested);
      orc_histo_free(running);
      return ORC_ERR_INVARIANT;
    }
    r->st.n_singleton_kmers = r->histo_vecs[(size_t)(r->n_chunks - 1) * len + 1]; /* :1096-1099 */
    uint64_t nk = orc_histo_n_kmers(running), nu = orc_histo_n_unique(running);
    orc_histo_free(running);
    if (nk != r->st.n_kmers_ingested) { /* io.rs:1120-1125 */
      snprintf(r->err, sizeof r->err,
               "The total count of kmers in the histogram (%llu) does not equal the total expected count of kmers (%llu)",
               (unsigned long long)nk, (unsigned long long)r->st.n_kmers_ingested);
      return ORC_ERR_INVARIANT;
    }
    if (nu != r->st.n_unique_kmers) { /* io.rs:1127-1132 */
      snprintf(r->err, sizeof r->err,
               "The total count of unique kmers in the histogram (%llu) does not equal the total count of hashed kmers (%llu)",
               (unsigned long long)nu, (unsigned long long)r->st.n_unique_kmers);
      return ORC_ERR_INVARIANT;
    }
  } else { /* io.rs:1133-1158 */
    for (uint32_t i = 0; i < r->n_chunks; i++) {
      rc = orc_counts_extend(r->merged, r->chunks[i].kmer_counts);
      if (rc != ORC_OK) return rc;
      orc_counts_free(r->chunks[i].kmer_counts);
      r->chunks[i].kmer_counts = NULL;
    }
    r->st.n_hashed_kmers = orc_counts_n_kmers(r->merged);
    r->st.n_unique_kmers = orc_counts_n_unique(r->merged);
    if (r->st.n_hashed_kmers != r->st.n_kmers_ingested) {
      snprintf(r->err, sizeof r->err,
               "The total count of hashed kmers (%llu) does not equal the number of ingested kmers (%llu)",
               (unsigned long long)r->st.n_hashed_kmers,
               (unsigned long long)r->st.n_kmers_ingested);
      return ORC_ERR_INVARIANT;
    }
  }
  r->finished = 1;
  return ORC_OK;
}

const orc_run_stats *orc_run_get_stats(const orc_run *r) { return &r->st; }
const uint64_t *orc_run_histograms(const orc_run *r, size_t *n_cols, size_t *len) {
  if (n_cols) *n_cols = r->chunks_arg;
  if (len) *len = (size_t)r->histo_max + 2;
  return r->histo_vecs;
}
const orc_counts *orc_run_merged(const orc_run *r) { return r->merged; }
const char *orc_run_error(const orc_run *r) { return r->err; }
uint8_t orc_run_bad_char(const orc_run *r) { return r->bad_char; }

/* ===================================================================== */
/* FASTQ reading: io.rs:161-198, 213-265, 271-352, 598-625               */
/* ===================================================================== */

/* An io::Error as BufRead::lines hands it to read_fastq: its ErrorKind (printed with {:?} by
 * stream_io_error, io.rs:213-265) and its Display text. */
enum { IOK_NONE = 0, IOK_UNEXPECTED_EOF, IOK_INVALID_INPUT, IOK_INVALID_DATA, IOK_OTHER };
typedef struct {
  int kind;
  char text[96];
} io_err;
static const char *iok_name(int k) {
  switch (k) {
    case IOK_UNEXPECTED_EOF: return "UnexpectedEof";
    case IOK_INVALID_INPUT: return "InvalidInput";
    case IOK_INVALID_DATA: return "InvalidData";
    default: return "Other";
  }
}
static void io_set(io_err *e, int kind, const char *text) {
  e->kind = kind;
  snprintf(e->text, sizeof e->text, "%s", text);
}

/* The Box<dyn BufRead> of open_fastq_reader (io.rs:598-625): the file itself, or
 * flate2::read::GzDecoder over it.  flate2 1.1.9 / miniz_oxide 0.8.9 (Cargo.lock:569-576,
 * 940-947) are not under /root/reference; what is restated here is their published behaviour
 * (flate2 src/gz/bufread.rs GzDecoder::{new,read} with multi = false, src/gz/mod.rs
 * GzHeaderParser::parse, src/zio.rs read):
 *   - the header is parsed when the decoder is made; an error is KEPT and returned by the first
 *     read: 1f 8b / CM 8 / reserved flag bits → InvalidInput "invalid gzip header", a NUL-ended
 *     field past 65535 bytes → InvalidInput "gzip header field too long", FHCRC mismatch →
 *     InvalidInput "corrupt gzip stream does not have a matching checksum", input ending inside
 *     it → UnexpectedEof;
 *   - the body is raw DEFLATE; a decoder error → InvalidInput "corrupt deflate stream"; input
 *     that ends inside the stream makes the decoder report no progress, zio::read returns Ok(0),
 *     and the decoder goes on to the trailer;
 *   - the trailer's 8 bytes: any missing → UnexpectedEof; CRC-32 or ISIZE not what was written →
 *     the "matching checksum" error;
 *   - then end of stream for ever: ONE member, whatever follows it.
 * The inflate engine is zlib's raw inflate (windowBits −15): member framing, checks and errors are
 * all done here, none of them by zlib's gz* layer. */
typedef struct {
  FILE *f;
  int use_gzip;
  int gz_state; /* 0 error pending (gz_err), 1 body, 2 trailer, 3 end */
  io_err gz_err;
  z_stream zs;
  int zs_init;
  unsigned char inbuf[1 << 15];
  uint32_t crc;
  uint64_t amount;
  int all_members; /* NOT the reference: flate2::read::MultiGzDecoder's behaviour, on request (orc_run_set_gzip_all_members) */
} byte_source;

/* the compressed input, with what inflate has not consumed yet served first */
static int src_getc(byte_source *s) {
  if (s->zs.avail_in == 0) {
    size_t n = fread(s->inbuf, 1, sizeof s->inbuf, s->f);
    if (n == 0) return -1;
    s->zs.next_in = s->inbuf;
    s->zs.avail_in = (uInt)n;
  }
  s->zs.avail_in--;
  return *s->zs.next_in++;
}

static void gz_parse_header(byte_source *s) {
  unsigned char h[10];
  uint32_t hcrc = 0;
  s->gz_state = 0;
  for (int i = 0; i < 10; i++) {
    int c = src_getc(s);
    if (c < 0) { io_set(&s->gz_err, IOK_UNEXPECTED_EOF, "unexpected end of file"); return; }
    h[i] = (unsigned char)c;
  }
  hcrc = (uint32_t)crc32(0L, h, 10);
  if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 0xE0)) {
    io_set(&s->gz_err, IOK_INVALID_INPUT, "invalid gzip header");
    return;
  }
  int flg = h[3];
  if (flg & 4) { /* FEXTRA */
    unsigned char x[2];
    for (int i = 0; i < 2; i++) {
      int c = src_getc(s);
      if (c < 0) { io_set(&s->gz_err, IOK_UNEXPECTED_EOF, "unexpected end of file"); return; }
      x[i] = (unsigned char)c;
    }
    hcrc = (uint32_t)crc32(hcrc, x, 2);
    unsigned xlen = x[0] | (x[1] << 8);
    for (unsigned i = 0; i < xlen; i++) {
      int c = src_getc(s);
      if (c < 0) { io_set(&s->gz_err, IOK_UNEXPECTED_EOF, "unexpected end of file"); return; }
      unsigned char b = (unsigned char)c;
      hcrc = (uint32_t)crc32(hcrc, &b, 1);
    }
  }
  for (int field = 0; field < 2; field++) { /* FNAME, FCOMMENT: read_to_nul */
    if (!(flg & (field == 0 ? 8 : 16))) continue;
    size_t held = 0;
    for (;;) {
      int c = src_getc(s);
      if (c < 0) { io_set(&s->gz_err, IOK_UNEXPECTED_EOF, "unexpected end of file"); return; }
      unsigned char b = (unsigned char)c;
      hcrc = (uint32_t)crc32(hcrc, &b, 1);
      if (b == 0) break;
      if (held == 65535) { io_set(&s->gz_err, IOK_INVALID_INPUT, "gzip header field too long"); return; }
      held++;
    }
  }
  if (flg & 2) { /* FHCRC */
    unsigned char x[2];
    for (int i = 0; i < 2; i++) {
      int c = src_getc(s);
      if (c < 0) { io_set(&s->gz_err, IOK_UNEXPECTED_EOF, "unexpected end of file"); return; }
      x[i] = (unsigned char)c;
    }
    if ((unsigned)(x[0] | (x[1] << 8)) != (hcrc & 0xFFFF)) {
      io_set(&s->gz_err, IOK_INVALID_INPUT, "corrupt gzip stream does not have a matching checksum");
      return;
    }
  }
  s->gz_state = 1;
}

/* Read::read: >0 bytes, 0 end of stream, -1 error (*err) */
static long source_read(byte_source *s, unsigned char *dst, size_t cap, io_err *err) {
  if (!s->use_gzip) {
    if (s->zs.avail_in) { /* what the peek had read */
      size_t n = s->zs.avail_in < cap ? s->zs.avail_in : cap;
      memcpy(dst, s->zs.next_in, n);
      s->zs.next_in += n;
      s->zs.avail_in -= (uInt)n;
      return (long)n;
    }
    size_t n = fread(dst, 1, cap, s->f);
    if (n == 0 && ferror(s->f)) { io_set(err, IOK_OTHER, strerror(errno)); return -1; }
    return (long)n;
  }
  for (;;) {
    if (s->gz_state == 0) { /* GzState::Err: handed out once, then End */
      *err = s->gz_err;
      s->gz_state = 3;
      return -1;
    }
    if (s->gz_state == 3) return 0;
    if (s->gz_state == 1) {
      if (s->zs.avail_in == 0) {
        size_t n = fread(s->inbuf, 1, sizeof s->inbuf, s->f);
        s->zs.next_in = s->inbuf;
        s->zs.avail_in = (uInt)n;
        if (n == 0) { /* the input ends inside the stream: Ok(0) from zio::read, on to the trailer */
          s->gz_state = 2;
          continue;
        }
      }
      s->zs.next_out = dst;
      s->zs.avail_out = (uInt)cap;
      int rc = inflate(&s->zs, Z_NO_FLUSH);
      size_t got = cap - s->zs.avail_out;
      if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
        /* flate2's zio::read: `Err(..) => return Err(io::Error::new(InvalidInput, "corrupt deflate
         * stream"))` — the bytes this very call had already written into the caller's buffer (`got`:
         * up to the 8 KiB of BufReader's fill_buf; the call began where the one before it ended, i.e.
         * at the last point where the output buffer was full or the 32 KiB of compressed input ran
         * out) are NOT handed out: the read fails, BufReader's buffer stays empty, read_until returns
         * the error.  Round 4: restated here as it is (rounds 2-3 kept those bytes, as the product does
         * — see DESIGN.md §2 on which record a corrupt stream is blamed on). */
        (void)got;
        io_set(err, IOK_INVALID_INPUT, "corrupt deflate stream");
        s->gz_state = 3;
        return -1;
      }
      s->crc = (uint32_t)crc32(s->crc, dst, (uInt)got);
      s->amount += got;
      if (rc == Z_STREAM_END) s->gz_state = 2;
      if (got) return (long)got;
      continue;
    }
    /* trailer: CRC-32, ISIZE, little endian, from what inflate left unread and then the file */
    unsigned char t[8];
    for (int i = 0; i < 8; i++) {
      int c = src_getc(s);
      if (c < 0) {
        io_set(err, IOK_UNEXPECTED_EOF, "unexpected end of file");
        s->gz_state = 3;
        return -1;
      }
      t[i] = (unsigned char)c;
    }
    uint32_t want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    uint32_t want_len = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
    s->gz_state = 3;
    if (want_crc != s->crc || want_len != (uint32_t)s->amount) {
      io_set(err, IOK_INVALID_INPUT, "corrupt gzip stream does not have a matching checksum");
      return -1;
    }
    if (s->all_members) { /* MultiGzDecoder: bytes behind a member are the next member's header */
      int c = src_getc(s);
      if (c < 0) return 0;
      s->zs.next_in--; /* (src_getc served it out of inbuf: back it goes) */
      s->zs.avail_in++;
      inflateReset(&s->zs);
      s->crc = 0;
      s->amount = 0;
      gz_parse_header(s); /* → body, or an error pending */
      continue;
    }
    return 0;
  }
}

/* str::from_utf8 on one line: decode scalar values, refuse overlong forms, surrogates and
 * anything above U+10FFFF */
static int line_is_utf8(const unsigned char *s, size_t n) {
  size_t i = 0;
  while (i < n) {
    unsigned c = s[i];
    unsigned len, cp, min;
    if (c < 0x80) { i++; continue; }
    if ((c & 0xE0) == 0xC0) { len = 2; cp = c & 0x1F; min = 0x80; }
    else if ((c & 0xF0) == 0xE0) { len = 3; cp = c & 0x0F; min = 0x800; }
    else if ((c & 0xF8) == 0xF0) { len = 4; cp = c & 0x07; min = 0x10000; }
    else return 0;
    if (i + len > n) return 0;
    for (unsigned j = 1; j < len; j++) {
      if ((s[i + j] & 0xC0) != 0x80) return 0;
      cp = (cp << 6) | (s[i + j] & 0x3F);
    }
    if (cp < min || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return 0;
    i += len;
  }
  return 1;
}

/* BufReader + BufRead::lines(): read_until '\n' (a failed read returns the error and what had been
 * gathered of the line is gone), the line must be UTF-8 (else InvalidData "stream did not contain
 * valid UTF-8"), then one trailing '\n' and — only then — one trailing '\r' are stripped. */
typedef struct {
  byte_source src;
  unsigned char rbuf[8192];
  size_t rpos, rlen;
  char *buf;
  size_t cap;
} line_reader;

/* returns 1 line read, 0 EOF, -1 error (*err) */
static int next_line(line_reader *lr, char **line, size_t *len, io_err *err) {
  size_t n = 0;
  int saw_nl = 0;
  for (;;) {
    if (lr->rpos == lr->rlen) {
      long got = source_read(&lr->src, lr->rbuf, sizeof lr->rbuf, err);
      if (got < 0) return -1;
      if (got == 0) break;
      lr->rpos = 0;
      lr->rlen = (size_t)got;
    }
    unsigned char *p = lr->rbuf + lr->rpos;
    size_t avail = lr->rlen - lr->rpos;
    unsigned char *nl = (unsigned char *)memchr(p, '\n', avail);
    size_t take = nl ? (size_t)(nl - p) + 1 : avail;
    if (lr->cap < n + take + 1) {
      size_t nc = lr->cap ? lr->cap * 2 : 1 << 16;
      while (nc < n + take + 1) nc *= 2;
      char *q = (char *)realloc(lr->buf, nc);
      if (!q) { io_set(err, IOK_OTHER, "out of memory"); return -1; }
      lr->buf = q;
      lr->cap = nc;
    }
    memcpy(lr->buf + n, p, take);
    n += take;
    lr->rpos += take;
    if (nl) { saw_nl = 1; break; }
  }
  if (n == 0) return 0;
  if (!line_is_utf8((const unsigned char *)lr->buf, n)) {
    io_set(err, IOK_INVALID_DATA, "stream did not contain valid UTF-8");
    return -1;
  }
  if (saw_nl) {
    n--;
    if (n > 0 && lr->buf[n - 1] == '\r') n--;
  }
  lr->buf[n] = 0;
  *line = lr->buf;
  *len = n;
  return 1;
}

static int fastq_fail(orc_run *r, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#include <stdarg.h>
static int fastq_fail(orc_run *r, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(r->err, sizeof r->err, fmt, ap);
  va_end(ap);
  return ORC_ERR_FASTQ;
}

/* stream_io_error, io.rs:213-265, for a local source (the names this reader sees are never URLs) */
static int stream_io_error(orc_run *r, const char *line_role, const io_err *e, const char *source_name) {
  unsigned long long recno = (unsigned long long)r->st.n_reads_read + 1;
  if (e->kind == IOK_UNEXPECTED_EOF)
    snprintf(r->err, sizeof r->err,
             "Local read stream ended unexpectedly while reading %s line of record %llu in %s "
             "(I/O error: %s \xE2\x80\x94 kind %s). The file may be truncated or corrupted.",
             line_role, recno, source_name, e->text, iok_name(e->kind));
  else
    snprintf(r->err, sizeof r->err, "Failed to read %s line of record %llu in %s: %s (kind %s)",
             line_role, recno, source_name, e->text, iok_name(e->kind));
  return ORC_ERR_IO;
}

/* header.chars().next().unwrap_or(' ') */
static void first_char(const char *s, size_t len, char out[5]) {
  if (len == 0) { out[0] = ' '; out[1] = 0; return; }
  unsigned char c = (unsigned char)s[0];
  size_t n = c < 0x80 ? 1 : c < 0xE0 ? 2 : c < 0xF0 ? 3 : 4;
  if (n > len) n = len;
  memcpy(out, s, n);
  out[n] = 0;
}

static char *keep_line(char **copy, size_t *cap, const char *line, size_t len) {
  if (*cap < len + 1) {
    *cap = len + 64;
    *copy = (char *)realloc(*copy, *cap);
  }
  memcpy(*copy, line, len + 1);
  return *copy;
}

/* test-only switch for the product's opt-in (shk_fastq_open_ex, SHK_FASTQ_GZIP_ALL_MEMBERS): every member of a gzip
 * file, as flate2::read::MultiGzDecoder would read it.  The reference itself reads the first member only. */
static int g_gzip_all_members = 0;
void orc_set_gzip_all_members(int on) { g_gzip_all_members = on; }

int orc_run_read_fastq(orc_run *r, const char *path, uint64_t max_reads,
                       uint64_t validate_every, int *reached_max) {
  if (reached_max) *reached_max = 0;
  line_reader *lr = (line_reader *)calloc(1, sizeof *lr);
  if (!lr) return ORC_ERR_NOMEM;
  /* open_fastq_reader, io.rs:598-625 */
  size_t plen = strlen(path);
  int has_gz_ext = (plen >= 3 && !strcmp(path + plen - 3, ".gz")) || (plen >= 5 && !strcmp(path + plen - 5, ".gzip"));
  lr->src.f = fopen(path, "rb");
  if (!lr->src.f) {
    snprintf(r->err, sizeof r->err, "Failed to open file: %s", path);
    free(lr);
    return ORC_ERR_IO;
  }
  int use_gzip = has_gz_ext;
  if (!use_gzip) { /* fill_buf() peek: the first two bytes are looked at, not consumed (they are
                    * served again below — the source may be a pipe, which cannot be rewound) */
    int c0 = fgetc(lr->src.f), c1 = c0 == EOF ? EOF : fgetc(lr->src.f);
    use_gzip = c0 == 0x1f && c1 == 0x8b;
    size_t n = 0;
    if (c0 != EOF) lr->src.inbuf[n++] = (unsigned char)c0;
    if (c1 != EOF) lr->src.inbuf[n++] = (unsigned char)c1;
    lr->src.zs.next_in = lr->src.inbuf;
    lr->src.zs.avail_in = (uInt)n;
  }
  lr->src.use_gzip = use_gzip;
  lr->src.all_members = g_gzip_all_members;
  if (use_gzip) {
    unsigned char *keep_in = lr->src.zs.next_in;
    uInt keep_n = lr->src.zs.avail_in;
    lr->src.zs.zalloc = Z_NULL;
    lr->src.zs.zfree = Z_NULL;
    lr->src.zs.opaque = Z_NULL;
    int zrc = inflateInit2(&lr->src.zs, -15);
    lr->src.zs.next_in = keep_in;
    lr->src.zs.avail_in = keep_n;
    if (zrc != Z_OK) {
      fclose(lr->src.f);
      free(lr);
      return ORC_ERR_NOMEM;
    }
    lr->src.zs_init = 1;
    gz_parse_header(&lr->src);
  }
  int rc = ORC_OK;
  char *hdr_copy = NULL, *seq_copy = NULL, *sep_copy = NULL;
  size_t hdr_cap = 0, seq_cap = 0, sep_cap = 0;
  io_err e;
  for (;;) { /* read_fastq, io.rs:282-349 */
    char *line;
    size_t len;
    int got = next_line(lr, &line, &len, &e);
    if (got == 0) break;
    if (got < 0) { rc = stream_io_error(r, "header", &e, path); break; }
    keep_line(&hdr_copy, &hdr_cap, line, len);
    size_t hdr_len = len;

    got = next_line(lr, &line, &len, &e);
    if (got == 0) { /* io.rs:291-295 */
      rc = fastq_fail(r, "Truncated FASTQ record at record %llu in %s: missing sequence line",
                      (unsigned long long)r->st.n_reads_read + 1, path);
      break;
    }
    if (got < 0) { rc = stream_io_error(r, "sequence", &e, path); break; }
    keep_line(&seq_copy, &seq_cap, line, len);
    size_t seq_len = len;

    got = next_line(lr, &line, &len, &e);
    if (got == 0) { /* io.rs:302-306 */
      rc = fastq_fail(r, "Truncated FASTQ record at record %llu in %s: missing separator line",
                      (unsigned long long)r->st.n_reads_read + 1, path);
      break;
    }
    if (got < 0) { rc = stream_io_error(r, "separator", &e, path); break; }
    keep_line(&sep_copy, &sep_cap, line, len);
    size_t sep_len = len;

    got = next_line(lr, &line, &len, &e);
    if (got == 0) { /* io.rs:313-317 */
      rc = fastq_fail(r, "Truncated FASTQ record at record %llu in %s: missing quality line",
                      (unsigned long long)r->st.n_reads_read + 1, path);
      break;
    }
    if (got < 0) { rc = stream_io_error(r, "quality", &e, path); break; }
    size_t qual_len = len;

    /* io.rs:321-332 */
    int should_validate = r->st.n_reads_read == 0 ||
                          (validate_every > 0 && r->st.n_reads_read % validate_every == 0);
    if (should_validate) { /* io.rs:161-198 */
      unsigned long long recno = (unsigned long long)r->st.n_reads_read + 1;
      char fc[5];
      if (hdr_len > 0 && hdr_copy[0] == '>') {
        rc = fastq_fail(r,
                        "Input appears to be FASTA format, not FASTQ (record %llu starts with '>'). "
                        "sharkmer requires FASTQ input with quality scores.",
                        recno);
        break;
      }
      if (!(hdr_len > 0 && hdr_copy[0] == '@')) {
        first_char(hdr_copy, hdr_len, fc);
        rc = fastq_fail(r, "FASTQ record %llu has invalid header (expected '@', got '%s'): %s",
                        recno, fc, hdr_copy);
        break;
      }
      if (!(sep_len > 0 && sep_copy[0] == '+')) {
        first_char(sep_copy, sep_len, fc);
        rc = fastq_fail(r, "FASTQ record %llu has invalid separator line (expected '+', got '%s'): %s",
                        recno, fc, sep_copy);
        break;
      }
      if (qual_len != seq_len) {
        rc = fastq_fail(r, "FASTQ record %llu has mismatched sequence (%zu) and quality (%zu) lengths",
                        recno, seq_len, qual_len);
        break;
      }
    }
    rc = orc_run_push_seq(r, (const uint8_t *)seq_copy, seq_len); /* io.rs:335-343 */
    if (rc != ORC_OK) break;
    if (max_reads > 0 && r->st.n_reads_read >= max_reads) { /* io.rs:345-348 */
      if (reached_max) *reached_max = 1;
      break;
    }
  }
  if (lr->src.zs_init) inflateEnd(&lr->src.zs);
  fclose(lr->src.f);
  free(lr->buf);
  free(lr);
  free(hdr_copy);
  free(seq_copy);
  free(sep_copy);
  return rc;
}

/* ===================================================================== */
/* writers: io.rs:1009-1014, 1051-1094                                   */
/* ===================================================================== */

int orc_run_write_histo(const orc_run *r, const char *path, const char *version) {
  if (!r->finished || r->chunks_arg == 0) return ORC_ERR_INVARIANT;
  FILE *f = fopen(path, "w");
  if (!f) return ORC_ERR_IO;
  size_t len = (size_t)r->histo_max + 2;
  fprintf(f, "# sharkmer %s k=%d chunks=%u\n", version, r->k, r->chunks_arg);
  fputs("count", f);
  for (uint32_t c = 1; c <= r->n_chunks; c++) fprintf(f, "\tchunk_%u", c);
  fputc('\n', f);
  for (size_t i = 1; i < len; i++) {
    fprintf(f, "%zu", i);
    for (uint32_t c = 0; c < r->n_chunks; c++)
      fprintf(f, "\t%llu", (unsigned long long)r->histo_vecs[(size_t)c * len + i]);
    fputc('\n', f);
  }
  fclose(f);
  return ORC_OK;
}

int orc_run_write_final_histo(const orc_run *r, const char *path, const char *version) {
  if (!r->finished || r->chunks_arg == 0) return ORC_ERR_INVARIANT;
  FILE *f = fopen(path, "w");
  if (!f) return ORC_ERR_IO;
  size_t len = (size_t)r->histo_max + 2;
  const uint64_t *last = r->histo_vecs + (size_t)(r->n_chunks - 1) * len;
  fprintf(f, "# sharkmer %s k=%d chunks=%u\n", version, r->k, r->chunks_arg);
  fputs("count\tfrequency\n", f);
  for (size_t i = 1; i < len; i++) fprintf(f, "%zu\t%llu\n", i, (unsigned long long)last[i]);
  fclose(f);
  return ORC_OK;
}

/* stats.rs:27-45,186-193 with main.rs:182-197 — RunStats as YAML, field order of the struct;
 * n_multi_kmers / n_singleton_kmers only when chunks > 0; pcr_results omitted when empty. */
int orc_run_write_stats_yaml(const orc_run *r, const char *path, const char *version,
                             const char *command, const char *sample, uint64_t peak_memory_bytes) {
  if (!r->finished) return ORC_ERR_INVARIANT;
  FILE *f = fopen(path, "w");
  if (!f) return ORC_ERR_IO;
  fprintf(f, "sharkmer_version: %s\n", version);
  fprintf(f, "command: %s\n", command);
  fprintf(f, "sample: %s\n", sample);
  fprintf(f, "kmer_length: %d\n", r->k);
  fprintf(f, "chunks: %u\n", r->chunks_arg);
  fprintf(f, "n_reads_read: %llu\n", (unsigned long long)r->st.n_reads_read);
  fprintf(f, "n_bases_read: %llu\n", (unsigned long long)r->st.n_bases_read);
  fprintf(f, "n_subreads_ingested: %llu\n", (unsigned long long)r->st.n_reads_ingested);
  fprintf(f, "n_bases_ingested: %llu\n", (unsigned long long)r->st.n_bases_ingested);
  fprintf(f, "n_kmers: %llu\n", (unsigned long long)r->st.n_kmers_ingested);
  if (r->chunks_arg > 0) {
    uint64_t s1 = r->st.n_singleton_kmers;
    uint64_t multi = r->st.n_kmers_ingested >= s1 ? r->st.n_kmers_ingested - s1 : 0; /* saturating_sub */
    fprintf(f, "n_multi_kmers: %llu\n", (unsigned long long)multi);
    fprintf(f, "n_singleton_kmers: %llu\n", (unsigned long long)s1);
  }
  fprintf(f, "peak_memory_bytes: %llu\n", (unsigned long long)peak_memory_bytes);
  fclose(f);
  return ORC_OK;
}

/* find_oligos_in_kmers, src/pcr/primers.rs:163-226 — the consumer-side scan of the merged
 * table.  oligos: 2-bit values of oligo_len bases.  Writes the matching
 * (k-mer, count) pairs (reverse-complemented for the rc orientation, :219-222); returns n. */
size_t orc_find_oligos(const orc_counts *c, const uint64_t *oligos, size_t n_oligos, int oligo_len,
                       uint32_t min_count, uint64_t *out_kmers, uint32_t *out_counts) {
  const int k = c->k;
  uint64_t mask = 0;
  for (int i = 0; i < 2 * oligo_len; i++) mask = (mask << 1) | 1; /* :195-198 */
  mask <<= 2 * k - 2 * oligo_len;
  const uint64_t rc_mask = (1ull << (2 * oligo_len)) - 1; /* :203 */
  size_t n = 0;
  for (size_t i = 0; i < c->kmers.cap; i++) {
    uint64_t kmer = c->kmers.keys[i];
    if (kmer == MAP_EMPTY) continue;
    uint32_t count = c->kmers.vals[i];
    if (count < min_count) continue; /* :213 */
    int hit = 0;
    for (size_t j = 0; j < n_oligos && !hit; j++)
      if ((oligos[j] << (2 * (k - oligo_len))) == (kmer & mask)) hit = 1; /* :214 */
    if (!hit)
      for (size_t j = 0; j < n_oligos && !hit; j++)
        if (orc_revcomp_kmer(oligos[j], oligo_len) == (kmer & rc_mask)) hit = 2; /* :216 */
    if (hit) {
      out_kmers[n] = hit == 1 ? kmer : orc_revcomp_kmer(kmer, k);
      out_counts[n] = count;
      n++;
    }
  }
  return n;
}

/* PrimerReadFilter::matches, src/pcr/read_filter.rs:43-49: kmers_from_ascii must succeed (an
 * invalid byte anywhere in the read → false) and some k-mer must be in the primer set (the union
 * of two KmerCounts, :24-41; here: any orc_counts holding the canonical primer k-mers). */
int orc_filter_matches(const orc_counts *primers, const uint8_t *seq, size_t len) {
  const int k = primers->k;
  if (len == 0) return 0;
  uint64_t *km = (uint64_t *)malloc(sizeof(uint64_t) * (len ? len : 1));
  if (!km) return 0;
  size_t n = 0;
  uint8_t bad = 0;
  int hit = 0;
  if (orc_kmers_from_ascii(seq, len, k, km, &n, &bad) == ORC_OK)
    for (size_t i = 0; i < n && !hit; i++) hit = orc_counts_contains(primers, km[i]);
  free(km);
  return hit;
}

/* ---- probe-set counting (test infrastructure for the full-size parity checks) ------------------------
 * For reads [0, n_seqs) of a batch whose first read has global index first_read_index: the k-mers of every
 * read come from orc_kmers_from_ascii (encoding.rs:332-371), a read's chunk lane is
 * (global read index / 1000) % n_lanes (io.rs:340-361), and every occurrence of a k-mer that is in
 * `probes` (sorted ascending, distinct) adds one to counts[lane * n_probes + its index] (u64: the caller
 * clamps like saturating_add, counting.rs:82-85).  Thread-safe on disjoint `counts`; callers shard reads
 * over threads with one counts array each.  Returns ORC_OK or the extractor's error. */
int orc_probe_count(const uint8_t *bases, const uint64_t *offsets, uint64_t n_seqs, int k,
                    uint64_t first_read_index, uint32_t n_lanes, const uint64_t *probes, uint64_t n_probes,
                    uint64_t *counts) {
  if (n_probes == 0 || n_seqs == 0) return ORC_OK;
  /* open-addressing index of the probes (power-of-two table, load ≤ 1/4, stays in cache) */
  uint64_t cap = 16;
  while (cap < n_probes * 4) cap <<= 1;
  uint64_t *tk = (uint64_t *)malloc(cap * sizeof(uint64_t));
  uint32_t *ti = (uint32_t *)malloc(cap * sizeof(uint32_t));
  if (!tk || !ti) {
    free(tk);
    free(ti);
    return ORC_ERR_NOMEM;
  }
  memset(tk, 0xFF, cap * sizeof(uint64_t));
  for (uint64_t i = 0; i < n_probes; i++) {
    uint64_t h = mix64(probes[i]) & (cap - 1);
    while (tk[h] != ~0ull) h = (h + 1) & (cap - 1);
    tk[h] = probes[i];
    ti[h] = (uint32_t)i;
  }
  /* a one-word-per-lookup Bloom filter in front of it (512 KiB: stays in L2; almost every k-mer of the
   * reads is no probe and never touches the index) */
  const uint64_t bloom_words = 1ull << 16;
  uint64_t *bloom = (uint64_t *)calloc(bloom_words, sizeof(uint64_t));
  if (!bloom) {
    free(tk);
    free(ti);
    return ORC_ERR_NOMEM;
  }
  for (uint64_t i = 0; i < n_probes; i++) {
    const uint64_t h = mix64(probes[i]);
    bloom[(h >> 40) & (bloom_words - 1)] |= (1ull << (h & 63)) | (1ull << ((h >> 6) & 63));
  }
  size_t km_cap = 1024;
  uint64_t *km = (uint64_t *)malloc(km_cap * sizeof(uint64_t));
  int rc = km ? ORC_OK : ORC_ERR_NOMEM;
  for (uint64_t r = 0; r < n_seqs && rc == ORC_OK; r++) {
    const size_t len = (size_t)(offsets[r + 1] - offsets[r]);
    if (len > km_cap) {
      km_cap = len * 2;
      uint64_t *nk = (uint64_t *)realloc(km, km_cap * sizeof(uint64_t));
      if (!nk) {
        rc = ORC_ERR_NOMEM;
        break;
      }
      km = nk;
    }
    size_t n = 0;
    uint8_t bad = 0;
    rc = orc_kmers_from_ascii(bases + offsets[r], len, k, km, &n, &bad);
    if (rc != ORC_OK) break;
    const uint32_t lane = (uint32_t)(((first_read_index + r) / 1000) % n_lanes);
    for (size_t j = 0; j < n; j++) {
      const uint64_t hh = mix64(km[j]);
      const uint64_t need = (1ull << (hh & 63)) | (1ull << ((hh >> 6) & 63));
      if ((bloom[(hh >> 40) & (bloom_words - 1)] & need) != need) continue;
      uint64_t h = hh & (cap - 1);
      while (tk[h] != ~0ull) {
        if (tk[h] == km[j]) {
          counts[(uint64_t)lane * n_probes + ti[h]]++;
          break;
        }
        h = (h + 1) & (cap - 1);
      }
    }
  }
  free(km);
  free(tk);
  free(ti);
  free(bloom);
  return rc;
}
