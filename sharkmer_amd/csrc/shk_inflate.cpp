// shk_inflate.cpp — see shk_inflate.h.  Table-driven DEFLATE decoder: 64-bit bit buffer refilled by one unaligned
// load, an 11-bit first-level table for literals/lengths and an 8-bit one for distances (second-level tables behind
// them for longer codes), up to three literals per refill, wide overshooting match copies.  RFC 1951 is the
// specification followed; which malformed streams are refused follows miniz_oxide's rules where RFC 1951 leaves a
// choice (a code set that is neither complete nor a single code is refused), since that is the reference's decoder.
#include <algorithm>
#include "shk_inflate.h"

#include <immintrin.h>

#include <cstring>

namespace shk {

const char *io_kind_name(IoKind k) {
  switch (k) {
    case IO_UNEXPECTED_EOF: return "UnexpectedEof";
    case IO_INVALID_INPUT: return "InvalidInput";
    case IO_INVALID_DATA: return "InvalidData";
    case IO_OTHER: return "Other";
    default: return "";
  }
}

namespace {

constexpr uint32_t E_LITERAL = 1u << 31;
constexpr uint32_t E_EXC = 1u << 15;      // anything that is not a literal, a length or a distance
constexpr uint32_t E_SUB = 1u << 14;      // with E_EXC: pointer to a second-level table
constexpr uint32_t E_EOB = 1u << 13;      // with E_EXC: end of block
constexpr uint32_t E_INVALID = E_EXC;     // E_EXC alone: no code word / a symbol that must not occur
constexpr int LIT_BITS = 11, DIST_BITS = 8, PRE_BITS = 7;
constexpr uint32_t LIT_CAP = 2048 + 1024, DIST_CAP = 256 + 512;

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t PRE_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint32_t payload_litlen(uint32_t s) {
  if (s < 256) return E_LITERAL | (s << 16);
  if (s == 256) return E_EXC | E_EOB;
  if (s < 286) return ((uint32_t)LEN_BASE[s - 257] << 16) | ((uint32_t)LEN_EXTRA[s - 257] << 8);
  return E_INVALID;
}
inline uint32_t payload_dist(uint32_t s) {
  if (s < 30) return ((uint32_t)DIST_BASE[s] << 16) | ((uint32_t)DIST_EXTRA[s] << 8);
  return E_INVALID;
}
inline uint32_t payload_pre(uint32_t s) { return s << 16; }

inline uint32_t bitrev(uint32_t v, int n) {
  uint32_t r = 0;
  for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
  return r;
}

// Canonical Huffman code of lens[0..n) → two-level decode table.  Entry = payload | bits to drop (the code word's
// length in the first level, its length beyond the first level in a second-level table).  false: over-subscribed, or
// incomplete with more than one code word in use.
template <class Payload>
bool build_table(const uint8_t *lens, uint32_t n, int tb, uint32_t cap, uint32_t *table, Payload payload) {
  uint32_t count[16] = {0};
  uint32_t used = 0;
  for (uint32_t s = 0; s < n; ++s) {
    ++count[lens[s]];
    used += lens[s] != 0;
  }
  int left = 1;
  for (int l = 1; l <= 15; ++l) {
    left <<= 1;
    left -= (int)count[l];
    if (left < 0) return false;
  }
  if (left > 0 && used > 1) return false;
  const uint32_t prim = 1u << tb;
  for (uint32_t i = 0; i < prim; ++i) table[i] = E_INVALID;
  if (used == 0) return true;
  uint32_t next_code[16];
  {
    uint32_t code = 0;
    count[0] = 0;
    for (int l = 1; l <= 15; ++l) {
      code = (code + count[l - 1]) << 1;
      next_code[l] = code;
    }
  }
  uint32_t rev_of[320];
  uint8_t sub_max[2048];
  bool any_long = false;
  for (uint32_t s = 0; s < n; ++s) {
    const int l = lens[s];
    if (!l) continue;
    const uint32_t rev = bitrev(next_code[l]++, l);
    rev_of[s] = rev;
    if (l <= tb) {
      const uint32_t e = payload(s) | (uint32_t)l;
      for (uint32_t i = rev; i < prim; i += 1u << l) table[i] = e;
    } else {
      if (!any_long) {
        memset(sub_max, 0, prim);
        any_long = true;
      }
      uint8_t &m = sub_max[rev & (prim - 1)];
      if (l > m) m = (uint8_t)l;
    }
  }
  if (!any_long) return true;
  uint32_t next_free = prim;
  for (uint32_t s = 0; s < n; ++s) {
    const int l = lens[s];
    if (l <= tb) continue;
    const uint32_t rev = rev_of[s], pfx = rev & (prim - 1);
    uint32_t &pe = table[pfx];
    const uint32_t sub_bits = (uint32_t)sub_max[pfx] - (uint32_t)tb;
    if (!(pe & E_SUB)) {  // first code word behind this prefix: lay out its second-level table
      if (next_free + (1u << sub_bits) > cap) return false;
      for (uint32_t i = 0; i < (1u << sub_bits); ++i) table[next_free + i] = E_INVALID;
      pe = E_EXC | E_SUB | (next_free << 16) | (sub_bits << 8) | (uint32_t)tb;
      next_free += 1u << sub_bits;
    }
    const uint32_t start = pe >> 16;
    const uint32_t e = payload(s) | (uint32_t)(l - tb);
    for (uint32_t i = rev >> tb; i < (1u << sub_bits); i += 1u << (l - tb)) table[start + i] = e;
  }
  return true;
}

inline uint64_t load64(const uint8_t *p) {
  uint64_t v;
  memcpy(&v, p, 8);
  return v;
}

}  // namespace

void Inflater::reset(const uint8_t *p, const uint8_t *e) {
  in = p;
  in_end = e;
  bitbuf = 0;
  bitcnt = 0;
  state = 0;
  last_block = false;
  stored_left = 0;
  total_out = 0;
}

// The header of the block at the current bit position: BFINAL, BTYPE, and for a Huffman block its decode tables.
// INF_OUTPUT_FULL stands for "go on" here (state says with what: 1 a stored block of stored_left bytes, 2 symbols).
InflateStatus Inflater::read_block_header() {
  const uint8_t *ip = in;
  uint64_t bb = bitbuf;
  uint32_t bc = bitcnt;
  auto need = [&](uint32_t n) -> bool {
    while (bc < n) {
      if (ip == in_end) return false;
      bb |= (uint64_t)*ip++ << bc;
      bc += 8;
    }
    return true;
  };
  auto fill_all = [&]() {
    while (bc <= 56 && ip != in_end) {
      bb |= (uint64_t)*ip++ << bc;
      bc += 8;
    }
  };
  struct Sync {  // (the members follow the locals on every way out)
    Inflater *self;
    const uint8_t *&ip;
    uint64_t &bb;
    uint32_t &bc;
    ~Sync() { self->in = ip, self->bitbuf = bb, self->bitcnt = bc; }
  } sync{this, ip, bb, bc};
      if (!need(3)) return INF_TRUNCATED;
      last_block = bb & 1;
      const uint32_t type = (uint32_t)(bb >> 1) & 3;
      bb >>= 3;
      bc -= 3;
      if (type == 0) {
        bb >>= bc & 7;
        bc -= bc & 7;
        if (!need(32)) return INF_TRUNCATED;
        const uint32_t len = (uint32_t)bb & 0xFFFF, nlen = (uint32_t)(bb >> 16) & 0xFFFF;
        bb >>= 32;
        bc -= 32;
        if ((len ^ 0xFFFF) != nlen) return INF_CORRUPT;
        ip -= bc >> 3;  // whole bytes still in the bit buffer are the block's first data bytes
        bb = 0;
        bc = 0;
        stored_left = len;
        state = 1;
      } else if (type == 1) {
        uint8_t lens[288 + 32];
        for (int i = 0; i < 144; ++i) lens[i] = 8;
        for (int i = 144; i < 256; ++i) lens[i] = 9;
        for (int i = 256; i < 280; ++i) lens[i] = 7;
        for (int i = 280; i < 288; ++i) lens[i] = 8;
        for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
        build_table(lens, 288, LIT_BITS, LIT_CAP, litlen, payload_litlen);
        build_table(lens + 288, 32, DIST_BITS, DIST_CAP, dist, payload_dist);
        state = 2;
      } else if (type == 2) {
        if (!need(14)) return INF_TRUNCATED;
        const uint32_t hlit = ((uint32_t)bb & 31) + 257, hdist = ((uint32_t)(bb >> 5) & 31) + 1, hclen = ((uint32_t)(bb >> 10) & 15) + 4;
        bb >>= 14;
        bc -= 14;
        uint8_t pre_lens[19] = {0};
        for (uint32_t i = 0; i < hclen; ++i) {
          if (!need(3)) return INF_TRUNCATED;
          pre_lens[PRE_ORDER[i]] = (uint8_t)(bb & 7);
          bb >>= 3;
          bc -= 3;
        }
        uint32_t pre[1u << PRE_BITS];
        if (!build_table(pre_lens, 19, PRE_BITS, 1u << PRE_BITS, pre, payload_pre)) return INF_CORRUPT;
        uint8_t lens[288 + 32 + 140];
        const uint32_t total = hlit + hdist;
        uint32_t i = 0;
        while (i < total) {
          fill_all();
          const uint32_t e = pre[bb & ((1u << PRE_BITS) - 1)];
          if (e & E_EXC) {  // no such code word — unless the bits are simply not there
            return ip == in_end && bc < 7 ? INF_TRUNCATED : INF_CORRUPT;
          }
          const uint32_t cl = e & 0xFF, sym = e >> 16;
          uint32_t extra_n = sym < 16 ? 0 : sym == 16 ? 2 : sym == 17 ? 3 : 7;
          if (bc < cl + extra_n) return INF_TRUNCATED;
          bb >>= cl;
          bc -= cl;
          if (sym < 16) {
            lens[i++] = (uint8_t)sym;
            continue;
          }
          const uint32_t extra = (uint32_t)bb & ((1u << extra_n) - 1);
          bb >>= extra_n;
          bc -= extra_n;
          uint32_t rep;
          uint8_t v = 0;
          if (sym == 16) {
            if (i == 0) return INF_CORRUPT;
            v = lens[i - 1];
            rep = 3 + extra;
          } else if (sym == 17) {
            rep = 3 + extra;
          } else {
            rep = 11 + extra;
          }
          if (i + rep > total) return INF_CORRUPT;
          memset(lens + i, v, rep);
          i += rep;
        }
        uint8_t ll[288] = {0}, dl[32] = {0};
        memcpy(ll, lens, hlit);
        memcpy(dl, lens + hlit, hdist);
        hdr_plausible = hlit <= 286 && hdist <= 30 && ll[256] != 0;
        if (!build_table(ll, 288, LIT_BITS, LIT_CAP, litlen, payload_litlen)) return INF_CORRUPT;
        if (!build_table(dl, 32, DIST_BITS, DIST_CAP, dist, payload_dist)) return INF_CORRUPT;
        state = 2;
      } else {
        return INF_CORRUPT;
      }
  return INF_OUTPUT_FULL;
}

InflateStatus Inflater::run(uint8_t *out_base, size_t *out_pos, size_t out_cap) {
  uint8_t *out = out_base + *out_pos;
  uint8_t *const out_end = out_base + out_cap;
  uint8_t *const out_start = out;
  // how far back a match may reach: never beyond the buffer's start — and never beyond the START OF THE STREAM where
  // the caller decodes several streams into one buffer (all-members mode: a member starts with empty history)
  uint8_t *const hist_first = out_start - (size_t)std::min<uint64_t>(history_bytes, (uint64_t)*out_pos);
  const uint8_t *ip = in;
  uint64_t bb = bitbuf;
  uint32_t bc = bitcnt;
  InflateStatus result = INF_OUTPUT_FULL;

  auto fill_all = [&]() {
    while (bc <= 56 && ip != in_end) {
      bb |= (uint64_t)*ip++ << bc;
      bc += 8;
    }
  };
#define SHK_DONE(st)     \
  do {                   \
    result = (st);       \
    goto finished;       \
  } while (0)

  stopped_between_blocks = false;
  for (;;) {
    if (state == 3) SHK_DONE(INF_STREAM_END);
    if (state == 0) {  // ---- block header ---------------------------------------------------------------------
      in = ip, bitbuf = bb, bitcnt = bc;
      if (stop_origin && bit_position(stop_origin) >= stop_bit) {
        stopped_between_blocks = true;
        SHK_DONE(INF_OUTPUT_FULL);
      }
      const InflateStatus hs = read_block_header();
      ip = in, bb = bitbuf, bc = bitcnt;
      if (hs != INF_OUTPUT_FULL) SHK_DONE(hs);
    }
    if (state == 1) {  // ---- stored block ---------------------------------------------------------------------
      while (stored_left) {
        const size_t room = (size_t)(out_end - out), have = (size_t)(in_end - ip);
        if (room == 0) SHK_DONE(INF_OUTPUT_FULL);
        if (have == 0) SHK_DONE(INF_TRUNCATED);
        size_t n = stored_left;
        if (n > room) n = room;
        if (n > have) n = have;
        memcpy(out, ip, n);
        out += n;
        ip += n;
        stored_left -= (uint32_t)n;
      }
      state = last_block ? 3 : 0;
      continue;
    }
    // ---- Huffman block ----------------------------------------------------------------------------------------
    for (;;) {
      if ((size_t)(out_end - out) < OUT_SLACK) SHK_DONE(INF_OUTPUT_FULL);
      uint32_t e;
      if (in_end - ip >= 16) {
        // fast path: whole-word refills; bits above bc in bb are the stream's own next bits (idempotent under the OR)
#define SHK_REFILL()                                    \
  do {                                                  \
    bb |= load64(ip) << bc;                             \
    const uint32_t nby_ = (63 - bc) >> 3;               \
    ip += nby_;                                         \
    bc += nby_ * 8;                                     \
  } while (0)
        SHK_REFILL();
        e = litlen[bb & ((1u << LIT_BITS) - 1)];
        if (e & E_LITERAL) {
          bb >>= (uint8_t)e;
          bc -= (uint8_t)e;
          *out++ = (uint8_t)(e >> 16);
          e = litlen[bb & ((1u << LIT_BITS) - 1)];
          if (e & E_LITERAL) {
            bb >>= (uint8_t)e;
            bc -= (uint8_t)e;
            *out++ = (uint8_t)(e >> 16);
            e = litlen[bb & ((1u << LIT_BITS) - 1)];
            if (e & E_LITERAL) {
              bb >>= (uint8_t)e;
              bc -= (uint8_t)e;
              *out++ = (uint8_t)(e >> 16);
              continue;
            }
          }
          if (bc < 48) SHK_REFILL();
        }
        if (e & E_EXC) {
          if (e & E_SUB) {
            bb >>= LIT_BITS;
            bc -= LIT_BITS;
            e = litlen[(e >> 16) + ((uint32_t)bb & ((1u << ((e >> 8) & 15)) - 1))];
            if (e & E_LITERAL) {
              bb >>= (uint8_t)e;
              bc -= (uint8_t)e;
              *out++ = (uint8_t)(e >> 16);
              continue;
            }
          }
          if (e & E_EXC) {
            if (!(e & E_EOB)) SHK_DONE(INF_CORRUPT);
            bb >>= (uint8_t)e;
            bc -= (uint8_t)e;
            break;
          }
        }
        {
          bb >>= (uint8_t)e;
          const uint32_t leb = (e >> 8) & 15;
          const uint32_t length = (e >> 16) + ((uint32_t)bb & ((1u << leb) - 1));
          bb >>= leb;
          bc -= (uint8_t)e + leb;
          uint32_t d = dist[bb & ((1u << DIST_BITS) - 1)];
          if (d & E_EXC) {
            if (!(d & E_SUB)) SHK_DONE(INF_CORRUPT);
            bb >>= DIST_BITS;
            bc -= DIST_BITS;
            d = dist[(d >> 16) + ((uint32_t)bb & ((1u << ((d >> 8) & 15)) - 1))];
            if (d & E_EXC) SHK_DONE(INF_CORRUPT);
          }
          bb >>= (uint8_t)d;
          const uint32_t deb = (d >> 8) & 15;
          const size_t distance = (d >> 16) + ((uint32_t)bb & ((1u << deb) - 1));
          bb >>= deb;
          bc -= (uint8_t)d + deb;
          if (distance > (size_t)(out - hist_first)) SHK_DONE(INF_CORRUPT);
          const uint8_t *src = out - distance;
          uint8_t *const end = out + length;
          if (distance >= 16) {
            do {
              memcpy(out, src, 16);
              out += 16;
              src += 16;
            } while (out < end);
          } else if (distance == 1) {
            memset(out, *src, length);
          } else if (distance >= 8) {
            do {
              memcpy(out, src, 8);
              out += 8;
              src += 8;
            } while (out < end);
          } else {
            do {
              *out++ = *src++;
            } while (out < end);
          }
          out = end;
        }
        continue;
      }
      // careful path (the last bytes of the input): a symbol is taken only when ALL its bits are there, so that
      // everything a truncated stream still says has been written when it runs out
      fill_all();
      uint64_t b2 = bb;
      uint32_t c2 = bc;
      auto drop = [&](uint32_t n) -> bool {
        if (c2 < n) return false;
        b2 >>= n;
        c2 -= n;
        return true;
      };
      const bool at_end = ip == in_end;
      e = litlen[b2 & ((1u << LIT_BITS) - 1)];
      if ((e & E_EXC) && (e & E_SUB)) {
        if (!drop(LIT_BITS)) SHK_DONE(at_end ? INF_TRUNCATED : INF_CORRUPT);
        e = litlen[(e >> 16) + ((uint32_t)b2 & ((1u << ((e >> 8) & 15)) - 1))];
      }
      // (a code word that is not there — or a symbol that must not occur — decoded from bits the input really has is
      // corruption; decoded with the help of the zero padding behind the last byte it is only the input ending)
      if ((e & E_EXC) && !(e & E_EOB)) SHK_DONE(at_end && c2 < ((e & 0xFF) ? (e & 0xFF) : 15u) ? INF_TRUNCATED : INF_CORRUPT);
      if (!drop(e & 0xFF)) SHK_DONE(INF_TRUNCATED);
      if (e & E_LITERAL) {
        *out++ = (uint8_t)(e >> 16);
        bb = b2;
        bc = c2;
        continue;
      }
      if (e & E_EXC) {  // end of block
        bb = b2;
        bc = c2;
        break;
      }
      const uint32_t leb = (e >> 8) & 15;
      const uint32_t length = (e >> 16) + ((uint32_t)b2 & ((1u << leb) - 1));
      if (!drop(leb)) SHK_DONE(INF_TRUNCATED);
      uint32_t d = dist[b2 & ((1u << DIST_BITS) - 1)];
      if ((d & E_EXC) && (d & E_SUB)) {
        if (!drop(DIST_BITS)) SHK_DONE(at_end ? INF_TRUNCATED : INF_CORRUPT);
        d = dist[(d >> 16) + ((uint32_t)b2 & ((1u << ((d >> 8) & 15)) - 1))];
      }
      if (d & E_EXC) SHK_DONE(at_end && c2 < ((d & 0xFF) ? (d & 0xFF) : 15u) ? INF_TRUNCATED : INF_CORRUPT);
      if (!drop(d & 0xFF)) SHK_DONE(INF_TRUNCATED);
      const uint32_t deb = (d >> 8) & 15;
      const size_t distance = (d >> 16) + ((uint32_t)b2 & ((1u << deb) - 1));
      if (!drop(deb)) SHK_DONE(INF_TRUNCATED);
      if (distance > (size_t)(out - hist_first)) SHK_DONE(INF_CORRUPT);
      bb = b2;
      bc = c2;
      const uint8_t *src = out - distance;
      for (uint32_t i = 0; i < length; ++i) out[i] = src[i];
      out += length;
    }
    state = last_block ? 3 : 0;
  }
finished:
  in = ip;
  bitbuf = bb;
  bitcnt = bc;
  total_out += (uint64_t)(out - out_start);
  *out_pos = (size_t)(out - out_base);
  return result;
#undef SHK_DONE
#undef SHK_REFILL
}

void Inflater::seek(const uint8_t *origin, const uint8_t *e, uint64_t bit) {
  reset(origin + (bit >> 3), e);
  const uint32_t skip = (uint32_t)(bit & 7);
  if (skip && in < in_end) {
    bitbuf = (uint64_t)*in++ >> skip;
    bitcnt = 8 - skip;
  }
}

InflateStatus Inflater::run_symbols(uint16_t *out, size_t *out_pos, size_t out_cap, const uint8_t *origin, uint64_t stop_at, bool *between_blocks) {
  size_t pos = *out_pos;
  *between_blocks = false;
  InflateStatus result = INF_OUTPUT_FULL;
  const uint8_t *ip = in;
  uint64_t bb = bitbuf;
  uint32_t bc = bitcnt;
  auto refill = [&]() {
    if (in_end - ip >= 8) {
      bb |= load64(ip) << bc;
      const uint32_t nby = (63 - bc) >> 3;
      ip += nby;
      bc += nby * 8;
    } else {
      while (bc <= 56 && ip != in_end) {
        bb |= (uint64_t)*ip++ << bc;
        bc += 8;
      }
    }
  };
#define SHK_SYM_DONE(st) \
  do {                   \
    result = (st);       \
    goto sym_finished;   \
  } while (0)
  for (;;) {
    if (state == 3) SHK_SYM_DONE(INF_STREAM_END);
    if (state == 0) {
      in = ip, bitbuf = bb, bitcnt = bc;
      if (bit_position(origin) >= stop_at) {
        *between_blocks = true;
        SHK_SYM_DONE(INF_OUTPUT_FULL);
      }
      const InflateStatus hs = read_block_header();
      ip = in, bb = bitbuf, bc = bitcnt;
      if (hs != INF_OUTPUT_FULL) SHK_SYM_DONE(hs);
    }
    if (state == 1) {
      while (stored_left) {
        if (pos == out_cap) SHK_SYM_DONE(INF_OUTPUT_FULL);
        if (ip == in_end) SHK_SYM_DONE(INF_TRUNCATED);
        size_t n = stored_left;
        if (n > out_cap - pos) n = out_cap - pos;
        if (n > (size_t)(in_end - ip)) n = (size_t)(in_end - ip);
        for (size_t i = 0; i < n; ++i) out[pos + i] = ip[i];
        pos += n;
        ip += n;
        stored_left -= (uint32_t)n;
      }
      state = last_block ? 3 : 0;
      continue;
    }
    for (;;) {  // symbols of a Huffman block
      if (out_cap - pos < OUT_SLACK) SHK_SYM_DONE(INF_OUTPUT_FULL);
      if (in_end - ip >= 16) {
        // away from the input's end every bit a symbol can need is in the buffer after one refill (run()'s fast path,
        // with 16-bit symbols going out)
        {
          bb |= load64(ip) << bc;
          const uint32_t nby = (63 - bc) >> 3;
          ip += nby;
          bc += nby * 8;
        }
        uint32_t e = litlen[bb & ((1u << LIT_BITS) - 1)];
        if (e & E_LITERAL) {
          bb >>= (uint8_t)e;
          bc -= (uint8_t)e;
          out[pos++] = (uint16_t)((e >> 16) & 0xFF);
          e = litlen[bb & ((1u << LIT_BITS) - 1)];
          if (e & E_LITERAL) {
            bb >>= (uint8_t)e;
            bc -= (uint8_t)e;
            out[pos++] = (uint16_t)((e >> 16) & 0xFF);
            e = litlen[bb & ((1u << LIT_BITS) - 1)];
            if (e & E_LITERAL) {
              bb >>= (uint8_t)e;
              bc -= (uint8_t)e;
              out[pos++] = (uint16_t)((e >> 16) & 0xFF);
              continue;
            }
          }
          if (bc < 48) {
            bb |= load64(ip) << bc;
            const uint32_t nby = (63 - bc) >> 3;
            ip += nby;
            bc += nby * 8;
          }
        }
        if (e & E_EXC) {
          if (e & E_SUB) {
            bb >>= LIT_BITS;
            bc -= LIT_BITS;
            e = litlen[(e >> 16) + ((uint32_t)bb & ((1u << ((e >> 8) & 15)) - 1))];
            if (e & E_LITERAL) {
              bb >>= (uint8_t)e;
              bc -= (uint8_t)e;
              out[pos++] = (uint16_t)((e >> 16) & 0xFF);
              continue;
            }
          }
          if (e & E_EXC) {
            if (!(e & E_EOB)) SHK_SYM_DONE(INF_CORRUPT);
            bb >>= (uint8_t)e;
            bc -= (uint8_t)e;
            break;
          }
        }
        bb >>= (uint8_t)e;
        const uint32_t leb = (e >> 8) & 15;
        const uint32_t length = (e >> 16) + ((uint32_t)bb & ((1u << leb) - 1));
        bb >>= leb;
        bc -= (uint8_t)e + leb;
        uint32_t d = dist[bb & ((1u << DIST_BITS) - 1)];
        if (d & E_EXC) {
          if (!(d & E_SUB)) SHK_SYM_DONE(INF_CORRUPT);
          bb >>= DIST_BITS;
          bc -= DIST_BITS;
          d = dist[(d >> 16) + ((uint32_t)bb & ((1u << ((d >> 8) & 15)) - 1))];
          if (d & E_EXC) SHK_SYM_DONE(INF_CORRUPT);
        }
        bb >>= (uint8_t)d;
        const uint32_t deb = (d >> 8) & 15;
        const size_t distance = (d >> 16) + ((uint32_t)bb & ((1u << deb) - 1));
        bb >>= deb;
        bc -= (uint8_t)d + deb;
        if (distance <= pos) {
          const uint16_t *src = out + pos - distance;
          uint16_t *dst = out + pos, *const dend = dst + length;
          if (distance >= 8) {  // eight symbols (16 bytes) at a time, overshooting into the slack
            do {
              memcpy(dst, src, 16);
              dst += 8;
              src += 8;
            } while (dst < dend);
          } else {
            do {
              *dst++ = *src++;
            } while (dst < dend);
          }
        } else {  // reaches in front of the entry point: bytes of the unknown window (distance ≤ 32768 always)
          for (uint32_t i = 0; i < length; ++i) {
            const size_t at = pos + i;
            out[at] = at >= distance ? out[at - distance] : (uint16_t)(256 + 32768 - (distance - at));
          }
        }
        pos += length;
        continue;
      }
      // the input's last bytes: every symbol taken only when all its bits are there
      refill();
      uint64_t b2 = bb;
      uint32_t c2 = bc;
      auto drop = [&](uint32_t n) -> bool {
        if (c2 < n) return false;
        b2 >>= n;
        c2 -= n;
        return true;
      };
      const bool at_end = ip == in_end;
      uint32_t e = litlen[b2 & ((1u << LIT_BITS) - 1)];
      if ((e & E_EXC) && (e & E_SUB)) {
        if (!drop(LIT_BITS)) SHK_SYM_DONE(at_end ? INF_TRUNCATED : INF_CORRUPT);
        e = litlen[(e >> 16) + ((uint32_t)b2 & ((1u << ((e >> 8) & 15)) - 1))];
      }
      if ((e & E_EXC) && !(e & E_EOB)) SHK_SYM_DONE(at_end && c2 < ((e & 0xFF) ? (e & 0xFF) : 15u) ? INF_TRUNCATED : INF_CORRUPT);
      if (!drop(e & 0xFF)) SHK_SYM_DONE(INF_TRUNCATED);
      if (e & E_LITERAL) {
        out[pos++] = (uint16_t)((e >> 16) & 0xFF);
        bb = b2;
        bc = c2;
        continue;
      }
      if (e & E_EXC) {  // end of block
        bb = b2;
        bc = c2;
        break;
      }
      const uint32_t leb = (e >> 8) & 15;
      const uint32_t length = (e >> 16) + ((uint32_t)b2 & ((1u << leb) - 1));
      if (!drop(leb)) SHK_SYM_DONE(INF_TRUNCATED);
      uint32_t d = dist[b2 & ((1u << DIST_BITS) - 1)];
      if ((d & E_EXC) && (d & E_SUB)) {
        if (!drop(DIST_BITS)) SHK_SYM_DONE(at_end ? INF_TRUNCATED : INF_CORRUPT);
        d = dist[(d >> 16) + ((uint32_t)b2 & ((1u << ((d >> 8) & 15)) - 1))];
      }
      if (d & E_EXC) SHK_SYM_DONE(at_end && c2 < ((d & 0xFF) ? (d & 0xFF) : 15u) ? INF_TRUNCATED : INF_CORRUPT);
      if (!drop(d & 0xFF)) SHK_SYM_DONE(INF_TRUNCATED);
      const uint32_t deb = (d >> 8) & 15;
      const size_t distance = (d >> 16) + ((uint32_t)b2 & ((1u << deb) - 1));
      if (!drop(deb)) SHK_SYM_DONE(INF_TRUNCATED);
      bb = b2;
      bc = c2;
      if (distance <= pos) {
        const uint16_t *src = out + pos - distance;
        for (uint32_t i = 0; i < length; ++i) out[pos + i] = src[i];
      } else {  // reaches in front of the entry point: bytes of the unknown window (distance ≤ 32768 always)
        for (uint32_t i = 0; i < length; ++i) {
          const size_t at = pos + i;
          out[at] = at >= distance ? out[at - distance] : (uint16_t)(256 + 32768 - (distance - at));
        }
      }
      pos += length;
    }
    state = last_block ? 3 : 0;
  }
sym_finished:
  in = ip;
  bitbuf = bb;
  bitcnt = bc;
  total_out += pos - *out_pos;
  *out_pos = pos;
  return result;
#undef SHK_SYM_DONE
}

// ---- CRC-32 ---------------------------------------------------------------------------------------------------
namespace {
struct CrcTables {
  uint32_t t[8][256];
  uint32_t x2n[32];  // x^(2^n) mod P, reflected
  static uint32_t mulmod(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
      if (a & m) {
        p ^= b;
        if ((a & (m - 1)) == 0) break;
      }
      m >>= 1;
      b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
    }
    return p;
  }
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    x2n[0] = 1u << 30;  // x^1
    for (int n = 1; n < 32; ++n) x2n[n] = mulmod(x2n[n - 1], x2n[n - 1]);
  }
};
const CrcTables &crc_tables() {
  static const CrcTables T;
  return T;
}
}  // namespace

// The same CRC by carry-less multiplication (PCLMULQDQ), four 16-byte lanes folded per step — the scheme of Intel's
// "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction" with the published constants for the
// reflected gzip polynomial (x^(4·128+32), x^(4·128−32), x^(128+32), x^(128−32), x^64, x^32 mod P; P and its Barrett
// quotient).  `c` is the running register (the complement of the CRC so far); n ≥ 64 and a multiple of 16.
// Slicing-by-8 does ≈ 2 GB/s a core, this ≈ 10: every decoded byte of a gzip member goes through it once.
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(uint32_t c, const uint8_t *p, size_t n) {
  alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
  alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
  alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
  alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
  __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
  x1 = _mm_loadu_si128((const __m128i *)(p + 0x00));
  x2 = _mm_loadu_si128((const __m128i *)(p + 0x10));
  x3 = _mm_loadu_si128((const __m128i *)(p + 0x20));
  x4 = _mm_loadu_si128((const __m128i *)(p + 0x30));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
  x0 = _mm_load_si128((const __m128i *)k1k2);
  p += 64;
  n -= 64;
  while (n >= 64) {  // four lanes, each folded across 64 bytes
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
    x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
    x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
    x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
    y5 = _mm_loadu_si128((const __m128i *)(p + 0x00));
    y6 = _mm_loadu_si128((const __m128i *)(p + 0x10));
    y7 = _mm_loadu_si128((const __m128i *)(p + 0x20));
    y8 = _mm_loadu_si128((const __m128i *)(p + 0x30));
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
    x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
    x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
    p += 64;
    n -= 64;
  }
  x0 = _mm_load_si128((const __m128i *)k3k4);  // the four lanes into one
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
  x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (n >= 16) {  // what is left, 16 bytes at a time
    x2 = _mm_loadu_si128((const __m128i *)p);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    p += 16;
    n -= 16;
  }
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);  // 128 → 64 bits
  x3 = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_srli_si128(x1, 8);
  x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_loadl_epi64((const __m128i *)k5k0);
  x2 = _mm_srli_si128(x1, 4);
  x1 = _mm_and_si128(x1, x3);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  x0 = _mm_load_si128((const __m128i *)poly);  // Barrett reduction, 64 → 32 bits
  x2 = _mm_and_si128(x1, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
  x2 = _mm_and_si128(x2, x3);
  x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}
static bool have_clmul() {
  static const bool on = getenv("SHK_NO_AVX2") == nullptr && __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
  return on;
}

uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
  const CrcTables &T = crc_tables();
  uint32_t c = ~crc;
  if (n >= 256 && have_clmul()) {
    const size_t m = n & ~(size_t)15;
    c = crc32_clmul(c, p, m);
    p += m;
    n -= m;
  }
  while (n && ((uintptr_t)p & 7)) {
    c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xFF];
    --n;
  }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = T.t[7][w & 0xFF] ^ T.t[6][(w >> 8) & 0xFF] ^ T.t[5][(w >> 16) & 0xFF] ^ T.t[4][(w >> 24) & 0xFF] ^
        T.t[3][(w >> 32) & 0xFF] ^ T.t[2][(w >> 40) & 0xFF] ^ T.t[1][(w >> 48) & 0xFF] ^ T.t[0][w >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xFF];
  return ~c;
}

uint32_t crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
  const CrcTables &T = crc_tables();
  // x^(8·len_b) mod P
  uint32_t p = 1u << 31;  // x^0
  uint64_t n = len_b;
  for (int k = 3; n; n >>= 1, k = (k + 1) & 31)
    if (n & 1) p = CrcTables::mulmod(T.x2n[k], p);
  return CrcTables::mulmod(p, crc_a) ^ crc_b;
}

// ---- gzip member ------------------------------------------------------------------------------------------------
// flate2 1.1.9, src/gz/mod.rs (GzHeaderParser::parse) and src/gz/bufread.rs (GzDecoder::{new, read}), restated:
//   header: 10 fixed bytes — ID1 ID2 must be 1f 8b, CM must be 8, reserved FLG bits must be 0, each else
//   io::Error(InvalidInput, "invalid gzip header"); then FEXTRA (2-byte length + that many bytes), FNAME and FCOMMENT
//   (to a NUL; more than 65535 bytes: InvalidInput "gzip header field too long"), FHCRC (2 bytes, the low half of the
//   header's CRC-32, else InvalidInput "corrupt gzip stream does not have a matching checksum"); the input ending
//   anywhere in it: io::ErrorKind::UnexpectedEof ("unexpected end of file").  GzDecoder::new keeps such an error
//   (GzState::Err) and the FIRST read returns it.
//   body: raw DEFLATE; a decoder error is zio::read's io::Error(InvalidInput, "corrupt deflate stream"); when the
//   input runs out inside the stream the decoder reports "no progress" and zio::read returns Ok(0), which GzDecoder
//   takes for the end of the body;
//   trailer (GzState::Crc): 8 bytes read with read_into — none left: UnexpectedEof; CRC-32 then ISIZE compared with
//   what was written: InvalidInput "corrupt gzip stream does not have a matching checksum"; then GzState::End —
//   Ok(0) for ever, whatever bytes follow (multi = false).
namespace {
const IoError ERR_EOF{IO_UNEXPECTED_EOF, "unexpected end of file"};
const IoError ERR_BAD_HEADER{IO_INVALID_INPUT, "invalid gzip header"};
const IoError ERR_FIELD_LONG{IO_INVALID_INPUT, "gzip header field too long"};
const IoError ERR_CHECKSUM{IO_INVALID_INPUT, "corrupt gzip stream does not have a matching checksum"};
const IoError ERR_DEFLATE{IO_INVALID_INPUT, "corrupt deflate stream"};
}  // namespace

void GzMember::open(const uint8_t *p, const uint8_t *e) {
  header_error = IoError();
  body_done = false;
  file_end = e;
  const uint8_t *const start = p;
  inf.reset(e, e);
  if (e - p < 10) {
    // (the identification bytes are only looked at once all ten are there)
    header_error = ERR_EOF;
    return;
  }
  if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) {
    header_error = ERR_BAD_HEADER;
    return;
  }
  const uint8_t flg = p[3];
  if (flg & 0xE0) {
    header_error = ERR_BAD_HEADER;
    return;
  }
  p += 10;
  if (flg & 4) {  // FEXTRA
    if (e - p < 2) {
      header_error = ERR_EOF;
      return;
    }
    const size_t xlen = (size_t)p[0] | ((size_t)p[1] << 8);
    p += 2;
    if ((size_t)(e - p) < xlen) {
      header_error = ERR_EOF;
      return;
    }
    p += xlen;
  }
  for (int field = 0; field < 2; ++field) {  // FNAME, FCOMMENT
    if (!(flg & (field == 0 ? 8 : 16))) continue;
    const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(e - p));
    const size_t n = z ? (size_t)(z - p) : (size_t)(e - p);
    if (n > 65535) {  // read_to_nul refuses the 65536th byte of a field that has not met its NUL
      header_error = ERR_FIELD_LONG;
      return;
    }
    if (!z) {
      header_error = ERR_EOF;
      return;
    }
    p = z + 1;
  }
  if (flg & 2) {  // FHCRC
    if (e - p < 2) {
      header_error = ERR_EOF;
      return;
    }
    const uint32_t want = (uint32_t)p[0] | ((uint32_t)p[1] << 8);
    if ((crc32_update(0, start, (size_t)(p - start)) & 0xFFFF) != want) {
      header_error = ERR_CHECKSUM;
      return;
    }
    p += 2;
  }
  inf.reset(p, e);
}

IoError GzMember::finish(InflateStatus last, uint32_t crc, uint64_t total_out) const {
  if (header_error.kind != IO_NONE) return header_error;
  if (last == INF_CORRUPT) return ERR_DEFLATE;
  const uint8_t *t = last == INF_STREAM_END ? inf.input_after_stream() : file_end;  // a short body: nothing is left to read
  if (file_end - t < 8) return ERR_EOF;  // (read_into: the trailer's bytes are not all there)
  const uint32_t want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
  const uint32_t want_len = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
  if (want_crc != crc) return ERR_CHECKSUM;
  if (want_len != (uint32_t)total_out) return ERR_CHECKSUM;
  return IoError();
}

}  // namespace shk
