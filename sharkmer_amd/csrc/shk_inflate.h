// shk_inflate.h — DEFLATE (RFC 1951) decoder + gzip member framing (RFC 1952) of the FASTQ front-end.
//
// Why hand-written: the reference opens a .gz with flate2::read::GzDecoder (src/io.rs:618-621;
// flate2 1.1.9 over miniz_oxide 0.8.9, Cargo.lock:569-576, 940-947 — neither vendored under /root/reference).
// That decoder (a) reads ONE gzip member and then reports end of stream whatever follows, (b) defers a header
// error to the first read, (c) verifies CRC32 and ISIZE of the trailer, (d) turns a stream that ends early into
// an io::Error the FASTQ loop reports through stream_io_error (src/io.rs:213-265).  zlib's gzread — what this
// file replaces — does none of that the same way (it concatenates members, passes non-gzip bytes through and
// treats a truncated stream as a clean end).  The oracle restates the same semantics separately over zlib's
// raw inflate (under oracle/), so the two share no decoding code.
//
// Plain C++17, no HIP: built into libshk.so and into the sanitizer builds of the host code.
#pragma once
#include <cstddef>
#include <cstdint>

namespace shk {

// io::ErrorKind of the error a flate2 reader would hand to BufRead::lines (what stream_io_error prints with {:?})
enum IoKind : int { IO_NONE = 0, IO_UNEXPECTED_EOF = 1, IO_INVALID_INPUT = 2, IO_INVALID_DATA = 3, IO_OTHER = 4 };
struct IoError {
  IoKind kind = IO_NONE;
  const char *text = "";  // Display of the io::Error
};
const char *io_kind_name(IoKind k);  // "UnexpectedEof", "InvalidInput", "InvalidData"

enum InflateStatus {
  INF_OUTPUT_FULL = 0,  // fewer than Inflater::OUT_SLACK bytes of room are left; call again with a fresh buffer
  INF_STREAM_END = 1,   // the final block's end-of-block symbol has been decoded
  INF_TRUNCATED = 2,    // the input ends inside the stream: everything decodable has been written
  INF_CORRUPT = 3       // not a valid DEFLATE stream from here on: everything before has been written
};

// Resumable over OUTPUT buffers, one contiguous INPUT (a mapped or slurped file).  History: the caller keeps the last
// 32 KiB of output (or all of it, if less) directly in front of out_pos when it switches buffers.
struct Inflater {
  static constexpr size_t OUT_SLACK = 320;  // a symbol is only decoded with this much room (258 + wide-copy overshoot)
  const uint8_t *in = nullptr, *in_end = nullptr;
  uint64_t bitbuf = 0;
  uint32_t bitcnt = 0;
  int state = 0;  // 0 block header, 1 stored, 2 Huffman, 3 done
  bool last_block = false;
  uint32_t stored_left = 0;
  uint64_t total_out = 0;
  // run() stops between two blocks once the stream position is at or behind stop_bit (bits from stop_origin; null: never)
  const uint8_t *stop_origin = nullptr;
  uint64_t stop_bit = ~0ull;
  bool stopped_between_blocks = false;
  bool hdr_plausible = true;  // the last dynamic header: ≤ 286 / 30 symbols and an end-of-block code (what real encoders write)
  uint32_t litlen[2048 + 1024];  // 11-bit primary + subtables
  uint32_t dist[256 + 512];      // 8-bit primary + subtables
  void reset(const uint8_t *p, const uint8_t *e);
  InflateStatus read_block_header();
  // The stream position in bits from `origin` (≤ in): what has been consumed so far.
  uint64_t bit_position(const uint8_t *origin) const { return (uint64_t)(in - origin) * 8 - bitcnt; }
  // Start (or go on) decoding at bit `bit` from `origin`, between blocks.
  void seek(const uint8_t *origin, const uint8_t *e, uint64_t bit);
  // SPECULATIVE decoding, for a stream entered at a block boundary somewhere in its middle: the 32 KiB in front of
  // the entry point are unknown, so the output is SYMBOLS — a byte value, or 256 + w for "byte w of that unknown
  // window" (w = 0: the oldest) — and a match copies symbols.  Decodes whole blocks into out[*out_pos, out_cap) and
  // stops in front of the first block that starts at or behind bit `stop_bit` (from `origin`): INF_OUTPUT_FULL then
  // (or when the room runs out: *out_pos says how far it got; the caller grows the buffer and calls again),
  // INF_STREAM_END behind the final block, INF_TRUNCATED / INF_CORRUPT as for run().  *block_start receives the bit
  // position of the block the decoder stands in front of (valid when it stopped between blocks).
  InflateStatus run_symbols(uint16_t *out, size_t *out_pos, size_t out_cap, const uint8_t *origin, uint64_t stop_bit, bool *between_blocks);
  // Decodes into out_base[out_pos, out_cap); bytes [out_pos - min(out_pos, 32768), out_pos) are the history.
  InflateStatus run(uint8_t *out_base, size_t *out_pos, size_t out_cap);
  // How many of the bytes in front of out_pos belong to THIS stream (the default: all of them).  A caller that decodes
  // one stream after another into the same buffer sets it before every run(): a match must not reach back into the
  // stream before (flate2's MultiGzDecoder starts every member with empty history: "corrupt deflate stream").
  uint64_t history_bytes = ~0ull;
  // After INF_STREAM_END: the first input byte behind the stream (whole unread bytes are given back).
  const uint8_t *input_after_stream() const { return in - (bitcnt >> 3); }
};

// CRC-32 (IEEE, reflected) — slicing-by-8, and the combination of two CRCs so that a buffer can be summed in pieces.
uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n);
uint32_t crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);

// One gzip member, the way flate2's GzDecoder (multi = false) reads it.
struct GzMember {
  Inflater inf;
  IoError header_error;  // GzState::Err: reported by the first read
  bool body_done = false;
  const uint8_t *file_end = nullptr;
  // Parses the header over [p, e).  Never fails here: a bad or short header is kept for the first read.
  void open(const uint8_t *p, const uint8_t *e);
  // What reading on after the last data byte reports: the trailer check (needs the CRC-32 and length of everything
  // written) after INF_STREAM_END, or the error a short or corrupt body turns into.  IO_NONE = clean end of stream.
  IoError finish(InflateStatus last, uint32_t crc, uint64_t total_out) const;
};

}  // namespace shk
