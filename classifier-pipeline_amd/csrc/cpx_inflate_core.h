// cpx_inflate_core.h -- DEFLATE (RFC 1951) decoding of a gzip member as scalar code that compiles for the device
// (one wavefront per file executes it uniformly, csrc/cpx_inflate.hip) and for the host (the same code behind a
// plain-array I/O policy: tests/native/inflate_host.cpp checks it against zlib on CPU).
//
// What it replaces: the gzip layer of the Rust CPTV reader the reference calls (python-cptv 0.0.8 -> flate2;
// /root/reference/src/track/cliptrackextractor.py:108-129,160-162).  The stream format is RFC 1951 / 1952; the
// table construction follows the published canonical-Huffman scheme with a root table and variable-size
// second-level tables (the bounds LL_ENOUGH / D_ENOUGH are the known maxima for 288 / 32 symbols, 15-bit
// codes and these root sizes).
//
// I/O policy `IO` (all calls uniform across the wave on the device):
//   uint32_t bits()          the next >= 32 unread bits, LSB first (refills itself)
//   void     drop(int n)     consume n bits (n <= 32, and <= what bits() returned)
//   bool     overrun()       more bits were consumed than the input holds
//   void     align_byte()    drop the bits up to the next byte boundary
//   bool     literal(uint32_t byte)            false: output full
//   int      match(int len, int dist)          OK / ERR_DISTANCE (before the start of the output) / ERR_OUTPUT
//   int      stored(int len)                   copy len bytes input -> output (input is byte aligned): OK / ERR_*
//   uint16_t* ll_table() / d_table()           LL_ENOUGH / D_ENOUGH entries
//   uint8_t* lens()  [LENS_SCRATCH]   uint16_t* work() [WORK_SCRATCH]   uint16_t* small() [32]    scratch
//   int      decode_symbols(ll, dt)            the symbols of a block: decode_symbols_generic(*this, ll, dt) or an
//                                              equivalent
//   static uint32_t ld16(const uint16_t*)      table entry read
//   static int uni(int)                        identity (on the device: moves a value every lane holds into a scalar
//                                              register, so that what depends on it stays scalar)
#pragma once
#include <stdint.h>

#ifndef CPX_HD
#ifdef __HIPCC__
#define CPX_HD __host__ __device__
#else
#define CPX_HD
#endif
#endif

#ifdef __HIPCC__
#define CPX_INFL_INLINE __attribute__((always_inline)) inline
#define CPX_INFL_NOINLINE __attribute__((noinline))
#else
#define CPX_INFL_INLINE inline
#define CPX_INFL_NOINLINE
#endif

namespace cpx {
namespace infl {

constexpr int LL_ROOT = 10;
constexpr int D_ROOT = 8;
constexpr int PRE_ROOT = 7;
constexpr int LL_ENOUGH = 1334;  // 288 symbols, root 10, max length 15
constexpr int D_ENOUGH = 402;    // 32 symbols, root 8, max length 15
constexpr int PRE_ENOUGH = 128;  // 19 symbols, root 7, max length 7 (no second level)
constexpr int LENS_SCRATCH = 352; // 32 (code-length code) + 286 + 30 code lengths
constexpr int WORK_SCRATCH = 320;

enum Status {
  OK = 0,
  ERR_BLOCK_TYPE = 1,     // reserved block type 3
  ERR_STORED_LEN = 2,     // LEN != ~NLEN
  ERR_HEADER = 3,         // HLIT > 286 / HDIST > 30, bad code-length repeat
  ERR_TABLE = 4,          // over-subscribed or incomplete code lengths
  ERR_SYMBOL = 5,         // a bit pattern no code maps to / length symbol 286-287 / distance symbol 30-31
  ERR_DISTANCE = 6,       // match before the start of the output
  ERR_OUTPUT = 7,         // output capacity exhausted
  ERR_INPUT = 8,          // ran past the end of the input
  ERR_NO_EOB = 9,         // litlen code without end-of-block symbol
};

// table entry (uint16): literal leaf  [15:12] = 0, [11:4] the byte, [3:0] bits to drop (1..15): entry < 0x1000 is the
//                                     whole test of the decoder's fast path
//                       other leaf    bit 14 = 1, [8:4] symbol - 256 (end of block, lengths), [3:0] bits to drop;
//                                     INVALID (0x4000, zero bits to drop) marks bit patterns no code maps to
//                       pointer       bit 15 = 1, [14:11] index bits of the second-level table, [10:0] its offset
// (the distance and code-length codes have fewer than 256 symbols: their leaves are all of the first kind)
constexpr uint16_t INVALID = 0x4000;
CPX_HD inline uint16_t leaf(int nbits, int sym) {
  return sym < 256 ? (uint16_t)(nbits | (sym << 4)) : (uint16_t)(0x4000 | nbits | ((sym - 256) << 4));
}
CPX_HD inline uint16_t pointer(int sub_bits, int offset) { return (uint16_t)(0x8000 | (sub_bits << 11) | offset); }
CPX_HD inline int entry_symbol(uint32_t e) { return (e & 0x4000) ? 256 + (int)((e >> 4) & 31) : (int)((e >> 4) & 255); }

// Decode table for `n` symbols with code lengths lens[0..n) (0 = unused), root table of 2^root entries followed
// by the second-level tables.  `single_ok`: an incomplete code is accepted when it consists of exactly one
// 1-bit code (RFC 1951 3.2.7 for the distance code; zlib accepts the same for the litlen code).
// `work` is scratch for n uint16.  Returns OK or ERR_TABLE.
// `small` is scratch for 32 uint16 (the per-length counts and offsets: dynamically indexed, so not in registers).
CPX_HD CPX_INFL_NOINLINE inline int build_table(const uint8_t* lens, int n, int root, uint16_t* table, int enough, uint16_t* work,
                              uint16_t* small, bool single_ok = true) {
  uint16_t* const count = small;
  uint16_t* const offs = small + 16;
  for (int i = 0; i < 16; ++i) count[i] = 0;
  for (int s = 0; s < n; ++s) count[lens[s]] += 1;
  int max = 15;
  while (max >= 1 && count[max] == 0) --max;
  const int root_size = 1 << root;
  if (max == 0) {  // no codes at all: every pattern is invalid (a block of literals only has such a distance code)
    for (int i = 0; i < root_size; ++i) table[i] = INVALID;
    return OK;
  }
  int min = 1;
  while (min < max && count[min] == 0) ++min;
  int left = 1;
  for (int len = 1; len <= 15; ++len) {
    left <<= 1;
    left -= (int)count[len];
    if (left < 0) return ERR_TABLE;  // over-subscribed
  }
  if (left > 0 && (max != 1 || !single_ok)) return ERR_TABLE;  // incomplete (accepted only for a single 1-bit code)
  // symbols sorted by (length, value)
  offs[1] = 0;
  for (int len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
  for (int s = 0; s < n; ++s)
    if (lens[s] != 0) work[offs[lens[s]]++] = (uint16_t)s;
  for (int i = 0; i < root_size; ++i) table[i] = INVALID;
  unsigned huff = 0;  // the current code, bit-reversed
  int sym = 0, len = min, curr = root, drop = 0, used = root_size;
  int next = 0;       // offset of the table being filled
  unsigned low = ~0u;
  const unsigned mask = (unsigned)root_size - 1u;
  for (;;) {
    const uint16_t here = leaf(len - drop, work[sym]);
    const unsigned incr = 1u << (len - drop);
    unsigned fill = 1u << curr;
    const unsigned tmin = fill;
    do {
      fill -= incr;
      table[next + (huff >> drop) + fill] = here;
    } while (fill != 0);
    unsigned inc2 = 1u << (len - 1);
    while (huff & inc2) inc2 >>= 1;
    if (inc2 != 0) {
      huff &= inc2 - 1;
      huff += inc2;
    } else {
      huff = 0;
    }
    ++sym;
    if (--count[len] == 0) {
      if (len == max) break;
      len = lens[work[sym]];
    }
    if (len > root && (huff & mask) != low) {
      if (drop == 0) drop = root;
      next += (int)tmin;
      curr = len - drop;
      int l2 = 1 << curr;
      while (curr + drop < max) {
        l2 -= (int)count[curr + drop];
        if (l2 <= 0) break;
        ++curr;
        l2 <<= 1;
      }
      used += 1 << curr;
      if (used > enough) return ERR_TABLE;
      for (int i = 0; i < (1 << curr); ++i) table[next + i] = INVALID;
      low = huff & mask;
      table[low] = pointer(curr, next);
    }
  }
  return OK;
}

// one symbol of a (root, table) code from the low bits of `b`; *nbits = bits to drop, 0 when no code matches.
// IO::ld16 reads a table entry (on the device: into a scalar register, the value being the same in every lane)
template <class IO>
CPX_HD CPX_INFL_INLINE int decode_sym(const uint16_t* table, int root, uint32_t b, int* nbits) {
  uint32_t e = IO::ld16(table + (b & ((1u << root) - 1u)));
  int used = 0;
  if (e & 0x8000) {
    const int sb = (e >> 11) & 15;
    used = root;
    e = IO::ld16(table + (e & 0x7FF) + ((b >> root) & ((1u << sb) - 1u)));
  }
  const int n = e & 15;
  *nbits = n == 0 ? 0 : used + n;
  return entry_symbol(e);
}

CPX_HD inline int length_base(int s) {  // s = litlen symbol - 257, 0..28
  return s < 8 ? 3 + s : (s == 28 ? 258 : ((4 + (s & 3)) << ((s >> 2) - 1)) + 3);
}
CPX_HD inline int length_extra(int s) { return (s < 8 || s == 28) ? 0 : (s >> 2) - 1; }
CPX_HD inline int dist_base(int s) {  // s = distance symbol 0..29
  return s < 4 ? 1 + s : ((2 + (s & 1)) << ((s >> 1) - 1)) + 1;
}
CPX_HD inline int dist_extra(int s) { return s < 4 ? 0 : (s >> 1) - 1; }

// the fixed code of block type 1 (RFC 1951 3.2.6)
CPX_HD inline void fixed_lengths(uint8_t* lens) {
  int s = 0;
  for (; s < 144; ++s) lens[s] = 8;
  for (; s < 256; ++s) lens[s] = 9;
  for (; s < 280; ++s) lens[s] = 7;
  for (; s < 288; ++s) lens[s] = 8;
  for (; s < 320; ++s) lens[s] = 5;  // 288..319: the 32 distance symbols (30 and 31 never occur in valid data)
}

// One symbol that is not a root-table literal: `b` = the unread bits, `e` = the root entry they select.
// Returns OK (symbol consumed), 1000 (end of block) or an error.
template <class IO>
CPX_HD CPX_INFL_INLINE int slow_symbol(IO& io, const uint16_t* ll, const uint16_t* dt, uint32_t b, uint32_t e) {
  int used = 0;
  if (e & 0x8000) {
    const int sb = (e >> 11) & 15;
    used = LL_ROOT;
    e = IO::ld16(ll + (e & 0x7FF) + ((b >> LL_ROOT) & ((1u << sb) - 1u)));
  }
  if ((e & 15) == 0) return ERR_SYMBOL;
  int nb = used + (int)(e & 15);
  const int s = entry_symbol(e);
  if (s < 256) {  // a literal with a code longer than the root table's index
    io.drop(nb);
    return io.literal((uint32_t)s) ? OK : ERR_OUTPUT;
  }
  if (s == 256) {
    io.drop(nb);
    return 1000;
  }
  const int ls = s - 257;
  if (ls > 28) return ERR_SYMBOL;
  const int le = length_extra(ls);
  const int len = length_base(ls) + (int)((b >> nb) & ((1u << le) - 1u));
  io.drop(nb + le);
  b = io.bits();
  const int ds = decode_sym<IO>(dt, D_ROOT, b, &nb);
  if (nb == 0 || ds > 29) return ERR_SYMBOL;
  const int de = dist_extra(ds);
  const int dist = dist_base(ds) + (int)((b >> nb) & ((1u << de) - 1u));
  io.drop(nb + de);
  if (io.overrun()) return ERR_INPUT;
  return io.match(len, dist);
}

// The symbols of one block up to its end-of-block code (the portable form; the device has its own literal loop).
template <class IO>
CPX_HD CPX_INFL_INLINE int decode_symbols_generic(IO& io, const uint16_t* ll, const uint16_t* dt) {
  for (;;) {
    const uint32_t b = io.bits();
    const uint32_t e = IO::ld16(ll + (b & ((1u << LL_ROOT) - 1u)));
    if (e < 0x1000) {
      io.drop((int)(e & 15));
      if (!io.literal(e >> 4)) return ERR_OUTPUT;
      continue;
    }
    const int rc = slow_symbol(io, ll, dt, b, e);
    if (rc == 1000) return OK;
    if (rc != OK) return rc;
  }
}

// Inflate one DEFLATE stream (all blocks up to and including the final one).
template <class IO>
CPX_HD CPX_INFL_INLINE int inflate(IO& io) {
  uint16_t* const ll = io.ll_table();
  uint16_t* const dt = io.d_table();
  uint8_t* const lens = io.lens();
  uint16_t* const work = io.work();
  uint16_t* const small = io.small();
  for (;;) {
    uint32_t b = io.bits();
    const int final_block = b & 1;
    const int type = (b >> 1) & 3;
    io.drop(3);
    if (type == 3) return ERR_BLOCK_TYPE;
    if (type == 0) {
      io.align_byte();
      b = io.bits();
      const int len = b & 0xFFFF, nlen = (b >> 16) & 0xFFFF;
      io.drop(32);
      if (io.overrun()) return ERR_INPUT;
      if ((len ^ 0xFFFF) != nlen) return ERR_STORED_LEN;
      const int rc = io.stored(len);
      if (rc != OK) return rc;
    } else {
      if (type == 1) {
        fixed_lengths(lens);
        int rc = IO::uni(build_table(lens, 288, LL_ROOT, ll, LL_ENOUGH, work, small));
        if (rc != OK) return rc;
        rc = IO::uni(build_table(lens + 288, 32, D_ROOT, dt, D_ENOUGH, work, small));
        if (rc != OK) return rc;
      } else {
        const int hlit = (b >> 3 & 31) + 257, hdist = (b >> 8 & 31) + 1, hclen = (b >> 13 & 15) + 4;
        io.drop(14);
        if (hlit > 286 || hdist > 30) return ERR_HEADER;
        for (int i = 0; i < 19; ++i) lens[i] = 0;
        for (int i = 0; i < hclen; ++i) {
          // order of the code-length code lengths: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
          const int pos = i < 3 ? 16 + i : (i == 3 ? 0 : ((i & 1) ? 8 - ((i - 3) >> 1) : 8 + ((i - 4) >> 1)));
          lens[pos] = (uint8_t)(io.bits() & 7);
          io.drop(3);
        }
        // the code-length code shares the distance table's storage (built before the distance table)
        int rc = IO::uni(build_table(lens, 19, PRE_ROOT, dt, PRE_ENOUGH, work, small, false));
        if (rc != OK) return rc;
        uint16_t* const pre = dt;
        // the code lengths of both codes, run-length coded; they may not be stored over the pre-code's lens[0..19)
        // before it is built, hence the offset: lens[32 + k]
        uint8_t* const cl = lens + 32;
        const int total = hlit + hdist;
        int k = 0;
        while (k < total) {
          b = io.bits();
          int nb;
          const int s = decode_sym<IO>(pre, PRE_ROOT, b, &nb);
          if (nb == 0) return ERR_SYMBOL;
          b >>= nb;
          if (s < 16) {
            io.drop(nb);
            cl[k++] = (uint8_t)s;
          } else {
            int rep, val = 0;
            if (s == 16) {
              if (k == 0) return ERR_HEADER;
              val = IO::uni((int)cl[k - 1]);
              rep = 3 + (b & 3);
              io.drop(nb + 2);
            } else if (s == 17) {
              rep = 3 + (b & 7);
              io.drop(nb + 3);
            } else {
              rep = 11 + (b & 127);
              io.drop(nb + 7);
            }
            if (k + rep > total) return ERR_HEADER;
            while (rep-- > 0) cl[k++] = (uint8_t)val;
          }
          if (io.overrun()) return ERR_INPUT;
        }
        if (IO::uni((int)cl[256]) == 0) return ERR_NO_EOB;
        rc = IO::uni(build_table(cl, hlit, LL_ROOT, ll, LL_ENOUGH, work, small));
        if (rc != OK) return rc;
        rc = IO::uni(build_table(cl + hlit, hdist, D_ROOT, dt, D_ENOUGH, work, small));
        if (rc != OK) return rc;
      }
      // ---- the block's symbols ----
      const int rc = io.decode_symbols(ll, dt);
      if (rc != OK) return rc;
      if (io.overrun()) return ERR_INPUT;
    }
    if (final_block) return OK;
  }
}

// ---- gzip member framing (RFC 1952) over a byte buffer; host and device --------------------------------------
// Returns the offset of the DEFLATE data, or -1 for a malformed / truncated header.
CPX_HD inline long gzip_header_end(const uint8_t* p, long n) {
  if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return -1;
  const int flg = p[3];
  long pos = 10;
  if (flg & 4) {
    if (pos + 2 > n) return -1;
    pos += 2 + (long)(p[pos] | (p[pos + 1] << 8));
  }
  if (flg & 8) {
    while (pos < n && p[pos] != 0) ++pos;
    ++pos;
  }
  if (flg & 16) {
    while (pos < n && p[pos] != 0) ++pos;
    ++pos;
  }
  if (flg & 2) pos += 2;
  return pos + 8 <= n ? pos : -1;
}

}  // namespace infl
}  // namespace cpx
