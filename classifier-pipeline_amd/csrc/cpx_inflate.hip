// cpx_inflate.hip -- gzip inflate + CPTV section index on the GPU: ONE WAVEFRONT PER FILE.
//
// A DEFLATE stream is a serial bit stream (every symbol's position depends on the lengths of all before it), so a
// file is one dependency chain; a batch of recordings is thousands of independent ones.  The wave executes the
// scalar decoder of cpx_inflate_core.h uniformly -- bit buffer, table look-ups and all control flow in scalar
// registers -- and uses its 64 lanes where the data is wide:
//   input    the lanes hold the next 256 bytes of the file as one dword each (one coalesced load per 256 B, the
//            following chunk requested while this one is consumed); the bit buffer takes dword k by v_readlane
//   literals are gathered into one register (v_writelane, lane = position in the run) and stored as ONE coalesced
//            byte store per run, before a match or when 64 are pending
//   matches  lane j copies byte j of the match (source index folded by the distance for overlapping copies)
//   tables   root + second-level Huffman tables of the block, 16-bit entries, in the wave's 4.4 KB of LDS
// The output itself is the LZ77 window (matches read the bytes this wave stored earlier through L1 / L2).
// After the last block the same wave checks the gzip trailer and walks the CPTV sections of what it wrote
// (cpx_cptv_index_core.h).  Throughput comes from the number of files in flight (up to 32 per CU, 8,192 on the chip),
// not from a file's own speed: a lone file inflates slower here than on one CPU core.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cpx_cptv_index_core.h"
#include "cpx_inflate_core.h"
#include "cpx_kernels.h"

namespace cpx {

namespace {

constexpr int WAVES_PER_WG = 1;
constexpr int ERR_GZIP_HEADER = 10;
constexpr int ERR_GZIP_TRAILER = 11;
constexpr int ERR_GZIP_CRC = 12;

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ long rfl64(long v) {
  const unsigned lo = (unsigned)rfl((int)(v & 0xFFFFFFFFl)), hi = (unsigned)rfl((int)((unsigned long)v >> 32));
  return (long)(((unsigned long)hi << 32) | lo);
}

struct WaveLds {
  uint32_t crc_tab[256];   // CRC-32 byte table (built by the wave after the last block)
  uint16_t ll[infl::LL_ENOUGH];
  uint16_t dt[infl::D_ENOUGH];
  uint16_t work[infl::WORK_SCRATCH];
  uint16_t small[32];
  uint8_t lens[infl::LENS_SCRATCH];
};

struct WaveIO {
  // ---- input: the file as dwords (the buffer is readable past the file's end; bits past it read as zero) ----
  const uint32_t* words;  // dword-aligned start of the file
  long n_words;           // dwords holding file bytes
  long file_bits;         // bits of the file
  long wbase;             // index (in the file) of dword 0 of `cur`; the next dword the bit buffer takes is wbase + widx
  int cur;                // per lane: dword `lane` of the current 64-dword chunk
  int widx;               // next dword of `cur`, 64 when the chunk is used up
  uint64_t buf;
  int cnt;
  int lane;
  // ---- output ----
  uint8_t* out;
  long cap;
  long lim;               // = cap, or 0 once the input is exhausted
  long n_flushed;         // bytes stored
  int lit, nlit;          // pending literals: lane k of `lit` holds literal k of the run as byte << 4 (a table entry)
  WaveLds* lds;
  const uint8_t* file_bytes;
  // lane-indexed constant tables (RFC 1951 3.2.5): lane s of len_tab = base | extra bits << 16 of length symbol
  // 257 + s, of dist_tab the same for distance symbol s -- one v_readlane instead of a branchy computation
  int len_tab, dist_tab;

  __device__ __forceinline__ long pos_bits() const { return (wbase + widx) * 32 - cnt; }  // position of the next bit
  __device__ __forceinline__ int load_chunk(long chunk) const {
    const long w = chunk * 64 + lane;
    return w < n_words ? (int)words[w] : 0;
  }
  __device__ __forceinline__ void seek_byte(long byte) {
    const long w = byte >> 2;
    wbase = w & ~63l;
    cur = load_chunk(wbase >> 6);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    widx = (int)(w & 63);
    buf = 0;
    cnt = 0;
    const int skip = (int)(byte & 3) * 8;
    if (skip) {
      (void)bits();
      buf >>= skip;
      cnt -= skip;
    }
  }
  // the next 64 dwords of the file (when the current ones are used up).  A stream that runs past its input decodes
  // the zero padding -- possibly as literals, for as long as the output has room: checked here, once per 256 bytes
  // of input (the literal path itself only tests the output limit)
  __device__ __forceinline__ void next_chunk() {
    // (loaded when it is needed: a register with a load in flight that lives across the decoder's control flow makes
    // the compiler wait for ALL outstanding memory operations -- the acknowledgements of recent stores included --
    // at every copy of it; one exposed load per 256 bytes of input is cheaper)
    wbase += 64;
    if (wbase * 32 > file_bits) lim = 0;
    cur = load_chunk(wbase >> 6);
    // waited for HERE (vmcnt(0), expcnt / lgkmcnt untouched): the compiler then knows the register holds its value and
    // does not put a wait -- for every outstanding store's acknowledgement -- in front of each later copy of it
    __builtin_amdgcn_s_waitcnt(0x0F70);
    widx = 0;
  }
  __device__ __forceinline__ uint32_t next_word() {
    if (widx == 64) next_chunk();
    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane(cur, rfl(widx));
    widx += 1;
    return v;
  }
  __device__ __forceinline__ uint32_t bits() {
    if (cnt < 32) {
      buf |= (uint64_t)next_word() << cnt;
      cnt += 32;
    }
    return (uint32_t)buf;
  }
  __device__ __forceinline__ void drop(int n) {  // n <= 32 bits of what the preceding bits() returned
    buf >>= n;
    cnt -= n;
  }
  __device__ __forceinline__ bool overrun() const { return pos_bits() > file_bits; }
  __device__ __forceinline__ void align_byte() {
    const int n = (int)((8 - (pos_bits() & 7)) & 7);
    if (n) {
      if (cnt < n) (void)bits();
      drop(n);
    }
  }
  __device__ __forceinline__ long n_out() const { return n_flushed + nlit; }
  // pending literals -> memory; false when they do not fit
  __device__ __forceinline__ bool flush() {
    if (nlit > 0) {
      if (n_flushed + nlit > lim) return false;
      if (lane < nlit) out[n_flushed + lane] = (uint8_t)(lit >> 4);  // `lit` holds table entries: byte = entry >> 4
      n_flushed += nlit;
      nlit = 0;
    }
    return true;
  }
  __device__ __forceinline__ bool literal(uint32_t b) {
    // lane nlit of `lit` = b (the lane select goes through M0: one other scalar operand per vector instruction)
    asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(lit) : "s"(rfl((int)(b << 4))), "s"(rfl(nlit)) : "m0");
    nlit += 1;
    if (nlit == 64) return flush();
    return true;
  }
  __device__ __forceinline__ int match(int len, int dist) {
    const long at = n_flushed + nlit;  // where the match goes
    if (dist > at) return infl::ERR_DISTANCE;
    if (at + len > lim) return infl::ERR_OUTPUT;
    uint8_t* const dst = out + at;
    const uint8_t* const src = dst - dist;
    // stores of the hand-written loop may still be in flight: complete before a load that may read their bytes
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (dist >= len + nlit && len <= 64) {
      // the common shape: its source lies before the pending literals, so the load is issued FIRST and the literal
      // store rides in its shadow; the wait before the match's own store then counts past that younger store
      // (vmcnt is in order) instead of waiting for its acknowledgement
      int v = 0;
      if (lane < len) v = src[lane];
      if (nlit > 0) {
        if (lane < nlit) out[n_flushed + lane] = (uint8_t)(lit >> 4);
        n_flushed += nlit;
        nlit = 0;
      }
      if (lane < len) dst[lane] = (uint8_t)v;
      n_flushed += len;
      return infl::OK;
    }
    if (!flush()) return infl::ERR_OUTPUT;
    if (dist >= len) {
      for (int k = 0; k < len; k += 64) {
        const int j = k + lane;
        if (j < len) dst[j] = src[j];
      }
    } else {
      // overlapping copy: byte j repeats byte j mod dist of the `dist` bytes before the match
      for (int k = 0; k < len; k += 64) {
        const int j = k + lane;
        if (j < len) dst[j] = src[j % dist];
      }
    }
    n_flushed += len;
    return infl::OK;
  }
  __device__ __forceinline__ int stored(int len) {
    if (!flush()) return infl::ERR_OUTPUT;
    const long byte = pos_bits() >> 3;
    if (byte + len > (file_bits >> 3)) return infl::ERR_INPUT;
    if (n_flushed + len > lim) return infl::ERR_OUTPUT;
    for (int k = 0; k < len; k += 64) {
      const int j = k + lane;
      if (j < len) out[n_flushed + j] = file_bytes[byte + j];
    }
    n_flushed += len;
    seek_byte(byte + len);
    return infl::OK;
  }
  __device__ __forceinline__ uint16_t* ll_table() { return lds->ll; }
  __device__ __forceinline__ uint16_t* d_table() { return lds->dt; }
  __device__ __forceinline__ uint8_t* lens() { return lds->lens; }
  __device__ __forceinline__ uint16_t* work() { return lds->work; }
  __device__ __forceinline__ uint16_t* small() { return lds->small; }
  static __device__ __forceinline__ uint32_t ld16(const uint16_t* p) { return (uint32_t)rfl((int)*p); }
  static __device__ __forceinline__ int uni(int v) { return rfl(v); }

  // The symbols of a block.  Root-table literals -- four of five symbols of a recording -- run in a hand-written
  // loop of 20 scalar instructions (the compiler's form of the same loop is about sixty, and a lone wave issues one
  // instruction per four to five cycles): look the low ten bits up, leave when the entry is not a literal, when
  // fewer than 32 bits are buffered or when 64 literals are pending; otherwise shift the buffer and put the byte into
  // lane `nlit` of the literal register.  (The bit buffer is pinned to s[90:91] for the block: the loop reads its low
  // half by name.)
  __device__ __forceinline__ int decode_symbols(const uint16_t* ll, const uint16_t* dt) {
    // The root table of the literal / length code lives in REGISTERS for the block: 1024 entries = 16 vector
    // registers x 64 lanes (entry i in lane i & 63 of register i >> 6).  A look-up is then a register-indexed move
    // (s_set_gpr_idx_on) + v_readlane into a scalar register: no LDS round trip (about 150 cycles of the 260 a
    // literal took with the table in LDS) on the decoder's dependency chain.  Second-level tables stay in LDS.
    typedef int v16i __attribute__((ext_vector_type(16)));
    v16i tab;
#pragma unroll
    for (int k = 0; k < 16; ++k) tab[k] = (int)ll[k * 64 + lane];
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i dtab;
#pragma unroll
    for (int k = 0; k < 4; ++k) dtab[k] = (int)dt[k * 64 + lane];
    const int dtab0 = dtab[0], dtab1 = dtab[1], dtab2 = dtab[2], dtab3 = dtab[3];
    for (;;) {
      uint32_t e = 0xFFFF;
      int t0, t1, vt;
      // (values the compiler may hold in vector registers although every lane agrees: moved to scalar ones here)
      uint64_t sbuf = (uint64_t)rfl64((long)buf);
      int scnt = rfl(cnt), snlit = rfl(nlit);
      // One literal: entry = tab[low ten bits]; leave when it is no literal, else shift the buffer and put the ENTRY
      // into lane nlit of `lit` (the byte is entry >> 4: taken when the run is stored).  Three literals per round:
      // 32 buffered bits cover three root-table codes.  The bit buffer is pinned to s[90:91] and the table to
      // v[40:55]: the loop names their parts.
#define CPX_INFL_LITERAL                                  \
  "s_and_b32 %[t0], s90, 0x3ff\n\t"                       \
  "s_lshr_b32 %[t1], %[t0], 6\n\t"                        \
  "s_set_gpr_idx_on %[t1], 0x1\n\t"                       \
  "v_mov_b32 %[vt], v40\n\t"                              \
  "s_set_gpr_idx_off\n\t"                                 \
  "v_readlane_b32 %[e], %[vt], %[t0]\n\t"                 \
  "s_cmp_ge_u32 %[e], 0x1000\n\t"                         \
  "s_cbranch_scc1 2f\n\t"                                 \
  "s_and_b32 %[t0], %[e], 15\n\t"                         \
  "s_mov_b32 m0, %[nlit]\n\t"                             \
  "s_lshr_b64 s[90:91], s[90:91], %[t0]\n\t"              \
  "s_sub_u32 %[cnt], %[cnt], %[t0]\n\t"                   \
  "v_writelane_b32 %[lit], %[e], m0\n\t"                  \
  "s_add_u32 %[nlit], %[nlit], 1\n\t"
      int swidx = rfl(widx);
      // the output position / limit as 32-bit scalars for the loop (a file whose output could pass 2 GiB gets lim32 = 0:
      // every store and match of the loop then declines and the C++ path below does them)
      int snf = rfl((int)(uint32_t)n_flushed);
      const int lim32 = rfl(lim < 0x7FFFFFFFl ? (int)lim : 0);
      const int out_lo = rfl((int)(uint32_t)(uintptr_t)out), out_hi = rfl((int)(uint32_t)((uintptr_t)out >> 32));
      int t2, t3, vb;
      // A match whose length AND distance codes are root-table leaves, of at most 64 bytes, with its source before the
      // pending literals (dist >= len + nlit) stays in the loop as well -- about 60 scalar instructions against some 180
      // through the compiler's form of match() and the loop's exit / re-entry: the length code is decoded on a COPY of
      // the bit state (saved in s[96:97], s98, s99) so that any other shape restores it and leaves with the entry, as
      // before.  The copy: source load first, the pending literals' store in its shadow, then the match's store.
      // A run of 62+ pending literals is stored here too.
      asm volatile(
          "1:\n\t"
          "s_cmp_gt_i32 %[cnt], 31\n\t"
          "s_cbranch_scc1 4f\n\t"
          // refill: the next dword of the chunk (lane widx of `cur`) goes above the cnt buffered bits
          "s_cmp_eq_u32 %[widx], 64\n\t"
          "s_cbranch_scc1 3f\n\t"
          "v_readlane_b32 s92, %[cur], %[widx]\n\t"
          "s_mov_b32 s93, 0\n\t"
          "s_add_u32 %[widx], %[widx], 1\n\t"
          "s_lshl_b64 s[92:93], s[92:93], %[cnt]\n\t"
          "s_add_u32 %[cnt], %[cnt], 32\n\t"
          "s_or_b64 s[90:91], s[90:91], s[92:93]\n\t"
          "4:\n\t"
          "s_cmp_gt_u32 %[nlit], 61\n\t"
          "s_cbranch_scc1 5f\n\t"
          "6:\n\t"
          CPX_INFL_LITERAL CPX_INFL_LITERAL CPX_INFL_LITERAL
          "s_branch 1b\n\t"
          // ---- 62+ literals pending: store the run ----
          "5:\n\t"
          "s_add_u32 %[t0], %[nf], %[nlit]\n\t"
          "s_cmp_gt_u32 %[t0], %[lim]\n\t"
          "s_cbranch_scc1 3f\n\t"
          "s_add_u32 s92, %[outlo], %[nf]\n\t"
          "s_addc_u32 s93, %[outhi], 0\n\t"
          "v_cmp_gt_u32 vcc, %[nlit], %[vlane]\n\t"
          "s_and_saveexec_b64 s[94:95], vcc\n\t"
          "v_lshrrev_b32 %[vt], 4, %[lit]\n\t"
          "global_store_byte %[vlane], %[vt], s[92:93]\n\t"
          "s_mov_b64 exec, s[94:95]\n\t"
          "s_mov_b32 %[nf], %[t0]\n\t"
          "s_mov_b32 %[nlit], 0\n\t"
          "s_branch 6b\n\t"
          // ---- e = a root entry that is no literal ----
          "2:\n\t"
          "s_cmp_gt_i32 %[cnt], 31\n\t"
          "s_cbranch_scc1 7f\n\t"
          "s_cmp_eq_u32 %[widx], 64\n\t"
          "s_cbranch_scc1 9f\n\t"
          "v_readlane_b32 s92, %[cur], %[widx]\n\t"
          "s_mov_b32 s93, 0\n\t"
          "s_add_u32 %[widx], %[widx], 1\n\t"
          "s_lshl_b64 s[92:93], s[92:93], %[cnt]\n\t"
          "s_add_u32 %[cnt], %[cnt], 32\n\t"
          "s_or_b64 s[90:91], s[90:91], s[92:93]\n\t"
          "7:\n\t"
          "s_and_b32 %[t0], %[e], 0xc000\n\t"
          "s_cmp_lg_u32 %[t0], 0x4000\n\t"
          "s_cbranch_scc1 9f\n\t"                       // a pointer to a second-level table
          "s_bfe_u32 %[t1], %[e], 0x50004\n\t"          // symbol - 256
          "s_sub_u32 %[t1], %[t1], 1\n\t"               // length symbol - 257 (end of block, invalid: wraps)
          "s_cmp_gt_u32 %[t1], 28\n\t"
          "s_cbranch_scc1 9f\n\t"
          "s_cmp_eq_u32 %[widx], 64\n\t"                // the distance may need one more dword
          "s_cbranch_scc1 9f\n\t"
          "s_mov_b64 s[96:97], s[90:91]\n\t"
          "s_mov_b32 s98, %[cnt]\n\t"
          "s_mov_b32 s99, %[widx]\n\t"
          "s_and_b32 %[t0], %[e], 15\n\t"
          "v_readlane_b32 %[t2], %[lentab], %[t1]\n\t"  // base | extra bits << 16
          "s_lshr_b64 s[90:91], s[90:91], %[t0]\n\t"
          "s_sub_u32 %[cnt], %[cnt], %[t0]\n\t"
          "s_lshr_b32 %[t0], %[t2], 16\n\t"
          "s_and_b32 %[t2], %[t2], 0xffff\n\t"
          "s_bfm_b32 %[t3], %[t0], 0\n\t"
          "s_and_b32 %[t3], %[t3], s90\n\t"
          "s_add_u32 %[t2], %[t2], %[t3]\n\t"           // t2 = length
          "s_lshr_b64 s[90:91], s[90:91], %[t0]\n\t"
          "s_sub_u32 %[cnt], %[cnt], %[t0]\n\t"
          "s_cmp_gt_i32 %[cnt], 31\n\t"
          "s_cbranch_scc1 10f\n\t"
          "v_readlane_b32 s92, %[cur], %[widx]\n\t"
          "s_mov_b32 s93, 0\n\t"
          "s_add_u32 %[widx], %[widx], 1\n\t"
          "s_lshl_b64 s[92:93], s[92:93], %[cnt]\n\t"
          "s_add_u32 %[cnt], %[cnt], 32\n\t"
          "s_or_b64 s[90:91], s[90:91], s[92:93]\n\t"
          "10:\n\t"
          "s_and_b32 %[t0], s90, 0xff\n\t"              // distance: root table (256 entries) in v[56:59]
          "s_lshr_b32 %[t1], %[t0], 6\n\t"
          "s_set_gpr_idx_on %[t1], 0x1\n\t"
          "v_mov_b32 %[vt], v56\n\t"
          "s_set_gpr_idx_off\n\t"
          "v_readlane_b32 %[t3], %[vt], %[t0]\n\t"
          "s_bitcmp1_b32 %[t3], 15\n\t"
          "s_cbranch_scc1 8f\n\t"                       // second-level distance code
          "s_and_b32 %[t0], %[t3], 15\n\t"
          "s_cmp_eq_u32 %[t0], 0\n\t"
          "s_cbranch_scc1 8f\n\t"                       // invalid code
          "s_bfe_u32 %[t1], %[t3], 0x80004\n\t"         // distance symbol
          "s_cmp_gt_u32 %[t1], 29\n\t"
          "s_cbranch_scc1 8f\n\t"
          "v_readlane_b32 %[t3], %[disttab], %[t1]\n\t"
          "s_lshr_b64 s[90:91], s[90:91], %[t0]\n\t"
          "s_sub_u32 %[cnt], %[cnt], %[t0]\n\t"
          "s_lshr_b32 %[t0], %[t3], 16\n\t"
          "s_and_b32 %[t3], %[t3], 0xffff\n\t"
          "s_bfm_b32 %[t1], %[t0], 0\n\t"
          "s_and_b32 %[t1], %[t1], s90\n\t"
          "s_add_u32 %[t3], %[t3], %[t1]\n\t"           // t3 = distance
          "s_lshr_b64 s[90:91], s[90:91], %[t0]\n\t"
          "s_sub_u32 %[cnt], %[cnt], %[t0]\n\t"
          // the copy's shape: len <= 64, dist <= at, dist >= len + nlit, at + len <= limit
          "s_cmp_gt_u32 %[t2], 64\n\t"
          "s_cbranch_scc1 8f\n\t"
          "s_add_u32 %[t0], %[nf], %[nlit]\n\t"         // t0 = at
          "s_cmp_gt_u32 %[t3], %[t0]\n\t"
          "s_cbranch_scc1 8f\n\t"
          "s_add_u32 %[t1], %[t2], %[nlit]\n\t"
          "s_cmp_lt_u32 %[t3], %[t1]\n\t"
          "s_cbranch_scc1 8f\n\t"
          "s_add_u32 %[t1], %[t0], %[t2]\n\t"           // t1 = at + len
          "s_cmp_gt_u32 %[t1], %[lim]\n\t"
          "s_cbranch_scc1 8f\n\t"
          // the previous match's bytes have arrived by now, or are waited for here: stored before this match's load
          // (which may read them) is issued
          // every earlier store of this wave complete before the source load: a byte store and a load of the same
          // bytes a few instructions later are not ordered by the memory pipeline under load (measured: with the
          // match's store deferred past the next literals, a few recordings in 10^5 decoded wrongly while other
          // kernels shared the chip; none in 1.5 x 10^5 with the store in place and this wait)
          "s_waitcnt vmcnt(0)\n\t"
          "s_sub_u32 %[t3], %[t0], %[t3]\n\t"
          "s_add_u32 s92, %[outlo], %[t3]\n\t"
          "s_addc_u32 s93, %[outhi], 0\n\t"
          "v_cmp_gt_u32 vcc, %[t2], %[vlane]\n\t"
          "s_and_saveexec_b64 s[94:95], vcc\n\t"
          "global_load_ubyte %[vb], %[vlane], s[92:93]\n\t"
          "s_mov_b64 exec, s[94:95]\n\t"
          "s_add_u32 s96, %[outlo], %[t0]\n\t"
          "s_addc_u32 s97, %[outhi], 0\n\t"
          "s_cmp_eq_u32 %[nlit], 0\n\t"
          "s_cbranch_scc1 11f\n\t"
          "s_add_u32 s92, %[outlo], %[nf]\n\t"
          "s_addc_u32 s93, %[outhi], 0\n\t"
          "v_cmp_gt_u32 vcc, %[nlit], %[vlane]\n\t"
          "s_and_saveexec_b64 s[94:95], vcc\n\t"
          "v_lshrrev_b32 %[vt], 4, %[lit]\n\t"
          "global_store_byte %[vlane], %[vt], s[92:93]\n\t"
          "s_mov_b64 exec, s[94:95]\n\t"
          "11:\n\t"
          "v_cmp_gt_u32 vcc, %[t2], %[vlane]\n\t"
          "s_and_saveexec_b64 s[94:95], vcc\n\t"
          "s_waitcnt vmcnt(0)\n\t"
          "global_store_byte %[vlane], %[vb], s[96:97]\n\t"
          "s_mov_b64 exec, s[94:95]\n\t"
          "s_mov_b32 %[nf], %[t1]\n\t"
          "s_mov_b32 %[nlit], 0\n\t"
          "s_branch 1b\n\t"
          // any other shape: the bit state as it was before the length code, the entry to the C++ path
          "8:\n\t"
          "s_mov_b64 s[90:91], s[96:97]\n\t"
          "s_mov_b32 %[cnt], s98\n\t"
          "s_mov_b32 %[widx], s99\n\t"
          "s_branch 9f\n\t"
          "3:\n\t"
          "s_mov_b32 %[e], 0xffff\n\t"
          "9:\n\t"
          : "+{s[90:91]}"(sbuf), [cnt] "+s"(scnt), [nlit] "+s"(snlit), [lit] "+v"(lit), [e] "+s"(e), [t0] "=&s"(t0),
            [t1] "=&s"(t1), [vt] "=&v"(vt), [widx] "+s"(swidx), [nf] "+s"(snf), [t2] "=&s"(t2), [t3] "=&s"(t3),
            [vb] "=&v"(vb)
          : "{v[40:55]}"(tab), [cur] "v"(cur), "{v[56:59]}"(dtab), [lentab] "v"(len_tab), [disttab] "v"(dist_tab),
            [vlane] "v"(lane), [outlo] "s"(out_lo), [outhi] "s"(out_hi), [lim] "s"(lim32)
          : "m0", "scc", "vcc", "memory", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
      n_flushed = (long)(((unsigned long)n_flushed & ~0xFFFFFFFFul) | (uint32_t)snf);
#undef CPX_INFL_LITERAL
      buf = sbuf;
      cnt = scnt;
      nlit = snlit;
      widx = swidx;
      if (e == 0xFFFF) {  // left for the next chunk of input or a store of the pending literals, not for a symbol
        if (cnt < 32) (void)bits();
        if (nlit > 61 && !flush()) return infl::ERR_OUTPUT;
        continue;
      }
      // e: the root entry of a symbol that is not a root-table literal (selected by the ten low bits, which were
      // valid; the second or third literal of a round may have left fewer than the 32 bits the slow path reads)
      if (cnt < 32) (void)bits();
      if ((e & 0xC000) == 0x4000 && (e & 15) != 0) {
        // ---- the common non-literal: a root-table length code (or the end of the block) ----
        const int nb = (int)(e & 15);
        const int ls = (int)((e >> 4) & 31) - 1;   // symbol - 257
        if (ls < 0) {                              // 256: end of block
          drop(nb);
          return infl::OK;
        }
        if (ls <= 28) {
          uint32_t b = (uint32_t)buf;
          const uint32_t lt = (uint32_t)__builtin_amdgcn_readlane(len_tab, ls);
          const int le = (int)(lt >> 16);
          const int len = (int)(lt & 0xFFFFu) + (int)((b >> nb) & ((1u << le) - 1u));
          drop(nb + le);
          b = bits();
          // distance: root table (256 entries) in four registers, entry i in lane i & 63 of register i >> 6
          const int di = (int)(b & 255u);
          const int dk = di >> 6;
          const int dsel = dk == 0 ? dtab0 : (dk == 1 ? dtab1 : (dk == 2 ? dtab2 : dtab3));
          const uint32_t de_ = (uint32_t)__builtin_amdgcn_readlane(dsel, di & 63);
          if (!(de_ & 0x8000u) && (de_ & 15u) != 0) {
            const int nbd = (int)(de_ & 15u);
            const int ds = (int)((de_ >> 4) & 255u);
            if (ds > 29) return infl::ERR_SYMBOL;
            const uint32_t dtv = (uint32_t)__builtin_amdgcn_readlane(dist_tab, ds);
            const int de = (int)(dtv >> 16);
            const int dist = (int)(dtv & 0xFFFFu) + (int)((b >> nbd) & ((1u << de) - 1u));
            drop(nbd + de);
            const int rcm = match(len, dist);
            if (rcm != infl::OK) return rcm;
            continue;
          }
          // a distance code longer than the root index (or an invalid one): the generic decoder finishes the symbol
          int nbd;
          const int ds = infl::decode_sym<WaveIO>(dt, infl::D_ROOT, b, &nbd);
          if (nbd == 0 || ds > 29) return infl::ERR_SYMBOL;
          const int de = infl::dist_extra(ds);
          const int dist = infl::dist_base(ds) + (int)((b >> nbd) & ((1u << de) - 1u));
          drop(nbd + de);
          if (overrun()) return infl::ERR_INPUT;
          const int rcm = match(len, dist);
          if (rcm != infl::OK) return rcm;
          continue;
        }
      }
      const int rc = infl::slow_symbol(*this, ll, dt, (uint32_t)buf, e);
      if (rc == 1000) return infl::OK;
      if (rc != infl::OK) return rc;
    }
  }
};

// ---- CRC-32 of the inflated bytes (RFC 1952: IEEE polynomial, reflected) by the file's own wave ----
// Lane k runs the byte-table recurrence over chunk k of the output (64 chunks of L bytes, L a multiple of 4); the 64
// partial states are folded in order with the operator "advance the state through L zero bytes" -- a 32 x 32 matrix over
// GF(2) kept one column per lane (lanes 32-63 mirror 0-31), built by squaring like zlib's crc32_combine.
__device__ __forceinline__ uint32_t xor_reduce32(uint32_t v) {   // over each half's 32 lanes -> every lane
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v ^= (uint32_t)__shfl_xor((int)v, o, 64);
  return v;
}
__device__ __forceinline__ uint32_t gf2_times(uint32_t col, uint32_t vec, int lane) {   // matrix (columns on lanes) x vec
  return xor_reduce32(((vec >> (lane & 31)) & 1u) ? col : 0u);
}
__device__ __forceinline__ uint32_t gf2_mul(uint32_t a, uint32_t b, int lane) {          // a x b: column i = a x (column i of b)
  uint32_t out = 0;
  for (int i = 0; i < 32; ++i) {
    const uint32_t r = gf2_times(a, (uint32_t)__builtin_amdgcn_readlane((int)b, i), lane);
    if ((lane & 31) == i) out = r;
  }
  return out;
}
__device__ __forceinline__ uint32_t gf2_zero_bytes(long n, int lane) {   // the operator for n zero bytes
  const int c = lane & 31;
  uint32_t base = c == 0 ? 0xEDB88320u : (1u << (c - 1));   // one zero BIT
  base = gf2_mul(base, base, lane);                         // 2 bits
  base = gf2_mul(base, base, lane);                         // 4
  base = gf2_mul(base, base, lane);                         // 8 = one byte
  uint32_t res = 1u << c;                                   // identity
  while (n > 0) {                                           // (uniform)
    if (n & 1) res = gf2_mul(base, res, lane);
    n >>= 1;
    if (n > 0) base = gf2_mul(base, base, lane);
  }
  return res;
}
__device__ uint32_t wave_crc32(const uint8_t* data, long n, uint32_t* tab, int lane) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint32_t c = (uint32_t)(lane * 4 + k);
#pragma unroll
    for (int b = 0; b < 8; ++b) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
    tab[lane * 4 + k] = c;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the table is this wave's own
  const long L = (((n + 63) >> 6) + 3) & ~3l;               // chunk length (the same for every lane but the last used)
  const long lo = (long)lane * L;
  const long len = lo >= n ? 0 : (n - lo < L ? n - lo : L);
  uint32_t c = lane == 0 ? 0xFFFFFFFFu : 0u;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(data + lo);
  const long nw = len >> 2;
  for (long j = 0; j < nw; ++j) {
    const uint32_t v = w[j];
    c = tab[(c ^ v) & 0xFFu] ^ (c >> 8);
    c = tab[(c ^ (v >> 8)) & 0xFFu] ^ (c >> 8);
    c = tab[(c ^ (v >> 16)) & 0xFFu] ^ (c >> 8);
    c = tab[(c ^ (v >> 24)) & 0xFFu] ^ (c >> 8);
  }
  for (long j = nw << 2; j < len; ++j) c = tab[(c ^ data[lo + j]) & 0xFFu] ^ (c >> 8);
  // fold: state = U_len(k)(state) ^ partial(k), k = 1 .. last used chunk
  const int used = (int)rfl((int)((n + L - 1) / (L > 0 ? L : 1)));       // chunks that hold bytes (n > 0 here)
  const long last_len = n - (long)(used - 1) * L;
  const uint32_t op_full = gf2_zero_bytes(L, lane);
  const uint32_t op_last = last_len == L ? op_full : gf2_zero_bytes(last_len, lane);
  uint32_t acc = (uint32_t)__builtin_amdgcn_readlane((int)c, 0);
  for (int k = 1; k < used; ++k)
    acc = gf2_times(k == used - 1 ? op_last : op_full, acc, lane) ^ (uint32_t)__builtin_amdgcn_readlane((int)c, k);
  return acc ^ 0xFFFFFFFFu;
}

struct GlobalBytes {
  const uint8_t* p;
  __device__ __forceinline__ uint32_t u8(long i) const { return (uint32_t)rfl((int)p[i]); }  // every lane reads the same byte
};

}  // namespace

__global__ __launch_bounds__(64 * WAVES_PER_WG, 6) void cpx_cptv_inflate_kernel(CptvInflateArgs a) {
  __shared__ WaveLds s_lds[WAVES_PER_WG];
#ifdef CPX_INFLATE_CLOCK_PROBE
  const unsigned long probe_c0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int lane = threadIdx.x & 63;
  const int f = rfl((int)(blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6)));
  if (f >= a.B) return;
  cpx_cptv_file fi;  // the same for every lane: kept in scalar registers
  fi.in_offset = rfl64(a.files[f].in_offset);
  fi.in_bytes = rfl64(a.files[f].in_bytes);
  fi.out_offset = rfl64(a.files[f].out_offset);
  fi.out_capacity = rfl64(a.files[f].out_capacity);
  fi.slot_offset = rfl64(a.files[f].slot_offset);
  fi.slot_capacity = rfl(a.files[f].slot_capacity);
  cpx_cptv_file_result res;
  res.status = 0;
  res.n_frames = 0;
  res.out_bytes = 0;
  res.in_consumed = 0;
  res.header_bytes = 0;
  res.width = res.height = 0;
  res.reserved = 0;
  const uint8_t* const file = a.in + fi.in_offset;
  uint8_t* const out = a.out + fi.out_offset;
  // ---- gzip member header (RFC 1952) ----
  long start = -1;
  {
    GlobalBytes gb{file};
    const long n = fi.in_bytes;
    if (n >= 18 && gb.u8(0) == 0x1f && gb.u8(1) == 0x8b && gb.u8(2) == 8) {
      const int flg = (int)gb.u8(3);
      // FHCRC (a CRC-16 of the header, which zlib verifies) and the reserved flag bits (which zlib refuses) are left to
      // the host reader: the member is reported as "not a gzip member" here, never silently accepted
      long pos = (flg & 0xE2) ? n : 10;
      if (flg & 4) pos += 2 + (long)(gb.u8(10) | (gb.u8(11) << 8));
      if (flg & 8) {
        while (pos < n && gb.u8(pos) != 0) ++pos;
        ++pos;
      }
      if (flg & 16) {
        while (pos < n && gb.u8(pos) != 0) ++pos;
        ++pos;
      }
      if (flg & 2) pos += 2;
      if (pos + 8 <= n) start = pos;
    }
  }
  int status = 0;
  long n_out = 0;
  if (start < 0) {
    status = ERR_GZIP_HEADER;
  } else {
    WaveIO io;
    io.words = reinterpret_cast<const uint32_t*>(file);
    io.n_words = (fi.in_bytes + 3) >> 2;
    io.file_bits = fi.in_bytes * 8;
    io.lane = lane;
    io.out = out;
    io.cap = fi.out_capacity;
    io.lim = fi.out_capacity;
    io.n_flushed = 0;
    io.lit = 0;
    io.nlit = 0;
    io.lds = &s_lds[rfl((int)(threadIdx.x >> 6))];
    io.len_tab = lane < 29 ? (infl::length_base(lane) | (infl::length_extra(lane) << 16)) : 0;
    io.dist_tab = lane < 30 ? (infl::dist_base(lane) | (infl::dist_extra(lane) << 16)) : 0;
    io.file_bytes = file;
    io.seek_byte(start);
    status = infl::inflate(io);
    if (!io.flush() && status == 0) status = infl::ERR_OUTPUT;
    if (status == infl::ERR_OUTPUT && io.overrun()) status = infl::ERR_INPUT;
    n_out = io.n_flushed;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // every store of the stream complete before the section walk reads it back
    if (status == 0) {
      // trailer: crc32, isize; what follows must be the end of the file (a further member goes to the host path)
      const long used = (io.pos_bits() + 7) >> 3;
      res.in_consumed = used + 8;
      if (used + 8 != fi.in_bytes) {
        status = ERR_GZIP_TRAILER;
      } else {
        GlobalBytes gb{file};
        const uint32_t isize = gb.u8(used + 4) | (gb.u8(used + 5) << 8) | (gb.u8(used + 6) << 16) | (gb.u8(used + 7) << 24);
        if (isize != (uint32_t)n_out) status = ERR_GZIP_TRAILER;
        else if (n_out > 0) {
          const uint32_t want = gb.u8(used) | (gb.u8(used + 1) << 8) | (gb.u8(used + 2) << 16) | (gb.u8(used + 3) << 24);
          __builtin_amdgcn_s_waitcnt(0);
          if (wave_crc32(out, n_out, io.lds->crc_tab, lane) != want) status = ERR_GZIP_CRC;
        }
      }
    }
  }
  res.out_bytes = n_out;
  // ---- CPTV sections of the inflated bytes (this wave's own stores: visible to it in program order) ----
  if (status == 0) {
    __builtin_amdgcn_s_waitcnt(0);  // every store of the file has left the wave
    GlobalBytes ob{out};
    cpx_cptv_frame_slot* const slots = a.slots + fi.slot_offset;
    status = cptvidx::index_file(ob, n_out, fi.out_offset, fi.slot_capacity,
                                 [&](int i, const cpx_cptv_frame_slot& s) {
                                   if (lane == 0) slots[i] = s;
                                 },
                                 &res);
    if (a.header) {
      uint8_t* hb = a.header + (size_t)f * CPX_CPTV_HEADER_BYTES;
      const long nb = n_out < CPX_CPTV_HEADER_BYTES ? n_out : CPX_CPTV_HEADER_BYTES;
      for (int k = lane; k < CPX_CPTV_HEADER_BYTES; k += 64) hb[k] = k < nb ? out[k] : (uint8_t)0;
    }
  }
  res.status = status;
#ifdef CPX_INFLATE_CLOCK_PROBE
  {  // diagnostic build only: the shader clock this wave ran at, in MHz (s_memrealtime ticks at 100 MHz)
    const unsigned long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    res.reserved = (int)((c1 - probe_c0) * 100 / (r1 - probe_r0 + 1));
  }
#endif
  if (lane == 0) a.results[f] = res;
}

__global__ __launch_bounds__(256) void cpx_cptv_gather_kernel(CptvGatherArgs a) {
  const int b = blockIdx.x;
  const int f0 = a.clip_offsets[b], f1 = a.clip_offsets[b + 1];
  const cpx_cptv_frame_slot* src = a.slots + a.slot_offsets[b];
  for (int i = threadIdx.x; i < f1 - f0; i += blockDim.x) {
    const cpx_cptv_frame_slot s = src[i];
    a.frame_offsets[f0 + i] = s.offset;
    a.bit_widths[f0 + i] = s.bit_width;
    if (a.slots_out) a.slots_out[f0 + i] = s;
  }
}

int launch_cptv_inflate(const CptvInflateArgs& a, hipStream_t s) {
  const int wgs = (a.B + WAVES_PER_WG - 1) / WAVES_PER_WG;
  hipLaunchKernelGGL(cpx_cptv_inflate_kernel, dim3(wgs), dim3(64 * WAVES_PER_WG), 0, s, a);
  return 0;
}

int launch_cptv_gather(const CptvGatherArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL(cpx_cptv_gather_kernel, dim3(B), dim3(256), 0, s, a);
  return 0;
}

}  // namespace cpx
