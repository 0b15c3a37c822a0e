// TEST INFRASTRUCTURE ONLY: host build of classifier-pipeline_amd/csrc/cpx_inflate_core.h (the DEFLATE decoder the
// GPU runs one wavefront per file) behind a plain-array I/O policy, so that the decoding logic can be checked against
// zlib without a GPU (tests/test_inflate_cpu.py).  The product never loads this; it runs the HIP build of the same
// header (cpx_inflate.hip).
#include <stdint.h>
#include <string.h>

#include "cpx_inflate_core.h"

namespace {

struct HostIO {
  const uint8_t* in;
  long n_in, pos;      // pos: next input byte to load into the bit buffer
  uint64_t buf;
  int cnt;             // valid bits in buf
  long over;           // bits handed out beyond the input
  uint8_t* out;
  long cap, n_out;
  uint16_t ll[cpx::infl::LL_ENOUGH], dt[cpx::infl::D_ENOUGH];
  uint8_t lens_[cpx::infl::LENS_SCRATCH];
  uint16_t work_[cpx::infl::WORK_SCRATCH];
  uint16_t small_[32];
  static uint32_t ld16(const uint16_t* p) { return *p; }
  static int uni(int v) { return v; }
  int decode_symbols(const uint16_t* l, const uint16_t* d) { return cpx::infl::decode_symbols_generic(*this, l, d); }
  uint16_t* small() { return small_; }

  uint32_t bits() {
    while (cnt < 32) {
      uint64_t v = 0;
      if (pos < n_in) v = in[pos];
      else over += 8;
      ++pos;
      buf |= v << cnt;
      cnt += 8;
    }
    return (uint32_t)buf;
  }
  void drop(int n) {
    if (cnt < n) bits();
    buf >>= n;
    cnt -= n;
  }
  bool overrun() const { return over > cnt; }  // consumed (not merely buffered) bits beyond the input
  void align_byte() { drop(cnt & 7); }
  bool literal(uint32_t b) {
    if (n_out >= cap) return false;
    out[n_out++] = (uint8_t)b;
    return true;
  }
  int match(int len, int dist) {
    if (dist > n_out) return cpx::infl::ERR_DISTANCE;
    if (n_out + len > cap) return cpx::infl::ERR_OUTPUT;
    for (int i = 0; i < len; ++i, ++n_out) out[n_out] = out[n_out - dist];
    return cpx::infl::OK;
  }
  int stored(int len) {
    // the bit buffer holds whole bytes here: give them back
    pos -= cnt >> 3;
    buf = 0;
    cnt = 0;
    if (pos + len > n_in) return cpx::infl::ERR_INPUT;
    if (n_out + len > cap) return cpx::infl::ERR_OUTPUT;
    memcpy(out + n_out, in + pos, (size_t)len);
    n_out += len;
    pos += len;
    return cpx::infl::OK;
  }
  uint16_t* ll_table() { return ll; }
  uint16_t* d_table() { return dt; }
  uint8_t* lens() { return lens_; }
  uint16_t* work() { return work_; }
};

}  // namespace

// raw DEFLATE stream -> out; returns the status, *n_out = bytes produced, *consumed = input bytes used
extern "C" int inflate_host_raw(const uint8_t* in, long n_in, uint8_t* out, long cap, long* n_out, long* consumed) {
  HostIO io;
  io.in = in;
  io.n_in = n_in;
  io.pos = 0;
  io.buf = 0;
  io.cnt = 0;
  io.over = 0;
  io.out = out;
  io.cap = cap;
  io.n_out = 0;
  const int rc = cpx::infl::inflate(io);
  *n_out = io.n_out;
  *consumed = io.pos - (io.cnt >> 3);
  return rc;
}

extern "C" long inflate_host_gzip_header(const uint8_t* in, long n) { return cpx::infl::gzip_header_end(in, n); }
