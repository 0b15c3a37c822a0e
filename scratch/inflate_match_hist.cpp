#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "cpx_inflate_core.h"
struct HostIO {
  const uint8_t* in; long n_in, pos; uint64_t buf; int cnt; long over; uint8_t* out; long cap, n_out;
  uint16_t ll[cpx::infl::LL_ENOUGH], dt[cpx::infl::D_ENOUGH]; uint8_t lens_[cpx::infl::LENS_SCRATCH]; uint16_t work_[cpx::infl::WORK_SCRATCH]; uint16_t small_[32];
  long n_lit = 0, n_match = 0, match_bytes = 0, dist_hist[16] = {0}, dist_bytes[16] = {0}, len_le64 = 0, near_after_store[4] = {0};
  long last_match_end = -1;
  static uint32_t ld16(const uint16_t* p) { return *p; }
  static int uni(int v) { return v; }
  int decode_symbols(const uint16_t* l, const uint16_t* d) { return cpx::infl::decode_symbols_generic(*this, l, d); }
  uint16_t* small() { return small_; }
  uint16_t* ll_table() { return ll; } uint16_t* d_table() { return dt; } uint8_t* lens() { return lens_; } uint16_t* work() { return work_; }
  uint32_t bits() { while (cnt < 32) { uint64_t v = 0; if (pos < n_in) v = in[pos]; else over += 8; ++pos; buf |= v << cnt; cnt += 8; } return (uint32_t)buf; }
  void drop(int n) { buf >>= n; cnt -= n; }
  bool overrun() const { return over > cnt; }
  void align_byte() { drop(cnt & 7); }
  bool literal(uint32_t b) { if (n_out >= cap) return false; out[n_out++] = (uint8_t)b; ++n_lit; return true; }
  int match(int len, int dist) {
    if (dist > n_out) return cpx::infl::ERR_DISTANCE;
    if (n_out + len > cap) return cpx::infl::ERR_OUTPUT;
    int b = 0; while ((1 << b) < dist) ++b;   // bucket: dist <= 2^b
    dist_hist[b]++; dist_bytes[b] += len; ++n_match; match_bytes += len; if (len <= 64) ++len_le64;
    for (int i = 0; i < len; ++i, ++n_out) out[n_out] = out[n_out - dist];
    return cpx::infl::OK;
  }
  int stored(int len) { pos -= cnt >> 3; buf = 0; cnt = 0; if (pos + len > n_in) return cpx::infl::ERR_INPUT; if (n_out + len > cap) return cpx::infl::ERR_OUTPUT; memcpy(out + n_out, in + pos, len); pos += len; n_out += len; return cpx::infl::OK; }
};
int main(int argc, char** argv) {
  for (int a = 1; a < argc; ++a) {
    FILE* f = fopen(argv[a], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> in(n); fread(in.data(), 1, n, f); fclose(f);
    long hdr = cpx::infl::gzip_header_end(in.data(), n);
    uint32_t isize; memcpy(&isize, in.data() + n - 4, 4);
    std::vector<uint8_t> out(isize + 64);
    HostIO io; io.in = in.data() + hdr; io.n_in = n - hdr; io.pos = 0; io.buf = 0; io.cnt = 0; io.over = 0; io.out = out.data(); io.cap = isize; io.n_out = 0;
    int rc = cpx::infl::inflate(io);
    printf("%s rc %d out %ld literals %ld matches %ld match_bytes %ld (%.1f%% of output) len<=64 %.1f%% avg len %.1f\n", argv[a], rc, io.n_out, io.n_lit, io.n_match, io.match_bytes,
           100.0 * io.match_bytes / io.n_out, 100.0 * io.len_le64 / (io.n_match ? io.n_match : 1), (double)io.match_bytes / (io.n_match ? io.n_match : 1));
    long cum = 0;
    for (int b = 0; b < 16; ++b) { cum += io.dist_hist[b]; printf("  dist <= %5d: %8ld matches (cum %.1f%%)\n", 1 << b, io.dist_hist[b], 100.0 * cum / (io.n_match ? io.n_match : 1)); }
  }
}
