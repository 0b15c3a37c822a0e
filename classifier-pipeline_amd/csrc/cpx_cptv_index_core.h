// cpx_cptv_index_core.h -- walk of the sections of an inflated CPTV v2 file: header fields, then per frame the
// field list (time on, last FFC time, temperatures, background flag, bits per delta, payload size) and the payload's
// offset.  Scalar code for the device (the wavefront that inflated the file runs it, csrc/cpx_inflate.hip) and the
// host (the zlib worker pool of csrc/cpx_cptv_host.cpp; tests/native/inflate_host.cpp).
//
// Replaces the section parser of the Rust reader the reference calls (python-cptv 0.0.8 CptvReader.get_header /
// next_frame; /root/reference/src/track/cliptrackextractor.py:108-129,160-162); same acceptance rules as the Python
// restatement cpx/cptv.py:CptvReader (which the tests compare it with).
//
// Byte reader `R`: uint32_t u8(long p) -- byte p of the inflated file.
#pragma once
#include <stdint.h>

#include "cpx.h"

#ifndef CPX_HD
#ifdef __HIPCC__
#define CPX_HD __host__ __device__
#else
#define CPX_HD
#endif
#endif

namespace cpx {
namespace cptvidx {

enum Status {
  OK = 0,
  ERR_NOT_CPTV = 20,     // magic
  ERR_VERSION = 21,      // version byte != 2
  ERR_NO_HEADER = 22,    // section H missing / header runs past the end / no resolution
  ERR_SECTION = 23,      // a frame section does not start with 'F'
  ERR_TRUNCATED = 24,    // fields or payload run past the end of the data
  ERR_FRAME = 25,        // bits per delta outside 1..32, payload shorter than the frame needs, 'w' / 'f' missing
  ERR_SLOTS = 26,        // more frames than slots
  ERR_NO_FRAMES = 27,
};

template <class R>
CPX_HD inline uint32_t u32le(R& r, long p) {
  return r.u8(p) | (r.u8(p + 1) << 8) | (r.u8(p + 2) << 16) | (r.u8(p + 3) << 24);
}

// n: inflated bytes; base: what is added to a payload's position to give cpx_cptv_frame_slot.offset (the file's
// offset inside the batch's payload buffer).  emit(i, slot) stores frame i's slot.
template <class R, class Emit>
CPX_HD inline int index_file(R& r, long n, long base, int slot_cap, Emit emit, cpx_cptv_file_result* res) {
  res->n_frames = 0;
  res->header_bytes = 0;
  res->width = res->height = 0;
  if (n < 7 || r.u8(0) != 'C' || r.u8(1) != 'P' || r.u8(2) != 'T' || r.u8(3) != 'V') return ERR_NOT_CPTV;
  if (r.u8(4) != 2) return ERR_VERSION;
  if (r.u8(5) != 'H') return ERR_NO_HEADER;
  long p = 7;
  int W = 0, H = 0;
  for (int k = (int)r.u8(6); k > 0; --k) {
    if (p + 2 > n) return ERR_NO_HEADER;
    const int len = (int)r.u8(p), code = (int)r.u8(p + 1);
    if (p + 2 + len > n) return ERR_NO_HEADER;
    if (code == 'X' && len >= 4) W = (int)u32le(r, p + 2);
    if (code == 'Y' && len >= 4) H = (int)u32le(r, p + 2);
    p += 2 + len;
  }
  if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return ERR_NO_HEADER;
  res->header_bytes = (int32_t)p;
  res->width = W;
  res->height = H;
  const long pix = (long)W * H;
  int nf = 0;
  while (p < n) {
    if (r.u8(p) != 'F') return ERR_SECTION;
    if (p + 2 > n) return ERR_TRUNCATED;
    int k = (int)r.u8(p + 1);
    p += 2;
    cpx_cptv_frame_slot s;
    s.offset = 0;
    s.bit_width = 0;
    s.time_on_ms = s.last_ffc_ms = 0;
    s.temp_c = s.last_ffc_temp_c = 0.0f;
    s.flags = 0;
    long nbytes = -1;
    for (; k > 0; --k) {
      if (p + 2 > n) return ERR_TRUNCATED;
      const int len = (int)r.u8(p), code = (int)r.u8(p + 1);
      if (p + 2 + len > n) return ERR_TRUNCATED;
      if (code == 't' && len >= 4) {
        s.time_on_ms = u32le(r, p + 2);
        s.flags |= CPX_CPTV_HAS_TIME_ON;
      } else if (code == 'c' && len >= 4) {
        s.last_ffc_ms = u32le(r, p + 2);
        s.flags |= CPX_CPTV_HAS_LAST_FFC;
      } else if (code == 'a' && len >= 4) {
        const uint32_t v = u32le(r, p + 2);
        s.temp_c = *reinterpret_cast<const float*>(&v);
      } else if (code == 'b' && len >= 4) {
        const uint32_t v = u32le(r, p + 2);
        s.last_ffc_temp_c = *reinterpret_cast<const float*>(&v);
      } else if (code == 'g' && len >= 1) {
        if (r.u8(p + 2) != 0) s.flags |= CPX_CPTV_BACKGROUND_FRAME;
      } else if (code == 'w' && len >= 1) {
        s.bit_width = (int32_t)r.u8(p + 2);
      } else if (code == 'f' && len >= 4) {
        nbytes = (long)u32le(r, p + 2);
      }
      p += 2 + len;
    }
    if (nbytes < 0 || s.bit_width < 1 || s.bit_width > 32) return ERR_FRAME;
    if (p + nbytes > n) return ERR_TRUNCATED;
    if (nbytes < 4 + ((pix - 1) * s.bit_width + 7) / 8) return ERR_FRAME;
    if (nf >= slot_cap) return ERR_SLOTS;
    s.offset = base + p;
    emit(nf, s);
    ++nf;
    res->n_frames = nf;
    p += nbytes;
  }
  return nf > 0 ? OK : ERR_NO_FRAMES;
}

}  // namespace cptvidx
}  // namespace cpx
