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

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ long rfl64(long v) {
  const unsigned lo = (unsigned)rfl((int)(v & 0xFFFFFFFFl)), hi = (unsigned)rfl((int)((unsigned long)v >> 32));
  return (long)(((unsigned long)hi << 32) | lo);
}

struct WaveLds {
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
  long bits_total;        // bits of the file from the start of the DEFLATE data
  long bits_used;         // bits dropped so far
  long next_chunk;        // index (in 64-dword chunks) of what `nxt` holds
  int cur, nxt;           // per lane: dword `lane` of the current / following chunk
  int widx;               // next dword of `cur`
  uint64_t buf;
  int cnt;
  int lane;
  // ---- output ----
  uint8_t* out;
  long cap, n_out;        // n_out counts the pending literals too
  long lim;               // = cap, or 0 once the input is exhausted
  int lit, nlit;
  WaveLds* lds;

  __device__ __forceinline__ int load_chunk(long chunk) const {
    const long w = chunk * 64 + lane;
    return w < n_words ? (int)words[w] : 0;
  }
  // position the reader at bit `bit` of the file (a multiple of 8)
  __device__ __forceinline__ void seek_byte(long byte) {
    const long w = byte >> 2;
    const long chunk = w >> 6;
    cur = load_chunk(chunk);
    nxt = load_chunk(chunk + 1);
    next_chunk = chunk + 1;
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
  __device__ __forceinline__ uint32_t next_word() {
    // a stream that runs past its input decodes the zero padding -- possibly as literals, for as long as the output
    // has room: checked here, once per 32 input bits (the literal path itself only tests the output limit)
    if (bits_used > bits_total) lim = 0;
    if (widx == 64) {
      cur = nxt;
      next_chunk += 1;
      nxt = load_chunk(next_chunk);
      widx = 0;
    }
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
    bits_used += n;
  }
  __device__ __forceinline__ bool overrun() const { return bits_used > bits_total; }
  __device__ __forceinline__ void align_byte() {
    const int n = (int)((8 - (bits_used & 7)) & 7);
    if (n) {
      if (cnt < n) (void)bits();
      drop(n);
    }
  }
  __device__ __forceinline__ void flush() {
    if (nlit > 0) {
      if (lane < nlit) out[n_out - nlit + lane] = (uint8_t)lit;
      nlit = 0;
    }
  }
  __device__ __forceinline__ bool literal(uint32_t b) {
    if (n_out >= lim) return false;
    // lane nlit of `lit` = b (the lane select goes through M0: one other scalar operand per vector instruction)
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(lit) : "s"(rfl((int)b)), "s"(rfl(nlit)) : "m0");
    nlit += 1;
    n_out += 1;
    if (nlit == 64) flush();
    return true;
  }
  __device__ __forceinline__ int match(int len, int dist) {
    if (dist > n_out) return infl::ERR_DISTANCE;
    if (n_out + len > cap) return infl::ERR_OUTPUT;
    flush();
    uint8_t* const dst = out + n_out;
    const uint8_t* const src = dst - dist;
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
    n_out += len;
    return infl::OK;
  }
  __device__ __forceinline__ int stored(int len, const uint8_t* file_bytes, long data_start) {
    flush();
    const long byte = data_start + (bits_used >> 3);
    if ((bits_used >> 3) + len > (bits_total >> 3)) return infl::ERR_INPUT;
    if (n_out + len > cap) return infl::ERR_OUTPUT;
    for (int k = 0; k < len; k += 64) {
      const int j = k + lane;
      if (j < len) out[n_out + j] = file_bytes[byte + j];
    }
    n_out += len;
    bits_used += (long)len * 8;
    seek_byte(byte + len);
    return infl::OK;
  }
  // the core's policy interface
  const uint8_t* file_bytes;
  long data_start;
  __device__ __forceinline__ int stored(int len) { return stored(len, file_bytes, data_start); }
  __device__ __forceinline__ uint16_t* ll_table() { return lds->ll; }
  __device__ __forceinline__ uint16_t* d_table() { return lds->dt; }
  __device__ __forceinline__ uint8_t* lens() { return lds->lens; }
  __device__ __forceinline__ uint16_t* work() { return lds->work; }
  __device__ __forceinline__ uint16_t* small() { return lds->small; }
  static __device__ __forceinline__ uint32_t ld16(const uint16_t* p) { return (uint32_t)rfl((int)*p); }
  static __device__ __forceinline__ int uni(int v) { return rfl(v); }
};

struct GlobalBytes {
  const uint8_t* p;
  __device__ __forceinline__ uint32_t u8(long i) const { return (uint32_t)rfl((int)p[i]); }  // every lane reads the same byte
};

}  // namespace

__global__ __launch_bounds__(64 * WAVES_PER_WG) void cpx_cptv_inflate_kernel(CptvInflateArgs a) {
  __shared__ WaveLds s_lds[WAVES_PER_WG];
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
      long pos = 10;
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
    io.bits_total = (fi.in_bytes - start) * 8;
    io.bits_used = 0;
    io.lane = lane;
    io.out = out;
    io.cap = fi.out_capacity;
    io.lim = fi.out_capacity;
    io.n_out = 0;
    io.lit = 0;
    io.nlit = 0;
    io.lds = &s_lds[threadIdx.x >> 6];
    io.file_bytes = file;
    io.data_start = start;
    io.seek_byte(start);
    status = infl::inflate(io);
    if (status == infl::ERR_OUTPUT && io.overrun()) status = infl::ERR_INPUT;
    io.flush();
    n_out = io.n_out;
    if (status == 0) {
      // trailer: crc32, isize; what follows must be the end of the file (a further member goes to the host path)
      const long used = start + ((io.bits_used + 7) >> 3);
      res.in_consumed = used + 8;
      if (used + 8 != fi.in_bytes) {
        status = ERR_GZIP_TRAILER;
      } else {
        GlobalBytes gb{file};
        const uint32_t isize = gb.u8(used + 4) | (gb.u8(used + 5) << 8) | (gb.u8(used + 6) << 16) | (gb.u8(used + 7) << 24);
        if (isize != (uint32_t)n_out) status = ERR_GZIP_TRAILER;
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
