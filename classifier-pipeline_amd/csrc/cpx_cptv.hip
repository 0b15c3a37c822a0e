// cpx_cptv.hip -- CPTV v2 frame payload decode on the GPU: one workgroup per clip walks its frames
// (frame = previous + decoded difference), the previous frame living in registers in scan order.
// Per frame: the payload bytes are staged in LDS with coalesced loads, every thread unpacks a
// contiguous run of signed big-endian bit fields, a block-wide inclusive scan turns deltas into the
// running sum, and the result is scattered to snake order (odd rows reversed).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cpx_kernels.h"

namespace cpx {

namespace {
constexpr int DT = 1024;          // threads
constexpr int DCH = 20;           // values per thread: W*H <= 20480
typedef unsigned int u32;
typedef unsigned long long u64;
}  // namespace

__global__ __launch_bounds__(DT) void cpx_cptv_unpack_kernel(CptvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_pay[];  // frame payload (+ 8 bytes slack)
  __shared__ int s_wsum[DT / 64];
  const int b = blockIdx.x;
  const int f0 = a.clip_offsets[b], f1 = a.clip_offsets[b + 1];
  const int W = a.W, P = a.W * a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = tid * DCH;  // first scan index of this thread
  int prev[DCH];
#pragma unroll
  for (int j = 0; j < DCH; ++j) prev[j] = 0;
  for (int f = f0; f < f1; ++f) {
    const int w = a.bit_widths[f];
    const unsigned char* src = a.payload + a.frame_offsets[f];
    const int nbytes = 4 + (int)(((long long)(P - 1) * w + 7) >> 3);
    __syncthreads();  // previous frame's readers are done with s_pay
    for (int k = tid; k < nbytes + 8; k += DT) s_pay[k] = (k < nbytes) ? src[k] : 0;
    __syncthreads();
    // ---- unpack this thread's run and its local inclusive sums ----
    int loc[DCH];
    int run = 0;
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      const int i = i0 + j;
      int d = 0;
      if (i == 0) {
        d = (int)((u32)s_pay[0] | ((u32)s_pay[1] << 8) | ((u32)s_pay[2] << 16) | ((u32)s_pay[3] << 24));
      } else if (i < P) {
        const long long bit = (long long)(i - 1) * w;
        const int by = 4 + (int)(bit >> 3), sh = (int)(bit & 7);
        u64 win = 0;  // 40 bits, big endian
#pragma unroll
        for (int k = 0; k < 5; ++k) win = (win << 8) | s_pay[by + k];
        const u32 field = (u32)((win >> (40 - sh - w)) & ((w == 32) ? 0xFFFFFFFFull : ((1ull << w) - 1ull)));
        d = (w == 32) ? (int)field : ((int)(field << (32 - w)) >> (32 - w));  // sign extension
      }
      run += d;
      loc[j] = run;
    }
    // ---- block-wide exclusive scan of the per-thread totals ----
    int incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wave; ++k) base += s_wsum[k];
    const int excl = base + incl - run;
    // ---- accumulate onto the previous frame and scatter to snake order ----
    uint16_t* out = a.frames_out + (size_t)f * P;
#pragma unroll
    for (int j = 0; j < DCH; ++j) {
      const int i = i0 + j;
      if (i < P) {
        prev[j] += excl + loc[j];
        const int y = i / W, xs = i - y * W;
        const int x = (y & 1) ? (W - 1 - xs) : xs;
        out[y * W + x] = (uint16_t)prev[j];
      }
    }
  }
}

int launch_cptv_unpack(const CptvArgs& a, int B, hipStream_t s) {
  if (a.W * a.H > DT * DCH) return -2;
  const size_t lds = 4 + ((size_t)(a.W * a.H - 1) * 32 + 7) / 8 + 16;  // worst case: 32-bit fields
  static bool lds_ready[64];
  if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(cpx_cptv_unpack_kernel), lds_ready, 160 * 1024 - 1024)) return -1;
  hipLaunchKernelGGL(cpx_cptv_unpack_kernel, dim3(B), dim3(DT), lds, s, a);
  return 0;
}

}  // namespace cpx
