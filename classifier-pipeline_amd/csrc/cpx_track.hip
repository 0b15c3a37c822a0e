// cpx_track.hip -- track-stage kernels for gfx950 (MI355X), one workgroup per
// clip-frame, the whole frame resident on chip between the single streaming
// pass over HBM and the (sparse) outputs.
//
// Follows, step for step, the reference arithmetic of
//   track/cliptrackextractor.py:198-247  (process_frame)
//   track/cliptracker.py:93-122          (_get_filtered_frame)
//   ml_tools/imageprocessing.py:151-169  (normalize), :240-248 (detect_objects)
//   piclassifier/motiondetector.py:197-248 (WeightedBackground)
//   track/cliptracker.py:249-261,316-318 (delta frame + np.var per region)
//   track/clip.py:474-487                (ClipStats.add_frame)
// with the OpenCV operator semantics of SURVEY.md Appendix A (8-bit binomial
// blur with one rounding, floor-threshold, 1x2 close, 8-connected labelling in
// 2x2-block raster order).  All image arithmetic is integer / IEEE f32 / f64 in
// the same order as NumPy evaluates it, so results are bit-identical.
//
// Data flow per workgroup (= one frame of one clip):
//   phase 1  one coalesced streaming pass: thermal (u16x4), window-leaving frame,
//            window sum, background (ping-pong), weight counters -> filtered
//            (registers + HBM), background update (HBM), block reductions
//   phase 2  shifted/clipped frame in registers -> min/max
//   phase 3  f32 normalise -> u8 image in LDS
//   phase 4  separable 5x5 binomial blur in LDS (u8 -> u16 -> 1 bit/pixel)
//   phase 5  1x2 close on 64-bit row words
//   phase 6  run-based union-find over row words (LDS atomics), 8-connectivity
//   phase 7  per-run statistics -> components, 2x2-block raster ordering
//   phase 8  label image (optional), one wave per component delta-variance
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <stdint.h>

#include "cpx_kernels.h"

namespace cpx {

namespace {

constexpr int NT = CPX_TRACK_THREADS;  // threads per workgroup
constexpr int NWAVE = NT / 64;
constexpr int NCH = CPX_TRACK_CHUNKS;  // 4-pixel chunks per thread
constexpr int RW = 3;                  // 64-bit words per bit row (W <= 192)
constexpr int CAP = CPX_TRACK_LDS_COMPONENTS;
typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// uniform base + 32-bit unsigned byte offset: the form the compiler turns into `global_load v, v_off, s[base]`
// (one offset register per lane instead of a 64-bit address pair per array: the streaming pass touches seven arrays)
template <typename T>
__device__ __forceinline__ T* at_off(T* base, unsigned bytes) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + bytes);
}
template <typename T>
__device__ __forceinline__ const T* at_off(const T* base, unsigned bytes) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + bytes);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T w = __shfl_xor(v, o);
    v = w < v ? w : v;
  }
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T w = __shfl_xor(v, o);
    v = w > v ? w : v;
  }
  return v;
}

// 32-bit reductions of a fully active wave on the DPP network (no LDS traffic, unlike the ds_bpermute behind
// __shfl_xor): butterflies inside quads, mirrors inside rows of 16, then row broadcasts; the total lands in lane 63
// and comes back as a scalar.  OP(x, x) must be x-neutral for `self` as the fill value (min / max / or), sums fill 0.
#define CPX_DPP_STEP(OP, X, FILL, CTRL, ROWMASK) X = OP(X, __builtin_amdgcn_update_dpp(FILL, X, CTRL, ROWMASK, 0xF, false))
#define CPX_DPP_REDUCE(OP, X, FILL)             \
  CPX_DPP_STEP(OP, X, FILL, 0xB1, 0xF);  /* quad_perm [1,0,3,2] */ \
  CPX_DPP_STEP(OP, X, FILL, 0x4E, 0xF);  /* quad_perm [2,3,0,1] */ \
  CPX_DPP_STEP(OP, X, FILL, 0x141, 0xF); /* row_half_mirror */     \
  CPX_DPP_STEP(OP, X, FILL, 0x140, 0xF); /* row_mirror */          \
  CPX_DPP_STEP(OP, X, FILL, 0x142, 0xA); /* row_bcast:15 -> rows 1, 3 */ \
  CPX_DPP_STEP(OP, X, FILL, 0x143, 0xC); /* row_bcast:31 -> rows 2, 3 */
__device__ __forceinline__ int dpp_add(int a, int b) { return a + b; }
__device__ __forceinline__ int dpp_imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int dpp_imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int dpp_umin(int a, int b) { return (u32)a < (u32)b ? a : b; }
__device__ __forceinline__ int dpp_umax(int a, int b) { return (u32)a > (u32)b ? a : b; }
__device__ __forceinline__ int wave_sum(int v) {
  CPX_DPP_REDUCE(dpp_add, v, 0)
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ u32 wave_sum(u32 v) { return (u32)wave_sum((int)v); }
__device__ __forceinline__ int wave_min(int v) {
  CPX_DPP_REDUCE(dpp_imin, v, v)
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max(int v) {
  CPX_DPP_REDUCE(dpp_imax, v, v)
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ u32 wave_min(u32 v) {
  int x = (int)v;
  CPX_DPP_REDUCE(dpp_umin, x, x)
  return (u32)__builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ u32 wave_max(u32 v) {
  int x = (int)v;
  CPX_DPP_REDUCE(dpp_umax, x, x)
  return (u32)__builtin_amdgcn_readlane(x, 63);
}
// lanes of a fully active wave for which `pred` holds (a scalar: one compare + s_bcnt1 per call)
__device__ __forceinline__ u32 wave_count(bool pred) { return (u32)__builtin_popcountll(__builtin_amdgcn_ballot_w64(pred)); }
// the same for both 16-bit halves of a packed register against a uniform bound: how many halves of the wave are
// <= bound.  Written out because the compiler, given ballots, issues every compare of an unrolled loop first and
// spills the lane masks (v_writelane per mask); here a mask lives in VCC for one instruction, and the halves are
// selected by the compare itself (SDWA), so the packed registers are never unpacked.
__device__ __forceinline__ u32 wave_count_le_halves(u32 packed, u32 bound) {
  u32 c0, c1;
  asm volatile(
      "v_cmp_ge_u32_sdwa vcc, %2, %3 src0_sel:DWORD src1_sel:WORD_0\n\t"
      "s_bcnt1_i32_b64 %0, vcc\n\t"
      "v_cmp_ge_u32_sdwa vcc, %2, %3 src0_sel:DWORD src1_sel:WORD_1\n\t"
      "s_bcnt1_i32_b64 %1, vcc"
      : "=&s"(c0), "=&s"(c1)
      : "s"(bound), "v"(packed)
      : "vcc", "scc");
  return c0 + c1;
}

__device__ __forceinline__ u32 pk_add_u16(u32 x, u32 y) {  // v_pk_add_u16
  typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(u32, (us2_t)(__builtin_bit_cast(us2_t, x) + __builtin_bit_cast(us2_t, y)));
}

// ---- bit-row helpers (rows are RW little-endian 64-bit words, bit x = pixel x) ----
__device__ __forceinline__ int run_start(const u64* row, int x) {
  int w = x >> 6;
  int b = x & 63;
  u64 inv = ~row[w] & ((1ull << b) - 1ull);
  while (true) {
    if (inv) return (w << 6) + 64 - __builtin_clzll(inv);
    if (w == 0) return 0;
    --w;
    inv = ~row[w];
  }
}
// last pixel of the run containing x (rows are zero beyond W, W < 64*RW)
__device__ __forceinline__ int run_end(const u64* row, int x) {
  int w = x >> 6;
  int b = x & 63;
  u64 inv = ~row[w] & ~((1ull << b) - 1ull);
  while (true) {
    if (inv) return (w << 6) + __builtin_ctzll(inv) - 1;
    if (w == RW - 1) return 64 * RW - 1;
    ++w;
    inv = ~row[w];
  }
}

__device__ __forceinline__ u32 uf_find(volatile u32* parent, u32 x) {
  u32 p = parent[x];
  while (p != x) {
    x = p;
    p = parent[x];
  }
  return x;
}
__device__ __forceinline__ void uf_union(u32* parent, u32 a, u32 b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a > b) {
      u32 t = a;
      a = b;
      b = t;
    }
    u32 old = atomicMin(&parent[b], a);
    if (old == b) return;
    b = old;
  }
}



}  // namespace

// ---------------------------------------------------------------------------
// init: WeightedBackground.__init__ + first process_frame
// (motiondetector.py:178-211; cliptrackextractor.py:129-139)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cpx_init_kernel(TrackArgs a, int keep) {
  const int b = blockIdx.x;
  const int W = a.W, H = a.H, P = W * H, e = a.edge;
  const uint16_t* F = a.frames + (size_t)a.clip_first[b] * P;
  uint16_t* bg0 = a.bg + (size_t)b * 2 * P;
  u32* ws = a.wsum + (size_t)b * P;
  uint16_t* kc = a.kcnt + (size_t)b * P;
  if (keep) {
    // CPX_TRACK_KEEP_BACKGROUND: background, weights and average continue from where the previous run left them
    // (its last frame wrote ping-pong slot n_done & 1); only the 45-frame window starts again
    const ClipState old = a.cstate[b];
    const int src = old.n_done & 1;
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
      bg0[(1 - src) * P + p] = bg0[src * P + p];
      ws[p] = 0;
    }
    if (threadIdx.x == 0) {
      ClipState st = old;
      st.prev_fmin = st.prev_fmax = 0;
      st.has_prev = 0;
      st.n_done = 0;
      a.cstate[b] = st;
      if (a.bgavg) a.bgavg[b] = st.bg_average;
    }
    return;
  }
  u64 s = 0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    int y = p / W, x = p - y * W;
    int cy = clampi(y, e, H - 1 - e), cx = clampi(x, e, W - 1 - e);
    int v = F[cy * W + cx];
    bg0[p] = (uint16_t)v;
    bg0[P + p] = (uint16_t)v;
    ws[p] = 0;
    kc[p] = 0;
    if (cy == y && cx == x) s += (u64)v;
  }
  s = wave_sum(s);
  __shared__ u64 part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 tot = part[0] + part[1] + part[2] + part[3];
    ClipState st;
    st.bg_average = (double)tot / (double)((W - 2 * e) * (H - 2 * e));  // np.average(frame), un-rounded
    st.prev_fmin = 0;
    st.prev_fmax = 0;
    st.has_prev = 0;
    st.n_done = 0;
    a.cstate[b] = st;
    if (a.bgavg) a.bgavg[b] = st.bg_average;
  }
}

// ---------------------------------------------------------------------------
// one processed frame of every clip
// ---------------------------------------------------------------------------
// mode 0: whole frame step.  With denoise the step is split around the NLM kernel: mode 1 = front
// (phases 1-3: streaming pass, normalised uint8 image -> HBM), mode 2 = back (phases 4-8 on the
// denoised image); the scalars crossing the split travel in FrameCarry.
namespace {
// one processed frame (number t) of clip b; cs = the clip's state before / after the frame (uniform)
// entries of the weight tables kept in LDS (two workgroups of 80 KB share a CU): the integer thresholds every pixel
// looks up (4 bytes each) and the float64 weights behind the rare exact comparison (8 bytes each)
constexpr int WTHR_LDS = 1024;
constexpr int WTAB_LDS = 512;
__host__ __device__ inline size_t wtab_lds_offset(int W, int H) {
  const size_t P = (size_t)W * H;
  const size_t used = 3 * P + 2 * (size_t)H * RW * 8 + (size_t)9 * CAP * 4 + (NWAVE + 1) * sizeof(Red1) + NWAVE * 2 * sizeof(int) + 16 +
                      3 * NWAVE * sizeof(u32) + 16 + 16 + 16;  // (+ 16: s_keep, the record's scalars parked across the labelling phases; + 16: s_mag)
  return (used + 15) & ~(size_t)15;
}
// values every lane holds alike (read from LDS or through a vector load) -> scalar registers
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
  const long long q = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(q & 0xFFFFFFFFll)), hi = __builtin_amdgcn_readfirstlane((int)(q >> 32));
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ ClipState uniform_state(const ClipState& s) {
  ClipState o;
  o.bg_average = uni(s.bg_average);
  o.prev_fmin = uni(s.prev_fmin);
  o.prev_fmax = uni(s.prev_fmax);
  o.has_prev = uni(s.has_prev);
  o.n_done = uni(s.n_done);
  return o;
}
// (the arguments are read through the kernel-argument segment, see the kernel below)
typedef __attribute__((address_space(4))) const TrackArgs KernArgs;
// PK (round 6): the per-pixel count of consecutive kept frames rides in the top ten bits of the pixel's window sum (22 bits: 64
// frames x 65535 at most) instead of in an array of its own -- 2 B read and 2 B written per pixel and frame less, a sixth of what
// this HBM-bound kernel moves.  Only for a fresh batch of at most 1023 processed frames per clip (the count cannot outgrow ten
// bits); the host unpacks the state (cpx_unpack_state_kernel) before anything continues from it (cpx_api.cpp: track_run).
template <bool PK>
__device__ __forceinline__ void frame_step(KernArgs& a, const int b, const int pbase, const int t, const int mode,
                                           ClipState& cs, unsigned char* smem) {
  const int W = a.W, H = a.H, P = W * H, e = a.edge;
  // opaque per frame: otherwise everything that depends only on the thread index (pixel coordinates, clamps, row
  // tests of every chunk) is hoisted out of the frame loop and kept live across it -- 100 bytes of scratch per lane
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
  const int SW = W >> 1;  // run-start slots per row
  const int nchunk = P >> 2;

  // ---- LDS ---------------------------------------------------------------
  // everything is carved from one dynamic region so that its base stays 16-byte aligned
  // [0, P)            u8 image           | later: component id per slot (u16 [H*SW])
  // [P, 3P)           u16 h-blur         | later: union-find parents (u32 [H*SW])
  unsigned char* s_u8 = smem;
  uint16_t* s_cid = reinterpret_cast<uint16_t*>(smem);
  uint16_t* s_tmp = reinterpret_cast<uint16_t*>(smem + P);
  u32* s_par = reinterpret_cast<u32*>(smem + P);
  u64* s_rowI = reinterpret_cast<u64*>(smem + 3 * P);
  u64* s_rowE = s_rowI + H * RW;
  u32* s_stat = reinterpret_cast<u32*>(s_rowE + H * RW);  // [8][CAP]
  u32* s_rank = s_stat + 8 * CAP;                         // [CAP]
  Red1* s_red = reinterpret_cast<Red1*>(s_rank + CAP);    // [NWAVE]
  int* s_red2 = reinterpret_cast<int*>(s_red + NWAVE);    // [NWAVE][2]
  u32* s_ncomp_p = reinterpret_cast<u32*>(s_red2 + 2 * NWAVE);
  Red1* s_R = reinterpret_cast<Red1*>(s_ncomp_p + 2);
  // [WTAB_LDS], filled by the kernel.  Typed as LDS: as a generic pointer the lookup below merges with its fall-back
  // into ONE flat load of a selected address, which waits for every outstanding vector-memory operation
  typedef __attribute__((address_space(3))) const double LdsDouble;
  typedef __attribute__((address_space(3))) const u32 LdsU32;
  // four scalars of the frame record (avg_change, norm_min, norm_max, threshold) are computed in phase 3 and stored at the very
  // end of the step; kept in registers in between they cost four of the 64 the labelling phases have and went to scratch
  // (36 B per lane of scratch = 12 % more HBM traffic than the layout needs, profiles/r05_e2e_pmc.json).  Parked here instead.
  int* s_keep = reinterpret_cast<int*>(smem + wtab_lds_offset(W, H)) - 4;
  u32* s_mag = reinterpret_cast<u32*>(s_keep) - 4;  // the frame's normalisation as an integer division: multiplier, shift (phase 3)
  LdsDouble* s_wtab = (LdsDouble*)(smem + wtab_lds_offset(W, H));
  LdsU32* s_wthr = (LdsU32*)(smem + wtab_lds_offset(W, H) + WTAB_LDS * sizeof(double));

  const int fidx = a.proc_idx[pbase + t];
  const int oidx = (t >= a.window) ? a.proc_idx[pbase + t - a.window] : -1;
  const int nwin = (t + 1 < a.window) ? (t + 1) : a.window;
  // window_sum // n for window_sum < 2^22, n <= 64: one v_mul_hi_u32 with M = floor(2^32 / n) + 1 (the error term
  // x * (M - 2^32 / n) / 2^32 < 2^-10 cannot carry the quotient over an integer: fractions are multiples of 1 / n)
  const u32 div_magic = (nwin > 1) ? (u32)(0xFFFFFFFFu / (u32)nwin + 1u) : 0u;
  const uint16_t* F = a.frames + (size_t)fidx * P;
  const uint16_t* O = (oidx >= 0) ? a.frames + (size_t)oidx * P : nullptr;
  const uint16_t* bg_old = a.bg + ((size_t)b * 2 + (t & 1)) * P;
  uint16_t* bg_new = a.bg + ((size_t)b * 2 + ((t + 1) & 1)) * P;
  u32* ws = a.wsum + (size_t)b * P;
  uint16_t* kc = a.kcnt + (size_t)b * P;
  float* filt_cur;
  const float* filt_prev;
  if (a.filtered_out) {
    filt_cur = a.filtered_out + (size_t)fidx * P;
    filt_prev = (t > 0) ? a.filtered_out + (size_t)a.proc_idx[pbase + t - 1] * P : nullptr;
  } else {
    filt_cur = a.filt_state + ((size_t)b * 2 + (t & 1)) * P;
    filt_prev = a.filt_state + ((size_t)b * 2 + ((t + 1) & 1)) * P;
  }
  // who owns the background (include/cpx.h, CPX_TRACK_*): a frozen frame leaves background, weights and average as
  // they are (post_process_file skips FFC-affected frames, clipclassifier.py:510-511; update_background = False)
  const bool freeze = (a.flags & CPX_TRACK_FREEZE_BACKGROUND) ||
                      ((a.flags & CPX_TRACK_FREEZE_ON_FFC) && a.proc_ffc[pbase + t] != 0);
  // split steps: the front half of step t+1 may run while the back half of step t has not yet written the clip
  // state, so the background average travels front -> front (a.bgavg) and front -> back (the carry)
  const int slot = t & 1;
  if (mode == 1) cs.bg_average = a.bgavg[b];

  int avg_change = 0, mn = 0, mx = 0, ithr = 0;
  float thresh = 0.0f;
  if (mode != 2) {
  // ---- phase 1a: thermal frame -> LDS (phase 1b reads it there), sum / min / max ----------------
  // (np.median(thermal) of ClipStats is not on the dependency chain of the clip's frames: cpx_median_kernel
  // computes it for all frames at once and writes its word of the record; the step's store of the record leaves that word alone)
  {
    u32 sumpix = 0, minpix = 0xFFFFFFFFu, maxpix = 0;
    // all loads first and unconditionally (clamped address): a load inside `if (c < nchunk)` is waited for before
    // the next one is issued -- five dependent trips to HBM instead of one
    uint2 q[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      q[i] = *reinterpret_cast<const uint2*>(at_off(F, (unsigned)min(tid + i * NT, nchunk - 1) << 3));
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * NT;
      if (c < nchunk) {
        *reinterpret_cast<uint2*>(s_tmp + (c << 2)) = q[i];
        const u32 v0 = q[i].x & 0xFFFFu, v1 = q[i].x >> 16, v2 = q[i].y & 0xFFFFu, v3 = q[i].y >> 16;
        sumpix += v0 + v1 + v2 + v3;
        minpix = min(min(minpix, v0), min(v1, min(v2, v3)));
        maxpix = max(max(maxpix, v0), max(v1, max(v2, v3)));
      }
    }
    sumpix = wave_sum(sumpix);
    minpix = wave_min(minpix);
    maxpix = wave_max(maxpix);
    if (lane == 0) {
      s_red[wave].sumpix = sumpix;
      s_red[wave].minpix = minpix;
      s_red[wave].maxpix = maxpix;
    }
    // zero the bit rows / counters while we are at a barrier anyway
    for (int i = tid; i < 2 * H * RW; i += NT) s_rowI[i] = 0ull;
    if (tid == 0) *s_ncomp_p = 0;
  }
  __syncthreads();
  u32 sumpix = 0, minpix = 0xFFFFFFFFu, maxpix = 0;
#pragma unroll
  for (int w = 0; w < NWAVE; ++w) {
    sumpix += s_red[w].sumpix;
    minpix = min(minpix, s_red[w].minpix);
    maxpix = max(maxpix, s_red[w].maxpix);
  }
  // (the same in every lane: keep them in scalar registers)
  sumpix = (u32)uni((int)sumpix);
  minpix = (u32)uni((int)minpix);
  maxpix = (u32)uni((int)maxpix);
  // avg_change = int(round(np.average(thermal) - background.average))  (cliptracker.py:103-105)
  const double mean_thermal = (double)sumpix / (double)P;
  avg_change = (int)rint(mean_thermal - cs.bg_average);

  // ---- phase 1b: the streaming pass over the clip state -----------------------------------------------
  // x = max(thermal - background - avg_change, 0) goes to LDS as 17 bits (u16 plane + 1-bit-in-a-byte plane)
  Red1 r;
  r.sumpix = sumpix;
  r.minpix = minpix;
  r.maxpix = maxpix;
  r.fmin = 0x7FFFFFFF;
  r.fmax = -0x7FFFFFFF - 1;
  r.sumbg = 0;
  r.changed = 0;
  r.sumabs = 0;
  u32 sabs = 0;  // this lane's share of sum |filtered|: <= 20 pixels x 65535
  // Two pixels per lane and step (4-byte lanes: a wave instruction still covers 256 contiguous bytes), software
  // pipelined: the state of step i+1 is requested before step i is computed, so a workgroup pays the trip to HBM once
  // instead of once per step (a frame's 42 us in isolation were half this pass: five dependent trips).  Everything is
  // branch-free around the loads -- a conditional load makes the compiler drain all outstanding loads at the join.
  {
    const int npair = P >> 1;
    const bool edge1 = (e == 1);  // the usual border: the clamped background of a border pixel is its neighbour in the pair
    // without a frame leaving the window any readable address does (the value is discarded)
    const uint16_t* Osafe = O ? O : F;
    const u32 omask = O ? 0xFFFFFFFFu : 0u;
    struct Pair { u32 bg, old, kc; uint2 ws; };
    // a lane's pair walks the frame in strides of 2 NT pixels: row / column advance by constants (one compare per
    // round) instead of a division by the run-time width per request and per step
    struct Coord { int p0, yx; };  // first pixel; row << 8 | column (W < 192)
    const int dY = (2 * NT) / W, dX = 2 * NT - dY * W;  // uniform
    auto coord_of = [&](int c) -> Coord {
      const int p0 = c << 1;
      const int y = p0 / W;
      return Coord{p0, (y << 8) | (p0 - y * W)};
    };
    auto next = [&](const Coord& q) -> Coord {
      int x = (q.yx & 0xFF) + dX, y = (q.yx >> 8) + dY;
      const bool wrap = x >= W;
      x = wrap ? x - W : x;
      y = wrap ? y + 1 : y;
      return Coord{q.p0 + 2 * NT, (y << 8) | x};
    };
    auto request = [&](const Coord& q0) -> Pair {
      unsigned p0u = (unsigned)q0.p0;
      asm volatile("" : "+v"(p0u));  // opaque: base + offset addressing instead of one pointer induction variable per array
      const int y = q0.yx >> 8, x0 = q0.yx & 0xFF;
      const int cy = clampi(y, e, H - 1 - e);
      Pair q;
      // aligned pair of the clamped row; the column clamp is resolved on the data (edge1) or by two loads (any other edge)
      if (edge1) {
        // the clamped row is the pixel's own, or (top / bottom row) the one next to it: no multiply
        const int rowadj = (y == 0) ? W : ((y == H - 1) ? -W : 0);
        q.bg = *reinterpret_cast<const u32*>(at_off(bg_old, (unsigned)((int)p0u + rowadj) << 1));
      } else {
        const u32 b0 = *at_off(bg_old, (unsigned)(cy * W + clampi(x0, e, W - 1 - e)) << 1);
        const u32 b1 = *at_off(bg_old, (unsigned)(cy * W + clampi(x0 + 1, e, W - 1 - e)) << 1);
        q.bg = b0 | (b1 << 16);
      }
      q.old = *reinterpret_cast<const u32*>(at_off(Osafe, p0u << 1));
      q.ws = *reinterpret_cast<const uint2*>(at_off(ws, p0u << 2));
      if constexpr (PK) q.kc = 0u;
      else q.kc = *reinterpret_cast<const u32*>(at_off(kc, p0u << 1));
      return q;
    };
    auto step = [&](const Coord& q0, const Pair& cur) {
      unsigned p0u = (unsigned)q0.p0;
      asm volatile("" : "+v"(p0u));
      const int p0 = (int)p0u;
      const int y = q0.yx >> 8, x0 = q0.yx & 0xFF;
      const bool row_in = (y >= e && y <= H - 1 - e);
      const u32 pq = *reinterpret_cast<const u32*>(s_tmp + p0);
      const int pix[2] = {(int)(pq & 0xFFFFu), (int)(pq >> 16)};
      int bgv[2] = {(int)(cur.bg & 0xFFFFu), (int)(cur.bg >> 16)};
      if (edge1) {  // x0 is even, W is even: only pixel 0 can sit on the left border, only pixel 1 on the right one
        bgv[0] = (x0 == 0) ? bgv[1] : bgv[0];
        bgv[1] = (x0 == W - 2) ? bgv[0] : bgv[1];
      }
      const u32 oldq = cur.old & omask;
      const int oldp[2] = {(int)(oldq & 0xFFFFu), (int)(oldq >> 16)};
      u32 wsv[2] = {cur.ws.x, cur.ws.y};
      int kv[2] = {(int)(cur.kc & 0xFFFFu), (int)(cur.kc >> 16)};
      if constexpr (PK) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          kv[j] = (int)(wsv[j] >> 22);
          wsv[j] &= 0x3FFFFFu;
        }
      }
      int nb[2];
      float fo[2];
      u32 xs2[2];
      // |pix - bg| of both pixels + the running sum in one v_sad_u16 on the packed pairs (written per pixel on the unpacked
      // values the compiler spent 11 instructions per two steps on SDWA min / max pairs)
#ifndef CPX_TRACK_SAD32
      sabs = __builtin_amdgcn_sad_u16(pq, (u32)bgv[0] | ((u32)bgv[1] << 16), sabs);
#else
      sabs = __usad((u32)pix[0], (u32)bgv[0], sabs);
      sabs = __usad((u32)pix[1], (u32)bgv[1], sabs);
#endif
      // interior test per pixel; with the usual one-pixel border the column tests are the ones the background selects made
      bool colin[2];
      if (edge1) {
        colin[0] = x0 != 0;
        colin[1] = x0 != W - 2;
      } else {
        colin[0] = x0 >= e && x0 <= W - 1 - e;
        colin[1] = x0 + 1 >= e && x0 + 1 <= W - 1 - e;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int d = pix[j] - bgv[j];  // filtered = float32(pix) - background (cliptrackextractor.py:212)
        fo[j] = (float)d;
        r.fmin = min(r.fmin, d);
        r.fmax = max(r.fmax, d);
        int xs = d - avg_change;  // cliptracker.py:109-114
        xs = xs < 0 ? 0 : xs;
        xs2[j] = (u32)xs;
        // background feed: np.int32(np.mean(last <=45 frames)) == window_sum // n (cliptrackextractor.py:173-176)
        wsv[j] = wsv[j] + (u32)pix[j] - (u32)oldp[j];
        const int f = (nwin > 1) ? (int)__umulhi(wsv[j], div_magic) : (int)wsv[j];  // == wsv / nwin exactly
        nb[j] = bgv[j];
#ifndef CPX_TRACK_COLTEST
        if (row_in && colin[j]) {
#else
        if (row_in && x0 + j >= e && x0 + j <= W - 1 - e) {
#endif
          // motiondetector.py:212-223: bg' = bg if bg < f - w else f ; w' = w + add if (same) else 0
          // keep <=> bg < fl64(f - w_k), w_k = k-fold float64 accumulation of weight_add.  For integers bg, f that is
          // d = f - bg >= hi_k with hi_k = floor(w_k) + 1 from a table (csrc/cpx_api.cpp:weight_thresholds) -- unless
          // w_k lies within 1e-6 of an integer m without being one: then d = m is decided by NumPy's float64
          // expression itself (rounding of f - w_k against bg), d > m keeps, d < m does not.  Both tables' first
          // entries sit in LDS (copied once per clip): no vector-memory operation on the common path, so nothing
          // waits for the prefetched loads; longer runs of kept frames go to the tables in memory.
          const int kk = kv[j];
          const int d2 = f - bgv[j];
          // (the table holds 2 hi_k - near_k: keep <=> 2 d + 1 > entry; the float64 expression decides on equality, which
          // only an odd entry -- a near one -- can reach)
          const int th = (int)s_wthr[min(kk, WTHR_LDS - 1)];
          const int c2 = 2 * d2 + 1;
          bool keep = c2 > th;
          if (kk >= WTHR_LDS || c2 == th) {
            // (the loads are consumed INSIDE the branch: a value merged after it would put the wait for them, and
            // with it for every prefetched load, on the common path)
            int far;
            if (kk < WTAB_LDS) far = (double)bgv[j] < (double)f - s_wtab[kk];
            else far = (double)bgv[j] < (double)f - a.wtab[kk];
            asm volatile("" : "+v"(far));
            keep = far != 0;
          }
          if (freeze) keep = true;
          const int nv = keep ? bgv[j] : f;
          if (!freeze) kv[j] = keep ? kv[j] + 1 : 0;
          r.changed |= (u32)(nv ^ bgv[j]);  // non-zero <=> some background pixel changed
          r.sumbg += (u32)nv;
          nb[j] = nv;
        }
      }
      // x as 17 bits: u16 plane + one byte plane (phase 3 reads them back)
      *reinterpret_cast<u32*>(s_tmp + p0) = (xs2[0] & 0xFFFFu) | (xs2[1] << 16);
      *reinterpret_cast<uint16_t*>(s_u8 + p0) = (uint16_t)((xs2[0] >> 16) | ((xs2[1] >> 16) << 8));
      *reinterpret_cast<float2*>(at_off(filt_cur, p0u << 2)) = make_float2(fo[0], fo[1]);
      if constexpr (PK) *reinterpret_cast<uint2*>(at_off(ws, p0u << 2)) = make_uint2(wsv[0] | ((u32)kv[0] << 22), wsv[1] | ((u32)kv[1] << 22));
      else *reinterpret_cast<uint2*>(at_off(ws, p0u << 2)) = make_uint2(wsv[0], wsv[1]);
      *reinterpret_cast<u32*>(at_off(bg_new, p0u << 1)) = ((u32)nb[0] & 0xFFFFu) | ((u32)nb[1] << 16);
      if constexpr (!PK) *reinterpret_cast<u32*>(at_off(kc, p0u << 1)) = ((u32)kv[0] & 0xFFFFu) | ((u32)kv[1] << 16);
    };
    // Full rounds (every lane has a pair) run pipelined and without any predicate around a memory operation, so that
    // the wait before a step's arithmetic counts exactly: the four loads of the next step and the four stores of the
    // previous one may still be in flight.  The first round is peeled -- the loop is then entered in the state the
    // back edge leaves (loads requested, then stores) and the compiler's wait count is not the conservative merge of
    // two different histories -- and the loop body holds two rounds.  The ragged last round (P / 2 is not a multiple of the workgroup) runs on its own.
    const int nfull = npair / NT;
    const Coord c0 = coord_of(tid);
    if (nfull > 0) {
      // two register sets take turns (no copies: a copy of a loaded register would wait for the load)
      Coord cA = c0, cB = nfull > 1 ? next(c0) : c0;
      Pair A = request(cA);
      Pair B = request(cB);
      step(cA, A);
      int k = 1;
#pragma unroll 1
      for (; k + 1 < nfull; k += 2) {
        cA = next(cB);
        A = request(cA);
        step(cB, B);
        cB = (k + 2 < nfull) ? next(cA) : cA;
        B = request(cB);
        step(cA, A);
      }
      if (k < nfull) step(cB, B);
    }
    if (tid + nfull * NT < npair) {
      const Coord ct = coord_of(tid + nfull * NT);
      const Pair last = request(ct);
      step(ct, last);
    }
  }
  r.fmin = wave_min(r.fmin);
  r.fmax = wave_max(r.fmax);
  r.sumbg = wave_sum(r.sumbg);
  r.changed = wave_max(r.changed);
  r.sumabs = (u64)wave_sum(sabs);
  if (lane == 0) s_red[wave] = r;
  __syncthreads();
  if (wave == 0) {
    // combine the per-wave partials once; the result stays in LDS (s_R) for the later phases
    Red1 q;
    if (lane < NWAVE) {
      q = s_red[lane];
    } else {
      q.sumpix = sumpix; q.minpix = minpix; q.maxpix = maxpix; q.fmin = 0x7FFFFFFF; q.fmax = -0x7FFFFFFF - 1;
      q.sumbg = 0; q.changed = 0; q.sumabs = 0;
    }
    q.fmin = wave_min(q.fmin);
    q.fmax = wave_max(q.fmax);
    q.sumbg = wave_sum(q.sumbg);
    q.changed = wave_max(q.changed);
    q.sumabs = (u64)wave_sum((u32)q.sumabs);  // <= 20480 pixels x 65535 < 2^31
    if (lane == 0) {
      q.sumpix = sumpix;
      q.minpix = minpix;
      q.maxpix = maxpix;
      q.changed = q.changed != 0;
      *s_R = q;
      {  // phase 3 divides by span = max - min of the shifted, clipped frame: multiplier and shift of that division, once per frame
        int lo = q.fmin - avg_change, hi = q.fmax - avg_change;
        lo = lo < 0 ? 0 : lo;
        hi = hi < 0 ? 0 : hi;
        const u32 d = (u32)(hi - lo);
        const int l = d > 1u ? 32 - __clz((int)(d - 1u)) : 0;   // ceil(log2 d)
        // floor(2^(24 + l) / d) + 1 < 2^25 + 1.  In float64: the quotient is below 2^25 and, unless it is an integer (d a power of
        // two: exact), at least 1 / d >= 2^-17 away from one -- the division's error (< 2^-28) cannot carry the floor
        s_mag[0] = d ? (u32)floor(__longlong_as_double((long long)(1023 + 24 + l) << 52) / (double)d) + 1u : 0u;
        s_mag[1] = (u32)l;
      }
    }
  }
  __syncthreads();
  // min / max of x = max(filtered - avg_change, 0) follow from those of filtered (monotone): no reduction of their own
  mn = uni(s_R->fmin) - avg_change;
  mx = uni(s_R->fmax) - avg_change;
  mn = mn < 0 ? 0 : mn;
  mx = mx < 0 ? 0 : mx;

  // ---- phase 3: normalise to 0..255 (float32, imageprocessing.py:151-169) -> uint8 in LDS
  // np.uint8(float32(255 * (x - mn)) / float32(mx - mn)).  With span = mx - mn <= 65793 the product n = 255 (x - mn) < 2^24 is exact
  // in float32, and the correctly rounded quotient truncates to floor(n / span): n / span <= 255 lies at least 1 / span > 2^-17
  // below the next integer unless it is one, and 2^-17 is half an ulp of [128, 256) -- rounding to nearest cannot reach it.  So
  // the phase is an integer division by a per-frame constant: floor(n m / 2^(24 + l)) with m = floor(2^(24 + l) / span) + 1,
  // l = ceil(log2 span), exact for every n < 2^24 (Granlund & Montgomery 1994, theorem 4.2: 2^(24+l) <= m span <= 2^(24+l) + 2^l);
  // as instructions: mad_u32_u24 ((x - mn) 255 2^8 < 2^32), mul_hi_u32, shift -- 3 instead of the 16 of an IEEE division.
  // Wider spans (pixels and background 65 k apart) take the float32 expression itself.
  {
    const float fmn = (float)mn, fmx = (float)mx;
    const float span = fmx - fmn;
    if (mx == mn) {
      thresh = (float)a.background_thresh;  // raw threshold (cliptracker.py:118-119)
    } else {
      thresh = __fmul_rn(__fdiv_rn((float)a.background_thresh, span), 255.0f);
    }
    if (mx == mn) {
      const u32 fill = (mx == 0) ? 0u : 0x01010101u;  // zeros, or data / max == 1
#pragma unroll 1
      for (int c = tid; c < nchunk; c += NT) *reinterpret_cast<u32*>(s_u8 + (c << 2)) = fill;
#ifndef CPX_NORM_FLOAT   // (experiment switch: -DCPX_NORM_FLOAT = the float32 expression for every span)
    } else if (mx - mn <= 65793) {
#else
    } else if (false) {
#endif
      const u32 mg = (u32)uni((int)s_mag[0]);
      const int lsh = uni((int)s_mag[1]);
      const u32 cadd = 0u - (u32)mn * 65280u;
#pragma unroll 1
      for (int c = tid; c < nchunk; c += NT) {
        const uint2 ql = *reinterpret_cast<const uint2*>(s_tmp + (c << 2));
        const u32 qh = *reinterpret_cast<const u32*>(s_u8 + (c << 2));
        const u32 xv[4] = {(ql.x & 0xFFFFu) | ((qh & 0xFFu) << 16), (ql.x >> 16) | (((qh >> 8) & 0xFFu) << 16),
                           (ql.y & 0xFFFFu) | (((qh >> 16) & 0xFFu) << 16), (ql.y >> 16) | ((qh >> 24) << 16)};
        u32 o = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) o |= (__umulhi(__umul24(xv[j], 65280u) + cadd, mg) >> lsh) << (8 * j);
        *reinterpret_cast<u32*>(s_u8 + (c << 2)) = o;
      }
    } else {
#pragma unroll 1
      for (int c = tid; c < nchunk; c += NT) {
        const uint2 ql = *reinterpret_cast<const uint2*>(s_tmp + (c << 2));
        const u32 qh = *reinterpret_cast<const u32*>(s_u8 + (c << 2));
        const int xv[4] = {(int)((ql.x & 0xFFFFu) | ((qh & 0xFFu) << 16)), (int)((ql.x >> 16) | (((qh >> 8) & 0xFFu) << 16)),
                           (int)((ql.y & 0xFFFFu) | (((qh >> 16) & 0xFFu) << 16)), (int)((ql.y >> 16) | ((qh >> 24) << 16))};
        unsigned char o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (unsigned char)(int)__fdiv_rn(__fmul_rn(255.0f, (float)xv[j] - fmn), span);  // np.uint8() truncation
        *reinterpret_cast<uchar4*>(s_u8 + (c << 2)) = make_uchar4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  ithr = (mx == mn) ? (int)floor(a.background_thresh) : (int)floorf(thresh);
  __syncthreads();
  } else {
    // ---- back half of a split step: scalars from the carry, the (denoised) uint8 image from HBM ----
    const FrameCarry fc = a.carry[(size_t)b * 2 + slot];
    cs.bg_average = fc.bg_avg_in;
    avg_change = fc.avg_change;
    mn = fc.mn;
    mx = fc.mx;
    ithr = fc.ithr;
    thresh = fc.thresh;
    if (tid == 0) {
      *s_R = fc.R;
      *s_ncomp_p = 0;
    }
    const uint4* src = reinterpret_cast<const uint4*>(a.u8_state + ((size_t)b * 2 + (slot ^ a.nlm_flip)) * P);
    for (int i = tid; i < (P >> 4); i += NT) reinterpret_cast<uint4*>(s_u8)[i] = src[i];
    for (int i = tid; i < 2 * H * RW; i += NT) s_rowI[i] = 0ull;
    __syncthreads();
  }
  if (mode == 1) {
    // ---- front half: hand the normalised uint8 image and the scalars to the NLM / back kernels ----
    uint4* dst = reinterpret_cast<uint4*>(a.u8_state + ((size_t)b * 2 + slot) * P);
    for (int i = tid; i < (P >> 4); i += NT) dst[i] = reinterpret_cast<const uint4*>(s_u8)[i];
    if (tid == 0) {
      FrameCarry fc;
      fc.R = *s_R;
      fc.avg_change = avg_change;
      fc.mn = mn;
      fc.mx = mx;
      fc.ithr = ithr;
      fc.thresh = thresh;
      fc.median = 0.0f;  // (cpx_median_kernel writes the record's median)
      fc.pad = 0;
      fc.bg_avg_in = cs.bg_average;
      a.carry[(size_t)b * 2 + slot] = fc;
      // motiondetector.py:224-226 (same expression as at the end of the back half): the next front half needs it
      a.bgavg[b] = fc.R.changed ? rint((double)fc.R.sumbg / (double)((W - 2 * e) * (H - 2 * e))) : cs.bg_average;
    }
    return;
  }

  if (tid == 0) {  // (read back by the same thread at the end of the step: no barrier needed for it)
    s_keep[0] = avg_change;
    s_keep[1] = mn;
    s_keep[2] = mx;
    s_keep[3] = __float_as_int(thresh);
  }
#ifndef CPX_BLUR_SCALAR
  // ---- phase 4a: horizontal [1 4 6 4 1], BORDER_REFLECT_101 -----------------------
  // Packed 16-bit arithmetic, two pixels per instruction (sums <= 16 * 255): a group of 8 output pixels reads the 16 bytes
  // around it with three aligned LDS loads (the reflections only exist at the two ends of a row: byte shuffles of the group
  // itself), spreads the 12 pixels it needs into pairs (p[2j], p[2j+1]) = E_j and (p[2j+1], p[2j+2]) = O_j with v_perm /
  // v_alignbit, and out pair j = E_j + E_j+2 + 4 (O_j + O_j+1) + 6 E_j+1: 4 packed instructions per two pixels.  (Per pixel
  // on 32-bit values this phase and the next were ~25 of the kernel's 145 vector instructions per pixel.)
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  const int ngroup = P >> 3;
  const int gpr = W >> 3;  // 8-pixel groups per row
  auto as_us2 = [](u32 v) -> us2 { return __builtin_bit_cast(us2, v); };
  auto as_u32 = [](us2 v) -> u32 { return __builtin_bit_cast(u32, v); };
  // row of group g without an integer division (run-time width): (g + 0.5) / gpr is at least 0.5 / gpr >= 0.02 away from an
  // integer, g < 2560 -- float32 cannot misplace it.  The group's pixels start at 8 g (W = 8 gpr).
  const float inv_gpr = 1.0f / (float)gpr;
  auto row_of = [&](int g) -> int { return (int)(((float)g + 0.5f) * inv_gpr); };
  for (int g = tid; g < ngroup; g += NT) {
    const int x0 = (g - (int)__umul24((u32)row_of(g), (u32)gpr)) << 3;
    const int at = g << 3;
    const uint2 mid = *reinterpret_cast<const uint2*>(s_u8 + at);              // pixels x0 .. x0 + 7
    u32 left = *reinterpret_cast<const u32*>(s_u8 + max(at - 4, 0));           // x0 - 4 .. x0 - 1 (bytes 2, 3 are used)
    u32 right = *reinterpret_cast<const u32*>(s_u8 + at + 8);                  // x0 + 8 .. x0 + 11 (bytes 0, 1 are used)
    // REFLECT_101: p[-1] = p[1], p[-2] = p[2]; p[W] = p[W - 2], p[W + 1] = p[W - 3]
    left = (x0 == 0) ? __builtin_amdgcn_perm(0u, mid.x, 0x01020C0Cu) : left;
    right = (x0 == W - 8) ? __builtin_amdgcn_perm(0u, mid.y, 0x0C0C0102u) : right;
    us2 E[6], O[5];
    E[0] = as_us2(__builtin_amdgcn_perm(0u, left, 0x0C030C02u));
    E[1] = as_us2(__builtin_amdgcn_perm(0u, mid.x, 0x0C010C00u));
    E[2] = as_us2(__builtin_amdgcn_perm(0u, mid.x, 0x0C030C02u));
    E[3] = as_us2(__builtin_amdgcn_perm(0u, mid.y, 0x0C010C00u));
    E[4] = as_us2(__builtin_amdgcn_perm(0u, mid.y, 0x0C030C02u));
    E[5] = as_us2(__builtin_amdgcn_perm(0u, right, 0x0C010C00u));
#pragma unroll
    for (int k = 0; k < 5; ++k) O[k] = as_us2(__builtin_amdgcn_alignbit(as_u32(E[k + 1]), as_u32(E[k]), 16));
    u32 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = as_u32((E[k] + E[k + 2]) + (O[k] + O[k + 1]) * (unsigned short)4 + E[k + 1] * (unsigned short)6);
    *reinterpret_cast<uint4*>(s_tmp + at) = make_uint4(o[0], o[1], o[2], o[3]);
  }
  __syncthreads();
  // ---- phase 4b: vertical pass, (S + 128) >> 8, floor-threshold -> bit rows ---------
  // packed as well (sums <= 16 * 4080 = 65280 fit 16 bits): rows y-2 + y+2, 4 (y-1 + y+1), 6 y.  ((S + 128) >> 8) > ithr is
  // S >= 256 (ithr + 1) - 128 =: T, i.e. the saturating difference S - (T - 1) is non-zero; T <= 0 sets every pixel, T > 65280 none.
  {
    const int ic = ithr < -2 ? -2 : (ithr > 300 ? 300 : ithr);
    const int T = 256 * (ic + 1) - 128;
    const u32 tm1 = (u32)(T <= 0 ? 0 : (T > 65535 ? 65535 : T - 1));
    const us2 tsub = as_us2(tm1 | (tm1 << 16)), one = as_us2(0x00010001u);
    for (int g = tid; g < ngroup; g += NT) {
      const int y = row_of(g), xg = g - (int)__umul24((u32)y, (u32)gpr);
      // rows y-2, y-1, y+1, y+2 reflected at the borders, as distances from this row (no multiplication by the run-time width)
      const int da = (y >= 2) ? -2 * W : (y == 0 ? 2 * W : 0);
      const int db = (y >= 1) ? -W : W;
      const int dc = (y + 1 < H) ? W : -W;
      const int dd = (y + 2 < H) ? 2 * W : (y + 2 == H ? 0 : -2 * W);
      const uint16_t* c0 = s_tmp + (g << 3);
      const uint4 qa = *reinterpret_cast<const uint4*>(c0 + da);
      const uint4 qb = *reinterpret_cast<const uint4*>(c0 + db);
      const uint4 q0 = *reinterpret_cast<const uint4*>(c0);
      const uint4 qc = *reinterpret_cast<const uint4*>(c0 + dc);
      const uint4 qd = *reinterpret_cast<const uint4*>(c0 + dd);
      const u32 ra[4] = {qa.x, qa.y, qa.z, qa.w}, rb[4] = {qb.x, qb.y, qb.z, qb.w}, r0[4] = {q0.x, q0.y, q0.z, q0.w},
                rc[4] = {qc.x, qc.y, qc.z, qc.w}, rd[4] = {qd.x, qd.y, qd.z, qd.w};
      u32 m = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const us2 S = (as_us2(ra[k]) + as_us2(rd[k])) + (as_us2(rb[k]) + as_us2(rc[k])) * (unsigned short)4 + as_us2(r0[k]) * (unsigned short)6;
        // 0 / 1 per half: min(saturating S - (T - 1), 1).  (As vector code the compiler turns this into a compare and a select per
        // half and a byte shuffle: twice the instructions.)
        u32 hit;
        asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, %3" : "=&v"(hit) : "v"(as_u32(S)), "v"(as_u32(tsub)), "v"(as_u32(one)));
        m |= hit << (2 * k);   // bits 2k and 16 + 2k
      }
      u32 bits = (m | (m >> 15)) & 0xFFu;
      bits = (T <= 0) ? 0xFFu : bits;
      reinterpret_cast<unsigned char*>(s_rowI)[(int)__umul24((u32)y, (u32)(RW * 8)) + xg] = (unsigned char)bits;
    }
  }
  __syncthreads();
#else
  // ---- phase 4a: horizontal [1 4 6 4 1], BORDER_REFLECT_101 -----------------------
  const int ngroup = P >> 3;
  const int gpr = W >> 3;  // 8-pixel groups per row
  for (int g = tid; g < ngroup; g += NT) {
    const int y = g / gpr, x0 = (g - y * gpr) << 3;
    const unsigned char* row = s_u8 + y * W;
    int pv[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      int x = x0 - 2 + k;
      x = x < 0 ? -x : (x >= W ? 2 * W - 2 - x : x);
      pv[k] = row[x];
    }
    uint16_t o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (uint16_t)(pv[k] + 4 * pv[k + 1] + 6 * pv[k + 2] + 4 * pv[k + 3] + pv[k + 4]);
    uint4 pk;
    pk.x = o[0] | ((u32)o[1] << 16);
    pk.y = o[2] | ((u32)o[3] << 16);
    pk.z = o[4] | ((u32)o[5] << 16);
    pk.w = o[6] | ((u32)o[7] << 16);
    *reinterpret_cast<uint4*>(s_tmp + y * W + x0) = pk;
  }
  __syncthreads();
  // ---- phase 4b: vertical pass, (S + 128) >> 8, floor-threshold -> bit rows ---------
  for (int g = tid; g < ngroup; g += NT) {
    const int y = g / gpr, x0 = (g - y * gpr) << 3;
    int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      int yy = y - 2 + k;
      yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
      const int wgt = (k == 0 || k == 4) ? 1 : ((k == 2) ? 6 : 4);
      const uint4 q = *reinterpret_cast<const uint4*>(s_tmp + yy * W + x0);
      acc[0] += wgt * (int)(q.x & 0xFFFF);
      acc[1] += wgt * (int)(q.x >> 16);
      acc[2] += wgt * (int)(q.y & 0xFFFF);
      acc[3] += wgt * (int)(q.y >> 16);
      acc[4] += wgt * (int)(q.z & 0xFFFF);
      acc[5] += wgt * (int)(q.z >> 16);
      acc[6] += wgt * (int)(q.w & 0xFFFF);
      acc[7] += wgt * (int)(q.w >> 16);
    }
    u32 bits = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) bits |= (u32)((((acc[k] + 128) >> 8) > ithr) ? 1 : 0) << k;
    reinterpret_cast<unsigned char*>(s_rowI)[y * (RW * 8) + (x0 >> 3)] = (unsigned char)bits;
  }
  __syncthreads();
#endif
  // ---- phase 5: MORPH_CLOSE with the 1x2 element ----------------------
  for (int i = tid; i < H * RW; i += NT) {
    const int y = i / RW;
    const u64 i0 = s_rowI[i];
    const u64 i1 = (y >= 1) ? s_rowI[i - RW] : 0ull;
    const u64 i2 = (y >= 2) ? s_rowI[i - 2 * RW] : 0ull;
    s_rowE[i] = (y == 0) ? i0 : (i1 | (i0 & i2));
  }
  __syncthreads();

  // ---- phase 6: 8-connected labelling on runs --------------------------------------
  // parents live on run-start slots (row * SW + start/2); initialise them
  for (int i = tid; i < H * RW; i += NT) {
    const int y = i / RW, w = i - y * RW;
    const u64 L = s_rowE[i];
    const u64 Lp = (w > 0) ? s_rowE[i - 1] : 0ull;
    u64 st = L & ~((L << 1) | (Lp >> 63));
    while (st) {
      const int x = (w << 6) + __builtin_ctzll(st);
      st &= st - 1;
      const u32 slot = (u32)(y * SW + (x >> 1));
      s_par[slot] = slot;
    }
  }
  __syncthreads();
  for (int i = tid; i < H * RW; i += NT) {
    const int y = i / RW, w = i - y * RW;
    if (y == 0) continue;
    const u64* rowL = s_rowE + y * RW;
    const u64* rowU = rowL - RW;
    const u64 L = rowL[w], U = rowU[w];
    const u64 Lp = (w > 0) ? rowL[w - 1] : 0ull, Ln = (w < RW - 1) ? rowL[w + 1] : 0ull;
    const u64 Up = (w > 0) ? rowU[w - 1] : 0ull, Un = (w < RW - 1) ? rowU[w + 1] : 0ull;
    const u64 Lshl = (L << 1) | (Lp >> 63), Lshr = (L >> 1) | (Ln << 63);
    const u64 Ushl = (U << 1) | (Up >> 63), Ushr = (U >> 1) | (Un << 63);
    const u64 startL = L & ~Lshl, startU = U & ~Ushl;
    u64 m = L & U & (startL | startU);  // north neighbour, first contact of the two runs
    while (m) {
      const int x = (w << 6) + __builtin_ctzll(m);
      m &= m - 1;
      uf_union(s_par, (u32)(y * SW + (run_start(rowL, x) >> 1)), (u32)((y - 1) * SW + (run_start(rowU, x) >> 1)));
    }
    m = startL & ~U & Ushl;  // north-west neighbour
    while (m) {
      const int x = (w << 6) + __builtin_ctzll(m);
      m &= m - 1;
      uf_union(s_par, (u32)(y * SW + (x >> 1)), (u32)((y - 1) * SW + (run_start(rowU, x - 1) >> 1)));
    }
    m = L & ~U & Ushr & ~Lshr;  // north-east neighbour
    while (m) {
      const int x = (w << 6) + __builtin_ctzll(m);
      m &= m - 1;
      uf_union(s_par, (u32)(y * SW + (run_start(rowL, x) >> 1)), (u32)((y - 1) * SW + ((x + 1) >> 1)));
    }
  }
  __syncthreads();
  // flatten + number the roots
  for (int i = tid; i < H * RW; i += NT) {
    const int y = i / RW, w = i - y * RW;
    const u64 L = s_rowE[i];
    const u64 Lp = (w > 0) ? s_rowE[i - 1] : 0ull;
    u64 st = L & ~((L << 1) | (Lp >> 63));
    while (st) {
      const int x = (w << 6) + __builtin_ctzll(st);
      st &= st - 1;
      const u32 slot = (u32)(y * SW + (x >> 1));
      const u32 root = uf_find(s_par, slot);
      if (root == slot) {
        const u32 cid = atomicAdd(s_ncomp_p, 1u);
        s_cid[slot] = (uint16_t)cid;  // (at most one component per 2 x 2 block: < 2^16 at any supported resolution)
      }
    }
  }
  __syncthreads();
  const int ncomp_all = uni((int)*s_ncomp_p);
  // More components than the LDS tables hold (CAP): a handle created with max_components > CAP brings per-clip tables in
  // HBM (a.big_stat: the same eight statistics rows + the rank row, a.cap_out entries each) and the frame takes the same
  // code on those -- slower (global atomics, an O(n^2) ranking over thousands), but a frame of hot pixels or rain is
  // a frame the reference processes too (cliptrackextractor.py:236-247 has no limit).  Without them, or beyond the
  // caller's own capacity, the frame reports CPX_ERR_OVERFLOW and the count it needs.
  const bool big = ncomp_all > CAP && a.big_stat != nullptr && ncomp_all <= a.cap_out;
  const bool overflow = !big && (ncomp_all > CAP || ncomp_all > a.cap_out);
  const int ncomp = overflow ? 0 : ncomp_all;
  const bool has_prev = cs.has_prev != 0;
  double* s_part = reinterpret_cast<double*>(s_rowI);  // [2][NWAVE][2]: the un-closed bit rows are dead by now

  // phases 7 and 8 over statistics tables `stat` [8][SC] and ranks `rnk` [SC]: the LDS ones (SC = CAP, every ordinary
  // frame) or the clip's HBM ones (SC = a.cap_out); a generic lambda, so that each call is compiled for its address space
  auto label_phases = [&](auto* stat, auto* rnk, const int SC, const bool in_hbm) __attribute__((always_inline)) {
  // tables in HBM: what one wave wrote (stores, atomics at L2) must be what another reads after the barrier, and the
  // CU's vector cache may still hold the lines from the clip's previous frame -- write back / invalidate around it
  auto phase_sync = [&]() __attribute__((always_inline)) {
    if (in_hbm) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (in_hbm) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  };
  // ---- phase 7: statistics per component ------------------------------------------------
  // stat rows: 0 area, 1 minx, 2 maxx, 3 miny, 4 maxy, 5 sumx, 6 sumy, 7 key
  for (int i = tid; i < ncomp; i += NT) {
    stat[0 * SC + i] = 0;
    stat[1 * SC + i] = 0xFFFFFFFFu;
    stat[2 * SC + i] = 0;
    stat[3 * SC + i] = 0xFFFFFFFFu;
    stat[4 * SC + i] = 0;
    stat[5 * SC + i] = 0;
    stat[6 * SC + i] = 0;
    stat[7 * SC + i] = 0xFFFFFFFFu;
  }
  phase_sync();
  if (ncomp > 0) {
    for (int i = tid; i < H * RW; i += NT) {
      const int y = i / RW, w = i - y * RW;
      const u64* rowL = s_rowE + y * RW;
      const u64 L = rowL[w];
      const u64 Lp = (w > 0) ? rowL[w - 1] : 0ull;
      u64 st = L & ~((L << 1) | (Lp >> 63));
      while (st) {
        const int xs = (w << 6) + __builtin_ctzll(st);
        st &= st - 1;
        const int xe = run_end(rowL, xs);
        const u32 slot = (u32)(y * SW + (xs >> 1));
        const u32 root = uf_find(s_par, slot);
        const u32 cid = s_cid[root];
        const u32 len = (u32)(xe - xs + 1);
        atomicAdd(&stat[0 * SC + cid], len);
        atomicMin(&stat[1 * SC + cid], (u32)xs);
        atomicMax(&stat[2 * SC + cid], (u32)xe);
        atomicMin(&stat[3 * SC + cid], (u32)y);
        atomicMax(&stat[4 * SC + cid], (u32)y);
        atomicAdd(&stat[5 * SC + cid], (u32)((xs + xe) * (int)len / 2));
        atomicAdd(&stat[6 * SC + cid], (u32)y * len);
        atomicMin(&stat[7 * SC + cid], (u32)((y >> 1) * SW + (xs >> 1)));
      }
    }
  }
  phase_sync();
  // OpenCV numbers components by the raster position of their first 2x2 block (SURVEY a7')
  for (int i = tid; i < ncomp; i += NT) {
    const u32 k = stat[7 * SC + i];
    u32 rank = 0;
    for (int j = 0; j < ncomp; ++j) rank += (stat[7 * SC + j] < k) ? 1u : 0u;
    rnk[i] = rank;
  }
  phase_sync();

  // ---- phase 8a: label image (Frame.mask) ------------------------------------------------------
  if (a.labels_out) {
    int32_t* Lout = a.labels_out + (size_t)fidx * P;
    for (int c = tid; c < nchunk; c += NT) {
      const int p0 = c << 2;
      const int y = p0 / W, x0 = p0 - y * W;
      const u64* rowL = s_rowE + y * RW;
      const u32 nib = (u32)((rowL[x0 >> 6] >> (x0 & 63)) & 0xFull);
      int lab[4] = {0, 0, 0, 0};
      if (nib && !overflow) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (nib & (1u << j)) {
            const int xs = run_start(rowL, x0 + j);
            const u32 root = uf_find(s_par, (u32)(y * SW + (xs >> 1)));
            lab[j] = (int)rnk[s_cid[root]] + 1;
          }
        }
      } else if (nib) {
        lab[0] = (nib & 1) ? -1 : 0; lab[1] = (nib & 2) ? -1 : 0; lab[2] = (nib & 4) ? -1 : 0; lab[3] = (nib & 8) ? -1 : 0;
      }
      *reinterpret_cast<int4*>(at_off(Lout, (unsigned)p0 << 2)) = make_int4(lab[0], lab[1], lab[2], lab[3]);
    }
  }

  // ---- phase 8b: np.var(delta_filtered[bbox]) per component ------------------------------------------
  // delta = |f32(norm255(cur.filtered)) - f32(norm255(prev.filtered))|  (cliptracker.py:249-261);
  // normalize() promotes to float64 for these float64 frames (NumPy >= 2 scalar promotion).
  // Small boxes: one wave each.  Large boxes: the whole workgroup, partials combined in wave order.
  Component* Cout = a.comps_out + (size_t)fidx * a.cap_out;
  const double cmin = (double)s_R->fmin, cmax = (double)s_R->fmax;
  const double pmin = (double)cs.prev_fmin, pmax = (double)cs.prev_fmax;
  auto delta_at = [&](int q) -> double {
    const float cv = *at_off(filt_cur, (unsigned)q << 2), pv = *at_off(filt_prev, (unsigned)q << 2);
    float an, bn;
    if (cmax == cmin) an = (cmax == 0.0) ? 0.0f : (float)((double)cv / cmax);
    else an = (float)((255.0 * ((double)cv - cmin)) / (cmax - cmin));
    if (pmax == pmin) bn = (pmax == 0.0) ? 0.0f : (float)((double)pv / pmax);
    else bn = (float)((255.0 * ((double)pv - pmin)) / (pmax - pmin));
    return (double)fabsf(an - bn);
  };
  auto emit = [&](int cidx, int bx, int by, int bw, int bh, float var) {
    Component o;
    o.x = bx;
    o.y = by;
    o.width = bw;
    o.height = bh;
    o.area = (int)stat[0 * SC + cidx];
    o.sum_x = (int)stat[5 * SC + cidx];
    o.sum_y = (int)stat[6 * SC + cidx];
    o.pixel_variance = var;
    Cout[rnk[cidx]] = o;
  };
#ifndef CPX_VAR_PER_WAVE
  // The components of a frame share the workgroup's waves: with nc components in a round (at most NWAVE), NWAVE / nc waves sum
  // each one's box, partials meet in LDS, the first lane of a group combines them in wave order.  (One wave per component left
  // eleven of twelve waves waiting on the usual one or two boxes, seven trips to L2 long: the phase was 9 % of the step.)
  // Box pixel k -> (row, column) by a float32 reciprocal and one correction step (k < 20480: the estimate is off by at most one);
  // normalised values by float32 division when it provably equals float32(float64 division): integer operands below 2^24,
  // i.e. spans <= 65793 (the quotient of such integers is 2^-41 relative away from any float32 rounding boundary it is not on,
  // float64's own rounding moves it by 2^-53).
  const int icmin = s_R->fmin, icmax = s_R->fmax, ipmin = cs.prev_fmin, ipmax = cs.prev_fmax;
  const bool narrow = (icmax - icmin <= 65793) && (ipmax - ipmin <= 65793) && icmax < 65536 && icmin > -65536 &&
                      ipmax < 65536 && ipmin > -65536;
  auto delta32_at = [&](int q) -> double {
    const float cv = *at_off(filt_cur, (unsigned)q << 2), pv = *at_off(filt_prev, (unsigned)q << 2);
    float an, bn;
    if (icmax == icmin) an = (icmax == 0) ? 0.0f : __fdiv_rn(cv, (float)icmax);
    else an = __fdiv_rn(__fmul_rn(255.0f, __fsub_rn(cv, (float)icmin)), (float)(icmax - icmin));
    if (ipmax == ipmin) bn = (ipmax == 0) ? 0.0f : __fdiv_rn(pv, (float)ipmax);
    else bn = __fdiv_rn(__fmul_rn(255.0f, __fsub_rn(pv, (float)ipmin)), (float)(ipmax - ipmin));
    return (double)fabsf(an - bn);
  };
  if (!has_prev) {
    for (int cidx = tid; cidx < ncomp; cidx += NT) {
      const int bx = (int)stat[1 * SC + cidx], by = (int)stat[3 * SC + cidx];
      emit(cidx, bx, by, (int)stat[2 * SC + cidx] - bx + 1, (int)stat[4 * SC + cidx] - by + 1, 0.0f);
    }
  } else {
    int par = 0;
    for (int base = 0; base < ncomp; base += NWAVE) {
      const int nc = min(NWAVE, ncomp - base);
      const int G = NWAVE / nc;                       // waves per component (uniform)
      const int ci = (int)((u32)wave / (u32)G), wi = wave - ci * G;   // (per wave: uniform within it)
      const bool mine = ci < nc;
      const int cidx = base + (mine ? ci : 0);
      const int bx = (int)stat[1 * SC + cidx], by = (int)stat[3 * SC + cidx];
      const int bw = (int)stat[2 * SC + cidx] - bx + 1, bh = (int)stat[4 * SC + cidx] - by + 1;
      const int n = bw * bh;
      if (mine) {
        const float inv_bw = 1.0f / (float)bw;
        double s1 = 0.0, s2 = 0.0;
        for (int k = wi * 64 + lane; k < n; k += 64 * G) {
          int yy = (int)((float)k * inv_bw);
          int xx = k - (int)__umul24((u32)yy, (u32)bw);
          if (xx < 0) {
            --yy;
            xx += bw;
          } else if (xx >= bw) {
            ++yy;
            xx -= bw;
          }
          const int q = (by + yy) * W + bx + xx;
          const double d = narrow ? delta32_at(q) : delta_at(q);
          s1 += d;
          s2 += d * d;
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
          s_part[(par * NWAVE + wave) * 2] = s1;
          s_part[(par * NWAVE + wave) * 2 + 1] = s2;
        }
      }
      phase_sync();
      if (mine && wi == 0 && lane == 0) {
        double t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < G; ++w) {
          t1 += s_part[(par * NWAVE + wave + w) * 2];
          t2 += s_part[(par * NWAVE + wave + w) * 2 + 1];
        }
        const double mean = t1 / (double)n;
        const double var = t2 / (double)n - mean * mean;
        emit(cidx, bx, by, bw, bh, (float)(var < 0.0 ? 0.0 : var));
      }
      par ^= 1;
    }
  }
#else
  constexpr int BIG = 512;  // pixels: above this a box is summed by the whole workgroup
  int par = 0;
  for (int cidx = 0; cidx < ncomp; ++cidx) {
    const int bx = (int)stat[1 * SC + cidx], by = (int)stat[3 * SC + cidx];
    const int bw = (int)stat[2 * SC + cidx] - bx + 1, bh = (int)stat[4 * SC + cidx] - by + 1;
    const int n = bw * bh;
    if (n > BIG && has_prev) {
      double s1 = 0.0, s2 = 0.0;
      for (int k = tid; k < n; k += NT) {
        const int yy = k / bw;
        const double d = delta_at((by + yy) * W + bx + (k - yy * bw));
        s1 += d;
        s2 += d * d;
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0) {
        s_part[(par * NWAVE + wave) * 2] = s1;
        s_part[(par * NWAVE + wave) * 2 + 1] = s2;
      }
      phase_sync();
      if (tid == 0) {
        double t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < NWAVE; ++w) {
          t1 += s_part[(par * NWAVE + w) * 2];
          t2 += s_part[(par * NWAVE + w) * 2 + 1];
        }
        const double mean = t1 / (double)n;
        double var = t2 / (double)n - mean * mean;
        emit(cidx, bx, by, bw, bh, (float)(var < 0.0 ? 0.0 : var));
      }
      par ^= 1;
    } else if ((cidx % NWAVE) == wave) {
      float var = 0.0f;
      if (has_prev) {
        double s1 = 0.0, s2 = 0.0;
        for (int k = lane; k < n; k += 64) {
          const int yy = k / bw;
          const double d = delta_at((by + yy) * W + bx + (k - yy * bw));
          s1 += d;
          s2 += d * d;
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const double mean = s1 / (double)n;
        const double v = s2 / (double)n - mean * mean;
        var = (float)(v < 0.0 ? 0.0 : v);
      }
      if (lane == 0) emit(cidx, bx, by, bw, bh, var);
    }
  }

#endif
  };
  if (big) label_phases(a.big_stat + (size_t)b * 9 * a.cap_out, a.big_stat + (size_t)b * 9 * a.cap_out + (size_t)8 * a.cap_out, a.cap_out, true);
  else label_phases(s_stat, s_rank, CAP, false);

  // ---- per-frame record + clip state -------------------------------------------------------------------
  // (written two phases ago; every thread keeps the new clip state for the clip's next frame.  Read through a volatile
  // pointer: thread 0 wrote the record itself and the compiler otherwise forwards the stored values -- i.e. carries seven
  // registers across all the labelling phases, which at 64 registers per lane means through scratch)
  Red1 R;
  {
    volatile Red1* vr = s_R;
    R.sumpix = vr->sumpix; R.minpix = vr->minpix; R.maxpix = vr->maxpix; R.fmin = vr->fmin; R.fmax = vr->fmax;
    R.sumbg = vr->sumbg; R.changed = vr->changed; R.sumabs = vr->sumabs;
  }
  ClipState ns;
  // motiondetector.py:224-226: average = int(round(np.average(background))) when any pixel changed
  ns.bg_average = R.changed ? rint((double)R.sumbg / (double)((W - 2 * e) * (H - 2 * e))) : cs.bg_average;
  ns.prev_fmin = R.fmin;
  ns.prev_fmax = R.fmax;
  ns.has_prev = 1;
  ns.n_done = t + 1;
  if (tid == 0) {
    FrameInfo fi;
    fi.frame_number = t;
    fi.n_components = overflow ? ncomp_all : ncomp;
    fi.status = overflow ? -5 : 0;
    fi.ffc_affected = a.proc_ffc[pbase + t];
    // (parked in LDS before the labelling phases; volatile: the values must not be carried in registers instead)
    volatile int* keep = s_keep;
    fi.avg_change = keep[0];
    fi.norm_min = keep[1];
    fi.norm_max = keep[2];
    fi.threshold = __int_as_float(keep[3]);
    fi.filt_min = R.fmin;
    fi.filt_max = R.fmax;
    fi.thermal_min = (int)R.minpix;
    fi.thermal_max = (int)R.maxpix;
    fi.thermal_sum = R.sumpix;
    fi.thermal_median = 0.0f;  // (not stored: the field is cpx_median_kernel's)
    fi.filtered_abs_sum = R.sumabs;
    fi.background_average = ns.bg_average;
    fi.background_changed = (int)R.changed;
    fi.reserved = 0;
    {  // the record without its thermal_median word (offset 52): cpx_median_kernel writes that one (no load of it here)
      static_assert(sizeof(cpx_frame_info) == 80 && offsetof(cpx_frame_info, thermal_median) == 52, "record layout");
      uint2 q[10];
      __builtin_memcpy(q, &fi, sizeof(fi));
      uint2* o = reinterpret_cast<uint2*>(&a.info_out[fidx]);   // (the record's alignment is 8)
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        if (i != 6) o[i] = q[i];
      }
      reinterpret_cast<u32*>(o)[12] = q[6].x;
    }
    a.cstate[b] = ns;
  }
  cs = uniform_state(ns);
}
}  // namespace

// One workgroup per clip walks the clip's processed frames [t0, t1) in order.  Frames of a clip depend on each other
// (background, window sum, previous filtered frame), clips do not -- so nothing orders workgroups against each
// other, and after a few frames the streaming pass of one workgroup runs beside the labelling phases of its
// neighbour on the CU instead of every workgroup of a per-frame launch moving through the same phase at once.
// a.order (optional) lists the clips by falling length: the dispatcher hands out workgroups in index order.
// The split forms (mode 1 / 2, around the NLM kernel or on two streams) are launched one step at a time.
template <bool PK>
__global__ __launch_bounds__(NT, CPX_TRACK_MIN_WAVES_PER_SIMD) void cpx_frame_kernel(TrackArgs a, int t0, int t1, int mode) {
  const int b = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
  const int pbase = a.proc_off[b];
  const int nproc = a.proc_off[b + 1] - pbase;
  const int tend = t1 < nproc ? t1 : nproc;
  if (t0 >= tend) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  ClipState cs = uniform_state(a.cstate[b]);
  {  // weight table -> LDS, once per clip (entries beyond the table's length are never indexed: k <= frames so far)
    double* wl = reinterpret_cast<double*>(smem + wtab_lds_offset(a.W, a.H));
    u32* tl = reinterpret_cast<u32*>(wl + WTAB_LDS);
    for (int i = threadIdx.x; i < WTAB_LDS; i += NT) wl[i] = a.wtab[min(i, a.wtab_len - 1)];
    for (int i = threadIdx.x; i < WTHR_LDS; i += NT) tl[i] = a.wthr[min(i, a.wtab_len - 1)];
    __syncthreads();
  }
  // The frame step reads its arguments from the kernel-argument segment (TrackArgs is the first parameter) through a
  // pointer the optimiser cannot see through: otherwise every pointer and constant of TrackArgs is hoisted out of the
  // frame loop, kept live across it, and the scalar registers spill into vector registers and those into scratch
  // (208 bytes per lane = +20 % HBM traffic of an HBM-bound pass).  A scalar load per use costs nothing.
  KernArgs* ap = (KernArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  for (int t = t0; t < tend; ++t) {
    asm volatile("" : "+s"(ap));
    frame_step<PK>(*ap, b, pbase, t, mode, cs, smem);
    if (t + 1 < tend) {
      // the next frame reads what other threads of this workgroup wrote (clamped background edges, the previous
      // filtered frame under a bounding box) and reuses the LDS image.  Workgroup scope is all it takes: the waves
      // of a workgroup share the CU's write-through L1.  (An agent-scope fence here writes back and invalidates the
      // XCD's whole L2 once per frame and workgroup: measured 4x slower.)
      __syncthreads();
    }
  }
}



// ---------------------------------------------------------------------------------------------
// np.median(thermal) of every processed frame (ClipStats.add_frame, track/clip.py:475).  It feeds the clip statistics
// only, so it does not have to sit on the frame-after-frame dependency chain of a clip: one 256-thread workgroup per
// frame, all frames of the batch at once.  Exact selection by bisection on the value range with the frame in
// registers; a step counts `value <= mid` with one compare + s_bcnt1 per register half (the wave's count is a scalar,
// no cross-lane reduction); median = mean of the two middle order statistics.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int NT_MED = 256;
constexpr int NW_MED = NT_MED / 64;
constexpr int MCH = (4 * NCH * NT / 4 + NT_MED - 1) / NT_MED;  // 4-pixel chunks per thread at the largest frame
#ifndef CPX_MED_SCALAR_PART
#define CPX_MED_SCALAR_PART 40   // per cent of a thread's registers counted through the scalar unit (see the bisection loop)
#endif
constexpr int MED_SCALAR = MCH * CPX_MED_SCALAR_PART / 100;
}  // namespace
__global__ __launch_bounds__(NT_MED) void cpx_median_kernel(TrackArgs a, int t0, int nsteps) {
  const int b = (int)(blockIdx.x / (unsigned)nsteps), t = t0 + (int)(blockIdx.x - (unsigned)b * (unsigned)nsteps);
  const int pbase = a.proc_off[b];
  if (t >= a.proc_off[b + 1] - pbase) return;
  const int fidx = a.proc_idx[pbase + t];
  const int P = a.W * a.H, nchunk = P >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint16_t* F = a.frames + (size_t)fidx * P;
  __shared__ u32 s_cnt[3 * NW_MED];
  __shared__ u32 s_mm[3 * NW_MED];
  u32 pk[MCH][2];
  // all loads first, branch-free (clamped address, padding selected afterwards): a conditional load makes the compiler
  // wait for each one before the next is issued -- twenty serialised trips to HBM
#pragma unroll
  for (int i = 0; i < MCH; ++i) {
    const int c = tid + i * NT_MED;
    const uint2 q = *reinterpret_cast<const uint2*>(at_off(F, (unsigned)min(c, nchunk - 1) << 3));
    pk[i][0] = q.x;
    pk[i][1] = q.y;
  }
  u32 minpix = 0xFFFFFFFFu, maxpix = 0, sumpix = 0;
#pragma unroll
  for (int i = 0; i < MCH; ++i) {
    const bool in = tid + i * NT_MED < nchunk;
    const u32 v0 = pk[i][0] & 0xFFFFu, v1 = pk[i][0] >> 16, v2 = pk[i][1] & 0xFFFFu, v3 = pk[i][1] >> 16;
    minpix = in ? min(min(minpix, v0), min(v1, min(v2, v3))) : minpix;
    maxpix = in ? max(max(maxpix, v0), max(v1, max(v2, v3))) : maxpix;
    sumpix += in ? (v0 + v1) + (v2 + v3) : 0u;
    pk[i][0] = in ? pk[i][0] : 0xFFFFFFFFu;  // padding: 65535, never below a bisection candidate
    pk[i][1] = in ? pk[i][1] : 0xFFFFFFFFu;
  }
  minpix = wave_min(minpix);
  maxpix = wave_max(maxpix);
  sumpix = wave_sum(sumpix);
  if (lane == 0) {
    s_mm[wave] = minpix;
    s_mm[NW_MED + wave] = maxpix;
    s_mm[2 * NW_MED + wave] = sumpix;
  }
  __syncthreads();
  u32 lo = 0xFFFFFFFFu, hi = 0, tsum = 0;
#pragma unroll
  for (int w = 0; w < NW_MED; ++w) {
    lo = min(lo, s_mm[w]);
    hi = max(hi, s_mm[NW_MED + w]);
    tsum += s_mm[2 * NW_MED + w];
  }
  lo = (u32)uni((int)lo);
  hi = (u32)uni((int)hi);
  const u32 meanv = (u32)uni((int)(tsum / (u32)P));
  const u32 k1 = (u32)((P - 1) >> 1), k2 = (u32)(P >> 1);
  int par = 0;
  // Selection of the value of rank k1 in [lo, hi] by counting `value <= mid`, one workgroup-wide count per round.  A thermal frame
  // is mostly background: its median sits within a few counts of its MEAN, so the first probe is the mean and the bracket is
  // found by galloping away from it (steps of 2, 8, 32, ...) before it is bisected -- 4-6 rounds where bisecting [min, max]
  // takes log2(max - min) = 10-14.  Every probe keeps the invariant (rank k1 lies in [lo, hi]); the order of the probes does not
  // change what is found.
#ifdef CPX_MED_PLAIN_BISECT   // (experiment switch: bisect [min, max] as rounds 2-5 did)
  int phase = 3;
#else
  int phase = 0;   // 0: probe the mean; 1 / 2: galloping down from hi / up from lo; 3: bisection
#endif
  u32 gstep = 2;
  while (lo < hi) {  // uniform: every thread sees the same block totals
    if (phase != 0 && hi - lo <= gstep) phase = 3;   // the bracket is narrower than the next step: bisect it (the step stops growing:
                                                     // grown through sixteen more rounds it would wrap to zero and probe lo - 1 for ever)
    u32 mid = (lo + hi) >> 1;
    if (phase == 0) mid = min(max(meanv, lo), hi - 1u);
    else if (phase == 1) mid = hi - gstep;
    else if (phase == 2) mid = lo + gstep - 1u;
    u32 cnt = 0;
    // The compare + s_bcnt1 form costs two scalar instructions per value and the CU has ONE scalar unit for its four SIMDs: alone
    // it ran that unit at ~80 % (profiles/r06_track_experiments.md).  So the last MCH - MED_SCALAR registers count on the vector
    // side instead, packed: saturating v - mid is non-zero exactly for v > mid, min(., 1) makes it a flag, a packed add keeps a
    // count per lane and half (3 vector instructions per two values); <= is the complement (the padding counts as > mid).
#pragma unroll
    for (int i = 0; i < MED_SCALAR; ++i) {
      cnt += wave_count_le_halves(pk[i][0], mid) + wave_count_le_halves(pk[i][1], mid);
    }
    {
      const u32 midpk = mid | (mid << 16);
      u32 gt = 0;
#pragma unroll
      for (int i = MED_SCALAR; i < MCH; ++i) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          u32 f;
          asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, %3" : "=&v"(f) : "v"(pk[i][hh]), "v"(midpk), "v"(0x00010001u));
          gt = pk_add_u16(gt, f);
        }
      }
      const u32 gtw = wave_sum((gt & 0xFFFFu) + (gt >> 16));
      cnt += (u32)(64 * 4 * (MCH - MED_SCALAR)) - gtw;
    }
    if (lane == 0) s_cnt[par * NW_MED + wave] = cnt;
    __syncthreads();
    u32 tot = 0;
#pragma unroll
    for (int w = 0; w < NW_MED; ++w) tot += s_cnt[par * NW_MED + w];
    tot = (u32)uni((int)tot);
    par ^= 1;
    const bool le = tot >= k1 + 1;   // the value of rank k1 is <= mid
    if (le) hi = mid;
    else lo = mid + 1;
    if (phase == 0) phase = le ? 1 : 2;
    else if (phase == 1) {
      if (le) gstep <<= 2;
      else phase = 3;
    } else if (phase == 2) {
      if (!le) gstep <<= 2;
      else phase = 3;
    }
  }
  // lo = value of rank k1; rank k2 is the same value unless exactly k1+1 pixels are <= lo
  u32 cnt = 0, nxt = 0xFFFFFFFFu;
#pragma unroll
  for (int i = 0; i < MCH; ++i) {
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
      const u32 val = (hh & 1) ? (pk[i][hh >> 1] >> 16) : (pk[i][hh >> 1] & 0xFFFFu);
      nxt = (val > lo && val < nxt) ? val : nxt;
    }
    cnt += wave_count_le_halves(pk[i][0], lo) + wave_count_le_halves(pk[i][1], lo);
  }
  nxt = wave_min(nxt);
  if (lane == 0) {
    s_cnt[par * NW_MED + wave] = cnt;
    s_cnt[2 * NW_MED + wave] = nxt;
  }
  __syncthreads();
  u32 tot = 0, mnx = 0xFFFFFFFFu;
#pragma unroll
  for (int w = 0; w < NW_MED; ++w) {
    tot += s_cnt[par * NW_MED + w];
    mnx = min(mnx, s_cnt[2 * NW_MED + w]);
  }
  // (padding lanes hold 65535: they only count when lo == 65535, where both ranks are 65535 anyway)
  const u32 v2 = (tot >= k2 + 1 || lo == 65535u) ? lo : mnx;
  if (tid == 0) a.info_out[fidx].thermal_median = 0.5f * (float)(lo + v2);
}

// ---------------------------------------------------------------------------------------------
// cv2.fastNlMeansDenoising(uint8, None) with the defaults h = 3, template 7, search 21
// (track/cliptracker.py:116-117; integer algorithm of SURVEY.md Appendix A.6): for every pixel and
// every offset in [-10,10]^2 the 7x7 sum of squared differences, weight = LUT[dist >> 6] (fixed
// point, 48 non-zero entries), result = (sum w*v + sum w / 2) / sum w.  One workgroup per frame;
// per offset: row-wise 7-sums of squared differences into LDS (pass A), column-wise sliding 7-sums +
// weight accumulation in registers (pass B).  In place on the uint8 image between the front and back
// halves of the frame step.
// ---------------------------------------------------------------------------------------------
namespace {
// double-buffered row sums (one barrier per offset pair, the row pass of pair q + 1 beside the column pass of pair q):
// 4.53 vs 4.56 us per frame in round 3, 3.80 vs 4.04 after the round-4 instruction trims (-DCPX_NLM_SINGLE: one buffer)
#if !defined(CPX_NLM_SINGLE) && !defined(CPX_NLM_DB)
#define CPX_NLM_DB 1
#endif
#ifndef CPX_NLM_ASM   // experiment switch: 0 = compiler-scheduled LDS reads, 1 = row sums by hand, 2 = the row loop too
#define CPX_NLM_ASM 1
#endif
constexpr bool NLM_ASM_HV = (CPX_NLM_ASM & 1) != 0, NLM_ASM_ROWS = (CPX_NLM_ASM & 2) != 0;
constexpr int NLM_R = 13;      // border = template radius 3 + search radius 10
constexpr int NT_NLM = 1024;  // threads of an NLM workgroup (independent of the frame kernel's)
__device__ __forceinline__ int refl101(int v, int n) { return v < 0 ? -v : (v >= n ? 2 * n - 2 - v : v); }
}  // namespace

// Offsets come in pairs: dist(p, p + d) = dist(p + d, p), i.e. the row sums of squared differences of offset -d are
// those of offset d shifted by d.  Pass A computes them once per pair over the band extended by d (rows -3 - dy ... ,
// columns -dx ... or ... W - dx), pass B accumulates both directions from the same array: 220 pairs + the zero offset
// instead of 441 passes.  Row sums are kept as uint16 saturated at 4095: a 7-row sum of them is exact below 4096 and
// >= 4095 otherwise, and every distance >= 3072 has weight zero (LUT index >= 48), so the weights are those of the
// exact sums.  Pass A works on packed 16-bit pairs (difference, square, saturating adds), pass B on two adjacent
// columns per thread with packed sliding sums and a zero-padded weight table (no branch per pixel).
//
// BH = output rows per thread in pass B (two adjacent columns each).  A workgroup owns a band of
// RB = BH * (NT_NLM / (W / 2)) image rows of one frame (blockIdx.y = band): BH = 10 covers a whole 120-row frame with
// one workgroup -- the throughput configuration for large batches; smaller BH spread a frame over several CUs, which
// is what the latency of a single clip (one frame per launch) needs.  Reads slot t & 1 of the hand-over image, writes
// the other slot (the bands of a frame overlap in what they read, so the result cannot go back in place).
namespace {
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
constexpr int NLM_LUT2 = 49;    // LDS weight table: pairs of the 48 non-zero weights + the zero weight
__device__ __forceinline__ u16x2 as_pk(u32 v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ u32 as_u32(u16x2 v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ int nlm_hs(int W) { return (W + 24 + 7) & ~7; }  // row stride of the row-sum array (uint16): W + 13 columns in segments of 12

// ---- LDS reads with the offsets in the instruction (cpx_nlm_kernel<.., 160>): the compiler pairs neighbouring rows into
// ds_read2_b32, whose 8-bit offsets reach three rows, and spends a v_add per pair on the base; written out, sixteen
// rows hang off ONE address register.  The loads of a block are all in flight before the single wait that follows
// (lds_wait ties the loaded registers to the wait, so no consumer can be scheduled ahead of it).
template <int STRIDE, int BASE>
__device__ __forceinline__ void lds_read8_b32(u32 addr, u32* v) {
  asm volatile("ds_read_b32 %0, %8 offset:%9\n\tds_read_b32 %1, %8 offset:%10\n\tds_read_b32 %2, %8 offset:%11\n\t"
               "ds_read_b32 %3, %8 offset:%12\n\tds_read_b32 %4, %8 offset:%13\n\tds_read_b32 %5, %8 offset:%14\n\t"
               "ds_read_b32 %6, %8 offset:%15\n\tds_read_b32 %7, %8 offset:%16"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
               : "v"(addr), "n"(BASE), "n"(BASE + STRIDE), "n"(BASE + 2 * STRIDE), "n"(BASE + 3 * STRIDE), "n"(BASE + 4 * STRIDE),
                 "n"(BASE + 5 * STRIDE), "n"(BASE + 6 * STRIDE), "n"(BASE + 7 * STRIDE));
}
template <int STRIDE, int BASE>
__device__ __forceinline__ void lds_read5_u16(u32 addr, u32* v) {
  asm volatile("ds_read_u16 %0, %5 offset:%6\n\tds_read_u16 %1, %5 offset:%7\n\tds_read_u16 %2, %5 offset:%8\n\t"
               "ds_read_u16 %3, %5 offset:%9\n\tds_read_u16 %4, %5 offset:%10"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
               : "v"(addr), "n"(BASE), "n"(BASE + STRIDE), "n"(BASE + 2 * STRIDE), "n"(BASE + 3 * STRIDE), "n"(BASE + 4 * STRIDE));
}
__device__ __forceinline__ void lds_gather5_b32(const u32* addr, u32* v) {
  asm volatile("ds_read_b32 %0, %5\n\tds_read_b32 %1, %6\n\tds_read_b32 %2, %7\n\tds_read_b32 %3, %8\n\tds_read_b32 %4, %9"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
               : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]));
}
__device__ __forceinline__ void lds_wait8(u32* v) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
__device__ __forceinline__ void lds_wait5(u32* v) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]));
}
// the LDS byte address of a pointer into the workgroup's shared memory
__device__ __forceinline__ u32 lds_addr(const void* p) { return (u32)reinterpret_cast<size_t>(p); }
}  // namespace

// WC = the frame width when it is known at compile time (160: row strides fold into instruction offsets and the
// pass-B row loads pair up as ds_read2_b32), 0 = any supported width.
#ifndef CPX_NLM_WPE5   // experiment switch: waves per SIMD the BH <= 5 forms are compiled for (8 = two workgroups per CU)
#define CPX_NLM_WPE5 4
#endif
template <int BH, int WC>
__global__ __launch_bounds__(NT_NLM) __attribute__((amdgpu_waves_per_eu(BH <= 5 ? CPX_NLM_WPE5 : 4, BH <= 5 ? CPX_NLM_WPE5 : 4)))
void cpx_nlm_kernel(TrackArgs a, int t) {
  const int b = blockIdx.x;
  const int nproc = a.proc_off[b + 1] - a.proc_off[b];
  if (t >= nproc) return;
  const int W = WC > 0 ? WC : a.W, H = a.H, P = W * H;
  const int tid = threadIdx.x;
  const int npair = W >> 1;                // pass-B column pairs
  const int nsub = NT_NLM / npair;         // pass-B sub-bands
  const int RB = BH * nsub;                // rows of this workgroup's band
  const int yb0 = blockIdx.y * RB;         // first image row of the band
  if (yb0 >= H) return;
  const int RBc = (H - yb0 < RB) ? (H - yb0) : RB;
  const int EW = W + 2 * NLM_R, EHb = RBc + 2 * NLM_R;
  const int ES = (EW + 8 + 7) & ~7;  // row stride of the padded image: multiple of 8, slack for the aligned fetches of a
                                     // shifted row (the zero padding is read, never used)
  const int HS = nlm_hs(W);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ext = smem;                                                   // [EHb][ES]: padded rows yb0-13 ...
  const int RBa = RB < H ? RB : H;   // rows the launch sized the LDS for
  uint16_t* Hh = reinterpret_cast<uint16_t*>(smem + (((size_t)(RBa + 2 * NLM_R) * ES + 32 + 15) & ~(size_t)15));  // [RBa+16+BH][HS]
  // weights of a PAIR of distances in one 4-byte entry (two uint16): [min(d0 >> 6, 48)][min(d1 >> 6, 48)] (entry 48 = weight 0)
#ifdef CPX_NLM_DB
  // double-buffered row sums: the row pass of pair q + 1 and the column pass of pair q run between the same two
  // barriers (one barrier per pair, and waves drift between an LDS-heavy and a VALU-heavy phase)
  uint16_t* const Hh1 = Hh + (size_t)(RBa + 16 + BH) * HS;
  u32* s_lut2 = reinterpret_cast<u32*>(Hh1 + (size_t)(RBa + 16 + BH) * HS);  // [49][49]
#else
  u32* s_lut2 = reinterpret_cast<u32*>(Hh + (size_t)(RBa + 16 + BH) * HS);  // [49][49]
#endif
  const unsigned char* img = a.u8_state + ((size_t)b * 2 + (t & 1)) * P;
  unsigned char* out = a.u8_state + ((size_t)b * 2 + ((t + 1) & 1)) * P;

  for (int i = tid; i < EHb * ES; i += NT_NLM) {
    const int ey = i / ES, ex = i - ey * ES;
    unsigned char v = 0;
    if (ex < EW) v = img[refl101(yb0 + ey - NLM_R, H) * W + refl101(ex - NLM_R, W)];
    ext[i] = v;
  }
  for (int i = tid; i < NLM_LUT2 * NLM_LUT2; i += NT_NLM) {
    const int i0 = i / NLM_LUT2, i1 = i - i0 * NLM_LUT2;
    s_lut2[i] = (u32)(i0 < 48 ? a.nlm_lut[i0] : 0) | ((u32)(i1 < 48 ? a.nlm_lut[i1] : 0) << 16);
  }
  // pass-B role: two adjacent image columns and a sub-band of BH rows
  const int pcol = tid % npair, sub = tid / npair;
  const int bx = pcol << 1;
  const bool active_b = sub < nsub && sub * BH < RBc;
  const int l0 = sub * BH;  // first band-local row of this thread
  int est[2 * BH], wsum[2 * BH];
  const u32 lut_off = (u32)(reinterpret_cast<const unsigned char*>(s_lut2) - smem);  // byte offset of the table in the workgroup's LDS
  __syncthreads();
  {  // the zero offset: distance 0 everywhere
    const int w0 = (int)(s_lut2[0] & 0xFFFFu);
#pragma unroll
    for (int i = 0; i < BH; ++i) {
      const int l = (l0 + i < RBc) ? l0 + i : 0;
      const unsigned char* px = ext + (l + NLM_R) * ES + bx + NLM_R;
      est[2 * i] = w0 * (int)px[0];
      est[2 * i + 1] = w0 * (int)px[1];
      wsum[2 * i] = wsum[2 * i + 1] = w0;
    }
  }

  // offset pair q: upper half of the search window: (0, 1..10), then (1..10, -10..10)
  //
  // Instruction census (round 4): every vector instruction of this kernel issues at the same
  // rate on gfx950 (scratch/valu_cost_probe.hip: v_perm, v_pk_*, v_dot2_u32_u16, v_mul_lo_u32, SDWA forms alike), so
  // the count is the cost.  What the two passes were trimmed by:
  //   pass A  the byte differences come straight out of the packed rows by SDWA byte operands (two v_sub_u16_sdwa per
  //           16-bit pair instead of two unpacking v_perm + one v_pk_sub), the a-row starts on a dword (the domain is
  //           extended to the left by 0..3 columns instead: four alignment v_perm and one LDS read fewer per item)
  //   pass B  both directions of a pair run together; the table index is ONE v_dot2_u32_u16 (i0 * 196 + i1 * 4 + base)
  //           instead of and / mul / shift / add; weights and pixels of the two directions are packed per PIXEL by one
  //           v_perm each, and a pixel's two estimate terms (and its two weights) are accumulated by one
  //           v_dot2_u32_u16 each instead of two multiply-adds (two adds)
  auto pair_offset = [](const int q, int& dy, int& dx) {
    if (q < 10) {
      dy = 0;
      dx = q + 1;
    } else {
      dy = (q - 10) / 21 + 1;
      dx = (q - 10) - (dy - 1) * 21 - 10;
    }
  };
  auto pass_a = [&](const int q, uint16_t* const Hh) __attribute__((always_inline)) {
    int dy, dx;
    pair_offset(q, dy, dx);
    const int adx = dx > 0 ? dx : -dx;
    const int c00 = dx > 0 ? -dx : 0;
    const int ash = (NLM_R - 3 + c00) & 3;     // columns the domain is extended to the left by: the a values start on a dword
    const int c0 = c00 - ash;                  // first column of the extended domain
    const int nrows = RBc + 6 + dy;            // rows -3 - dy .. RBc + 2 of the band
    const int nseg = (W + adx + ash + 11) / 12; // 12-column segments
    // ---- pass A: Hh[rr][c - c0] = min(4095, sum_{v=-3..3} (ext(r, c+v) - ext(r+dy, c+dx+v))^2), r = yb0 - 3 - dy + rr ----
    // An item is a segment of 12 columns (18 values of each row: five dwords of a, six of b): the per-item index
    // arithmetic is paid once per 12 columns and a band of 120 rows is at most two items per thread.
    {
      const int sa0 = NLM_R - 3 + c0, sb0 = sa0 + dx;   // byte of a row where the 18 values of segment 0 start (>= 0; sa0 = 0 mod 4)
      const int bbase = sb0 & ~3;                       // 4-byte aligned start of the b values; the shift is wave-uniform
      const u32 bsel = 0x03020100u + 0x01010101u * (u32)(sb0 & 3);
      // (row, segment) of this thread's first item, without a division
      const u32 mg = (1u << 20) / (u32)nseg + 1u;        // floor(i / nseg) = (i * mg) >> 20 for i < 1024, nseg < 1024
      int rr = (int)((__umul24((u32)tid, mg)) >> 20);
      int sg = tid - rr * nseg;
      const int drr = NT_NLM / nseg;   // rows one sweep of the block covers
      // a thread keeps its SEGMENT and walks rows in a constant stride of drr rows (the threads past drr * nseg idle): the
      // three per-item addresses advance by wave-uniform constants instead of being rebuilt from (row, segment) per item
      // (measured 3.667 -> 3.623 us per frame, profiles/r06_nlm_experiments.md)
      const bool has_seg = rr < drr;
      const int so = __umul24((u32)sg, 12u);
      const u32* pa = reinterpret_cast<const u32*>(ext + __umul24((u32)(rr + NLM_R - 3 - dy), (u32)ES) + sa0 + so);
      const u32* pb = reinterpret_cast<const u32*>(ext + __umul24((u32)(rr + NLM_R - 3), (u32)ES) + bbase + so);
      uint2* hp = reinterpret_cast<uint2*>(Hh + __umul24((u32)rr, (u32)HS) + so);
      const int stepE = (drr * ES) >> 2, stepH = (drr * HS) >> 2;   // (uniform; in dwords / uint2: ES and HS are multiples of 8)
      for (; has_seg && rr < nrows; rr += drr, pa += stepE, pb += stepE, hp += stepH) {
        u32 ra[5], qb[6];
#pragma unroll
        for (int k = 0; k < 5; ++k) ra[k] = pa[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) qb[k] = pb[k];
        u32 rb[5];  // the 18 (20) b values, byte-aligned with the a values
#pragma unroll
        for (int k = 0; k < 5; ++k) rb[k] = __builtin_amdgcn_perm(qb[k + 1], qb[k], bsel);
        // D[i] = (d^2 of value 2i, d^2 of value 2i+1), E[i] = (d^2 of 2i+1, d^2 of 2i+2); the differences wrap in 16
        // bits and their squares are exact there (<= 255^2)
        u16x2 D[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          u32 d;
          if (i & 1)
            asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2\n\t"
                "v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:BYTE_3"
                : "=&v"(d) : "v"(ra[i >> 1]), "v"(rb[i >> 1]));
          else
            asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n\t"
                "v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1"
                : "=&v"(d) : "v"(ra[i >> 1]), "v"(rb[i >> 1]));
          const u16x2 dd = as_pk(d);
          D[i] = dd * dd;
        }
        // output pair (2m, 2m+1) = sum of the seven pairs P[2m .. 2m+6], P[2i] = D[i], P[2i+1] = E[i]
        //                        = G[m] + G[m+1] + G[m+2] + D[m+3] with G[i] = D[i] + E[i]; saturating adds
        auto sadd = [](u16x2 x, u16x2 y) { return __builtin_elementwise_add_sat(x, y); };
        u16x2 G[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) G[i] = sadd(D[i], as_pk(__builtin_amdgcn_alignbit(as_u32(D[i + 1]), as_u32(D[i]), 16)));
        const u16x2 cap = {4095, 4095};
        u32 o[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) o[m] = as_u32(__builtin_elementwise_min(sadd(sadd(G[m], sadd(G[m + 1], G[m + 2])), D[m + 3]), cap));
        hp[0] = make_uint2(o[0], o[1]);
        hp[1] = make_uint2(o[2], o[3]);
        hp[2] = make_uint2(o[4], o[5]);
      }
    }
  };
  // row sums of one direction: the seven-row window of every output row of this thread, as packed column pairs
  auto load_hv = [&](const uint16_t* const Hh, const int rbase, const int coff, u32* hv) __attribute__((always_inline)) {
    const int cc = bx + coff;
    // (rows past the band's last one feed only outputs that are never stored; the array has BH spare rows for them)
    const uint16_t* hr0 = Hh + (cc & ~1) + __umul24((u32)rbase, (u32)HS);
    if constexpr (WC == 160 && BH == 10 && NLM_ASM_HV) {  // sixteen rows off one address register (row stride 2 HS = 368 bytes: HS = W + 24 uint16 entries)
      constexpr int RS = 2 * ((160 + 24 + 7) & ~7);
      const u32 ad = lds_addr(hr0);
      if (coff & 1) {  // the pair straddles two aligned words (bx is even: the parity is the offset's)
        u32 lo[16], hi[16];
        lds_read8_b32<RS, 0>(ad, lo);
        lds_read8_b32<RS, 4>(ad, hi);
        lds_read8_b32<RS, 8 * RS>(ad, lo + 8);
        lds_read8_b32<RS, 8 * RS + 4>(ad, hi + 8);
        lds_wait8(lo); lds_wait8(hi); lds_wait8(lo + 8); lds_wait8(hi + 8);
#pragma unroll
        for (int k = 0; k < 16; ++k) hv[k] = __builtin_amdgcn_alignbit(hi[k], lo[k], 16);
      } else {
        lds_read8_b32<RS, 0>(ad, hv);
        lds_read8_b32<RS, 8 * RS>(ad, hv + 8);
        lds_wait8(hv); lds_wait8(hv + 8);
      }
      return;
    }
    if (coff & 1) {  // the pair straddles two aligned words (bx is even: the parity is the offset's)
#pragma unroll
      for (int k = 0; k < BH + 6; ++k) {
        const uint16_t* hr = hr0 + (size_t)k * HS;
        const u32 w0 = *reinterpret_cast<const u32*>(hr), w1 = *reinterpret_cast<const u32*>(hr + 2);
        hv[k] = __builtin_amdgcn_alignbit(w1, w0, 16);
      }
    } else {
#pragma unroll
      for (int k = 0; k < BH + 6; ++k) hv[k] = *reinterpret_cast<const u32*>(hr0 + (size_t)k * HS);
    }
  };
  auto pass_b = [&](const int q, const uint16_t* const Hh) __attribute__((always_inline)) {
    int dy, dx;
    pair_offset(q, dy, dx);
    // ---- pass B: dist = sum of seven row sums; weight; accumulate -- offset +d and offset -d together ----
    if (active_b) {
      const int ash = (NLM_R - 3 + (dx > 0 ? -dx : 0)) & 3;   // pass A's left extension
      // offset +d reads rows l + k + dy at column x - c0, offset -d rows l + k at column x - dx - c0
      const int coff0 = (dx > 0 ? dx : 0) + ash, coff1 = (dx > 0 ? 0 : -dx) + ash;   // uniform
      u32 hv0[BH + 6], hv1[BH + 6];
      load_hv(Hh, l0 + dy, coff0, hv0);
      load_hv(Hh, l0, coff1, hv1);
      u16x2 V0 = as_pk(hv0[0]), V1 = as_pk(hv1[0]);
#pragma unroll
      for (int k = 1; k < 7; ++k) {
        V0 = V0 + as_pk(hv0[k]);
        V1 = V1 + as_pk(hv1[k]);
      }
      // the two pixels of a row as ALIGNED 16-bit reads (a merged ds_read_u16 at an odd column is a misaligned LDS
      // access: measured 3.3 of 8.3 us per frame in round 2).  Rows past the band's end (last band of a frame only)
      // read whatever follows in LDS and are never stored: no test per row.  +d and -d shift the column by the same
      // parity, so one flag serves both directions.
      const unsigned char* px0 = ext + __umul24((u32)(l0 + NLM_R + dy), (u32)ES) + ((bx + NLM_R + dx) & ~1);
      const unsigned char* px1 = ext + __umul24((u32)(l0 + NLM_R - dy), (u32)ES) + ((bx + NLM_R - dx) & ~1);
      const bool podd = ((NLM_R + dx) & 1) != 0;
      const u16x2 lut_k = {NLM_LUT2 * 4, 4};   // bytes per step of the first / second table index
      const u16x2 ones = {1, 1};
      auto rows = [&](auto podd_c) __attribute__((always_inline)) {
        constexpr bool PODD = decltype(podd_c)::value;
        if constexpr (WC == 160 && BH == 10 && NLM_ASM_ROWS) {
          // three phases instead of ten dependent chains: all table addresses (the sliding sums are cheap), then all
          // LDS reads in flight at once, then the accumulation
          constexpr int ESC = (160 + 2 * NLM_R + 8 + 7) & ~7;   // row stride of the padded image (bytes)
          const u32 lut_ad = lds_addr(s_lut2);
          u32 T0[BH], T1[BH];
#pragma unroll
          for (int i = 0; i < BH; ++i) {
            const u16x2 cap48 = {48, 48};
            T0[i] = __builtin_amdgcn_udot2(__builtin_elementwise_min(V0 >> 6, cap48), lut_k, lut_ad, false);
            T1[i] = __builtin_amdgcn_udot2(__builtin_elementwise_min(V1 >> 6, cap48), lut_k, lut_ad, false);
            if (i + 1 < BH) {
              V0 = V0 + as_pk(hv0[i + 7]) - as_pk(hv0[i]);
              V1 = V1 + as_pk(hv1[i + 7]) - as_pk(hv1[i]);
            }
          }
          u32 W0[BH], W1[BH], A0[BH], B0[BH], A1[BH], B1[BH];
          const u32 pa0 = lds_addr(px0), pa1 = lds_addr(px1);
          lds_gather5_b32(T0, W0); lds_gather5_b32(T0 + 5, W0 + 5);
          lds_gather5_b32(T1, W1); lds_gather5_b32(T1 + 5, W1 + 5);
          lds_read5_u16<ESC, 0>(pa0, A0); lds_read5_u16<ESC, 5 * ESC>(pa0, A0 + 5);
          lds_read5_u16<ESC, 0>(pa1, B0); lds_read5_u16<ESC, 5 * ESC>(pa1, B0 + 5);
          if (PODD) {
            lds_read5_u16<ESC, 2>(pa0, A1); lds_read5_u16<ESC, 5 * ESC + 2>(pa0, A1 + 5);
            lds_read5_u16<ESC, 2>(pa1, B1); lds_read5_u16<ESC, 5 * ESC + 2>(pa1, B1 + 5);
            lds_wait5(A1); lds_wait5(A1 + 5); lds_wait5(B1); lds_wait5(B1 + 5);
          }
          lds_wait5(W0); lds_wait5(W0 + 5); lds_wait5(W1); lds_wait5(W1 + 5);
          lds_wait5(A0); lds_wait5(A0 + 5); lds_wait5(B0); lds_wait5(B0 + 5);
#pragma unroll
          for (int i = 0; i < BH; ++i) {
            const u32 WA = __builtin_amdgcn_perm(W1[i], W0[i], 0x05040100u), WB = __builtin_amdgcn_perm(W1[i], W0[i], 0x07060302u);
            u32 PA, PB;
            if (PODD) {
              PA = __builtin_amdgcn_perm(B0[i], A0[i], 0x0c050c01u);
              PB = __builtin_amdgcn_perm(B1[i], A1[i], 0x0c040c00u);
            } else {
              PA = __builtin_amdgcn_perm(B0[i], A0[i], 0x0c040c00u);
              PB = __builtin_amdgcn_perm(B0[i], A0[i], 0x0c050c01u);
            }
            est[2 * i] = (int)__builtin_amdgcn_udot2(as_pk(WA), as_pk(PA), (u32)est[2 * i], false);
            est[2 * i + 1] = (int)__builtin_amdgcn_udot2(as_pk(WB), as_pk(PB), (u32)est[2 * i + 1], false);
            wsum[2 * i] = (int)__builtin_amdgcn_udot2(as_pk(WA), ones, (u32)wsum[2 * i], false);
            wsum[2 * i + 1] = (int)__builtin_amdgcn_udot2(as_pk(WB), ones, (u32)wsum[2 * i + 1], false);
          }
          return;
        }
#pragma unroll
        for (int i = 0; i < BH; ++i) {
          const u16x2 cap48 = {48, 48};
          const u16x2 i0 = __builtin_elementwise_min(V0 >> 6, cap48), i1 = __builtin_elementwise_min(V1 >> 6, cap48);
          // entry = (weight of the left pixel, weight of the right pixel) as two uint16 (weights < 2^15)
          const u32 W0 = *reinterpret_cast<const u32*>(smem + __builtin_amdgcn_udot2(i0, lut_k, lut_off, false));
          const u32 W1 = *reinterpret_cast<const u32*>(smem + __builtin_amdgcn_udot2(i1, lut_k, lut_off, false));
          // per pixel: (weight towards +d, weight towards -d) and (pixel at +d, pixel at -d)
          const u32 WA = __builtin_amdgcn_perm(W1, W0, 0x05040100u), WB = __builtin_amdgcn_perm(W1, W0, 0x07060302u);
          u32 PA, PB;
          if (PODD) {  // left pixel = high byte of the first aligned pair, right pixel = low byte of the next
            const u32 a0 = *reinterpret_cast<const uint16_t*>(px0 + i * ES), a1 = *reinterpret_cast<const uint16_t*>(px0 + i * ES + 2);
            const u32 b0 = *reinterpret_cast<const uint16_t*>(px1 + i * ES), b1 = *reinterpret_cast<const uint16_t*>(px1 + i * ES + 2);
            PA = __builtin_amdgcn_perm(b0, a0, 0x0c050c01u);
            PB = __builtin_amdgcn_perm(b1, a1, 0x0c040c00u);
          } else {
            const u32 a0 = *reinterpret_cast<const uint16_t*>(px0 + i * ES), b0 = *reinterpret_cast<const uint16_t*>(px1 + i * ES);
            PA = __builtin_amdgcn_perm(b0, a0, 0x0c040c00u);
            PB = __builtin_amdgcn_perm(b0, a0, 0x0c050c01u);
          }
          est[2 * i] = (int)__builtin_amdgcn_udot2(as_pk(WA), as_pk(PA), (u32)est[2 * i], false);
          est[2 * i + 1] = (int)__builtin_amdgcn_udot2(as_pk(WB), as_pk(PB), (u32)est[2 * i + 1], false);
          wsum[2 * i] = (int)__builtin_amdgcn_udot2(as_pk(WA), ones, (u32)wsum[2 * i], false);
          wsum[2 * i + 1] = (int)__builtin_amdgcn_udot2(as_pk(WB), ones, (u32)wsum[2 * i + 1], false);
          if (i + 1 < BH) {
            V0 = V0 + as_pk(hv0[i + 7]) - as_pk(hv0[i]);
            V1 = V1 + as_pk(hv1[i + 7]) - as_pk(hv1[i]);
          }
        }
      };
      if (podd) rows(std::true_type{});
      else rows(std::false_type{});
    }
  };
#ifdef CPX_NLM_DB
  pass_a(0, Hh);
  __syncthreads();
#ifdef CPX_NLM_STAGGER   // experiment: half of the waves take the two passes of an interval in the other order
  const bool b_first = ((tid >> 6) & CPX_NLM_STAGGER) != 0;
  for (int q = 0; q < 220; ++q) {
    if (b_first) {
      pass_b(q, (q & 1) ? Hh1 : Hh);
      if (q + 1 < 220) pass_a(q + 1, ((q + 1) & 1) ? Hh1 : Hh);
    } else {
      if (q + 1 < 220) pass_a(q + 1, ((q + 1) & 1) ? Hh1 : Hh);
      pass_b(q, (q & 1) ? Hh1 : Hh);
    }
    __syncthreads();
  }
#else
  for (int q = 0; q < 220; ++q) {
    if (q + 1 < 220) pass_a(q + 1, ((q + 1) & 1) ? Hh1 : Hh);
    pass_b(q, (q & 1) ? Hh1 : Hh);
    __syncthreads();
  }
#endif
#else
  for (int q = 0; q < 220; ++q) {
    pass_a(q, Hh);
    __syncthreads();
    pass_b(q, Hh);
    __syncthreads();
  }
#endif
  if (active_b) {
#pragma unroll
    for (int i = 0; i < BH; ++i) {
      const int l = l0 + i;
      if (l < RBc) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const u32 ws = (u32)wsum[2 * i + j];
          u32 v = ((u32)est[2 * i + j] + ws / 2u) / ws;  // the zero offset always contributes LUT[0] > 0
          out[(yb0 + l) * W + bx + j] = (unsigned char)(v > 255u ? 255u : v);
        }
      }
    }
  }
}

namespace {
size_t nlm_lds_rows(int W, int rows, int bh) {
  const size_t ES = ((size_t)W + 2 * NLM_R + 8 + 7) & ~(size_t)7;
  const size_t HS = ((size_t)W + 24 + 7) & ~(size_t)7;
#ifdef CPX_NLM_DB
  const size_t nbuf = 2;
#else
  const size_t nbuf = 1;
#endif
  return ((((size_t)rows + 2 * NLM_R) * ES + 32 + 15) & ~(size_t)15) + nbuf * ((size_t)rows + 16 + bh) * HS * 2 + NLM_LUT2 * NLM_LUT2 * 4 + 64;
}
template <int BH, int WC>
void launch_nlm_tw(const TrackArgs& a, int B, int t, hipStream_t s) {
  static bool lds_ready[64];
  (void)cpx_dyn_lds_ready(reinterpret_cast<const void*>(cpx_nlm_kernel<BH, WC>), lds_ready, 160 * 1024 - 2048);
  const int RB = BH * (NT_NLM / (a.W / 2));
  hipLaunchKernelGGL((cpx_nlm_kernel<BH, WC>), dim3(B, (a.H + RB - 1) / RB), dim3(NT_NLM),
                     nlm_lds_rows(a.W, RB < a.H ? RB : a.H, BH), s, a, t);
}
template <int BH>
void launch_nlm_t(const TrackArgs& a, int B, int t, hipStream_t s) {
  if (a.W == 160) launch_nlm_tw<BH, 160>(a, B, t, s);
  else launch_nlm_tw<BH, 0>(a, B, t, s);
}
}  // namespace

size_t nlm_lds_bytes(int W, int H) {
  const int rb = 10 * (NT_NLM / (W / 2));
  return nlm_lds_rows(W, rb < H ? rb : H, 10);
}
int nlm_supported(int W, int H) {
  const int nsub = NT_NLM / (W / 2);
  // any height works (bands), the width must leave at least one pass-B sub-band per workgroup
  return nsub >= 1 && (W % 8) == 0 && W >= 8 && H >= 1 && nlm_lds_bytes(W, H) <= 160 * 1024 - 2048;
}
void launch_nlm(const TrackArgs& a, int B, int t, hipStream_t s) {
  // enough workgroups to fill the chip: whole frames per workgroup for big batches, bands of a frame for small
  // ones (a single clip is one frame per launch)
  const int nsub = NT_NLM / (a.W / 2);
#ifdef CPX_NLM_BANDS   // experiment switch: bands per frame for big batches (scratch/build_variant.sh)
  const int want = (B >= 384) ? CPX_NLM_BANDS : (512 + B - 1) / B;
#else
  const int want = (B >= 384) ? 1 : (512 + B - 1) / B;  // bands per frame that would give ~2 workgroups per CU
#endif
  const int rows = (a.H + want - 1) / want;              // rows per band for that
  if (rows > 5 * nsub) launch_nlm_t<10>(a, B, t, s);
  else if (rows > 2 * nsub) launch_nlm_t<5>(a, B, t, s);
  else if (rows > nsub) launch_nlm_t<2>(a, B, t, s);
  else launch_nlm_t<1>(a, B, t, s);
}

// final background of each clip as float, edges replicated (motiondetector.py:239-244)
__global__ __launch_bounds__(256) void cpx_export_background_kernel(TrackArgs a, float* out) {
  const int b = blockIdx.x;
  const int W = a.W, H = a.H, P = W * H, e = a.edge;
  const uint16_t* bg = a.bg + ((size_t)b * 2 + (a.cstate[b].n_done & 1)) * P;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    int y = p / W, x = p - y * W;
    out[(size_t)b * P + p] = (float)bg[clampi(y, e, H - 1 - e) * W + clampi(x, e, W - 1 - e)];
  }
}

size_t track_lds_bytes(int W, int H) { return wtab_lds_offset(W, H) + (size_t)WTAB_LDS * sizeof(double) + (size_t)WTHR_LDS * sizeof(u32); }

void launch_init(const TrackArgs& a, int B, int keep, hipStream_t s) {
  hipLaunchKernelGGL(cpx_init_kernel, dim3(B), dim3(256), 0, s, a, keep);
}
void launch_frame(const TrackArgs& a, int B, int t0, int t1, int mode, hipStream_t s) {
  if (a.packed_state) hipLaunchKernelGGL(cpx_frame_kernel<true>, dim3(B), dim3(NT), track_lds_bytes(a.W, a.H), s, a, t0, t1, mode);
  else hipLaunchKernelGGL(cpx_frame_kernel<false>, dim3(B), dim3(NT), track_lds_bytes(a.W, a.H), s, a, t0, t1, mode);
}
// the state a packed call left (count in the window sum's top ten bits) -> the two arrays every other path reads
__global__ __launch_bounds__(256) void cpx_unpack_state_kernel(uint32_t* wsum, uint16_t* kcnt, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t v = wsum[i];
    kcnt[i] = (uint16_t)(v >> 22);
    wsum[i] = v & 0x3FFFFFu;
  }
}
void launch_unpack_state(uint32_t* wsum, uint16_t* kcnt, size_t n, hipStream_t s) {
  if (n == 0) return;
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 65535u * 16u);
  hipLaunchKernelGGL(cpx_unpack_state_kernel, dim3(blocks), dim3(256), 0, s, wsum, kcnt, n);
}
void launch_median(const TrackArgs& a, int B, int t0, int t1, hipStream_t s) {
  if (t1 <= t0) return;
  hipLaunchKernelGGL(cpx_median_kernel, dim3((unsigned)(t1 - t0) * (unsigned)B), dim3(NT_MED), 0, s, a, t0, t1 - t0);
}
void launch_export_background(const TrackArgs& a, int B, float* out, hipStream_t s) {
  hipLaunchKernelGGL(cpx_export_background_kernel, dim3(B), dim3(256), 0, s, a, out);
}
int track_max_pixels() { return 4 * NCH * NT; }
int track_lds_components() { return CAP; }
int frame_kernel_attr_setup() {
  const int r0 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(cpx_frame_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  const int r1 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(cpx_frame_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  return r0 != 0 ? r0 : r1;
}

}  // namespace cpx
