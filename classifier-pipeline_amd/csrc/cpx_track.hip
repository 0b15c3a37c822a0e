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

struct Red1 {
  u32 sumpix;
  u32 minpix, maxpix;
  int fmin, fmax;
  u32 sumbg;
  u32 changed;
  u64 sumabs;
};

}  // namespace

// ---------------------------------------------------------------------------
// init: WeightedBackground.__init__ + first process_frame
// (motiondetector.py:178-211; cliptrackextractor.py:129-139)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cpx_init_kernel(TrackArgs a) {
  const int b = blockIdx.x;
  const int W = a.W, H = a.H, P = W * H, e = a.edge;
  const uint16_t* F = a.frames + (size_t)a.clip_first[b] * P;
  int32_t* bg0 = a.bg + (size_t)b * 2 * P;
  u32* ws = a.wsum + (size_t)b * P;
  uint16_t* kc = a.kcnt + (size_t)b * P;
  u64 s = 0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    int y = p / W, x = p - y * W;
    int cy = clampi(y, e, H - 1 - e), cx = clampi(x, e, W - 1 - e);
    int v = F[cy * W + cx];
    bg0[p] = v;
    bg0[P + p] = v;
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
    st.pad = 0;
    a.cstate[b] = st;
  }
}

// ---------------------------------------------------------------------------
// one processed frame of every clip
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT, CPX_TRACK_MIN_WAVES_PER_SIMD) void cpx_frame_kernel(TrackArgs a, int t) {
  const int b = blockIdx.x;
  const int pbase = a.proc_off[b];
  const int nproc = a.proc_off[b + 1] - pbase;
  if (t >= nproc) return;

  const int W = a.W, H = a.H, P = W * H, e = a.edge;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int SW = W >> 1;  // run-start slots per row
  const int nchunk = P >> 2;

  // ---- LDS ---------------------------------------------------------------
  // everything is carved from one dynamic region so that its base stays 16-byte aligned
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
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

  const int fidx = a.proc_idx[pbase + t];
  const int oidx = (t >= a.window) ? a.proc_idx[pbase + t - a.window] : -1;
  const int nwin = (t + 1 < a.window) ? (t + 1) : a.window;
  const uint16_t* F = a.frames + (size_t)fidx * P;
  const uint16_t* O = (oidx >= 0) ? a.frames + (size_t)oidx * P : nullptr;
  const int32_t* bg_old = a.bg + ((size_t)b * 2 + (t & 1)) * P;
  int32_t* bg_new = a.bg + ((size_t)b * 2 + ((t + 1) & 1)) * P;
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
  const ClipState cs = a.cstate[b];

  // ---- phase 1: streaming pass ------------------------------------------------
  int v[NCH][4];
  Red1 r;
  r.sumpix = 0;
  r.minpix = 0xFFFFFFFFu;
  r.maxpix = 0;
  r.fmin = 0x7FFFFFFF;
  r.fmax = -0x7FFFFFFF - 1;
  r.sumbg = 0;
  r.changed = 0;
  r.sumabs = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NT;
    if (c < nchunk) {
      const int p0 = c << 2;
      const int y = p0 / W, x0 = p0 - y * W;
      const ushort4 px = *reinterpret_cast<const ushort4*>(F + p0);
      const int pix[4] = {px.x, px.y, px.z, px.w};
      int bgv[4];
      const int cy = clampi(y, e, H - 1 - e);
      const bool row_in = (cy == y);
      if (row_in && x0 >= e && x0 + 3 <= W - 1 - e) {
        const int4 q = *reinterpret_cast<const int4*>(bg_old + p0);
        bgv[0] = q.x; bgv[1] = q.y; bgv[2] = q.z; bgv[3] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bgv[j] = bg_old[cy * W + clampi(x0 + j, e, W - 1 - e)];
      }
      int oldp[4] = {0, 0, 0, 0};
      if (O) {
        const ushort4 q = *reinterpret_cast<const ushort4*>(O + p0);
        oldp[0] = q.x; oldp[1] = q.y; oldp[2] = q.z; oldp[3] = q.w;
      }
      uint4 wq = *reinterpret_cast<const uint4*>(ws + p0);
      u32 wsv[4] = {wq.x, wq.y, wq.z, wq.w};
      ushort4 kq = *reinterpret_cast<const ushort4*>(kc + p0);
      int kv[4] = {kq.x, kq.y, kq.z, kq.w};
      int nb[4];
      float fo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d = pix[j] - bgv[j];  // filtered = float32(pix) - background (cliptrackextractor.py:212)
        v[i][j] = d;
        fo[j] = (float)d;
        r.sumpix += (u32)pix[j];
        r.minpix = min(r.minpix, (u32)pix[j]);
        r.maxpix = max(r.maxpix, (u32)pix[j]);
        r.fmin = min(r.fmin, d);
        r.fmax = max(r.fmax, d);
        r.sumabs += (u64)(d < 0 ? -d : d);
        // background feed: np.int32(np.mean(last <=45 frames)) == window_sum // n (cliptrackextractor.py:173-176)
        wsv[j] = wsv[j] + (u32)pix[j] - (u32)oldp[j];
        const int f = (int)(wsv[j] / (u32)nwin);
        const int x = x0 + j;
        nb[j] = bgv[j];
        if (row_in && x >= e && x <= W - 1 - e) {
          // motiondetector.py:212-223: bg' = bg if bg < f - w else f ; w' = w + add if (same) else 0
          const double wgt = a.wtab[kv[j]];
          const bool keep = (double)bgv[j] < (double)f - wgt;
          const int nv = keep ? bgv[j] : f;
          kv[j] = keep ? kv[j] + 1 : 0;
          r.changed |= (u32)(nv != bgv[j]);
          r.sumbg += (u32)nv;
          nb[j] = nv;
        }
      }
      *reinterpret_cast<float4*>(filt_cur + p0) = make_float4(fo[0], fo[1], fo[2], fo[3]);
      *reinterpret_cast<uint4*>(ws + p0) = make_uint4(wsv[0], wsv[1], wsv[2], wsv[3]);
      *reinterpret_cast<int4*>(bg_new + p0) = make_int4(nb[0], nb[1], nb[2], nb[3]);
      *reinterpret_cast<ushort4*>(kc + p0) =
          make_ushort4((unsigned short)kv[0], (unsigned short)kv[1], (unsigned short)kv[2], (unsigned short)kv[3]);
      // keep the five chunk bodies from being interleaved: 16 waves per CU already cover the
      // HBM latency, and interleaving them blows the 128-VGPR budget of a 1024-thread workgroup
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[i][j] = 0;
    }
  }
  // block reduction of the phase-1 scalars
  r.sumpix = wave_sum(r.sumpix);
  r.minpix = wave_min(r.minpix);
  r.maxpix = wave_max(r.maxpix);
  r.fmin = wave_min(r.fmin);
  r.fmax = wave_max(r.fmax);
  r.sumbg = wave_sum(r.sumbg);
  r.changed = wave_max(r.changed);
  r.sumabs = wave_sum(r.sumabs);
  if (lane == 0) s_red[wave] = r;
  // zero the bit rows while we are at a barrier anyway
  for (int i = tid; i < 2 * H * RW; i += NT) s_rowI[i] = 0ull;
  if (tid == 0) *s_ncomp_p = 0;
  __syncthreads();
  if (wave == 0) {
    // combine the per-wave partials once; the result stays in LDS (s_R) for the later phases
    Red1 q;
    if (lane < NWAVE) q = s_red[lane];
    else {
      q.sumpix = 0; q.minpix = 0xFFFFFFFFu; q.maxpix = 0; q.fmin = 0x7FFFFFFF; q.fmax = -0x7FFFFFFF - 1;
      q.sumbg = 0; q.changed = 0; q.sumabs = 0;
    }
    q.sumpix = wave_sum(q.sumpix);
    q.minpix = wave_min(q.minpix);
    q.maxpix = wave_max(q.maxpix);
    q.fmin = wave_min(q.fmin);
    q.fmax = wave_max(q.fmax);
    q.sumbg = wave_sum(q.sumbg);
    q.changed = wave_max(q.changed);
    q.sumabs = wave_sum(q.sumabs);
    if (lane == 0) *s_R = q;
  }
  __syncthreads();
  // avg_change = int(round(np.average(thermal) - background.average))  (cliptracker.py:103-105)
  const double mean_thermal = (double)s_R->sumpix / (double)P;
  const int avg_change = (int)rint(mean_thermal - cs.bg_average);

  // ---- phase 2: shifted + clipped frame, min / max ------------------------------
  int mn = 0x7FFFFFFF, mx = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x = v[i][j] - avg_change;
      x = x < 0 ? 0 : x;
      v[i][j] = x;
      if (c < nchunk) {
        mn = min(mn, x);
        mx = max(mx, x);
      }
    }
  }
  mn = wave_min(mn);
  mx = wave_max(mx);
  if (lane == 0) {
    s_red2[2 * wave] = mn;
    s_red2[2 * wave + 1] = mx;
  }
  __syncthreads();
  mn = s_red2[0];
  mx = s_red2[1];
#pragma unroll
  for (int w = 1; w < NWAVE; ++w) {
    mn = min(mn, s_red2[2 * w]);
    mx = max(mx, s_red2[2 * w + 1]);
  }

  // ---- phase 3: normalise to 0..255 (float32, imageprocessing.py:151-169) -> uint8 in LDS
  float thresh;
  {
    const float fmn = (float)mn, fmx = (float)mx;
    const float span = fmx - fmn;
    if (mx == mn) {
      thresh = (float)a.background_thresh;  // raw threshold (cliptracker.py:118-119)
    } else {
      thresh = __fmul_rn(__fdiv_rn((float)a.background_thresh, span), 255.0f);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * NT;
      if (c < nchunk) {
        unsigned char o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float val;
          if (mx == mn) {
            val = (mx == 0) ? 0.0f : 1.0f;  // zeros, or data / max == 1
          } else {
            val = __fdiv_rn(__fmul_rn(255.0f, (float)v[i][j] - fmn), span);
          }
          o[j] = (unsigned char)(int)val;  // np.uint8() truncation
        }
        *reinterpret_cast<uchar4*>(s_u8 + (c << 2)) = make_uchar4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  const int ithr = (mx == mn) ? (int)floor(a.background_thresh) : (int)floorf(thresh);
  __syncthreads();

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
        if (cid < (u32)CAP) s_cid[slot] = (uint16_t)cid;
      }
    }
  }
  __syncthreads();
  const int ncomp_all = (int)*s_ncomp_p;
  const bool overflow = ncomp_all > CAP || ncomp_all > a.cap_out;
  const int ncomp = overflow ? 0 : ncomp_all;

  // ---- phase 7: statistics per component ------------------------------------------------
  // s_stat rows: 0 area, 1 minx, 2 maxx, 3 miny, 4 maxy, 5 sumx, 6 sumy, 7 key
  for (int i = tid; i < ncomp; i += NT) {
    s_stat[0 * CAP + i] = 0;
    s_stat[1 * CAP + i] = 0xFFFFFFFFu;
    s_stat[2 * CAP + i] = 0;
    s_stat[3 * CAP + i] = 0xFFFFFFFFu;
    s_stat[4 * CAP + i] = 0;
    s_stat[5 * CAP + i] = 0;
    s_stat[6 * CAP + i] = 0;
    s_stat[7 * CAP + i] = 0xFFFFFFFFu;
  }
  __syncthreads();
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
        atomicAdd(&s_stat[0 * CAP + cid], len);
        atomicMin(&s_stat[1 * CAP + cid], (u32)xs);
        atomicMax(&s_stat[2 * CAP + cid], (u32)xe);
        atomicMin(&s_stat[3 * CAP + cid], (u32)y);
        atomicMax(&s_stat[4 * CAP + cid], (u32)y);
        atomicAdd(&s_stat[5 * CAP + cid], (u32)((xs + xe) * (int)len / 2));
        atomicAdd(&s_stat[6 * CAP + cid], (u32)y * len);
        atomicMin(&s_stat[7 * CAP + cid], (u32)((y >> 1) * SW + (xs >> 1)));
      }
    }
  }
  __syncthreads();
  // OpenCV numbers components by the raster position of their first 2x2 block (SURVEY a7')
  for (int i = tid; i < ncomp; i += NT) {
    const u32 k = s_stat[7 * CAP + i];
    u32 rank = 0;
    for (int j = 0; j < ncomp; ++j) rank += (s_stat[7 * CAP + j] < k) ? 1u : 0u;
    s_rank[i] = rank;
  }
  __syncthreads();

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
            lab[j] = (int)s_rank[s_cid[root]] + 1;
          }
        }
      } else if (nib) {
        lab[0] = (nib & 1) ? -1 : 0; lab[1] = (nib & 2) ? -1 : 0; lab[2] = (nib & 4) ? -1 : 0; lab[3] = (nib & 8) ? -1 : 0;
      }
      *reinterpret_cast<int4*>(Lout + p0) = make_int4(lab[0], lab[1], lab[2], lab[3]);
    }
  }

  // ---- phase 8b: np.var(delta_filtered[bbox]) -- one wave per component ---------------------------
  // delta = |f32(norm255(cur.filtered)) - f32(norm255(prev.filtered))|  (cliptracker.py:249-261);
  // normalize() promotes to float64 for these float64 frames (NumPy >= 2 scalar promotion).
  Component* Cout = a.comps_out + (size_t)fidx * a.cap_out;
  const bool has_prev = cs.has_prev != 0;
  const double cmin = (double)s_R->fmin, cmax = (double)s_R->fmax;
  const double pmin = (double)cs.prev_fmin, pmax = (double)cs.prev_fmax;
  for (int cidx = wave; cidx < ncomp; cidx += NWAVE) {
    const int bx = (int)s_stat[1 * CAP + cidx], by = (int)s_stat[3 * CAP + cidx];
    const int bw = (int)s_stat[2 * CAP + cidx] - bx + 1, bh = (int)s_stat[4 * CAP + cidx] - by + 1;
    float var = 0.0f;
    if (has_prev) {
      const int n = bw * bh;
      double s1 = 0.0;
      for (int k = lane; k < n; k += 64) {
        const int yy = by + k / bw, xx = bx + (k - (k / bw) * bw);
        const int q = yy * W + xx;
        const float cv = filt_cur[q], pv = filt_prev[q];
        float an, bn;
        if (cmax == cmin) an = (cmax == 0.0) ? 0.0f : (float)((double)cv / cmax);
        else an = (float)((255.0 * ((double)cv - cmin)) / (cmax - cmin));
        if (pmax == pmin) bn = (pmax == 0.0) ? 0.0f : (float)((double)pv / pmax);
        else bn = (float)((255.0 * ((double)pv - pmin)) / (pmax - pmin));
        s1 += (double)fabsf(an - bn);
      }
      s1 = wave_sum(s1);
      const double mean = s1 / (double)n;
      double s2 = 0.0;
      for (int k = lane; k < n; k += 64) {
        const int yy = by + k / bw, xx = bx + (k - (k / bw) * bw);
        const int q = yy * W + xx;
        const float cv = filt_cur[q], pv = filt_prev[q];
        float an, bn;
        if (cmax == cmin) an = (cmax == 0.0) ? 0.0f : (float)((double)cv / cmax);
        else an = (float)((255.0 * ((double)cv - cmin)) / (cmax - cmin));
        if (pmax == pmin) bn = (pmax == 0.0) ? 0.0f : (float)((double)pv / pmax);
        else bn = (float)((255.0 * ((double)pv - pmin)) / (pmax - pmin));
        const double d = (double)fabsf(an - bn) - mean;
        s2 += d * d;
      }
      s2 = wave_sum(s2);
      var = (float)(s2 / (double)n);
    }
    if (lane == 0) {
      Component o;
      o.x = bx;
      o.y = by;
      o.width = bw;
      o.height = bh;
      o.area = (int)s_stat[0 * CAP + cidx];
      o.sum_x = (int)s_stat[5 * CAP + cidx];
      o.sum_y = (int)s_stat[6 * CAP + cidx];
      o.pixel_variance = var;
      Cout[s_rank[cidx]] = o;
    }
  }

  // ---- per-frame record + clip state -------------------------------------------------------------------
  if (tid == 0) {
    const Red1 R = *s_R;
    FrameInfo fi;
    fi.frame_number = t;
    fi.n_components = overflow ? ncomp_all : ncomp;
    fi.status = overflow ? -5 : 0;
    fi.ffc_affected = a.proc_ffc[pbase + t];
    fi.avg_change = avg_change;
    fi.norm_min = mn;
    fi.norm_max = mx;
    fi.threshold = thresh;
    fi.filt_min = R.fmin;
    fi.filt_max = R.fmax;
    fi.thermal_min = (int)R.minpix;
    fi.thermal_max = (int)R.maxpix;
    fi.thermal_sum = R.sumpix;
    fi.thermal_median = -1.0f;
    fi.filtered_abs_sum = R.sumabs;
    ClipState ns;
    // motiondetector.py:224-226: average = int(round(np.average(background))) when any pixel changed
    ns.bg_average = R.changed ? rint((double)R.sumbg / (double)((W - 2 * e) * (H - 2 * e))) : cs.bg_average;
    ns.prev_fmin = R.fmin;
    ns.prev_fmax = R.fmax;
    ns.has_prev = 1;
    ns.pad = 0;
    fi.background_average = ns.bg_average;
    fi.background_changed = (int)R.changed;
    fi.reserved = 0;
    a.info_out[fidx] = fi;
    a.cstate[b] = ns;
  }
}

// final background of each clip as float, edges replicated (motiondetector.py:239-244)
__global__ __launch_bounds__(256) void cpx_export_background_kernel(TrackArgs a, float* out) {
  const int b = blockIdx.x;
  const int W = a.W, H = a.H, P = W * H, e = a.edge;
  const int nproc = a.proc_off[b + 1] - a.proc_off[b];
  const int32_t* bg = a.bg + ((size_t)b * 2 + (nproc & 1)) * P;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    int y = p / W, x = p - y * W;
    out[(size_t)b * P + p] = (float)bg[clampi(y, e, H - 1 - e) * W + clampi(x, e, W - 1 - e)];
  }
}

size_t track_lds_bytes(int W, int H) {
  const size_t P = (size_t)W * H;
  return 3 * P + 2 * (size_t)H * RW * 8 + (size_t)9 * CAP * 4 + (NWAVE + 1) * sizeof(Red1) + NWAVE * 2 * sizeof(int) + 16;
}

void launch_init(const TrackArgs& a, int B, hipStream_t s) { hipLaunchKernelGGL(cpx_init_kernel, dim3(B), dim3(256), 0, s, a); }
void launch_frame(const TrackArgs& a, int B, int t, hipStream_t s) {
  hipLaunchKernelGGL(cpx_frame_kernel, dim3(B), dim3(NT), track_lds_bytes(a.W, a.H), s, a, t);
}
void launch_export_background(const TrackArgs& a, int B, float* out, hipStream_t s) {
  hipLaunchKernelGGL(cpx_export_background_kernel, dim3(B), dim3(256), 0, s, a, out);
}
int track_max_pixels() { return 4 * NCH * NT; }
int track_lds_components() { return CAP; }
int frame_kernel_attr_setup() {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(cpx_frame_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
}

}  // namespace cpx
