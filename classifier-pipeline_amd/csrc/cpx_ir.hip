// cpx_ir.hip -- detection stage of the IR (640x480) tracker, SURVEY section 8 f4: detect_objects_ir
// (ml_tools/imageprocessing.py:185-199) = uint8 image -> MORPH_OPEN with the reference's tuple kernel (a 1 x 2 element,
// see cpx_track.hip phase 5 for the close) -> threshold -> 8-connected components with OpenCV's statistics and label
// numbering (first 2x2 block in block-raster order).
//
// One workgroup per frame.  A 640 x 480 frame does not fit LDS as bytes, but everything after the threshold is binary:
// the frame lives in LDS as bit rows (10 x u64 per row, 38 KB), the open is three word operations per word, and the
// labelling works on RUNS: run r of row y is identified by row_off[y] + (number of run starts before it in the row),
// found for any pixel with one popcount (word prefix counts are kept per row).  Union-find (atomicMin) over the runs,
// per-component statistics with atomics, OpenCV's numbering from a bitmap of first blocks (a component's rank is
// the number of set bits before its own: one popcount after a word prefix scan, no sort).
// Tables: up to 8192 runs and 1024 components live in LDS.  A frame with more (dense noise) takes one of the
// handle's scratch slots in HBM for the table that does not fit -- same code, global atomics -- so the only overflow
// left is the caller's own max_components.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cpx_kernels.h"

namespace cpx {

namespace {
typedef unsigned int u32;
typedef unsigned long long u64;
constexpr int IT = 1024;     // threads
constexpr int RCAP = 8192;   // runs per frame held in LDS
constexpr int CCAP = 1024;   // components per frame held in LDS

struct IrLds {
  u64* rowI;      // [max(H*NW, 4*CCAP)] thresholded input; dead after the open, reused for the statistics
  u64* rowO;      // [H*NW] after the open
  uint16_t* wpre; // [H*NW] run starts of the row before word w
  u32* row_off;   // [H+1]
  u32* par;       // [RCAP]
  u64* kbits;     // [NKW] bitmap of first blocks
  u32* kpre;      // [NKW] set bits before word k
  u32* misc;      // [8]: 0 total runs, 1 components, 3 scratch slot + 1
};

// run starts of word (y, w)
__device__ __forceinline__ u64 starts_of(const u64* O, int NW, int y, int w) {
  const u64 o = O[y * NW + w];
  const u64 carry = w > 0 ? (O[y * NW + w - 1] >> 63) : 0ull;
  return o & ~((o << 1) | carry);
}
// index of the run of row y that contains (or, scanning left, last started at or before) pixel x
__device__ __forceinline__ int run_index(const IrLds& L, int NW, int y, int x) {
  const int w = x >> 6, b = x & 63;
  const u64 s = starts_of(L.rowO, NW, y, w);
  const u64 m = (b == 63) ? ~0ull : ((2ull << b) - 1ull);
  return (int)L.row_off[y] + (int)L.wpre[y * NW + w] + __popcll(s & m) - 1;
}
__device__ __forceinline__ int uf_find(u32* par, int i) {
  int p = (int)par[i];
  while (p != i) {
    i = p;
    p = (int)par[i];
  }
  return i;
}
__device__ __forceinline__ void uf_union(u32* par, int a, int b) {
  for (;;) {
    a = uf_find(par, a);
    b = uf_find(par, b);
    if (a == b) return;
    if (a > b) {
      const int t = a;
      a = b;
      b = t;
    }
    const u32 old = atomicMin(&par[b], (u32)a);  // a < b
    if (old == (u32)b) return;
    b = (int)old;
  }
}

// a free scratch slot (thread 0 only); holders never wait on anything, so the spin ends
__device__ int slot_acquire(u32* bitmap, int nslots) {
  for (;;) {
    for (int s = 0; s < nslots; ++s) {
      const u32 bit = 1u << (s & 31);
      if (__hip_atomic_load(&bitmap[s >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) continue;
      if (!(atomicOr(&bitmap[s >> 5], bit) & bit)) return s;
    }
    __builtin_amdgcn_s_sleep(32);
  }
}
__device__ void slot_release(u32* bitmap, int s) {
  __threadfence();
  atomicAnd(&bitmap[s >> 5], ~(1u << (s & 31)));
}

// ---- labelling: unions at the first pixel of every contact with the row above, flatten, roots -> component slots.
// Afterwards par[i] = R + (component slot of run i) for every run.
__device__ __forceinline__ void label_runs(const IrLds& L, u32* par, int R, int W, int H, int NW, int tid) {
  for (int i = tid; i < R; i += IT) par[i] = (u32)i;
  __syncthreads();
  for (int i = tid; i < H * NW; i += IT) {
    const int y = i / NW, w = i - y * NW;
    if (y == 0) continue;
    const u64 oy = L.rowO[i];
    if (!oy) continue;
    const u64 up = L.rowO[i - NW];
    const u64 upl = (up << 1) | (w > 0 ? (L.rowO[i - NW - 1] >> 63) : 0ull);       // pixel x-1 of the row above
    const u64 upr = (up >> 1) | (w + 1 < NW ? (L.rowO[i - NW + 1] << 63) : 0ull);  // pixel x+1 of the row above
    const u64 kn = oy & up;
    const u64 kn_prev = w > 0 ? ((L.rowO[i - 1] & L.rowO[i - NW - 1]) >> 63) : 0ull;
    u64 first = kn & ~((kn << 1) | kn_prev);  // one union per stretch of vertical contact
    while (first) {
      const int b = __ffsll((long long)first) - 1;
      first &= first - 1;
      const int x = w * 64 + b;
      uf_union(par, run_index(L, NW, y, x), run_index(L, NW, y - 1, x));
    }
    u64 knw = oy & upl & ~up;  // diagonal-only contacts
    while (knw) {
      const int b = __ffsll((long long)knw) - 1;
      knw &= knw - 1;
      const int x = w * 64 + b;
      uf_union(par, run_index(L, NW, y, x), run_index(L, NW, y - 1, x - 1));
    }
    u64 kne = oy & upr & ~up;
    while (kne) {
      const int b = __ffsll((long long)kne) - 1;
      kne &= kne - 1;
      const int x = w * 64 + b;
      uf_union(par, run_index(L, NW, y, x), run_index(L, NW, y - 1, x + 1));
    }
  }
  __syncthreads();
  for (int i = tid; i < R; i += IT) par[i] = (u32)uf_find(par, i);  // roots never change here
  __syncthreads();
  for (int i = tid; i < R; i += IT)
    if (par[i] == (u32)i) par[i] = (u32)R + atomicAdd(&L.misc[1], 1u);  // root: its slot, marked by >= R
  __syncthreads();
  for (int i = tid; i < R; i += IT) {
    const u32 p = par[i];
    if (p < (u32)R) par[i] = par[p];  // p is a root: already encoded
  }
  __syncthreads();
}

// ---- statistics, numbering, outputs.  st: [8][cs] u32: area (then label), minx, maxx, miny, maxy, sumx, sumy, key
__device__ __forceinline__ void measure_runs(const IrLds& L, const u32* par, u32* st, int cs, int R, int C, int W, int H,
                                             int NW, int tid, cpx_component* out, int32_t* lab) {
  for (int i = tid; i < C; i += IT) {
    st[0 * cs + i] = 0;
    st[1 * cs + i] = 0xFFFFFFFFu;
    st[2 * cs + i] = 0;
    st[3 * cs + i] = 0xFFFFFFFFu;
    st[4 * cs + i] = 0;
    st[5 * cs + i] = 0;
    st[6 * cs + i] = 0;
    st[7 * cs + i] = 0xFFFFFFFFu;
  }
  const int BW2 = (W + 1) >> 1, NKW = (BW2 * ((H + 1) >> 1) + 63) >> 6;
  for (int i = tid; i < NKW; i += IT) L.kbits[i] = 0;
  __syncthreads();
  for (int i = tid; i < H * NW; i += IT) {
    const int y = i / NW, w = i - y * NW;
    u64 s = starts_of(L.rowO, NW, y, w);
    int k = 0;
    while (s) {
      const int b = __ffsll((long long)s) - 1;
      s &= s - 1;
      const int xs = w * 64 + b;
      const u64 rest = ~L.rowO[i] >> b;  // bit 0 <-> pixel xs, which is set: bit 0 of rest is clear
      int len;
      if (rest) {
        len = __ffsll((long long)rest) - 1;
      } else {
        len = 64 - b;
        for (int ww = w + 1; ww < NW; ++ww) {
          const u64 inv = ~L.rowO[y * NW + ww];
          if (inv) {
            len += __ffsll((long long)inv) - 1;
            break;
          }
          len += 64;
        }
      }
      const int xe = xs + len - 1;
      const int run = (int)L.row_off[y] + (int)L.wpre[i] + k;
      ++k;
      const int c = (int)(par[run] - (u32)R);
      atomicAdd(&st[0 * cs + c], (u32)len);
      atomicMin(&st[1 * cs + c], (u32)xs);
      atomicMax(&st[2 * cs + c], (u32)xe);
      atomicMin(&st[3 * cs + c], (u32)y);
      atomicMax(&st[4 * cs + c], (u32)y);
      atomicAdd(&st[5 * cs + c], (u32)((xs + xe) * len / 2));
      atomicAdd(&st[6 * cs + c], (u32)(y * len));
      atomicMin(&st[7 * cs + c], (u32)((y >> 1) * BW2 + (xs >> 1)));
    }
  }
  __syncthreads();
  // OpenCV numbers components by their first 2x2 block in block-raster order; all pixels of a block belong to one
  // 8-connected component, so the keys are distinct and a component's rank is the number of keys below its own
  for (int c = tid; c < C; c += IT) {
    const u32 key = st[7 * cs + c];
    atomicOr(&L.kbits[key >> 6], 1ull << (key & 63));
  }
  __syncthreads();
  if (tid < 64) {
    const int per = (NKW + 63) >> 6;
    int sum = 0;
    for (int k = tid * per; k < min(NKW, (tid + 1) * per); ++k) sum += __popcll(L.kbits[k]);
    int incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (tid >= d) incl += v;
    }
    int run = incl - sum;
    for (int k = tid * per; k < min(NKW, (tid + 1) * per); ++k) {
      L.kpre[k] = (u32)run;
      run += __popcll(L.kbits[k]);
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += IT) {
    const u32 key = st[7 * cs + c];
    const int rank = (int)L.kpre[key >> 6] + __popcll(L.kbits[key >> 6] & ((1ull << (key & 63)) - 1ull));
    cpx_component o;
    o.x = (int)st[1 * cs + c];
    o.y = (int)st[3 * cs + c];
    o.width = (int)st[2 * cs + c] - o.x + 1;
    o.height = (int)st[4 * cs + c] - o.y + 1;
    o.area = (int)st[0 * cs + c];
    o.sum_x = (int)st[5 * cs + c];
    o.sum_y = (int)st[6 * cs + c];
    o.pixel_variance = 0.0f;
    out[rank] = o;
    st[0 * cs + c] = (u32)(rank + 1);  // the area is written out: its cell carries the label for the image pass
  }
  if (!lab) return;
  __syncthreads();
  for (int q = tid; q < ((W * H) >> 2); q += IT) {  // every pixel finds its run with one popcount
    const int p0 = q << 2;
    const int y = p0 / W, x0 = p0 - y * W;
    const u64 o = L.rowO[y * NW + (x0 >> 6)];
    int v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = x0 + j;
      v[j] = 0;
      if ((o >> (x & 63)) & 1ull) v[j] = (int)st[par[run_index(L, NW, y, x)] - (u32)R];
    }
    *reinterpret_cast<int4*>(lab + p0) = make_int4(v[0], v[1], v[2], v[3]);
  }
}
}  // namespace

__global__ __launch_bounds__(IT) void cpx_ir_detect_kernel(IrArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int f = blockIdx.x;
  const int W = a.W, H = a.H, NW = W >> 6, P = W * H;
  const int tid = threadIdx.x;
  const int NKW = (((W + 1) >> 1) * ((H + 1) >> 1) + 63) >> 6;
  IrLds L;
  L.rowI = reinterpret_cast<u64*>(smem);
  L.rowO = L.rowI + (H * NW > CCAP * 4 ? H * NW : CCAP * 4);  // the statistics ([8][CCAP] u32) reuse the input rows
  L.kbits = L.rowO + H * NW;
  L.kpre = reinterpret_cast<u32*>(L.kbits + NKW);
  L.par = L.kpre + ((NKW + 1) & ~1);
  L.row_off = L.par + RCAP;
  L.misc = L.row_off + ((H + 1 + 3) & ~3);
  L.wpre = reinterpret_cast<uint16_t*>(L.misc + 8);
  const unsigned char* img = a.images + (size_t)f * P;

  // ---- 1. np.uint8 image -> bit rows of (pixel > threshold); min / max of the 1 x 2 open commute with the threshold
  for (int i = tid; i < H * NW; i += IT) {
    const uint4* src = reinterpret_cast<const uint4*>(img + (size_t)i * 64);
    u64 bits = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = src[q];
      const u32 ws[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
          bits |= (u64)(((ws[k] >> (8 * bb)) & 0xFFu) > (u32)a.threshold) << (q * 16 + k * 4 + bb);
    }
    L.rowI[i] = bits;
  }
  if (tid < 8) L.misc[tid] = 0;
  __syncthreads();
  // ---- 2. MORPH_OPEN with the 1 x 2 element: erode E[y] = I[y] & I[y-1] (E[0] = I[0]), dilate O[y] = E[y] | E[y-1]
  for (int i = tid; i < H * NW; i += IT) {
    const int y = i / NW;
    const u64 i0 = L.rowI[i];
    const u64 i1 = y >= 1 ? L.rowI[i - NW] : ~0ull;
    const u64 i2 = y >= 2 ? L.rowI[i - 2 * NW] : ~0ull;
    const u64 e0 = i0 & i1;                    // E[y]   (row -1 is outside: & with all ones)
    const u64 e1 = y >= 1 ? (i1 & i2) : 0ull;  // E[y-1] (absent for y = 0)
    L.rowO[i] = e0 | e1;
  }
  __syncthreads();
  // ---- 3. run starts per row: word prefix counts, row offsets ----
  for (int y = tid; y < H; y += IT) {
    int c = 0;
    for (int w = 0; w < NW; ++w) {
      L.wpre[y * NW + w] = (uint16_t)c;
      c += __popcll(starts_of(L.rowO, NW, y, w));
    }
    L.row_off[y + 1] = (u32)c;  // per-row count for now
  }
  __syncthreads();
  if (tid < 64) {  // exclusive scan of the per-row counts by one wave: a few rows per lane, then a shuffle scan
    const int per = (H + 63) >> 6;
    const int r0 = tid * per, r1 = min(H, r0 + per);
    u32 sum = 0;
    for (int y = r0; y < r1; ++y) sum += L.row_off[y + 1];
    u32 incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
      const u32 v = __shfl_up(incl, d, 64);
      if (tid >= d) incl += v;
    }
    u32 run = incl - sum;
    for (int y = r0; y < r1; ++y) {
      const u32 c = L.row_off[y + 1];
      run += c;
      L.row_off[y + 1] = run;  // inclusive: offset of row y + 1 (lanes only touch their own rows)
    }
    if (tid == 0) L.row_off[0] = 0;
    if (tid == 63) {
      L.misc[0] = incl;
      if (incl > (u32)RCAP) L.misc[3] = (u32)slot_acquire(a.slot_bitmap, a.n_slots) + 1u;
    }
  }
  __syncthreads();
  const int R = (int)L.misc[0];
  const bool big_runs = R > RCAP;
  // scratch slot: par u32 [P/2 + 64], statistics u32 [8][cs_big]
  const int cs_big = ((W + 1) >> 1) * ((H + 1) >> 1);
  u32* slot = L.misc[3] ? reinterpret_cast<u32*>(a.slots + (size_t)(L.misc[3] - 1) * a.slot_bytes) : nullptr;
  // ---- 4. labelling ----
  if (!big_runs)
    label_runs(L, L.par, R, W, H, NW, tid);
  else
    label_runs(L, slot, R, W, H, NW, tid);
  const int C = (int)L.misc[1];
  if (C > a.max_components) {  // the caller's table is too small: report, never truncate
    if (tid == 0) {
      a.counts[f] = C;
      a.status[f] = CPX_ERR_OVERFLOW;
      if (L.misc[3]) slot_release(a.slot_bitmap, (int)L.misc[3] - 1);
    }
    return;
  }
  const bool big_comps = C > CCAP;
  if (big_comps && !big_runs) {
    if (tid == 0) L.misc[3] = (u32)slot_acquire(a.slot_bitmap, a.n_slots) + 1u;
    __syncthreads();
    slot = reinterpret_cast<u32*>(a.slots + (size_t)(L.misc[3] - 1) * a.slot_bytes);
  }
  // ---- 5. statistics, numbering, outputs ----
  cpx_component* out = a.comps + (size_t)f * a.max_components;
  int32_t* lab = a.labels ? a.labels + (size_t)f * P : nullptr;
  u32* st_lds = reinterpret_cast<u32*>(L.rowI);
  u32* st_big = slot + (P / 2 + 64);
  if (!big_runs && !big_comps)
    measure_runs(L, L.par, st_lds, CCAP, R, C, W, H, NW, tid, out, lab);
  else if (big_runs && !big_comps)
    measure_runs(L, slot, st_lds, CCAP, R, C, W, H, NW, tid, out, lab);
  else if (!big_runs)
    measure_runs(L, L.par, st_big, cs_big, R, C, W, H, NW, tid, out, lab);
  else
    measure_runs(L, slot, st_big, cs_big, R, C, W, H, NW, tid, out, lab);
  __syncthreads();
  if (tid == 0) {
    a.counts[f] = C;
    a.status[f] = 0;
    if (L.misc[3]) slot_release(a.slot_bitmap, (int)L.misc[3] - 1);
  }
}

size_t ir_lds_bytes(int W, int H) {
  const size_t NW = (size_t)W >> 6, HW = (size_t)H * NW;
  const size_t in_words = HW > (size_t)CCAP * 4 ? HW : (size_t)CCAP * 4;
  const size_t NKW = ((size_t)((W + 1) >> 1) * ((H + 1) >> 1) + 63) >> 6;
  return (in_words + HW + NKW) * 8 + ((NKW + 1) & ~(size_t)1) * 4 + (size_t)RCAP * 4 + (((size_t)H + 1 + 3) & ~(size_t)3) * 4 +
         32 + ((HW + 7) & ~(size_t)7) * 2;
}
int ir_supported(int W, int H) {
  return W >= 64 && (W % 64) == 0 && H >= 1 && ir_lds_bytes(W, H) <= 160 * 1024 - 1024;
}
size_t ir_slot_bytes(int W, int H) {
  const size_t P = (size_t)W * H, cs = (size_t)((W + 1) >> 1) * ((H + 1) >> 1);
  return ((P / 2 + 64 + 8 * cs) * 4 + 255) & ~(size_t)255;
}

// np.var(np.abs(frame.thermal - prev.thermal)[region]) of the IR tracker (track/irtrackextractor.py:638-655,
// track/cliptracker.py:303-312): both frames are uint8, so the difference wraps modulo 256 and np.abs leaves it as
// it is.  One wavefront per region; the sums are exact integers (n * S2 - S1^2 < 2^63 for a 640 x 480 box), one
// division at the end -- NumPy's two-pass float64 evaluation agrees to rounding noise.
__global__ __launch_bounds__(64) void cpx_ir_delta_var_kernel(IrVarArgs a) {
  const int r = blockIdx.x;
  const int x0 = a.rects[4 * r], y0 = a.rects[4 * r + 1];
  const int x1 = min(x0 + a.rects[4 * r + 2], a.W), y1 = min(y0 + a.rects[4 * r + 3], a.H);  // slicing clips
  const int w = x1 - x0, hgt = y1 - y0;
  unsigned long long s1 = 0, s2 = 0;
  if (w > 0 && hgt > 0 && x0 >= 0 && y0 >= 0) {
    const int n = w * hgt;
    for (int k = threadIdx.x; k < n; k += 64) {
      const int yy = k / w, xx = k - yy * w;
      const int p = (y0 + yy) * a.W + x0 + xx;
      const unsigned d = ((unsigned)a.cur[p] - (unsigned)a.prev[p]) & 255u;
      s1 += d;
      s2 += (unsigned long long)d * d;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if (threadIdx.x == 0) {
    const long long n = (w > 0 && hgt > 0 && x0 >= 0 && y0 >= 0) ? (long long)w * hgt : 0;
    a.out[r] = n ? (double)((unsigned long long)n * s2 - s1 * s1) / ((double)n * (double)n) : NAN;  // np.var([]) is nan
  }
}

// merge_components of the IR tracker (track/irtrackextractor.py:324-389) for one frame per workgroup, then the
// variance of the frame difference over every merged box (the kernel above, per box) -- the two host steps between
// cpx_ir_detect and the association when a batch of videos advances in lockstep.
//   rows with area > 40 or (width > 16 and height > 16) survive, largest area first (stable);
//   a row absorbs every other row whose ORIGINAL box is closer than 40 pixels to its original box (gap measured per
//   axis, 0 where the extents overlap: dx^2 + dy^2 < 1600 -- the reference compares the square root with 40) or
//   overlaps it on both axes -- except rows with the same x (the reference's self test also skips those); the merged
//   box takes the union, its height measured from the already-updated top (as the reference computes it); the scan
//   restarts from the first row after any merge.
// Thread 0 does the merge (a few dozen rows: scalar work, kept off the host so that no synchronisation separates the
// steps of a frame); every wave then takes boxes for the variance.
__global__ __launch_bounds__(256) void cpx_ir_merge_kernel(IrMergeArgs a) {
  extern __shared__ int s_rows[];          // [cap_out][5] original boxes | [cap_out][5] merged boxes
  __shared__ int s_n, s_status;
  const int v = blockIdx.x;
  const int cap = a.cap_out;
  int* anchor = s_rows;
  int* box = s_rows + cap * 5;
  const size_t row = (size_t)v * a.out_stride + a.frame_number;
  if (threadIdx.x == 0) {
    const cpx_component* in = a.comps + (size_t)v * a.cap_in;
    const int nin = min(a.counts[v], a.cap_in);
    int n = 0, status = 0;
    for (int i = 0; i < nin; ++i) {
      const cpx_component c = in[i];
      if (!(c.area > 40 || (c.width > 16 && c.height > 16))) continue;
      if (n >= cap) {
        status = CPX_ERR_OVERFLOW;
        break;
      }
      // insertion by area, descending, stable (sorted(..., key=area, reverse=True) keeps equal areas in input order)
      int k = n;
      while (k > 0 && anchor[(k - 1) * 5 + 4] < c.area) {
        for (int j = 0; j < 5; ++j) anchor[k * 5 + j] = anchor[(k - 1) * 5 + j];
        --k;
      }
      anchor[k * 5 + 0] = c.x;
      anchor[k * 5 + 1] = c.y;
      anchor[k * 5 + 2] = c.width;
      anchor[k * 5 + 3] = c.height;
      anchor[k * 5 + 4] = c.area;
      ++n;
    }
    for (int i = 0; i < n * 5; ++i) box[i] = anchor[i];
    int i = 0;
    while (i < n) {
      const int ax = anchor[i * 5], ay = anchor[i * 5 + 1], aw = anchor[i * 5 + 2], ah = anchor[i * 5 + 3];
      bool absorbed = false;
      int j = 0;
      while (j < n) {
        const int ox = anchor[j * 5], oy = anchor[j * 5 + 1], ow = anchor[j * 5 + 2], oh = anchor[j * 5 + 3];
        if (ox == ax) {
          ++j;
          continue;
        }
        const bool over_x = ow + aw > max(ox + ow, ax + aw) - min(ox, ax);
        const bool over_y = oh + ah > max(oy + oh, ay + ah) - min(oy, ay);
        int dx = 0, dy = 0;
        if (!over_x) dx = ax < ox ? (ax + aw) - ox : (ox + ow) - ax;
        if (!over_y) dy = ay < oy ? (ay + ah) - oy : (oy + oh) - ay;
        if (!((over_x && over_y) || dx * dx + dy * dy < 1600)) {
          ++j;
          continue;
        }
        int* b = box + i * 5;
        const int right = b[0] + b[2];
        b[0] = min(b[0], ox);
        b[1] = min(b[1], oy);
        const int bottom = max(b[1] + b[3], oy + oh);   // (with the top already updated, as the reference does)
        b[2] = max(right, ox + ow) - b[0];
        b[3] = bottom - b[1];
        b[4] += anchor[j * 5 + 4];
        absorbed = true;
        for (int k = j; k + 1 < n; ++k)
          for (int q = 0; q < 5; ++q) {
            anchor[k * 5 + q] = anchor[(k + 1) * 5 + q];
            box[k * 5 + q] = box[(k + 1) * 5 + q];
          }
        --n;
        if (j < i) --i;   // the anchor moved down with the list
      }
      i = absorbed ? 0 : i + 1;
    }
    s_n = n;
    s_status = status;
  }
  __syncthreads();
  const int n = s_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  cpx_component* out = a.out_comps + row * cap;
  const unsigned char* cur = a.cur + (size_t)v * a.W * a.H;
  const unsigned char* prev = a.prev ? a.prev + (size_t)v * a.W * a.H : nullptr;
  for (int r = wave; r < n; r += 4) {
    const int x0 = box[r * 5], y0 = box[r * 5 + 1], bw = box[r * 5 + 2], bh = box[r * 5 + 3], mass = box[r * 5 + 4];
    double var = 0.0;
    if (prev) {
      const int x1 = min(x0 + bw, a.W), y1 = min(y0 + bh, a.H);
      const int w = x1 - x0, hgt = y1 - y0;
      unsigned long long s1 = 0, s2 = 0;
      const bool ok = w > 0 && hgt > 0 && x0 >= 0 && y0 >= 0;
      if (ok) {
        const int cnt = w * hgt;
        for (int k = lane; k < cnt; k += 64) {
          const int yy = k / w, xx = k - yy * w;
          const int p = (y0 + yy) * a.W + x0 + xx;
          const unsigned d = ((unsigned)cur[p] - (unsigned)prev[p]) & 255u;
          s1 += d;
          s2 += (unsigned long long)d * d;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
      }
      const long long cnt = ok ? (long long)w * hgt : 0;
      var = cnt ? (double)((unsigned long long)cnt * s2 - s1 * s1) / ((double)cnt * (double)cnt) : NAN;
    }
    if (lane == 0) {
      cpx_component c;
      c.x = x0;
      c.y = y0;
      c.width = bw;
      c.height = bh;
      c.area = mass;
      const int cx = (int)(x0 + bw / 2.0), cy = (int)(y0 + bh / 2.0);   // the IR tracker's centroid: box centre, truncated
      c.sum_x = cx * mass;
      c.sum_y = cy * mass;
      c.pixel_variance = (float)var;
      out[r] = c;
    }
  }
  if (threadIdx.x == 0) {
    a.status[v] = s_status;
    if (a.out_info) {
      cpx_frame_info fi;
      memset(&fi, 0, sizeof(fi));
      fi.frame_number = a.frame_number;
      fi.n_components = n;
      fi.status = s_status;
      a.out_info[row] = fi;
    }
  }
}

void launch_ir_merge(const IrMergeArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(cpx_ir_merge_kernel, dim3(a.n), dim3(256), (size_t)a.cap_out * 10 * sizeof(int), s, a);
}


// ---- per-frame statistics of the IR clip (Clip.add_frame's min / max / median / mean of the frame and the sum of the
// foreground image, track/clip.py:330-347) ------------------------------------------------------------------------------
// Pass 1: blockIdx = (slice, frame); a block histograms its slice of the frame in LDS (one 256-bin table per wave) and
// adds it to the frame's global histogram; the foreground bytes of the slice are summed on the way.
constexpr int STAT_SPLIT = 16;
__global__ __launch_bounds__(256) void cpx_ir_hist_kernel(IrStatsArgs a) {
  __shared__ unsigned int hist[4][256];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 4 * 256; i += 256) (&hist[0][0])[i] = 0;
  __syncthreads();
  const int f = blockIdx.y;
  const size_t P = (size_t)a.pixels;
  const unsigned char* fr = a.frames + (size_t)f * P;
  const unsigned char* mk = a.masks ? a.masks + (size_t)f * P : nullptr;
  unsigned long long msum = 0;
  if (a.vec16) {
    const size_t nv = P / 16, per = (nv + STAT_SPLIT - 1) / STAT_SPLIT;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < nv ? lo + per : nv;
    const uint4* fv = reinterpret_cast<const uint4*>(fr);
    const uint4* mv = reinterpret_cast<const uint4*>(mk);
    for (size_t i = lo + tid; i < hi; i += 256) {
      const uint4 q = fv[i];
      const unsigned int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(&hist[wave][w[k] & 255u], 1u);
        atomicAdd(&hist[wave][(w[k] >> 8) & 255u], 1u);
        atomicAdd(&hist[wave][(w[k] >> 16) & 255u], 1u);
        atomicAdd(&hist[wave][w[k] >> 24], 1u);
      }
      if (mk) {
        const uint4 m = mv[i];
        const unsigned int u[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) msum += (u[k] & 255u) + ((u[k] >> 8) & 255u) + ((u[k] >> 16) & 255u) + (u[k] >> 24);
      }
    }
  } else {
    const size_t per = (P + STAT_SPLIT - 1) / STAT_SPLIT;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < P ? lo + per : P;
    for (size_t i = lo + tid; i < hi; i += 256) {
      atomicAdd(&hist[wave][fr[i]], 1u);
      if (mk) msum += mk[i];
    }
  }
  __syncthreads();
  const unsigned int c = hist[0][tid] + hist[1][tid] + hist[2][tid] + hist[3][tid];
  if (c) atomicAdd(&a.hist[(size_t)f * 256 + tid], c);
  if (mk) {
    for (int off = 32; off > 0; off >>= 1) msum += __shfl_down(msum, off, 64);
    if ((tid & 63) == 0 && msum) atomicAdd(reinterpret_cast<unsigned long long*>(&a.out[f].filtered_sum), msum);
  }
}

// Pass 2: one block per frame reads the 256 bins: minimum / maximum = the first / last occupied bin, sum = sum of
// bin * count, median (np.median of an even or odd count) = the mean of order statistics (P - 1) / 2 and P / 2.
__global__ __launch_bounds__(256) void cpx_ir_stats_kernel(IrStatsArgs a) {
  __shared__ unsigned int cum[256];
  __shared__ int mn, mx, below_lo, below_hi;
  __shared__ unsigned long long total;
  const int tid = threadIdx.x, f = blockIdx.x;
  const unsigned int c = a.hist[(size_t)f * 256 + tid];
  if (tid == 0) { mn = 255; mx = 0; below_lo = 0; below_hi = 0; total = 0; }
  cum[tid] = c;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {   // inclusive scan
    const unsigned int add = tid >= off ? cum[tid - off] : 0u;
    __syncthreads();
    cum[tid] += add;
    __syncthreads();
  }
  const unsigned long long P = (unsigned long long)a.pixels;
  if (c) { atomicMin(&mn, tid); atomicMax(&mx, tid); atomicAdd(&total, (unsigned long long)c * (unsigned)tid); }
  if ((unsigned long long)cum[tid] < (P + 1) / 2) atomicAdd(&below_lo, 1);
  if ((unsigned long long)cum[tid] < P / 2 + 1) atomicAdd(&below_hi, 1);
  __syncthreads();
  if (tid == 0) {
    cpx_ir_frame_stats& o = a.out[f];
    o.min = mn; o.max = mx; o.sum = (long long)total; o.median_x2 = below_lo + below_hi; o.reserved = 0;
  }
}

void launch_ir_frame_stats(const IrStatsArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(cpx_ir_hist_kernel, dim3(STAT_SPLIT, a.n), dim3(256), 0, s, a);
  hipLaunchKernelGGL(cpx_ir_stats_kernel, dim3(a.n), dim3(256), 0, s, a);
}

void launch_ir_delta_variance(const IrVarArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(cpx_ir_delta_var_kernel, dim3(a.n), dim3(64), 0, s, a);
}

int launch_ir_detect(const IrArgs& a, int n_frames, hipStream_t s) {
  static bool lds_ready[64];
  if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(cpx_ir_detect_kernel), lds_ready, 160 * 1024 - 1024)) return -1;
  hipLaunchKernelGGL(cpx_ir_detect_kernel, dim3(n_frames), dim3(IT), ir_lds_bytes(a.W, a.H), s, a);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// cv2.resize(uint8, (W / f, H / f), interpolation=cv2.INTER_AREA) for an integer factor f: what the IR tracker does to
// its foreground image when it is given a `scale` (track/irtrackextractor.py:445-451; the Pi runs scale = 0.25).  With
// an integer ratio OpenCV averages the f x f block: factor 2 as (sum + 2) >> 2 (its SIMD path), any other factor as
// the sum times the float32 1 / f^2, rounded to nearest-even (saturate_cast<uchar>).  One thread per output pixel;
// lanes of a wave read adjacent f-byte runs of the same source rows.
__global__ __launch_bounds__(256) void cpx_ir_resize_area_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                                 int n, int W, int H, int f) {
  const int Wo = W / f, Ho = H / f;
  const long long total = (long long)n * Wo * Ho;
  const float inv = 1.0f / (float)(f * f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wo);
    const long long r = i / Wo;
    const int y = (int)(r % Ho);
    const int k = (int)(r / Ho);
    const unsigned char* p = src + ((size_t)k * H + (size_t)y * f) * W + (size_t)x * f;
    unsigned sum = 0;
    for (int dy = 0; dy < f; ++dy)
      for (int dx = 0; dx < f; ++dx) sum += p[(size_t)dy * W + dx];
    unsigned v;
    if (f == 2) v = (sum + 2u) >> 2;
    else v = (unsigned)__float2int_rn((float)sum * inv);
    dst[i] = (unsigned char)(v > 255u ? 255u : v);
  }
}
void launch_ir_resize_area(const unsigned char* src, unsigned char* dst, int n, int W, int H, int f, hipStream_t s) {
  const long long total = (long long)n * (W / f) * (H / f);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
  hipLaunchKernelGGL(cpx_ir_resize_area_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, src, dst, n, W, H, f);
}

}  // namespace cpx
