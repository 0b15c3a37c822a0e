// cpx_classify.hip -- classification pre-processing kernels (HBM-bound):
//   cpx_limits_kernel : per track, min / max of the filtered crops + clip_thermals_at_zero
//   cpx_crop_kernel   : per tile, crop -> bilinear resize -> pad -> median shift / clip ->
//                       normalise -> write into the 5x5 tiled NHWC sample
// All arithmetic is float32 in the order NumPy evaluates it in the reference
// (ml_tools/preprocess.py:56-113, ml_tools/imageprocessing.py:11-82,151-169); cv2.resize
// semantics are SURVEY.md A.8 (half-pixel centres, float32 weights, edge clamp).
#include <hip/hip_runtime.h>
#include <math.h>

#include "cpx_kernels.h"

namespace cpx {

namespace {
typedef unsigned int u32;

template <typename T>
__device__ __forceinline__ T wsum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <typename T>
__device__ __forceinline__ T wmin(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T w = __shfl_xor(v, o);
    v = w < v ? w : v;
  }
  return v;
}
template <typename T>
__device__ __forceinline__ T wmax(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T w = __shfl_xor(v, o);
    v = w > v ? w : v;
  }
  return v;
}

constexpr int LT = 256;  // threads of both kernels
// the limits kernel: sixteen waves per track -- a wave walks its share of the track's regions through three dependent trips to
// memory each, and the longest track of a batch is the kernel's time (four waves: 1.6 ms per 4096 clips; sixteen: see profiles/r06_track_experiments.md)
#ifndef CPX_LIMITS_THREADS
#define CPX_LIMITS_THREADS 1024
#endif
constexpr int LTL = CPX_LIMITS_THREADS, LWL = LTL / 64;
constexpr int LW = LT / 64;

// block-wide reductions through a small LDS scratch (all threads get the result)
__device__ __forceinline__ float block_min(float v, float* sc) {
  v = wmin(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sc[0];
  for (int w = 1; w < LW; ++w) r = fminf(r, sc[w]);
  return r;
}
__device__ __forceinline__ float block_max(float v, float* sc) {
  v = wmax(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sc[0];
  for (int w = 1; w < LW; ++w) r = fmaxf(r, sc[w]);
  return r;
}
__device__ __forceinline__ u32 block_sum_u32(u32 v, u32* sc) {
  v = wsum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = v;
  __syncthreads();
  u32 r = 0;
  for (int w = 0; w < LW; ++w) r += sc[w];
  return r;
}
}  // namespace

// ---------------------------------------------------------------------------------------
// get_limits + the clip_thermals_at_zero test, one workgroup per track
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(LTL) void cpx_limits_kernel(ClassifyArgs a) {
  // the track's regions are dealt to the workgroup's waves; a wave handles a region with shuffles only (no
  // workgroup barrier inside the per-region median bisection), the waves meet once at the end
  __shared__ float s_mn[LWL], s_mx[LWL], s_tmn[LWL], s_tmx[LWL];
  __shared__ int s_clip[LWL];
  const int t = blockIdx.x;
  const int r0 = a.track_offsets[t], r1 = a.track_offsets[t + 1];
  const int W = a.W, P = a.W * a.H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // get_limits: max_diff starts at 0, min_diff at None (interpreter.py:316-317).  post_process_file: the first sampled
  // crop's own min / max start both, and only sampled frames count (clipclassifier.py:484-497)
  const bool post = (a.limits_flags & CPX_LIMITS_POST_PROCESS) != 0;
  float mn = INFINITY, mx = post ? -INFINITY : 0.0f;
  float tmn = INFINITY, tmx = -INFINITY;  // thermal_norm_limits: thermal - median over the whole frame (get_limits)
  int clip0 = 1;
  for (int r = r0 + wave; r < r1; r += LWL) {
    const cpx_region_ref ref = a.refs[r];
    if (ref.width <= 0 || ref.height <= 0) continue;
    if (post && !ref.in_segment) continue;
    {
      // float32(thermal) - np.median(float32 frame): integers minus a half-integer, exact in float32
      const FrameInfo fi = a.info[ref.frame];
      tmn = fminf(tmn, __fsub_rn((float)fi.thermal_min, fi.thermal_median));
      tmx = fmaxf(tmx, __fsub_rn((float)fi.thermal_max, fi.thermal_median));
    }
    const int n = ref.width * ref.height;
    const float* F = a.filtered + (size_t)ref.frame * P;
    for (int k = lane; k < n; k += 64) {
      const int yy = k / ref.width;
      const float v = F[(ref.y + yy) * W + ref.x + (k - yy * ref.width)];
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
    if (ref.in_segment && clip0 && !post) {
      // np.median(float32(crop) - median) <= 0  <=>  a + b <= M, with a <= b the two middle order statistics of
      // the crop (a == b for an odd count) and M = 2 * median (an integer: the frame median is k/2).  One counting
      // pass decides it: with t = floor(M / 2) and c = #{v <= t},
      //   c >= k2 + 1        -> b <= t          -> a + b <= 2t <= M           -> true
      //   c <= k1            -> a >= t + 1      -> a + b >= 2t + 2 > M        -> false
      //   otherwise (even n) -> a <= t < b, a = max{v <= t}, b = min{v > t}   -> compare a + b with M
      // (all values are integers below 2^17, so the reference's float32 evaluation has the same sign)
      const uint16_t* T = a.frames + (size_t)ref.frame * P;
      const u32 k1 = (u32)((n - 1) >> 1), k2 = (u32)(n >> 1);
      const u32 M = (u32)(2.0f * a.info[ref.frame].thermal_median);
      const u32 tt = M >> 1;
      u32 c = 0, lowmax = 0, highmin = 0xFFFFFFFFu;
      for (int k = lane; k < n; k += 64) {
        const int yy = k / ref.width;
        const u32 v = T[(ref.y + yy) * W + ref.x + (k - yy * ref.width)];
        if (v <= tt) {
          ++c;
          lowmax = max(lowmax, v);
        } else {
          highmin = min(highmin, v);
        }
      }
      c = wsum(c);
      bool le;
      if (c >= k2 + 1) le = true;
      else if (c <= k1) le = false;
      else le = wmax(lowmax) + wmin(highmin) <= M;
      const float m2 = le ? 0.0f : 1.0f;
      if (m2 <= 0.0f) clip0 = 0;
    }
  }
  mn = wmin(mn);
  mx = wmax(mx);
  if (lane == 0) {
    s_mn[wave] = mn;
    s_mx[wave] = mx;
    s_tmn[wave] = tmn;   // (uniform across the wave: every lane read the same frame records)
    s_tmx[wave] = tmx;
    s_clip[wave] = clip0;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    cpx_track_limits o;
    float fmn = s_mn[0], fmx = s_mx[0];
    int c0 = s_clip[0];
    for (int w = 1; w < LWL; ++w) {
      fmn = fminf(fmn, s_mn[w]);
      fmx = fmaxf(fmx, s_mx[w]);
      c0 &= s_clip[w];
    }
    float ftmn = s_tmn[0], ftmx = s_tmx[0];
    for (int w = 1; w < LWL; ++w) {
      ftmn = fminf(ftmn, s_tmn[w]);
      ftmx = fmaxf(ftmx, s_tmx[w]);
    }
    o.filt_min = fmn;
    o.filt_max = fmx;
    // preprocess_frame's default clip_thermals_at_zero = True (preprocess.py:68): post_process_file and the
    // single-frame path (preprocess_frames, interpreter.py:297-306) do not pass the tested value
    o.clip_at_zero = (post || (a.limits_flags & CPX_LIMITS_ALWAYS_CLIP)) ? 1 : c0;
    o.flags = a.limits_flags;
    o.therm_min = ftmn;
    o.therm_max = ftmx;
    o.reserved[0] = o.reserved[1] = 0;
    a.limits[t] = o;
  }
}

// ---------------------------------------------------------------------------------------
// one tile per workgroup
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void lin_coord(int d, int dn, int sn, int* i0, int* i1, float* w) {
  // cv2.resize INTER_LINEAR source coordinate
  const double scale = (double)sn / (double)dn;
  const float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  float a = __fsub_rn(f, (float)s);
  if (s < 0) {
    a = 0.0f;
    s = 0;
  }
  if (s >= sn - 1) {
    a = 0.0f;
    s = sn - 1;
  }
  *i0 = s;
  *i1 = (s + 1 < sn) ? s + 1 : sn - 1;
  *w = a;
}

__global__ __launch_bounds__(LT) void cpx_crop_kernel(ClassifyArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_t = reinterpret_cast<float*>(smem);  // [fs*fs] thermal tile
  float* s_f = s_t + a.frame_size * a.frame_size;  // [fs*fs] filtered tile (diff_norm = False: normalised per tile)
  __shared__ float sc[LW];
  const cpx_crop_req q = a.reqs[blockIdx.x];
  const int fs = a.frame_size, sq = a.square_width;
  const int W = a.W, P = a.W * a.H;
  const uint16_t* T = a.frames + (size_t)q.frame * P;
  const float* F = a.filtered + (size_t)q.frame * P;
  const int rw = q.width, rh = q.height;
  const cpx_track_limits lim = a.limits[q.track];
  const float median = a.info[q.frame].thermal_median;
  // ---- resize_and_pad geometry (imageprocessing.py:24-60) ----
  const double sh = (double)fs / (double)rh, sw = (double)fs / (double)rw;
  const double scale = sh < sw ? sh : sw;
  int dw = (int)rint((double)rw * scale), dh = (int)rint((double)rh * scale);  // Python round(): half to even
  dw = dw < 1 ? 1 : (dw > fs ? fs : dw);
  dh = dh < 1 ? 1 : (dh > fs ? fs : dh);
  int ox = (fs - dw) / 2, oy = (fs - dh) / 2;
  const int cx = a.crop_x, cy = a.crop_y, cr = a.crop_x + a.crop_w, cb = a.crop_y + a.crop_h;
  if (q.x <= cx) ox = 0;                      // min(edge_offset[0] = 0, fs - dw)
  else if (q.x + rw >= cr) ox = fs - dw;      // max(fs - 0 - dw, 0)
  if (q.y <= cy) oy = 0;
  else if (q.y + rh >= cb) oy = fs - dh;
  // ---- pad value of the thermal channel: min of the crop (imageprocessing.py:38-39) ----
  float pmin = INFINITY;
  for (int k = threadIdx.x; k < rw * rh; k += LT) {
    const int yy = k / rw;
    pmin = fminf(pmin, (float)T[(q.y + yy) * W + q.x + (k - yy * rw)]);
  }
  pmin = block_min(pmin, sc);
  // post_process_file hands preprocess_frame a crop that already had the frame median subtracted (float32, exact:
  // integer pixels, half-integer median), with sub_median = False: the subtraction happens BEFORE the resize
  const bool pre = (lim.flags & CPX_LIMITS_POST_PROCESS) != 0;
  const bool tdn = (lim.flags & CPX_LIMITS_THERMAL_DIFF_NORM) != 0;   // thermal_norm_limits given: no clip at zero
  const bool own = (lim.flags & CPX_LIMITS_NO_DIFF_NORM) != 0;        // Frame.normalize(): both channels per tile
  const int cth = (lim.flags & CPX_LIMITS_SWAP_CHANNELS) ? 1 : 0, cfi = cth ^ 1;
  // the input scaling of the Keras 'tf'-mode families and inceptionv3 (interpreter.py:64-98,563-566: x /= 127.5; x -= 1.0
  // in float32, applied to the finished sample, preprocess.py:142-143,200-201)
  // A tiled sample is a float64 array at that point (square_clip pastes the float32 tiles into np.zeros,
  // imageprocessing.py:85-104) and is cast to float32 after the scaling; a single frame (square_width 1) is the
  // float32 stack of its channels and is scaled in float32.
  const bool tfs = (lim.flags & CPX_LIMITS_TF_SCALING) != 0;
  const bool tiled = sq > 1;
  auto fin = [tfs, tiled](float v) {
    if (!tfs) return v;
    return tiled ? (float)((double)v / 127.5 - 1.0) : __fsub_rn(__fdiv_rn(v, 127.5f), 1.0f);
  };
  if (pre) pmin = __fsub_rn(pmin, median);
  // ---- both channels: bilinear sample (two float32 passes), paste, per-pixel ops ----
  const int n = fs * fs;
  const int ty = q.tile / sq, tx = q.tile - ty * sq;
  const int OW = sq * fs;
  float* out = a.out + ((size_t)q.sample * OW + (size_t)ty * fs) * OW * 2 + (size_t)tx * fs * 2;
  float tmn = INFINITY, tmx = -INFINITY, fmn = INFINITY, fmx = -INFINITY;
  const float fspan = __fsub_rn(lim.filt_max, lim.filt_min);
  // the filtered channel's value waits in LDS for the thermal one (which needs the tile's own minimum and maximum): the two
  // channels of a pixel then leave in ONE 8-byte store -- written one channel per pass, every 32-byte sector of the sample was
  // written twice, half each time (the kernel was bound by exactly that: 319 MB in 0.9 ms)
  for (int k = threadIdx.x; k < n; k += LT) {
    const int yy = k / fs, xx = k - yy * fs;
    float tv = pmin, fv = 0.0f;
    const int ry = yy - oy, rx = xx - ox;
    if (ry >= 0 && ry < dh && rx >= 0 && rx < dw) {
      int x0, x1, y0, y1;
      float ax, ay;
      lin_coord(rx, dw, rw, &x0, &x1, &ax);
      lin_coord(ry, dh, rh, &y0, &y1, &ay);
      const float bx = __fsub_rn(1.0f, ax), by = __fsub_rn(1.0f, ay);
      const int r0 = (q.y + y0) * W + q.x, r1 = (q.y + y1) * W + q.x;
      float t00 = (float)T[r0 + x0], t01 = (float)T[r0 + x1], t10 = (float)T[r1 + x0], t11 = (float)T[r1 + x1];
      if (pre) {
        t00 = __fsub_rn(t00, median); t01 = __fsub_rn(t01, median);
        t10 = __fsub_rn(t10, median); t11 = __fsub_rn(t11, median);
      }
      const float th0 = __fadd_rn(__fmul_rn(t00, bx), __fmul_rn(t01, ax));
      const float th1 = __fadd_rn(__fmul_rn(t10, bx), __fmul_rn(t11, ax));
      tv = __fadd_rn(__fmul_rn(th0, by), __fmul_rn(th1, ay));
      const float f00 = F[r0 + x0], f01 = F[r0 + x1], f10 = F[r1 + x0], f11 = F[r1 + x1];
      const float fh0 = __fadd_rn(__fmul_rn(f00, bx), __fmul_rn(f01, ax));
      const float fh1 = __fadd_rn(__fmul_rn(f10, bx), __fmul_rn(f11, ax));
      fv = __fadd_rn(__fmul_rn(fh0, by), __fmul_rn(fh1, ay));
    }
    // thermal: -= median ; clip at 0 (preprocess.py:87-90)
    if (!pre) tv = __fsub_rn(tv, median);
    if (lim.clip_at_zero && !tdn && tv < 0.0f) tv = 0.0f;
    s_t[k] = tv;
    tmn = fminf(tmn, tv);
    tmx = fmaxf(tmx, tv);
    if (own) {
      s_f[k] = fv;
      fmn = fminf(fmn, fv);
      fmx = fmaxf(fmx, fv);
      continue;
    }
    // filtered: normalize(min, max of the track, new_max = 255) (preprocess.py:92-98)
    float fo;
    if (pre) {
      // post_process_file's limits are np.float64 scalars (np.min / np.max of a float64 crop), which promote
      // new_max * (float32(data) - min) / (max - min) to float64 (imageprocessing.py:168); one rounding at the end
      const double dmin = (double)lim.filt_min, dmax = (double)lim.filt_max;
      if (dmax == dmin) fo = (dmax == 0.0) ? 0.0f : (float)((double)fv / dmax);
      else fo = (float)(255.0 * ((double)fv - dmin) / (dmax - dmin));
    } else if (lim.filt_max == lim.filt_min) fo = (lim.filt_max == 0.0f) ? 0.0f : __fdiv_rn(fv, lim.filt_max);
    else fo = __fdiv_rn(__fmul_rn(255.0f, __fsub_rn(fv, lim.filt_min)), fspan);
    s_f[k] = fin(fo);
  }
  tmn = block_min(tmn, sc);
  tmx = block_max(tmx, sc);
  // thermal: normalize over the tile itself (preprocess.py:99-106), or with the track's thermal limits
  // (thermal_diff_norm: normalize(thermal, min, max, 255) with the np.float32 limits of get_limits) -- unless
  // diff_norm is off, where Frame.normalize() ignores them
  if (tdn && !own) {
    tmn = lim.therm_min;
    tmx = lim.therm_max;
  }
  const float tspan = __fsub_rn(tmx, tmn);
  if (own) {
    fmn = block_min(fmn, sc);
    fmx = block_max(fmx, sc);
  }
  const float ospan = __fsub_rn(fmx, fmn);
  for (int k = threadIdx.x; k < n; k += LT) {
    const int yy = k / fs, xx = k - yy * fs;
    const float tv = s_t[k];
    float to;
    if (tmx == tmn) to = (tmx == 0.0f) ? 0.0f : __fdiv_rn(tv, tmx);
    else to = __fdiv_rn(__fmul_rn(255.0f, __fsub_rn(tv, tmn)), tspan);
    float fo = s_f[k];  // (finished above unless the tile normalises it itself)
    if (own) {
      const float fv = fo;
      if (fmx == fmn) fo = (fmx == 0.0f) ? 0.0f : __fdiv_rn(fv, fmx);
      else fo = __fdiv_rn(__fmul_rn(255.0f, __fsub_rn(fv, fmn)), ospan);
      fo = fin(fo);
    }
    const float tf = fin(to);
    *reinterpret_cast<float2*>(out + ((size_t)yy * OW + xx) * 2) = cth == 0 ? make_float2(tf, fo) : make_float2(fo, tf);
  }
}

// class_best_score per track (trackprediction.py:127-171) + low-evidence cap (interpreter.py:151-168)
__global__ __launch_bounds__(64) void cpx_aggregate_kernel(AggregateArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n_tracks) return;
  // samples of track t: [s0, s1) in the sorted sample_track array
  int lo = 0, hi = a.n_samples;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a.sample_track[mid] < t) lo = mid + 1;
    else hi = mid;
  }
  const int s0 = lo;
  hi = a.n_samples;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a.sample_track[mid] <= t) lo = mid + 1;
    else hi = mid;
  }
  const int s1 = lo;
  const int L = a.n_labels;
  float* out = a.scores + (size_t)t * L;
  float total = 0.0f;
  // np.sum(predictions, axis=0) over at most a few segments: plain float32 accumulation in segment order
  for (int l = 0; l < L; ++l) {
    float s = 0.0f;
    for (int k = s0; k < s1; ++k) s += a.probs[(size_t)k * L + l];
    out[l] = s;
  }
  // np.sum(class_best_score): pairwise over L <= 128 labels collapses to NumPy's 8-lane scheme; L < 8 is a loop
  if (L < 8) {
    for (int l = 0; l < L; ++l) total += out[l];
  } else {
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = out[j];
    int i;
    for (i = 8; i < L - (L % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += out[i + j];
    total = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < L; ++i) total += out[i];
  }
  int best = 0;
  for (int l = 0; l < L; ++l) {
    out[l] = (s1 > s0) ? out[l] / total : 0.0f;
    if (out[l] > out[best]) best = l;
  }
  if (s1 - s0 == 1) {
    const int per = a.square_width * a.square_width;
    const cpx_crop_req* q = a.reqs + (size_t)s0 * per;
    int distinct = 0;
    for (int j = 0; j < per; ++j) distinct += (j == 0 || q[j].frame != q[j - 1].frame);  // frames are sorted
    if ((float)distinct < (float)per / 4.0f && best != a.fp_index) {
      float tot2 = 0.0f;
      for (int l = 0; l < L; ++l) tot2 += out[l];
      if (tot2 > 0.5f) {
        const float scale = 0.5f / tot2;
        for (int l = 0; l < L; ++l) out[l] *= scale;
      }
    }
  }
  a.best[t] = (s1 > s0) ? best : -1;
}

void launch_aggregate(const AggregateArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(cpx_aggregate_kernel, dim3((a.n_tracks + 63) / 64), dim3(64), 0, s, a);
}

void launch_limits(const ClassifyArgs& a, int n_tracks, hipStream_t s) {
  hipLaunchKernelGGL(cpx_limits_kernel, dim3(n_tracks), dim3(LTL), 0, s, a);
}
void launch_crop(const ClassifyArgs& a, int n_reqs, hipStream_t s) {
  hipLaunchKernelGGL(cpx_crop_kernel, dim3(n_reqs), dim3(LT), (size_t)2 * a.frame_size * a.frame_size * sizeof(float), s, a);
}

}  // namespace cpx
