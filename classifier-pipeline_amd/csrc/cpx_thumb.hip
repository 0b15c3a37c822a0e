// cpx_thumb.hip -- thumbnail stage (reference classify/thumbnail.py:13-188).
//   cpx_thumb_kernel     : one wavefront per (track, frame) region: the label mask inside the region goes to LDS,
//                          lane 0 runs the raster scan / Suzuki-Abe border following of the external contours and the
//                          Teh-Chin (TC89_L1) dominant-point passes OpenCV applies for CHAIN_APPROX_TC89_L1, keeping the
//                          largest point count; the whole wave then selects the median of the thermal values under the
//                          mask by bisection.  Integer work throughout: results are exact.
//   cpx_trackless_kernel : 64x64 box sums of the hottest frame and of its (uint16-wrapping) difference with the clip
//                          background, then the reference's sequential choice rule over the 56x96 positions.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cpx_kernels.h"

namespace cpx {

namespace {
typedef unsigned int u32;
typedef long long i64;

__device__ const signed char kDx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
__device__ const signed char kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
__device__ const signed char kAbsDiff[15] = {1, 2, 3, 4, 3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1};

struct TcArrays {
  unsigned char* px;
  unsigned char* py;
  signed char* s;
  short* k;
  short* nxt;  // -1 = end of list
};

// Teh-Chin approximation of a closed chain (OpenCV icvApproximateChainTC89, method TC89_L1): number of points kept.
__device__ int tc89_count(const signed char* chain, int n, int ox, int oy, TcArrays A) {
  if (n == 0) return 1;
  const int len = n;
  int head = -1, tail = -1;
  {
    int x = ox, y = oy;
    for (int i = 0; i < n; ++i) {
      const int prev_code = chain[i == 0 ? n - 1 : i - 1];
      const int code = chain[i];
      const int sv = kAbsDiff[code - prev_code + 7];
      A.px[i] = (unsigned char)x;
      A.py[i] = (unsigned char)y;
      A.s[i] = (signed char)sv;
      A.nxt[i] = -1;
      if (sv != 0) {
        if (tail < 0) head = i; else A.nxt[tail] = (short)i;
        tail = i;
      }
      x += kDx[code];
      y += kDy[code];
    }
  }
  if (head < 0) return 0;  // cannot happen for a closed chain
  // Pass 1: support regions
  for (int cur = head; cur >= 0; cur = A.nxt[cur]) {
    const int i = cur;
    const int x0 = A.px[i], y0 = A.py[i];
    int l = 0, d_num = 0, kk = 1;
    for (;; ++kk) {
      int i1 = i - kk;
      i1 += i1 < 0 ? len : 0;
      int i2 = i + kk;
      i2 -= i2 >= len ? len : 0;
      const int dx = (int)A.px[i2] - (int)A.px[i1], dy = (int)A.py[i2] - (int)A.py[i1];
      const int lk = dx * dx + dy * dy;
      const int dk_num = (x0 - (int)A.px[i1]) * dy - (y0 - (int)A.py[i1]) * dx;
      const i64 d = (i64)d_num * lk - (i64)dk_num * l;
      if (kk > 1 && (l >= lk || (d_num > 0 && d <= 0) || (d_num < 0 && d >= 0))) break;
      d_num = dk_num;
      l = lk;
      if (kk >= len) { ++kk; break; }  // guard (OpenCV asserts k <= len)
    }
    A.k[cur] = (short)(kk - 1);
  }
  // Pass 2: non-maxima suppression
  {
    int prev = -1;
    for (int cur = head; cur >= 0;) {
      const int k2 = A.k[cur] >> 1, sv = A.s[cur], i = cur;
      int j = 1;
      for (; j <= k2; ++j) {
        int i2 = i - j;
        i2 += i2 < 0 ? len : 0;
        if (A.s[i2] > sv) break;
        i2 = i + j;
        i2 -= i2 >= len ? len : 0;
        if (A.s[i2] > sv) break;
      }
      const int nx = A.nxt[cur];
      if (j <= k2) {
        if (prev < 0) head = nx; else A.nxt[prev] = (short)nx;
        A.s[cur] = 0;
      } else {
        prev = cur;
      }
      cur = nx;
    }
  }
  // Pass 3: non-dominant points with a 1-length support region
  {
    int prev = -1;
    for (int cur = head; cur >= 0;) {
      const int nx = A.nxt[cur];
      bool removed = false;
      if (A.k[cur] == 1) {
        const int sv = A.s[cur], i = cur;
        int i1 = i - 1;
        i1 += i1 < 0 ? len : 0;
        int i2 = i + 1;
        i2 -= i2 >= len ? len : 0;
        if (sv <= A.s[i1] || sv <= A.s[i2]) {
          if (prev < 0) head = nx; else A.nxt[prev] = (short)nx;
          A.s[cur] = 0;
          removed = true;
        }
      }
      if (!removed) prev = cur;
      cur = nx;
    }
  }
  if (head < 0) return 0;
  // Pass 4: clean the remaining couples of neighbouring points
  bool all_survived = false;
  if (A.s[0] != 0 && A.s[len - 1] != 0) {
    int i1 = 1;
    for (; i1 < len && A.s[i1] != 0; ++i1) A.s[i1 - 1] = 0;
    if (i1 == len) {
      all_survived = true;
    } else {
      --i1;
      int i2 = len - 2;
      for (; i2 > 0 && A.s[i2] != 0; --i2) {
        A.nxt[i2] = -1;
        A.s[i2 + 1] = 0;
      }
      ++i2;
      if (i1 == 0 && i2 == len - 1) {  // only two points
        i1 = A.nxt[0];
        A.px[len] = A.px[0];
        A.py[len] = A.py[0];
        A.s[len] = A.s[0];
        A.k[len] = A.k[0];
        A.nxt[len] = -1;
        A.nxt[len - 1] = (short)len;
      }
      head = i1;
    }
  }
  if (!all_survived) {
    // `first` / `prev` = -2 stand for the list head (whose s and k read as 0)
    int cur = head, first = -2, prev = -2, count = 1;
    while (cur >= 0) {
      const int nx = A.nxt[cur];
      if (nx < 0 || nx - cur != 1) {
        if (count >= 2) {
          if (count == 2) {
            const int s1 = prev == -2 ? 0 : A.s[prev], s2 = A.s[cur];
            const int k1 = prev == -2 ? 0 : A.k[prev];
            if (s1 > s2 || (s1 == s2 && k1 <= A.k[cur])) {
              if (prev == -2) head = nx; else A.nxt[prev] = (short)nx;  // remove the second
            } else {
              if (first == -2) head = cur; else A.nxt[first] = (short)cur;  // remove the first
            }
          } else {
            const int fn = first == -2 ? head : A.nxt[first];
            A.nxt[fn] = (short)cur;
          }
        }
        first = cur;
        count = 1;
      } else {
        ++count;
      }
      prev = cur;
      cur = A.nxt[cur];
    }
  }
  int cnt = 0;
  for (int cur = head; cur >= 0; cur = A.nxt[cur]) ++cnt;
  return cnt;
}
}  // namespace

__global__ __launch_bounds__(64) void cpx_thumb_kernel(ThumbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int r = blockIdx.x;
  const cpx_region_ref ref = a.refs[r];
  const int lane = threadIdx.x;
  const int w = ref.width, h = ref.height;
  const int W = a.W, P = a.W * a.H;
  cpx_thumb_stat out;
  out.contours = 0;
  out.status = 0;
  out.median_diff = 0.0;
  if (w <= 0 || h <= 0) {
    if (lane == 0) a.out[r] = out;
    return;
  }
  if (w > a.max_w || h > a.max_h) {   // this launch's LDS was sized for smaller regions: the caller runs it again
    out.status = CPX_ERR_OVERFLOW;
    if (lane == 0) a.out[r] = out;
    return;
  }
  const int PW = w + 2, PH = h + 2;
  signed char* img = (signed char*)s_raw;                       // [PH][PW], zero border
  const int img_bytes = (PH * PW + 15) & ~15;
  const int cap = a.chain_cap;
  signed char* chain = (signed char*)(s_raw + img_bytes);       // [cap]
  TcArrays A;
  A.px = (unsigned char*)(chain + cap);                         // [cap+1] each
  A.py = A.px + (cap + 16);
  A.s = (signed char*)(A.py + (cap + 16));
  A.k = (short*)(A.s + (cap + 16));
  A.nxt = A.k + (cap + 16);
  const int32_t* lab = a.labels + (size_t)ref.frame * P;
  const uint16_t* th = a.frames + (size_t)ref.frame * P;
  for (int i = lane; i < PH * PW; i += 64) {
    const int yy = i / PW - 1, xx = i % PW - 1;
    signed char v = 0;
    // np.uint8(labels) != 0 (thumbnail.py:91): a label that is a multiple of 256 would read as background
    if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = ((lab[(ref.y + yy) * W + ref.x + xx] & 0xFF) != 0) ? 1 : 0;
    img[i] = v;
  }
  __syncthreads();
  // ---- masked median of the thermal values (whole wave) -- before lane 0 rewrites the mask with border marks,
  // which keep non-zero pixels non-zero, so the order does not matter; done first to keep the wave convergent ----
  int n_mask = 0;
  for (int i = lane; i < h * w; i += 64) n_mask += img[(i / w + 1) * PW + (i % w) + 1] != 0;
  for (int o = 32; o > 0; o >>= 1) n_mask += __shfl_xor(n_mask, o);
  if (n_mask == 0) {
    if (lane == 0) a.out[r] = out;
    return;
  }
  auto count_le = [&](int v) {  // masked values <= v
    int c = 0;
    for (int i = lane; i < h * w; i += 64) {
      const int yy = i / w, xx = i - yy * w;
      if (img[(yy + 1) * PW + xx + 1] != 0) c += (int)th[(ref.y + yy) * W + ref.x + xx] <= v;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    return c;
  };
  auto kth = [&](int k) {  // k-th smallest (0-based): smallest v with count_le(v) >= k + 1
    int lo = 0, hi = 65535;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (count_le(mid) >= k + 1) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  const int v_hi = kth(n_mask >> 1);
  int v_lo = v_hi;
  if ((n_mask & 1) == 0) v_lo = kth((n_mask >> 1) - 1);
  out.median_diff = 0.5 * ((double)v_lo + (double)v_hi) - (double)a.info[ref.frame].thermal_median;
  // ---- external contours: raster scan + border following + TC89_L1 (lane 0) ----
  if (lane == 0) {
    int best = 0;
    for (int y = 1; y <= h && out.status == 0; ++y) {
      int prev = 0, lnbd_x = 0;
      signed char* row = img + y * PW;
      for (int x = 1; x <= w; ++x) {
        int p = row[x];
        if (p == prev) continue;
        if (prev == 0 && p == 1 && !(row[lnbd_x] > 0)) {
          // ---- follow the outer border that starts at (x, y) ----
          int n = 0;
          bool overflow = false;
          int s_end = 4, s = 4;
          signed char* i0 = row + x;
          signed char* i1;
          do {
            s = (s - 1) & 7;
            i1 = i0 + kDy[s] * PW + kDx[s];
          } while (*i1 == 0 && s != s_end);
          if (s == s_end) {
            *i0 = (signed char)-126;  // single pixel
          } else {
            signed char* i3 = i0;
            signed char* i4 = i0;
            for (;;) {
              s_end = s;
              while (s < 15) {
                ++s;
                i4 = i3 + kDy[s & 7] * PW + kDx[s & 7];
                if (*i4 != 0) break;
              }
              s &= 7;
              if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)-126;
              else if (*i3 == 1) *i3 = 2;
              if (n < cap) chain[n] = (signed char)s; else overflow = true;
              ++n;
              if (i4 == i0 && i3 == i1) break;
              i3 = i4;
              s = (s + 4) & 7;
            }
          }
          if (overflow) {
            out.status = CPX_ERR_OVERFLOW;
            break;
          }
          const int pts = tc89_count(chain, n, x - 1, y - 1, A);
          best = pts > best ? pts : best;
          p = row[x];
        }
        prev = p;
        if (prev & -2) lnbd_x = x;
      }
    }
    out.contours = best;
    a.out[r] = out;
  }
}

size_t thumb_lds_bytes(int W, int H, int cap) {
  const size_t img = ((size_t)(W + 2) * (H + 2) + 15) & ~(size_t)15;
  return img + cap + (size_t)(cap + 16) * (1 + 1 + 1 + 2 + 2) + 64;
}

int launch_thumb(const ThumbArgs& a, int n_refs, hipStream_t s) {
  const size_t lds = thumb_lds_bytes(a.max_w, a.max_h, a.chain_cap);
  if (lds > 160 * 1024) return -2;
  static bool lds_ready[64];
  if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(cpx_thumb_kernel), lds_ready, 160 * 1024 - 1024)) return -1;
  hipLaunchKernelGGL(cpx_thumb_kernel, dim3(n_refs), dim3(64), lds, s, a);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// best_trackless_thumb (thumbnail.py:26-64): one workgroup
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int TT = 256;
constexpr int TS = 64;  // THUMBNAIL_SIZE
}  // namespace

__global__ __launch_bounds__(TT) void cpx_trackless_kernel(TracklessArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  const int W = a.W, H = a.H, P = W * H;
  const int NX = W - TS + 1, BX = W - TS, BY = H - TS;  // row sums per row; box positions (range(H-64), range(W-64))
  u32* hs = (u32*)s_raw;                 // [H][NX]
  u32* box_t = hs + H * NX;              // [BY][BX] thermal
  u32* box_f = box_t + BY * BX;          // [BY][BX] filtered
  // one workgroup per (frame, background) pair: the single pair of the arguments, or row blockIdx.x of a.pairs
  const int frame = a.pairs ? a.pairs[2 * blockIdx.x] : a.frame;
  const int background = a.pairs ? a.pairs[2 * blockIdx.x + 1] : a.background;
  int32_t* out = a.out + 2 * (size_t)blockIdx.x;
  const uint16_t* fr = a.frames + (size_t)frame * P;
  const uint16_t* bg = a.frames + (size_t)background * P;
  for (int plane = 0; plane < 2; ++plane) {
    __syncthreads();
    for (int i = threadIdx.x; i < H * NX; i += TT) {
      const int y = i / NX, x = i - y * NX;
      u32 acc = 0;
      for (int j = 0; j < TS; ++j) {
        const u32 t = fr[y * W + x + j];
        acc += plane == 0 ? t : ((t - (u32)bg[y * W + x + j]) & 0xFFFFu);  // uint16 arithmetic wraps
      }
      hs[i] = acc;
    }
    __syncthreads();
    u32* box = plane == 0 ? box_t : box_f;
    for (int i = threadIdx.x; i < BY * BX; i += TT) {
      const int y = i / BX, x = i - y * BX;
      u32 acc = 0;
      for (int j = 0; j < TS; ++j) acc += hs[(y + j) * NX + x];
      box[i] = acc;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // means are sums / 4096: compare the integer sums.  The reference swaps the two stored values on update.
    int bx = 0, by = 0;
    u32 v1 = 0, v2 = 0;
    bool have = false;
    for (int y = 0; y < BY; ++y)
      for (int x = 0; x < BX; ++x) {
        const u32 ts = box_t[y * BX + x], fs = box_f[y * BX + x];
        if (!have) {
          have = true;
          bx = x; by = y; v1 = fs; v2 = ts;
        } else if (v1 > 0) {
          if (v1 < fs) { bx = x; by = y; v1 = ts; v2 = fs; }
        } else if (v2 < ts) {
          bx = x; by = y; v1 = ts; v2 = fs;
        }
      }
    out[0] = have ? bx : -1;
    out[1] = have ? by : -1;
  }
}

int launch_trackless(const TracklessArgs& a, hipStream_t s) {
  if (a.W <= TS || a.H <= TS) return -2;
  const size_t lds = ((size_t)a.H * (a.W - TS + 1) + 2 * (size_t)(a.H - TS) * (a.W - TS)) * 4;
  if (lds > 160 * 1024 - 1024) return -2;
  static bool lds_ready[64];
  if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(cpx_trackless_kernel), lds_ready, 160 * 1024 - 1024)) return -1;
  hipLaunchKernelGGL(cpx_trackless_kernel, dim3(a.pairs ? a.n : 1), dim3(TT), lds, s, a);
  return 0;
}

}  // namespace cpx
