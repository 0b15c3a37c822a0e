// cpx_assoc_core.h -- region filtering, track association and the Kalman
// tracker for ONE clip, as plain scalar code that runs as one GPU lane per clip
// (cpx_assoc.hip).  It is a header so that tests can also compile it for the
// host to debug it against the oracle without a GPU; the product only ever
// runs the HIP build.
//
// Follows the reference (paths relative to /root/reference/src):
//   track/cliptracker.py:263-365   _get_regions_of_interest (thermal branch)
//   track/cliptracker.py:124-247   _apply_region_matchings and helpers
//   track/track.py:107-326         RegionTracker.match / add_region / add_blank_frame / gates
//   track/track.py:646-735         Track.add_region / update_velocity / average_area / average_mass
//   track/kalman.py:5-26           cv2.KalmanFilter(4, 2), float32
//   ml_tools/rectangle.py:85-146   overlap_area / crop / enlarge
//   track/region.py:154-209        set_is_along_border / average_distance
// NumPy scalar typing is reproduced where it changes arithmetic: velocities and
// Kalman predictions are float32 when both operands were float32 (two
// consecutive Kalman-predicted blank regions), float64 otherwise.
#pragma once
#include <math.h>
#include <stdint.h>

#include "cpx.h"

#ifndef CPX_HD
#if defined(__HIPCC__)
#define CPX_HD __host__ __device__
#else
#define CPX_HD
#endif
#endif

namespace cpx {

enum { RF_BLANK = 1, RF_CROPPED = 2, RF_BORDER = 4, RF_CENTROID_F32 = 8, RF_W_PY = 16, RF_H_PY = 32 };
enum { VK_INT = 0, VK_F64 = 1, VK_F32 = 2 };

typedef cpx_region RegionRec;
typedef cpx_track_params AssocParams;
typedef cpx_track_record TrackRec;

struct ActiveTrack {
  int id, start_frame, n;
  int rt_frames, blank_frames, since_seen, tracking;
  int slot;
  float kx[4];
  float kP[16];
  float pm[2];  // predicted_mid
  double vx, vy;
  int vel_kind;
  int matched;  // scratch: 1 matched this frame, 2 created this frame
};

struct ScoreRec {
  double score;
  double key;
  int track;   // index into active[]
  int region;  // index into regs[]
};

// ---- rectangle helpers ---------------------------------------------------------
CPX_HD inline int imin(int a, int b) { return a < b ? a : b; }
CPX_HD inline int imax(int a, int b) { return a > b ? a : b; }

// Rectangle.crop (rectangle.py:91-96; the left/top setters keep right/bottom)
// The reference's coordinates are a mix of np.int32 (component statistics) and Python ints (crop rectangle, int()
// results).  Which of the two a width / height is decides whether `predicted_mid - width / 2.0` of a Kalman blank region
// is float64 arithmetic (np.int32 / 2.0 -> np.float64) or float32 (a Python float is a weak scalar), so the crop keeps
// track of it: *_py = "is a Python int in the reference".  Python's max(a, b) is b only if b > a, min(a, b) is b only
// if b < a; the bounds are Python ints.
CPX_HD inline void crop_axis(int& pos, bool& pos_py, int& ext, bool& ext_py, int lo, int size) {
  const int hi = lo + size;
  // left / top = min(bounds.hi, max(self.pos, bounds.lo)); the setter keeps right / bottom
  int m = pos;
  bool m_py = pos_py;
  if (lo > pos) {
    m = lo;
    m_py = true;
  }
  int npos = hi;
  bool npos_py = true;
  if (m < hi) {
    npos = m;
    npos_py = m_py;
  }
  const int old_far = pos + ext;
  const bool old_far_py = pos_py && ext_py;
  pos = npos;
  pos_py = npos_py;
  ext = old_far - pos;
  ext_py = old_far_py && pos_py;
  // right / bottom = max(bounds.lo, min(self.far, bounds.hi))
  const int far = pos + ext;
  const bool far_py = pos_py && ext_py;
  int mm = far;
  bool mm_py = far_py;
  if (hi < far) {
    mm = hi;
    mm_py = true;
  }
  int v = lo;
  bool v_py = true;
  if (mm > lo) {
    v = mm;
    v_py = mm_py;
  }
  ext = v - pos;
  ext_py = v_py && pos_py;
}

// Rectangle.crop (rectangle.py:91-96; the left/top setters keep right/bottom)
CPX_HD inline void rect_crop(int& x, int& y, int& w, int& h, bool& x_py, bool& y_py, bool& w_py, bool& h_py, int bx,
                             int by, int bw, int bh) {
  crop_axis(x, x_py, w, w_py, bx, bw);
  crop_axis(y, y_py, h, h_py, by, bh);
}

CPX_HD inline int overlap_area(const RegionRec& a, const RegionRec& b) {
  const int xo = imax(0, imin(a.x + a.width, b.x + b.width) - imax(a.x, b.x));
  const int yo = imax(0, imin(a.y + a.height, b.y + b.height) - imax(a.y, b.y));
  return xo * yo;
}

// ---- cv2.KalmanFilter(4,2) in float32 (kalman.py; OpenCV gemm semantics: every
// product is accumulated in double and rounded to float once) --------------------------
CPX_HD inline void kalman_init(ActiveTrack& t) {
  for (int i = 0; i < 4; ++i) t.kx[i] = 0.f;
  for (int i = 0; i < 16; ++i) t.kP[i] = 0.f;
}

// statePre = A x ; errorCovPre = A P A^T + Q ; post := pre.  A = [[1,0,1,0],[0,1,0,1],[0,0,1,0],[0,0,0,1]], Q = 0.03 I
CPX_HD inline void kalman_predict(ActiveTrack& t) {
  const float q = 0.03f;  // np.eye(4, dtype=float32) * 0.03 -> float32(0.03)
  static const int A[4][4] = {{1, 0, 1, 0}, {0, 1, 0, 1}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  float x[4], t1[16], P[16];
  for (int i = 0; i < 4; ++i) {
    double s = 0.0;
    for (int k = 0; k < 4; ++k) s += (double)A[i][k] * (double)t.kx[k];
    x[i] = (float)s;
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += (double)A[i][k] * (double)t.kP[k * 4 + j];
      t1[i * 4 + j] = (float)s;
    }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += (double)t1[i * 4 + k] * (double)A[j][k];
      s += (i == j) ? (double)q : 0.0;
      P[i * 4 + j] = (float)s;
    }
  for (int i = 0; i < 4; ++i) t.kx[i] = x[i];
  for (int i = 0; i < 16; ++i) t.kP[i] = P[i];
  t.pm[0] = x[0];
  t.pm[1] = x[1];
}

// correct(z): H = eye(2,4), R = I
CPX_HD inline void kalman_correct(ActiveTrack& t, float z0, float z1) {
  float t2[8];  // H * P  (2x4) = first two rows of P (exact)
  for (int j = 0; j < 4; ++j) {
    t2[j] = t.kP[j];
    t2[4 + j] = t.kP[4 + j];
  }
  // temp3 = temp2 * H^T + R  (2x2)
  const float a = (float)((double)t2[0] + 1.0), b = t2[1];
  const float c = t2[4], d = (float)((double)t2[5] + 1.0);
  // temp4 = solve(temp3, temp2): 2x2 Cramer in double, rounded to float (oracle definition)
  const double det = (double)a * (double)d - (double)b * (double)c;
  float t4[8];
  for (int j = 0; j < 4; ++j) {
    const double r0 = (double)t2[j], r1 = (double)t2[4 + j];
    t4[j] = (float)(((double)d * r0 - (double)b * r1) / det);
    t4[4 + j] = (float)(((double)a * r1 - (double)c * r0) / det);
  }
  // gain = temp4^T (4x2); temp5 = z - H x
  const float e0 = (float)((double)z0 - (double)t.kx[0]);
  const float e1 = (float)((double)z1 - (double)t.kx[1]);
  float x[4], P[16];
  for (int i = 0; i < 4; ++i) {
    const double s = (double)t4[i] * (double)e0 + (double)t4[4 + i] * (double)e1;
    x[i] = (float)((double)t.kx[i] + s);
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const double s = (double)t4[i] * (double)t2[j] + (double)t4[4 + i] * (double)t2[4 + j];
      P[i * 4 + j] = (float)((double)t.kP[i * 4 + j] - s);
    }
  for (int i = 0; i < 4; ++i) t.kx[i] = x[i];
  for (int i = 0; i < 16; ++i) t.kP[i] = P[i];
}

// ---- per-clip working set ----------------------------------------------------------------
struct AssocClip {
  const AssocParams* p;
  int cap;        // regions per frame
  int max_active; // slots
  int max_tracks;
  RegionRec* pool;      // [n_frames][max_active]
  ActiveTrack* active;  // [max_active]
  int n_active;
  TrackRec* tracks;     // [max_tracks]
  int n_tracks;
  int next_id;
  RegionRec* regs;      // [cap] regions of the current frame
  ScoreRec* scores;     // [max_active * cap]
  unsigned char* used;  // [cap]
  int status;
};

CPX_HD inline RegionRec& pool_at(AssocClip& c, const ActiveTrack& t, int i) {
  return c.pool[(size_t)(t.start_frame + i) * c.max_active + t.slot];
}

// RegionTracker.add_region (track.py:194-214) + Track.add_region bookkeeping (:646-669)
CPX_HD inline void track_add_region(AssocClip& c, ActiveTrack& t, const RegionRec& r) {
  t.rt_frames += 1;
  if (r.flags & RF_BLANK) {
    t.blank_frames += 1;
    t.since_seen += 1;
    const int stop = imin(2 * (t.rt_frames - t.since_seen), c.p->max_blanks);
    t.tracking = t.since_seen < stop;
  } else {
    t.tracking = 1;
    kalman_correct(t, (float)r.cx, (float)r.cy);
    t.since_seen = 0;
  }
  kalman_predict(t);
  pool_at(c, t, t.n) = r;
  t.n += 1;
  // update_velocity (track.py:657-669)
  if (t.n >= 2) {
    const RegionRec& cur = pool_at(c, t, t.n - 1);
    const RegionRec& prv = pool_at(c, t, t.n - 2);
    if ((cur.flags & RF_CENTROID_F32) && (prv.flags & RF_CENTROID_F32)) {
      t.vx = (double)((float)cur.cx - (float)prv.cx);
      t.vy = (double)((float)cur.cy - (float)prv.cy);
      t.vel_kind = VK_F32;
    } else {
      t.vx = cur.cx - prv.cx;
      t.vy = cur.cy - prv.cy;
      t.vel_kind = VK_F64;
    }
  } else {
    t.vx = 0.0;
    t.vy = 0.0;
    t.vel_kind = VK_INT;
  }
}

// RegionTracker.add_blank_frame (track.py:239-264)
CPX_HD inline void track_add_blank(AssocClip& c, ActiveTrack& t) {
  const RegionRec last = pool_at(c, t, t.n - 1);
  const int kalman_amount = t.rt_frames - 18 - t.since_seen * 2;
  RegionRec r;
  if (kalman_amount > 0) {
    // int(predicted_mid - last_bound.width / 2.0): float32 when the width is a Python int in the reference, float64
    // when it is np.int32 (track.py:247-253)
    bool w_py = (last.flags & RF_W_PY) != 0, h_py = (last.flags & RF_H_PY) != 0;
    r.x = w_py ? (int)(t.pm[0] - (float)(last.width / 2.0)) : (int)((double)t.pm[0] - last.width / 2.0);
    r.y = h_py ? (int)(t.pm[1] - (float)(last.height / 2.0)) : (int)((double)t.pm[1] - last.height / 2.0);
    r.width = last.width;
    r.height = last.height;
    r.cx = (double)t.pm[0];
    r.cy = (double)t.pm[1];
    r.id = 0;
    bool x_py = true, y_py = true;
    rect_crop(r.x, r.y, r.width, r.height, x_py, y_py, w_py, h_py, c.p->crop_x, c.p->crop_y, c.p->crop_w,
              c.p->crop_h);
    r.flags = RF_CENTROID_F32 | (w_py ? RF_W_PY : 0) | (h_py ? RF_H_PY : 0);
  } else {
    r = last;
  }
  r.flags |= RF_BLANK;
  r.mass = 0;
  r.pixel_variance = 0.f;
  r.frame_number = last.frame_number + 1;
  r.pad = 0;
  track_add_region(c, t, r);
}

CPX_HD inline void avg_last5(AssocClip& c, const ActiveTrack& t, double& avg_mass, double& avg_area) {
  long long m = 0, a = 0;
  int cnt = 0;
  for (int i = t.n - 1; i >= 0; --i) {
    const RegionRec& b = pool_at(c, t, i);
    if (!(b.flags & RF_BLANK)) {
      m += b.mass;
      a += (long long)b.width * b.height;
      cnt += 1;
    }
    if (cnt == 5) break;
  }
  avg_mass = cnt ? (double)m / cnt : 0.0;
  avg_area = cnt ? (double)a / cnt : 0.0;
}

// |vx| + |vy| in the dtype NumPy would use for np.sum(np.abs(velocity))
CPX_HD inline double vel_abs_sum(const ActiveTrack& t) {
  if (t.vel_kind == VK_F32) return (double)(fabsf((float)t.vx) + fabsf((float)t.vy));
  return fabs(t.vx) + fabs(t.vy);
}

// RegionTracker.match for one (track, region) pair; returns false if a gate rejects it
CPX_HD inline bool match_pair(AssocClip& c, const ActiveTrack& t, const RegionRec& last, const RegionRec& r,
                              double avg_mass, double avg_area, double max_distance, double* score) {
  const AssocParams& p = *c.p;
  const double area = (double)((long long)r.width * r.height);
  const double size_change = fabs(area - avg_area) / (avg_area + 50.0);
  const long long dx0 = r.x - last.x, dy0 = r.y - last.y;
  const long long dx2 = (r.x + r.width) - (last.x + last.width), dy2 = (r.y + r.height) - (last.y + last.height);
  const double distance = (double)((dx0 * dx0 + dy0 * dy0) + (dx2 * dx2 + dy2 * dy2)) / 2.0;
  // get_max_size_change (track.py:312-326)
  const bool rb = (r.flags & RF_BORDER) != 0, lb = (last.flags & RF_BORDER) != 0;
  const bool exiting = rb && !lb;
  const bool entering = !exiting && lb;
  double max_size = 1.5;
  if (t.n < 5) max_size = 2.0;
  const double vel = vel_abs_sum(t);
  if (entering || exiting) {
    max_size = 2.0;
    if (vel > 10.0) max_size *= 3.0;
  } else if (vel > 10.0) {
    max_size *= 2.0;
  }
  // get_max_mass_change_percent (track.py:295-309)
  if (p.has_mass_change_percent && (double)t.n > p.restrict_mass_after * (double)p.fps) {
    double pct = p.mass_change_percent;
    if (vel > 5.0) pct = pct + 0.1;
    double max_mass = avg_mass * pct;
    if (p.has_min_mass_change && p.min_mass_change >= max_mass) max_mass = p.min_mass_change;  // max(min_mass_change, .)
    if (max_mass != 0.0 && fabs(avg_mass - (double)r.mass) > max_mass) return false;
  }
  if (distance > max_distance) return false;
  if (size_change > max_size) return false;
  *score = distance;
  return true;
}

// get_max_distance_change (track.py:269-293) -> the single distance gate that is used
CPX_HD inline double max_distance_for(AssocClip& c, const ActiveTrack& t, const RegionRec& last) {
  const AssocParams& p = *c.p;
  // velocity_distance, typed like NumPy: int when the track has one region, float32 for two
  // consecutive Kalman centroids, float64 otherwise
  double vel_d;
  int kind;
  if (t.n == 1) {
    const double x = p.velocity_multiplier * p.base_velocity;
    vel_d = x * x + x * x;
    kind = VK_F64;  // python numbers: exact in double
  } else if (t.vel_kind == VK_F32) {
    const float x = (float)p.velocity_multiplier * (float)t.vx, y = (float)p.velocity_multiplier * (float)t.vy;
    const float xx = x * x, yy = y * y;
    vel_d = (double)(xx + yy);
    kind = VK_F32;
  } else {
    const double x = p.velocity_multiplier * t.vx, y = p.velocity_multiplier * t.vy;
    vel_d = x * x + y * y;
    kind = VK_F64;
  }
  // predicted_velocity (track.py:228-237)
  double pred_d = 0.0;
  int pkind = VK_F64;
  if (t.rt_frames - t.blank_frames > 18) {
    if (last.flags & RF_CENTROID_F32) {
      const float px = t.pm[0] - (float)last.cx, py = t.pm[1] - (float)last.cy;
      const float xx = px * px, yy = py * py;
      pred_d = (double)(xx + yy);
      pkind = VK_F32;
    } else {
      const double px = (double)t.pm[0] - last.cx, py = (double)t.pm[1] - last.cy;
      pred_d = px * px + py * py;
      pkind = VK_F64;
    }
  }
  // pred_distance = max(velocity_distance, pred_distance); max_distance = base + max(velocity_distance, pred_distance)
  double m = vel_d;
  int mk = kind;
  if (pred_d > vel_d) {
    m = pred_d;
    mk = pkind;
  }
  if (mk == VK_F32) return (double)((float)p.base_distance_change + (float)m);
  return p.base_distance_change + m;
}

// _get_regions_of_interest for one frame: components -> regs[0..n)
CPX_HD inline int build_regions(AssocClip& c, const cpx_component* comps, int ncomp, int frame_number, bool has_prev) {
  const AssocParams& p = *c.p;
  int n = 0;
  const int padding = imax(3, p.frame_padding);
  const int edge = (int)ceil(p.crop_w * 0.03);
  for (int i = 0; i < ncomp; ++i) {
    const cpx_component& k = comps[i];
    RegionRec r;
    r.x = k.x;
    r.y = k.y;
    r.width = k.width;
    r.height = k.height;
    r.mass = k.area;
    r.id = i;
    r.frame_number = frame_number;
    r.cx = (double)k.sum_x / (double)k.area;
    r.cy = (double)k.sum_y / (double)k.area;
    r.pixel_variance = has_prev ? k.pixel_variance : 0.f;
    r.flags = 0;
    r.pad = 0;
    if (r.width < p.min_dimension || r.height < p.min_dimension) continue;
    const int ox = r.x, oy = r.y, ow = r.width, oh = r.height;
    bool x_py = false, y_py = false, w_py = false, h_py = false;  // component statistics are np.int32
    rect_crop(r.x, r.y, r.width, r.height, x_py, y_py, w_py, h_py, p.crop_x, p.crop_y, p.crop_w, p.crop_h);
    const bool cropped = (ox != r.x) || (oy != r.y) || (ow != r.width) || (oh != r.height);
    if (cropped) r.flags |= RF_CROPPED;
    if (p.cropped_regions_strategy == 0) {  // cautious
      if ((double)(ow - r.width) / (double)ow > 0.25 || (double)(oh - r.height) / (double)oh > 0.25) continue;
    } else if (p.cropped_regions_strategy == 1) {  // none
      if (cropped) continue;
    }
    if (p.filter_regions_pre_match && ((double)r.pixel_variance < p.aoi_pixel_variance && (double)r.mass < p.aoi_min_mass))
      continue;
    // enlarge(padding, max=crop_rectangle)
    r.x -= padding;
    r.width += 2 * padding;
    r.y -= padding;
    r.height += 2 * padding;
    rect_crop(r.x, r.y, r.width, r.height, x_py, y_py, w_py, h_py, p.crop_x, p.crop_y, p.crop_w, p.crop_h);
    if (w_py) r.flags |= RF_W_PY;
    if (h_py) r.flags |= RF_H_PY;
    // set_is_along_border(bounds = crop_rectangle, edge) -- note: bounds.width / bounds.height, not right / bottom
    if (cropped || r.x <= p.crop_x + edge || r.y <= p.crop_y + edge || r.x + r.width >= p.crop_w - edge ||
        r.y + r.height >= p.crop_h - edge)
      r.flags |= RF_BORDER;
    c.regs[n++] = r;
  }
  return n;
}

// float(".{id}") -- the decimal fraction the reference uses as a tie-breaker (cliptracker.py:147-150)
CPX_HD inline double dot_id(int id) {
  double den = 1.0;
  int v = id;
  do {
    den *= 10.0;
    v /= 10;
  } while (v > 0);
  return (double)id / den;
}

// one processed frame: _apply_region_matchings (cliptracker.py:124-247)
CPX_HD inline void assoc_frame(AssocClip& c, int nreg, int frame_number) {
  (void)frame_number;
  // ---- scores of every (track, region) pair; tracks in id order ----------------------
  // active[] is kept sorted by id (ids are handed out in increasing order and removal preserves order)
  int ns = 0;
  for (int ti = 0; ti < c.n_active; ++ti) {
    ActiveTrack& t = c.active[ti];
    t.matched = 0;
    if (nreg == 0) continue;
    const RegionRec last = pool_at(c, t, t.n - 1);
    double avg_mass, avg_area;
    avg_last5(c, t, avg_mass, avg_area);
    const double max_distance = max_distance_for(c, t, last);
    const double key = (double)t.since_seen + dot_id(t.id);
    for (int ri = 0; ri < nreg; ++ri) {
      double s;
      if (match_pair(c, t, last, c.regs[ri], avg_mass, avg_area, max_distance, &s)) {
        ScoreRec& q = c.scores[ns++];
        q.score = s;
        q.key = key;
        q.track = ti;
        q.region = ri;
      }
    }
  }
  for (int ri = 0; ri < nreg; ++ri) c.used[ri] = 0;
  // ---- greedy assignment in (score, key, insertion) order == the two stable sorts ------
  int remaining = ns;
  while (remaining > 0) {
    int best = -1;
    for (int i = 0; i < ns; ++i) {
      const ScoreRec& q = c.scores[i];
      if (q.track < 0) continue;
      if (best < 0 || q.score < c.scores[best].score ||
          (q.score == c.scores[best].score && q.key < c.scores[best].key))
        best = i;
    }
    if (best < 0) break;
    ScoreRec q = c.scores[best];
    c.scores[best].track = -1;
    remaining -= 1;
    if (c.active[q.track].matched || c.used[q.region]) continue;  // (matched == 2: blanked below)
    c.used[q.region] = 1;
    // cliptracker.py:164-199: with filter_regions_pre_match off the area-of-interest filter runs AFTER the matching: a
    // region too faint or too small still takes its track's match (and is used up), but the track gets a blank frame
    // instead of it -- "rather than if we filter earlier and match this track to a different region"
    const RegionRec& mr = c.regs[q.region];
    if (!c.p->filter_regions_pre_match &&
        ((double)mr.pixel_variance < c.p->aoi_pixel_variance || (double)mr.mass < c.p->aoi_min_mass)) {
      c.active[q.track].matched = 2;
      continue;
    }
    track_add_region(c, c.active[q.track], c.regs[q.region]);
    c.active[q.track].matched = 1;
  }
  for (int ti = 0; ti < c.n_active; ++ti)
    if (c.active[ti].matched == 2) c.active[ti].matched = 0;  // a blanked track is an unmatched one from here on
  // ---- new tracks for unmatched regions, in region-id order -----------------
  const int n_old = c.n_active;
  for (int ri = 0; ri < nreg; ++ri) {
    if (c.used[ri]) continue;
    const RegionRec& r = c.regs[ri];
    int max_ov = -1;
    for (int ti = 0; ti < c.n_active; ++ti) {
      const ActiveTrack& t = c.active[ti];
      const int ov = overlap_area(pool_at(c, t, t.n - 1), r);
      if (ov > max_ov) max_ov = ov;
    }
    if (c.n_active > 0 && (double)max_ov > (double)((long long)r.width * r.height) * 0.25) continue;
    if (c.n_active >= c.max_active || c.n_tracks >= c.max_tracks) {
      c.status = CPX_ERR_OVERFLOW;
      continue;
    }
    // a free slot: not used by any active track
    int slot = 0;
    for (;; ++slot) {
      bool taken = false;
      for (int ti = 0; ti < c.n_active; ++ti) taken |= (c.active[ti].slot == slot);
      if (!taken) break;
    }
    ActiveTrack& t = c.active[c.n_active];
    t.id = c.next_id++;
    t.start_frame = r.frame_number;
    t.n = 0;
    t.rt_frames = 0;
    t.blank_frames = 0;
    t.since_seen = 0;
    t.tracking = 0;
    t.slot = slot;
    t.vx = t.vy = 0.0;
    t.vel_kind = VK_INT;
    kalman_init(t);
    t.pm[0] = t.pm[1] = 0.f;
    c.n_active += 1;
    track_add_region(c, t, r);
    t.matched = 2;
    TrackRec& o = c.tracks[c.n_tracks];
    o.id = t.id;
    o.slot = slot;
    o.start_frame = t.start_frame;
    o.track_index = c.n_tracks;
    c.n_tracks += 1;
  }
  (void)n_old;
  // ---- unmatched tracks: blank frame, keep while tracking (cliptracker.py:237-247) ---------------
  int w = 0;
  for (int ti = 0; ti < c.n_active; ++ti) {
    ActiveTrack& t = c.active[ti];
    bool keep = true;
    if (!t.matched) {
      track_add_blank(c, t);
      keep = t.tracking != 0;
    }
    // persist the counters the host needs for trim / stats
    for (int k = c.n_tracks - 1; k >= 0; --k)
      if (c.tracks[k].id == t.id) {
        TrackRec& o = c.tracks[k];
        o.n_frames = t.n;
        o.blank_frames = t.blank_frames;
        o.since_seen = t.since_seen;
        o.rt_frames = t.rt_frames;
        break;
      }
    if (keep) {
      if (w != ti) c.active[w] = t;
      w += 1;
    }
  }
  c.n_active = w;
}

}  // namespace cpx
