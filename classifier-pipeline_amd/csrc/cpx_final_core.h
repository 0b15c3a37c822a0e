// cpx_final_core.h -- end-of-clip work for ONE clip as scalar code (one GPU lane per clip in
// cpx_assoc.hip; also host-compilable for tests): trim, movement statistics, score, rejects, score
// ordering, and the deterministic segment plan for classification.
//
// Follows the reference (paths relative to /root/reference/src):
//   track/track.py:873-905   Track.trim
//   track/track.py:737-833   Track.calculate_stats
//   track/cliptracker.py:367-486  filter_tracks / filter_track
// NumPy reductions are reproduced as NumPy evaluates them (add.reduce = pairwise summation; float32
// for the per-frame variances, float64 otherwise).
#pragma once
#include <math.h>
#include <stdint.h>

#include "cpx.h"
#include "cpx_assoc_core.h"

namespace cpx {

// ---- numpy pairwise summation (loops_utils.h.src), contiguous input ----------------------
template <typename T>
CPX_HD inline T np_pairwise(const T* a, int n) {
  if (n < 8) {
    T res = (T)0;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    T r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}
// np.add.reduce over a contiguous 1-D array == the pairwise sum of all n elements (checked against
// NumPy 2.2 for n = 1..300)
template <typename T>
CPX_HD inline T np_sum(const T* a, int n) {
  return np_pairwise(a, n);
}

// np.median without the full sort: quickselect of the upper middle element (reorders `a`), then the largest
// element below it for even n.  Same value as sorting: the two middle order statistics are what they are.
CPX_HD inline double np_median_select(double* a, int n) {
  const int k = n >> 1;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const double pivot = a[(lo + hi) >> 1];
    int i = lo, j = hi;
    while (i <= j) {
      while (a[i] < pivot) ++i;
      while (a[j] > pivot) --j;
      if (i <= j) {
        const double tmp = a[i];
        a[i] = a[j];
        a[j] = tmp;
        ++i;
        --j;
      }
    }
    if (k <= j) hi = j;
    else if (k >= i) lo = i;
    else break;
  }
  const double upper = a[k];
  if (n & 1) return upper;
  double lower = a[0];
  for (int i = 1; i < k; ++i) lower = a[i] > lower ? a[i] : lower;
  return (lower + upper) / 2.0;
}

struct FinalScratch {
  double* d;  // [max_frames]
  float* f;   // [max_frames]
};

CPX_HD inline const RegionRec& treg(const RegionRec* pool, int max_active, const cpx_track_summary& t, int i) {
  return pool[(size_t)(t.start_frame + i) * max_active + t.slot];
}

// velocity between bounds i-1 and i, typed like NumPy (float32 for two Kalman centroids)
CPX_HD inline void vel_at(const RegionRec* pool, int ma, const cpx_track_summary& t, int i, int first, double* vx,
                          double* vy) {
  // `first`: index (in the untrimmed history) this track started at; velocity 0 for the very first bound
  if (i + first == 0) {
    *vx = 0.0;
    *vy = 0.0;
    return;
  }
  const RegionRec& c = treg(pool, ma, t, i);
  const RegionRec& p = pool[(size_t)(t.start_frame + i - 1) * ma + t.slot];
  if ((c.flags & RF_CENTROID_F32) && (p.flags & RF_CENTROID_F32)) {
    *vx = (double)((float)c.cx - (float)p.cx);
    *vy = (double)((float)c.cy - (float)p.cy);
  } else {
    *vx = c.cx - p.cx;
    *vy = c.cy - p.cy;
  }
}

// trim + stats + reject decision of one track; `t` enters with the untrimmed record
CPX_HD inline void finalize_track(const cpx_filter_params& fp, const RegionRec* pool, cpx_track_summary& t,
                                  FinalScratch sc) {
  const int ma = fp.max_active_tracks;
  // ---- trim (track.py:873-905) ----
  int n = t.n_frames;
  for (int i = 0; i < n; ++i) sc.d[i] = (double)treg(pool, ma, t, i).mass;
  const double median_all = np_median_select(sc.d, n);
  double filter_mass = 0.005 * median_all;
  if (!(filter_mass > 2.0)) filter_mass = 2.0;  // max(filter_mass, 2)
  int start = 0;
  while (start < n && (double)treg(pool, ma, t, start).mass <= filter_mass) ++start;
  int end = n - 1;
  while (end > 0 && (double)treg(pool, ma, t, end).mass <= filter_mass) {
    if (t.since_seen > 0) {
      t.since_seen -= 1;
      t.blank_frames -= 1;
    }
    --end;
  }
  int first = 0;
  if (end < start) {
    t.n_frames = 0;
    t.blank_frames = 0;
  } else {
    first = start;
    t.start_frame += start;
    t.n_frames = end - start + 1;
  }
  n = t.n_frames;
  // ---- stats (track.py:737-833) ----
  t.movement = t.max_offset = t.score = t.average_mass = t.median_mass = t.delta_std = t.mass_std = 0.0;
  t.average_velocity = 0.0;
  t.frames_moved = t.region_jitter = t.jitter_bigger = t.jitter_smaller = t.blank_percent = 0;
  if (n > 1) {
    double movement = 0.0, max_offset = 0.0, avg_vel = 0.0;
    int frames_moved = 0, n_seen = 0, n_var = 0;
    const RegionRec& r0 = treg(pool, ma, t, 0);
    const double fx = r0.x + r0.width / 2.0, fy = r0.y + r0.height / 2.0;
    for (int i = 0; i < n; ++i) {
      const RegionRec& r = treg(pool, ma, t, i);
      double vx, vy;
      vel_at(pool, ma, t, i, first, &vx, &vy);
      if (!(r.flags & RF_BLANK)) {
        avg_vel += fabs(vx) + fabs(vy);
        sc.d[n_seen++] = (double)r.mass;
        if (r.pixel_variance != 0.0f) sc.f[n_var++] = r.pixel_variance;
      }
      if (i == 0) continue;
      const RegionRec& p = treg(pool, ma, t, i - 1);
      if ((r.flags & RF_BLANK) || (p.flags & RF_BLANK)) continue;
      const bool moved = (r.x != p.x && r.x + r.width != p.x + p.width) || (r.y != p.y && r.y + r.height != p.y + p.height);
      if (moved || (r.flags & RF_BORDER)) {
        movement += sqrt(vx * vx + vy * vy);
        const double mx = r.x + r.width / 2.0, my = r.y + r.height / 2.0;
        const double off = (fx - mx) * (fx - mx) + (fy - my) * (fy - my);
        if (off > max_offset) max_offset = off;
        frames_moved += 1;
      }
    }
    avg_vel = avg_vel / (double)n_seen;
    max_offset = sqrt(max_offset);
    // delta_std = float(np.mean(float32 variances)) ** 0.5
    const float vmean = n_var > 0 ? np_sum(sc.f, n_var) / (float)n_var : NAN;
    const double delta_std = sqrt((double)vmean);
    int bigger = 0, smaller = 0;
    for (int i = 1; i < n; ++i) {
      const RegionRec& p = treg(pool, ma, t, i - 1);
      const RegionRec& c = treg(pool, ma, t, i);
      if ((p.flags & RF_BORDER) || (c.flags & RF_BORDER)) continue;
      const int dh = c.height - p.height, dw = p.width - c.width;
      const double th = fmax(5.0, p.height * 0.3), tw = fmax(5.0, p.width * 0.3);
      if (fabs((double)dh) > th) {
        if (dh > 0) bigger += 1; else smaller += 1;
      } else if (fabs((double)dw) > tw) {
        if (dw > 0) bigger += 1; else smaller += 1;
      }
    }
    const int frames = n;  // end_frame + 1 - start_frame
    const double movement_points = sqrt(movement) + max_offset;
    const double delta_points = delta_std * 25.0;
    const int jitter_percent = (int)rint(100.0 * (bigger + smaller) / (double)frames);
    const int blank_percent = (int)rint(100.0 * t.blank_frames / (double)frames);
    // Python min(x, 100) keeps a NaN x (no variance recorded): 100 < nan is False
    const double mp = (100.0 < movement_points) ? 100.0 : movement_points;
    const double dp = (100.0 < delta_points) ? 100.0 : delta_points;
    const double score = mp + dp + (100 - jitter_percent) + (100 - blank_percent);
    // mass statistics over the non-blank regions (np.mean / np.median / np.std of int lists)
    const double msum = np_sum(sc.d, n_seen);
    const double mmean = msum / (double)n_seen;
    double* dev2 = sc.d + n_seen;  // second half of the scratch
    for (int i = 0; i < n_seen; ++i) {
      const double d = sc.d[i] - mmean;
      dev2[i] = d * d;
    }
    const double mstd = sqrt(np_sum(dev2, n_seen) / (double)n_seen);
    t.movement = movement;
    t.max_offset = max_offset;
    t.score = score;
    t.average_mass = mmean;
    t.median_mass = np_median_select(sc.d, n_seen);
    t.delta_std = delta_std;
    t.mass_std = mstd;
    t.average_velocity = avg_vel;
    t.frames_moved = frames_moved;
    t.region_jitter = jitter_percent;
    t.jitter_bigger = bigger;
    t.jitter_smaller = smaller;
    t.blank_percent = blank_percent;
  }
  // ---- reject decision (cliptracker.py:422-486) ----
  int reject = CPX_TRACK_KEPT;
  if ((double)n < fp.min_duration_secs * fp.fps) reject = CPX_REJECT_TOO_SHORT;
  else if (t.max_offset < fp.track_min_offset || t.frames_moved < fp.min_moving_frames) reject = CPX_REJECT_DIDNT_MOVE;
  else if (t.blank_percent > fp.max_blank_percent) reject = CPX_REJECT_TOO_MANY_BLANKS;
  else if (t.region_jitter > fp.max_jitter) reject = CPX_REJECT_TOO_JITTERY;
  else if (t.delta_std < fp.track_min_delta) reject = CPX_REJECT_TOO_STATIC;
  else if (t.delta_std > fp.track_max_delta) reject = CPX_REJECT_TOO_DYNAMIC;
  else if (t.average_mass < fp.track_min_mass) reject = CPX_REJECT_MASS_TOO_SMALL;
  t.reject = reject;
}

CPX_HD inline bool region_usable(const RegionRec& r, bool has_no_mass, const int* proc_ffc) {
  // get_segments' frame filter (datasetstructures.py:1022-1037) with skip_ffc, no frame_min_mass
  return (has_no_mass || r.mass > 0) && !proc_ffc[r.frame_number] && !(r.flags & RF_BLANK) && r.width > 0 &&
         r.height > 0;
}

// ---- segment selection (ml_tools/datasetstructures.py:972-1301 as Interpreter.frames_for_prediction calls it through
// Track.get_segments, ml_tools/interpreter.py:178-243, track/track.py:480-545: ALL_RANDOM_MASKED, segment_width =
// square_width^2 = 25, segment_frame_spacing 9, min_segments 1, no max_segments, repeats 1, dont_filter False,
// segment_min_mass None).  The reference draws at random; this is its outcome when every draw is the
// identity (shuffle keeps the order, choice without replacement takes the first k, choice with replacement cycles
// through the array) -- pinned against the reference itself run under such draws,
// tests/golden/segments_identity_golden.json.
//   positions p = 0 .. n_frames-1 index the (trimmed) track's bounds; U = number of usable positions
//   segment_count = max(1, U / 9); for segment i the pool is every usable position not yet taken -- and, when
//   U >= 40, outside the window [25 i, 25 i + 25) (datasetstructures.py:1189-1197); a pool shorter than a quarter
//   segment, or than half a segment once a segment exists, ends the loop (:1200-1209; the first segment is exempt:
//   min_segments = 1); the segment takes the first 25 of the pool (:1223-1226); a short one is padded with its own
//   first frames (:1236-1244) and, if still short, by cycling through the sorted padded list (:1276-1282); a segment
//   whose mean mass (over the first padding) is below 1 is dropped but keeps its frames taken (:1249-1255).
// emit(segment index, tile j, position) is called for the 25 sorted frames of every kept segment; `taken` is scratch of
// n_frames bytes and holds, on return, 1 for positions used by a KEPT segment.  Returns the number of kept segments.
template <typename Emit>
CPX_HD inline int plan_track_segments(const cpx_filter_params& fp, const RegionRec* pool, const cpx_track_summary& t,
                                      const int* proc_ffc, int per, unsigned char* taken, Emit emit) {
  const int n = t.n_frames;
  long long msum = 0;
  for (int i = 0; i < n; ++i) msum += (unsigned short)treg(pool, fp.max_active_tracks, t, i).mass;  // np.uint16 masses
  const bool has_no_mass = msum == 0;
  int usable = 0;
  for (int i = 0; i < n; ++i) {
    taken[i] = 0;
    usable += region_usable(treg(pool, fp.max_active_tracks, t, i), has_no_mass, proc_ffc);
  }
  if (usable == 0) return 0;  // "Nothing to load"
  int segment_count = usable / 9;
  if (segment_count < 1) segment_count = 1;
  const bool masked = usable >= 40;
  int made = 0;
  // bit 1 of taken[]: position consumed by some segment (kept or dropped); bit 0: by a kept one
  for (int i = 0; i < segment_count; ++i) {
    const int m0 = masked ? i * 25 : -1, m1 = masked ? i * 25 + 25 : -1;
    int plen = 0;
    for (int p = 0; p < n; ++p)
      if (!(taken[p] & 2) && !(p >= m0 && p < m1) && region_usable(treg(pool, fp.max_active_tracks, t, p), has_no_mass, proc_ffc))
        plen += 1;
    if (plen == 0 || made >= 1) {
      if ((2 * plen < per && made > 0) || 4 * plen < per) break;
    }
    int sel[32];
    int have = 0;
    for (int p = 0; p < n && have < per; ++p)
      if (!(taken[p] & 2) && !(p >= m0 && p < m1) && region_usable(treg(pool, fp.max_active_tracks, t, p), has_no_mass, proc_ffc)) {
        sel[have++] = p;
        taken[p] |= 2;
      }
    // first padding: the first min(remaining, have) frames once more, then sorted (sel is ascending)
    int fr[32];
    int len = 0;
    const int extra = (per - have) < have ? (per - have) : have;
    for (int j = 0; j < have; ++j) {
      fr[len++] = sel[j];
      if (j < extra) fr[len++] = sel[j];
    }
    long long smass = 0;
    for (int j = 0; j < len; ++j) smass += (unsigned short)treg(pool, fp.max_active_tracks, t, fr[j]).mass;
    if (smass < len) continue;  // segment_avg_mass < 1: dropped ("segment_mass")
    if (len < per) {
      // second padding: frames[k % len] of the sorted list for k = 0 .. per-len-1, then sorted again
      int cnt[32];
      const int len0 = len, add = per - len0;
      for (int j = 0; j < len0; ++j) cnt[j] = 1 + add / len0 + (j < add % len0 ? 1 : 0);
      int tmp[32];
      for (int j = 0; j < len0; ++j) tmp[j] = fr[j];
      len = 0;
      for (int j = 0; j < len0; ++j)
        for (int c = 0; c < cnt[j]; ++c) fr[len++] = tmp[j];
    }
    for (int j = 0; j < have; ++j) taken[sel[j]] |= 1;
    for (int j = 0; j < per; ++j) emit(made, j, fr[j]);
    made += 1;
  }
  for (int p = 0; p < n; ++p) taken[p] &= 1;
  return made;
}

// all tracks of a clip: finalize each, order by score (stable, descending), apply max_tracks,
// count the classification work.  counts = {kept, refs, samples, 0}
CPX_HD inline void finalize_clip(const cpx_filter_params& fp, const RegionRec* pool, const cpx_track_record* recs,
                                 int n_tracks, const int* proc_ffc, int square_width, cpx_track_summary* out,
                                 int* counts, FinalScratch sc) {
  for (int k = 0; k < n_tracks; ++k) {
    cpx_track_summary& t = out[k];
    t.id = recs[k].id;
    t.slot = recs[k].slot;
    t.start_frame = recs[k].start_frame;
    t.n_frames = recs[k].n_frames;
    t.blank_frames = recs[k].blank_frames;
    t.since_seen = recs[k].since_seen;
    t.n_segments = 0;
    finalize_track(fp, pool, t, sc);
  }
  // clip.tracks.sort(reverse=True, key=score): stable descending
  for (int k = 0; k < n_tracks; ++k) {
    int rank = 0;
    for (int j = 0; j < n_tracks; ++j)
      if (out[j].score > out[k].score || (out[j].score == out[k].score && j < k)) rank += 1;
    out[k].rank = rank;
  }
  int kept = 0;
  if (fp.max_tracks >= 0) {
    // the best max_tracks kept tracks survive (cliptracker.py:403-414)
    for (int k = 0; k < n_tracks; ++k) {
      if (out[k].reject != CPX_TRACK_KEPT) continue;
      int better = 0;
      for (int j = 0; j < n_tracks; ++j)
        if (out[j].reject == CPX_TRACK_KEPT && out[j].rank < out[k].rank) better += 1;
      if (better >= fp.max_tracks) out[k].reject = CPX_REJECT_TOO_MANY_TRACKS;
    }
  }
  int refs = 0, samples = 0;
  const int per = square_width * square_width;
  for (int k = 0; k < n_tracks; ++k) {
    cpx_track_summary& t = out[k];
    if (t.reject != CPX_TRACK_KEPT) continue;
    kept += 1;
    int nonblank = 0;
    for (int i = 0; i < t.n_frames; ++i) {
      const RegionRec& r = treg(pool, fp.max_active_tracks, t, i);
      if (!(r.flags & RF_BLANK) && r.width > 0 && r.height > 0) nonblank += 1;
    }
    t.n_segments = plan_track_segments(fp, pool, t, proc_ffc, per, reinterpret_cast<unsigned char*>(sc.d),
                                       [](int, int, int) {});
    refs += nonblank;
    samples += t.n_segments;
  }
  counts[0] = kept;
  counts[1] = refs;
  counts[2] = samples;
  counts[3] = 0;
}

// fill pass: prefix = exclusive prefix sums of counts over the clips of the batch; `taken` is scratch of max_frames bytes
CPX_HD inline void plan_clip(const cpx_filter_params& fp, const RegionRec* pool, const cpx_track_summary* sums,
                             int n_tracks, const int* proc_ffc, const int* proc_idx, int square_width, int clip,
                             const int* prefix, cpx_region_ref* refs, int* track_offsets, cpx_crop_req* reqs,
                             int* sample_track, int* track_clip, unsigned char* taken, bool last_clip = false) {
  int ti = prefix[0], ri = prefix[1], si = prefix[2];
  const int per = square_width * square_width;
  // kept tracks in score order
  for (int want = 0; want < n_tracks; ++want) {
    int k = -1;
    for (int j = 0; j < n_tracks; ++j)
      if (sums[j].rank == want) k = j;
    if (k < 0 || sums[k].reject != CPX_TRACK_KEPT) continue;
    const cpx_track_summary& t = sums[k];
    track_offsets[ti] = ri;
    track_clip[2 * ti] = clip;
    track_clip[2 * ti + 1] = t.id;
    // segments: one crop request per tile, in the sorted frame order of the segment
    const int made = plan_track_segments(fp, pool, t, proc_ffc, per, taken, [&](int s, int j, int p) {
      const RegionRec& r = treg(pool, fp.max_active_tracks, t, p);
      cpx_crop_req q;
      q.frame = proc_idx[r.frame_number];
      q.x = r.x; q.y = r.y; q.width = r.width; q.height = r.height;
      q.track = ti;
      q.sample = si + s;
      q.tile = j;
      reqs[(size_t)(si + s) * per + j] = q;
    });
    for (int s = 0; s < made; ++s) sample_track[si + s] = ti;
    si += made;
    // refs: every non-blank region (get_limits walks the whole track); in_segment marks the frames a segment uses
    // (the clip_thermals_at_zero test, interpreter.py:372-399)
    for (int i = 0; i < t.n_frames; ++i) {
      const RegionRec& r = treg(pool, fp.max_active_tracks, t, i);
      if ((r.flags & RF_BLANK) || r.width <= 0 || r.height <= 0) continue;
      cpx_region_ref q;
      q.frame = proc_idx[r.frame_number];
      q.x = r.x; q.y = r.y; q.width = r.width; q.height = r.height;
      q.in_segment = taken[i] ? 1 : 0;
      refs[ri++] = q;
    }
    ti += 1;
  }
  // track_offsets has one entry more than there are kept tracks: the end of the last track's refs
  if (last_clip) track_offsets[ti] = ri;
}

}  // namespace cpx
