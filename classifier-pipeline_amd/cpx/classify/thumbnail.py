"""Thumbnail choice -- drop-in for the reference's classify/thumbnail.py:13-188.

get_track_thumb_stats() sends the track's usable regions to the HIP thumbnail kernel
(cpx_thumb_stats: external contours + Teh-Chin approximation + masked median, per region) and
best_trackless_thumb()'s window search runs in cpx_trackless_thumb; the ranking arithmetic
(score, thumbnail.py:163-197) is a handful of flops per region and is done here exactly as the
reference does it.  There is no CPU implementation of the per-pixel work."""

import logging
from collections import namedtuple

import numpy as np

from .._lib import REGION_REF_DTYPE
from ..ml_tools import tools
from ..track.region import Region

Stat = namedtuple("Stat", "region contours median_diff")
THUMBNAIL_SIZE = 64


def _device_state(clip):
    st = getattr(clip, "device_state", None)
    if st is None or st.track_result.labels_dev is None:
        raise RuntimeError("thumbnails need the clip's label masks on the device: track the clip with "
                           "keep_frames=True (ClipTrackExtractor) before asking for thumbnails")
    return st


def _ir_clip(clip):
    """IR clips are tracked from uint8 frames by IRTrackExtractor and keep no label masks (the reference hands
    Clip.add_frame its saliency map, which is None with DO_SALIENCY = False: its own loop raises there at this
    snapshot, irtrackextractor.py:432).  The contour / median ranking is not built for them: a track's thumbnail is
    what get_thumbnail_info returns when no region yields statistics -- its first bound, score 0."""
    return getattr(clip, "type", None) == "IR" and getattr(clip, "device_state", None) is None


def get_track_thumb_stats(clip, track):
    """thumbnail.py:70-135 -> (stats, max_mass, max_median_diff, min_median_diff, max_contour)."""
    if _ir_clip(clip):
        return [], 0, 0, 0, 0
    usable = []
    st = None
    for region in track.bounds_history:
        if region.blank or region.mass == 0:
            continue
        if st is None:
            st = _device_state(clip)
        f = st.frame_index(region.frame_number)
        if f is None:  # frame not kept (thumbnail.py:81-82)
            continue
        usable.append((region, f))
    max_mass = 0
    max_median_diff = 0
    min_median_diff = 0
    max_contour = 0
    stats = []
    if not usable:
        return stats, max_mass, max_median_diff, min_median_diff, max_contour
    refs = np.zeros(len(usable), REGION_REF_DTYPE)
    for i, (r, f) in enumerate(usable):
        refs[i] = (f, r.x, r.y, r.width, r.height, 0)
    got = st.engine.thumb_stats(st.frames_dev, st.track_result, refs)
    for (region, _), g in zip(usable, got):
        points = int(g["contours"])
        if points == 0:  # no contour in the region ("shouldnt happen", thumbnail.py:97-100)
            continue
        if points > max_contour:
            max_contour = points
        median_diff = np.float64(g["median_diff"])
        if region.mass > max_mass:
            max_mass = region.mass
        if median_diff > max_median_diff:
            max_median_diff = median_diff
        if median_diff < min_median_diff:
            min_median_diff = median_diff
        stats.append(Stat(region, points, median_diff))
    return stats, max_mass, max_median_diff, min_median_diff, max_contour


def get_thumbnail_info(clip, track):
    stats, max_mass, max_median_diff, min_median_diff, max_contour = get_track_thumb_stats(clip, track)
    if len(stats) == 0:
        if len(track.bounds_history) == 0:
            return None, 0
        return Stat(track.bounds_history[0], 0, 0), 0
    scored = sorted(stats, key=lambda s: score(s, max_mass, max_median_diff, min_median_diff, max_contour),
                    reverse=True)
    return scored[0], score(scored[0], max_mass, max_median_diff, min_median_diff, max_contour)


def score(stat, max_mass, max_median_diff, min_median_diff, max_contour):
    region = stat.region
    mass_percent = region.mass / max_mass * 40          # mass out of 40
    pts = stat.contours / max_contour * 50              # contours out of 50
    centroid_mid = tools.eucl_distance_sq(region.centroid, region.mid) ** 0.5 * 2
    if max_median_diff == 0:
        diff = 0
        if min_median_diff != 0:
            diff = (stat.median_diff + abs(min_median_diff)) / abs(min_median_diff) * 40
    else:
        diff = stat.median_diff / max_median_diff * 40  # median difference out of 40
    total = mass_percent + pts + diff - centroid_mid
    if region.x <= 1 or region.y <= 1 or region.bottom >= 119 or region.right >= 159:
        total = total - 1000                            # prefer frames not on the border
    return total


def best_trackless_thumb(clip):
    """Region for clips without any track (thumbnail.py:13-64)."""
    best_region = None
    for regions in clip.region_history:
        for region in regions:
            if best_region is None or region.mass > best_region.mass:
                best_region = region
    if best_region is not None:
        return best_region
    if _ir_clip(clip):
        return None
    st = getattr(clip, "device_state", None)
    if st is None:
        raise RuntimeError("best_trackless_thumb needs a clip tracked by ClipTrackExtractor")
    best_frame_i = int(np.argmax(clip.stats.frame_stats_mean))
    f = st.frame_index(best_frame_i)
    if f is None:
        logging.warning("best_trackless_thumb: frame %s is not on the device", best_frame_i)
        return None
    x, y = st.engine.trackless_thumb(st.frames_dev, f, st.first_frame)  # clip.background = the file's first frame
    return Region(x, y, THUMBNAIL_SIZE, THUMBNAIL_SIZE, frame_number=best_frame_i,
                  centroid=(x + THUMBNAIL_SIZE // 2, y + THUMBNAIL_SIZE // 2))
