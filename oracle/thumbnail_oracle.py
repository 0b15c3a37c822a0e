"""TEST INFRASTRUCTURE ONLY (never imported by the product package): CPU restatement of the
reference's thumbnail stage, SURVEY section 8 f3 -- classify/thumbnail.py:13-188
(best_trackless_thumb :13-64, get_track_thumb_stats :70-135, get_thumbnail_info :138-160,
score :163-197).  cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_TC89_L1) is restated in
oracle/cv2_shim.py; the pair is pinned by the reference's own golden tests/clips/possum.txt
(thumbnail region / contours / median_diff / score of both tracks) and by the vectors the reference
produced under the harness (tests/golden/*_thumbs.json, tests/golden/make_golden_thumbs.py)."""

from collections import namedtuple

import numpy as np

import cv2_shim as cv2

Stat = namedtuple("Stat", "region contours median_diff")
THUMBNAIL_SIZE = 64


def region_stat(region, mask, thermal):
    """(contour points of the largest external contour, median difference) for one region, or None
    when the mask has no contour inside the region (thumbnail.py:89-129)."""
    sub = mask[region.y:region.y + region.height, region.x:region.x + region.width]
    contours, _ = cv2.findContours(np.uint8(sub), cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_TC89_L1)
    if len(contours) == 0:
        return None
    points = max(len(c) for c in contours)
    sub_mask = sub > 0
    thermal_sub = thermal[region.y:region.y + region.height, region.x:region.x + region.width]
    masked = thermal_sub[sub_mask]
    return points, np.median(masked) - np.median(thermal)


def track_thumb_stats(bounds, get_mask, get_thermal):
    """thumbnail.py:70-135 over a track's bounds_history."""
    max_mass = 0
    max_md = 0
    min_md = 0
    max_contour = 0
    stats = []
    for region in bounds:
        if region.blank or region.mass == 0:
            continue
        got = region_stat(region, get_mask(region.frame_number), get_thermal(region.frame_number))
        if got is None:
            continue
        points, md = got
        max_contour = max(max_contour, points)
        if region.mass > max_mass:
            max_mass = region.mass
        if md > max_md:
            max_md = md
        if md < min_md:
            min_md = md
        stats.append(Stat(region, points, md))
    return stats, max_mass, max_md, min_md, max_contour


def score(stat, max_mass, max_md, min_md, max_contour):
    """thumbnail.py:163-197."""
    r = stat.region
    mass_percent = r.mass / max_mass * 40
    pts = stat.contours / max_contour * 50
    mid = (r.x + r.width / 2, r.y + r.height / 2)
    dx = r.centroid[0] - mid[0]
    dy = r.centroid[1] - mid[1]
    centroid_mid = (dx * dx + dy * dy) ** 0.5 * 2
    if max_md == 0:
        diff = 0
        if min_md != 0:
            diff = (stat.median_diff + abs(min_md)) / abs(min_md) * 40
    else:
        diff = stat.median_diff / max_md * 40
    s = mass_percent + pts + diff - centroid_mid
    if r.x <= 1 or r.y <= 1 or r.y + r.height >= 119 or r.x + r.width >= 159:
        s = s - 1000
    return s


def thumbnail_info(bounds, get_mask, get_thermal):
    """thumbnail.py:138-160 -> (Stat or None, best score)."""
    stats, max_mass, max_md, min_md, max_contour = track_thumb_stats(bounds, get_mask, get_thermal)
    if len(stats) == 0:
        if len(bounds) == 0:
            return None, 0
        return Stat(bounds[0], 0, 0), 0
    ranked = sorted(stats, key=lambda s: score(s, max_mass, max_md, min_md, max_contour), reverse=True)
    return ranked[0], score(ranked[0], max_mass, max_md, min_md, max_contour)


def trackless_thumb(region_history, frame_means, get_thermal, background):
    """thumbnail.py:13-64 -> (x, y, width, height, frame_number, centroid, mass).  `background` is the
    clip background (first frame, uint16): the subtraction wraps in uint16 exactly as NumPy's does."""
    best = None
    for regions in region_history:
        for region in regions:
            if best is None or region.mass > best.mass:
                best = region
    if best is not None:
        return (best.x, best.y, best.width, best.height, best.frame_number, best.centroid, best.mass)
    best_i = int(np.argmax(frame_means))
    frame = get_thermal(best_i)
    H, W = frame.shape
    filt = frame - background
    best_region = None
    for y in range(H - THUMBNAIL_SIZE):
        for x in range(W - THUMBNAIL_SIZE):
            thermal_sum = np.mean(frame[y:y + THUMBNAIL_SIZE, x:x + THUMBNAIL_SIZE])
            filtered_sum = np.mean(filt[y:y + THUMBNAIL_SIZE, x:x + THUMBNAIL_SIZE])
            if best_region is None:
                best_region = ((x, y), filtered_sum, thermal_sum)
            elif best_region[1] > 0:
                if best_region[1] < filtered_sum:
                    best_region = ((x, y), thermal_sum, filtered_sum)
            elif best_region[2] < thermal_sum:
                best_region = ((x, y), thermal_sum, filtered_sum)
    (x, y) = best_region[0]
    return (x, y, THUMBNAIL_SIZE, THUMBNAIL_SIZE, best_i, (x + THUMBNAIL_SIZE // 2, y + THUMBNAIL_SIZE // 2), 0)
