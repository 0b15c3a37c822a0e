"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy) of the reference's
classification pre-processing and prediction aggregation; the checker for the
HIP crop/tile kernel.  Pinned by tests/golden/*_classify_fs*.npz, which were
produced by the reference's own Interpreter.classify_track
(tests/golden/make_golden_classify.py); tests/test_oracle_golden.py compares
bit for bit.

Follows (paths relative to /root/reference/src):
  ml_tools/interpreter.py:315-363   get_limits
  ml_tools/interpreter.py:365-474   preprocess_segments
  ml_tools/preprocess.py:56-113     preprocess_frame
  ml_tools/imageprocessing.py:11-82 resize_and_pad / resize_cv
  ml_tools/preprocess.py:151-202    preprocess_movement (+ imageprocessing.py:85-104 square_clip)
  classify/trackprediction.py:127-171 classified_track
cv2.resize is oracle/cv2_shim.py:resize (not pinned by any reference golden).
"""

import numpy as np

import cv2_shim as cv


def get_limits(regions, filtered_of):
    """min / max of region.subimage(frame.filtered) over the track's non-blank regions
    (max starts at 0) -- interpreter.py:315-363 with diff_norm=True, thermal_diff_norm=False.
    get_limits first calls Frame.float_arrays() (frame.py:316-324), which turns the stored
    filtered frames into float32 IN PLACE: the limits are float32 scalars and everything
    downstream of it (crop, resize, normalise) is float32 arithmetic."""
    min_diff, max_diff = None, 0
    for r in reversed(regions):
        if r.blank or r.width <= 0 or r.height <= 0:
            continue
        f = filtered_of(r.frame_number)
        if f is None:
            continue
        sub = np.float32(f[r.y : r.y + r.height, r.x : r.x + r.width])
        new_max, new_min = np.amax(sub), np.amin(sub)
        if min_diff is None or new_min < min_diff:
            min_diff = new_min
        if new_max > max_diff:
            max_diff = new_max
    return min_diff, max_diff


def resize_and_pad(frame, fs, region, crop, pad=None, interpolation=cv.INTER_LINEAR):
    """imageprocessing.py:11-70 with keep_edge=True, edge_offset=(0,0,0,0)."""
    scale = (np.array((fs, fs)) / np.array(frame.shape[:2])).min()
    width = min(max(round(frame.shape[1] * scale), 1), fs)
    height = min(max(round(frame.shape[0] * scale), 1), fs)
    if pad is None:
        pad = np.min(frame)
    out = np.full((fs, fs), pad, dtype=frame.dtype)
    small = cv.resize(np.float32(frame), (width, height), interpolation=interpolation)
    fh, fw = small.shape[:2]
    ox = (fs - fw) // 2
    oy = (fs - fh) // 2
    cx, cy, cw, ch = crop
    if region.x <= cx:
        ox = min(0, fs - fw)
    elif region.x + region.width >= cx + cw:
        ox = max((fs - 0) - fw, 0)
    if region.y <= cy:
        oy = min(0, fs - fh)
    elif region.y + region.height >= cy + ch:
        oy = max(fs - fh - 0, 0)
    out[oy : oy + fh, ox : ox + fw] = small
    return out


def normalize(data, mn=None, mx=None, new_max=1):
    """imageprocessing.py:151-169."""
    if data.size == 0:
        return np.zeros(data.shape)
    if mx is None:
        mx = np.amax(data)
    if mn is None:
        mn = np.amin(data)
    if mx == mn:
        if mx == 0:
            return np.zeros(data.shape)
        return data / mx
    return new_max * (np.float32(data) - mn) / (mx - mn)


def preprocess_frame(thermal, filtered, region, fs, crop, median, limits, clip_at_zero):
    """preprocess.py:56-113 for the classify call (calculate_filtered=False, sub_median=True,
    filtered_norm_limits given, thermal_norm_limits None) -> (thermal f32 [fs,fs], filtered f32 [fs,fs])."""
    t = np.float32(thermal[region.y : region.y + region.height, region.x : region.x + region.width])
    f = np.float32(filtered[region.y : region.y + region.height, region.x : region.x + region.width])
    t = resize_and_pad(t, fs, region, crop)
    f = resize_and_pad(f, fs, region, crop, pad=0)
    t -= median
    if clip_at_zero:
        np.clip(t, 0, None, out=t)
    f = normalize(f, mn=limits[0], mx=limits[1], new_max=255)
    t = normalize(t, new_max=255)
    return t, f


def preprocess_segments(thermal_of, filtered_of, regions_by_frame, track_regions, segments, fs, crop):
    """interpreter.py:365-474 + preprocess.py:151-202 for channels (thermal, filtered),
    wr-resnet (no preprocess_fn).  segments: list of 25 frame numbers each.
    filtered frames are integer valued (float32 after Frame.float_arrays())."""
    medians, unique = {}, {}
    clip_at_zero = True
    for seg in segments:
        for fn in seg:
            fn = int(fn)
            if fn in unique:
                continue
            r = regions_by_frame[fn]
            unique[fn] = r
            th = thermal_of(fn)
            medians[fn] = np.median(th)
            if clip_at_zero:
                sub = np.float32(th[r.y : r.y + r.height, r.x : r.x + r.width]) - medians[fn]
                if np.median(sub) <= 0:
                    clip_at_zero = False
    limits = get_limits(track_regions, filtered_of)
    data = {}
    for fn, r in unique.items():
        data[fn] = preprocess_frame(thermal_of(fn), filtered_of(fn), r, fs, crop, medians[fn], limits, clip_at_zero)
    out = np.zeros((len(segments), 5 * fs, 5 * fs, 2), dtype=np.float64)
    for s, seg in enumerate(segments):
        assert len(seg) == 25, "parity runs use full 25-frame segments (short ones are padded at random, F13)"
        for i, fn in enumerate(seg):
            ty, tx = divmod(i, 5)
            t, f = data[int(fn)]
            out[s, ty * fs : (ty + 1) * fs, tx * fs : (tx + 1) * fs, 0] = np.float32(t)
            out[s, ty * fs : (ty + 1) * fs, tx * fs : (tx + 1) * fs, 1] = np.float32(f)
    info = dict(medians=medians, limits=limits, clip_at_zero=clip_at_zero)
    return np.float32(out), info


def classified_track(predictions, smooth_masses=None, prediction_frames=None, labels=None, square_width=5):
    """class_best_score of a track from its segment predictions: trackprediction.py:127-171 plus the
    low-evidence cap of Interpreter.track_prediction_from_raw (interpreter.py:151-168): a single segment
    with fewer than square_width**2 / 4 distinct frames that is not 'false-positive' is capped at 0.5."""
    predictions = np.asarray(predictions)
    if smooth_masses is not None:
        masses = np.array(smooth_masses)
        smoothed = predictions * masses[:, None]
        score = np.sum(smoothed, axis=0) / np.sum(masses)
    else:
        score = np.sum(predictions, axis=0)
        score = score / np.sum(score)
    if prediction_frames is not None and len(prediction_frames) == 1 and \
            len(set(int(f) for f in prediction_frames[0])) < square_width**2 / 4:
        if labels is None or labels[int(np.argmax(score))] != "false-positive":
            total = np.sum(score)
            if total > 0.5:
                score = score * (0.5 / total)
    return score
