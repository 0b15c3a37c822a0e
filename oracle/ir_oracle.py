"""TEST INFRASTRUCTURE ONLY (never imported by the product package): CPU restatement of the detection stage of the
reference's IR tracker, SURVEY section 8 f4 --
  detect_objects_ir      ml_tools/imageprocessing.py:185-199  (uint8 cast, MORPH_OPEN with the tuple kernel, threshold,
                                                               connectedComponentsWithStats)
  merge_components       track/irtrackextractor.py:324-389    (fragment merging by midpoint distance / overlap)
OpenCV's calls are the restatements of oracle/cv2_shim.py (MORPH_OPEN is the same 1x2 element as the close the golden
pins, but itself not covered by a golden).  Pinned by tests/golden/ir_detect_golden.json: the reference's own
detect_objects_ir (under the harness) and its merge_components (pure Python) on seeded 640x480 foreground masks.
The IR background model (cv2.createBackgroundSubtractorMOG2) and the MP4 decoder are not restated: parity unpinned,
not built."""

import numpy as np

import cv2_shim as cv2


def detect_objects_ir(image, threshold=0, kernel=(15, 15)):
    image = np.uint8(image)
    image = cv2.morphologyEx(image, cv2.MORPH_OPEN, kernel)
    _, image = cv2.threshold(image, threshold, 255, cv2.THRESH_BINARY)
    components, mask, stats, _ = cv2.connectedComponentsWithStats(image)
    return components, mask, stats


def rect_distance(r_a, r_b):
    """irtrackextractor.py:789-818: per axis the gap between the two extents (0 when they overlap on that axis),
    combined as a Euclidean distance."""
    x_1 = x_2 = y_1 = y_2 = 0
    if r_a[2] + r_b[2] > max(r_a[0] + r_a[2], r_b[2] + r_b[0]) - min(r_a[0], r_b[0]):
        pass
    elif r_a[0] < r_b[0]:
        x_1, x_2 = r_a[0] + r_a[2], r_b[0]
    else:
        x_1, x_2 = r_b[0] + r_b[2], r_a[0]
    if r_a[3] + r_b[3] > max(r_a[1] + r_a[3], r_b[1] + r_b[3]) - min(r_a[1], r_b[1]):
        pass
    elif r_a[1] < r_b[1]:
        y_1, y_2 = r_a[1] + r_a[3], r_b[1]
    else:
        y_1, y_2 = r_b[1] + r_b[3], r_a[1]
    dx, dy = x_1 - x_2, y_1 - y_2
    return (dx * dx + dy * dy) ** 0.5


def merge_components(rectangles, scale=None):
    """irtrackextractor.py:324-389, statement for statement (including cur_bottom being computed from the x extent
    and never used): rectangles are [x, y, w, h, area] rows; returns the merged rows."""
    min_mass = 10 * 4
    min_size = 16
    max_gap = 40
    if scale:
        min_mass = int(min_mass * scale)
        min_size = int(min_size * scale)
        max_gap *= scale
    rectangles = [np.array(r).copy() for r in rectangles if r[4] > min_mass or (r[2] > min_size and r[3] > min_size)]
    rectangles = sorted(rectangles, key=lambda s: s[4], reverse=True)
    rectangles = [(r, r.copy()) for r in rectangles]
    rect_i = 0
    while rect_i < len(rectangles):
        rect, merged_r = rectangles[rect_i]
        merged = False
        index = 0
        while index < len(rectangles):
            within = False
            r_2 = rectangles[index][0]
            if r_2[0] == rect[0]:
                index += 1
                continue
            if r_2[2] + rect[2] > max(r_2[0] + r_2[2], rect[2] + rect[0]) - min(r_2[0], rect[0]):
                within = r_2[3] + rect[3] > max(r_2[1] + r_2[3], rect[1] + rect[3]) - min(r_2[1], rect[1])
            distance = rect_distance(rect, r_2)
            if distance < max_gap or within:
                cur_right = merged_r[0] + merged_r[2]
                merged_r[0] = min(merged_r[0], r_2[0])
                merged_r[1] = min(merged_r[1], r_2[1])
                merged_r[2] = max(cur_right, r_2[0] + r_2[2])
                merged_r[3] = max(merged_r[1] + merged_r[3], r_2[1] + r_2[3])
                merged_r[2] -= merged_r[0]
                merged_r[3] -= merged_r[1]
                merged_r[4] += r_2[4]
                merged = True
                del rectangles[index]
            else:
                index += 1
        if merged:
            rect_i = 0
        else:
            rect_i += 1
    return [r[1] for r in rectangles]
