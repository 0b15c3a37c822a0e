"""Detection stage of the IR tracker: detect_objects_ir on the device (cpx_ir_detect) and the
host-side fragment merge that follows it in IRTrackExtractor (reference ml_tools/imageprocessing.py:185-199 and
track/irtrackextractor.py:324-389, 789-818).  The IR background model (cv2 MOG2) and the MP4 decoder are not built;
this module takes the foreground image the background subtractor hands over."""

import numpy as np


def detect_objects_ir(engine, images_dev, threshold=0, max_components=1024, want_labels=False):
    """images_dev: uint8 [n, H, W] (or [H, W]) foreground on the engine's device.
    -> per frame (components, labels, stats) with the reference's return shape: components counts the background
    label too, stats is an int32 [components, 5] array (x, y, w, h, area) whose row 0 is the background row OpenCV
    reports (whole-image box of the zero pixels is NOT reproduced: row 0 is zeros and the callers skip it,
    irtrackextractor.py:262), labels an int32 [H, W] device tensor or None."""
    single = images_dev.dim() == 2
    if single:
        images_dev = images_dev[None]
    counts, comps, labels = engine.ir_detect(images_dev, threshold, max_components, want_labels)
    out = []
    for i in range(len(counts)):
        c = comps[i, : counts[i]]
        stats = np.zeros((int(counts[i]) + 1, 5), np.int32)
        for k, name in enumerate(("x", "y", "width", "height", "area")):
            stats[1:, k] = c[name]
        out.append((int(counts[i]) + 1, labels[i] if labels is not None else None, stats))
    return out[0] if single else out


def _axis_overlap(a0, a_len, b0, b_len):
    """True when the two extents share more than a boundary: the sum of the lengths exceeds the joint span."""
    return a_len + b_len > max(a0 + a_len, b0 + b_len) - min(a0, b0)


def rect_distance(r_a, r_b):
    """Euclidean gap between two [x, y, w, h, ...] boxes; an axis on which they overlap contributes 0."""
    gap = [0, 0]
    for axis in (0, 1):
        a0, al, b0, bl = r_a[axis], r_a[axis + 2], r_b[axis], r_b[axis + 2]
        if _axis_overlap(a0, al, b0, bl):
            continue
        gap[axis] = (a0 + al) - b0 if a0 < b0 else (b0 + bl) - a0
    return (gap[0] * gap[0] + gap[1] * gap[1]) ** 0.5


def merge_components(rectangles, scale=None):
    """Merges the fragments of one object: [x, y, w, h, area] rows, small ones dropped, largest first; a row absorbs
    every other row that is closer than max_gap to its ORIGINAL box or overlaps it on both axes, and the scan
    restarts after any merge.  Same results as the reference, including its quirks: rows that share the anchor's x
    are never merged with it, and the merged height is measured from the already-updated top edge."""
    min_mass, min_size, max_gap = 40, 16, 40
    if scale:
        min_mass, min_size, max_gap = int(min_mass * scale), int(min_size * scale), max_gap * scale
    kept = [np.array(r).copy() for r in rectangles if r[4] > min_mass or (r[2] > min_size and r[3] > min_size)]
    kept.sort(key=lambda r: r[4], reverse=True)  # stable, like the reference's sorted()
    anchors = kept
    boxes = [r.copy() for r in kept]
    i = 0
    while i < len(anchors):
        anchor, box = anchors[i], boxes[i]
        absorbed = False
        j = 0
        while j < len(anchors):
            other = anchors[j]
            if other[0] == anchor[0]:
                j += 1
                continue
            inside = (_axis_overlap(other[0], other[2], anchor[0], anchor[2])
                      and _axis_overlap(other[1], other[3], anchor[1], anchor[3]))
            if not (inside or rect_distance(anchor, other) < max_gap):
                j += 1
                continue
            right = box[0] + box[2]
            box[0] = min(box[0], other[0])
            box[1] = min(box[1], other[1])
            bottom = max(box[1] + box[3], other[1] + other[3])
            box[2] = max(right, other[0] + other[2]) - box[0]
            box[3] = bottom - box[1]
            box[4] += other[4]
            absorbed = True
            del anchors[j], boxes[j]
            # (deleting an entry before i shifts the list under the cursor exactly as it does in the reference,
            # which restarts from 0 after a merge anyway)
        i = 0 if absorbed else i + 1
    return boxes


class MOG2Background:
    """The IR tracker's background model on the device -- the role of CVBackground("mog2") in the reference
    (track/cliptracker.py:561-613: cv2.createBackgroundSubtractorMOG2(history=1000, detectShadows=False)), for
    `n_streams` videos advancing in lockstep.  Frames and results are uint8 device tensors [n_streams, H, W]
    (or [H, W] for one stream).  OpenCV's algorithm is restated, not linked: see include/cpx.h (parity with cv2 unpinned)."""

    def __init__(self, engine, width, height, n_streams=1, history=1000, var_threshold=16.0):
        import ctypes as C

        self.eng = engine
        self.shape = (n_streams, height, width)
        self._frames = 0
        self._background = None  # the foreground mask of the last update, as the reference names it
        m = C.c_void_p()
        rc = engine.lib.cpx_mog2_create(engine.h, n_streams, width, height, history, float(var_threshold), C.byref(m))
        if rc != 0:
            from .._lib import CpxError

            raise CpxError(rc, engine._err())
        self._m = m

    def _as_batch(self, frame):
        t = self.eng.torch
        if frame.dim() == 2:
            frame = frame[None]
        if frame.dtype != t.uint8 or tuple(frame.shape) != self.shape or not frame.is_contiguous():
            raise ValueError("MOG2Background wants contiguous uint8 frames of shape %s" % (self.shape,))
        return frame

    def set_background(self, background, frames=1):
        self.update_background(background, learning_rate=1)

    def update_background(self, thermal, filtered=None, learning_rate=-1):
        import ctypes as C

        t = self.eng.torch
        single = thermal.dim() == 2
        frame = self._as_batch(thermal)
        mask = t.empty(self.shape, dtype=t.uint8, device=self.eng.device)
        t.cuda.current_stream(self.eng.device).synchronize()
        rc = self.eng.lib.cpx_mog2_apply(self._m, C.c_void_p(frame.data_ptr()), float(learning_rate),
                                         C.c_void_p(mask.data_ptr()))
        if rc != 0:
            from .._lib import CpxError

            raise CpxError(rc, self.eng._err())
        self.eng.synchronize()
        self._background = mask[0] if single else mask
        self._frames += 1
        return self._background

    @property
    def background(self):
        import ctypes as C

        t = self.eng.torch
        out = t.empty(self.shape, dtype=t.uint8, device=self.eng.device)
        rc = self.eng.lib.cpx_mog2_background(self._m, C.c_void_p(out.data_ptr()))
        if rc != 0:
            from .._lib import CpxError

            raise CpxError(rc, self.eng._err())
        self.eng.synchronize()
        return out[0] if self.shape[0] == 1 else out

    def compute_filtered(self, thermal=None):
        return self._background

    def close(self):
        if self._m and self.eng.h:
            self.eng.lib.cpx_mog2_destroy(self._m)
        self._m = None
