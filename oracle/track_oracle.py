"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy) of the reference's
thermal track-extraction path, the checker for the HIP kernels and the timed
``cpu_baseline`` ("port") of bench.py.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product package never does.

Pinned: this restatement reproduces, stage by stage and bit for bit, the
vectors in tests/golden/*.npz that were produced by running the reference itself
(tests/golden/make_golden.py), and through them the reference's own golden
tests/clips/possum.txt (tests/test_oracle_golden.py).

Each function cites the reference file:line (relative to /root/reference/src)
it follows.  The OpenCV operators come from oracle/cv2_shim.py.
"""

import math

import numpy as np

import cv2_shim as cv

# ---------------------------------------------------------------------------
# configuration (config/trackingconfig.py:126-177, trackingmotionconfig.py:24-59)
# ---------------------------------------------------------------------------


class OracleConfig:
    def __init__(self, model="lepton3"):
        self.edge_pixels = 1
        self.frame_padding = 4
        self.min_dimension = 0
        self.denoise = False
        self.aoi_min_mass = 4.0
        self.aoi_pixel_variance = 2.0
        self.cropped_regions_strategy = "cautious"
        self.filter_regions_pre_match = True
        self.track_min_offset = 4.0
        self.track_min_mass = 2.0
        self.min_moving_frames = 2
        self.max_blank_percent = 30
        self.max_jitter = 20
        self.min_duration_secs = 0
        self.max_tracks = None
        self.base_distance_change = 450
        self.min_mass_change = 20
        self.restrict_mass_after = 1.5
        self.mass_change_percent = 0.55
        self.max_distance = 2000
        self.max_blanks = 18
        self.velocity_multiplier = 2
        self.base_velocity = 2
        self.fps = 9
        self.window = 45  # cliptrackextractor.py:173-175
        self.set_model(model)

    def set_model(self, model):
        self.model = model
        if model == "lepton3.5":
            self.background_thresh = 50
            self.weight_add = 1.0  # cliptrackextractor.py:124-125
        else:
            self.background_thresh = 20
            self.weight_add = 0.1  # cliptrackextractor.py:126-127
        self.track_min_delta = 1.0
        self.track_max_delta = 150


# ---------------------------------------------------------------------------
# geometry (ml_tools/rectangle.py, track/region.py)
# ---------------------------------------------------------------------------


class Region:
    """Plain-int region; mirrors track/region.py:27-42 + rectangle.py:6-13.

    The reference's coordinates are a mix of np.int32 (component statistics) and Python ints (the crop rectangle,
    int() results); which one a width / height is decides whether `predicted_mid - width / 2.0` of a Kalman blank
    region is evaluated in float64 (np.int32 / 2.0 -> np.float64) or float32 (Python float is a weak scalar).  The
    values here are Python ints; `py` = (x, y, width, height) records which of them are Python ints in the reference."""

    __slots__ = ("x", "y", "width", "height", "centroid", "mass", "frame_number",
                 "pixel_variance", "id", "was_cropped", "blank", "is_along_border", "py")

    def __init__(self, x, y, width, height, centroid=None, mass=0, frame_number=0,
                 pixel_variance=0, id=0, was_cropped=False, blank=False, is_along_border=False,
                 py=(False, False, False, False)):
        self.x, self.y, self.width, self.height = int(x), int(y), int(width), int(height)
        self.centroid = centroid
        self.mass = mass
        self.frame_number = frame_number
        self.pixel_variance = pixel_variance
        self.id = id
        self.was_cropped = was_cropped
        self.blank = blank
        self.is_along_border = is_along_border
        self.py = tuple(py)

    right = property(lambda s: s.x + s.width)
    bottom = property(lambda s: s.y + s.height)
    mid_x = property(lambda s: s.x + s.width / 2)
    mid_y = property(lambda s: s.y + s.height / 2)
    area = property(lambda s: int(s.width) * s.height)

    def copy(self):
        return Region(self.x, self.y, self.width, self.height, self.centroid, self.mass,
                      self.frame_number, self.pixel_variance, self.id, self.was_cropped,
                      self.blank, self.is_along_border, self.py)

    @staticmethod
    def _crop_axis(pos, pos_py, ext, ext_py, lo, size):
        """One axis of Rectangle.crop (rectangle.py:91-96, setters :50-70) with Python's max / min operand choice:
        max(a, b) is b only if b > a, min(a, b) is b only if b < a; the bounds are Python ints."""
        hi = lo + size
        # left / top = min(bounds.hi, max(self.pos, bounds.lo)); the setter keeps right / bottom
        m, m_py = (lo, True) if lo > pos else (pos, pos_py)
        npos, npos_py = (m, m_py) if m < hi else (hi, True)
        old_far, old_far_py = pos + ext, pos_py and ext_py
        pos, pos_py = npos, npos_py
        ext, ext_py = old_far - pos, old_far_py and pos_py
        # right / bottom = max(bounds.lo, min(self.far, bounds.hi))
        far, far_py = pos + ext, pos_py and ext_py
        mm, mm_py = (hi, True) if hi < far else (far, far_py)
        v, v_py = (mm, mm_py) if mm > lo else (lo, True)
        return pos, pos_py, v - pos, v_py and pos_py

    def crop(self, bx, by, bw, bh):
        """rectangle.py:91-96 with the left/top setters keeping right/bottom."""
        self.x, xpy, self.width, wpy = self._crop_axis(self.x, self.py[0], self.width, self.py[2], bx, bw)
        self.y, ypy, self.height, hpy = self._crop_axis(self.y, self.py[1], self.height, self.py[3], by, bh)
        self.py = (xpy, ypy, wpy, hpy)

    def enlarge(self, border, crop):
        """rectangle.py:138-146."""
        self.x -= border
        self.width += 2 * border
        self.y -= border
        self.height += 2 * border
        self.crop(*crop)

    def overlap_area(self, o):
        xo = max(0, min(self.right, o.right) - max(self.x, o.x))
        yo = max(0, min(self.bottom, o.bottom) - max(self.y, o.y))
        return xo * yo

    def has_moved(self, o):
        return (self.x != o.x and self.right != o.right) or (self.y != o.y and self.bottom != o.bottom)

    def key(self):
        return (self.x, self.y, self.width, self.height)


# ---------------------------------------------------------------------------
# background (piclassifier/motiondetector.py:178-248)
# ---------------------------------------------------------------------------


class WeightedBackground:
    def __init__(self, res_x, res_y, weight_add, edge=1):
        self.edge = edge
        self.weight_add = weight_add
        self.background = None  # float64 [H,W], integer valued
        self.weight = np.zeros((res_y - 2 * edge, res_x - 2 * edge))
        self.average = None

    def _set_edges(self):
        b, e = self.background, self.edge
        for i in range(e):
            b[i] = b[e]
            b[-i - 1] = b[-e - 1]
            b[:, i] = b[:, e]
            b[:, -i - 1] = b[:, -1 - e]

    def process_frame(self, frame):
        """motiondetector.py:197-237.  `frame` is uint16 or the float64 window mean."""
        e = self.edge
        H, W = frame.shape
        inner_of = lambda a: a[e:H - e, e:W - e]  # (edge_pixels may be 0: a [e:-e] slice would be empty)
        f = np.int32(inner_of(frame))
        if self.background is None:
            self.background = np.empty(frame.shape)
            inner_of(self.background)[:, :] = f
            self.average = np.average(f)  # un-rounded float (:209)
            self._set_edges()
            return
        inner = inner_of(self.background)
        cond = inner < f - self.weight
        new_bg = np.where(cond, inner, f)
        self.weight = np.where(cond, self.weight + self.weight_add, 0)
        if np.any(new_bg != inner):
            inner[:, :] = new_bg
            self.average = int(round(np.average(inner)))
            self._set_edges()


# ---------------------------------------------------------------------------
# per-frame pixel stage
# ---------------------------------------------------------------------------


def normalize(data, new_max=1):
    """ml_tools/imageprocessing.py:151-169 (min/max of the data itself)."""
    if data.size == 0:
        return np.zeros(data.shape), (False, None, None)
    mx = np.amax(data)
    mn = np.amin(data)
    if mx == mn:
        if mx == 0:
            return np.zeros(data.shape), (False, mx, mn)
        return data / mx, (True, mx, mn)
    data = new_max * (np.float32(data) - mn) / (mx - mn)
    return data, (True, mx, mn)


def is_affected_by_ffc(time_on, last_ffc_time):
    """piclassifier/cptvmotiondetector.py:211-223 with int ms:
    FFC_PERIOD.seconds == 9 (timedelta(seconds=9.9).seconds)."""
    if time_on is None or last_ffc_time is None:
        return False
    return (time_on - last_ffc_time) < 9


def filtered_frame(thermal, bg, cfg):
    """track/cliptracker.py:93-122.  Returns (float image 0..255, mapped threshold)."""
    f = np.float32(thermal.copy())
    avg_change = int(round(np.average(thermal) - bg.average))
    np.clip(f - bg.background - avg_change, 0, None, out=f)
    f, stats = normalize(f, new_max=255)
    if cfg.denoise:
        f = cv.fastNlMeansDenoising(np.uint8(f), None)
    if stats[1] == stats[2]:
        thresh = cfg.background_thresh
    else:
        thresh = cfg.background_thresh / (stats[1] - stats[2]) * 255
    return f, thresh, avg_change, stats


def detect_objects(image, threshold):
    """ml_tools/imageprocessing.py:240-248 with kernel=(5,5), otsus=False."""
    image = np.uint8(image)
    u8 = image
    image = cv.GaussianBlur(image, (5, 5), 0)
    _, image = cv.threshold(image, threshold, 255, cv.THRESH_BINARY)
    image = cv.morphologyEx(image, cv.MORPH_CLOSE, (5, 5))
    n, labels, stats, cents = cv.connectedComponentsWithStats(image)
    return u8, n, labels, stats, cents


def delta_frame(cur_filtered, prev_filtered):
    """track/cliptracker.py:249-261 (filtered channel only)."""
    if prev_filtered is None:
        return None
    a, _ = normalize(cur_filtered, new_max=255)
    b, _ = normalize(prev_filtered, new_max=255)
    return np.abs(np.float32(a) - np.float32(b))


def regions_of_interest(stats, cents, delta, frame_number, cfg, crop):
    """track/cliptracker.py:263-365 for clip.type == 'thermal'."""
    padding = max(3, cfg.frame_padding)
    regions = []
    for i in range(len(stats)):
        x, y, w, h, area = (int(v) for v in stats[i])
        r = Region(x, y, w, h, mass=area, id=i, frame_number=frame_number, centroid=cents[i])
        if r.width < cfg.min_dimension or r.height < cfg.min_dimension:
            continue
        if delta is not None:
            r.pixel_variance = np.var(delta[y : y + h, x : x + w])
        old = r.copy()
        r.crop(*crop)
        r.was_cropped = old.key() != r.key()
        if cfg.cropped_regions_strategy == "cautious":
            if (old.width - r.width) / old.width > 0.25 or (old.height - r.height) / old.height > 0.25:
                continue
        elif cfg.cropped_regions_strategy in ("none", None):
            if r.was_cropped:
                continue
        if cfg.filter_regions_pre_match and (
            r.pixel_variance < cfg.aoi_pixel_variance and r.mass < cfg.aoi_min_mass
        ):
            continue
        r.enlarge(padding, crop)
        edge = math.ceil(crop[2] * 0.03)
        r.is_along_border = (
            r.was_cropped
            or r.x <= crop[0] + edge
            or r.y <= crop[1] + edge
            or r.right >= crop[2] - edge
            or r.bottom >= crop[3] - edge
        )
        regions.append(r)
    return regions


# ---------------------------------------------------------------------------
# tracker (track/track.py, track/kalman.py)
# ---------------------------------------------------------------------------


class Track:
    MIN_KALMAN_FRAMES = 18
    JITTER_THRESHOLD = 0.3
    MIN_JITTER_CHANGE = 5

    def __init__(self, tid, region, cfg, crop):
        self.id = tid
        self.cfg = cfg
        self.crop = crop
        self.start_frame = region.frame_number
        self.bounds = []
        self.vel_x = []
        self.vel_y = []
        self.prev_frame_num = None
        # RegionTracker state (track.py:65-97)
        self.kalman = cv.KalmanFilter(4, 2)
        self.kalman.measurementMatrix = np.eye(2, 4, dtype=np.float32)
        self.kalman.transitionMatrix = np.array(
            [[1, 0, 1, 0], [0, 1, 0, 1], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
        self.kalman.processNoiseCov = np.eye(4, dtype=np.float32) * 0.03
        self.since_seen = 0
        self.rt_frames = 0
        self.blank_frames = 0
        self.tracking = False
        self.predicted_mid = None
        self.stats = None
        self.add_region(region)

    def __len__(self):
        return len(self.bounds)

    last_bound = property(lambda s: s.bounds[-1])
    end_frame = property(lambda s: s.bounds[-1].frame_number if s.bounds else s.start_frame)
    frames = property(lambda s: s.end_frame + 1 - s.start_frame)
    velocity = property(lambda s: (s.vel_x[-1], s.vel_y[-1]))

    def _tracker_add(self, region):
        """RegionTracker.add_region, track.py:194-214."""
        self.rt_frames += 1
        if region.blank:
            self.blank_frames += 1
            self.since_seen += 1
            stop = min(2 * (self.rt_frames - self.since_seen), self.cfg.max_blanks)
            self.tracking = self.since_seen < stop
        else:
            self.tracking = True
            pts = np.array([np.float32(region.centroid[0]), np.float32(region.centroid[1])], np.float32)
            self.kalman.correct(pts)
            self.since_seen = 0
        p = self.kalman.predict()
        self.predicted_mid = (p[0][0], p[1][0])

    def add_region(self, region):
        """Track.add_region, track.py:646-669."""
        if self.prev_frame_num and region.frame_number:
            for _ in range(region.frame_number - self.prev_frame_num - 1):
                self.add_blank_frame()
        self._tracker_add(region)
        self.bounds.append(region)
        self.prev_frame_num = region.frame_number
        self._update_velocity()

    def _update_velocity(self):
        if len(self.bounds) >= 2:
            self.vel_x.append(self.bounds[-1].centroid[0] - self.bounds[-2].centroid[0])
            self.vel_y.append(self.bounds[-1].centroid[1] - self.bounds[-2].centroid[1])
        else:
            self.vel_x.append(0)
            self.vel_y.append(0)

    def add_blank_frame(self):
        """RegionTracker.add_blank_frame track.py:239-264 + Track.add_blank_frame :729-735."""
        last = self.last_bound
        kalman_amount = self.rt_frames - Track.MIN_KALMAN_FRAMES - self.since_seen * 2
        if kalman_amount > 0:
            # the reference's widths are usually np.int32 (float64 arithmetic); a width that came out of a crop as a
            # Python int makes it float32 (see Region)
            half_w = last.width / 2.0 if last.py[2] else np.float64(last.width / 2.0)
            half_h = last.height / 2.0 if last.py[3] else np.float64(last.height / 2.0)
            r = Region(
                int(np.float32(self.predicted_mid[0]) - half_w),
                int(np.float32(self.predicted_mid[1]) - half_h),
                last.width, last.height,
                centroid=[self.predicted_mid[0], self.predicted_mid[1]],
                py=(True, True, last.py[2], last.py[3]),
            )
            r.crop(*self.crop)
        else:
            r = last.copy()
        r.blank = True
        r.mass = 0
        r.pixel_variance = 0
        r.frame_number = last.frame_number + 1
        self._tracker_add(r)
        self.bounds.append(r)
        self.prev_frame_num = r.frame_number
        self._update_velocity()

    def _avg_last5(self, attr):
        tot, cnt = 0, 0
        for b in reversed(self.bounds):
            if not b.blank:
                tot += getattr(b, attr)
                cnt += 1
            if cnt == 5:
                break
        return 0 if cnt == 0 else tot / cnt

    def predicted_velocity(self):
        nonblank = self.rt_frames - self.blank_frames
        if nonblank <= Track.MIN_KALMAN_FRAMES:
            return (0, 0)
        return (self.predicted_mid[0] - self.last_bound.centroid[0],
                self.predicted_mid[1] - self.last_bound.centroid[1])

    def match(self, regions):
        """RegionTracker.match track.py:118-192 (with the builtin-`type` quirk)."""
        cfg = self.cfg
        scores = []
        avg_mass = self._avg_last5("mass")
        avg_area = self._avg_last5("area")
        vx, vy = self.velocity
        if len(self) == 1:
            vx = vy = cfg.base_velocity
        vx, vy = cfg.velocity_multiplier * vx, cfg.velocity_multiplier * vy
        vel_d = vx * vx + vy * vy
        pv = self.predicted_velocity()
        pred_d = max(vel_d, pv[0] * pv[0] + pv[1] * pv[1])
        max_distance = cfg.base_distance_change + max(vel_d, pred_d)
        last = self.last_bound
        for region in regions:
            size_change = abs(region.area - avg_area) / (avg_area + 50)
            d0 = (region.x - last.x) ** 2 + (region.y - last.y) ** 2
            d2 = (region.right - last.right) ** 2 + (region.bottom - last.bottom) ** 2
            distance = (d0 + d2) / 2
            max_size = self._max_size_change(region)
            max_mass = None
            if cfg.mass_change_percent is not None and len(self) > cfg.restrict_mass_after * cfg.fps:
                pct = cfg.mass_change_percent
                v = self.velocity
                if np.sum(np.abs(v)) > 5:
                    pct = pct + 0.1
                max_mass = max(cfg.min_mass_change, avg_mass * pct)
            if max_mass and abs(avg_mass - region.mass) > max_mass:
                continue
            if distance > max_distance:
                continue
            if size_change > max_size:
                continue
            scores.append((distance, self, region))
        return scores

    def _max_size_change(self, region):
        """get_max_size_change track.py:312-326."""
        last = self.last_bound
        exiting = region.is_along_border and not last.is_along_border
        entering = not exiting and last.is_along_border
        pct = 1.5
        if len(self) < 5:
            pct = 2
        vel = np.sum(np.abs(self.velocity))
        if entering or exiting:
            pct = 2
            if vel > 10:
                pct *= 3
        elif vel > 10:
            pct *= 2
        return pct

    # ---- end of clip (track.py:737-905) ------------------------------------
    def trim(self):
        mass_history = [int(b.mass) for b in self.bounds]
        median_mass = np.median(mass_history)
        filter_mass = max(0.005 * median_mass, 2)
        start = 0
        while start < len(self) and mass_history[start] <= filter_mass:
            start += 1
        end = len(self) - 1
        while end > 0 and mass_history[end] <= filter_mass:
            if self.since_seen > 0:
                self.since_seen -= 1
                self.blank_frames -= 1
            end -= 1
        if end < start:
            self.bounds, self.vel_x, self.vel_y = [], [], []
            self.blank_frames = 0
        else:
            self.start_frame += start
            self.bounds = self.bounds[start : end + 1]
            self.vel_x = self.vel_x[start : end + 1]
            self.vel_y = self.vel_y[start : end + 1]

    def calculate_stats(self):
        if len(self) <= 1:
            self.stats = dict(movement=0, max_offset=0, score=0, average_mass=0, median_mass=0,
                              delta_std=0, region_jitter=0, jitter_smaller=0, jitter_bigger=0,
                              blank_percent=0, frames_moved=0, mass_std=0, average_velocity=0)
            return
        non_blank = [b for b in self.bounds if not b.blank]
        mass_history = [int(b.mass) for b in non_blank]
        variance_history = [b.pixel_variance for b in non_blank if b.pixel_variance]
        movement = 0
        max_offset = 0
        frames_moved = 0
        avg_vel = 0
        first = (self.bounds[0].mid_x, self.bounds[0].mid_y)
        for i, (vx, vy) in enumerate(zip(self.vel_x, self.vel_y)):
            region = self.bounds[i]
            if not region.blank:
                avg_vel += abs(vx) + abs(vy)
            if i == 0:
                continue
            if region.blank or self.bounds[i - 1].blank:
                continue
            if region.has_moved(self.bounds[i - 1]) or region.is_along_border:
                movement += (vx**2 + vy**2) ** 0.5
                off = (first[0] - region.mid_x) ** 2 + (first[1] - region.mid_y) ** 2
                max_offset = max(max_offset, off)
                frames_moved += 1
        avg_vel = avg_vel / len(mass_history)
        max_offset = math.sqrt(max_offset)
        delta_std = float(np.mean(variance_history)) ** 0.5
        jb = js = 0
        for i, bound in enumerate(self.bounds[1:]):
            prev = self.bounds[i]
            if prev.is_along_border or bound.is_along_border:
                continue
            hd = bound.height - prev.height
            wd = prev.width - bound.width
            th = max(Track.MIN_JITTER_CHANGE, prev.height * Track.JITTER_THRESHOLD)
            tv = max(Track.MIN_JITTER_CHANGE, prev.width * Track.JITTER_THRESHOLD)
            if abs(hd) > th:
                if hd > 0:
                    jb += 1
                else:
                    js += 1
            elif abs(wd) > tv:
                if wd > 0:
                    jb += 1
                else:
                    js += 1
        movement_points = (movement**0.5) + max_offset
        delta_points = delta_std * 25.0
        jitter_percent = int(round(100 * (jb + js) / float(self.frames)))
        blank_percent = int(round(100.0 * self.blank_frames / self.frames))
        score = (min(movement_points, 100) + min(delta_points, 100)
                 + (100 - jitter_percent) + (100 - blank_percent))
        self.stats = dict(
            movement=float(movement), max_offset=float(max_offset), score=float(score),
            average_mass=float(np.mean(mass_history)), median_mass=float(np.median(mass_history)),
            delta_std=float(delta_std), region_jitter=jitter_percent, jitter_smaller=js,
            jitter_bigger=jb, blank_percent=blank_percent, frames_moved=frames_moved,
            mass_std=float(np.std(mass_history)), average_velocity=float(avg_vel))


def apply_matchings(state, regions):
    """track/cliptracker.py:124-247."""
    cfg = state["cfg"]
    active = sorted(state["active"], key=lambda t: t.id)
    scores = []
    for t in active:
        scores.extend(t.match(regions))
    scores.sort(key=lambda r: r[1].since_seen + float(".{}".format(r[1].id)))
    scores.sort(key=lambda r: r[0])
    matched, used, blanked = [], set(), []
    unmatched = list(regions)
    for score, track, region in scores:
        if track in matched or id(region) in used or track in blanked:
            continue
        used.add(id(region))
        unmatched.remove(region)
        # cliptracker.py:164-199: with filter_regions_pre_match off the area-of-interest filter runs AFTER the matching: a
        # region too faint or too small still takes its track's match (and is used up), but the track gets a blank frame
        # instead of it -- "rather than if we filter earlier and match this track to a different region"
        if not cfg.filter_regions_pre_match and (
            region.pixel_variance < cfg.aoi_pixel_variance or region.mass < cfg.aoi_min_mass
        ):
            blanked.append(track)
            continue
        track.add_region(region)
        matched.append(track)
    new_tracks = []
    # the reference iterates a set of eq=False objects; canonical order = region id
    for region in sorted(unmatched, key=lambda r: r.id):
        overlaps = [t.last_bound.overlap_area(region) for t in state["active"]]
        if len(overlaps) > 0 and max(overlaps) > region.area * 0.25:
            continue
        t = Track(state["next_id"], region, cfg, state["crop"])
        state["next_id"] += 1
        new_tracks.append(t)
        state["active"].append(t)
        state["tracks"].append(t)
    unactive = [t for t in state["active"] if t not in matched and t not in new_tracks]
    state["active"] = matched + new_tracks
    for t in unactive:
        t.add_blank_frame()
        if t.tracking:
            state["active"].append(t)
    return new_tracks


def filter_tracks(state):
    """track/cliptracker.py:367-486."""
    cfg = state["cfg"]
    for t in state["tracks"]:
        t.trim()
    for t in state["tracks"]:
        t.calculate_stats()
    state["tracks"].sort(reverse=True, key=lambda t: t.stats["score"])
    good, filtered = [], []
    for t in state["tracks"]:
        s = t.stats
        reason = None
        if len(t) < cfg.min_duration_secs * cfg.fps:
            reason = "Track filtered.  Too short"
        elif s["max_offset"] < cfg.track_min_offset or s["frames_moved"] < cfg.min_moving_frames:
            reason = "Track filtered.  Didn't move"
        elif s["blank_percent"] > cfg.max_blank_percent:
            reason = "Track filtered. Too Many Blanks"
        elif s["region_jitter"] > cfg.max_jitter:
            reason = "Track filtered.  Too Jittery"
        elif s["delta_std"] < cfg.track_min_delta:
            reason = "Track filtered.  Too static"
        elif s["delta_std"] > cfg.track_max_delta:
            reason = "Track filtered.  Too Dynamic"
        elif s["average_mass"] < cfg.track_min_mass:
            reason = "Track filtered.  Mass too small"
        if reason:
            filtered.append((reason, t))
        else:
            good.append(t)
    if cfg.max_tracks is not None and cfg.max_tracks < len(good):
        filtered.extend(("Too many tracks", t) for t in good[cfg.max_tracks :])
        good = good[: cfg.max_tracks]
    state["tracks"] = good
    state["filtered"] = filtered


# ---------------------------------------------------------------------------
# whole clip (track/cliptrackextractor.py:98-247)
# ---------------------------------------------------------------------------


def track_clip(frames, time_on=None, last_ffc=None, background_flags=None, cfg=None,
               keep=False, do_tracking=True, apply_filter=True):
    """frames: uint16 [N,H,W] -- every frame of the file in order (the first one
    initialises the background even when it is a background frame,
    cliptrackextractor.py:129-139).  Returns a dict of per-frame results."""
    cfg = cfg or OracleConfig()
    n, H, W = frames.shape
    e = cfg.edge_pixels
    crop = (e, e, W - 2 * e, H - 2 * e)
    bg = WeightedBackground(W, H, cfg.weight_add, e)
    bg.process_frame(frames[0])
    out = dict(init_bg=bg.background.copy(), init_avg=bg.average, frames=[])
    state = dict(cfg=cfg, crop=crop, active=[], tracks=[], next_id=1, filtered=[])
    window = []
    prev_filtered = None
    cur = -1
    ffc_frames = []
    region_history = []
    for i in range(n):
        if background_flags is not None and background_flags[i]:
            continue
        thermal = frames[i]
        cur += 1
        ffc = False
        if time_on is not None:
            ffc = is_affected_by_ffc(time_on[i], last_ffc[i])
        filtered = np.float32(thermal) - bg.background  # cliptrackextractor.py:212
        bg_used_avg = bg.average
        img, thresh, avg_change, nstats = filtered_frame(thermal, bg, cfg)
        u8, ncomp, labels, stats, cents = detect_objects(img, thresh)
        if ffc:
            ffc_frames.append(cur)
        regions = []
        if do_tracking:
            if ffc:
                state["active"] = []
            else:
                delta = delta_frame(filtered, prev_filtered)
                regions = regions_of_interest(stats[1:], cents[1:], delta, cur, cfg, crop)
                if do_tracking != "regions":  # "regions": pixel stage + region lists only
                    apply_matchings(state, regions)
            region_history.append(regions)
        prev_filtered = filtered
        window.append(thermal)
        if len(window) > cfg.window:
            window.pop(0)
        last_avg = np.mean(window, axis=0)
        bg.process_frame(last_avg)
        rec = dict(index=cur, ffc=ffc, threshold=float(thresh), avg_change=avg_change,
                   n_components=ncomp - 1, stats=stats[1:].copy(), centroids=cents[1:].copy(),
                   regions=[r.copy() for r in regions], bg_used_avg=float(bg_used_avg),
                   bg_after_avg=float(bg.average), norm_max=float(nstats[1]), norm_min=float(nstats[2]))
        rec["filtered"] = filtered.astype(np.int32)
        rec["obj_u8"] = u8
        rec["mask"] = labels
        rec["bg_after"] = bg.background.astype(np.int32)
        rec["weight_after"] = bg.weight.copy()
        if not keep:
            # keep memory bounded: callers that want images pass keep=True
            pass
        out["frames"].append(rec)
    if do_tracking and do_tracking != "regions" and apply_filter:
        filter_tracks(state)
    out["tracks"] = state["tracks"]
    out["filtered_tracks"] = state["filtered"]
    out["ffc_frames"] = ffc_frames
    out["region_history"] = region_history
    return out
