"""Host-side Track object.  The per-frame association (matching, Kalman, blank
frames) runs on the GPU (csrc/cpx_assoc_core.h); this class holds the resulting
history and does the end-of-clip work of the reference's Track: trim,
movement statistics / score, smoothing, metadata
(reference src/track/track.py:372-1031)."""

import math
from collections import namedtuple

import numpy as np

from ..ml_tools.rectangle import Rectangle
from ..ml_tools.tools import eucl_distance_sq
from .region import Region

TrackMovementStatistics = namedtuple(
    "TrackMovementStatistics",
    "movement max_offset score average_mass median_mass delta_std region_jitter jitter_smaller jitter_bigger "
    "blank_percent frames_moved mass_std average_velocity",
)
TrackMovementStatistics.__new__.__defaults__ = (0,) * len(TrackMovementStatistics._fields)


class RegionTracker:
    """Counters of the device-side tracker that the end-of-clip code reads (track.py:65-75,216-226)."""

    MIN_KALMAN_FRAMES = 18
    MASS_CHANGE_PERCENT = 0.55

    def __init__(self, track_id, frames=0, blank_frames=0, frames_since_target_seen=0, tracking=False):
        self.track_id = track_id
        self.frames = frames
        self._blank_frames = blank_frames
        self._frames_since_target_seen = frames_since_target_seen
        self._tracking = tracking
        self._last_bound = None

    blank_frames = property(lambda s: s._blank_frames)
    frames_since_target_seen = property(lambda s: s._frames_since_target_seen)
    tracking = property(lambda s: s._tracking)
    last_bound = property(lambda s: s._last_bound)
    nonblank_frames = property(lambda s: s.frames - s._blank_frames)


class Track:
    JITTER_THRESHOLD = 0.3  # a region growing / shrinking by 30 % counts as jitter
    MIN_JITTER_CHANGE = 5   # ... if it changes by at least 5 pixels
    _track_id = 1

    def __init__(self, clip_id, id=None, fps=9, tracking_config=None, crop_rectangle=None, tracker_version=None):
        if not id:
            self._id = Track._track_id
            Track._track_id += 1
        else:
            self._id = id
        self.clip_id = clip_id
        self.start_frame = None
        self.start_s = None
        self.end_s = None
        self.fps = fps
        self.frame_list = []
        self.bounds_history = []
        self.vel_x = []
        self.vel_y = []
        self.tag = "unknown"
        self.prev_frame_num = None
        self.confidence = None
        self.from_metadata = False
        self.tags = None
        self.predictions = None
        self.predicted_tag = None
        self.predicted_confidence = None
        self.all_class_confidences = None
        self.prediction_classes = None
        self.crop_rectangle = crop_rectangle
        self.tracker_version = tracker_version
        self.tracker = RegionTracker(self._id) if tracking_config is not None else None
        self.thumb_info = None
        self.score = None
        self.stats = None
        self.in_trap = False
        self.trigger_frame = None
        self.trap_reported = False
        self.direction = 0
        self.trap_tag = None

    # ---- construction from the device records ---------------------------------------------
    @classmethod
    def from_device(cls, clip, record, regions, tracker_version=None, tracking_config=None):
        t = cls(clip.get_id(), id=int(record["id"]), fps=clip.frames_per_second, tracking_config=tracking_config or True,
                crop_rectangle=clip.crop_rectangle, tracker_version=tracker_version)
        t.start_frame = int(record["start_frame"])
        t.start_s = t.start_frame / float(clip.frames_per_second)
        t.bounds_history = [Region.from_record(r) for r in regions]
        # a blank region that copies its predecessor shares that predecessor's centroid object
        for i in range(1, len(t.bounds_history)):
            cur, prv = t.bounds_history[i], t.bounds_history[i - 1]
            if cur.blank and type(cur.centroid) is type(prv.centroid) and np.array_equal(cur.centroid, prv.centroid):
                cur.centroid = prv.centroid
        t._velocities_from_history()
        rt = t.tracker
        rt.frames = int(record["rt_frames"])
        rt._blank_frames = int(record["blank_frames"])
        rt._frames_since_target_seen = int(record["since_seen"])
        rt._last_bound = t.bounds_history[-1] if t.bounds_history else None
        t.prev_frame_num = t.bounds_history[-1].frame_number if t.bounds_history else None
        return t

    def append_from_device(self, record, region_record):
        """Incremental tracking: the region the device added to this track for the newest frame."""
        region = Region.from_record(region_record)
        if self.bounds_history:
            prv = self.bounds_history[-1]
            if region.blank and type(region.centroid) is type(prv.centroid) and \
                    np.array_equal(region.centroid, prv.centroid):
                region.centroid = prv.centroid
            self.vel_x.append(region.centroid[0] - prv.centroid[0])
            self.vel_y.append(region.centroid[1] - prv.centroid[1])
        else:
            self.vel_x.append(0)
            self.vel_y.append(0)
        self.bounds_history.append(region)
        rt = self.tracker
        rt.frames = int(record["rt_frames"])
        rt._blank_frames = int(record["blank_frames"])
        rt._frames_since_target_seen = int(record["since_seen"])
        rt._last_bound = region
        self.prev_frame_num = region.frame_number

    def _velocities_from_history(self):
        """Track.update_velocity (track.py:657-669) for the whole history."""
        self.vel_x, self.vel_y = [], []
        for i, b in enumerate(self.bounds_history):
            if i == 0:
                self.vel_x.append(0)
                self.vel_y.append(0)
            else:
                p = self.bounds_history[i - 1]
                self.vel_x.append(b.centroid[0] - p.centroid[0])
                self.vel_y.append(b.centroid[1] - p.centroid[1])

    # ---- simple accessors -----------------------------------------------------------------------
    def get_id(self):
        return self._id

    def __len__(self):
        return len(self.bounds_history)

    def __repr__(self):
        return "Track: {} frames# {}".format(self.get_id(), len(self))

    @property
    def blank_frames(self):
        return 0 if self.tracker is None else self.tracker.blank_frames

    @property
    def tracking(self):
        return self.tracker.tracking

    @property
    def frames_since_target_seen(self):
        return self.tracker.frames_since_target_seen

    @property
    def end_frame(self):
        return self.bounds_history[-1].frame_number if self.bounds_history else self.start_frame

    @property
    def frames(self):
        return self.end_frame + 1 - self.start_frame

    @property
    def nonblank_frames(self):
        return self.end_frame + 1 - self.start_frame - self.blank_frames

    @property
    def last_mass(self):
        return self.bounds_history[-1].mass

    @property
    def velocity(self):
        return self.vel_x[-1], self.vel_y[-1]

    @property
    def last_bound(self):
        return self.bounds_history[-1]

    def update_trapped_state(self):
        """track.py:951-958: trapped once the last two bounds are inside the trap."""
        if self.in_trap:
            return self.in_trap
        min_frames = 2
        if len(self.bounds_history) < min_frames:
            return False
        self.in_trap = all(r.in_trap for r in self.bounds_history[-min_frames:])
        return self.in_trap

    def get_stats(self):
        """IRTrackExtractor asks a live track for its statistics (irtrackextractor.py:481,528); the reference's Track
        has no such method at this snapshot (an AttributeError there): read as calculate_stats() -> stats."""
        self.calculate_stats()
        return self.stats

    def crop_regions(self):
        if self.crop_rectangle is not None:
            for region in self.bounds_history:
                region.crop(self.crop_rectangle)

    def average_mass(self):
        masses = [b.mass for b in reversed(self.bounds_history) if not b.blank][:5]
        return sum(masses) / len(masses) if masses else 0

    def average_area(self):
        areas = [b.area for b in reversed(self.bounds_history) if not b.blank][:5]
        return sum(areas) / len(areas) if areas else 0

    # ---- end of clip (track.py:737-905) --------------------------------------------------------------
    def trim(self):
        """Drop low-mass frames from both ends: mass <= max(2, 0.5 % of the median mass)."""
        mass_history = [int(b.mass) for b in self.bounds_history]
        filter_mass = max(0.005 * np.median(mass_history), 2)
        n = len(mass_history)
        start = 0
        while start < n and mass_history[start] <= filter_mass:
            start += 1
        end = n - 1
        while end > 0 and mass_history[end] <= filter_mass:
            if self.tracker and self.frames_since_target_seen > 0:
                self.tracker._frames_since_target_seen -= 1
                self.tracker._blank_frames -= 1
            end -= 1
        if end < start:
            self.bounds_history, self.vel_x, self.vel_y = [], [], []
            if self.tracker:
                self.tracker._blank_frames = 0
        else:
            self.start_frame += start
            self.bounds_history = self.bounds_history[start : end + 1]
            self.vel_x = self.vel_x[start : end + 1]
            self.vel_y = self.vel_y[start : end + 1]
        self.start_s = self.start_frame / float(self.fps)

    def set_end_s(self, fps):
        self.end_s = self.start_s if len(self) == 0 else (self.end_frame + 1) / fps

    def calculate_stats(self):
        if len(self) <= 1:
            self.stats = TrackMovementStatistics()
            return
        hist = self.bounds_history
        seen = [b for b in hist if not b.blank]
        mass_history = [int(b.mass) for b in seen]
        variance_history = [b.pixel_variance for b in seen if b.pixel_variance]
        movement = 0
        max_offset = 0
        frames_moved = 0
        avg_vel = 0
        origin = hist[0].mid
        for i, (vx, vy) in enumerate(zip(self.vel_x, self.vel_y)):
            region = hist[i]
            if not region.blank:
                avg_vel += abs(vx) + abs(vy)
            if i == 0 or region.blank or hist[i - 1].blank:
                continue
            if region.has_moved(hist[i - 1]) or region.is_along_border:
                movement += (vx**2 + vy**2) ** 0.5
                max_offset = max(max_offset, eucl_distance_sq(origin, region.mid))
                frames_moved += 1
        avg_vel = avg_vel / len(mass_history)
        max_offset = math.sqrt(max_offset)
        # std of the inter-frame delta: sqrt of the mean per-frame variance
        delta_std = float(np.mean(variance_history)) ** 0.5
        bigger = smaller = 0
        for prev, cur in zip(hist[:-1], hist[1:]):
            if prev.is_along_border or cur.is_along_border:
                continue
            dh = cur.height - prev.height
            dw = prev.width - cur.width
            if abs(dh) > max(Track.MIN_JITTER_CHANGE, prev.height * Track.JITTER_THRESHOLD):
                if dh > 0:
                    bigger += 1
                else:
                    smaller += 1
            elif abs(dw) > max(Track.MIN_JITTER_CHANGE, prev.width * Track.JITTER_THRESHOLD):
                if dw > 0:
                    bigger += 1
                else:
                    smaller += 1
        movement_points = (movement**0.5) + max_offset
        delta_points = delta_std * 25.0
        jitter_percent = int(round(100 * (bigger + smaller) / float(self.frames)))
        blank_percent = int(round(100.0 * self.blank_frames / self.frames))
        score = min(movement_points, 100) + min(delta_points, 100) + (100 - jitter_percent) + (100 - blank_percent)
        self.stats = TrackMovementStatistics(
            movement=float(movement), max_offset=float(max_offset), average_mass=float(np.mean(mass_history)),
            median_mass=float(np.median(mass_history)), delta_std=float(delta_std), score=float(score),
            region_jitter=jitter_percent, jitter_bigger=bigger, jitter_smaller=smaller, blank_percent=blank_percent,
            frames_moved=frames_moved, mass_std=float(np.std(mass_history)), average_velocity=float(avg_vel))

    def smooth(self, frame_bounds: Rectangle):
        if not self.bounds_history:
            return
        hist = self.bounds_history
        out = []
        for i, cur in enumerate(hist):
            prv = hist[max(0, i - 1)]
            nxt = hist[min(len(hist) - 1, i + 1)]
            w = (prv.width + cur.width + nxt.width) / 3
            h = (prv.height + cur.height + nxt.height) / 3
            r = Region(int(cur.centroid[0] - w / 2), int(cur.centroid[1] - h / 2), int(w), int(h))
            r.crop(frame_bounds)
            out.append(r)
        self.bounds_history = out

    def start_and_end_in_secs(self):
        if self.end_s is None:
            self.end_s = self.start_s if len(self) == 0 else (self.end_frame + 1) / self.fps
        return (self.start_s, self.end_s)

    def get_metadata(self, predictions_per_model=None):
        start_s, end_s = self.start_and_end_in_secs()
        info = {"id": self.get_id()}
        if self.in_trap:
            info["trap_triggered"] = self.in_trap
            info["trigger_frame"] = self.trigger_frame
            if self.trap_tag is not None:
                info["trap_tag"] = self.trap_tag
        info["tracker_version"] = self.tracker_version
        info["start_s"] = round(start_s, 2)
        info["end_s"] = round(end_s, 2)
        info["num_frames"] = len(self)
        info["frame_start"] = self.start_frame
        info["frame_end"] = self.end_frame
        info["positions"] = self.bounds_history
        if self.thumb_info is not None:
            info["thumbnail"] = self.thumb_info.to_metadata()
        info["tracking_score"] = 0 if self.stats is None else self.stats.score
        prediction_info = []
        if predictions_per_model:
            for model_id, predictions in predictions_per_model.items():
                prediction = predictions.prediction_for(self.get_id())
                if prediction is None:
                    continue
                meta = prediction.get_metadata(predictions.thresholds)
                meta["model_id"] = model_id
                prediction_info.append(meta)
        info["predictions"] = prediction_info
        return info

    def load_track_meta(self, track_meta, frames_per_second, tag_precedence=None, min_confidence=0.8):
        """Rebuild a track from saved metadata (track.py:568-627); positions only."""
        self.tracker_version = track_meta.get("tracker_version", "unknown")
        self.from_metadata = True
        self._id = track_meta["id"]
        extra = track_meta.get("data", track_meta)
        self.start_s = extra.get("start_s", extra.get("start"))
        self.end_s = extra.get("end_s", extra.get("end"))
        self.fps = frames_per_second
        self.tags = track_meta.get("tags")
        self.stats = TrackMovementStatistics(score=track_meta.get("tracking_score", 0))
        positions = track_meta.get("positions")
        if not positions:
            return False
        self.bounds_history, self.frame_list = [], []
        for i, position in enumerate(positions):
            if isinstance(position, list):
                region = Region.region_from_array(position[1])
                if region.frame_number is None:
                    region.frame_number = round(position[0] * frames_per_second)
            else:
                region = Region.region_from_json(position)
                if region.frame_number is None:
                    if "frameTime" not in position:
                        raise Exception("No frame number info for track")
                    region.frame_number = position["frameTime"] * 9 if i == 0 else self.bounds_history[0].frame_number + i
            if self.start_frame is None:
                self.start_frame = region.frame_number
            self.bounds_history.append(region)
            self.frame_list.append(region.frame_number)
        return True
