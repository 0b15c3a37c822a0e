"""Clip: per-recording state shared by tracker and classifier
(reference src/track/clip.py:38-492)."""

import datetime
import logging
import os

import numpy as np
import pytz

from ..ml_tools.rectangle import Rectangle
from .framebuffer import FrameBuffer
from .track import Track

RES_X = 160
RES_Y = 120


class ClipStats:
    """Per-frame thermal statistics (clip.py:455-492); the per-frame numbers come
    from the track kernel's reductions."""

    def __init__(self):
        self.mean_background_value = 0
        self.max_temp = None
        self.min_temp = None
        self.mean_temp = None
        self.frame_stats_min = []
        self.frame_stats_max = []
        self.frame_stats_median = []
        self.frame_stats_mean = []
        self.filtered_deviation = None
        self.filtered_sum = 0
        self.temp_thresh = 0
        self.threshold = None
        self.average_delta = None
        self.is_static_background = None

    def add_frame_stats(self, f_min, f_max, f_median, f_mean, filtered_abs_sum=None):
        self.max_temp = f_max if self.max_temp is None else max(self.max_temp, f_max)
        self.min_temp = f_min if self.min_temp is None else min(self.min_temp, f_min)
        self.frame_stats_min.append(f_min)
        self.frame_stats_max.append(f_max)
        self.frame_stats_median.append(f_median)
        self.frame_stats_mean.append(f_mean)
        if filtered_abs_sum is not None:
            self.filtered_sum = self.filtered_sum + np.float64(filtered_abs_sum)

    def completed(self):
        # (sic) the reference casts the float64 totals to uint16 before averaging (clip.py:489-492)
        with np.errstate(all="ignore"):
            if self.filtered_sum is not None:
                self.filtered_deviation = np.mean(np.float64(self.filtered_sum).astype(np.uint16))
            self.mean_temp = np.mean(np.asarray(self.frame_stats_mean, dtype=np.float64).astype(np.uint16))


class Clip:
    PREVIEW = "preview"
    FRAMES_PER_SECOND = 9
    local_tz = pytz.timezone("Pacific/Auckland")
    CLIP_ID = 1

    def __init__(self, trackconfig, sourcefile, background=None, calc_stats=True, model=None, type="thermal",
                 fps=FRAMES_PER_SECOND):
        # The attribute set is the reference's (track/clip.py:56-120: extractor, classifier and metadata writers read
        # them by name); grouped here by what owns them.
        # -- identity and source --
        self._id, Clip.CLIP_ID = Clip.CLIP_ID, Clip.CLIP_ID + 1
        Track._track_id = 1                      # track ids restart with every clip
        self.source_file, self.type, self.config = sourcefile, type, trackconfig
        self.frames_per_second = fps
        self.from_metadata = False
        self.video_start_time = self.location = self.device = self.station_id = self.tags = None
        # -- camera geometry and per-model thresholds (filled by set_res / set_model) --
        self.res_x = self.res_y = self.crop_rectangle = None
        self.camera_model = self.threshold_config = None
        self.background_thresh = self.temp_thresh = None
        self.track_min_delta = self.track_max_delta = None
        # -- background state --
        self._background = None
        self.background_calculated = False
        self.background_frames = 0
        self.disable_background_subtraction = False
        # -- frames as they stream through --
        self.current_frame = -1
        self.frame_buffer = None
        self.ffc_affected = False
        self.ffc_frames = []
        self.calc_stats = calc_stats
        self.stats = ClipStats()
        # -- what tracking produces --
        self.region_history = []
        self.active_tracks = set()
        self.tracks, self.filtered_tracks = [], []
        self.thumb_info = None
        self.set_model(model)
        if background is not None:
            self.set_background(background)

    # ---- identity / camera ----
    def get_id(self):
        return str(self._id)

    def set_model(self, camera_model):
        self.camera_model = camera_model
        threshold = self.config.motion.threshold_for_model(camera_model)
        if threshold:
            self.threshold_config = threshold
            self.background_thresh = threshold.background_thresh
            self.temp_thresh = threshold.temp_thresh
            self.stats.threshold = self.background_thresh
            self.track_min_delta = threshold.track_min_delta
            self.track_max_delta = threshold.track_max_delta

    def set_res(self, res_x, res_y):
        self.res_x = res_x or RES_X
        self.res_y = res_y or RES_Y
        edge = self.config.edge_pixels
        self.crop_rectangle = Rectangle(edge, edge, self.res_x - 2 * edge, self.res_y - 2 * edge)
        for track in self.tracks:
            track.crop_rectangle = self.crop_rectangle

    def set_video_stats(self, video_start_time):
        self.video_start_time = video_start_time
        self.stats.date_time = video_start_time.astimezone(Clip.local_tz)
        self.stats.is_night = video_start_time.astimezone(Clip.local_tz).time().hour >= 2

    # ---- background ----
    @property
    def background(self):
        return self._background

    def set_background(self, frame):
        self._background = frame
        self._background_calculated()

    def update_background(self, frame):
        self._background = frame if self._background is None else np.minimum(self._background, frame)
        self.background_frames += 1

    def _background_calculated(self):
        if self.type != "IR" or self.calc_stats:
            self.stats.mean_background_value = np.average(self._background)
        self.background_calculated = True

    def on_preview(self):
        return not self.background_calculated

    # ---- frames ----
    def set_frame_buffer(self, high_quality_flow, cache_to_disk, use_flow, keep_frames, max_frames=None):
        self.frame_buffer = FrameBuffer(self.source_file, high_quality_flow, cache_to_disk, use_flow, keep_frames,
                                        max_frames)

    def get_frame(self, frame_number):
        return self.frame_buffer.get_frame(frame_number)

    def frames_kept(self):
        return self.frame_buffer.max_frames

    def add_frame(self, thermal, filtered, mask=None, ffc_affected=False, stats=None):
        """stats = (min, max, median, mean, sum|filtered|) from the device when available."""
        self.current_frame += 1
        if ffc_affected:
            self.ffc_frames.append(self.current_frame)
        f = self.frame_buffer.add_frame(thermal, filtered, mask, self.current_frame, ffc_affected)
        if self.calc_stats:
            if stats is None:
                stats = (np.min(thermal), np.max(thermal), np.median(thermal), np.nanmean(thermal),
                         None if filtered is None else np.sum(np.abs(filtered)))
            self.stats.add_frame_stats(*stats)
        return f

    def _add_active_track(self, track):
        self.active_tracks.add(track)
        self.tracks.append(track)

    # ---- metadata ----
    def start_and_end_in_secs(self, track):
        if track.end_s is None:
            track.end_s = (track.end_frame + 1) / self.frames_per_second
        return (track.start_s, track.end_s)

    def start_and_end_time_absolute(self, start_s=0, end_s=None):
        if not end_s:
            end_s = len(self.frame_buffer.frames) / self.frames_per_second
        return (self.video_start_time + datetime.timedelta(seconds=start_s),
                self.video_start_time + datetime.timedelta(seconds=end_s))

    def load_metadata(self, metadata, tag_precedence=None):
        self._id = metadata.get("id", 0)
        device_meta = metadata.get("Device")
        self.tags = metadata.get("Tags")
        if device_meta:
            self.device = device_meta.get("devicename")
        else:
            self.device = os.path.splitext(os.path.basename(str(self.source_file)))[0].split("-")[-1]
        self.location = metadata.get("location")
        self.station_id = metadata.get("stationId")
        tracks = []
        for track_meta in metadata.get("Tracks", metadata.get("tracks", [])):
            track = Track(self.get_id())
            if track.load_track_meta(track_meta, self.frames_per_second, tag_precedence, self.config.min_tag_confidence):
                tracks.append(track)
        self.from_metadata = True
        self.tracks = tracks

    def get_metadata(self, predictions_per_model=None):
        meta = {}
        if self.camera_model:
            meta["camera_model"] = self.camera_model
        meta["background_thresh"] = self.background_thresh
        start, end = self.start_and_end_time_absolute()
        meta["id"] = self._id
        meta["start_time"] = start.isoformat()
        meta["end_time"] = end.isoformat()
        meta["tracks"] = [t.get_metadata(predictions_per_model) for t in self.tracks]
        return meta

    def print_if_verbose(self, info_string):
        if getattr(self.config, "verbose", False):
            logging.info(info_string)
