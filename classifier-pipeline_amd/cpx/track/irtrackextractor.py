"""IRTrackExtractor -- the 640 x 480 IR tracker around the device detection stage (reference
src/track/irtrackextractor.py:94-787).

Per frame, on the GPU: MOG2 background model (cpx_mog2_apply, the role of CVBackground / cv2 MOG2,
track/cliptracker.py:561-613) -> foreground mask -> detect_objects_ir (cpx_ir_detect) -> [host: merge_components,
a few dozen rectangles] -> per-region variance of the frame-to-frame change (cpx_ir_delta_variance) -> region
filter + matching + Kalman on the shared association core with the IR TrackingConfig (cpx_associate_frame).  Host
side, as in the reference: trap geometry (Line, get_trap_lines, filter_components, inside_trap_top / _bottom),
filter_track, end-of-clip trim.

What the reference itself cannot pin at this snapshot (both are AttributeErrors there, so its IR tracker does not
run): ``clip.frame_buffer.get_frame_ago`` (irtrackextractor.py:647; FrameBuffer has no such method, only a comment
at framebuffer.py:101) and ``track.get_stats()`` (irtrackextractor.py:481,528; Track has calculate_stats only).
Here get_frame_ago(n) is read as "the frame n frames before the current one" and get_stats() as calculate_stats()
-> stats; everything else follows the reference line by line and its pure functions are pinned by a golden made
from the reference's own code (tests/golden/make_golden_irtrap.py).  MP4 decoding (cv2.VideoCapture) is third-party
code that is not part of the build: parse_clip needs cv2, parse_frames takes the gray frames themselves."""

import logging
import os
import time
from datetime import datetime

import numpy as np

from .._lib import COMPONENT_DTYPE
from ..engine import ComponentStream, TrackEngine
from ..tracking import make_track_params
from .cliptrackextractor import ClipTracker
from .irdetect import MOG2Background, detect_objects_ir, merge_components, rect_distance  # noqa: F401
from .region import Region
from .track import Track


class Line:
    """y = m x + c in the trap's coordinates (irtrackextractor.py:40-74)."""

    def __init__(self, m, c):
        self.m = m
        self.c = c

    def is_above(self, point):
        return point[1] > self.y_res(point[0])

    def is_below(self, point):
        return not self.is_above(point)

    def is_left(self, point):
        return point[0] < self.x_res(point[1])

    def is_right(self, point):
        return not self.is_left(point)

    def y_res(self, x):
        return x * self.m + self.c

    def x_res(self, y):
        return (y - self.c) / self.m

    def __str__(self):
        return f"y={self.m}x + {self.c}"


def get_trap_lines(trap_size):
    """irtrackextractor.py:77-91."""
    if trap_size == "S":
        lb, rb = Line(1.3, 297.5), Line(-1.4, 1148)
    else:
        lb, rb = Line(1.28, 180), Line(-1.2, 979)
    logging.info("Getting trap lines for trap size %s left bottom %s right bottom %s", trap_size, lb, rb)
    return lb, rb


class Direction:
    LEFT = 1
    BOTTOM = 2
    RIGHT = 4
    TOP = 8
    MIDDLE = 16


class _LazyRegionHistory:
    """clip.region_history of a batch-tracked clip: per frame the list of Region objects, built from the association's
    region records on first access (a sequence like the list the one-video path fills)."""

    def __init__(self, assoc, first, n):
        self._assoc, self._first, self._n = assoc, first, n
        self._made = {}

    def __len__(self):
        return self._n

    def _frame(self, q):
        regs = self._made.get(q)
        if regs is None:
            regs = []
            for rec in self._assoc.frame_regions(self._first + q):
                r = Region.from_record(rec)
                r.centroid = [int(r.centroid[0]), int(r.centroid[1])]
                regs.append(r)
            self._made[q] = regs
        return regs

    def __getitem__(self, q):
        if isinstance(q, slice):
            return [self._frame(i) for i in range(*q.indices(self._n))]
        if q < 0:
            q += self._n
        if not 0 <= q < self._n:
            raise IndexError(q)
        return self._frame(q)

    def __iter__(self):
        return (self._frame(q) for q in range(self._n))


class IRTrackExtractor(ClipTracker):
    PREVIEW = "preview"
    VERSION = 10
    TYPE = "IR"
    FRAMES_AGO = 10  # get_delta_frame compares with the frame this many frames back (irtrackextractor.py:641)

    @property
    def type(self):
        return IRTrackExtractor.TYPE

    @property
    def tracker_version(self):
        return self.version

    @property
    def tracking_time(self):
        return self._tracking_time

    def __init__(self, config, cache_to_disk=False, keep_frames=True, calc_stats=True, verbose=False, scale=None,
                 do_tracking=True, on_trapped=None, update_background=True, trap_size="L", tracking_alg="mog2",
                 check_trapped=False, from_pi=False, device=0, max_frames=4096):
        super().__init__(config, cache_to_disk, keep_frames, calc_stats, verbose, do_tracking=do_tracking, scale=scale)
        self._factor = None
        if scale:
            # the foreground is detected on cv2.resize(filtered, (int(res_x * scale), int(res_y * scale)), INTER_AREA)
            # (irtrackextractor.py:445-451): built for the integer ratios (the reference's only caller passes 0.25,
            # piclassifier.py:225), where INTER_AREA is the block mean (cpx_ir_resize_area)
            f = 1.0 / float(scale)
            if abs(f - round(f)) > 1e-9 or round(f) < 1 or round(f) > 16:
                raise NotImplementedError("scale %r: only 1 / integer (0.5, 0.25, ...) is built" % (scale,))
            self._factor = int(round(f))
        if tracking_alg != "mog2":
            raise NotImplementedError("tracking_alg %r: only the MOG2 background model is built (SuBSENSE is pybgs)"
                                      % tracking_alg)
        self.version = f"PI-IR-{IRTrackExtractor.VERSION}" if from_pi else f"IR-{IRTrackExtractor.VERSION}"
        self.check_trapped = check_trapped
        self.tracking_alg = tracking_alg
        self.on_trapped = on_trapped
        self.saliency = None
        self.background = None
        self.res_x = None
        self.res_y = None
        self.update_background = update_background
        self.trap_size = trap_size
        self.left_bottom, self.right_bottom = get_trap_lines(self.trap_size)
        self.learning_rate = -1
        self.device = device
        self.capacity = int(max_frames)
        self._engine = None
        self._stream = None
        self._ring = []      # the last FRAMES_AGO + 1 frames on the device (number, tensor)
        self._tracks = {}

    # ---- file / frame-list drivers ------------------------------------------------------------------
    def parse_clip(self, clip, process_background=False):
        """irtrackextractor.py:166-231.  Needs cv2.VideoCapture for the MP4 container."""
        try:
            import cv2
        except ImportError:
            raise NotImplementedError("MP4 decoding is cv2's (not part of this build): decode the recording elsewhere "
                                      "and call parse_frames(clip, gray_frames)") from None
        vidcap = cv2.VideoCapture(str(clip.source_file))

        def frames():
            fail_count = 0
            while True:
                success, image = vidcap.read()
                if not success:
                    if fail_count < 1:
                        fail_count += 1
                        continue
                    break
                fail_count = 0
                grey = bool(np.all(image[:, :, 0] == image[:, :, 1]) and np.all(image[:, :, 1] == image[:, :, 2]))
                yield cv2.cvtColor(image, cv2.COLOR_BGR2GRAY), grey
        try:
            return self._parse(clip, frames())
        finally:
            vidcap.release()

    def parse_frames(self, clip, frames, first_is_background=False):
        """parse_clip for already decoded gray uint8 [H, W] frames.  first_is_background: the first frame is the
        recorder's tracking background (a grey-scale frame in the MP4: 500 background frames, not tracked)."""
        return self._parse(clip, ((f, first_is_background and i == 0) for i, f in enumerate(frames)))

    def _parse(self, clip, frames):
        clip.type = self.type
        self._tracking_time = None
        start = time.time()
        clip.set_frame_buffer(False, self.cache_to_disk, False, self.keep_frames,
                              max_frames=None if self.keep_frames else 51)
        background = None
        for gray, is_background_frame in frames:
            gray = np.ascontiguousarray(gray, dtype=np.uint8)
            if background is None:
                self.res_x, self.res_y = gray.shape[0], gray.shape[1]  # (sic, irtrackextractor.py:199-200)
                clip.set_res(gray.shape[1], gray.shape[0])
                if clip.from_metadata:
                    for track in clip.tracks:
                        track.crop_regions()
                background = gray
                self.start_tracking(clip, background_frame=gray, background_frames=500 if is_background_frame else 1)
                if is_background_frame:
                    continue
            self.process_frame(clip, gray)
        if not clip.from_metadata and self.do_tracking:
            self.apply_track_filtering(clip)
        if self.calc_stats:
            clip.stats.completed()
        self._tracking_time = time.time() - start
        return True

    def parse_frames_batch(self, clips, videos, calc_stats=None):
        """parse_frames for V videos as ONE device batch: the videos advance in lockstep through
        the MOG2 model (cpx_mog2_apply, V streams), detect_objects_ir (cpx_ir_detect), the fragment merge and the
        per-region variance of the frame difference (cpx_ir_merge) -- four launches per frame step for all videos, no
        host synchronisation inside the walk -- and ONE association call (cpx_associate_batch with the IR tracking
        parameters) follows the last step.  Host work afterwards, per clip: the Track / Region objects, the trap
        geometry replayed over every track's history, the end-of-clip trim.
        videos: uint8 arrays [T_v, H, W], or one uint8 device tensor [T, V, H, W] (all the same H x W; lengths may differ: a shorter video idles on its last
        frame and those steps are not recorded).  Same tracks as parse_frames clip by clip
        (tests/test_irtrack_gpu.py), with the region variances in float32 (the device's component record) instead of
        float64."""
        import ctypes as C

        from .._lib import IR_FRAME_STATS_DTYPE, CpxError
        from ..engine import AssocBatchResult

        start = time.time()
        if len(videos) == 0:
            return True
        if self.scale:   # (the lockstep batch merges on the device at full resolution: a scaled tracker walks its videos one by one)
            if hasattr(videos, "data_ptr"):
                videos = [videos[:, v].cpu().numpy() for v in range(int(videos.shape[1]))]
            return all(self.parse_frames(clip, np.asarray(video)) for clip, video in zip(clips, videos))
        resident = hasattr(videos, "data_ptr")   # a uint8 device tensor [T, V, H, W]: frames already in HBM
        if resident:
            T, V, H, W = (int(x) for x in videos.shape)
            lens = [T] * V
        else:
            V = len(videos)
            H, W = (int(v) for v in np.asarray(videos[0]).shape[1:])
            lens = [int(len(v)) for v in videos]
            T = max(lens)
        if len(clips) != V:
            raise ValueError("one clip per video")
        calc_stats = self.calc_stats if calc_stats is None else calc_stats
        if self._engine is not None:
            self._engine.close()
        eng = self._engine = TrackEngine(width=160, height=120, device=self.device, max_components=256,
                                         max_frames=max(T, 1024))
        t, dev, lib, h = eng.torch, eng.device, eng.lib, eng.h
        for clip in clips:
            clip.type = self.type
            clip.set_frame_buffer(False, self.cache_to_disk, False, False, max_frames=51)
            clip.set_res(W, H)
            clip.set_model("IR")
            clip.set_video_stats(datetime.now())
            clip.calc_stats = bool(calc_stats)
        self.res_x, self.res_y = H, W   # (sic, irtrackextractor.py:199-200)
        if resident:
            video = videos.contiguous()
        else:
            # frames resident as [T, V, H, W]: step t reads one contiguous [V, H, W] slab
            host = np.zeros((T, V, H, W), np.uint8)
            for v, vid in enumerate(videos):
                a = np.ascontiguousarray(vid, dtype=np.uint8)
                host[: lens[v], v] = a
                host[lens[v]:, v] = a[-1]
            video = t.from_numpy(host).to(dev)
            del host
        bg = MOG2Background(eng, W, H, n_streams=V, history=1000)
        cap_det, cap = 4096, eng.cap
        mask = t.empty((V, H, W), dtype=t.uint8, device=dev)
        det = t.empty((V, cap_det, 8), dtype=t.int32, device=dev)
        counts = t.zeros((T, V), dtype=t.int32, device=dev)
        dstatus = t.zeros((T, V), dtype=t.int32, device=dev)
        mstatus = t.zeros((T, V), dtype=t.int32, device=dev)
        comps = t.zeros((V * T, cap, 8), dtype=t.int32, device=dev)
        info = t.zeros((V * T, 20), dtype=t.int32, device=dev)
        stats = hist = None
        if calc_stats:   # one 32-byte cpx_ir_frame_stats record per (step, video)
            stats = t.empty((T, V, 32), dtype=t.uint8, device=dev)
            hist = t.empty((V, 256), dtype=t.int32, device=dev)
        p = lambda x: C.c_void_p(x.data_ptr())

        def check(rc):
            if rc != 0:
                raise CpxError(rc, eng._err())

        t.cuda.current_stream(dev).synchronize()
        with t.cuda.stream(eng.torch_stream()):
            check(lib.cpx_mog2_apply(bg._m, p(video[0]), 1.0, p(mask)))   # start_tracking: the first frame seeds the model
            for q in range(T):
                cur = video[q]
                check(lib.cpx_mog2_apply(bg._m, p(cur), float(self.learning_rate), p(mask)))
                if q == 0:   # clip.background is the model's image after the first frame (irtrackextractor.py:415-416)
                    first_background = bg.background.clone()
                check(lib.cpx_ir_detect(h, p(mask), V, W, H, 0, cap_det, p(det), p(counts[q]), p(dstatus[q]), None))
                # get_delta_frame (irtrackextractor.py:638-659): the frame FRAMES_AGO back, frame 1 before that
                prev_i = q - 1 if q < self.FRAMES_AGO else self.FRAMES_AGO
                want = q - prev_i
                prev = video[want] if (prev_i != q and want != q and 0 <= want < q and q - want <= self.FRAMES_AGO) else None
                check(lib.cpx_ir_merge(h, p(det), p(counts[q]), V, cap_det, cap, p(cur), p(prev) if prev is not None else None,
                                       W, H, q, T, p(comps), p(info), p(mstatus[q])))
                if stats is not None:
                    check(lib.cpx_ir_frame_statistics(h, p(cur), p(mask), V, H * W, p(hist), p(stats[q])))
            eng.synchronize()
        bad = t.nonzero((dstatus != 0) | (mstatus != 0))
        if bad.numel():
            q, v = (int(x) for x in bad[0])
            raise CpxError(-5, "video %d, frame %d: more components than the detection / merge capacity" % (v, q))
        cfg = self.config
        c0 = clips[0]
        params = make_track_params(c0.res_x, c0.res_y, cfg.edge_pixels, cfg.frame_padding, self.min_dimension,
                                   cfg.cropped_regions_strategy, cfg.filter_regions_pre_match, cfg.aoi_min_mass,
                                   cfg.aoi_pixel_variance, cfg.params, c0.frames_per_second)
        offs = (np.arange(V + 1, dtype=np.int64) * T).astype(np.int32)
        meta = eng.make_meta(V * T)
        for v in range(V):   # the steps a shorter video idled through are not frames of it
            meta["background_frame"][v * T + lens[v]:(v + 1) * T] = 1

        class _Res:  # what associate_batch reads of a track result
            comps_dev, info_dev = comps.view(-1), info.view(-1)

        assoc = eng.associate_batch(_Res, offs, meta, params=params, want_regions=True)
        assoc.check()
        if stats is not None:
            st = stats.cpu().numpy().view(IR_FRAME_STATS_DTYPE).reshape(T, V)
        background = first_background.cpu().numpy()
        P = H * W
        for v, clip in enumerate(clips):
            clip.set_background(background[v] if V > 1 else background)
            for q in range(lens[v]):
                sv = None
                if stats is not None:
                    rec = st[q, v]
                    sv = (np.uint8(rec["min"]), np.uint8(rec["max"]), np.float64(rec["median_x2"]) / 2.0, rec["sum"] / P,
                          int(rec["filtered_sum"]))
                clip.ffc_affected = False
                clip.add_frame(None, None, None, False, stats=sv)
            # (Region objects of the per-frame region lists are made when somebody reads them: thousands per clip,
            # and most callers only want the tracks)
            clip.region_history = _LazyRegionHistory(assoc, v * T, lens[v])
            tracks = []
            for rec, regs in assoc.clip_tracks(v):
                track = Track.from_device(clip, rec, regs, self.tracker_version, self.config)
                for r in track.bounds_history:
                    r.centroid = [int(r.centroid[0]), int(r.centroid[1])]
                track._velocities_from_history()
                track.direction, track.trap_reported = 0, False
                tracks.append(track)
            clip.tracks = tracks
            self._replay_trap(clip)
            last = clip.current_frame
            clip.active_tracks = set(tr for tr in clip.tracks if tr.end_frame == last and self._active(tr.tracker.frames,
                                                                                                   tr.frames_since_target_seen))
            if not clip.from_metadata and self.do_tracking:
                self.apply_track_filtering(clip)
            if calc_stats:
                clip.stats.completed()
        bg.close()
        self._tracking_time = (time.time() - start) / V
        return True

    @staticmethod
    def _active(frames, since):
        return since == 0 or since < min(2 * (frames - since), 18)

    def _replay_trap(self, clip):
        """The per-frame trap test of _process_frame (irtrackextractor.py:470-490) replayed over finished tracks: at
        every frame a track was active, inside_trap_top on its bound of that frame and -- once the last two bounds are
        inside -- filter_track on its statistics so far; the first frame that passes is the trigger frame."""
        for track in clip.tracks:
            full, vx, vy = track.bounds_history, track.vel_x, track.vel_y
            rt = track.tracker
            saved = (rt.frames, rt._blank_frames, rt._frames_since_target_seen, rt._last_bound)
            since = blanks = 0
            try:
                for i, region in enumerate(full):
                    since = since + 1 if region.blank else 0
                    blanks += 1 if region.blank else 0
                    if not self._active(i + 1, since) or track.trap_reported:
                        continue
                    track.bounds_history, track.vel_x, track.vel_y = full[: i + 1], vx[: i + 1], vy[: i + 1]
                    rt.frames, rt._blank_frames, rt._frames_since_target_seen, rt._last_bound = i + 1, blanks, since, region
                    self.inside_trap_top(track, self.scale)
                    if track.in_trap and not self.filter_track(clip, track, track.get_stats()):
                        track.trigger_frame = region.frame_number
                        if self.on_trapped is not None:
                            track.trap_reported = True
                            self.on_trapped(track)
            finally:
                track.bounds_history, track.vel_x, track.vel_y = full, vx, vy
                rt.frames, rt._blank_frames, rt._frames_since_target_seen, rt._last_bound = saved

    def start_tracking(self, clip, frames=None, track_frames=-1, background_alg=None, background_frame=None,
                       background_frames=1, retrack_back=True):
        """irtrackextractor.py:233-283."""
        self.res_x, self.res_y = clip.res_x, clip.res_y
        clip.set_model("IR")
        clip.set_video_stats(datetime.now())
        self._open(clip)
        if background_alg is None:
            self.background = MOG2Background(self._engine, clip.res_x, clip.res_y, history=1000)
            if background_frame is not None:
                self.background.set_background(self._upload(background_frame), background_frames)
        else:
            self.background = background_alg
        if frames is not None:
            do_tracking, update_background = self.do_tracking, self.update_background
            remaining = len(frames)
            for frame in frames:
                self.do_tracking = do_tracking and ((track_frames == -1) or (remaining <= track_frames))
                self.learning_rate = 0
                self.update_background = self.do_tracking and retrack_back
                self.process_frame(clip, frame)
                remaining -= 1
            self.learning_rate = -1
            self.update_background = update_background
            self.do_tracking = do_tracking

    def _open(self, clip):
        if self._engine is not None:
            self._engine.close()
        # the IR calls take their geometry per call; the handle only carries the component capacity and the stream
        self._engine = TrackEngine(width=160, height=120, device=self.device, max_components=256,
                                   max_frames=max(self.capacity, 1024))
        cfg = self.config
        params = make_track_params(clip.res_x, clip.res_y, cfg.edge_pixels, cfg.frame_padding, self.min_dimension,
                                   cfg.cropped_regions_strategy, cfg.filter_regions_pre_match, cfg.aoi_min_mass,
                                   cfg.aoi_pixel_variance, cfg.params, clip.frames_per_second)
        self._stream = ComponentStream(self._engine, self.capacity, params)
        self._ring = []
        self._tracks = {}

    def _upload(self, frame):
        t = self._engine.torch
        return t.from_numpy(np.array(frame, dtype=np.uint8, order="C")).to(self._engine.device)   # (a copy: memory-mapped .npy frames are read-only)

    def process_frame(self, clip, frame, ffc_affected=False):
        """irtrackextractor.py:295-312."""
        frame = np.asarray(frame)
        if frame.ndim == 3:
            raise NotImplementedError("colour frames: convert to gray first (cv2.cvtColor is not part of this build)")
        if ffc_affected:
            self.print_if_verbose("{} ffc_affected".format(clip.current_frame))
        clip.ffc_affected = ffc_affected
        self._process_frame(clip, frame, ffc_affected)

    def merge_components(self, rectangles):
        return merge_components(rectangles, self.scale)

    # ---- one frame -----------------------------------------------------------------------------------
    def _process_frame(self, clip, frame, ffc_affected=False):
        """irtrackextractor.py:391-492."""
        if self._stream is None:
            self._open(clip)
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        frame_dev = self._upload(frame)
        filtered_dev = None
        if self.do_tracking:
            if self.background is None:
                self.background = MOG2Background(self._engine, clip.res_x, clip.res_y, history=1000)
            if self.background._background is None:
                self.background.set_background(frame_dev.clone())
            if self.update_background:
                self.background.update_background(frame_dev, learning_rate=self.learning_rate)
            filtered_dev = self.background.compute_filtered(frame_dev)
            if not clip.background_calculated:
                clip.set_background(self.background.background.cpu().numpy())
        filtered = filtered_dev.cpu().numpy() if (filtered_dev is not None and self.keep_frames) else None
        clip.add_frame(frame, filtered, None, ffc_affected)
        q = clip.current_frame
        self._ring.append((q, frame_dev))
        del self._ring[: -(self.FRAMES_AGO + 1)]
        if not self.do_tracking:
            return
        re_f = filtered_dev
        if self.scale and self._factor > 1:
            if clip.res_x % self._factor or clip.res_y % self._factor:
                raise NotImplementedError("scale %r does not divide a %d x %d frame" % (self.scale, clip.res_x, clip.res_y))
            # cpx_ir_detect packs 64 pixels per word: zero columns on the right up to a multiple of 64 change nothing
            # (the open's element is vertical -- the (15, 15) tuple quirk, SURVEY F3 -- and zeros join no component)
            re_f = self._engine.ir_resize_area(filtered_dev, self._factor, pad_width_to=64)
        _, _, stats = detect_objects_ir(self._engine, re_f, threshold=0, max_components=4096)
        component_details = self.merge_components(list(stats[1:]))   # (thresholds scaled: irdetect.merge_components)
        if clip.from_metadata:
            return  # tracks come from the metadata: nothing is matched (the reference re-reads their frames only)
        regions = []
        if ffc_affected:
            clip.active_tracks = set()
            self._stream.append(np.zeros(0, COMPONENT_DTYPE), ffc_affected=True)
        else:
            regions = self._associate(clip, component_details, q)
        for track in clip.active_tracks:
            if getattr(track, "trap_reported", False):
                continue
            self.inside_trap_top(track, self.scale)
            if track.in_trap:
                if not self.filter_track(clip, track, track.get_stats()):
                    track.trigger_frame = q
                    if self.on_trapped is not None:   # fire trapping event
                        track.trap_reported = True
                        self.on_trapped(track)
        clip.region_history.append(regions)

    def get_delta_frame(self, clip):
        """irtrackextractor.py:638-659 -> (frame number compared with, its device frame) or (None, None)."""
        cur = clip.current_frame
        prev_i = cur - 1 if cur < self.FRAMES_AGO else self.FRAMES_AGO
        want = cur - prev_i          # "prev_i frames before the current one"
        if prev_i == cur or want == cur:
            return None, None
        for q, dev in self._ring:
            if q == want:
                return q, dev
        return None, None

    def _associate(self, clip, component_details, q):
        """_get_regions_of_interest with the IR branch (track/cliptracker.py:263-365) + _apply_region_matchings, on
        the device association core; -> the frame's regions."""
        n = len(component_details)
        comps = np.zeros(n, COMPONENT_DTYPE)
        variances = np.zeros(n, np.float64)
        if n:
            # with a scale the components are in the down-scaled image: Region.rescale(1 / scale) (region.py:44-50,
            # cliptracker.py:297-298) truncates x, y, width, height times the factor and multiplies the mass by its
            # square; the centroid was taken from the box BEFORE that and is left where it was (as in the reference)
            f = float(1 / self.scale) if self.scale else 1.0
            cents = [(int(c[0] + c[2] / 2), int(c[1] + c[3] / 2)) for c in component_details]
            if self.scale:
                component_details = [[int(c[0] * f), int(c[1] * f), int(c[2] * f), int(c[3] * f), int(c[4] * f * f)]
                                     for c in component_details]
            rects = np.array([[int(c[0]), int(c[1]), int(c[2]), int(c[3])] for c in component_details], np.int32)
            _, prev_dev = self.get_delta_frame(clip)
            if prev_dev is not None:
                variances = self._engine.ir_delta_variance(self._ring[-1][1], prev_dev, rects)
            for i, c in enumerate(component_details):
                x, y, w, h, mass = (int(v) for v in c[:5])
                cx, cy = cents[i]     # the IR tracker's centroid: the box centre, truncated
                comps[i] = (x, y, w, h, mass, cx * mass, cy * mass, np.float32(variances[i]))
        f = self._stream.append(comps, ffc_affected=False)
        stream = self._stream
        regions = []
        for rec in stream.frame_regions(f):
            r = Region.from_record(rec)
            r.centroid = [int(r.centroid[0]), int(r.centroid[1])]
            r.pixel_variance = variances[int(rec["id"])]      # float64, as np.var returns it
            regions.append(r)
        records = stream.track_records()
        row = stream.pool_row(f)
        active = set()
        for rec in records:
            tid = int(rec["id"])
            if int(rec["start_frame"]) + int(rec["n_frames"]) - 1 != f:
                continue  # not touched by this frame
            track = self._tracks.get(tid)
            if track is None:
                track = Track(clip.get_id(), id=tid, fps=clip.frames_per_second, tracking_config=self.config,
                              crop_rectangle=clip.crop_rectangle, tracker_version=self.tracker_version)
                track.start_frame = int(rec["start_frame"])
                track.start_s = track.start_frame / float(clip.frames_per_second)
                track.direction, track.trap_reported = 0, False
                self._tracks[tid] = track
                clip.tracks.append(track)
            track.append_from_device(rec, row[int(rec["slot"])])
            rt = track.tracker
            since = rt.frames_since_target_seen
            if since == 0 or since < min(2 * (rt.frames - since), 18):
                active.add(track)
        clip.active_tracks = active
        return regions

    # ---- trap geometry and filters (host, as in the reference) -----------------------------------------
    def filter_components(self, component_details):
        """irtrackextractor.py:564-594."""
        kept = []
        for component in component_details:
            region = Region(component[0], component[1], component[2], component[3])
            p = (region.right, 480 - region.bottom)
            flt = self.left_bottom.is_above(p) and self.left_bottom.is_left(p)
            p = (region.left, 480 - region.bottom)
            flt = flt or (self.right_bottom.is_above(p) and self.right_bottom.is_right(p))
            if not flt:
                kept.append(component)
            else:
                logging.info("Filtered components %s", region)
        return kept

    def filter_track(self, clip, track, stats):
        """irtrackextractor.py:596-636."""
        if len(track) < self.config.min_duration_secs * clip.frames_per_second:
            self.print_if_verbose("Track filtered. Too short, {}".format(len(track)))
            clip.filtered_tracks.append(("Track filtered.  Too short", track))
            return True
        if stats.max_offset < self.config.track_min_offset or stats.frames_moved < self.config.min_moving_frames:
            self.print_if_verbose("Track filtered.  Didn't move {}".format(stats.max_offset))
            clip.filtered_tracks.append(("Track filtered.  Didn't move", track))
            return True
        return False

    def filter_tracks(self, clip):
        """The IR tracker's end of clip: trim only (track/cliptracker.py:367-371 as IRTrackExtractor inherits it)."""
        for track in clip.tracks:
            track.trim()
            track.set_end_s(clip.frames_per_second)
        return False

    # --- trap geometry (reference behaviour: irtrackextractor.py:660-778, pinned by tests/golden/irtrap_golden.json) ---
    # The picture is 640 x 480 with y growing downwards; the two trap walls are lines in a y-up frame, so a box corner
    # (x, y_picture) is probed at (x, 480 - y_picture).  Both tests share three steps: (1) boxes under 60 x 40 are not
    # judged; (2) the first judged box fixes the side the track came from; (3) two corners of the box are probed against
    # the wedge between the walls.  They differ in the margins of step 2, in which corners are probed, and in the verdict.
    PICTURE_W, PICTURE_H = 640, 480
    MIN_JUDGED_W, MIN_JUDGED_H = 60, 40

    @staticmethod
    def _entry_side(box, side_margin, bottom_margin=100, top_limit=300):
        """Step 2: the Direction bits of the picture edges the box reaches into, else TOP / MIDDLE by how low it hangs."""
        touched = ((box.left < side_margin, Direction.LEFT),
                   (box.right > IRTrackExtractor.PICTURE_W - side_margin, Direction.RIGHT),
                   (box.bottom > IRTrackExtractor.PICTURE_H - bottom_margin, Direction.BOTTOM))
        bits = 0
        for reaches, bit in touched:
            if reaches:
                bits |= bit
        if bits:
            return bits
        return Direction.TOP if box.bottom < top_limit else Direction.MIDDLE

    def _judged_box(self, track, side_margin):
        """Steps 1 and 2 -> a copy of the track's last box, or None when it is too small to judge."""
        box = track.last_bound.copy()
        if box.width < self.MIN_JUDGED_W or box.height < self.MIN_JUDGED_H:
            return None
        if track.direction == 0:
            track.direction = self._entry_side(box, side_margin)
        return box

    def _wall_probes(self, x_at_left_wall, x_at_right_wall, y_picture):
        """Step 3 -> (both probes inside the wedge, |x distance of each probe to its wall|).  A probe is inside when it is
        not above its wall and on the wedge's side of it (a probe exactly on the left wall is inside, one exactly on the
        right wall is not: Line.is_left is the strict comparison)."""
        y = self.PICTURE_H - y_picture
        pl, pr = (x_at_left_wall, y), (x_at_right_wall, y)
        lw, rw = self.left_bottom, self.right_bottom
        inside = (not lw.is_above(pl)) and (not lw.is_left(pl)) and (not rw.is_above(pr)) and rw.is_left(pr)
        return inside, abs(pl[0] - lw.x_res(y)), abs(pr[0] - rw.x_res(y))

    def inside_trap_bottom(self, track, scale=None):
        """The box's BOTTOM corners sit inside the wedge and its right one is more than 150 px from the right wall."""
        box = self._judged_box(track, side_margin=100)
        if box is None:
            return False
        inside, _, to_right_wall = self._wall_probes(box.left, box.right, box.bottom)
        inside = inside and to_right_wall > 150
        track.last_bound.in_trap = inside
        track.update_trapped_state()
        return inside

    def inside_trap_top(self, track, scale=None):
        """The box's TOP corners, crossed over (its right corner against the left wall, its left one against the right
        wall), sit inside the wedge; the verdict then depends on the side the track came from."""
        box = self._judged_box(track, side_margin=150)
        if box is None:
            return False
        inside, to_left_wall, to_right_wall = self._wall_probes(box.right, box.left, box.top)
        share_l, share_r = to_left_wall / box.width, to_right_wall / box.width
        if not inside or (share_l < 0.5 and share_r < 0.5):
            return False                       # (the track's trapped state is left as it was, as the reference leaves it)
        came = track.direction
        clear_l, clear_r = box.left > 40, box.right < 580
        verdicts = (
            bool(came & Direction.LEFT) and clear_l and share_l > 0.5,
            bool(came & Direction.RIGHT) and clear_r and share_r > 0.5,
            came == Direction.TOP and box.bottom > 300,
            came == Direction.BOTTOM and box.bottom < self.PICTURE_H - 50,
            came == Direction.MIDDLE and clear_l and clear_r,
        )
        in_trap = any(verdicts)
        track.last_bound.in_trap = in_trap
        track.update_trapped_state()
        return in_trap

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
