"""ClipTrackExtractor -- drop-in for the reference class of the same name
(reference src/track/cliptrackextractor.py:34-247, src/track/cliptracker.py:14-491).

parse_clip() inflates the CPTV container on the host, decodes the frame payloads on
the GPU (cpx_cptv_unpack) and runs the clip through the HIP track stage
(cpx_track_batch) and the HIP association stage
(cpx_associate_batch); the Clip / Frame / Track / Region objects the
reference's callers read are then built from the device records.  End-of-clip
filtering (trim, statistics, score ordering, rejects) is host work as in the
reference.  There is no CPU implementation of the per-frame arithmetic."""

import logging
import time
from datetime import datetime

import numpy as np

from .._lib import TRACK_FREEZE_BACKGROUND
from ..cptv import CptvReader, decode_clips_on_device
from ..engine import TrackEngine
from ..ml_tools.rectangle import Rectangle
from ..tracking import make_track_params
from .clip import Clip
from .region import Region
from .track import Track

_ENGINES = {}


def get_engine(width, height, background_thresh, weight_add, edge_pixels=1, device=0, max_frames=4096,
               max_components=64, denoise=False, lane=0):
    """One device engine per (geometry, thresholds, denoise) in this process -- per `lane`: a handle is one HIP stream and
    one tracking workspace, so two threads that drive the device at the same time (the bulk path's two device lanes) take
    a lane each."""
    key = (width, height, float(background_thresh), float(weight_add), edge_pixels, device, max_components,
           bool(denoise)) + ((lane,) if lane else ())
    eng = _ENGINES.get(key)
    if eng is None or eng.cfg.max_frames < max_frames:
        if eng is not None:
            eng.close()
        eng = TrackEngine(width=width, height=height, device=device, edge_pixels=edge_pixels,
                          background_thresh=background_thresh, weight_add=weight_add,
                          max_components=max_components, max_frames=max(max_frames, 1024), denoise=denoise)
        _ENGINES[key] = eng
    return eng


def default_engine(device=0):
    """Any live engine on `device` (the CNN kernels do not depend on the tracking thresholds)."""
    for key, eng in _ENGINES.items():
        if key[5] == device:
            return eng
    return get_engine(160, 120, 20.0, 0.1, device=device)


class DeviceClipState:
    """Device-resident data of a tracked clip that the classifier reads: the frames, the filtered
    frames and per-frame medians (cpx_frame_info) -- nothing is copied back for classification."""

    def __init__(self, engine, frames_dev, track_result, proc_frames, first_frame=0):
        self.first_frame = first_frame  # index of the file's first frame (the clip background) in frames_dev
        self.engine = engine
        self.frames_dev = frames_dev
        self.track_result = track_result
        self._index = {q: f for q, f in enumerate(proc_frames)}

    def frame_index(self, frame_number):
        return self._index.get(int(frame_number))


class WeightedBackgroundView:
    """What callers read from ``extractor.background_alg`` after tracking
    (reference piclassifier/motiondetector.py:178-248): the final background and its average."""

    def __init__(self, background, average, weight_add):
        self._background = background
        self.average = average
        self.weight_add = weight_add

    @property
    def background(self):
        return self._background

    def get_average(self):
        return self.average

    @property
    def background_weight(self):
        """The model's float64 weights after parse_clip (motiondetector.py:218-222): three blocking device-to-host
        copies, made only when a caller asks (post_process_file hands them on)."""
        src = getattr(self, "_weights_from", None)
        if src is None:
            raise AttributeError("background_weight")
        return src.final_state()[1]


class StreamBackgroundView:
    """extractor.background_alg during incremental tracking: the device's current background, fetched on demand."""

    def __init__(self, stream, frame_info, weight_add):
        self._stream = stream
        self.average = float(frame_info["background_average"])
        self.weight_add = weight_add

    @property
    def background(self):
        return self._stream.background[0].cpu().numpy().astype(np.float64)

    def get_average(self):
        return self.average


class ClipTracker:
    """Configuration handling + end-of-clip filtering shared by the extractors (cliptracker.py:14-491)."""

    def __init__(self, config, cache_to_disk=False, keep_frames=True, calc_stats=True, verbose=False,
                 do_tracking=True, scale=None, calculate_thumbnail_info=False, max_frames=None):
        self.max_frames = max_frames
        config = config.get(self.type)
        self.scale = scale
        self.calculate_thumbnail_info = calculate_thumbnail_info
        self.do_tracking = do_tracking
        self.verbose = verbose
        self.config = config
        self.stats = None
        self.cache_to_disk = cache_to_disk
        self.max_tracks = config.max_tracks
        self.frame_padding = max(3, self.config.frame_padding)
        self.keep_frames = keep_frames
        self.calc_stats = calc_stats
        self._tracking_time = None
        self.min_dimension = config.min_dimension
        self.background_alg = None

    def print_if_verbose(self, info_string):
        if self.verbose:
            logging.info(info_string)

    def apply_track_filtering(self, clip):
        filtered_tracks = self.filter_tracks(clip)
        if self.config.track_smoothing and clip.current_frame > 0:
            for track in clip.active_tracks:
                track.smooth(Rectangle(0, 0, clip.res_x, clip.res_y))
        return filtered_tracks

    def filter_tracks(self, clip):
        for track in clip.tracks:
            track.trim()
            track.set_end_s(clip.frames_per_second)
        for track in clip.tracks:
            track.calculate_stats()
        clip.tracks.sort(reverse=True, key=lambda t: t.stats.score)
        good, rejected = [], []
        for track in clip.tracks:
            (rejected if self.filter_track(clip, track) else good).append(track)
        clip.tracks = good
        if self.max_tracks is not None and self.max_tracks < len(clip.tracks):
            logging.warning(" -using only %s tracks out of %s", self.max_tracks, len(clip.tracks))
            clip.filtered_tracks.extend(("Too many tracks", t) for t in clip.tracks[self.max_tracks:])
            clip.tracks = clip.tracks[: self.max_tracks]
        return rejected

    def filter_track(self, clip, track):
        """True (and a reason on clip.filtered_tracks) when the track is noise (cliptracker.py:422-486)."""
        s, cfg = track.stats, self.config
        reason = None
        if len(track) < cfg.min_duration_secs * clip.frames_per_second:
            reason = "Track filtered.  Too short"
        elif s.max_offset < cfg.track_min_offset or s.frames_moved < cfg.min_moving_frames:
            reason = "Track filtered.  Didn't move"
        elif s.blank_percent > cfg.max_blank_percent:
            reason = "Track filtered. Too Many Blanks"
        elif s.region_jitter > cfg.max_jitter:
            reason = "Track filtered.  Too Jittery"
        elif s.delta_std < clip.track_min_delta:
            reason = "Track filtered.  Too static"
        elif s.delta_std > clip.track_max_delta:
            reason = "Track filtered.  Too Dynamic"
        elif s.average_mass < cfg.track_min_mass:
            reason = "Track filtered.  Mass too small"
        if reason is None:
            return False
        self.print_if_verbose("{} (track {})".format(reason, track.get_id()))
        clip.filtered_tracks.append((reason, track))
        return True


class ClipTrackExtractor(ClipTracker):
    PREVIEW = "preview"
    VERSION = 11
    TYPE = "thermal"

    @property
    def tracker_version(self):
        return self.version

    @property
    def type(self):
        return ClipTrackExtractor.TYPE

    @property
    def tracking_time(self):
        return self._tracking_time

    def __init__(self, config, use_opt_flow, cache_to_disk=False, keep_frames=True, calc_stats=True,
                 high_quality_optical_flow=False, verbose=False, do_tracking=True, update_background=True,
                 calculate_filtered=False, calculate_thumbnail_info=False, from_pi=False, max_frames=None,
                 device=0):
        super().__init__(config, cache_to_disk, keep_frames=keep_frames, calc_stats=calc_stats, verbose=verbose,
                         do_tracking=do_tracking, calculate_thumbnail_info=calculate_thumbnail_info,
                         max_frames=max_frames)
        if use_opt_flow:
            raise NotImplementedError("optical flow is outside the cpx hot path")
        self.version = f"PI-{ClipTrackExtractor.VERSION}" if from_pi else ClipTrackExtractor.VERSION
        self.use_opt_flow = use_opt_flow
        self.high_quality_optical_flow = high_quality_optical_flow
        self.update_background = update_background
        self.calculate_filtered = calculate_filtered
        self.weighting_percent = 1
        self.device = device
        self.host_images = True  # parse_clips: copy thermal / filtered / mask images into the frame buffer
        self._stream = None
        self._external_bg = None   # a caller-owned background model (start_tracking(background_alg=...))
        self._imported = None      # (background, average) last handed to the device from it
        self._meta = None          # cpx_frame_meta of the clip parse_clip tracked (post_process_file re-runs it)
        self._final_state = None   # WeightedBackground state (background, weights, average) after parse_clip
        self._frames = None
        self._frames_dev = None
        self._engine = None
        self._weight_add = None
        self._header = None
        self.timings = {}

    # ---- reference API -------------------------------------------------------------------------
    def _read_header(self, clip):
        """Open the file, take resolution / model / start time from its header (cliptrackextractor.py:98-127) and
        index its frame sections.  -> dict(reader, frames (metadata only), offsets, widths, weight_add)."""
        clip.set_frame_buffer(self.high_quality_optical_flow, self.cache_to_disk, self.use_opt_flow,
                              self.keep_frames, self.max_frames)
        clip.type = self.type
        blob = getattr(clip, "source_bytes", None)  # an in-memory recording (trackextractor.extract_file(blob=...))
        reader = CptvReader(blob if blob is not None else str(clip.source_file))
        header = reader.get_header()
        clip.set_res(header.x_resolution, header.y_resolution)
        if clip.from_metadata:
            for track in clip.tracks:
                track.crop_regions()
        clip.set_model(header.model if header.model else None)
        start = datetime.fromtimestamp(header.timestamp / 1000000).astimezone(Clip.local_tz)
        clip.set_video_stats(start)
        cam35 = clip.camera_model == "lepton3.5"
        weight_add = (1 if cam35 else 0.1) / self.weighting_percent
        frames, offsets, widths = reader.scan()
        if not frames:
            raise Exception("CPTV file has no frames: {}".format(clip.source_file))
        return dict(reader=reader, header=header, frames=frames, offsets=offsets, widths=widths, weight_add=weight_add)

    def init_clip(self, clip):
        """Header, resolution, camera thresholds, first frame -> clip background (cliptrackextractor.py:98-139)."""
        item = self._read_header(clip)
        reader, frames = item["reader"], item["frames"]
        self._header = item["header"]
        self._weight_add = item["weight_add"]
        eng = get_engine(clip.res_x, clip.res_y, clip.background_thresh, self._weight_add, self.config.edge_pixels,
                         self.device, max_frames=len(frames), denoise=bool(self.config.denoise))
        t0 = time.time()
        self._frames_dev = eng.cptv_unpack(np.frombuffer(reader.inflated + bytes(16), np.uint8), item["offsets"],
                                           item["widths"], np.array([0, len(frames)], np.int32))
        pix = self._frames_dev.cpu().numpy().view(np.uint16)
        self.timings["decode_s"] = time.time() - t0
        for f, p in zip(frames, pix):
            f.pix = p
        self._frames = frames
        self._engine = eng
        clip.update_background(self._frames[0].pix)
        clip._background_calculated()

    def parse_clip(self, clip, process_background=False):
        self._tracking_time = None
        start = time.time()
        self.init_clip(clip)
        self._track_clip(clip, process_background=process_background)
        if self.calc_stats:
            clip.stats.completed()
        self._tracking_time = time.time() - start
        return True

    def parse_clips(self, clips, process_background=False):
        """parse_clip for many files at once: the files are inflated in threads, and decoded, tracked and associated
        on the GPU as ONE batch per camera geometry / threshold group; the per-clip objects (tracks, regions,
        statistics, end-of-clip filtering) are then built exactly as parse_clip builds them.  With host_images=False
        the frame buffer gets its entries (frame count, flags) but no host copies of the images: thumbnails and
        classification read the frames, filtered frames and label masks on the device."""
        from concurrent.futures import ThreadPoolExecutor

        self._tracking_time = None
        start = time.time()
        clips = list(clips)
        if not clips:
            return True
        with ThreadPoolExecutor(max_workers=min(8, len(clips))) as pool:
            items = list(pool.map(self._read_header, clips))
        groups = {}
        for i, (clip, item) in enumerate(zip(clips, items)):
            key = (clip.res_x, clip.res_y, float(clip.background_thresh), float(item["weight_add"]))
            groups.setdefault(key, []).append(i)
        for key, members in groups.items():
            longest = max(len(items[i]["frames"]) for i in members)
            eng = get_engine(key[0], key[1], key[2], key[3], self.config.edge_pixels, self.device,
                             max_frames=longest, denoise=bool(self.config.denoise))
            t0 = time.time()
            offs, base, chunks, poffs, widths = [0], 0, [], [], []
            for i in members:
                it = items[i]
                chunks.append(np.frombuffer(it["reader"].inflated, np.uint8))
                poffs.append(it["offsets"] + base)
                widths.append(it["widths"])
                base += len(it["reader"].inflated)
                offs.append(offs[-1] + len(it["frames"]))
            offs = np.asarray(offs, np.int32)
            frames_dev = eng.cptv_unpack(np.concatenate(chunks + [np.zeros(16, np.uint8)]), np.concatenate(poffs),
                                         np.concatenate(widths), offs)
            host_images = self.keep_frames and self.host_images
            if host_images:
                pix_all = frames_dev.cpu().numpy().view(np.uint16)
            else:  # only the first frame of every clip is needed on the host (clip.background)
                firsts = frames_dev[eng.torch.from_numpy(offs[:-1].astype(np.int64)).to(frames_dev.device)]
                pix_first = firsts.cpu().numpy().view(np.uint16)
            self.timings["decode_s"] = time.time() - t0
            metas = []
            for k, i in enumerate(members):
                fr = items[i]["frames"]
                bgf = [bool(f.background_frame) and not process_background for f in fr]
                metas.append(eng.make_meta(len(fr), [f.time_on for f in fr], [f.last_ffc_time for f in fr], bgf))
                clips[i].update_background(pix_all[offs[k]] if host_images else pix_first[k])
                clips[i]._background_calculated()
            meta = np.concatenate(metas)
            t0 = time.time()
            track_flags = 0 if self.update_background else TRACK_FREEZE_BACKGROUND
            res = eng.track_batch(frames_dev, offs, meta, want_labels=True, want_filtered=True, want_background=True,
                                  flags=track_flags)
            assoc = params = None
            if self.do_tracking and not any(clips[i].from_metadata for i in members):
                c0 = clips[members[0]]
                params = make_track_params(
                    c0.res_x, c0.res_y, self.config.edge_pixels, self.config.frame_padding, self.min_dimension,
                    self.config.cropped_regions_strategy, self.config.filter_regions_pre_match,
                    self.config.aoi_min_mass, self.config.aoi_pixel_variance, self.config.params, c0.frames_per_second)
                assoc = eng.associate_batch(res, offs, meta, params=params)
            # a clip with a frame of more components, or with more tracks, than the batch's tables hold is run again
            # alone on tables grown to fit (TrackEngine.track_clip_grown); the others keep the batch's results
            over = res.overflowed(offs)
            regrow = set(over) | (set(assoc.overflowed()) if assoc is not None else set())
            self.timings["device_s"] = time.time() - t0
            labels = res.labels() if host_images else None
            filtered = res.filtered() if (host_images or self.calculate_filtered) else None
            backgrounds = res.background()
            for k, i in enumerate(members):
                f0, n = int(offs[k]), int(offs[k + 1] - offs[k])
                thermal = pix_all[f0:f0 + n] if host_images else None
                if k in regrow:
                    logging.info("%s: %s; tracking it again with larger tables", clips[i].source_file,
                                 ("%d components in a frame" % over[k]) if k in over else "track capacity exceeded")
                    fr_k = frames_dev[f0:f0 + n]
                    eng_k, res_k, assoc_k, _ = eng.track_clip_grown(
                        fr_k, meta[f0:f0 + n], params=params, need_components=over.get(k, 0), want_labels=True,
                        want_filtered=True, want_background=True, flags=track_flags, associate=assoc is not None)
                    self._collect_clip(clips[i], 0, 0, n, thermal, eng_k, fr_k, res_k, assoc_k,
                                       res_k.labels() if host_images else None,
                                       res_k.filtered() if (host_images or self.calculate_filtered) else None,
                                       res_k.background()[0], key[3])
                    continue
                self._collect_clip(clips[i], k, f0, n, thermal, eng, frames_dev, res, assoc, labels, filtered,
                                   backgrounds[k], key[3])
                if self.calc_stats:
                    clips[i].stats.completed()
        self._tracking_time = (time.time() - start) / len(clips)
        return True

    def start_tracking(self, clip, frames, track_frames=True, background_alg=None, **args):
        """Feed `frames` one by one (cliptrackextractor.py:181-193); with track_frames=False they only build the
        frame buffer.  background_alg: a caller-owned background model (the Pi motion detector's WeightedBackground,
        piclassifier.py:423-431) -- any object with .background ([H, W], integer-valued) and .average; the extractor
        reads it before every frame and never updates it, exactly as the reference's process_frame does."""
        do_tracking = self.do_tracking
        self._external_bg = background_alg
        self._imported = None
        self.background_alg = background_alg
        self.do_tracking = self.do_tracking and track_frames
        new_tracks = []
        try:
            for frame in frames:
                new_tracks.extend(self.process_frame(clip, frame))
        finally:
            self.do_tracking = do_tracking
        return new_tracks

    def process_frame(self, clip, frame):
        """One frame through the device: the reference's process_frame (cliptrackextractor.py:195-247).  Who updates
        the background afterwards depends on who owns it: with a caller-owned model (start_tracking(background_alg=),
        the Pi loop) or update_background=False nobody here does -- the device reads the model as the owner left it;
        otherwise the device applies the update _track_clip would (45-frame mean, cliptrackextractor.py:169-176).
        The first call opens the stream: the clip background (clip.update_background, else this frame) seeds the
        model.  Returns the tracks created by this frame."""
        st = self._stream
        if st is None or st["clip"] is not clip:
            st = self._open_stream(clip, frame)
        stream = st["stream"]
        flags = 0
        ext = self._external_bg
        if ext is not None:
            flags = TRACK_FREEZE_BACKGROUND
            bg = np.asarray(ext.background)
            avg = float(ext.average)
            last = self._imported
            if last is None or last[1] != avg or not np.array_equal(last[0], bg):
                # the model is frozen (TRACK_FREEZE_BACKGROUND): the device neither reads nor updates the weights, so
                # only background and average cross -- a detector whose weight_add differs from the extractor's
                # (or whose weights were adjusted while recording) must not make the frame loop raise
                stream.engine.set_background(0, bg, None, avg)
                self._imported = (np.array(bg, copy=True), avg)
        elif not self.update_background:
            flags = TRACK_FREEZE_BACKGROUND
        f = stream.append(frame.pix, frame.time_on, frame.last_ffc_time, associate=self.do_tracking, flags=flags)
        fi = stream.frame_info(f)
        if int(fi["status"]) != 0 or (self.do_tracking and int(stream.status.item()) != 0):
            stream = self._regrow_stream(st, flags)
            fi = stream.frame_info(f)
        thermal = np.array(frame.pix, dtype=np.uint16, copy=True)
        P = clip.res_x * clip.res_y
        stats = (np.uint16(fi["thermal_min"]), np.uint16(fi["thermal_max"]), np.float64(fi["thermal_median"]),
                 fi["thermal_sum"] / P, float(fi["filtered_abs_sum"]))
        clip.ffc_affected = bool(fi["ffc_affected"])
        if clip.ffc_affected:
            self.print_if_verbose("{} ffc_affected".format(clip.current_frame))
        filtered = stream.filtered[f].cpu().numpy() if (self.keep_frames or self.calculate_filtered) else None
        mask = stream.labels[f].cpu().numpy() if (self.keep_frames and stream.labels is not None) else None
        clip.add_frame(thermal, filtered, mask, clip.ffc_affected, stats=stats)
        st["device_state"]._index[clip.current_frame] = f
        clip.device_state = st["device_state"]
        self.background_alg = ext if ext is not None else StreamBackgroundView(stream, fi, st["weight_add"])
        if not self.do_tracking:
            return []
        new_tracks = []
        if not clip.from_metadata:
            if clip.ffc_affected:
                clip.active_tracks = set()
                clip.region_history.append([])
                return []
            clip.region_history.append([Region.from_record(r) for r in stream.frame_regions(f)])
            q = int(fi["frame_number"])
            records = stream.track_records()
            row = stream.pool_row(q)
            by_id = st["tracks"]
            active = set()
            for rec in records:
                tid = int(rec["id"])
                last = int(rec["start_frame"]) + int(rec["n_frames"]) - 1
                if last != q:
                    continue  # not touched by this frame
                track = by_id.get(tid)
                if track is None:
                    track = Track(clip.get_id(), id=tid, fps=clip.frames_per_second, tracking_config=self.config,
                                  crop_rectangle=clip.crop_rectangle, tracker_version=self.tracker_version)
                    track.start_frame = int(rec["start_frame"])
                    track.start_s = track.start_frame / float(clip.frames_per_second)
                    by_id[tid] = track
                    clip.tracks.append(track)
                    new_tracks.append(track)
                track.append_from_device(rec, row[int(rec["slot"])])
                if self._still_tracking(track):
                    active.add(track)
            clip.active_tracks = active
        return new_tracks

    def _open_stream(self, clip, frame):
        if clip.res_x is None or clip.res_y is None:
            clip.set_res(frame.pix.shape[1], frame.pix.shape[0])
        if clip.background_thresh is None:
            clip.set_model(None)
        if clip.frame_buffer is None:
            clip.set_frame_buffer(self.high_quality_optical_flow, self.cache_to_disk, self.use_opt_flow,
                                  self.keep_frames, self.max_frames)
        if clip.background is None:
            clip.update_background(np.array(frame.pix, dtype=np.uint16, copy=True))
            clip._background_calculated()
        cam35 = clip.camera_model == "lepton3.5"
        weight_add = (1 if cam35 else 0.1) / self.weighting_percent
        capacity = (self.max_frames or 2047) + 1
        # a stream's background / window / association state lives in its handle's workspace, so every stream gets a
        # handle of its own: two extractors (or two live clips) fed in lockstep can never resume on each other's state
        if self._stream is not None:
            self._stream["engine"].close()
            self._stream = None
        eng = TrackEngine(width=clip.res_x, height=clip.res_y, device=self.device, edge_pixels=self.config.edge_pixels,
                          background_thresh=clip.background_thresh, weight_add=weight_add, max_components=64,
                          max_frames=max(capacity, 1024), denoise=bool(self.config.denoise))  # (grown on demand: _regrow_stream)
        params = make_track_params(
            clip.res_x, clip.res_y, self.config.edge_pixels, self.config.frame_padding, self.min_dimension,
            self.config.cropped_regions_strategy, self.config.filter_regions_pre_match, self.config.aoi_min_mass,
            self.config.aoi_pixel_variance, self.config.params, clip.frames_per_second)
        stream = eng.open_stream(capacity, params, want_labels=True)
        self._imported = None
        stream.append(clip.background, init_only=True)  # slot 0 seeds the background model, it is not tracked
        self._stream = dict(clip=clip, stream=stream, tracks={}, weight_add=weight_add, engine=eng,
                            device_state=DeviceClipState(eng, stream.frames_dev, stream.result, []))
        return self._stream

    def _regrow_stream(self, st, flags):
        """A frame with more components, or a clip with more tracks, than the open stream's tables hold (the reference
        has no such limit): a new stream with tables grown to fit takes over the frames consumed so far and runs them
        again, so the clip stands where it stood, on larger tables.  A caller-owned background model cannot be replayed
        (its past states are gone): that case reports the overflow."""
        from ..engine import CpxError
        from ..tracking import TrackParams

        if self._external_bg is not None:
            raise CpxError(-5, "a frame exceeds the stream's component / track capacity and the background model is "
                               "caller-owned: its earlier states cannot be replayed")
        cur, cur_eng = st["stream"], st["engine"]
        while True:
            need, tracks_full = cur.overflow()
            if need <= cur_eng.cap and not tracks_full:
                break
            params = cur.params
            if tracks_full:
                params = TrackParams.from_buffer_copy(params)
                params.max_active_tracks, params.max_tracks = cur.params.max_active_tracks * 2, cur.params.max_tracks * 2
            eng = cur_eng.sibling(cur_eng.grown_capacity(max(need, cur_eng.cap)), max_frames=cur.cap_frames)
            new = eng.open_stream(cur.cap_frames, params, want_labels=True)
            new.replay(cur, flags=flags, associate=self.do_tracking)
            cur_eng.close()  # (a stream owns its handle: the outgrown one goes)
            cur, cur_eng = new, eng
        index = st["device_state"]._index
        st["stream"], st["engine"] = cur, cur_eng
        st["device_state"] = DeviceClipState(cur_eng, cur.frames_dev, cur.result, [])
        st["device_state"]._index = index
        logging.info("stream tables grown to %d components, %d / %d tracks", cur_eng.cap,
                     cur.params.max_active_tracks, cur.params.max_tracks)
        return cur

    # ---- device path --------------------------------------------------------------------------------
    def _track_clip(self, clip, process_background=False):
        if clip.background is None:
            raise Exception("Clip has no background have you called init_clip first")
        frames = self._frames
        n = len(frames)
        eng = self._engine
        t0 = time.time()
        bgf = [bool(f.background_frame) and not process_background for f in frames]
        meta = eng.make_meta(n, [f.time_on for f in frames], [f.last_ffc_time for f in frames], bgf)
        offs = np.array([0, n], np.int32)
        want_images = self.keep_frames
        frames_dev = self._frames_dev
        # update_background = False: the model stays as init_clip seeded it (cliptrackextractor.py:169)
        track_flags = 0 if self.update_background else TRACK_FREEZE_BACKGROUND
        res = eng.track_batch(frames_dev, offs, meta, want_labels=want_images, want_filtered=True,
                              want_background=True, flags=track_flags)
        self._meta = meta
        assoc = params = None
        if self.do_tracking and not clip.from_metadata:
            params = make_track_params(
                clip.res_x, clip.res_y, self.config.edge_pixels, self.config.frame_padding, self.min_dimension,
                self.config.cropped_regions_strategy, self.config.filter_regions_pre_match, self.config.aoi_min_mass,
                self.config.aoi_pixel_variance, self.config.params, clip.frames_per_second)
            assoc = eng.associate_batch(res, offs, meta, params=params)
        over = res.overflowed(offs)
        if over or (assoc is not None and assoc.overflowed()):
            # more components in a frame, or more tracks, than the tables hold: the reference has no such limit
            # (cliptrackextractor.py:236-247, cliptracker.py:202-247) -- run the clip again on tables grown to fit
            logging.info("%s: %s; tracking it again with larger tables", clip.source_file,
                         ("%d components in a frame" % over[0]) if over else "track capacity exceeded")
            eng, res, assoc, params = eng.track_clip_grown(
                frames_dev, meta, params=params, need_components=over.get(0, 0), want_labels=want_images,
                want_filtered=True, want_background=True, flags=track_flags, associate=assoc is not None)
        self._final_state = None   # read on demand (final_state), after the stream has drained anyway
        self._final_engine = eng
        self._final_call = eng.track_calls
        self.timings["device_s"] = time.time() - t0
        labels = res.labels() if want_images else None
        filtered = res.filtered() if (want_images or self.calculate_filtered) else None
        thermal = [f.pix for f in frames] if frames and frames[0].pix is not None else None
        self._collect_clip(clip, 0, 0, n, thermal, eng, frames_dev, res, assoc, labels, filtered,
                           res.background()[0], self._weight_add)

    def _collect_clip(self, clip, b, f0, n, thermal, eng, frames_dev, res, assoc, labels, filtered, background,
                      weight_add):
        """Build what the reference's callers read for clip `b` of a tracked batch (its frames are the batch
        frames [f0, f0 + n)): frame buffer / statistics, region history, tracks, end-of-clip filtering."""
        info = res.info
        P = clip.res_x * clip.res_y
        proc = []
        for i in range(n):
            f = f0 + i
            fi = info[f]
            if fi["frame_number"] < 0:
                continue
            proc.append(f)
            stats = (np.uint16(fi["thermal_min"]), np.uint16(fi["thermal_max"]), np.float64(fi["thermal_median"]),
                     fi["thermal_sum"] / P, float(fi["filtered_abs_sum"]))
            clip.ffc_affected = bool(fi["ffc_affected"])
            clip.add_frame(None if thermal is None else thermal[i], None if filtered is None else filtered[f],
                           None if labels is None else labels[f], clip.ffc_affected, stats=stats)
            if assoc is not None:
                clip.region_history.append([] if clip.ffc_affected else
                                           [Region.from_record(r) for r in assoc.frame_regions(f)])
        clip.device_state = DeviceClipState(eng, frames_dev, res, proc, first_frame=f0)
        last = info[proc[-1]] if proc else None
        self.background_alg = WeightedBackgroundView(np.asarray(background, dtype=np.float64),
                                                     None if last is None else last["background_average"], weight_add)
        if getattr(self, "_final_engine", None) is eng and b == 0 and n == len(self._frames or ()):
            self.background_alg._weights_from = self  # background_weight is fetched when somebody reads it
        if assoc is not None:
            assoc.check(b)
            clip.tracks = [Track.from_device(clip, rec, regs, self.tracker_version, self.config)
                           for rec, regs in assoc.clip_tracks(b)]
            last_frame = clip.current_frame
            clip.active_tracks = set(t for t in clip.tracks if t.end_frame == last_frame and self._still_tracking(t))
            self.apply_track_filtering(clip)

    def final_state(self):
        """(background, weights, average) the model held after parse_clip's last frame (cpx_get_background)."""
        if self._final_state is None:
            if getattr(self, "_final_engine", None) is None:
                return None
            if self._final_engine.track_calls != self._final_call:
                raise RuntimeError("the device engine has tracked another clip since this one: read the background "
                                   "weights (extractor.final_state()) before tracking the next clip")
            self._final_state = self._final_engine.get_background(0)
        return self._final_state

    def close(self):
        """Release the device handle of an open frame-by-frame stream."""
        if self._stream is not None:
            self._stream["engine"].close()
            self._stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _still_tracking(track):
        rt = track.tracker
        if rt.frames_since_target_seen == 0:
            return True
        return rt.frames_since_target_seen < min(2 * (rt.frames - rt.frames_since_target_seen), 18)
