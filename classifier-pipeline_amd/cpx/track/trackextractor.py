"""TrackExtractor / extract_file -- file-level drivers of the track stage
(reference src/track/trackextractor.py:25-251), including the thumbnail entries
of the metadata (classify/thumbnail.py via the HIP thumbnail kernels)."""

import json
import logging
import os
from pathlib import Path

from ..ml_tools import tools
from ..sharding import rank_world, shard_files
from .clip import Clip
from .cliptrackextractor import ClipTrackExtractor


class TrackExtractor:
    def __init__(self, config, cache_to_disk=None, retrack=False):
        self.config = config
        self.worker_threads = max(1, config.worker_threads)
        self.retrack = retrack
        self.cache_to_disk = config.classify.cache_to_disk if cache_to_disk is None else cache_to_disk
        self.batch_files = None  # files per device batch of extract(directory); None: bulk.auto_batch_files
        self.last_run = None     # timings of the last extract(directory) (cpx.track.bulk.BulkTracker.timings)
        # metadata worker processes (bulk.MetaPool) for directories at least this large (their start costs seconds);
        # None = never
        self.meta_pool_min_files = 8192   # (track-only metadata is 0.07 ms per recording in-process: workers pay late)

    def extract(self, base, to_stdout=False):
        base = Path(base)
        if not base.exists():
            logging.error("Could not find file or directory %s", base)
            return
        if base.is_file():
            extract_file(base, self.config, self.cache_to_disk, self.retrack, to_stdout)
            return
        # one GPU per process (the reference forks a pool of CPU workers, trackextractor.py:60-120): the files of a
        # directory go through the device in batches -- decoded, tracked and associated together
        todo, gray, videos = [], [], []
        for folder, _, files in os.walk(base):
            for name in sorted(files):
                ext = os.path.splitext(name)[1]
                if ext == ".cptv":
                    todo.append(os.path.join(folder, name))
                elif ext in (".y4m",):  # raw-gray IR recordings (cpx/track/grayvideo.py): one at a time
                    gray.append(os.path.join(folder, name))
                elif ext in (".mp4", ".avi"):
                    videos.append(os.path.join(folder, name))
        if videos:
            # the reference walks these too (trackextractor.py:72-76) and decodes them with cv2.VideoCapture; no video
            # decoder is built here (out of scope) -- convert to .y4m / .npy gray frames first
            logging.warning("%d .mp4 / .avi recordings skipped (no video decoder in this build): %s ...", len(videos),
                            videos[0])
        # under torchrun (one process per GPU) every rank takes its share of the files and its own device; each file's
        # metadata is written by the rank that tracked it, no collective is needed
        rank, world, local_rank = rank_world()
        todo = shard_files(todo, rank, world)
        device = local_rank if world > 1 else 0
        if world > 1:  # the reader / staging threads started below stay on the CPUs next to this rank's GPU
            from ..sharding import pin_to_gpu_numa

            logging.info("rank %d: %s", rank, pin_to_gpu_numa(local_rank))
        # a large directory: the metadata text of the file-fed path is formatted by worker processes.  They are spawned
        # children that never touch the GPU; they are started here, BEFORE this process's first GPU work (the .y4m
        # recordings below run on the device), so that nothing of HIP exists in the parent when they start
        pool = None
        if not self.retrack and self.meta_pool_min_files is not None and len(todo) >= self.meta_pool_min_files:
            from .bulk import MetaPool

            pool = MetaPool.make()
        for path in shard_files(gray, rank, world):
            try:
                extract_file(path, self.config, self.cache_to_disk, self.retrack, to_stdout)
            except Exception:  # noqa: BLE001 -- one bad recording does not stop a directory run
                logging.exception("%s failed", path)
        if self.retrack:  # existing tracks are re-used per file: no batch form
            step = self.batch_files or 1024
            for i in range(0, len(todo), step):
                extract_files(todo[i:i + step], self.config, self.cache_to_disk, self.retrack, to_stdout,
                              device=device)
            return
        # the file-fed path at device speed (cpx/track/bulk.py): gzip inflate, section index, decode, tracking,
        # end-of-clip statistics and thumbnails on the device for `batch_files` recordings at a time, the next batch
        # read from disk meanwhile; a recording that fails is logged, retried on its own and otherwise skipped
        from .bulk import extract_files_bulk

        try:
            _, tracker = extract_files_bulk(todo, self.config, to_stdout=to_stdout, device=device,
                                            batch_files=self.batch_files, meta_pool=pool)
        finally:
            if pool is not None:
                pool.close()
        self.last_run = tracker.timings


def extract_file(filename, config, cache_to_disk, retrack=False, to_stdout=False, max_frames=None, save_meta=True,
                 blob=None):
    """trackextractor.py:122-202.  blob: the recording's bytes when it lives in memory (`filename` then only names
    it in the metadata): the host reader (zlib + section walk) decodes it, as it does a file."""
    filename = Path(filename)
    if blob is None and not filename.is_file():
        raise Exception("File {} not found.".format(filename))
    logging.info("Tracking %s", filename)
    if filename.suffix != ".cptv":
        return extract_ir_file(filename, config, cache_to_disk, retrack, to_stdout, save_meta)
    track_extractor = ClipTrackExtractor(config.tracking, config.use_opt_flow, cache_to_disk, verbose=config.verbose,
                                         max_frames=max_frames)
    clip = Clip(track_extractor.config, filename)
    clip.frames_per_second = 9
    if blob is not None:
        clip.source_bytes = blob
    existing = None
    meta_filename = filename.with_suffix(".txt")
    if blob is None and meta_filename.exists():
        existing = tools.load_clip_metadata(meta_filename)
    if retrack:
        clip.load_metadata(existing)
    if not track_extractor.parse_clip(clip):
        logging.error("Could not parse %s", filename)
        return
    if retrack:
        for track in clip.tracks:
            track.trim()
            track.set_end_s(clip.frames_per_second)
    metadata = get_metadata(existing, filename, meta_filename, clip, track_extractor, to_stdout, save_meta)
    return clip, track_extractor, metadata


def extract_ir_file(filename, config, cache_to_disk, retrack=False, to_stdout=False, save_meta=True):
    """The other branch of extract_file (trackextractor.py:148-156): every recording that is not a .cptv goes to the
    IR tracker at 10 frames per second.  The reference decodes the container with cv2.VideoCapture and subtracts the
    background with SuBSENSE (pybgs); here the containers are the raw-gray ones of cpx/track/grayvideo.py (.npy, .y4m;
    an MP4 goes through IRTrackExtractor.parse_clip where cv2 exists) and the background model is MOG2 -- the one the
    reference's Pi runs and the one built on the device (include/cpx.h: cpx_mog2_*; cv2 parity unpinned)."""
    from .grayvideo import GRAY_SUFFIXES, read_gray_frames
    from .irtrackextractor import IRTrackExtractor

    track_extractor = IRTrackExtractor(config.tracking, cache_to_disk, verbose=config.verbose, keep_frames=True,
                                       tracking_alg="mog2")
    clip = Clip(track_extractor.config, filename)
    clip.frames_per_second = 10
    existing = None
    meta_filename = filename.with_suffix(".txt")
    if meta_filename.exists():
        existing = tools.load_clip_metadata(meta_filename)
    if retrack:
        clip.load_metadata(existing)
    if filename.suffix in GRAY_SUFFIXES:
        frames, _ = read_gray_frames(filename)
        track_extractor.capacity = max(track_extractor.capacity, int(len(frames)) + 4)
        ok = track_extractor.parse_frames(clip, frames)
    else:
        ok = track_extractor.parse_clip(clip)   # cv2.VideoCapture (NotImplementedError without OpenCV)
    if not ok:
        logging.error("Could not parse %s", filename)
        return
    if retrack:
        for track in clip.tracks:
            track.trim()
            track.set_end_s(clip.frames_per_second)
    metadata = get_metadata(existing, filename, meta_filename, clip, track_extractor, to_stdout, save_meta)
    return clip, track_extractor, metadata


def extract_files(filenames, config, cache_to_disk, retrack=False, to_stdout=False, max_frames=None, save_meta=True,
                  device=0):
    """extract_file for a list of files as one device batch (ClipTrackExtractor.parse_clips); same metadata per file.
    -> list of (clip, track_extractor, metadata)."""
    filenames = [Path(f) for f in filenames]
    for filename in filenames:
        if not filename.is_file():
            raise Exception("File {} not found.".format(filename))
        if filename.suffix != ".cptv":
            raise NotImplementedError("only thermal .cptv clips are handled (IR path: SURVEY section 8 f4)")
    if retrack:  # existing tracks are re-used per file: no batch form
        return [extract_file(f, config, cache_to_disk, retrack, to_stdout, max_frames, save_meta) for f in filenames]
    track_extractor = ClipTrackExtractor(config.tracking, config.use_opt_flow, cache_to_disk, verbose=config.verbose,
                                         max_frames=max_frames, device=device)
    track_extractor.host_images = False  # the consumers below (thumbnails, classification) read device memory
    clips, existing = [], []
    for filename in filenames:
        logging.info("Tracking %s", filename)
        clip = Clip(track_extractor.config, filename)
        clip.frames_per_second = 9
        meta_filename = filename.with_suffix(".txt")
        existing.append(tools.load_clip_metadata(meta_filename) if meta_filename.exists() else None)
        clips.append(clip)
    track_extractor.parse_clips(clips)
    out = []
    for filename, clip, ex in zip(filenames, clips, existing):
        metadata = get_metadata(ex, filename, filename.with_suffix(".txt"), clip, track_extractor, to_stdout, save_meta)
        out.append((clip, track_extractor, metadata))
    return out


def get_metadata(existing_metadata, filename, meta_filename, clip, track_extractor, to_stdout=False, save=True):
    from ..classify.thumbnail import best_trackless_thumb, get_thumbnail_info

    metadata = clip.get_metadata()
    for i, track in enumerate(clip.tracks):
        best_thumb, best_score = get_thumbnail_info(clip, track)
        if best_thumb is None:
            metadata["tracks"][i]["thumbnail"] = None
            continue
        metadata["tracks"][i]["thumbnail"] = {
            "region": best_thumb.region,
            "contours": best_thumb.contours,
            "median_diff": best_thumb.median_diff,
            "score": round(best_score),
        }
    if len(clip.tracks) == 0:
        metadata["thumbnail_region"] = best_trackless_thumb(clip)  # if no tracks choose a clip thumb
    metadata["source"] = str(filename)
    metadata["tracking_time"] = round(track_extractor.tracking_time, 1)
    metadata["algorithm"] = {"tracker_version": track_extractor.tracker_version,
                             "tracker_config": track_extractor.config.as_dict()}
    if existing_metadata is not None:
        existing_metadata.pop("tracks", None)
        existing_metadata.pop("Tracks", None)
        existing_metadata.update(metadata)
        metadata = existing_metadata
    if to_stdout:
        print(json.dumps(metadata, cls=tools.CustomJSONEncoder))
    elif save:
        with open(meta_filename, "w") as fh:
            json.dump(metadata, fh, indent=4, cls=tools.CustomJSONEncoder)
    return metadata
