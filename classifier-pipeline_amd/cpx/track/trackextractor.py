"""TrackExtractor / extract_file -- file-level drivers of the track stage
(reference src/track/trackextractor.py:25-251), including the thumbnail entries
of the metadata (classify/thumbnail.py via the HIP thumbnail kernels)."""

import json
import logging
import os
from pathlib import Path

from ..ml_tools import tools
from .clip import Clip
from .cliptrackextractor import ClipTrackExtractor


class TrackExtractor:
    def __init__(self, config, cache_to_disk=None, retrack=False):
        self.config = config
        self.worker_threads = max(1, config.worker_threads)
        self.retrack = retrack
        self.cache_to_disk = config.classify.cache_to_disk if cache_to_disk is None else cache_to_disk

    def extract(self, base, to_stdout=False):
        base = Path(base)
        if not base.exists():
            logging.error("Could not find file or directory %s", base)
            return
        if base.is_file():
            extract_file(base, self.config, self.cache_to_disk, self.retrack, to_stdout)
            return
        # one GPU per process: files of a directory are walked in this process, one device engine reused
        for folder, _, files in os.walk(base):
            for name in sorted(files):
                if os.path.splitext(name)[1] == ".cptv":
                    extract_file(os.path.join(folder, name), self.config, self.cache_to_disk, self.retrack, to_stdout)


def extract_file(filename, config, cache_to_disk, retrack=False, to_stdout=False, max_frames=None, save_meta=True):
    filename = Path(filename)
    if not filename.is_file():
        raise Exception("File {} not found.".format(filename))
    logging.info("Tracking %s", filename)
    if filename.suffix != ".cptv":
        raise NotImplementedError("only thermal .cptv clips are handled (IR path: SURVEY section 8 f4)")
    track_extractor = ClipTrackExtractor(config.tracking, config.use_opt_flow, cache_to_disk, verbose=config.verbose,
                                         max_frames=max_frames)
    clip = Clip(track_extractor.config, filename)
    clip.frames_per_second = 9
    existing = None
    meta_filename = filename.with_suffix(".txt")
    if meta_filename.exists():
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
