"""Segment selection for classification (reference src/ml_tools/datasetstructures.py:25-35,771-822,
972-1301).  Host side by design: the choice is random in the reference; parity runs pass
explicit ``segment_frames``."""

import logging
from enum import Enum

import numpy as np


class SegmentType(Enum):
    IMPORTANT_RANDOM = 0
    ALL_RANDOM = 1
    IMPORTANT_SEQUENTIAL = 2
    ALL_SEQUENTIAL = 3
    TOP_SEQUENTIAL = 4
    ALL_SECTIONS = 5
    TOP_RANDOM = 6
    ALL_RANDOM_NOMIN = 7
    ALL_RANDOM_MASKED = 8
    ELONGATION = 9


class SegmentHeader:
    """The frames of one classification sample."""

    def __init__(self, clip_id, track_id, start_frame, frames, weight, mass, label, regions, frame_indices=None,
                 filtered=False, **_):
        self.clip_id = clip_id
        self.track_id = track_id
        self.start_frame = start_frame
        self.frames = np.uint16(frames)
        self.weight = np.float16(weight)
        self._mass = np.uint16(mass)
        self.label = label
        self.regions = regions
        self.frame_numbers = np.uint16(frame_indices)
        self.filtered = filtered

    @property
    def frame_indices(self):
        return self.frame_numbers

    @property
    def mass(self):
        return self._mass


_RANDOM_TYPES = (SegmentType.IMPORTANT_RANDOM, SegmentType.ALL_RANDOM, SegmentType.ALL_RANDOM_NOMIN,
                 SegmentType.TOP_RANDOM, SegmentType.ALL_RANDOM_MASKED, None)


def get_segments(clip_id, track_id, start_frame, regions, segment_width=25, segment_frame_spacing=9, label=None,
                 segment_min_mass=None, ffc_frames=(), repeats=1, min_frames=None,
                 segment_types=(SegmentType.ALL_RANDOM_MASKED,), max_segments=None, dont_filter=False, skip_ffc=True,
                 frame_min_mass=None, repeat_frame_indices=True, min_segments=None, seed=None, **_):
    """Random 25-frame subsets of a track's usable frames.  ALL_RANDOM_MASKED (the default) removes each
    segment's frames from the pool and stops when fewer than half a segment remains."""
    if min_frames is None:
        min_frames = segment_width / 4.0
    regions = np.asarray(regions, dtype=object)
    segments = []
    stats = {"segment_mass": 0, "too short": 0}
    mass_history = np.uint16([r.mass for r in regions])
    has_no_mass = np.sum(mass_history) == 0
    for segment_type in segment_types:
        if segment_type not in _RANDOM_TYPES:
            raise NotImplementedError("segment type %s is a training-time selection" % segment_type)
        s_min_mass = None if segment_type == SegmentType.ALL_RANDOM_NOMIN else segment_min_mass
        usable = [r.frame_number for r in regions
                  if (has_no_mass or r.mass > 0)
                  and (ffc_frames is None or not skip_ffc or r.frame_number not in ffc_frames)
                  and not r.blank and r.width > 0 and r.height > 0
                  and (has_no_mass or frame_min_mass is None or r.mass >= frame_min_mass)]
        if not usable:
            logging.warning("Nothing to load for %s - %s", clip_id, track_id)
            return [], stats
        usable = np.array(usable)
        if s_min_mass is not None:
            s_min_mass = min(s_min_mass, np.median(mass_history[usable - start_frame]))
        else:
            s_min_mass = 1
        rng = np.random.default_rng(seed=seed)
        if segment_type == SegmentType.TOP_RANDOM:
            usable = np.array(sorted(sorted(usable, key=lambda f: mass_history[f - start_frame], reverse=True)[:50]))
        if len(usable) < min_frames and not min_segments:
            stats["too short"] += 1
            continue
        segment_count = int(max(1, len(usable) // segment_frame_spacing))
        mask_length = 25
        if max_segments is not None:
            segment_count = min(max_segments, segment_count)
            mask_length = max(mask_length, len(usable) // segment_count)
        whole = usable
        masked = segment_type == SegmentType.ALL_RANDOM_MASKED
        for _ in range(repeats):
            if masked:
                positions = np.arange(len(regions))
                all_frames = positions + start_frame
                available = np.full(len(regions), False)
                available[whole - start_frame] = True
            pool = None
            if not masked or len(whole) < 40:
                pool = whole.copy()
                rng.shuffle(pool)
            for i in range(segment_count):
                if masked:
                    if len(whole) < 40:
                        pool = positions[available]
                    else:
                        m = available.copy()
                        m[i * mask_length : (i + 1) * mask_length] = False
                        pool = np.uint32(positions[m])
                        np.random.shuffle(pool)  # (sic) the reference uses the global RNG here
                if len(pool) == 0 or min_segments is None or len(segments) >= min_segments:
                    if (len(pool) < segment_width / 2.0 and len(segments) > 0) or len(pool) < segment_width / 4:
                        break
                if masked:
                    idx = pool[:segment_width]
                    available[idx] = False
                    frames = all_frames[idx]
                else:
                    frames = pool[:segment_width]
                    pool = pool[segment_width:]
                remaining = segment_width - len(frames)
                if remaining > 0:
                    frames = np.concatenate([frames, rng.choice(frames, min(remaining, len(frames)), replace=False)])
                frames.sort()
                rel = frames - start_frame
                seg_mass = np.sum(mass_history[rel])
                avg_mass = seg_mass / len(rel)
                filtered = False
                if s_min_mass and avg_mass < s_min_mass:
                    if not dont_filter:
                        stats["segment_mass"] += 1
                        continue
                    filtered = True
                region_slice = regions[rel]
                weight = 0.75 if avg_mass < 50 else (1 if avg_mass < 100 else 1.2)
                if repeat_frame_indices and len(frames) < segment_width:
                    frames = sorted(list(frames) + list(rng.choice(frames, segment_width - len(frames))))
                segments.append(SegmentHeader(clip_id, track_id, start_frame, segment_width, weight, seg_mass, label,
                                              region_slice, frame_indices=frames, filtered=filtered))
    return segments, stats


def segments_from_frames(clip_id, track_id, start_frame, regions, segment_frames):
    """Track.get_segments with explicit frame lists (track.py:508-526)."""
    regions = np.asarray(regions, dtype=object)
    mass_history = np.uint16([r.mass for r in regions])
    out = []
    for frames in segment_frames:
        frames = np.asarray(frames)
        rel = frames - start_frame
        out.append(SegmentHeader(clip_id, track_id, start_frame, len(frames), 1, np.sum(mass_history[rel]), None,
                                 regions[rel], frame_indices=frames))
    return out
