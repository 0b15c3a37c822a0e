"""Host side of the association stage: parameter marshalling for
cpx_associate_batch and decoding of its outputs (cpx_region / cpx_track_record)."""

import ctypes as C

import numpy as np

REGION_BLANK, REGION_CROPPED, REGION_BORDER, REGION_CENTROID_F32 = 1, 2, 4, 8

REGION_DTYPE = np.dtype(
    [("x", "<i4"), ("y", "<i4"), ("width", "<i4"), ("height", "<i4"), ("mass", "<i4"),
     ("frame_number", "<i4"), ("pixel_variance", "<f4"), ("flags", "<i4"), ("cx", "<f8"), ("cy", "<f8"),
     ("id", "<i4"), ("pad", "<i4")]
)
TRACK_RECORD_DTYPE = np.dtype(
    [("id", "<i4"), ("slot", "<i4"), ("start_frame", "<i4"), ("n_frames", "<i4"), ("blank_frames", "<i4"),
     ("since_seen", "<i4"), ("rt_frames", "<i4"), ("track_index", "<i4")]
)
assert REGION_DTYPE.itemsize == 56 and TRACK_RECORD_DTYPE.itemsize == 32


class TrackParams(C.Structure):
    _fields_ = [
        ("crop_x", C.c_int32), ("crop_y", C.c_int32), ("crop_w", C.c_int32), ("crop_h", C.c_int32),
        ("frame_padding", C.c_int32), ("min_dimension", C.c_int32),
        ("cropped_regions_strategy", C.c_int32), ("filter_regions_pre_match", C.c_int32),
        ("aoi_min_mass", C.c_double), ("aoi_pixel_variance", C.c_double),
        ("base_distance_change", C.c_double), ("min_mass_change", C.c_double),
        ("restrict_mass_after", C.c_double), ("mass_change_percent", C.c_double),
        ("velocity_multiplier", C.c_double), ("base_velocity", C.c_double),
        ("has_min_mass_change", C.c_int32), ("has_mass_change_percent", C.c_int32),
        ("max_blanks", C.c_int32), ("fps", C.c_int32),
        ("max_active_tracks", C.c_int32), ("max_tracks", C.c_int32),
    ]


assert C.sizeof(TrackParams) == 120

_STRATEGY = {"cautious": 0, "none": 1, None: 1, "all": 2}


def make_track_params(width=160, height=120, edge_pixels=1, frame_padding=4, min_dimension=0,
                      cropped_regions_strategy="cautious", filter_regions_pre_match=True, aoi_min_mass=4.0,
                      aoi_pixel_variance=2.0, params=None, fps=9, max_active_tracks=16, max_tracks=128):
    """Defaults = config/trackingconfig.py:126-177 (thermal)."""
    p = dict(base_distance_change=450, min_mass_change=20, restrict_mass_after=1.5, mass_change_percent=0.55,
             max_distance=2000, max_blanks=18, velocity_multiplier=2, base_velocity=2)
    if params:
        p.update(params)
    if cropped_regions_strategy not in _STRATEGY:
        raise ValueError(
            "Invalid mode for CROPPED_REGIONS_STRATEGY, expected ['all','cautious','none'] but found {}".format(
                cropped_regions_strategy))
    e = edge_pixels
    return TrackParams(
        e, e, width - 2 * e, height - 2 * e, frame_padding, min_dimension,
        _STRATEGY[cropped_regions_strategy], 1 if filter_regions_pre_match else 0,
        float(aoi_min_mass), float(aoi_pixel_variance),
        float(p["base_distance_change"]), float(p["min_mass_change"] or 0.0), float(p["restrict_mass_after"]),
        float(p["mass_change_percent"] or 0.0), float(p["velocity_multiplier"]), float(p["base_velocity"]),
        0 if p["min_mass_change"] is None else 1, 0 if p["mass_change_percent"] is None else 1,
        int(p["max_blanks"]), int(fps), int(max_active_tracks), int(max_tracks))


def track_regions(pool, rec, max_active):
    """Region records of one track: pool is the clip's [n_frames * max_active] array."""
    idx = (int(rec["start_frame"]) + np.arange(int(rec["n_frames"]))) * max_active + int(rec["slot"])
    return pool[idx]


TRACK_SUMMARY_DTYPE = np.dtype(
    [("id", "<i4"), ("slot", "<i4"), ("start_frame", "<i4"), ("n_frames", "<i4"), ("blank_frames", "<i4"),
     ("since_seen", "<i4"), ("reject", "<i4"), ("rank", "<i4"), ("frames_moved", "<i4"), ("region_jitter", "<i4"),
     ("jitter_bigger", "<i4"), ("jitter_smaller", "<i4"), ("blank_percent", "<i4"), ("n_segments", "<i4"),
     ("movement", "<f8"), ("max_offset", "<f8"), ("score", "<f8"), ("average_mass", "<f8"), ("median_mass", "<f8"),
     ("delta_std", "<f8"), ("mass_std", "<f8"), ("average_velocity", "<f8")]
)
assert TRACK_SUMMARY_DTYPE.itemsize == 120

REJECT_REASONS = {
    1: "Track filtered.  Too short", 2: "Track filtered.  Didn't move", 3: "Track filtered. Too Many Blanks",
    4: "Track filtered.  Too Jittery", 5: "Track filtered.  Too static", 6: "Track filtered.  Too Dynamic",
    7: "Track filtered.  Mass too small", 8: "Too many tracks",
}


class FilterParams(C.Structure):
    _fields_ = [
        ("min_duration_secs", C.c_double), ("track_min_offset", C.c_double), ("track_min_mass", C.c_double),
        ("track_min_delta", C.c_double), ("track_max_delta", C.c_double),
        ("min_moving_frames", C.c_int32), ("max_blank_percent", C.c_int32), ("max_jitter", C.c_int32),
        ("fps", C.c_int32), ("max_tracks", C.c_int32), ("max_active_tracks", C.c_int32),
        ("max_tracks_per_clip", C.c_int32), ("reserved", C.c_int32),
    ]


assert C.sizeof(FilterParams) == 72


def make_filter_params(min_duration_secs=0, track_min_offset=4.0, track_min_mass=2.0, track_min_delta=1.0,
                       track_max_delta=150, min_moving_frames=2, max_blank_percent=30, max_jitter=20, fps=9,
                       max_tracks=None, max_active_tracks=16, max_tracks_per_clip=128):
    """Defaults = config/trackingconfig.py:126-177 and trackingmotionconfig.py:24-59 (thermal)."""
    return FilterParams(float(min_duration_secs), float(track_min_offset), float(track_min_mass),
                        float(track_min_delta), float(track_max_delta), int(min_moving_frames),
                        int(max_blank_percent), int(max_jitter), int(fps), -1 if max_tracks is None else int(max_tracks),
                        int(max_active_tracks), int(max_tracks_per_clip), 0)
