"""Shared by tests/test_config_tracks_{cpu,gpu}.py: the golden of tests/golden/make_golden_config_tracks.py (the
REFERENCE run with three non-default thermal tracking configurations) and how its scenes and configurations are rebuilt."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden():
    z = np.load(os.path.join(GOLDEN, "config_tracks.npz"))
    with open(os.path.join(GOLDEN, "config_tracks_info.json")) as fh:
        info = json.load(fh)
    return z["rows"], z["offsets"], int(z["frames"]), info["variants"], info["cases"]


def scene_frames(case, T):
    from cpx import synth
    from helpers import load_clip

    if case["scene"] == "possum":
        frames, t_on, ffc, bgf, hdr = load_clip("possum")
        return frames, t_on, ffc, bgf, hdr.model
    frames = synth.make_clip(np.random.default_rng(1000 + int(case["seed"])), T, model=case["model"], max_blobs=8)
    return frames, [100000 + 111 * i for i in range(T)], [40000] * T, None, case["model"]


def load_config(yaml_text, tmp_path, name):
    """The drop-in Config from the SAME YAML text the reference loaded."""
    from cpx.config import Config

    p = os.path.join(str(tmp_path), name + ".yaml")
    with open(p, "w") as fh:
        fh.write(yaml_text)
    return Config.load_from_file(p)


def oracle_config(cfg, model):
    """oracle/track_oracle.OracleConfig with the thermal section of a drop-in Config."""
    import track_oracle as to

    th = cfg.tracking["thermal"]
    oc = to.OracleConfig(model or "lepton3")
    oc.edge_pixels, oc.frame_padding, oc.min_dimension = th.edge_pixels, max(3, th.frame_padding), th.min_dimension
    oc.denoise = bool(th.denoise)
    oc.aoi_min_mass, oc.aoi_pixel_variance = th.aoi_min_mass, th.aoi_pixel_variance
    oc.cropped_regions_strategy = th.cropped_regions_strategy
    oc.filter_regions_pre_match = th.filter_regions_pre_match
    oc.track_min_offset, oc.track_min_mass = th.track_min_offset, th.track_min_mass
    oc.min_moving_frames, oc.max_blank_percent, oc.max_jitter = th.min_moving_frames, th.max_blank_percent, th.max_jitter
    oc.min_duration_secs, oc.max_tracks = th.min_duration_secs, th.max_tracks
    for k in ("base_distance_change", "min_mass_change", "restrict_mass_after", "mass_change_percent", "max_distance",
              "max_blanks", "velocity_multiplier", "base_velocity"):
        setattr(oc, k, th.params[k])
    return oc
