#!/usr/bin/env python3
"""Generates tests/golden/config_tracks.npz / config_tracks_info.json by RUNNING THE REFERENCE (under
oracle/refharness.py) with tracking configurations that move every thermal knob of
config/trackingconfig.py:126-177 off its default -- the defaults are what every other golden exercises:

  cropped_regions_strategy none / all, min_dimension, filter_regions_pre_match false, edge_pixels, frame_padding,
  max_tracks, filters.track_overlap_ratio (read by nothing at this snapshot: cliptracker.py:479 is commented out),
  min_duration_secs, track_min_offset, track_min_mass, min_moving_frames, max_blank_percent, max_jitter,
  areas_of_interest.min_mass / pixel_variance, the RegionTracker `params` block, and the lepton3.5 thresholds.

Scenes: three seeded busy scenes (cpx.synth, the generator of make_golden_busy.py), the reference's own possum.cptv, and
a seeded lepton3.5 busy scene.  For every (variant, scene): every track the reference created -- kept or filtered --
with all bounds, the kept order, the reject reasons.  Seeds where the reference's same-frame track births are not in
component order (it iterates a set) are skipped.  The YAML texts travel inside the JSON so that the tests
load the SAME configuration into the drop-in Config.

Build container only:   python tests/golden/make_golden_config_tracks.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in ("oracle", "tests", "classifier-pipeline_amd"):
    sys.path.insert(0, os.path.join(REPO, p))
import refharness as rh  # noqa: E402
from cpx import synth  # noqa: E402
from helpers import encode_cptv  # noqa: E402

T = 110

VARIANTS = {
    "none_tight": """
tracking:
  thermal:
    denoise: false
    edge_pixels: 3
    frame_padding: 8
    min_dimension: 6
    max_tracks: 2
    filter_regions_pre_match: false
    min_moving_frames: 3
    max_blank_percent: 10
    max_jitter: 5
    filters:
      track_overlap_ratio: 0.9
      min_duration_secs: 1
      track_min_offset: 6.0
      track_min_mass: 3.0
      moving_vel_thresh: 4
    areas_of_interest:
      min_mass: 6.0
      pixel_variance: 3.0
      cropped_regions_strategy: "none"
    params:
      base_distance_change: 300
      min_mass_change: 10
      restrict_mass_after: 1.0
      mass_change_percent: 0.4
      max_distance: 1500
      max_blanks: 6
      velocity_multiplier: 3
      base_velocity: 1
""",
    "all_loose": """
tracking:
  thermal:
    denoise: false
    edge_pixels: 0
    frame_padding: 2
    min_dimension: 3
    max_tracks: 20
    filter_regions_pre_match: true
    max_blank_percent: 60
    max_jitter: 40
    filters:
      track_overlap_ratio: 0.1
      min_duration_secs: 0.5
      track_min_offset: 2.0
      track_min_mass: 1.0
      moving_vel_thresh: 4
    areas_of_interest:
      min_mass: 2.0
      pixel_variance: 1.0
      cropped_regions_strategy: "all"
    params:
      base_distance_change: 900
      min_mass_change: null
      restrict_mass_after: 2.5
      mass_change_percent: null
      max_distance: 4000
      max_blanks: 25
      velocity_multiplier: 1
      base_velocity: 4
""",
    "blanks_jitter": """
tracking:
  thermal:
    denoise: false
    min_moving_frames: 1
    max_blank_percent: 5
    max_jitter: 2
    filters:
      track_overlap_ratio: 0.5
      min_duration_secs: 0
      track_min_offset: 1.0
      track_min_mass: 8.0
      moving_vel_thresh: 4
    params:
      base_distance_change: 450
      min_mass_change: 20
      restrict_mass_after: 1.5
      mass_change_percent: 0.55
      max_distance: 2000
      max_blanks: 3
      velocity_multiplier: 2
      base_velocity: 2
""",
}


def busy_clip(seed, model="lepton3"):
    return synth.make_clip(np.random.default_rng(1000 + int(seed)), T, model=model, max_blobs=8)


def run_reference(cfg, path):
    cte = rh.ref("track.cliptrackextractor")
    clipmod = rh.ref("track.clip")
    ex = cte.ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False)
    rc = clipmod.Clip(cfg.tracking["thermal"], path)
    ex.parse_clip(rc)
    return rc


def main():
    rh.install()
    confmod = rh.ref("config.config")
    tmp = tempfile.mkdtemp()
    t_on, ffc = [100000 + 111 * i for i in range(T)], [40000] * T
    scenes = []
    for seed in range(0, 40):
        scenes.append(("busy%d" % seed, "lepton3", seed))
    scenes35 = [("busy35_%d" % seed, "lepton3.5", seed) for seed in range(100, 120)]
    rows, offsets, info = [], [0], []
    for vname, yaml_text in VARIANTS.items():
        ypath = os.path.join(tmp, vname + ".yaml")
        with open(ypath, "w") as fh:
            fh.write(yaml_text)
        cfg = confmod.Config.load_from_file(ypath)
        done = {"lepton3": 0, "lepton3.5": 0}
        todo = [("possum", None, None)] + scenes + scenes35
        for sname, model, seed in todo:
            if model is not None and done[model] >= (3 if model == "lepton3" else 1):
                continue
            if sname == "possum":
                path = os.path.join(HERE, "possum.cptv")
            else:
                path = os.path.join(tmp, "%s_%s.cptv" % (vname, sname))
                encode_cptv(path, busy_clip(seed, model), [16] * T, time_on=t_on, last_ffc=ffc, model=model.encode())
            rc = run_reference(cfg, path)
            tracks = sorted(list(rc.tracks) + [t for _, t in rc.filtered_tracks], key=lambda t: t.get_id())
            # (a track trimmed to nothing keeps its id and reason but has no bounds to order or store)
            births = [(t.bounds_history[0].frame_number, t.bounds_history[0].id) for t in tracks if t.bounds_history]
            if births != sorted(births):  # same-frame births in set order (F14): nothing to pin the ids on
                print(vname, sname, "skipped: same-frame births out of component order")
                continue
            if sname != "possum" and len(tracks) < 3:
                continue
            if model is not None:
                done[model] += 1
            for t in tracks:
                for r in t.bounds_history:
                    rows.append((t.get_id(), int(r.x), int(r.y), int(r.width), int(r.height), int(r.mass),
                                 int(r.frame_number), int(bool(r.blank))))
            offsets.append(len(rows))
            info.append({"variant": vname, "scene": sname, "model": model, "seed": seed,
                         "kept": [[int(t.get_id()), float(t.stats.score)] for t in rc.tracks],
                         "filtered": [[str(reason), int(t.get_id())] for reason, t in rc.filtered_tracks]})
            print(vname, sname, "tracks", len(tracks), "kept", len(rc.tracks), "rows", offsets[-1] - offsets[-2],
                  sorted(set(r for r, _ in info[-1]["filtered"])))
    np.savez_compressed(os.path.join(HERE, "config_tracks.npz"), rows=np.asarray(rows, np.int32),
                        offsets=np.asarray(offsets, np.int32), frames=np.int32(T))
    with open(os.path.join(HERE, "config_tracks_info.json"), "w") as fh:
        json.dump({"variants": VARIANTS, "cases": info}, fh, indent=1)
    print("cases", len(info), "rows", len(rows))


if __name__ == "__main__":
    main()
