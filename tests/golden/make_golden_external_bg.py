#!/usr/bin/env python3
"""Golden for an externally owned background: the REFERENCE's ClipTrackExtractor
(under oracle/refharness.py) constructed with update_background=False and fed frame by frame through
start_tracking(clip, [frame], background_alg=<a WeightedBackground the CALLER owns>), the way PiClassifier drives it
(piclassifier.py:322-333,423-431, 907-975).  The owner here updates its model with the 45-frame mean only after
every third frame -- any policy will do, the point is that the extractor must read the model as the owner left it.
Also: update_background=False through parse_clip (the model stays as init_clip seeded it).
-> external_bg_golden.json

    python tests/golden/make_golden_external_bg.py      (build container only)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import refharness as rh  # noqa: E402


def regions_of(rs):
    return [[int(r.x), int(r.y), int(r.width), int(r.height), int(r.mass), bool(r.blank)] for r in rs]


def tracks_of(clip):
    return [{"id": t.get_id(), "start_frame": int(t.start_frame), "bounds": regions_of(t.bounds_history)}
            for t in sorted(clip.tracks, key=lambda t: t.get_id())]


def run_external(name):
    rh.install()
    cfg = rh.default_config()
    cfg.tracking["thermal"].denoise = False
    cte, clipmod = rh.ref("track.cliptrackextractor"), rh.ref("track.clip")
    md = rh.ref("piclassifier.motiondetector")
    from cptv_rs_python_bindings import CptvReader

    path = os.path.join(HERE, name + ".cptv")
    ex = cte.ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False, update_background=False)
    clip = clipmod.Clip(cfg.tracking["thermal"], path)
    ex.init_clip(clip)
    reader = CptvReader(path)
    first = reader.next_frame()
    owner = md.WeightedBackground(clip.crop_rectangle.x, clip.crop_rectangle, clip.res_x, clip.res_y,
                                  ex.background_alg.weight_add)
    owner.process_frame(first.pix)
    frames = [first]
    while True:
        f = reader.next_frame()
        if f is None:
            break
        frames.append(f)
    per_frame = []
    seen = []
    t = 0
    for f in frames:
        if f.background_frame:
            continue
        ex.start_tracking(clip, [f], background_alg=owner)
        per_frame.append({"regions": regions_of(clip.region_history[-1]), "average": float(owner.average),
                          "filtered_sum": float(np.abs(clip.frame_buffer.get_last_x(1)[0].filtered).sum())})
        seen.append(f.pix)
        if t % 3 == 2:   # the owner's own policy
            owner.process_frame(np.mean(seen[-45:], axis=0))
        t += 1
    ex.apply_track_filtering(clip)
    return {"frames": per_frame, "tracks": tracks_of(clip)}


def run_frozen(name):
    cfg = rh.default_config()
    cfg.tracking["thermal"].denoise = False
    cte, clipmod = rh.ref("track.cliptrackextractor"), rh.ref("track.clip")
    path = os.path.join(HERE, name + ".cptv")
    ex = cte.ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False, update_background=False)
    clip = clipmod.Clip(cfg.tracking["thermal"], path)
    ex.parse_clip(clip)
    return {"tracks": tracks_of(clip), "regions": [regions_of(r) for r in clip.region_history],
            "filtered_sum": float(clip.stats.filtered_sum)}


if __name__ == "__main__":
    out = {}
    for name in ("possum", "hedgehog"):
        out[name] = {"external": run_external(name), "frozen": run_frozen(name)}
        print(name, len(out[name]["external"]["frames"]), [len(t["bounds"]) for t in out[name]["external"]["tracks"]],
              [len(t["bounds"]) for t in out[name]["frozen"]["tracks"]])
    with open(os.path.join(HERE, "external_bg_golden.json"), "w") as fh:
        json.dump(out, fh)
