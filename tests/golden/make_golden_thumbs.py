#!/usr/bin/env python3
"""Generates tests/golden/<clip>_dn<0|1>_thumbs.json by RUNNING THE REFERENCE's thumbnail stage
(classify/thumbnail.py under oracle/refharness.py) on its own fixture clips: per kept track the list
of per-frame statistics (frame number, contour points, median difference), the chosen thumbnail and
its score; plus, for a trackless variant of the clip (first frames only), best_trackless_thumb.

Build container only:   python tests/golden/make_golden_thumbs.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import refharness as rh  # noqa: E402


def region_dict(r):
    return dict(x=int(r.x), y=int(r.y), width=int(r.width), height=int(r.height), mass=int(r.mass),
                frame_number=int(r.frame_number),
                centroid=None if r.centroid is None else [float(r.centroid[0]), float(r.centroid[1])])


def run(clip_name, denoise, max_frames=None, path=None):
    rh.install()
    cte = rh.ref("track.cliptrackextractor")
    clipmod = rh.ref("track.clip")
    thumb = rh.ref("classify.thumbnail")
    cfg = rh.default_config()
    cfg.tracking["thermal"].denoise = bool(denoise)
    ex = cte.ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False, calculate_thumbnail_info=True,
                                max_frames=None)
    clip = clipmod.Clip(cfg.tracking["thermal"], path or os.path.join(HERE, clip_name + ".cptv"))
    if max_frames is not None:
        # stop the reader after max_frames frames: a trackless clip for best_trackless_thumb
        import cptv_rs_python_bindings as rs

        orig = rs.CptvReader

        class Short(orig):
            def __init__(self, path):
                super().__init__(path)
                self._left = max_frames

            def next_frame(self):
                if self._left <= 0:
                    return None
                self._left -= 1
                return super().next_frame()

        cte.CptvReader = Short
        try:
            ex.parse_clip(clip)
        finally:
            cte.CptvReader = orig
    else:
        ex.parse_clip(clip)
    out = {"clip": clip_name, "denoise": int(denoise), "max_frames": max_frames, "tracks": []}
    for track in clip.tracks:
        stats, max_mass, max_md, min_md, max_contour = thumb.get_track_thumb_stats(clip, track)
        best, score = thumb.get_thumbnail_info(clip, track)
        out["tracks"].append({
            "id": int(track.get_id()),
            "start_frame": int(track.start_frame),
            "first": [int(v) for v in (track.bounds_history[0].x, track.bounds_history[0].y,
                                       track.bounds_history[0].width, track.bounds_history[0].height)],
            "stats": [[int(s.region.frame_number), int(s.contours), float(s.median_diff)] for s in stats],
            "max_mass": float(max_mass), "max_median_diff": float(max_md), "min_median_diff": float(min_md),
            "max_contour": int(max_contour),
            "best": None if best is None else {"region": region_dict(best.region), "contours": int(best.contours),
                                               "median_diff": float(best.median_diff), "score": float(score)},
        })
    if len(clip.tracks) == 0:
        out["n_region_history"] = int(sum(len(r) for r in clip.region_history))
        out["trackless"] = region_dict(thumb.best_trackless_thumb(clip))
    return out


def main():
    for name in ("possum", "hedgehog"):
        for dn in (0, 1):
            out = run(name, dn)
            with open(os.path.join(HERE, "%s_dn%d_thumbs.json" % (name, dn)), "w") as f:
                json.dump(out, f, indent=1)
            print(name, dn, [(t["id"], t["best"]["region"]["frame_number"], t["best"]["contours"],
                              t["best"]["median_diff"], round(t["best"]["score"])) for t in out["tracks"]])
    for name, n in (("possum", 30), ("hedgehog", 8)):
        out = run(name, 0, max_frames=n)
        with open(os.path.join(HERE, "%s_trackless_thumbs.json" % name), "w") as f:
            json.dump(out, f, indent=1)
        print(name, "trackless", len(out["tracks"]), out.get("n_region_history"), out.get("trackless"))


def busy():
    """Two seeded busy synthetic scenes (see make_golden_busy.py): ~20 tracks, many contour shapes."""
    import tempfile

    import numpy as np

    sys.path.insert(0, os.path.join(REPO, "tests"))
    sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
    from cpx import synth
    from helpers import encode_cptv

    T = 110
    clips = []
    with tempfile.TemporaryDirectory() as td:
        for seed in (15, 42):
            frames = synth.make_clip(np.random.default_rng(1000 + seed), T, max_blobs=8)
            path = os.path.join(td, "c%d.cptv" % seed)
            encode_cptv(path, frames, [16] * T, time_on=[100000 + 111 * i for i in range(T)], last_ffc=[40000] * T,
                        model=b"lepton3")
            out = run("busy%d" % seed, 0, path=path)
            out["seed"] = seed
            clips.append(out)
            print("busy", seed, [(t["id"], len(t["stats"]), t["best"]["contours"]) for t in out["tracks"]])
    with open(os.path.join(HERE, "busy_thumbs.json"), "w") as f:
        json.dump({"frames": T, "clips": clips}, f)


if __name__ == "__main__":
    if "--busy" in sys.argv:
        busy()
    else:
        main()
