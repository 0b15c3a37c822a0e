#!/usr/bin/env python3
"""Golden for the IR tracker's host-side pure functions, from the REFERENCE's own code under
oracle/refharness.py: IRTrackExtractor.inside_trap_top / inside_trap_bottom / filter_track,
Line, get_trap_lines, rect_distance (src/track/irtrackextractor.py:40-91,564-787) and Track.update_trapped_state
(src/track/track.py:951-958) on seeded boxes and box sequences, for both trap sizes.
(The reference's frame loop itself cannot be run at this snapshot: it calls FrameBuffer.get_frame_ago and
Track.get_stats, which do not exist.)  -> irtrap_golden.json

    python tests/golden/make_golden_irtrap.py      (build container only)
"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import refharness as rh  # noqa: E402


def boxes(rng, n):
    out = []
    for _ in range(n):
        w, h = int(rng.integers(20, 400)), int(rng.integers(20, 300))
        x, y = int(rng.integers(0, 640 - w)), int(rng.integers(0, 480 - h))
        out.append([x, y, w, h, int(rng.integers(1, 5000))])
    return out


def main():
    rh.install()
    cfg = rh.default_config()
    irt = rh.ref("track.irtrackextractor")
    region_mod = rh.ref("track.region")
    track_mod = rh.ref("track.track")
    rng = np.random.default_rng(77)
    out = {}
    for size in ("L", "S"):
        ex = irt.IRTrackExtractor(cfg.tracking, trap_size=size)
        # (filter_components, irtrackextractor.py:564-594, builds a Region without its required centroid: a TypeError
        # in the reference itself, so it is not part of this golden)
        seqs = []
        for _ in range(300):
            seq = boxes(rng, 3)
            if rng.random() < 0.5:   # a box sliding a little: realistic consecutive bounds
                for k in (1, 2):
                    seq[k] = [seq[0][0] + int(rng.integers(-8, 9)), seq[0][1] + int(rng.integers(-8, 9)), seq[0][2], seq[0][3], 9]
                    seq[k][0] = max(0, min(seq[k][0], 640 - seq[k][2]))
                    seq[k][1] = max(0, min(seq[k][1], 480 - seq[k][3]))
            res = {"top": [], "bottom": []}
            for which in ("top", "bottom"):
                track = track_mod.Track("c", id=1, tracking_config=cfg.tracking["IR"])
                for i, b in enumerate(seq):
                    r = region_mod.Region(b[0], b[1], b[2], b[3], centroid=[b[0] + b[2] // 2, b[1] + b[3] // 2], mass=b[4], frame_number=i)
                    track.bounds_history.append(r)
                    got = (ex.inside_trap_top if which == "top" else ex.inside_trap_bottom)(track)
                    res[which].append([bool(got), bool(track.in_trap), int(track.direction), bool(r.in_trap)])
            seqs.append({"boxes": seq, "top": res["top"], "bottom": res["bottom"]})
        ft = []
        clip = SimpleNamespace(frames_per_second=10, filtered_tracks=[])
        for _ in range(40):
            n = int(rng.integers(0, 6))
            stats = SimpleNamespace(max_offset=float(rng.uniform(0, 40)), frames_moved=int(rng.integers(0, 5)))
            track = track_mod.Track("c", id=1, tracking_config=cfg.tracking["IR"])
            track.bounds_history = [None] * n
            ft.append({"len": n, "max_offset": stats.max_offset, "frames_moved": stats.frames_moved,
                       "filtered": bool(ex.filter_track(clip, track, stats))})
        out[size] = {"lines": [[ex.left_bottom.m, ex.left_bottom.c], [ex.right_bottom.m, ex.right_bottom.c]],
                     "sequences": seqs,
                     "filter_track": ft, "reasons": [r for r, _ in clip.filtered_tracks]}
    pairs = [(boxes(rng, 1)[0], boxes(rng, 1)[0]) for _ in range(200)]
    out["rect_distance"] = [[a, b, float(irt.rect_distance(a, b))] for a, b in pairs]
    with open(os.path.join(HERE, "irtrap_golden.json"), "w") as fh:
        json.dump(out, fh)
    for size in ("L", "S"):
        s = out[size]
        print(size,
              sum(any(r[1] for r in q["top"]) for q in s["sequences"]), "sequences trapped (top),",
              sum(any(r[1] for r in q["bottom"]) for q in s["sequences"]), "(bottom)")


if __name__ == "__main__":
    main()
