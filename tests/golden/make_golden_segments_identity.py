#!/usr/bin/env python3
"""Segment-plan golden (device planner): the REFERENCE's get_segments (imported under
oracle/refharness.py) exactly as Interpreter.frames_for_prediction / ClipClassifier.classify_clip call it
(ALL_RANDOM_MASKED, segment_width 25, spacing 9, min_segments = 1, no max, dont_filter False), with every random draw
replaced by the IDENTITY draw (tests/helpers.py:IdentityDraws): shuffle leaves the order, choice without replacement
takes the first k, choice with replacement cycles through the array.  cpx_plan_segments is that member of the
reference's random family, so its frame lists, padding order and mass-drop decisions must equal these.
-> segments_identity_golden.json

    python tests/golden/make_golden_segments_identity.py      (build container only)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import refharness as rh  # noqa: E402
from helpers import IdentityDraws  # noqa: E402


def tracks(rng):
    """(start_frame, [(mass, blank, width, height)], ffc_frames): lengths on both sides of the 40-usable-frame switch
    (datasetstructures.py:1189), of the half / quarter segment cut-offs, blanks, zero masses, all-zero tracks."""
    out = []
    lengths = [1, 3, 6, 7, 12, 13, 24, 25, 26, 37, 38, 39, 40, 41, 45, 49, 50, 60, 62, 63, 65, 74, 75, 77, 88, 100, 101,
               120, 137, 138, 150, 200, 269, 270]
    for n in lengths:
        for variant in range(3):
            start = int(rng.integers(0, 50))
            regs = []
            for i in range(n):
                if variant == 0:      # every frame usable
                    regs.append((int(rng.integers(1, 400)), False, int(rng.integers(1, 40)), int(rng.integers(1, 40))))
                    continue
                blank = bool(rng.random() < (0.12 if variant == 1 else 0.3))
                mass = 0 if blank else int(rng.integers(0, 400) if rng.random() < 0.9 else 0)
                regs.append((mass, blank, int(rng.integers(0, 40)), int(rng.integers(0, 40))))
            ffc = [] if variant == 0 else sorted(set(int(start + v) for v in rng.integers(0, n, size=int(rng.integers(0, 5)))))
            out.append((start, regs, ffc))
    out.append((5, [(0, False, 9, 9)] * 30, []))    # has_no_mass: every segment is dropped by the mass test
    out.append((5, [(0, False, 9, 9)] * 70, [9, 10]))
    return out


def main():
    rh.install()
    ds = rh.ref("ml_tools.datasetstructures")
    region_mod = rh.ref("track.region")
    rng = np.random.default_rng(2026)
    cases = []
    for ti, (start, regs, ffc) in enumerate(tracks(rng)):
        regions = np.array([region_mod.Region(5, 6, w, h, centroid=[5, 6], mass=m, frame_number=start + i, blank=b)
                            for i, (m, b, w, h) in enumerate(regs)])
        with IdentityDraws():
            segs, stats = ds.get_segments(7, ti + 1, start, regions=regions, segment_width=25, segment_frame_spacing=9,
                                          ffc_frames=ffc, repeats=1, min_frames=0,
                                          segment_types=[ds.SegmentType.ALL_RANDOM_MASKED], max_segments=None,
                                          dont_filter=False, min_segments=1, seed=None)
        cases.append({"start": start, "regions": regs, "ffc": ffc,
                      "dropped_for_mass": int(stats["segment_mass"]),
                      "segments": [[int(f) for f in s.frame_indices] for s in segs]})
    with open(os.path.join(HERE, "segments_identity_golden.json"), "w") as fh:
        json.dump({"cases": cases}, fh)
    print(len(cases), "tracks,", sum(len(c["segments"]) for c in cases), "segments,",
          sum(c["dropped_for_mass"] for c in cases), "dropped for mass")


if __name__ == "__main__":
    main()
