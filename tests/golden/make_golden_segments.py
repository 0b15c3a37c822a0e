#!/usr/bin/env python3
"""Segment-selection golden: the REFERENCE's ml_tools.datasetstructures.get_segments (imported
under oracle/refharness.py) on seeded synthetic tracks, with both of its random sources pinned -- the per-call
generator (`seed=`) and the global NumPy generator it also shuffles with (`np.random.seed`).  -> segments_golden.json

    python tests/golden/make_golden_segments.py      (build container only)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import refharness as rh  # noqa: E402

TYPES = ["ALL_RANDOM_MASKED", "ALL_RANDOM", "ALL_RANDOM_NOMIN", "IMPORTANT_RANDOM", "TOP_RANDOM"]


def tracks(rng):
    """(start_frame, [(mass, blank, width, height)], ffc_frames) recipes: short / long tracks, blanks, zero masses."""
    out = []
    for n in (3, 8, 12, 24, 25, 26, 40, 49, 50, 77, 120, 200):
        start = int(rng.integers(0, 50))
        regs = []
        for i in range(n):
            blank = bool(rng.random() < 0.12)
            mass = 0 if blank else int(rng.integers(0, 400) if rng.random() < 0.9 else 0)
            regs.append((mass, blank, int(rng.integers(0, 40)), int(rng.integers(0, 40))))
        ffc = sorted(set(int(start + v) for v in rng.integers(0, n, size=int(rng.integers(0, 4)))))
        out.append((start, regs, ffc))
    return out


def main():
    rh.install()
    ds = rh.ref("ml_tools.datasetstructures")
    region_mod = rh.ref("track.region")
    rng = np.random.default_rng(2025)
    cases = []
    failed = {}
    for ti, (start, regs, ffc) in enumerate(tracks(rng)):
        regions = np.array([region_mod.Region(5, 6, w, h, centroid=[5, 6], mass=m, frame_number=start + i, blank=b)
                            for i, (m, b, w, h) in enumerate(regs)])
        for tname in TYPES:
            for min_segments, max_segments, dont_filter in ((1, None, False), (None, None, False), (1, 2, True)):
                seed = 100 * ti + len(cases)
                np.random.seed(seed)
                try:
                        segs, _ = ds.get_segments(7, ti + 1, start, regions=regions, segment_width=25,
                                              segment_frame_spacing=9,
                                              ffc_frames=ffc, repeats=1, min_frames=0,
                                              segment_types=[ds.SegmentType[tname]], max_segments=max_segments,
                                              dont_filter=dont_filter, min_segments=min_segments, seed=seed)
                except Exception as e:  # some segment types do not survive some inputs in the reference itself
                    failed[tname] = failed.get(tname, 0) + 1
                    continue
                cases.append({"track": ti, "start": start, "regions": regs, "ffc": ffc, "type": tname, "seed": seed,
                              "min_segments": min_segments, "max_segments": max_segments, "dont_filter": dont_filter,
                              "segments": [{"frames": [int(f) for f in s.frame_indices], "mass": int(s.mass),
                                            "weight": float(s.weight), "filtered": bool(getattr(s, "filtered", False))}
                                           for s in segs]})
    with open(os.path.join(HERE, "segments_golden.json"), "w") as fh:
        json.dump({"cases": cases}, fh)
    print("reference raised for:", failed)
    print(len(cases), "cases,", sum(len(c["segments"]) for c in cases), "segments")


if __name__ == "__main__":
    main()
