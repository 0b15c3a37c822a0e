#!/usr/bin/env python3
"""Generates tests/golden/busy_tracks.npz by RUNNING THE REFERENCE (under oracle/refharness.py) on seeded synthetic
busy scenes (cpx.synth.make_clip with up to 8 objects, written as CPTV files): every track the reference created --
kept or filtered -- with all its bounds (x, y, width, height, mass, frame, blank) and whether width / height are
Python ints in the reference (the dtype switch of Kalman blank regions).  The reference creates same-frame tracks in
set-iteration order (varies from process to process); seeds where that order is not component order are
skipped, the rest pin the association / Kalman / blank-region logic beyond the two fixture clips.

Build container only:   python tests/golden/make_golden_busy.py
"""
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

T, PLAIN, LAST_SEED = 110, 5, 59  # 5 ordinary clips + every clip of seeds 0..59 with Python-int sized regions


def clip_for(seed):
    return synth.make_clip(np.random.default_rng(1000 + seed), T, max_blobs=8)


def times():
    return [100000 + 111 * i for i in range(T)], [40000] * T


def main():
    rh.install()
    cte = rh.ref("track.cliptrackextractor")
    clipmod = rh.ref("track.clip")
    tmp = tempfile.mkdtemp()
    seeds, rows, offsets, info = [], [], [0], []
    seed = 0
    n_plain = 0
    while seed <= LAST_SEED:
        clip = clip_for(seed)
        path = os.path.join(tmp, "c%d.cptv" % seed)
        t_on, ffc = times()
        encode_cptv(path, clip, [16] * T, time_on=t_on, last_ffc=ffc, model=b"lepton3")
        cfg = rh.default_config()
        cfg.tracking["thermal"].denoise = False
        ex = cte.ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False)
        rc = clipmod.Clip(cfg.tracking["thermal"], path)
        ex.parse_clip(rc)
        tracks = sorted(list(rc.tracks) + [t for _, t in rc.filtered_tracks], key=lambda t: t.get_id())
        # canonical birth order: tracks born in the same frame in component (region id) order
        births = [(t.bounds_history[0].frame_number, t.bounds_history[0].id) for t in tracks]
        seed += 1
        if births != sorted(births) or len(tracks) < 3:
            continue
        has_py = any((type(r.width) is int) or (type(r.height) is int) for t in tracks for r in t.bounds_history)
        if not has_py:
            if n_plain >= PLAIN:
                continue
            n_plain += 1
        seeds.append(seed - 1)
        for t in tracks:
            for r in t.bounds_history:
                rows.append((t.get_id(), int(r.x), int(r.y), int(r.width), int(r.height), int(r.mass),
                             int(r.frame_number), int(bool(r.blank)), int(type(r.width) is int),
                             int(type(r.height) is int)))
        info.append({"seed": seed - 1,
                     "kept": [[int(t.get_id()), float(t.stats.score), int(t.stats.frames_moved),
                               float(t.stats.max_offset), float(t.stats.average_mass), float(t.stats.delta_std)]
                              for t in rc.tracks],
                     "filtered": [[str(reason), int(t.get_id())] for reason, t in rc.filtered_tracks]})
        offsets.append(len(rows))
        print("seed", seed - 1, "tracks", len(tracks), "regions", offsets[-1] - offsets[-2])
    rows = np.asarray(rows, np.int32)
    print("python-int sized regions:", int((rows[:, 8] | rows[:, 9]).sum()), "blank:", int(rows[:, 7].sum()))
    np.savez_compressed(os.path.join(HERE, "busy_tracks.npz"), seeds=np.asarray(seeds, np.int32), rows=rows,
                        offsets=np.asarray(offsets, np.int32), frames=np.int32(T))
    import json

    with open(os.path.join(HERE, "busy_tracks_info.json"), "w") as fh:
        json.dump(info, fh)
    print("filter reasons:", sorted(set(r for c in info for r, _ in c["filtered"])))


if __name__ == "__main__":
    main()
