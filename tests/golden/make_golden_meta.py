#!/usr/bin/env python3
"""Whole-file metadata goldens: the REFERENCE's own extract_file (track/trackextractor.py, under oracle/refharness.py)
on seeded synthetic recordings -> the JSON it writes (tracks, positions, scores, thumbnails).  -> synth_meta.json

    python tests/golden/make_golden_meta.py      (build container only)
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
from helpers import SYNTH_CLIPS, encode_cptv, synth_clip  # noqa: E402

T = 110
BUSY_SEEDS = (0, 3, 6)   # busy scenes whose same-frame births came out in component order in this run (F14)


def main():
    rh.install()
    te = rh.ref("track.trackextractor")
    out = {"busy_frames": T, "clips": {}}
    with tempfile.TemporaryDirectory() as td:
        jobs = []
        for name in SYNTH_CLIPS:
            frames, t_on, ffc, bgf, hdr = synth_clip(name)
            path = os.path.join(td, name + ".cptv")
            encode_cptv(path, frames, [16] * len(frames), time_on=t_on, last_ffc=ffc, model=hdr.model.encode(),
                        background_first=bgf[0])
            jobs.append((name, path))
        for seed in BUSY_SEEDS:
            frames = synth.make_clip(np.random.default_rng(1000 + seed), T, max_blobs=8)
            path = os.path.join(td, "busy%d.cptv" % seed)
            encode_cptv(path, frames, [16] * T, time_on=[100000 + 111 * i for i in range(T)], last_ffc=[40000] * T,
                        model=b"lepton3")
            jobs.append(("busy%d" % seed, path))
        for dn in (False, True):
            for name, path in jobs:
                if dn and not name.startswith("busy0"):
                    continue  # one default-config (NLM) run is enough: the NumPy NLM stand-in is slow
                cfg = rh.default_config()
                cfg.tracking["thermal"].denoise = dn
                te.extract_file(path, cfg, False)
                with open(os.path.splitext(path)[0] + ".txt") as fh:
                    meta = json.load(fh)
                births = [(t["positions"][0]["frame_number"], t["positions"][0]["x"]) for t in meta["tracks"]]
                out["clips"]["%s_dn%d" % (name, int(dn))] = meta
                print(name, dn, "tracks", [(t["id"], t["frame_start"], t["frame_end"]) for t in meta["tracks"]],
                      "thumbnail_region" in meta)
    with open(os.path.join(HERE, "synth_meta.json"), "w") as fh:
        json.dump(out, fh)


if __name__ == "__main__":
    main()
