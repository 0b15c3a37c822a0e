#!/usr/bin/env python3
"""Runs THE REFERENCE (oracle/refharness.py) on the seeded synthetic recordings of tests/helpers.py:SYNTH_CLIPS and
stores the same per-stage vectors as make_golden.py (`<name>_dn0.npz`, `<name>_dn0_tracks.json`).  The recordings are
not committed: the tests rebuild them from the seed.  Covers what the fixture clips do not: lepton3.5 thresholds,
FFC-affected frames inside a clip.

Build container only:   python tests/golden/make_golden_synth.py
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
for p in ("oracle", "tests", "classifier-pipeline_amd"):
    sys.path.insert(0, os.path.join(REPO, p))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402
from helpers import SYNTH_CLIPS, encode_cptv, synth_clip  # noqa: E402

if __name__ == "__main__":
    tmp = tempfile.mkdtemp()
    for name, spec in SYNTH_CLIPS.items():
        frames, t_on, ffc, bgf, hdr = synth_clip(name)
        path = os.path.join(tmp, name + ".cptv")
        encode_cptv(path, frames, [16] * len(frames), time_on=t_on, last_ffc=ffc, model=hdr.model.encode(),
                    background_first=bgf[0])
        make_golden.run(name, 0, [0, 29, 34, 60], path=path)
