#!/usr/bin/env python3
"""Golden for ClipClassifier.post_process_file: the REFERENCE's own method
(imported under oracle/refharness.py) on the fixture clips, in both of its modes --
  * no <clip>.txt next to the recording: it tracks the file first and re-reads it with the background model the
    tracking left behind;
  * with a <clip>.txt (written by the reference's extract_file): tracks come from the metadata, the re-read starts
    from a fresh model;
with every random draw of the segment choice pinned to the identity (tests/helpers.py:IdentityDraws) and a stand-in
classifier (TensorFlow and the weights are absent) that records every chunk handed to predict().
-> postprocess_golden.npz (network inputs per track), postprocess_golden.json (segments, chunk sizes, metadata)

    python tests/golden/make_golden_postprocess.py      (build container only)
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import refharness as rh  # noqa: E402
from helpers import IdentityDraws  # noqa: E402
from make_golden_classify import LABELS, fake_predict  # noqa: E402


def run(name, with_metadata, out, info):
    interp_mod = rh.ref("ml_tools.interpreter")
    cc = rh.ref("classify.clipclassifier")
    te = rh.ref("track.trackextractor")
    rh.ref("config.config")
    cfgmod = rh.ref("config.classifyconfig")
    tools = rh.ref("ml_tools.tools")
    config = rh.default_config()
    config.tracking["thermal"].denoise = False
    with tempfile.TemporaryDirectory() as td:
        clip_file = os.path.join(td, name + ".cptv")
        shutil.copy(os.path.join(HERE, name + ".cptv"), clip_file)
        mfile = os.path.join(td, "model.json")
        json.dump({"labels": LABELS, "hyperparams": {"frame_size": 32}, "type": "thermal", "version": "golden"}, open(mfile, "w"))
        config.classify.models = [cfgmod.ModelConfig.load({"id": 7, "name": "wr-test", "model_file": mfile})]
        if with_metadata:
            te.extract_file(clip_file, config, False)  # writes <clip>.txt, as extract.py does
            assert os.path.exists(os.path.join(td, name + ".txt"))
        chunks = []

        class Capture(interp_mod.Interpreter):
            TYPE = "capture"

            def shape(self):
                return 1, (None, 160, 160, 2)

            def predict(self, frames):
                chunks.append(np.array(frames, dtype=np.float32, copy=True))
                return fake_predict(frames)

        interp = Capture(mfile)
        interp.id, interp.port = 7, 8123
        classifier = cc.ClipClassifier(config)
        classifier.get_classifier = lambda model, location=None: interp
        seg_log = {}
        orig_ffp = interp.frames_for_prediction

        def logged(clip, track, **args):
            segs = orig_ffp(clip, track, **args)
            seg_log[track.get_id()] = [[int(f) for f in s.frame_indices] for s in segs]
            return segs

        interp.frames_for_prediction = logged
        with IdentityDraws():
            classifier.post_process_file(clip_file, None)
        meta = tools.load_clip_metadata(os.path.join(td, name + ".txt"))
    key = "%s_%s" % (name, "meta" if with_metadata else "nometa")
    # chunks arrive track by track, <= 5 segments each
    pos = 0
    tracks = []
    for t in meta["tracks"]:
        segs = seg_log.get(t["id"], [])
        need = len(segs)
        mine = []
        while need > 0:
            c = chunks[pos]
            pos += 1
            mine.append(c)
            need -= len(c)
        assert need == 0
        if mine:
            out["%s_t%d_input" % (key, t["id"])] = np.concatenate(mine)
        preds = [p for p in t.get("predictions", [])]
        for p in preds:
            p.pop("classify_time", None)
        tracks.append({"id": t["id"], "start_s": t.get("start_s"), "end_s": t.get("end_s"), "segments": segs,
                       "chunks": [int(len(c)) for c in mine], "predictions": preds,
                       "positions": [[p["x"], p["y"], p["width"], p["height"], p["frame_number"]] for p in t["positions"]]})
    assert pos == len(chunks)
    info[key] = {"tracks": tracks, "models": [{k: v for k, v in m.items() if k != "classify_time"} for m in meta["models"]]}
    print(key, [(t["id"], len(t["segments"]), t["chunks"]) for t in tracks])


if __name__ == "__main__":
    out, info = {}, {}
    for name in ("possum", "hedgehog"):
        for with_metadata in (False, True):
            run(name, with_metadata, out, info)
    np.savez_compressed(os.path.join(HERE, "postprocess_golden.npz"), **out)
    tools = rh.ref("ml_tools.tools")
    with open(os.path.join(HERE, "postprocess_golden.json"), "w") as fh:
        json.dump({"labels": LABELS, "runs": info}, fh, indent=1, cls=tools.CustomJSONEncoder)
