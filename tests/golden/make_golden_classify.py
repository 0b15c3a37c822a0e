#!/usr/bin/env python3
"""Golden vectors for the classification pre-processing,
produced by RUNNING THE REFERENCE's own Interpreter.classify_track (imported from
/root/reference/src under oracle/refharness.py) on the tracks of the fixture
clips, with explicit ``segment_frames`` (the reference's segment choice is random,
SURVEY F13) and a stand-in ``predict`` (TensorFlow and the model weights are not
available) that records the network input.

    python tests/golden/make_golden_classify.py      (build container only)
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import refharness as rh  # noqa: E402

LABELS = ["bird", "cat", "deer", "dog", "false-positive", "hedgehog", "human", "kiwi", "leporidae", "mustelid",
          "penguin", "possum", "rodent", "sheep", "vehicle", "wallaby", "land-bird"]


def fake_predict(x):
    """Deterministic stand-in for the CNN: depends on the input, sums to ~1 per row."""
    x = np.asarray(x, dtype=np.float64)
    feats = np.stack([x[:, i::17, :, :].mean(axis=(1, 2, 3)) for i in range(17)], axis=1)
    e = np.exp((feats - feats.max(axis=1, keepdims=True)) / 8.0)
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def segment_plan(track):
    """Deterministic 25-frame segments over the valid frames of a track (repeats pad a short tail)."""
    valid = [r.frame_number for r in track.bounds_history if not r.blank and r.mass > 0 and r.width > 0 and r.height > 0]
    segs = []
    for i in range(0, len(valid), 25):
        chunk = valid[i : i + 25]
        if len(chunk) < 25:
            chunk = sorted(chunk + [chunk[j % len(chunk)] for j in range(25 - len(chunk))])
        segs.append(np.array(chunk, dtype=np.int64))
    return segs


def run(name, frame_size):
    interp_mod = rh.ref("ml_tools.interpreter")
    clip, ex = rh.run_tracking(os.path.join(HERE, name + ".cptv"), denoise=False)
    with tempfile.TemporaryDirectory() as td:
        meta = {"labels": LABELS, "hyperparams": {"frame_size": frame_size}, "type": "thermal", "version": "golden"}
        mfile = os.path.join(td, "model.json")
        json.dump(meta, open(mfile, "w"))
        captured = {}

        class Capture(interp_mod.Interpreter):
            TYPE = "capture"

            def shape(self):
                return 1, (None, frame_size * 5, frame_size * 5, 2)

            def predict(self, frames):
                captured["x"] = np.array(frames, dtype=np.float32, copy=True)
                return fake_predict(frames)

        interp = Capture(mfile)
        out = {}
        tracks_meta = []
        for ti, track in enumerate(clip.tracks):
            segs = segment_plan(track)
            tp = interp.classify_track(clip, track, segment_frames=segs)
            key = "t%d" % ti
            out[key + "_segments"] = np.stack(segs).astype(np.int32)
            out[key + "_input"] = captured["x"]
            out[key + "_pred"] = fake_predict(captured["x"])
            m = tp.get_metadata(None)
            m.pop("classify_time", None)
            for p in m["predictions"]:
                p.pop("predicted_time", None)
            tracks_meta.append({"track_id": track.get_id(), "start_frame": int(track.start_frame),
                                "n_regions": len(track.bounds_history), "meta": m,
                                "class_best_score": [float(v) for v in tp.class_best_score]})
        np.savez_compressed(os.path.join(HERE, "%s_classify_fs%d.npz" % (name, frame_size)), **out)
        tools = rh.ref("ml_tools.tools")
        with open(os.path.join(HERE, "%s_classify_fs%d.json" % (name, frame_size)), "w") as fh:
            json.dump({"labels": LABELS, "tracks": tracks_meta}, fh, indent=1, cls=tools.CustomJSONEncoder)
        print(name, frame_size, [(t["track_id"], out["t%d_input" % i].shape) for i, t in enumerate(tracks_meta)])


if __name__ == "__main__":
    for name in ("possum", "hedgehog"):
        run(name, 32)
    run("hedgehog", 64)
