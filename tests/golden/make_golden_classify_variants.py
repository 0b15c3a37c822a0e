#!/usr/bin/env python3
"""Golden vectors for the model variants of the classification pre-processing (reference
src/ml_tools/preprocess.py:56-144, src/ml_tools/interpreter.py:240-474): thermal_diff_norm, diff_norm = False, both,
swapped channel order, and single-frame models (square_width = 1: frames_for_prediction's region list, preprocess_frames,
preprocess_single_frame).  Produced by RUNNING THE REFERENCE's own Interpreter.classify_track under oracle/refharness.py
with a stand-in predict that records the network input; stored as CRC32 + min / max / mean per sample (bit-exact checks
need no more) and the prediction metadata.

    python tests/golden/make_golden_classify_variants.py      (build container only)
"""
import json
import os
import sys
import tempfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, HERE)
import refharness as rh  # noqa: E402
from make_golden_classify import LABELS, fake_predict, segment_plan  # noqa: E402

VARIANTS = {
    "thermal_diff_norm": {"thermal_diff_norm": True},
    "no_diff_norm": {"diff_norm": False},
    "thermal_diff_norm_no_diff_norm": {"thermal_diff_norm": True, "diff_norm": False},
    "channels_swapped": {"channels": ["filtered", "thermal"]},
    "single_frame": {"square_width": 1},
    "single_frame_thermal_diff_norm": {"square_width": 1, "thermal_diff_norm": True},
    # the input scaling of the other model families (interpreter.py:64-98: get_preprocess_fn; inceptionv3 and the
    # Keras 'tf'-mode families get x / 127.5 - 1 applied to the tiled sample, preprocess.py:200-201)
    "inceptionv3_scaling": {"model_name": "inceptionv3"},
    "inceptionv3_scaling_single_frame": {"model_name": "inceptionv3", "square_width": 1},
}


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def main():
    interp_mod = rh.ref("ml_tools.interpreter")
    out = {"labels": LABELS, "variants": {}}
    for name in ("hedgehog", "possum"):
        clip, ex = rh.run_tracking(os.path.join(HERE, name + ".cptv"), denoise=False)
        for vname, hp in VARIANTS.items():
            with tempfile.TemporaryDirectory() as td:
                params = dict({"frame_size": 32}, **hp)
                mfile = os.path.join(td, "model.json")
                json.dump({"labels": LABELS, "hyperparams": params, "type": "thermal", "version": "golden"}, open(mfile, "w"))
                captured = {}

                class Capture(interp_mod.Interpreter):
                    TYPE = "capture"

                    def shape(self):
                        return 1, (None, 160, 160, 2)

                    def predict(self, frames):
                        captured["x"] = np.array(frames, dtype=np.float32, copy=True)
                        return fake_predict(frames)

                interp = Capture(mfile)
                tracks = []
                for track in clip.tracks:
                    single = params.get("square_width", 5) == 1
                    segs = None if single else segment_plan(track)
                    if single:
                        # the reference's own entry, Interpreter.preprocess (interpreter.py:110-130), calls
                        # preprocess_frames(clip, track) WITHOUT the samples at this snapshot and raises TypeError;
                        # the functions behind it run: called here as that entry evidently means to
                        samples = interp.frames_for_prediction(clip, track, frames_per_classify=1)
                        frames, pre, masses = interp.preprocess_frames(clip, track, samples)
                        tp = interp.track_prediction_from_raw(track.get_id(), frames, interp.predict(pre), masses) \
                            if len(frames) > 1 else None
                    else:
                        tp = interp.classify_track(clip, track, segment_frames=segs)
                    x = captured["x"]
                    m = tp.get_metadata(None)
                    m.pop("classify_time", None)
                    for p in m["predictions"]:
                        p.pop("predicted_time", None)
                    tracks.append({"track_id": track.get_id(), "shape": list(x.shape),
                                   "segments": [int(f) for f in frames] if single else [[int(f) for f in s] for s in segs],
                                   "masses": [int(v) for v in masses] if single else None,
                                   "crc": [crc(s) for s in x],
                                   "min": [float(s.min()) for s in x], "max": [float(s.max()) for s in x],
                                   "mean": [float(s.astype(np.float64).mean()) for s in x],
                                   "class_best_score": [float(v) for v in tp.class_best_score], "meta": m})
                out["variants"].setdefault(vname, {"hyperparams": params})[name] = tracks
                print(name, vname, [t["shape"] for t in tracks])
    tools = rh.ref("ml_tools.tools")
    with open(os.path.join(HERE, "classify_variants_golden.json"), "w") as fh:
        json.dump(out, fh, cls=tools.CustomJSONEncoder)


if __name__ == "__main__":
    main()
