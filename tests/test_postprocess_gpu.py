"""ClipClassifier.post_process_file (what classify.py runs today) on the GPU against the
REFERENCE's own method run under the harness (tests/golden/make_golden_postprocess.py): both modes (tracks from
tracking, background model continued; tracks from <clip>.txt, fresh model), segment choice pinned to identity draws,
a recording stand-in classifier.  Required: the same segments, the network inputs BIT FOR BIT (the second walk's
background, limits over the sampled crops, median subtracted before the resize, thermals clipped at zero), the same
chunking (<= 5 segments per predict) and the same prediction metadata."""
import json
import os
import shutil

import numpy as np
import pytest

from helpers import GOLDEN, IdentityDraws, fake_predict

pytestmark = pytest.mark.gpu

LABELS = ["bird", "cat", "deer", "dog", "false-positive", "hedgehog", "human", "kiwi", "leporidae", "mustelid",
          "penguin", "possum", "rodent", "sheep", "vehicle", "wallaby", "land-bird"]


@pytest.mark.parametrize("name,with_metadata", [("possum", False), ("possum", True), ("hedgehog", False),
                                                ("hedgehog", True)])
def test_post_process_file_equals_reference(tmp_path, name, with_metadata):
    import torch

    from cpx.classify.clipclassifier import ClipClassifier
    from cpx.config.config import Config, ModelConfig
    from cpx.ml_tools.interpreter import Interpreter
    from cpx.ml_tools.tools import load_clip_metadata
    from cpx.track.trackextractor import extract_file

    with open(os.path.join(GOLDEN, "postprocess_golden.json")) as fh:
        gold = json.load(fh)["runs"]["%s_%s" % (name, "meta" if with_metadata else "nometa")]
    z = np.load(os.path.join(GOLDEN, "postprocess_golden.npz"))
    clip_file = tmp_path / (name + ".cptv")
    shutil.copy(os.path.join(GOLDEN, name + ".cptv"), clip_file)
    mfile = tmp_path / "model.json"
    mfile.write_text(json.dumps({"labels": LABELS, "hyperparams": {"frame_size": 32}, "type": "thermal",
                                 "version": "golden"}))
    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = False
    cfg.classify.models = [ModelConfig.load({"id": 7, "name": "wr-test", "model_file": str(mfile)})]
    if with_metadata:
        extract_file(clip_file, cfg, False)
        assert clip_file.with_suffix(".txt").exists()
    chunks = []

    class Capture(Interpreter):
        TYPE = "capture"

        def shape(self):
            return 1, (None, 160, 160, 2)

        def predict(self, frames):
            x = frames.cpu().numpy() if isinstance(frames, torch.Tensor) else np.asarray(frames, dtype=np.float32)
            chunks.append(np.array(x, dtype=np.float32, copy=True))
            return fake_predict(x)

    interp = Capture(mfile)
    interp.id, interp.port = 7, 8123
    seg_log = {}
    orig = interp.frames_for_prediction

    def logged(clip, track, **args):
        segs = orig(clip, track, **args)
        seg_log[track.get_id()] = [[int(f) for f in s.frame_indices] for s in segs]
        return segs

    interp.frames_for_prediction = logged
    classifier = ClipClassifier(cfg)
    classifier.get_classifier = lambda model, location=None: interp
    with IdentityDraws():
        meta = classifier.post_process_file(clip_file, None)
    assert meta is not False
    meta = load_clip_metadata(clip_file.with_suffix(".txt"))
    assert [t["id"] for t in meta["tracks"]] == [t["id"] for t in gold["tracks"]]
    pos = 0
    for t, g in zip(meta["tracks"], gold["tracks"]):
        assert [[p["x"], p["y"], p["width"], p["height"], p["frame_number"]] for p in t["positions"]] == g["positions"]
        assert seg_log.get(t["id"], []) == g["segments"], t["id"]
        mine = []
        for n in g["chunks"]:          # the same chunking: at most 5 segments per predict() call
            assert len(chunks[pos]) == n
            mine.append(chunks[pos])
            pos += 1
        if mine:
            want = z["%s_%s_t%d_input" % (name, "meta" if with_metadata else "nometa", t["id"])]
            got = np.concatenate(mine)
            assert got.shape == want.shape
            assert np.array_equal(got, want), (t["id"], float(np.abs(got - want).max()))
        preds = t.get("predictions", [])
        assert len(preds) == len(g["predictions"])
        for p, q in zip(preds, g["predictions"]):
            assert p["model_id"] == q["model_id"] and p.get("tag") == q.get("tag")
            assert p.get("confident") == q.get("confident")
            assert list(p["all_class_confidences"]) == list(q["all_class_confidences"])
            assert np.allclose(list(p["all_class_confidences"].values()), list(q["all_class_confidences"].values()),
                               atol=1e-6)
            assert abs(p["confidence"] - q["confidence"]) <= 1e-6 and abs(p["clarity"] - q["clarity"]) <= 1e-6
            for sp, sq in zip(p["predictions"], q["predictions"]):
                assert sp["frames"] == sq["frames"] and sp["prediction"] == sq["prediction"] and sp["mass"] == sq["mass"]
    assert pos == len(chunks)
    assert [m["id"] for m in meta["models"]] == [m["id"] for m in gold["models"]]
