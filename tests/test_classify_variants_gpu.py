"""The model variants of the classification pre-processing -- thermal_diff_norm, diff_norm = False, both, swapped
channels, single-frame models (reference src/ml_tools/preprocess.py:56-144, src/ml_tools/interpreter.py:240-474) --
against network inputs the REFERENCE built (tests/golden/classify_variants_golden.json, made by running its own
Interpreter under oracle/refharness.py with a stand-in predict): bit-exact by CRC32 per sample, and the aggregated
class scores of the same stand-in network."""
import json
import os
import shutil
import zlib

import numpy as np
import pytest

from helpers import GOLDEN, fake_predict

pytestmark = pytest.mark.gpu


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(GOLDEN, "classify_variants_golden.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def clips(tmp_path_factory):
    from cpx.config import Config
    from cpx.track.trackextractor import extract_file

    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = False
    out = {}
    d = tmp_path_factory.mktemp("clips")
    for name in ("hedgehog", "possum"):
        src = d / (name + ".cptv")
        shutil.copy(os.path.join(GOLDEN, name + ".cptv"), src)
        out[name] = extract_file(src, cfg, False, save_meta=False)[0]
    return out


VARIANTS = ["thermal_diff_norm", "no_diff_norm", "thermal_diff_norm_no_diff_norm", "channels_swapped", "single_frame",
            "single_frame_thermal_diff_norm", "inceptionv3_scaling", "inceptionv3_scaling_single_frame"]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", ["hedgehog", "possum"])
def test_variant_inputs_equal_reference(tmp_path, golden, clips, name, variant):
    from cpx.ml_tools.interpreter import WRResNetInterpreter

    g = golden["variants"][variant]
    with open(tmp_path / "m.json", "w") as fh:
        json.dump({"labels": golden["labels"], "hyperparams": g["hyperparams"], "type": "thermal", "version": "test"}, fh)
    # (the families other than wr-resnet classify through a model server: run_over_network; predict is replaced below)
    other = g["hyperparams"].get("model_name", "wr-resnet") != "wr-resnet"
    if other:
        with pytest.raises(NotImplementedError):
            WRResNetInterpreter(tmp_path / "m.npz", load_model=False)
    interp = WRResNetInterpreter(tmp_path / "m.npz", run_over_network=other, load_model=False)
    seen = {}

    def predict(x):
        seen["x"] = x.cpu().numpy()
        return fake_predict(seen["x"])

    interp.predict = predict
    clip = clips[name]
    assert len(clip.tracks) == len(g[name])
    single = g["hyperparams"].get("square_width", 5) == 1
    for track, want in zip(clip.tracks, g[name]):
        assert track.get_id() == want["track_id"]
        segs = None if single else [np.array(s) for s in want["segments"]]
        pred = interp.classify_track(clip, track, segment_frames=segs)
        x = seen["x"]
        assert list(x.shape) == want["shape"]
        got = [crc(s) for s in x]
        if got != want["crc"]:
            bad = [i for i, (a, b) in enumerate(zip(got, want["crc"])) if a != b]
            i = bad[0]
            raise AssertionError("%s %s track %s: %d of %d samples differ; sample %d min/max/mean %.6f %.6f %.6f, reference "
                                 "%.6f %.6f %.6f" % (name, variant, want["track_id"], len(bad), len(got), i, x[i].min(),
                                                     x[i].max(), x[i].astype(np.float64).mean(), want["min"][i],
                                                     want["max"][i], want["mean"][i]))
        if single:  # one Prediction record (the reference zips the predictions with its one-element mass list)
            assert [int(p.frames) for p in pred.predictions] == want["segments"][:1]
        assert np.allclose(pred.class_best_score, want["class_best_score"], rtol=0, atol=1e-6)
        m = pred.get_metadata(None)
        assert m["tag"] == want["meta"]["tag"] and m["all_class_confidences"] == want["meta"]["all_class_confidences"]
        assert len(m["predictions"]) == len(want["meta"]["predictions"])


def test_preprocess_fn_families(tmp_path):
    """interpreter.py:64-98: which model names scale their input, which cannot be prepared here, which get the warning."""
    from cpx.ml_tools.interpreter import WRResNetInterpreter, inc3_preprocess

    def make(name):
        with open(tmp_path / "m.json", "w") as fh:
            json.dump({"labels": ["a", "b"], "hyperparams": {"model_name": name}, "type": "thermal"}, fh)
        return WRResNetInterpreter(tmp_path / "m.npz", run_over_network=True, load_model=False)

    for name in ("inceptionv3", "nasnet", "resnetv2", "mobilenet", "inceptionresnetv2"):
        it = make(name)
        assert it.preprocess_fn is inc3_preprocess and it.limits_flags() & 32
    for name in ("wr-resnet", "efficientnetv2b3", "something-new"):
        it = make(name)
        assert it.preprocess_fn is None and not it.limits_flags() & 32
    for name in ("resnet", "vgg16", "densenet121"):
        with pytest.raises(NotImplementedError):
            make(name)
    x = np.array([0.0, 127.5, 255.0], np.float32)
    assert inc3_preprocess(x).tolist() == [-1.0, 0.0, 1.0]


def test_model_by_country_picks_the_country_directory(tmp_path):
    """clipclassifier.py:60-83: with a recording location inside a country's box, <models>/<country>/<file name> is
    loaded when it exists; model_by_country = False keeps the configured file."""
    from cpx.classify.clipclassifier import ClipClassifier, country_by_location
    from cpx.config import Config
    from cpx.config.config import ModelConfig
    from cpx.ml_tools import wrresnet as wr

    labels = ["a", "false-positive", "c"]
    base = tmp_path / "models"
    (base / "default").mkdir(parents=True)
    (base / "NZ").mkdir()
    wr.save_model(base / "default" / "wr", wr.random_weights(3, seed=1), labels, hyperparams={"frame_size": 32})
    wr.save_model(base / "NZ" / "wr", wr.random_weights(3, seed=2), ["nz-a", "false-positive", "nz-c"], hyperparams={"frame_size": 32})
    assert country_by_location(-36.85, 174.76) == "NZ" and country_by_location(-33.87, 151.21) == "AU"
    assert country_by_location(51.5, -0.1) is None
    auckland = {"type": "Point", "coordinates": [174.76, -36.85]}
    for by_country, want in ((True, "nz-a"), (False, "a")):
        cfg = Config.get_defaults()
        cfg.classify.models = [ModelConfig.load({"id": 3, "name": "wr", "model_file": str(base / "default" / "wr.npz")})]
        cc = ClipClassifier(cfg, model_by_country=by_country)
        assert cc.get_classifier(cfg.classify.models[0], auckland).labels[0] == want
    cfg = Config.get_defaults()
    cfg.classify.models = [ModelConfig.load({"id": 3, "name": "wr", "model_file": str(base / "default" / "wr.npz")})]
    london = {"type": "Point", "coordinates": [-0.1, 51.5]}
    assert ClipClassifier(cfg).get_classifier(cfg.classify.models[0], london).labels[0] == "a"
