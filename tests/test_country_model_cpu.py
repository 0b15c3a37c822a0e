"""A directory run picks the country model from the FIRST recording's location, as the reference does
(src/classify/clipclassifier.py:60-83,254-256): ADVICE r03 -- process(directory, track=True) used to load every
classifier without a location, so <models>/../NZ/<file> was never chosen on a directory run.  No GPU: the interpreter
factory is replaced by a recorder that stops the run."""
import json
import os
import shutil
import sys

import pytest

from helpers import GOLDEN

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "classifier-pipeline_amd"))


class _Stop(Exception):
    pass


def _setup(tmp_path, with_location):
    clips = tmp_path / "clips"
    clips.mkdir()
    shutil.copy(os.path.join(GOLDEN, "hedgehog.cptv"), clips / "a.cptv")
    shutil.copy(os.path.join(GOLDEN, "hedgehog.cptv"), clips / "b.cptv")
    meta = {"tracks": []}
    if with_location:
        meta["location"] = {"type": "Point", "coordinates": [172.6, -43.5]}  # Christchurch: (lng, lat)
    (clips / "a.txt").write_text(json.dumps(meta))
    models = tmp_path / "models"
    (models / "default").mkdir(parents=True)
    (models / "NZ").mkdir()
    for d in ("default", "NZ"):
        (models / d / "model.npz").write_bytes(b"")
    return clips, models


@pytest.mark.parametrize("with_location", [True, False])
def test_directory_run_loads_the_country_model(tmp_path, monkeypatch, with_location):
    from cpx.classify import clipclassifier as cc
    from cpx.config import Config
    from cpx.config.config import ModelConfig

    clips, models = _setup(tmp_path, with_location)
    loaded = []

    def fake_get_interpreter(model, run_over_network=False):
        loaded.append(model.model_file)
        raise _Stop()

    monkeypatch.setattr(cc, "get_interpreter", fake_get_interpreter)
    cfg = Config.get_defaults()
    model = ModelConfig.load({"id": 1, "name": "m", "model_file": str(models / "default" / "model.npz")})
    classifier = cc.ClipClassifier(cfg, model)
    with pytest.raises(_Stop):
        classifier.process(str(clips), track=True)
    want = models / ("NZ" if with_location else "default") / "model.npz"
    assert loaded == [str(want)]


def test_country_by_location_boxes():
    from cpx.classify.clipclassifier import country_by_location

    assert country_by_location(-43.5, 172.6) == "NZ"
    assert country_by_location(-33.9, 151.2) == "AU"
    assert country_by_location(51.5, 0.0) is None
