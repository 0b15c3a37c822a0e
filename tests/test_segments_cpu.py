"""Segment selection (host side): cpx.ml_tools.datasetstructures.get_segments against what the
REFERENCE's get_segments returned for the same seeded tracks with both random sources pinned
(tests/golden/segments_golden.json, make_golden_segments.py)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN


class _R:
    def __init__(self, mass, blank, width, height, frame_number):
        self.mass, self.blank, self.width, self.height, self.frame_number = mass, blank, width, height, frame_number


def test_get_segments_equals_reference():
    from cpx.ml_tools import datasetstructures as ds

    with open(os.path.join(GOLDEN, "segments_golden.json")) as fh:
        cases = json.load(fh)["cases"]
    n_segments = 0
    seen_types = set()
    for c in cases:
        regions = np.array([_R(m, b, w, h, c["start"] + i) for i, (m, b, w, h) in enumerate(c["regions"])],
                           dtype=object)
        np.random.seed(c["seed"])
        segs, _ = ds.get_segments(7, c["track"] + 1, c["start"], regions=regions, segment_width=25,
                                  segment_frame_spacing=9, ffc_frames=c["ffc"], repeats=1, min_frames=0,
                                  segment_types=[ds.SegmentType[c["type"]]], max_segments=c["max_segments"],
                                  dont_filter=c["dont_filter"], min_segments=c["min_segments"], seed=c["seed"])
        got = [{"frames": [int(f) for f in s.frame_indices], "mass": int(s.mass), "weight": float(s.weight),
                "filtered": bool(s.filtered)} for s in segs]
        assert got == c["segments"], (c["type"], c["track"], c["seed"])
        n_segments += len(got)
        seen_types.add(c["type"])
    assert n_segments > 200 and {"ALL_RANDOM_MASKED", "ALL_RANDOM"} <= seen_types


def test_get_segments_with_identity_draws_equals_reference():
    """The member of the random family cpx_plan_segments implements: every draw the identity (helpers.IdentityDraws).
    The host port under those draws must return what the REFERENCE returned under the same draws
    (tests/golden/segments_identity_golden.json, make_golden_segments_identity.py) -- frame lists incl. padding order,
    and the segments dropped by the mass test."""
    from cpx.ml_tools import datasetstructures as ds
    from helpers import IdentityDraws

    with open(os.path.join(GOLDEN, "segments_identity_golden.json")) as fh:
        cases = json.load(fh)["cases"]
    n_segments = dropped = 0
    for ci, c in enumerate(cases):
        regions = np.array([_R(m, b, w, h, c["start"] + i) for i, (m, b, w, h) in enumerate(c["regions"])],
                           dtype=object)
        with IdentityDraws():
            segs, stats = ds.get_segments(7, ci + 1, c["start"], regions=regions, segment_width=25,
                                          segment_frame_spacing=9, ffc_frames=c["ffc"], repeats=1, min_frames=0,
                                          segment_types=[ds.SegmentType.ALL_RANDOM_MASKED], max_segments=None,
                                          dont_filter=False, min_segments=1, seed=None)
        assert [[int(f) for f in s.frame_indices] for s in segs] == c["segments"], ci
        assert int(stats["segment_mass"]) == c["dropped_for_mass"], ci
        n_segments += len(segs)
        dropped += c["dropped_for_mass"]
    assert n_segments >= 200 and dropped >= 2


def test_hyperparams_defaults_equal_reference():
    """cpx.ml_tools.hyperparams.HyperParams against the reference's defaults / derived values
    (tests/golden/defaults_golden.json, make_golden_defaults.py)."""
    from enum import Enum

    from cpx.ml_tools.hyperparams import HyperParams

    def plain(v):
        if isinstance(v, Enum):
            return v.name
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        return v

    with open(os.path.join(GOLDEN, "defaults_golden.json")) as fh:
        cases = json.load(fh)
    for c in cases:
        h = HyperParams(dict(c["input"]))
        got = {k: plain(v) for k, v in h.items()}
        assert got == c["params"], c["input"]
        if c["output_dim"] is not None:
            assert list(h.output_dim) == c["output_dim"], c["input"]


def _plain(o):
    from enum import Enum
    from pathlib import Path

    if isinstance(o, dict):
        return {str(k): _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    if isinstance(o, Enum):
        return o.name
    if isinstance(o, Path):
        return str(o)
    if hasattr(o, "as_dict") and not isinstance(o, (int, float, str, bool, type(None))):
        return _plain(o.as_dict())
    if hasattr(o, "__dict__") and not isinstance(o, (int, float, str, bool, type(None))):
        return {k: _plain(v) for k, v in vars(o).items() if not k.startswith("_")}
    return o


@pytest.mark.parametrize("which", ["defaults", "overrides"])
def test_config_equals_reference(which):
    """cpx.config.Config (defaults, and a YAML with overrides in every section) against the reference's Config for the
    same input (tests/golden/config_golden.json, make_golden_config.py): tracking (thermal + IR) and classify."""
    import io

    from cpx.config import Config

    with open(os.path.join(GOLDEN, "config_golden.json")) as fh:
        gold = json.load(fh)
    cfg = Config.get_defaults() if which == "defaults" else Config.load_from_stream(io.StringIO(gold["overrides_yaml"]))
    want = gold[which]

    def check(got, exp, path):
        if isinstance(exp, dict):
            assert isinstance(got, dict), path
            for k, v in exp.items():
                assert k in got, path + "/" + k
                check(got[k], v, path + "/" + k)
        elif isinstance(exp, list):
            assert len(got) == len(exp), path
            for i, (g, e) in enumerate(zip(got, exp)):
                check(g, e, "%s[%d]" % (path, i))
        else:
            assert got == exp or str(got) == str(exp), (path, got, exp)

    for t in ("thermal", "IR"):
        check(_plain(cfg.tracking[t]), want["tracking"][t], "tracking/" + t)
    check(_plain(cfg.classify), want["classify"], "classify")
    assert list(cfg.labels) == want["labels"]
    assert (cfg.use_opt_flow, cfg.verbose, cfg.worker_threads, cfg.reprocess) == (
        want["use_opt_flow"], want["verbose"], want["worker_threads"], want["reprocess"])


@pytest.mark.parametrize("name,fs", [("possum", 32), ("hedgehog", 32), ("hedgehog", 64)])
def test_track_prediction_metadata_equals_reference(name, fs):
    """cpx.classify.trackprediction.TrackPrediction (via Interpreter.track_prediction_from_raw's low-evidence cap) fed
    with the per-segment predictions of the classify goldens -> the metadata dictionary the reference's
    TrackPrediction.get_metadata produced (tests/golden/*_classify_fs*.json)."""
    from cpx.ml_tools.hyperparams import HyperParams
    from cpx.ml_tools.interpreter import Interpreter
    from cpx.ml_tools.tools import CustomJSONEncoder

    z = np.load(os.path.join(GOLDEN, "%s_classify_fs%d.npz" % (name, fs)))
    with open(os.path.join(GOLDEN, "%s_classify_fs%d.json" % (name, fs))) as fh:
        gold = json.load(fh)
    for ti, t in enumerate(gold["tracks"]):
        preds = z["t%d_pred" % ti]
        segs = z["t%d_segments" % ti]
        masses = [p["mass"] for p in t["meta"]["predictions"]]
        # through Interpreter.track_prediction_from_raw (interpreter.py:151-168): aggregation + low-evidence cap
        class _I:
            labels = gold["labels"]
            params = HyperParams({"frame_size": fs})

        tp = Interpreter.track_prediction_from_raw(_I(), t["track_id"], [np.array(s) for s in segs], preds, masses)
        got = json.loads(json.dumps(tp.get_metadata(None), cls=CustomJSONEncoder))
        got.pop("classify_time", None)
        for p in got["predictions"]:
            p.pop("predicted_time", None)
        want = t["meta"]
        assert [p["frames"] for p in got["predictions"]] == [p["frames"] for p in want["predictions"]]
        for k in ("tag", "threshold_used", "confident", "confidence", "clarity"):
            assert got[k] == pytest.approx(want[k], abs=1e-9) if isinstance(want[k], float) else got[k] == want[k], (ti, k)
        for a, b in zip(got["predictions"], want["predictions"]):
            assert a == b, ti
        assert got["all_class_confidences"] == pytest.approx(want["all_class_confidences"], abs=1e-9), ti
        assert np.allclose(tp.class_best_score, np.array(t["class_best_score"]), rtol=0, atol=1e-12)
