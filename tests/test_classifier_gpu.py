"""Drop-in classify API on the GPU: ClipClassifier.process_file(track=True) / Interpreter.classify_track
with a seeded WR-ResNet, against (a) the reference's own network inputs for explicit segments and
(b) the oracle chain (NumPy pre-processing + PyTorch CPU forward + aggregation) for the segments chosen."""
import json
import os
import shutil

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu

LABELS = ["bird", "cat", "deer", "dog", "false-positive", "hedgehog", "human", "kiwi", "leporidae", "mustelid",
          "penguin", "possum", "rodent", "sheep", "vehicle", "wallaby", "land-bird"]


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    import cnn_oracle as co
    from cpx.ml_tools import wrresnet as wr

    d = tmp_path_factory.mktemp("model")
    w = wr.random_weights(len(LABELS), seed=11)
    z = np.load(os.path.join(GOLDEN, "hedgehog_classify_fs32.npz"))
    w = co.calibrate_bn(w, z["t0_input"])
    wr.save_model(d / "wr", w, LABELS, hyperparams={"frame_size": 32})
    return d, w


def _config(model_dir):
    from cpx.config import Config
    from cpx.config.config import ModelConfig

    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = False
    cfg.classify.models = [ModelConfig.load({"id": 7, "name": "wr-test", "model_file": str(model_dir / "wr.npz")})]
    return cfg


@pytest.mark.parametrize("name", ["possum", "hedgehog"])
def test_process_file_track_and_classify(tmp_path, model_dir, name):
    import classify_oracle as co
    import cnn_oracle as cnn
    from cpx.classify.clipclassifier import ClipClassifier
    from cpx.track.trackextractor import extract_file

    mdir, w = model_dir
    src = tmp_path / (name + ".cptv")
    shutil.copy(os.path.join(GOLDEN, name + ".cptv"), src)
    cfg = _config(mdir)
    meta = ClipClassifier(cfg).process_file(str(src), track=True)
    assert meta and os.path.exists(src.with_suffix(".txt"))
    assert meta["models"][0]["id"] == 7 and "classify_time" in meta["models"][0]
    # independent re-computation with the oracle chain on the same segments
    clip, _, _ = extract_file(src, cfg, False, save_meta=False)
    H, W = clip.res_y, clip.res_x
    assert len(meta["tracks"]) == len(clip.tracks) > 0
    for tm, track in zip(meta["tracks"], clip.tracks):
        assert tm["id"] == track.get_id()
        (pm,) = tm["predictions"]
        assert pm["model_id"] == 7 and set(pm["all_class_confidences"]) == set(LABELS)
        assert {"tag", "threshold_used", "confident", "confidence", "clarity", "all_class_confidences",
                "predictions", "classify_time"} <= set(pm)
        segs = [np.array(p["frames"]) for p in pm["predictions"]]
        by_frame = {r.frame_number: r for r in track.bounds_history}
        x, _ = co.preprocess_segments(lambda q: clip.frame_buffer.get_frame(q).thermal,
                                      lambda q: clip.frame_buffer.get_frame(q).filtered.astype(np.float64),
                                      by_frame, track.bounds_history, segs, 32, (1, 1, W - 2, H - 2))
        _, probs = cnn.forward(w, x)
        score = co.classified_track(probs, prediction_frames=segs, labels=LABELS)
        got = np.array([pm["all_class_confidences"][l] for l in LABELS])
        assert np.abs(got - np.round(score, 3)).max() <= 1.5e-3
        assert pm["tag"] == LABELS[int(np.argmax(score))]


def test_classify_track_same_scores_in_the_two_plane_math_mode(tmp_path, model_dir):
    """CPX_CNN_MATH=bf16x2 (opt-in, include/cpx.h: the stride-1 3x3 layers of stages 2-4 and the stride-2 one of stage 3
    on two rounded bf16 planes and three products) is read when an engine is created; here the cached engines are
    switched directly.  Same explicit segments (process_file draws them at random, as the reference does): the same
    per-segment predictions and track scores to 1e-5."""
    from cpx.ml_tools.interpreter import WRResNetInterpreter
    from cpx.track import cliptrackextractor as cte
    from cpx.track.trackextractor import extract_file

    mdir, _ = model_dir
    src = tmp_path / "hedgehog.cptv"
    shutil.copy(os.path.join(GOLDEN, "hedgehog.cptv"), src)
    clip, _, _ = extract_file(src, _config(mdir), False, save_meta=False)
    z = np.load(os.path.join(GOLDEN, "hedgehog_classify_fs32.npz"))
    interp = WRResNetInterpreter(mdir / "wr.npz")
    interp.classify_track(clip, clip.tracks[0], segment_frames=z["t0_segments"])   # (creates / finds the engine)
    got = {}
    try:
        for mode in ("bf16x3", "bf16x2", "fp16x2"):
            assert cte._ENGINES
            for eng in cte._ENGINES.values():
                eng.set_cnn_math(mode)
            pred = interp.classify_track(clip, clip.tracks[0], segment_frames=z["t0_segments"])
            got[mode] = (np.array(pred.class_best_score, dtype=np.float64),
                         np.array([p.prediction for p in pred.predictions], dtype=np.float64))
    finally:
        for eng in cte._ENGINES.values():
            eng.set_cnn_math(eng.DEFAULT_CNN_MATH)
    for other in ("bf16x2", "fp16x2"):
        (sa, pa), (sb, pb) = got["bf16x3"], got[other]
        assert pa.shape == pb.shape and pa.size > 0
        assert float(np.abs(pa - pb).max()) <= 1e-5, (other, float(np.abs(pa - pb).max()))
        assert float(np.abs(sa - sb).max()) <= 1e-5
        assert float(np.abs(pa - pb).max()) > 0.0  # (it IS the other arithmetic)


def test_classify_track_inputs_equal_reference(tmp_path, model_dir):
    """Interpreter.classify_track with the reference's segment frames: the tensor handed to predict()
    is bit-identical to what the reference's own Interpreter built."""
    from cpx.ml_tools.interpreter import WRResNetInterpreter
    from cpx.track.trackextractor import extract_file

    mdir, w = model_dir
    src = tmp_path / "hedgehog.cptv"
    shutil.copy(os.path.join(GOLDEN, "hedgehog.cptv"), src)
    clip, _, _ = extract_file(src, _config(mdir), False, save_meta=False)
    z = np.load(os.path.join(GOLDEN, "hedgehog_classify_fs32.npz"))
    interp = WRResNetInterpreter(mdir / "wr.npz")
    seen = {}
    orig = interp.predict
    interp.predict = lambda x: seen.setdefault("x", x.cpu().numpy()) is None or orig(x)
    pred = interp.classify_track(clip, clip.tracks[0], segment_frames=z["t0_segments"])
    assert np.array_equal(seen["x"], z["t0_input"])
    assert pred is not None and abs(float(np.sum(pred.class_best_score)) - 1.0) < 1e-5
    assert [list(p.frames) for p in pred.predictions] == [list(s) for s in z["t0_segments"]]


def test_process_files_equals_process_file(tmp_path, model_dir):
    """ClipClassifier.process(directory, track=True): the recordings of a device batch get the metadata files the
    one-file-at-a-time path writes.  Segment choice is random in the reference; the batched path plans the
    member of that family in which every draw is the identity (cpx_plan_segments, pinned by
    tests/golden/segments_identity_golden.json), so the one-file path runs under the same draws here.  Also the
    explicit list form, ClipClassifier.process_files."""
    from cpx.classify.clipclassifier import ClipClassifier
    from helpers import IdentityDraws

    mdir, w = model_dir
    cfg = _config(mdir)
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    for d in (a, b, c):
        d.mkdir()
        for name in ("possum", "hedgehog"):
            shutil.copy(os.path.join(GOLDEN, name + ".cptv"), d / (name + ".cptv"))
    with IdentityDraws():
        for p in sorted(a.glob("*.cptv")):
            ClipClassifier(cfg).process_file(str(p), track=True, calculate_thumbnails=True)
        ClipClassifier(cfg).process_files(sorted(str(p) for p in c.glob("*.cptv")), calculate_thumbnails=True)
    cc = ClipClassifier(cfg)
    cc.process(str(b), track=True, calculate_thumbnails=True)
    assert cc.last_run["files"] == 2

    def load(p):
        with open(p) as fh:
            m = json.load(fh)
        for k in ("tracking_time", "source", "id"):
            m.pop(k, None)
        for model in m["models"]:
            model.pop("classify_time", None)
        for t in m["tracks"]:
            for pm in t["predictions"]:
                pm.pop("classify_time", None)
                for seg in pm.get("predictions", []):
                    seg.pop("predicted_time", None)  # wall clock
        return m

    for name in ("possum", "hedgehog"):
        ma = load(a / (name + ".txt"))
        assert len(ma["tracks"]) > 0 and ma["tracks"][0]["predictions"] and ma["tracks"][0]["thumbnail"]
        for other in (b, c):
            mb = load(other / (name + ".txt"))
            assert len(ma["tracks"]) == len(mb["tracks"]) and list(ma) == list(mb)
            for ta, tb in zip(ma["tracks"], mb["tracks"]):
                assert list(ta) == list(tb)
                for k in ta:
                    assert ta[k] == tb[k], (name, other.name, ta["id"], k)
            assert ma["models"] == mb["models"]
        assert ma == mb, name


def test_classify_existing_metadata_reuses_tracks_and_frames(tmp_path, model_dir):
    """process_file(track=False, reuse_frames=True) (clipclassifier.py:186-214): tracks come from the recording's
    metadata file, the frames are decoded and background-subtracted on the device without tracking, and the segments of
    the earlier run are reused -> the same class scores as the run that tracked."""
    from cpx.classify.clipclassifier import ClipClassifier

    mdir, w = model_dir
    cfg = _config(mdir)
    src = tmp_path / "hedgehog.cptv"
    shutil.copy(os.path.join(GOLDEN, "hedgehog.cptv"), src)
    first = ClipClassifier(cfg).process_file(str(src), track=True)
    assert first["tracks"] and first["tracks"][0]["predictions"]
    # the metadata file as an upstream tracker would leave it: predictions stored as tags with prediction_frames
    with open(src.with_suffix(".txt")) as fh:
        meta = json.load(fh)
    for t in meta["tracks"]:
        pm = t["predictions"][0]
        t["tags"] = [{"data": {"name": "wr-test", "prediction_frames": [p["frames"] for p in pm["predictions"]]}}]
    with open(src.with_suffix(".txt"), "w") as fh:
        json.dump(meta, fh)
    second = ClipClassifier(cfg).process_file(str(src), track=False, reuse_frames=True)
    assert len(second["tracks"]) == len(first["tracks"])
    for a, b in zip(first["tracks"], second["tracks"]):
        pa, pb = a["predictions"][0], b["predictions"][0]
        assert pa["all_class_confidences"] == pb["all_class_confidences"] and pa["tag"] == pb["tag"]
        assert [list(map(int, p["frames"])) for p in pa["predictions"]] == [
            list(map(int, p["frames"])) for p in pb["predictions"]]


def test_model_server_wire_format(model_dir):
    """cpx.servemodel: GET /ready, POST /predict with raw float32 [N,H,W,C] -> raw float32 [N,n_labels]
    (piclassifier/servemodel.py:17-36), and the run_over_network client (interpreter.py:53-62) against it."""
    import threading
    import urllib.request

    from cpx import servemodel
    from cpx.config.config import ModelConfig
    from cpx.ml_tools.interpreter import get_interpreter

    mdir, w = model_dir
    local = get_interpreter(ModelConfig.load({"id": 3, "name": "wr", "model_file": str(mdir / "wr.npz")}))
    server = servemodel.make_server(local, 0)
    port = server.server_address[1]
    th = threading.Thread(target=server.serve_forever, daemon=True)
    th.start()
    try:
        with urllib.request.urlopen("http://127.0.0.1:%d/ready" % port) as r:
            assert json.loads(r.read()) == {"ready": True}
        rng = np.random.default_rng(2)
        x = rng.uniform(0, 255, (3, 160, 160, 2)).astype(np.float32)
        want = local.predict(x)
        req = urllib.request.Request("http://127.0.0.1:%d/predict" % port, data=x.tobytes(),
                                     headers={"content-type": "application/octet-stream"}, method="POST")
        with urllib.request.urlopen(req) as r:
            assert r.headers["Content-Type"] == "application/octet-stream"
            got = np.frombuffer(r.read(), dtype=np.float32).reshape(3, -1)
        assert got.shape == want.shape and np.array_equal(got, want)
        remote = get_interpreter(ModelConfig.load({"id": 4, "name": "wr-net", "model_file": str(mdir / "wr.npz"),
                                                   "port": port}), run_over_network=True, load_model=False)
        assert np.array_equal(remote.predict(x), want)
        bad = urllib.request.Request("http://127.0.0.1:%d/predict" % port, data=b"123", method="POST")
        with pytest.raises(urllib.error.HTTPError):
            urllib.request.urlopen(bad)
    finally:
        server.shutdown()
        server.server_close()


def test_classify_service_socket(tmp_path, model_dir):
    """cpx.classifyservice: one JSON job per Unix-socket connection -> metadata JSON (classifyservice.py:73-122),
    a malformed job and a failing job -> {"error": ...}; the service survives both."""
    import socket
    import threading

    from cpx.classifyservice import ClassifyService

    mdir, w = model_dir
    cfg = _config(mdir)
    src = tmp_path / "hedgehog.cptv"
    shutil.copy(os.path.join(GOLDEN, "hedgehog.cptv"), src)
    path = str(tmp_path / "svc.sock")
    service = ClassifyService(cfg)
    th = threading.Thread(target=service.run, args=(path,), daemon=True)
    th.start()
    for _ in range(100):
        if os.path.exists(path):
            break
        import time
        time.sleep(0.05)

    def ask(payload):
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        s.connect(path)
        s.sendall(payload)
        chunks = []
        while True:
            b = s.recv(65536)
            if not b:
                break
            chunks.append(b)
        s.close()
        return json.loads(b"".join(chunks))

    try:
        assert "error" in ask(b'{"file": "x"}')                      # missing keys
        out = ask(json.dumps({"file": str(tmp_path / "missing.cptv"), "cache": None, "track": True,
                              "calculate_thumbnails": False}).encode())
        assert out is False or "error" in out                        # the reference's process_file returns False
        meta = ask(json.dumps({"file": str(src), "cache": None, "track": True,
                               "calculate_thumbnails": True}).encode())
        assert len(meta["tracks"]) == 1 and meta["tracks"][0]["predictions"][0]["model_id"] == 7
        assert meta["tracks"][0]["thumbnail"]["contours"] > 0
        assert os.path.exists(src.with_suffix(".txt"))
    finally:
        service.stop()


def test_process_directory_with_a_served_model(tmp_path, model_dir):
    """ClipClassifier.process(directory) with a model served over the network (run_over_network: what the families whose
    networks are not built here need): the bulk path has no device network for it, so the recordings go per file and
    their samples to the server -- the same predictions as the local model, one device pass per file."""
    import threading

    from cpx import servemodel
    from cpx.classify.clipclassifier import ClipClassifier
    from cpx.config import Config
    from cpx.config.config import ModelConfig
    from cpx.ml_tools.interpreter import get_interpreter

    mdir, w = model_dir
    local = get_interpreter(ModelConfig.load({"id": 3, "name": "wr", "model_file": str(mdir / "wr.npz")}))
    server = servemodel.make_server(local, 0)
    port = server.server_address[1]
    th = threading.Thread(target=server.serve_forever, daemon=True)
    th.start()
    try:
        a, b = tmp_path / "served", tmp_path / "local"
        for d in (a, b):
            d.mkdir()
            for name in ("possum", "hedgehog"):
                shutil.copy(os.path.join(GOLDEN, name + ".cptv"), d / (name + ".cptv"))
        metas = {}
        for d, served in ((a, True), (b, False)):
            cfg = Config.get_defaults()
            cfg.tracking["thermal"].denoise = False
            cfg.classify.models = [ModelConfig.load({"id": 7, "name": "wr-test", "model_file": str(mdir / "wr.npz"),
                                                     "run_over_network": served, "port": port})]
            cc = ClipClassifier(cfg)
            cc.process(str(d), track=True)
            for name in ("possum", "hedgehog"):
                with open(d / (name + ".txt")) as fh:
                    metas[(served, name)] = json.load(fh)
        for name in ("possum", "hedgehog"):
            ma, mb = metas[(True, name)], metas[(False, name)]
            assert len(ma["tracks"]) == len(mb["tracks"]) >= 1
            for ta, tb in zip(ma["tracks"], mb["tracks"]):
                assert len(ta["predictions"]) == len(tb["predictions"]) == 1
                # (process_file draws its segments at random, the bulk path plans them with identity draws: the two runs
                # classify different segments of the same track -- what must agree is that both classified it)
                assert ta["predictions"][0]["model_id"] == tb["predictions"][0]["model_id"] == 7
    finally:
        server.shutdown()
        server.server_close()
