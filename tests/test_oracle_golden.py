"""The oracle restatement (oracle/track_oracle.py) against the vectors the
reference itself produced (tests/golden/make_golden.py) -- every frame of both
fixture clips, every stage, bit for bit; and against the reference's own golden
tests/clips/possum.txt."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, crc, load_clip, load_golden


def _run(name, dn):
    import track_oracle as to

    frames, t_on, ffc, bgf, hdr = load_clip(name)
    cfg = to.OracleConfig(hdr.model)
    cfg.denoise = bool(dn)
    return to.track_clip(frames, t_on, ffc, bgf, cfg, keep=True), frames


@pytest.mark.parametrize("name,dn", [("possum", 0), ("hedgehog", 0), ("possum", 1)])
def test_oracle_matches_reference_vectors(name, dn):
    out, frames = _run(name, dn)
    z, gt = load_golden(name, dn)
    fr = out["frames"]
    assert len(fr) == int(z["n_frames"])
    assert np.array_equal(out["init_bg"].astype(np.int32), z["init_bg"])
    assert float(out["init_avg"]) == float(z["init_avg"])
    for i, f in enumerate(fr):
        assert f["ffc"] == bool(z["ffc"][i]), i
        assert f["bg_used_avg"] == z["bg_used_avg"][i], i
        assert f["bg_after_avg"] == z["bg_after_avg"][i], i
        assert f["threshold"] == z["threshold"][i], i
        assert crc(f["filtered"]) == z["crc_filtered"][i], i
        if dn == 0:
            assert crc(f["obj_u8"]) == z["crc_obj_u8"][i], i
        assert crc(f["mask"].astype(np.int32)) == z["crc_mask"][i], i
        assert crc(f["bg_after"]) == z["crc_bg_after"][i], i
        assert crc(f["weight_after"].astype(np.float64)) == z["crc_weight_after"][i], i
        a, b = z["comp_offsets"][i], z["comp_offsets"][i + 1]
        assert f["n_components"] == b - a, i
        assert np.array_equal(f["stats"], z["comp_stats"][a:b]), i
        assert np.array_equal(f["centroids"], z["comp_centroids"][a:b]), i
        a, b = z["region_offsets"][i], z["region_offsets"][i + 1]
        got = np.array(
            [
                (r.x, r.y, r.width, r.height, r.mass, r.id, float(r.pixel_variance), r.was_cropped,
                 r.is_along_border, float(r.centroid[0]), float(r.centroid[1]))
                for r in f["regions"]
            ],
            dtype=np.float64,
        ).reshape(-1, 11)
        assert np.array_equal(got, z["regions"][a:b]), i
    # final tracks
    assert len(out["tracks"]) == len(gt["tracks"])
    for t, g in zip(out["tracks"], gt["tracks"]):
        assert (t.id, t.start_frame, t.end_frame) == (g["id"], g["start_frame"], g["end_frame"])
        assert t.stats["score"] == g["score"]
        assert len(t.bounds) == len(g["positions"])
        for r, p in zip(t.bounds, g["positions"]):
            assert (r.x, r.y, r.width, r.height, int(r.mass), r.frame_number, r.blank) == (
                p["x"], p["y"], p["width"], p["height"], p["mass"], p["frame_number"], p["blank"])
            assert float(r.pixel_variance) == p["pixel_variance"]
    assert [(r, t.id) for r, t in out["filtered_tracks"]] == [(f["reason"], f["id"]) for f in gt["filtered"]]


def test_oracle_matches_possum_txt():
    """Default config (denoise on) vs the reference's committed golden metadata."""
    out, _ = _run("possum", 1)
    with open(os.path.join(GOLDEN, "possum.txt")) as fh:
        gold = json.load(fh)
    assert len(out["tracks"]) == len(gold["tracks"]) == 2
    for t, g in zip(out["tracks"], gold["tracks"]):
        assert t.id == g["id"]
        assert t.start_frame == g["frame_start"] and t.end_frame == g["frame_end"]
        assert len(t.bounds) == g["num_frames"]
        assert abs(t.stats["score"] - g["tracking_score"]) <= 1e-6 * g["tracking_score"]
        for r, p in zip(t.bounds, g["positions"]):
            assert (r.x, r.y, r.width, r.height, int(r.mass), r.frame_number, r.blank) == (
                p["x"], p["y"], p["width"], p["height"], p["mass"], p["frame_number"], p["blank"])
            assert abs(round(float(r.pixel_variance), 2) - p["pixel_variance"]) < 1e-3


@pytest.mark.parametrize("name,fs", [("possum", 32), ("hedgehog", 32), ("hedgehog", 64)])
def test_classify_oracle_matches_reference_inputs(name, fs):
    """oracle/classify_oracle.py against the network inputs the reference's own
    Interpreter.classify_track built (tests/golden/make_golden_classify.py)."""
    import classify_oracle as co

    out, frames = _run(name, 0)
    z = np.load(os.path.join(GOLDEN, "%s_classify_fs%d.npz" % (name, fs)))
    with open(os.path.join(GOLDEN, "%s_classify_fs%d.json" % (name, fs))) as fh:
        meta = json.load(fh)
    fr = out["frames"]
    proc_thermal = {f["index"]: None for f in fr}
    _, _, _, bgf, _ = load_clip(name)
    proc = [i for i in range(frames.shape[0]) if not bgf[i]]
    thermal_of = lambda q: frames[proc[q]]
    filtered_of = lambda q: fr[q]["filtered"].astype(np.float64)
    H, W = frames.shape[1:]
    crop = (1, 1, W - 2, H - 2)
    assert len(out["tracks"]) == len(meta["tracks"])
    for ti, (t, m) in enumerate(zip(out["tracks"], meta["tracks"])):
        assert t.id == m["track_id"]
        segs = z["t%d_segments" % ti]
        by_frame = {r.frame_number: r for r in t.bounds}
        x, info = co.preprocess_segments(thermal_of, filtered_of, by_frame, t.bounds, segs, fs, crop)
        want = z["t%d_input" % ti]
        assert x.shape == want.shape
        assert np.array_equal(x, want), (name, ti, np.abs(x - want).max())
        score = co.classified_track(z["t%d_pred" % ti], prediction_frames=segs, labels=meta["labels"])
        assert np.array_equal(score, np.array(m["class_best_score"]))
