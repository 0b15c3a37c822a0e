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


@pytest.mark.parametrize("name,dn", [("possum", 0), ("hedgehog", 0), ("possum", 1), ("synth35", 0)])
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


@pytest.mark.parametrize("name,dn", [("possum", 0), ("hedgehog", 0), ("possum", 1)])
def test_thumbnail_oracle_matches_reference(name, dn):
    """oracle/thumbnail_oracle.py (+ the findContours / TC89_L1 restatement in cv2_shim) against the
    per-frame thumbnail statistics the reference produced, and for the default config against the
    thumbnail entries of the reference's own golden possum.txt."""
    import thumbnail_oracle as th

    out, frames = _run(name, dn)
    fr = out["frames"]
    with open(os.path.join(GOLDEN, "%s_dn%d_thumbs.json" % (name, dn))) as fh:
        gold = json.load(fh)
    _, t_on, ffc, bgf, hdr = load_clip(name)
    proc = [i for i in range(frames.shape[0]) if not bgf[i]]
    assert len(out["tracks"]) == len(gold["tracks"])
    for t, g in zip(out["tracks"], gold["tracks"]):
        stats, max_mass, max_md, min_md, max_contour = th.track_thumb_stats(
            t.bounds, lambda q: fr[q]["mask"], lambda q: frames[proc[q]])
        assert [[s.region.frame_number, s.contours, float(s.median_diff)] for s in stats] == g["stats"]
        assert (max_mass, max_md, min_md, max_contour) == (
            g["max_mass"], g["max_median_diff"], g["min_median_diff"], g["max_contour"])
        best, score = th.thumbnail_info(t.bounds, lambda q: fr[q]["mask"], lambda q: frames[proc[q]])
        gb = g["best"]
        assert best.region.frame_number == gb["region"]["frame_number"]
        assert (best.contours, float(best.median_diff), score) == (gb["contours"], gb["median_diff"], gb["score"])
    if (name, dn) == ("possum", 1):
        with open(os.path.join(GOLDEN, "possum.txt")) as fh:
            ref_gold = json.load(fh)
        for t, g in zip(out["tracks"], ref_gold["tracks"]):
            best, score = th.thumbnail_info(t.bounds, lambda q: fr[q]["mask"], lambda q: frames[proc[q]])
            gt = g["thumbnail"]
            r = best.region
            assert (r.x, r.y, r.width, r.height, int(r.mass), r.frame_number) == tuple(
                gt["region"][k] for k in ("x", "y", "width", "height", "mass", "frame_number"))
            assert (best.contours, float(best.median_diff), round(score)) == (
                gt["contours"], gt["median_diff"], gt["score"])


@pytest.mark.parametrize("name,n", [("possum", 30), ("hedgehog", 8)])
def test_trackless_thumbnail_oracle_matches_reference(name, n):
    import thumbnail_oracle as th
    import track_oracle as to

    frames, t_on, ffc, bgf, hdr = load_clip(name)
    with open(os.path.join(GOLDEN, "%s_trackless_thumbs.json" % name)) as fh:
        gold = json.load(fh)
    frames, t_on, ffc, bgf = frames[:n], t_on[:n], ffc[:n], bgf[:n]
    out = to.track_clip(frames, t_on, ffc, bgf, to.OracleConfig(hdr.model), keep=True)
    assert len(out["tracks"]) == 0 and sum(len(r) for r in out["region_history"]) == gold["n_region_history"]
    proc = [i for i in range(n) if not bgf[i]]
    means = [frames[i].mean() for i in proc]
    x, y, w, h, fn, centroid, mass = th.trackless_thumb(out["region_history"], means, lambda q: frames[proc[q]],
                                                        frames[0])
    g = gold["trackless"]
    assert (x, y, w, h, fn, mass) == (g["x"], g["y"], g["width"], g["height"], g["frame_number"], g["mass"])
    assert [float(centroid[0]), float(centroid[1])] == g["centroid"]


def _busy_golden():
    z = np.load(os.path.join(GOLDEN, "busy_tracks.npz"))
    return z["seeds"], z["rows"], z["offsets"], int(z["frames"])


def busy_clip(seed, T):
    from cpx import synth

    return synth.make_clip(np.random.default_rng(1000 + int(seed)), T, max_blobs=8)


def test_oracle_matches_reference_on_busy_scenes():
    """Association / Kalman / blank regions beyond the fixture clips: every track the reference created on 11 seeded
    busy scenes (tests/golden/make_golden_busy.py), bounds and the Python-int typing of widths / heights included."""
    import track_oracle as to

    seeds, rows, offsets, T = _busy_golden()
    t_on, ffc = [100000 + 111 * i for i in range(T)], [40000] * T
    n_py = 0
    for k, seed in enumerate(seeds):
        out = to.track_clip(busy_clip(seed, T), t_on, ffc, None, to.OracleConfig("lepton3"), keep=True)
        mine = sorted(list(out["tracks"]) + [t for _, t in out["filtered_tracks"]], key=lambda t: t.id)
        got = [(t.id, r.x, r.y, r.width, r.height, int(r.mass), r.frame_number, int(bool(r.blank)), int(r.py[2]),
                int(r.py[3])) for t in mine for r in t.bounds]
        want = rows[offsets[k]:offsets[k + 1]]
        assert np.array_equal(np.asarray(got, np.int32), want), seed
        n_py += int((want[:, 8] | want[:, 9]).sum())
    assert n_py > 10


def _busy_classify_cases():
    """(clip frames, oracle output, [(oracle track, golden track entry)]) per clip of busy_classify_fs32.json; tracks
    are paired by birth (start frame + first box): the reference numbers same-frame births in set order (F14)."""
    import track_oracle as to

    with open(os.path.join(GOLDEN, "busy_classify_fs32.json")) as fh:
        gold = json.load(fh)
    T = gold["frames"]
    t_on, ffc = [100000 + 111 * i for i in range(T)], [40000] * T
    cases = []
    for c in gold["clips"]:
        frames = busy_clip(c["seed"], T)
        out = to.track_clip(frames, t_on, ffc, None, to.OracleConfig("lepton3"), keep=True)
        births = {(t.start_frame, t.bounds[0].x, t.bounds[0].y, t.bounds[0].width, t.bounds[0].height): t
                  for t in out["tracks"]}
        pairs = []
        for g in c["tracks"]:
            t = births.get((g["start_frame"],) + tuple(g["first"]))
            assert t is not None, (c["seed"], g["id"])
            pairs.append((t, g))
        assert len(pairs) == len(out["tracks"])
        cases.append((frames, out, pairs))
    return cases


def test_classify_oracle_matches_reference_on_busy_scenes():
    """Network inputs for 19 tracks of two busy synthetic scenes (many regions on the frame border: the edge-anchored
    branches of resize_and_pad) against the CRCs of what the reference's Interpreter.classify_track fed its model."""
    import classify_oracle as co

    n_samples = n_edge = 0
    for frames, out, pairs in _busy_classify_cases():
        fr = out["frames"]
        for t, g in pairs:
            by_frame = {r.frame_number: r for r in t.bounds}
            segs = [np.array(s) for s in g["segments"]]
            x, _ = co.preprocess_segments(lambda q: frames[q], lambda q: fr[q]["filtered"].astype(np.float64),
                                          by_frame, t.bounds, segs, 32, (1, 1, 158, 118))
            assert [crc(x[i]) for i in range(x.shape[0])] == g["crc"], g["id"]
            n_samples += x.shape[0]
            n_edge += g["edge_regions"]
    assert n_samples >= 30 and n_edge > 100


def test_thumbnail_oracle_matches_reference_on_busy_scenes():
    """Per-frame contour counts / median differences and the chosen thumbnail for the ~20 tracks of two busy synthetic
    scenes (tests/golden/busy_thumbs.json: reference run, make_golden_thumbs.py --busy): ~700 more contours for the
    findContours + TC89_L1 restatement than the fixture clips hold."""
    import thumbnail_oracle as th
    import track_oracle as to

    with open(os.path.join(GOLDEN, "busy_thumbs.json")) as fh:
        gold = json.load(fh)
    T = gold["frames"]
    t_on, ffc = [100000 + 111 * i for i in range(T)], [40000] * T
    n = 0
    for c in gold["clips"]:
        frames = busy_clip(c["seed"], T)
        out = to.track_clip(frames, t_on, ffc, None, to.OracleConfig("lepton3"), keep=True)
        fr = out["frames"]
        births = {(t.start_frame, t.bounds[0].x, t.bounds[0].y, t.bounds[0].width, t.bounds[0].height): t
                  for t in out["tracks"]}
        assert len(c["tracks"]) == len(out["tracks"])
        for g in c["tracks"]:
            t = births[(g["start_frame"],) + tuple(g["first"])]
            stats, max_mass, max_md, min_md, max_contour = th.track_thumb_stats(
                t.bounds, lambda q: fr[q]["mask"], lambda q: frames[q])
            assert [[s.region.frame_number, s.contours, float(s.median_diff)] for s in stats] == g["stats"], g["id"]
            best, score = th.thumbnail_info(t.bounds, lambda q: fr[q]["mask"], lambda q: frames[q])
            gb = g["best"]
            assert (best.region.frame_number, best.contours, float(best.median_diff), score) == (
                gb["region"]["frame_number"], gb["contours"], gb["median_diff"], gb["score"])
            n += len(stats)
    assert n > 500


def test_oracle_track_scores_and_rejects_on_busy_scenes():
    """End-of-clip statistics, score order and reject reasons of the busy scenes against the reference
    (tests/golden/busy_tracks_info.json)."""
    import track_oracle as to

    with open(os.path.join(GOLDEN, "busy_tracks_info.json")) as fh:
        info = json.load(fh)
    _, _, _, T = _busy_golden()
    t_on, ffc = [100000 + 111 * i for i in range(T)], [40000] * T
    n_rej = 0
    for c in info:
        out = to.track_clip(busy_clip(c["seed"], T), t_on, ffc, None, to.OracleConfig("lepton3"), keep=True)
        got = [[t.id, t.stats["score"], t.stats["frames_moved"], t.stats["max_offset"], t.stats["average_mass"],
                t.stats["delta_std"]] for t in out["tracks"]]
        assert len(got) == len(c["kept"])
        for g, w in zip(got, c["kept"]):
            assert g[0] == w[0] and g[2] == w[2], c["seed"]
            assert g[1] == pytest.approx(w[1], rel=1e-12) and g[3] == pytest.approx(w[3], rel=1e-12)
            assert g[4] == pytest.approx(w[4], rel=1e-12) and g[5] == pytest.approx(w[5], rel=1e-6)
        assert [[r, t.id] for r, t in out["filtered_tracks"]] == c["filtered"], c["seed"]
        n_rej += len(c["filtered"])
    assert n_rej > 0
