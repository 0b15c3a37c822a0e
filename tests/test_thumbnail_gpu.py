"""Thumbnail stage through the C-ABI: cpx_thumb_stats / cpx_trackless_thumb against
(1) the vectors the reference produced (tests/golden/*_thumbs.json) and its own golden possum.txt,
(2) the oracle (findContours + TC89_L1 restatement) on random masks.  Integer work: everything exact."""
import json
import os

import numpy as np
import pytest

from helpers import load_clip

pytestmark = pytest.mark.gpu


def _config(denoise):
    from cpx.config import Config

    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = bool(denoise)
    return cfg


def _extract(golden_dir, name, denoise, tmp_path):
    import shutil

    from cpx.track.trackextractor import extract_file

    src = os.path.join(golden_dir, name + ".cptv")
    dst = tmp_path / (name + ".cptv")
    shutil.copy(src, dst)
    return extract_file(dst, _config(denoise), cache_to_disk=False, save_meta=False)


@pytest.mark.parametrize("name,dn", [("possum", 0), ("hedgehog", 0), ("possum", 1), ("hedgehog", 1)])
def test_thumbnail_stats_match_reference(golden_dir, tmp_path, name, dn):
    from cpx.classify.thumbnail import get_thumbnail_info, get_track_thumb_stats

    clip, ex, meta = _extract(golden_dir, name, dn, tmp_path)
    with open(os.path.join(golden_dir, "%s_dn%d_thumbs.json" % (name, dn))) as fh:
        gold = json.load(fh)
    assert len(clip.tracks) == len(gold["tracks"]) > 0
    for track, g, tm in zip(clip.tracks, gold["tracks"], meta["tracks"]):
        stats, max_mass, max_md, min_md, max_contour = get_track_thumb_stats(clip, track)
        assert [[s.region.frame_number, s.contours, float(s.median_diff)] for s in stats] == g["stats"]
        assert (max_mass, max_md, min_md, max_contour) == (
            g["max_mass"], g["max_median_diff"], g["min_median_diff"], g["max_contour"])
        best, score = get_thumbnail_info(clip, track)
        gb = g["best"]
        assert best.region.frame_number == gb["region"]["frame_number"]
        assert (best.contours, float(best.median_diff)) == (gb["contours"], gb["median_diff"])
        assert score == pytest.approx(gb["score"], rel=1e-12)
        assert tm["thumbnail"]["score"] == round(gb["score"]) and tm["thumbnail"]["contours"] == gb["contours"]


def test_default_config_thumbnails_equal_the_references_own_golden(golden_dir, tmp_path):
    clip, ex, meta = _extract(golden_dir, "possum", True, tmp_path)
    from cpx.ml_tools.tools import CustomJSONEncoder

    meta = json.loads(json.dumps(meta, cls=CustomJSONEncoder))
    with open(os.path.join(golden_dir, "possum.txt")) as fh:
        gold = json.load(fh)
    for t, g in zip(meta["tracks"], gold["tracks"]):
        assert t["thumbnail"] == g["thumbnail"]


@pytest.mark.parametrize("name,n", [("possum", 30), ("hedgehog", 8)])
def test_trackless_thumbnail_matches_reference(golden_dir, tmp_path, name, n):
    """A clip cut to its first frames has no tracks and no regions: the 64x64 window search decides."""
    from cpx.classify.thumbnail import best_trackless_thumb
    from cpx.track.trackextractor import extract_file
    from helpers import encode_cptv

    frames, t_on, ffc, bgf, hdr = load_clip(name)
    p = tmp_path / "short.cptv"
    encode_cptv(p, frames[:n], [16] * n, time_on=t_on[:n] if t_on[0] is not None else None,
                last_ffc=ffc[:n] if t_on[0] is not None else None, model=hdr.model.encode() if hdr.model else None,
                background_first=bool(bgf[0]))
    clip, ex, meta = extract_file(p, _config(False), cache_to_disk=False, save_meta=False)
    with open(os.path.join(golden_dir, "%s_trackless_thumbs.json" % name)) as fh:
        g = json.load(fh)["trackless"]
    assert len(clip.tracks) == 0
    r = best_trackless_thumb(clip)
    assert (r.x, r.y, r.width, r.height, r.frame_number, r.mass) == (
        g["x"], g["y"], g["width"], g["height"], g["frame_number"], g["mass"])
    assert [float(r.centroid[0]), float(r.centroid[1])] == g["centroid"]
    mr = meta["thumbnail_region"]
    assert (mr.x, mr.y) == (g["x"], g["y"])


def test_random_masks_match_oracle():
    """Random blobs / noise / thin structures / nested shapes: contour point counts and masked medians of
    the kernel equal the oracle's for every region."""
    import thumbnail_oracle as th
    import track_oracle as to
    from cpx._lib import REGION_REF_DTYPE
    from cpx.engine import TrackEngine
    from scipy import ndimage

    rng = np.random.default_rng(11)
    H, W, N = 120, 160, 24
    eng = TrackEngine(width=W, height=H, max_frames=64)
    frames = rng.integers(2800, 3400, (N, H, W)).astype(np.uint16)
    labels = np.zeros((N, H, W), np.int32)
    for i in range(N):
        kind = i % 4
        if kind == 0:      # smooth blobs
            m = ndimage.gaussian_filter(rng.standard_normal((H, W)), 3 + i // 4) > 0.35
        elif kind == 1:    # salt noise + lines
            m = rng.random((H, W)) > 0.6
        elif kind == 2:    # rings (holes with islands)
            yy, xx = np.mgrid[:H, :W]
            d = np.hypot(yy - 60 - i, xx - 80 + i)
            m = ((d < 50) & (d > 38)) | ((d < 30) & (d > 22)) | (d < 9)
        else:              # thin diagonal / comb structures
            yy, xx = np.mgrid[:H, :W]
            m = ((xx + yy) % (5 + i % 3) == 0) | ((yy % 7 == 0) & (xx % 2 == 0))
        labels[i] = np.where(m, rng.integers(1, 60, (H, W)), 0)
    frames_dev = eng.upload_frames(frames)
    res = eng.track_batch(frames_dev, np.array([0, N], np.int32), eng.make_meta(N), want_labels=True)
    eng.synchronize()
    import torch

    res.labels_dev.copy_(torch.from_numpy(labels).to(res.labels_dev.device))  # our masks instead of the tracker's
    info = res.info
    boxes = []
    for i in range(N):
        boxes.append((i, 0, 0, W, H))
        for _ in range(12):
            w, h = int(rng.integers(1, 70)), int(rng.integers(1, 60))
            x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
            boxes.append((i, x, y, w, h))
    refs = np.zeros(len(boxes), REGION_REF_DTYPE)
    for k, b in enumerate(boxes):
        refs[k] = b + (0,)
    got = eng.thumb_stats(frames_dev, res, refs)
    n_with = 0
    for k, (i, x, y, w, h) in enumerate(boxes):
        region = to.Region(x, y, w, h, mass=1, frame_number=i)
        want = th.region_stat(region, labels[i], frames[i])
        if want is None:
            assert got[k]["contours"] == 0, k
            continue
        n_with += 1
        assert got[k]["contours"] == want[0], (k, boxes[k])
        assert float(info[i]["thermal_median"]) == float(np.median(frames[i]))
        assert got[k]["median_diff"] == float(want[1]), (k, boxes[k])
    assert n_with > 200
    eng.close()


def test_thumbnail_stats_match_reference_on_busy_scenes(golden_dir, tmp_path):
    """The thumbnail kernels against the reference's per-frame statistics for ~20 tracks of two busy synthetic scenes
    (tests/golden/busy_thumbs.json)."""
    from cpx import synth
    from cpx.classify.thumbnail import get_thumbnail_info, get_track_thumb_stats
    from cpx.track.trackextractor import extract_file
    from helpers import encode_cptv

    with open(os.path.join(golden_dir, "busy_thumbs.json")) as fh:
        gold = json.load(fh)
    T = gold["frames"]
    n = 0
    for c in gold["clips"]:
        frames = synth.make_clip(np.random.default_rng(1000 + c["seed"]), T, max_blobs=8)
        p = tmp_path / ("busy%d.cptv" % c["seed"])
        encode_cptv(p, frames, [16] * T, time_on=[100000 + 111 * i for i in range(T)], last_ffc=[40000] * T,
                    model=b"lepton3")
        clip, ex, meta = extract_file(p, _config(False), cache_to_disk=False, save_meta=False)
        births = {(t.start_frame, t.bounds_history[0].x, t.bounds_history[0].y, t.bounds_history[0].width,
                   t.bounds_history[0].height): t for t in clip.tracks}
        assert len(births) == len(c["tracks"])
        for g in c["tracks"]:
            track = births[(g["start_frame"],) + tuple(g["first"])]
            stats, max_mass, max_md, min_md, max_contour = get_track_thumb_stats(clip, track)
            assert [[s.region.frame_number, s.contours, float(s.median_diff)] for s in stats] == g["stats"], g["id"]
            best, score = get_thumbnail_info(clip, track)
            gb = g["best"]
            assert (best.region.frame_number, best.contours, float(best.median_diff)) == (
                gb["region"]["frame_number"], gb["contours"], gb["median_diff"])
            assert score == pytest.approx(gb["score"], rel=1e-12)
            n += len(stats)
    assert n > 500
