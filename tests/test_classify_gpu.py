"""GPU parity of the classification pre-processing (limits, crop / resize /
normalise / 5x5 tiling): HIP kernels against the network inputs the reference's
own Interpreter.classify_track produced for the fixture clips (bit for bit) and
against the oracle on synthetic tracks."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, load_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3")
    yield eng
    eng.close()


def _requests(tracks_regions, segments_per_track, frame_of):
    """tracks_regions: list of region lists (objects with x,y,width,height,frame_number,blank);
    segments_per_track: list of [S,25] frame-number arrays -> (refs, offsets, reqs, n_samples)."""
    from cpx._lib import CROP_REQ_DTYPE, REGION_REF_DTYPE

    refs, offs, reqs = [], [0], []
    sample = 0
    for ti, (regions, segs) in enumerate(zip(tracks_regions, segments_per_track)):
        used = set(int(f) for s in segs for f in s)
        by_frame = {}
        for r in regions:
            by_frame[r.frame_number] = r
            if r.blank or r.width <= 0 or r.height <= 0:
                continue
            refs.append((frame_of(r.frame_number), r.x, r.y, r.width, r.height, 1 if r.frame_number in used else 0))
        offs.append(len(refs))
        for s in segs:
            for tile, fn in enumerate(s):
                r = by_frame[int(fn)]
                reqs.append((frame_of(int(fn)), r.x, r.y, r.width, r.height, ti, sample, tile))
            sample += 1
    return (np.array(refs, dtype=REGION_REF_DTYPE), np.array(offs, np.int32), np.array(reqs, dtype=CROP_REQ_DTYPE),
            sample)


@pytest.mark.parametrize("name,fs", [("possum", 32), ("hedgehog", 32), ("hedgehog", 64)])
def test_crop_tile_matches_reference_network_inputs(engine, name, fs):
    import track_oracle as to

    frames, t_on, ffc, bgf, hdr = load_clip(name)
    z = np.load(os.path.join(GOLDEN, "%s_classify_fs%d.npz" % (name, fs)))
    with open(os.path.join(GOLDEN, "%s_classify_fs%d.json" % (name, fs))) as fh:
        meta = json.load(fh)
    n = frames.shape[0]
    dev = engine.upload_frames(frames)
    offs = np.array([0, n], np.int32)
    m = engine.make_meta(n, t_on, ffc, bgf)
    res = engine.track_batch(dev, offs, m, want_filtered=True)
    res.check()
    # the tracks (regions) come from the oracle tracker == the reference's tracks (test_oracle_golden)
    out = to.track_clip(frames, t_on, ffc, bgf, to.OracleConfig(hdr.model), keep=True)
    proc = [i for i in range(n) if not bgf[i]]
    regions = [t.bounds for t in out["tracks"]]
    segs = [z["t%d_segments" % i] for i in range(len(regions))]
    refs, toffs, reqs, ns = _requests(regions, segs, lambda q: proc[q])
    x, limits = engine.preprocess_segments(dev, res, refs, toffs, reqs, ns, frame_size=fs)
    x = x.cpu().numpy()
    s0 = 0
    for i in range(len(regions)):
        want = z["t%d_input" % i]
        got = x[s0:s0 + want.shape[0]]
        assert np.array_equal(got, want), (name, i, float(np.abs(got - want).max()))
        s0 += want.shape[0]
    assert len(meta["tracks"]) == len(regions)


def test_crop_tile_matches_oracle_on_synthetic_tracks(engine):
    """Synthetic clips: regions from the oracle tracker, random 25-frame segments (with repeats),
    regions touching the frame edges included."""
    import classify_oracle as co
    import track_oracle as to
    from cpx import synth

    rng = np.random.default_rng(21)
    clips = [synth.make_clip(rng, 120, max_blobs=3) for _ in range(4)]
    T = 120
    offs = (np.arange(5) * T).astype(np.int32)
    allf = np.concatenate(clips)
    dev = engine.upload_frames(allf)
    res = engine.track_batch(dev, offs, engine.make_meta(4 * T), want_filtered=True)
    res.check()
    tracks_regions, segs_all, frame_of_track, oracle_x = [], [], [], []
    for b in range(4):
        out = to.track_clip(clips[b], cfg=to.OracleConfig("lepton3"), keep=True)
        fr = out["frames"]
        for t in out["tracks"]:
            valid = [r.frame_number for r in t.bounds if not r.blank and r.mass > 0 and r.width > 0 and r.height > 0]
            if len(valid) < 6:
                continue
            segs = [np.sort(rng.choice(valid, 25, replace=True)) for _ in range(2)]
            by_frame = {r.frame_number: r for r in t.bounds}
            x, _ = co.preprocess_segments(lambda q: clips[b][q], lambda q: fr[q]["filtered"].astype(np.float64),
                                          by_frame, t.bounds, segs, 32, (1, 1, 158, 118))
            tracks_regions.append(t.bounds)
            segs_all.append(segs)
            frame_of_track.append(b)
            oracle_x.append(x)
    assert len(tracks_regions) >= 3
    # frame index mapping differs per track (clip b): build requests track by track
    from cpx._lib import CROP_REQ_DTYPE, REGION_REF_DTYPE

    refs_l, offs_l, reqs_l, sample = [], [0], [], 0
    for ti, (regions, segs) in enumerate(zip(tracks_regions, segs_all)):
        base = int(offs[frame_of_track[ti]])
        r, o, q, ns = _requests([regions], [segs], lambda fn: base + fn)
        q["track"] = ti
        q["sample"] += sample
        sample += ns
        refs_l.append(r)
        offs_l.append(offs_l[-1] + len(r))
        reqs_l.append(q)
    x, limits = engine.preprocess_segments(dev, res, np.concatenate(refs_l), np.array(offs_l, np.int32),
                                           np.concatenate(reqs_l), sample)
    x = x.cpu().numpy()
    s0 = 0
    for want in oracle_x:
        got = x[s0:s0 + want.shape[0]]
        assert np.array_equal(got, want), float(np.abs(got - want).max())
        s0 += want.shape[0]


def test_limits_and_clip_at_zero_on_crafted_crops(engine):
    """cpx_track_limits_batch on crafted frames: per 'track' one in-segment region whose crop median sits below, at,
    just above (by 0.5) and above the frame median, odd and even pixel counts, and the straddling case where the two
    middle order statistics fall on both sides of the frame median.  Expectation: NumPy, as the reference evaluates
    it (interpreter.py:372-399: np.median(float32(crop) - np.median(frame)) <= 0 turns clipping off)."""
    from cpx._lib import CROP_REQ_DTYPE, REGION_REF_DTYPE

    rng = np.random.default_rng(99)
    H, W = 120, 160
    cases = []
    frames = []
    for trial in range(60):
        f = rng.integers(2950, 3050, (H, W)).astype(np.uint16)
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        x, y = int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1))
        med = np.median(f)
        kind = trial % 6
        crop = f[y:y + h, x:x + w]
        n = w * h
        if kind == 0:      # clearly above
            crop += 40
        elif kind == 1:    # clearly below
            crop -= 40
        elif kind == 2:    # constant crop exactly at floor(median)
            crop[:] = int(np.floor(med))
        elif kind == 3:    # constant crop one above floor(median)
            crop[:] = int(np.floor(med)) + 1
        elif kind == 4 and n >= 2:  # half low, half high around the median: straddle (even n) or decided by the middle
            flat = np.empty(n, np.uint16)
            flat[: n // 2] = int(np.floor(med)) - int(rng.integers(0, 3))
            flat[n // 2:] = int(np.floor(med)) + int(rng.integers(0, 4))
            crop[:] = rng.permutation(flat).reshape(h, w)
        frames.append(f)
        cases.append((x, y, w, h))
    frames = np.stack(frames)
    n = len(cases)
    dev = engine.upload_frames(frames)
    res = engine.track_batch(dev, np.arange(n + 1, dtype=np.int32), engine.make_meta(n), want_filtered=True)
    res.check()
    filt = res.filtered()
    refs = np.zeros(n, REGION_REF_DTYPE)
    for i, (x, y, w, h) in enumerate(cases):
        refs[i] = (i, x, y, w, h, 1)
    reqs = np.zeros(0, CROP_REQ_DTYPE)
    _, limits = engine.preprocess_segments(dev, res, refs, np.arange(n + 1, dtype=np.int32), reqs, 0)
    seen = set()
    for i, (x, y, w, h) in enumerate(cases):
        th = frames[i]
        sub = np.float32(th[y:y + h, x:x + w]) - np.median(th)
        want_clip = 0 if np.median(sub) <= 0 else 1
        assert limits[i]["clip_at_zero"] == want_clip, (i, cases[i], float(np.median(sub)))
        fsub = filt[i][y:y + h, x:x + w]
        assert limits[i]["filt_min"] == fsub.min() and limits[i]["filt_max"] == max(0.0, float(fsub.max()))
        seen.add((want_clip, float(np.median(sub)) == 0.0, (w * h) % 2))
    assert {s[0] for s in seen} == {0, 1} and any(s[1] for s in seen)


def test_crop_tile_matches_reference_on_busy_scenes(engine, golden_dir):
    """cpx_track_limits_batch + cpx_crop_tile against the CRC32 of the network inputs the REFERENCE built for 19
    tracks of two busy synthetic scenes (tests/golden/busy_classify_fs32.json): regions on every frame border."""
    import json
    import os

    import track_oracle as to
    from cpx import synth
    from helpers import crc

    with open(os.path.join(golden_dir, "busy_classify_fs32.json")) as fh:
        gold = json.load(fh)
    T = gold["frames"]
    t_on, ffc = [100000 + 111 * i for i in range(T)], [40000] * T
    n = 0
    for c in gold["clips"]:
        frames = synth.make_clip(np.random.default_rng(1000 + c["seed"]), T, max_blobs=8)
        dev = engine.upload_frames(frames)
        res = engine.track_batch(dev, np.array([0, T], np.int32), engine.make_meta(T, t_on, ffc), want_filtered=True)
        res.check()
        out = to.track_clip(frames, t_on, ffc, None, to.OracleConfig("lepton3"), keep=True)  # tracks / regions only
        births = {(t.start_frame, t.bounds[0].x, t.bounds[0].y, t.bounds[0].width, t.bounds[0].height): t
                  for t in out["tracks"]}
        regions, segs, want = [], [], []
        for g in c["tracks"]:
            t = births[(g["start_frame"],) + tuple(g["first"])]
            regions.append(t.bounds)
            segs.append([np.array(s) for s in g["segments"]])
            want.extend(g["crc"])
        refs, toffs, reqs, ns = _requests(regions, segs, lambda fn: fn)
        x, _ = engine.preprocess_segments(dev, res, refs, toffs, reqs, ns, frame_size=32)
        x = x.cpu().numpy()
        assert [crc(x[i]) for i in range(ns)] == want, c["seed"]
        n += ns
    assert n >= 30


@pytest.mark.parametrize("name,fs", [("possum", 32), ("hedgehog", 32), ("hedgehog", 64)])
def test_aggregate_kernel_matches_reference_scores(engine, golden_dir, name, fs):
    """cpx_aggregate_predictions on the per-segment predictions of the classify goldens -> the class_best_score the
    reference's TrackPrediction produced (sum over segments, normalisation, low-evidence cap)."""
    import ctypes as C
    import json
    import os

    import torch

    from cpx._lib import CROP_REQ_DTYPE

    z = np.load(os.path.join(golden_dir, "%s_classify_fs%d.npz" % (name, fs)))
    with open(os.path.join(golden_dir, "%s_classify_fs%d.json" % (name, fs))) as fh:
        gold = json.load(fh)
    labels = gold["labels"]
    probs, sample_track, reqs, want = [], [], [], []
    for ti, t in enumerate(gold["tracks"]):
        p, segs = z["t%d_pred" % ti], z["t%d_segments" % ti]
        for s in range(p.shape[0]):
            probs.append(p[s])
            sample_track.append(ti)
            for tile, fn in enumerate(segs[s]):
                reqs.append((int(fn), 0, 0, 1, 1, ti, len(probs) - 1, tile))
        want.append(t["class_best_score"])
    dev = engine.device
    probs_d = torch.from_numpy(np.asarray(probs, np.float32)).to(dev)
    st_d = torch.from_numpy(np.asarray(sample_track, np.int32)).to(dev)
    reqs_d = engine._to_dev(np.array(reqs, dtype=CROP_REQ_DTYPE))
    n_tracks, L = len(want), len(labels)
    scores = torch.zeros((n_tracks, L), dtype=torch.float32, device=dev)
    best = torch.zeros(n_tracks, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    rc = engine.lib.cpx_aggregate_predictions(
        engine.h, C.c_void_p(probs_d.data_ptr()), C.c_void_p(st_d.data_ptr()), len(probs), C.c_void_p(reqs_d.data_ptr()),
        n_tracks, L, labels.index("false-positive"), 5, C.c_void_p(scores.data_ptr()), C.c_void_p(best.data_ptr()))
    assert rc == 0, engine._err()
    engine.synchronize()
    got = scores.cpu().numpy()
    assert np.abs(got - np.asarray(want)).max() <= 2e-7
    assert list(best.cpu().numpy()) == [int(np.argmax(w)) for w in want]
