"""The IR tracker end to end on the GPU (BASELINE configs[4]): IRTrackExtractor.parse_frames on a
seeded synthetic 640 x 480 video against the oracle chain -- MOG2 restatement (C) -> detect_objects_ir / merge_components
(NumPy, both pinned by the reference-generated ir_detect_golden.json) -> NumPy uint8-wrapping delta variance -> the
association restatement (pinned by the thermal goldens) with the IR TrackingConfig.  What no reference run can pin
(its own frame loop raises at this snapshot, see cpx/track/irtrackextractor.py) is stated there.  Plus the mixed
lepton3.5 + IR batch of configs[4]: both cameras' clips on one GPU, each equal to its own single-camera run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def ir_video(seed, n=60, blobs=2):
    rng = np.random.default_rng(seed)
    H, W = 480, 640
    yy, xx = np.mgrid[0:H, 0:W]
    scene = (90 + 40 * np.sin(xx / 37.0) * np.cos(yy / 53.0) + rng.normal(0, 6, (H, W))).clip(0, 255)
    frames = np.empty((n, H, W), np.uint8)
    objs = []
    for _ in range(blobs):
        objs.append(dict(x=float(rng.uniform(-80, 100)), y=float(rng.uniform(120, 330)), w=int(rng.integers(70, 130)),
                         h=int(rng.integers(50, 90)), vx=float(rng.uniform(6, 11)), vy=float(rng.uniform(-1.5, 1.5)),
                         start=int(rng.integers(3, 15)), level=int(rng.integers(200, 250))))
    for t in range(n):
        f = scene + rng.normal(0, 2.0, (H, W))
        for o in objs:
            if t < o["start"]:
                continue
            x0, y0 = int(o["x"] + o["vx"] * (t - o["start"])), int(o["y"] + o["vy"] * (t - o["start"]))
            xa, xb, ya, yb = max(x0, 0), min(x0 + o["w"], W), max(y0, 0), min(y0 + o["h"], H)
            if xa < xb and ya < yb:
                f[ya:yb, xa:xb] = o["level"] + rng.normal(0, 3.0, (yb - ya, xb - xa))
        frames[t] = f.clip(0, 255).astype(np.uint8)
    return frames


def _ir_oracle_config():
    import track_oracle as to

    cfg = to.OracleConfig("lepton3")
    # the reference's IR TrackingConfig (config/trackingconfig.py:179-208)
    cfg.edge_pixels, cfg.frame_padding, cfg.min_dimension = 0, 10, 10
    # (areas_of_interest is zeroed for IR there, but the ATTRIBUTES the tracker reads keep the thermal placeholders 4.0 /
    # 2.0 unless a YAML is loaded -- pinned by tests/golden/config_golden.json; with filter_regions_pre_match off they
    # act after the matching: a faint matched region becomes a blank frame of its track, cliptracker.py:164-199)
    cfg.aoi_min_mass, cfg.aoi_pixel_variance, cfg.filter_regions_pre_match = 4.0, 2.0, False
    cfg.base_distance_change, cfg.min_mass_change, cfg.mass_change_percent = 12000, None, None
    cfg.max_distance, cfg.velocity_multiplier, cfg.base_velocity, cfg.fps = 30752, 8, 10, 10
    return cfg


def _oracle_ir_track(frames, scale=None):
    import cv2_shim
    import ir_oracle as iro
    import mog2_oracle as mo
    import track_oracle as to

    cfg = _ir_oracle_config()
    H, W = frames.shape[1:]
    crop = (0, 0, W, H)
    bg = mo.MOG2(W, H, history=1000)
    bg.apply(frames[0], 1)            # start_tracking: set_background(background_frame), learning rate 1
    state = dict(cfg=cfg, crop=crop, active=[], tracks=[], next_id=1, filtered=[])
    history = []
    for q, frame in enumerate(frames):
        mask = bg.apply(frame, -1)
        if scale:   # irtrackextractor.py:445-451: the foreground is detected at int(res * scale), INTER_AREA
            mask = cv2_shim.resize(mask, (int(W * scale), int(H * scale)), interpolation=cv2_shim.INTER_AREA)
        _, _, stats = iro.detect_objects_ir(mask, threshold=0)
        merged = iro.merge_components(list(stats[1:]), scale)
        delta = None
        prev_i = q - 1 if q < 10 else 10
        want = q - prev_i
        if prev_i != q and want != q and 0 <= want < q:
            delta = np.abs(frame - frames[want])          # uint8 arithmetic wraps, as in the reference
        cents = [[int(m[0] + m[2] / 2), int(m[1] + m[3] / 2)] for m in merged]
        if scale:   # Region.rescale(1 / scale) (region.py:44-50) before anything else looks at the region; the centroid stays
            f = 1 / scale
            merged = [[int(m[0] * f), int(m[1] * f), int(m[2] * f), int(m[3] * f), m[4] * f ** 2] for m in merged]
        st = np.array([[int(v) for v in m[:5]] for m in merged], np.int64).reshape(-1, 5)
        regions = to.regions_of_interest(st, cents, delta, q, cfg, crop)
        to.apply_matchings(state, regions)
        history.append(regions)
    for t in state["tracks"]:
        t.trim()
    return history, state["tracks"]


def _rt(r):
    return (int(r.x), int(r.y), int(r.width), int(r.height), int(r.mass), bool(r.blank))


@pytest.mark.parametrize("seed", [3, 11])
def test_ir_tracker_matches_oracle_chain(seed):
    from cpx.config import Config
    from cpx.track.clip import Clip
    from cpx.track.irtrackextractor import IRTrackExtractor

    frames = ir_video(seed)
    cfg = Config.get_defaults()
    ex = IRTrackExtractor(cfg.tracking, max_frames=frames.shape[0] + 4)
    clip = Clip(ex.config, "synthetic-ir.mp4", type="IR")
    clip.frames_per_second = 10
    assert ex.parse_frames(clip, frames)
    history, tracks = _oracle_ir_track(frames)
    assert len(clip.region_history) == len(history) == frames.shape[0]
    n_regions = 0
    for q, (got, want) in enumerate(zip(clip.region_history, history)):
        assert [_rt(r) for r in got] == [_rt(r) for r in want], q
        for a, b in zip(got, want):
            assert abs(float(a.pixel_variance) - float(b.pixel_variance)) <= 1e-9 * max(1.0, float(b.pixel_variance)), q
            assert list(a.centroid) == list(b.centroid)
        n_regions += len(got)
    assert n_regions >= 40
    got_tracks = sorted(clip.tracks, key=lambda t: t.get_id())
    want_tracks = sorted(tracks, key=lambda t: t.id)
    assert [t.get_id() for t in got_tracks] == [t.id for t in want_tracks] and len(got_tracks) >= 1
    for a, b in zip(got_tracks, want_tracks):
        assert a.start_frame == b.start_frame
        assert [_rt(r) for r in a.bounds_history] == [_rt(r) for r in b.bounds], a.get_id()
    assert clip.background is not None and clip.background.shape == (480, 640)
    ex.close()


@pytest.mark.parametrize("scale", [0.25, 0.5])
def test_ir_tracker_with_scale_matches_oracle_chain(scale):
    """IRTrackExtractor(scale=...) (the Pi's configuration, piclassifier.py:219-226): foreground down-scaled by
    INTER_AREA on the device (cpx_ir_resize_area), detection and the fragment merge at that size with the scaled
    thresholds, regions scaled back -- against the same steps in the oracle chain (cv2_shim.resize INTER_AREA)."""
    from cpx.config import Config
    from cpx.track.clip import Clip
    from cpx.track.irtrackextractor import IRTrackExtractor

    frames = ir_video(5, n=50)
    cfg = Config.get_defaults()
    ex = IRTrackExtractor(cfg.tracking, max_frames=frames.shape[0] + 4, scale=scale)
    clip = Clip(ex.config, "synthetic-ir.mp4", type="IR")
    clip.frames_per_second = 10
    assert ex.parse_frames(clip, frames)
    history, tracks = _oracle_ir_track(frames, scale=scale)
    n_regions = 0
    for q, (got, want) in enumerate(zip(clip.region_history, history)):
        assert [_rt(r) for r in got] == [_rt(r) for r in want], q
        for a, b in zip(got, want):
            assert list(a.centroid) == list(b.centroid)
            assert abs(float(a.pixel_variance) - float(b.pixel_variance)) <= 1e-9 * max(1.0, float(b.pixel_variance)), q
        n_regions += len(got)
    assert n_regions >= 30
    got_tracks = sorted(clip.tracks, key=lambda t: t.get_id())
    want_tracks = sorted(tracks, key=lambda t: t.id)
    assert [t.get_id() for t in got_tracks] == [t.id for t in want_tracks] and len(got_tracks) >= 1
    for a, b in zip(got_tracks, want_tracks):
        assert [_rt(r) for r in a.bounds_history] == [_rt(r) for r in b.bounds], a.get_id()
    with pytest.raises(NotImplementedError):
        IRTrackExtractor(cfg.tracking, scale=0.3)
    ex.close()


def test_extract_file_routes_raw_gray_recordings_to_the_ir_tracker(tmp_path):
    """extract_file on a recording that is not a .cptv (trackextractor.py:148-156): the IR tracker at 10 frames per
    second, metadata written next to the file -- here for the two raw-gray containers (cpx/track/grayvideo.py); same
    tracks as IRTrackExtractor.parse_frames on the frames themselves."""
    import json

    from test_grayvideo_cpu import write_y4m

    from cpx.config import Config
    from cpx.track.clip import Clip
    from cpx.track.irtrackextractor import IRTrackExtractor
    from cpx.track.trackextractor import extract_file

    frames = ir_video(3, n=40)
    cfg = Config.get_defaults()
    ex = IRTrackExtractor(cfg.tracking, max_frames=frames.shape[0] + 4)
    ref = Clip(ex.config, "direct.npy", type="IR")
    ref.frames_per_second = 10
    assert ex.parse_frames(ref, frames)
    want = [[(r.x, r.y, r.width, r.height, r.mass) for r in t.bounds_history] for t in sorted(ref.tracks, key=lambda t: t.get_id())]
    ex.close()
    assert want and sum(len(w) for w in want) > 20
    np.save(tmp_path / "a.npy", frames)
    write_y4m(tmp_path / "b.y4m", frames, "420jpeg")
    for name in ("a.npy", "b.y4m"):
        clip, extractor, meta = extract_file(tmp_path / name, cfg, False)
        assert clip.frames_per_second == 10 and clip.type == "IR"
        got = [[(r.x, r.y, r.width, r.height, r.mass) for r in t.bounds_history] for t in sorted(clip.tracks, key=lambda t: t.get_id())]
        assert got == want, name
        with open((tmp_path / name).with_suffix(".txt")) as fh:
            saved = json.load(fh)
        assert len(saved["tracks"]) == len(want) and saved["algorithm"]["tracker_version"].startswith("IR-")
        assert [len(t["positions"]) for t in saved["tracks"]] == [len(w) for w in want]
    with pytest.raises((NotImplementedError, ValueError)):
        (tmp_path / "c.mp4").write_bytes(b"not a video")
        extract_file(tmp_path / "c.mp4", cfg, False)


def test_ir_resize_area_matches_the_shim():
    import cv2_shim
    import torch

    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3")
    rng = np.random.default_rng(4)
    for f in (2, 4, 5, 8):
        img = rng.integers(0, 256, (3, 480, 640), dtype=np.uint8)
        img[1] = (rng.random((480, 640)) < 0.3).astype(np.uint8) * 255      # a foreground mask
        got = eng.ir_resize_area(torch.from_numpy(img).to(eng.device), f)
        eng.synchronize()
        got = got.cpu().numpy()
        for k in range(3):
            want = cv2_shim.resize(img[k], (640 // f, 480 // f), interpolation=cv2_shim.INTER_AREA)
            assert np.array_equal(got[k], want), (f, k)
    import ctypes as C

    src = torch.zeros((480, 640), dtype=torch.uint8, device=eng.device)
    dst = torch.zeros((480, 640), dtype=torch.uint8, device=eng.device)
    rc = eng.lib.cpx_ir_resize_area(eng.h, C.c_void_p(src.data_ptr()), 1, 640, 480, 7, C.c_void_p(dst.data_ptr()))
    assert rc != 0      # 7 divides neither side: refused, not approximated
    eng.close()


def test_mixed_lepton35_and_ir_batch():
    """BASELINE configs[4]: lepton3.5 160 x 120 clips and 640 x 480 IR videos handled side by side on one GPU -- the
    thermal clips as one device batch, the IR videos through the IR tracker -- each result equal to the same clip run
    alone."""
    from cpx import synth
    from cpx.config import Config
    from cpx.engine import TrackEngine
    from cpx.track.clip import Clip
    from cpx.track.irtrackextractor import IRTrackExtractor

    rng = np.random.default_rng(8)
    T = 50
    thermal = [synth.make_clip(rng, T, model="lepton3.5", max_blobs=3) for _ in range(4)]
    ir = [ir_video(21, n=30, blobs=1), ir_video(22, n=30, blobs=2)]
    cfg = Config.get_defaults()
    eng = TrackEngine(model="lepton3.5", max_frames=T)
    offs = (np.arange(5) * T).astype(np.int32)
    meta = np.concatenate([eng.make_meta(T) for _ in range(4)])
    dev = eng.upload_frames(np.concatenate(thermal))
    # interleave: thermal batch enqueued, IR videos tracked meanwhile on their own handles, then the batch is read
    res = eng.track_batch(dev, offs, meta, want_labels=True, want_filtered=True)
    ir_clips = []
    for k, frames in enumerate(ir):
        ex = IRTrackExtractor(cfg.tracking, max_frames=40)
        clip = Clip(ex.config, "ir-%d.mp4" % k, type="IR")
        clip.frames_per_second = 10
        ex.parse_frames(clip, frames)
        ir_clips.append((clip, ex))
    res.check()
    labels, filt = res.labels(), res.filtered()
    for b in range(4):   # every thermal clip of the mixed run == the clip alone
        solo = eng.track_batch(eng.upload_frames(thermal[b]), np.array([0, T], np.int32), eng.make_meta(T),
                               want_labels=True, want_filtered=True)
        assert np.array_equal(solo.labels(), labels[b * T:(b + 1) * T])
        assert np.array_equal(solo.filtered(), filt[b * T:(b + 1) * T])
    for (clip, ex), frames in zip(ir_clips, ir):   # every IR video == the oracle chain
        history, tracks = _oracle_ir_track(frames)
        assert [[_rt(r) for r in g] for g in clip.region_history] == [[_rt(r) for r in w] for w in history]
        assert sorted(len(t.bounds_history) for t in clip.tracks) == sorted(len(t.bounds) for t in tracks)
        ex.close()
    eng.close()


def _ir_batch(videos, **kw):
    from cpx.config import Config
    from cpx.track.clip import Clip
    from cpx.track.irtrackextractor import IRTrackExtractor

    cfg = Config.get_defaults()
    ex = IRTrackExtractor(cfg.tracking, **kw)
    clips = []
    for k in range(len(videos)):
        clip = Clip(ex.config, "ir-%d.mp4" % k, type="IR")
        clip.frames_per_second = 10
        clips.append(clip)
    assert ex.parse_frames_batch(clips, videos)
    return ex, clips


def test_ir_device_batch_matches_oracle_and_one_by_one():
    """IRTrackExtractor.parse_frames_batch: V videos of different lengths in lockstep on the device (MOG2, detection,
    cpx_ir_merge, one cpx_associate_batch) -- every clip's regions and tracks equal the oracle chain's and the
    one-video-at-a-time path's, statistics included."""
    from cpx.config import Config
    from cpx.track.clip import Clip
    from cpx.track.irtrackextractor import IRTrackExtractor

    videos = [ir_video(3), ir_video(11, n=45), ir_video(21, n=30, blobs=1), ir_video(5, n=52, blobs=3)]
    ex, clips = _ir_batch(videos)
    cfg = Config.get_defaults()
    for frames, clip in zip(videos, clips):
        history, tracks = _oracle_ir_track(frames)
        assert len(clip.region_history) == len(history) == frames.shape[0] == clip.current_frame + 1
        for q, (got, want) in enumerate(zip(clip.region_history, history)):
            assert [_rt(r) for r in got] == [_rt(r) for r in want], q
            for a, b in zip(got, want):
                # the device record carries the variance as float32
                assert abs(float(a.pixel_variance) - float(b.pixel_variance)) <= 1e-6 * max(1.0, float(b.pixel_variance)), q
                assert list(a.centroid) == list(b.centroid)
        got_tracks = sorted(clip.tracks, key=lambda t: t.get_id())
        want_tracks = sorted(tracks, key=lambda t: t.id)
        assert [t.get_id() for t in got_tracks] == [t.id for t in want_tracks] and len(got_tracks) >= 1
        for a, b in zip(got_tracks, want_tracks):
            assert a.start_frame == b.start_frame
            assert [_rt(r) for r in a.bounds_history] == [_rt(r) for r in b.bounds], a.get_id()
        # against the per-video path: trap state and clip statistics too
        solo_ex = IRTrackExtractor(cfg.tracking, max_frames=frames.shape[0] + 4)
        solo = Clip(solo_ex.config, "solo.mp4", type="IR")
        solo.frames_per_second = 10
        solo_ex.parse_frames(solo, frames)
        st = sorted(solo.tracks, key=lambda t: t.get_id())
        assert [(t.get_id(), t.in_trap, t.trigger_frame, t.direction) for t in got_tracks] == \
               [(t.get_id(), t.in_trap, t.trigger_frame, t.direction) for t in st]
        assert [[r.in_trap for r in t.bounds_history] for t in got_tracks] == [[r.in_trap for r in t.bounds_history] for t in st]
        assert np.array_equal(clip.background, solo.background)
        for k in ("frame_stats_min", "frame_stats_max", "frame_stats_median", "frame_stats_mean"):
            assert [float(v) for v in getattr(clip.stats, k)] == [float(v) for v in getattr(solo.stats, k)], k
        assert float(clip.stats.filtered_sum) == float(solo.stats.filtered_sum)
        assert clip.stats.mean_temp == solo.stats.mean_temp
        solo_ex.close()
    ex.close()


def test_mixed_lepton35_and_ir_device_batches():
    """BASELINE configs[4] with BOTH cameras as device batches: four lepton3.5 clips through cpx_track_batch and two IR
    videos through parse_frames_batch on the same GPU, each result equal to its own single-camera run."""
    from cpx import synth
    from cpx.engine import TrackEngine

    rng = np.random.default_rng(8)
    T = 50
    thermal = [synth.make_clip(rng, T, model="lepton3.5", max_blobs=3) for _ in range(4)]
    ir = [ir_video(21, n=30, blobs=1), ir_video(22, n=30, blobs=2)]
    eng = TrackEngine(model="lepton3.5", max_frames=T)
    offs = (np.arange(5) * T).astype(np.int32)
    meta = np.concatenate([eng.make_meta(T) for _ in range(4)])
    dev = eng.upload_frames(np.concatenate(thermal))
    res = eng.track_batch(dev, offs, meta, want_labels=True, want_filtered=True)   # enqueued; the IR batch runs meanwhile
    ex, clips = _ir_batch(ir)
    res.check()
    labels = res.labels()
    for b in range(4):
        solo = eng.track_batch(eng.upload_frames(thermal[b]), np.array([0, T], np.int32), eng.make_meta(T), want_labels=True)
        assert np.array_equal(solo.labels(), labels[b * T:(b + 1) * T])
    for clip, frames in zip(clips, ir):
        history, tracks = _oracle_ir_track(frames)
        assert [[_rt(r) for r in g] for g in clip.region_history] == [[_rt(r) for r in w] for w in history]
        assert sorted(len(t.bounds_history) for t in clip.tracks) == sorted(len(t.bounds) for t in tracks)
    ex.close()
    eng.close()
