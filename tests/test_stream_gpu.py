"""Incremental tracking (ClipTrackExtractor.process_frame / start_tracking over cpx_track_frame /
cpx_associate_frame) must give, frame by frame, exactly what the whole-clip path (parse_clip) gives."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _config(denoise):
    from cpx.config import Config

    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = bool(denoise)
    return cfg


def _whole(path, cfg):
    from cpx.track.clip import Clip
    from cpx.track.cliptrackextractor import ClipTrackExtractor

    ex = ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False)
    clip = Clip(ex.config, path)
    clip.frames_per_second = 9
    ex.parse_clip(clip)
    return clip, ex


def _region_tuple(r):
    c = r.centroid
    return (r.x, r.y, r.width, r.height, int(r.mass), r.frame_number, bool(r.blank), bool(r.was_cropped),
            bool(r.is_along_border), float(r.pixel_variance), type(c[0]).__name__, float(c[0]), float(c[1]))


def _stream(path, cfg, untracked_prefix=0):
    from cpx.cptv import CptvReader
    from cpx.track.clip import Clip
    from cpx.track.cliptrackextractor import ClipTrackExtractor

    ex = ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False)
    clip = Clip(ex.config, path)
    clip.frames_per_second = 9
    reader = CptvReader(path)
    header = reader.get_header()
    clip.set_res(header.x_resolution, header.y_resolution)
    clip.set_model(header.model if header.model else None)
    frames = reader.read_all()
    clip.update_background(frames[0].pix)
    clip._background_calculated()
    todo = [f for f in frames if not f.background_frame]
    created = []
    if untracked_prefix:
        assert ex.start_tracking(clip, todo[:untracked_prefix], track_frames=False) == []
        todo = todo[untracked_prefix:]
    for fr in todo:
        created.extend(ex.process_frame(clip, fr))
    return clip, ex, created


@pytest.mark.parametrize("name,dn", [("possum", 0), ("hedgehog", 0), ("possum", 1)])
def test_frame_by_frame_equals_whole_clip(golden_dir, name, dn):
    path = os.path.join(golden_dir, name + ".cptv")
    cfg = _config(dn)
    whole, ex_w = _whole(path, cfg)
    clip, ex, created = _stream(path, cfg)
    assert clip.current_frame == whole.current_frame
    assert clip.ffc_frames == whole.ffc_frames
    # frames and statistics
    for q in range(whole.current_frame + 1):
        a, b = clip.frame_buffer.get_frame(q), whole.frame_buffer.get_frame(q)
        assert np.array_equal(a.thermal, b.thermal) and np.array_equal(a.filtered, b.filtered)
        assert np.array_equal(a.mask, b.mask)
    for k in ("frame_stats_min", "frame_stats_max", "frame_stats_median", "frame_stats_mean"):
        assert getattr(clip.stats, k) == getattr(whole.stats, k)
    assert clip.stats.filtered_sum == whole.stats.filtered_sum
    assert np.array_equal(ex.background_alg.background, ex_w.background_alg.background)
    assert ex.background_alg.get_average() == ex_w.background_alg.get_average()
    # regions of interest per frame
    assert len(clip.region_history) == len(whole.region_history)
    for ra, rb in zip(clip.region_history, whole.region_history):
        assert [_region_tuple(r) for r in ra] == [_region_tuple(r) for r in rb]
    # tracks: every track ever created, then the end-of-clip filtering
    assert [t.get_id() for t in created] == sorted(t.get_id() for t in created)
    ex.apply_track_filtering(clip)
    assert [t.get_id() for t in clip.tracks] == [t.get_id() for t in whole.tracks]
    assert len(whole.tracks) > 0
    for ta, tb in zip(clip.tracks, whole.tracks):
        assert (ta.start_frame, ta.end_frame) == (tb.start_frame, tb.end_frame)
        assert [_region_tuple(r) for r in ta.bounds_history] == [_region_tuple(r) for r in tb.bounds_history]
        assert ta.stats.score == tb.stats.score and ta.stats.frames_moved == tb.stats.frames_moved
        assert list(ta.vel_x) == list(tb.vel_x) and list(ta.vel_y) == list(tb.vel_y)
    assert [(r, t.get_id()) for r, t in clip.filtered_tracks] == [(r, t.get_id()) for r, t in whole.filtered_tracks]
    assert {t.get_id() for t in clip.active_tracks} == {t.get_id() for t in whole.active_tracks}


def test_untracked_preview_frames(golden_dir):
    """start_tracking(track_frames=False): the frames update the background and the frame buffer but never reach the
    association -- pixel results are unchanged, no track starts inside the preview."""
    path = os.path.join(golden_dir, "possum.cptv")
    cfg = _config(0)
    whole, _ = _whole(path, cfg)
    k = 45
    clip, ex, created = _stream(path, cfg, untracked_prefix=k)
    assert clip.current_frame == whole.current_frame
    for q in (0, k - 1, k, whole.current_frame):
        a, b = clip.frame_buffer.get_frame(q), whole.frame_buffer.get_frame(q)
        assert np.array_equal(a.filtered, b.filtered) and np.array_equal(a.mask, b.mask)
    assert len(clip.region_history) == whole.current_frame + 1 - k
    assert created and all(t.start_frame >= k for t in created)
    # from frame k on, the regions of interest are those of the whole-clip run
    for ra, rb in zip(clip.region_history, whole.region_history[k:]):
        assert [_region_tuple(r) for r in ra] == [_region_tuple(r) for r in rb]


def test_stream_sequence_errors():
    from cpx._lib import CpxError
    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3", max_frames=64)
    st = eng.open_stream(8)
    rng = np.random.default_rng(0)
    fr = rng.integers(2900, 3000, (120, 160)).astype(np.uint16)
    st.append(fr, init_only=True)
    for _ in range(7):
        st.append(fr)
    with pytest.raises(CpxError):
        st.append(fr)  # capacity
    # a batch call ends the stream: the next frame is refused instead of running on foreign state
    st2 = eng.open_stream(8)
    st2.append(fr, init_only=True)
    st2.append(fr)
    dev = eng.upload_frames(np.stack([fr, fr]))
    eng.track_batch(dev, np.array([0, 2], np.int32), eng.make_meta(2)).check()
    with pytest.raises(CpxError):
        st2.append(fr)
    eng.close()
