"""No recording is lost to a capacity.  The reference has no limit on components per frame
(track/cliptrackextractor.py:236-247), on simultaneous tracks or on tracks per clip (track/cliptracker.py:202-247); the
kernels work on caller-sized tables and report CPX_ERR_OVERFLOW, and the host layers that own the sizes grow them and
run THAT clip again: TrackEngine.track_clip_grown, ClipTrackExtractor (parse_clip / parse_clips / process_frame),
BatchPipeline, the bulk directory drivers.  Everything here must equal the oracle instead of raising."""
import json
import os

import numpy as np
import pytest

from helpers import encode_cptv
from test_track_gpu import _compare_assoc, _compare_with_oracle, _oracle_clip

pytestmark = pytest.mark.gpu

H, W = 120, 160


def hot_pixel_clip(n=6, step=7):
    """A clip whose frames 2.. hold a grid of single hot pixels far above threshold: hundreds of components per frame
    (352 at step 7 -- beyond the 64 of the default tables AND the 256 of the frame kernel's LDS tables; 130 at step 12.
    At step 6 the blurred pixels merge into 26 stripes: the grid of tests/test_track_gpu.py::test_degenerate_frames)."""
    cb = np.full((n, H, W), 3000, np.uint16)
    cb[2:, 4:116:step, 4:156:step] = 9000
    return cb


def crowded_clip(T=100, n_obj=24, life=3, period=10, seed=3, movers=5):
    """`n_obj` small warm objects at once, each alive `life` of every `period` frames (its track dies in the gap) and
    re-appearing somewhere else -- more than 16 simultaneous tracks and more than 128 in all -- plus `movers` objects that
    cross the frame steadily (tracks that survive the end-of-clip filter)."""
    rng = np.random.default_rng(seed)
    frames = np.full((T, H, W), 2900.0, np.float32)
    frames += rng.normal(0.0, 2.0, size=frames.shape).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    cols, rows = 6, 4
    cw, ch = W // cols, H // rows
    for k in range(n_obj):
        cx0, cy0 = (k % cols) * cw, (k // cols) * ch
        for t in range(T):
            g, ph = divmod(t, period)
            if ph >= life:
                continue
            r2 = np.random.default_rng(seed * 1000 + k * 97 + g)
            px = cx0 + 6 + r2.uniform(0, cw - 12) + 0.5 * ph
            py = cy0 + 6 + r2.uniform(0, ch - 12)
            y0, y1, x0, x1 = int(max(0, py - 8)), int(min(H, py + 9)), int(max(0, px - 8)), int(min(W, px + 9))
            d2 = (yy[y0:y1, x0:x1] - py) ** 2 + (xx[y0:y1, x0:x1] - px) ** 2
            frames[t, y0:y1, x0:x1] += 300.0 * np.exp(-d2 / (2 * 2.0 * 2.0))
    for m in range(movers):
        py = 12.0 + m * (H - 24.0) / max(movers - 1, 1)
        for t in range(T):
            px = 5.0 + 1.5 * t + 7.0 * m
            if px > W - 5:
                break
            y0, y1, x0, x1 = int(max(0, py - 12)), int(min(H, py + 13)), int(max(0, px - 12)), int(min(W, px + 13))
            d2 = (yy[y0:y1, x0:x1] - py) ** 2 + (xx[y0:y1, x0:x1] - px) ** 2
            frames[t, y0:y1, x0:x1] += 250.0 * np.exp(-d2 / (2 * 3.0 * 3.0))
    return np.clip(np.rint(frames), 0, 65535).astype(np.uint16)


@pytest.fixture(scope="module")
def eng():
    from cpx.engine import TrackEngine

    e = TrackEngine(model="lepton3")
    yield e
    e.close()


@pytest.mark.parametrize("step", [7, 12])
def test_hot_pixel_grid_equals_oracle(eng, step):
    """352 components in a frame (step 7: beyond the frame kernel's LDS tables, the HBM tables take it) and 130 (step 12:
    beyond the default 64, inside the LDS tables): the first pass reports the overflow and the count, the grown one
    reproduces the oracle's label images, statistics, centroids and variances."""
    cb = hot_pixel_clip(step=step)
    n = cb.shape[0]
    offs = np.array([0, n], np.int32)
    meta = eng.make_meta(n)
    dev = eng.upload_frames(cb)
    first = eng.track_batch(dev, offs, meta, want_labels=True)
    out, _ = _oracle_clip(cb, "lepton3")
    want = max(f["n_components"] for f in out["frames"])
    assert want > eng.cap and (want > 256) == (step == 7)
    assert first.overflowed(offs) == {0: want}
    g, res, assoc, _ = eng.track_clip_grown(dev, meta, need_components=want, want_labels=True, want_filtered=True,
                                            associate=False)
    assert g is not eng and g.cap >= want and not res.overflowed(offs)
    _compare_with_oracle(res, 0, out, n)
    # without a hint the growth finds the size itself (one more pass)
    g2, res2, _, _ = eng.track_clip_grown(dev, meta, want_labels=True, want_filtered=True, associate=False)
    assert np.array_equal(res2.info, res.info)


def test_crowded_scene_equals_oracle(eng):
    """24 objects at once, ~240 tracks in all (16 / 128 are the default tables): the association grown until it fits equals
    the oracle track by track."""
    import track_oracle as to

    clip = crowded_clip()
    T = clip.shape[0]
    offs = np.array([0, T], np.int32)
    meta = eng.make_meta(T)
    dev = eng.upload_frames(clip)
    res = eng.track_batch(dev, offs, meta)
    assoc = eng.associate_batch(res, offs, meta)
    assert assoc.overflowed() == [0] and not res.overflowed(offs)
    g, res2, assoc2, params = eng.track_clip_grown(dev, meta)
    assert params.max_active_tracks > 16 and params.max_tracks > 128 and not assoc2.overflowed()
    out = to.track_clip(clip, cfg=to.OracleConfig("lepton3"), keep=True, apply_filter=False)
    n_tracks, _ = _compare_assoc(assoc2, 0, out, 0, list(range(T)))
    assert n_tracks > 128
    every = out["tracks"] + [t for _, t in out.get("filtered_tracks", [])]
    live = max(sum(1 for t in every if t.start_frame <= q < t.start_frame + len(t.bounds)) for q in range(T))
    assert live > 16


def _write(path, frames):
    n = frames.shape[0]
    encode_cptv(path, frames, [16] * n, time_on=[100000 + 111 * i for i in range(n)], last_ffc=[40000] * n)


def _strip(meta):
    from cpx.ml_tools.tools import CustomJSONEncoder

    m = json.loads(json.dumps(meta, cls=CustomJSONEncoder))
    for k in ("tracking_time", "source", "id"):
        m.pop(k, None)
    return m


def test_extractors_grow_instead_of_raising(tmp_path):
    """extract_file on the hot-pixel and the crowded recording: metadata instead of CpxError; the frame-by-frame path
    (process_frame: a stream that outgrows its tables is replayed on larger ones) builds the same tracks; a directory run
    with the two among ordinary recordings writes every metadata file."""
    import track_oracle as to
    from cpx import synth
    from cpx.config import Config
    from cpx.cptv import CptvReader
    from cpx.track.clip import Clip
    from cpx.track.cliptrackextractor import ClipTrackExtractor
    from cpx.track.trackextractor import extract_file, extract_files

    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = False
    hot, crowd = tmp_path / "hot.cptv", tmp_path / "crowd.cptv"
    _write(hot, hot_pixel_clip(n=12))
    crowd_frames = crowded_clip(T=80)
    _write(crowd, crowd_frames)
    clip_h, _, meta_h = extract_file(hot, cfg, False, save_meta=False)
    assert clip_h.current_frame == 11
    masks = [clip_h.frame_buffer.get_frame(q).mask for q in range(2, 12)]
    assert all(int(m.max()) == 352 for m in masks)
    clip_c, _, meta_c = extract_file(crowd, cfg, False, save_meta=False)
    out = to.track_clip(crowd_frames, cfg=to.OracleConfig("lepton3"), keep=True, apply_filter=True)
    assert len(meta_c["tracks"]) == len(out["tracks"]) and len(meta_c["tracks"]) >= 1
    # frame by frame: the same tracks as the whole-clip path
    ex = ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False)
    clip_s = Clip(ex.config, str(crowd))
    clip_s.frames_per_second = 9
    reader = CptvReader(str(crowd))
    header = reader.get_header()
    clip_s.set_res(header.x_resolution, header.y_resolution)
    clip_s.set_model(header.model if header.model else None)
    frames = reader.read_all()
    clip_s.update_background(frames[0].pix)
    clip_s._background_calculated()
    for fr in frames:
        ex.process_frame(clip_s, fr)
    ex_w = ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False)
    whole = Clip(ex_w.config, str(crowd))
    whole.frames_per_second = 9
    ex_w.do_tracking = True
    ex_w.parse_clip(whole)
    sig = lambda tr: (tr.get_id(), tr.start_frame, [(r.x, r.y, r.width, r.height, int(r.mass)) for r in tr.bounds_history])
    streamed = sorted(sig(t) for t in clip_s.tracks)
    assert len(streamed) > 128
    # every track of the whole-clip path (trimmed there) is a streamed track's history with the blanks cut off its ends
    by_id = {s[0]: s for s in streamed}
    for t in whole.tracks:
        full = by_id[t.get_id()][2]
        got = [(r.x, r.y, r.width, r.height, int(r.mass)) for r in t.bounds_history]
        k = t.start_frame - by_id[t.get_id()][1]
        assert full[k:k + len(got)] == got
    ex.close()
    # a directory with the two among ordinary recordings: one metadata file per recording
    rng = np.random.default_rng(4)
    paths = [hot, crowd]
    for k in range(6):
        p = tmp_path / ("plain%d.cptv" % k)
        _write(p, synth.make_clip(rng, 40, max_blobs=2))
        paths.append(p)
    batch = extract_files(paths, cfg, False, save_meta=False)
    assert len(batch) == len(paths)
    by_name = {os.path.basename(str(c.source_file)): m for c, _, m in batch}
    assert _strip(by_name["hot.cptv"]) == _strip(meta_h)
    assert _strip(by_name["crowd.cptv"]) == _strip(meta_c)
    # the bulk directory driver (TrackExtractor.extract -> cpx.track.bulk): device decode + one tracking group; the two
    # recordings whose tables overflow there are taken out of the group and tracked alone -- eight metadata files
    from cpx.track.trackextractor import TrackExtractor

    TrackExtractor(cfg).extract(tmp_path)
    written = sorted(q.name for q in tmp_path.glob("*.txt"))
    assert written == sorted(os.path.splitext(os.path.basename(str(q)))[0] + ".txt" for q in paths)
    for name, want in (("hot.txt", meta_h), ("crowd.txt", meta_c)):
        with open(tmp_path / name) as fh:
            got = json.load(fh)
        assert _strip(got)["tracks"] == _strip(want)["tracks"], name


def test_batch_pipeline_regrows_the_clip_that_overflowed(eng):
    """BatchPipeline over four clips of which one is crowded: the crowded clip reports no tracks in the batch pass, is
    run again alone on larger tables, and its tracks join the batch's -- equal to running it alone from the start."""
    import torch
    from cpx import synth
    from cpx.ml_tools import wrresnet as wr
    from cpx.pipeline import BatchPipeline
    from cpx.tracking import make_filter_params, make_track_params

    rng = np.random.default_rng(12)
    T = 60
    clips = [synth.make_clip(rng, T, max_blobs=3), crowded_clip(T=T), synth.make_clip(rng, T, max_blobs=3),
             synth.make_clip(rng, T, max_blobs=2)]
    frames = eng.upload_frames(np.concatenate(clips))
    offs = (np.arange(5) * T).astype(np.int32)
    meta = eng.make_meta(4 * T)
    x = torch.rand((8, 160, 160, 2), device=eng.device) * 255
    w = wr.calibrate_bn_device(eng, wr.random_weights(17, seed=2), x)
    net = wr.WRResNetDevice(eng, w, 17)
    pipe = BatchPipeline(eng, net, n_labels=17, fp_index=4, cnn_chunk=64)
    res = pipe.run(frames, offs, meta)
    assert res.overflowed == [1] and [r[0] for r in res.regrown] == [1]
    tc = res.track_clip.cpu().numpy()
    assert res.n_tracks == tc.shape[0] == res.scores.shape[0] and (tc[:, 0] == 1).sum() >= 1
    # the crowded clip alone, on tables that fit from the start
    tp = make_track_params(max_active_tracks=res.regrown[0][2], max_tracks=res.regrown[0][3])
    fp = make_filter_params(max_active_tracks=tp.max_active_tracks, max_tracks_per_clip=tp.max_tracks)
    alone = BatchPipeline(eng, net, n_labels=17, fp_index=4, cnn_chunk=64, track_params=tp, filter_params=fp)
    r1 = alone.run(frames[T:2 * T], np.array([0, T], np.int32), meta[T:2 * T])
    assert not r1.overflowed
    mine = tc[:, 0] == 1
    assert np.array_equal(tc[mine][:, 1], r1.track_clip.cpu().numpy()[:, 1])
    assert torch.equal(res.scores[torch.from_numpy(mine).to(res.scores.device)], r1.scores)
    # the other clips are what the batch pass gave them
    plain = BatchPipeline(eng, net, n_labels=17, fp_index=4, cnn_chunk=64)
    keep = [0, 2, 3]
    fr3 = torch.cat([frames[k * T:(k + 1) * T] for k in keep])
    r3 = plain.run(fr3, (np.arange(4) * T).astype(np.int32), meta[: 3 * T])
    others = res.scores[torch.from_numpy(~mine).to(res.scores.device)]
    assert torch.equal(others, r3.scores)
    # the overlapped form (sub_batches > 1, the network on a second engine = a second stream) regrows too: the crowded
    # clip sits in the first of two groups; same tracks, same scores as the one-stream run
    from cpx.engine import TrackEngine

    ceng = TrackEngine(model="lepton3", device=0, max_frames=45)
    net2 = wr.WRResNetDevice(ceng, w, 17)
    pipe2 = BatchPipeline(eng, net2, n_labels=17, fp_index=4, cnn_chunk=64)
    res2 = pipe2.run(frames, offs, meta, sub_batches=2)
    assert res2.overflowed == [1] and [r[0] for r in res2.regrown] == [1]
    tc2 = res2.track_clip.cpu().numpy()
    order, order2 = np.lexsort((tc[:, 1], tc[:, 0])), np.lexsort((tc2[:, 1], tc2[:, 0]))
    assert np.array_equal(tc[order], tc2[order2])
    assert torch.equal(res.scores[torch.from_numpy(order).to(res.scores.device)],
                       res2.scores[torch.from_numpy(order2).to(res2.scores.device)])
    net2.close()
    ceng.close()
    net.close()
