"""The batched pipeline (what bench.py times) against the oracle chain, clip by clip: kept tracks and
their order, the segment plan (frames, padding order) of the reference's get_segments under identity draws, network inputs (bit-exact vs the NumPy oracle for the
planned segments), and the per-track class scores (1e-3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LABELS = ["bird", "cat", "deer", "dog", "false-positive", "hedgehog", "human", "kiwi", "leporidae", "mustelid",
          "penguin", "possum", "rodent", "sheep", "vehicle", "wallaby", "land-bird"]


def test_batched_pipeline_matches_oracle_chain():
    import torch

    import classify_oracle as co
    import cnn_oracle as cnn
    import track_oracle as to
    from cpx import synth
    from cpx._lib import CROP_REQ_DTYPE
    from cpx.engine import TrackEngine
    from cpx.ml_tools import wrresnet as wr
    from cpx.pipeline import BatchPipeline
    from cpx.ml_tools import datasetstructures as ds
    from helpers import IdentityDraws, load_clip

    eng = TrackEngine(model="lepton3")
    rng = np.random.default_rng(31)
    T = 130
    clips = [synth.make_clip(rng, T, max_blobs=3) for _ in range(5)]
    # + a real clip with a leading background frame and rejected tracks
    pos, t_on, ffc, bgf, hdr = load_clip("possum")
    clips.append(pos)
    lens = [c.shape[0] for c in clips]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    metas = [eng.make_meta(T) for _ in range(5)] + [eng.make_meta(pos.shape[0], t_on, ffc, bgf)]
    meta = np.concatenate(metas)
    w = wr.random_weights(len(LABELS), seed=2)
    w = cnn.calibrate_bn(w, rng.uniform(0, 255, size=(2, 160, 160, 2)).astype(np.float32))
    net = wr.WRResNetDevice(eng, w, len(LABELS))
    pipe = BatchPipeline(eng, net, n_labels=len(LABELS), fp_index=LABELS.index("false-positive"))
    frames_dev = eng.upload_frames(np.concatenate(clips))
    res = pipe.run(frames_dev, offs, meta, keep_samples=True)
    res.track.check()
    res.assoc.check()
    tc = res.track_clip.cpu().numpy()
    scores = res.scores.cpu().numpy()
    x_all = res.samples_dev.cpu().numpy()
    reqs = res.reqs_dev.cpu().numpy().view(CROP_REQ_DTYPE).reshape(-1, 25)
    st = res.sample_track_dev.cpu().numpy()
    ti = 0
    total_tracks = n_multi = 0
    for b, clip in enumerate(clips):
        is_pos = b == 5
        out = to.track_clip(clip, t_on if is_pos else None, ffc if is_pos else None, bgf if is_pos else None,
                            to.OracleConfig("lepton3"), keep=True)
        proc = [i for i in range(clip.shape[0]) if not (is_pos and bgf[i])]
        fr = out["frames"]
        assert res.counts[b, 0] == len(out["tracks"]), b
        for t in out["tracks"]:
            assert (tc[ti, 0], tc[ti, 1]) == (b, t.id)
            mine = np.nonzero(st == ti)[0]
            # device frame index -> processed frame number of this clip
            segs = [np.array([proc.index(int(f) - int(offs[b])) for f in reqs[s]["frame"]]) for s in mine]
            # the planner's choice = the host port of the reference's get_segments (pinned by segments_golden.json and,
            # under these draws, by segments_identity_golden.json) with every random draw the identity
            with IdentityDraws():
                want_segs, _ = ds.get_segments(b, t.id, t.bounds[0].frame_number, regions=np.array(t.bounds, dtype=object),
                                               segment_width=25, segment_frame_spacing=9, ffc_frames=out["ffc_frames"],
                                               repeats=1, min_frames=0, segment_types=[ds.SegmentType.ALL_RANDOM_MASKED],
                                               max_segments=None, dont_filter=False, min_segments=1, seed=None)
            assert [list(map(int, sg)) for sg in segs] == [[int(f) for f in sg.frame_indices] for sg in want_segs], (b, t.id)
            n_multi += len(segs) > 1
            by_frame = {r.frame_number: r for r in t.bounds}
            x, _ = co.preprocess_segments(lambda q: clip[proc[q]], lambda q: fr[q]["filtered"].astype(np.float64),
                                          by_frame, t.bounds, segs, 32, (1, 1, 158, 118))
            assert np.array_equal(x_all[mine], x), (b, t.id)
            _, probs = cnn.forward(w, x)
            want = co.classified_track(probs, prediction_frames=segs, labels=LABELS)
            assert np.abs(scores[ti] - want).max() <= 1e-3, (b, t.id)
            ti += 1
            total_tracks += 1
    assert ti == res.n_tracks and total_tracks >= 3 and n_multi >= 1
    eng.close()


def test_overlapped_sub_batches_equal_single_pass():
    """BatchPipeline(sub_batches=k): the track stage of one group of clips overlaps the network of the previous
    group on a second stream -- the results must be those of the plain pass, bit for bit."""
    import torch

    import cnn_oracle as cnn
    from cpx import synth
    from cpx.engine import TrackEngine
    from cpx.ml_tools import wrresnet as wr
    from cpx.pipeline import BatchPipeline

    K, T = 11, 130
    frames, offs = synth.make_batch(K, T, seed=5)
    eng = TrackEngine(model="lepton3", max_frames=T)
    ceng = TrackEngine(model="lepton3", max_frames=45)
    rng = np.random.default_rng(3)
    w = wr.random_weights(17, seed=2)
    w = cnn.calibrate_bn(w, rng.uniform(0, 255, size=(2, 160, 160, 2)).astype(np.float32))
    meta = np.concatenate([eng.make_meta(T) for _ in range(K)])
    dev = eng.upload_frames(frames)
    plain = BatchPipeline(eng, wr.WRResNetDevice(eng, w, 17), n_labels=17, fp_index=4, cnn_chunk=7).run(dev, offs, meta)
    assert plain.n_tracks >= 4
    pipe = BatchPipeline(eng, wr.WRResNetDevice(ceng, w, 17), n_labels=17, fp_index=4, cnn_chunk=7)
    # n_sub = 1 on the two-engine pipeline: the plain (non-overlapped) pass with the network on another stream than the
    # crop / aggregation kernels -- ordered by events (cnn_chunk = 7: the shared sample buffer is reused per chunk)
    for n_sub in (1, 2, 3, 11):
        got = pipe.run(dev, offs, meta, sub_batches=n_sub)
        got.track.check()
        got.assoc.check()
        assert (got.n_tracks, got.n_samples) == (plain.n_tracks, plain.n_samples)
        assert np.array_equal(got.counts, plain.counts)
        assert torch.equal(got.track_clip, plain.track_clip)
        assert torch.equal(got.scores, plain.scores) and torch.equal(got.best, plain.best)
        assert torch.equal(got.probs, plain.probs)
    eng.close()
    ceng.close()
