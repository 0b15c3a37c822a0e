"""Size-independent properties at BASELINE.json's full sizes (the oracle covers the same kernels at small sizes in
test_track_gpu / test_pipeline_gpu):
  * configs[1] -- 4096 clips x 270 frames through the track + association kernels: the batch is K distinct clips
    (two of them checked against the oracle here) replicated on the device; every replica must produce byte-identical
    component, frame-info and track records (no cross-clip interference at the full grid), and a second run must
    reproduce the first (determinism / idempotence of the clip state handling).
  * configs[2] -- 1024-clip batch through the whole pipeline (>= 1024 tracks): identical class scores per replica and
    kept-track counts that add up."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _valid_comps(res, cap):
    """component records with the slots past n_components (unwritten, caller-allocated memory) zeroed"""
    import torch

    info = res.info_dev.reshape(-1, 20)
    comps = res.comps_dev.reshape(info.shape[0], cap, 8)
    keep = torch.arange(cap, device=comps.device)[None, :] < info[:, 1:2]
    return comps * keep[:, :, None].to(comps.dtype)


def _digest(t, groups):
    """rows of t viewed as [groups, -1] all equal to row 0?"""
    v = t.reshape(groups, -1)
    return bool((v == v[0:1]).all().item())


def test_track_stage_replication_at_4096x270():
    import torch

    import track_oracle as to
    from cpx import synth
    from cpx.engine import TrackEngine

    K, T, R = 4, 270, 1024  # K * R = 4096 clips
    frames, offs = synth.make_batch(K, T, seed=1234)
    eng = TrackEngine(model="lepton3", max_frames=T)
    t_on, ffc = synth.frame_times(T)
    meta1 = np.concatenate([eng.make_meta(T, t_on, ffc) for _ in range(K)])
    base = eng.upload_frames(frames)
    # small run first: the oracle pins two of the K clips
    small = eng.track_batch(base, offs, meta1)
    small.check()
    for b in (0, K - 1):
        out = to.track_clip(frames[offs[b]:offs[b + 1]], t_on, ffc, None, to.OracleConfig("lepton3"),
                            do_tracking=False)
        for i, fr in enumerate(out["frames"]):
            comps = small.components(int(offs[b]) + i)
            assert len(comps) == fr["n_components"]
            for c, s in zip(comps, fr["stats"]):
                assert (c["x"], c["y"], c["width"], c["height"], c["area"]) == tuple(int(v) for v in s)
    small_comps = _valid_comps(small, eng.cap).clone()
    small_info = small.info_dev.clone()
    # full size: replicate on the device
    big = base.unsqueeze(0).expand(R, -1, -1, -1).reshape(R * K * T, eng.height, eng.width).contiguous()
    offs_big = (np.arange(R * K + 1, dtype=np.int64) * T).astype(np.int32)
    meta_big = np.tile(meta1, R)
    res = eng.track_batch(big, offs_big, meta_big)
    assoc = eng.associate_batch(res, offs_big, meta_big, want_regions=False)
    eng.synchronize()
    assert int(res.info_dev.reshape(-1, 20)[:, 2].abs().max().item()) == 0  # cpx_frame_info.status
    comps = _valid_comps(res, eng.cap)
    assert _digest(comps, R) and _digest(res.info_dev, R)
    assert torch.equal(comps.reshape(R, -1)[0], small_comps.reshape(-1))
    assert torch.equal(res.info_dev.reshape(R, -1)[0], small_info.reshape(-1))
    assert _digest(assoc.tracks_dev, R) and _digest(assoc.ntracks_dev, R) and _digest(assoc.pool_dev, R)
    assert int(assoc.status_dev.abs().max().item()) == 0
    assert int(assoc.ntracks_dev.sum().item()) >= R  # the batch did contain objects
    # same batch again on the same handle: clip state is rebuilt per call
    first = comps
    res2 = eng.track_batch(big, offs_big, meta_big)
    eng.synchronize()
    assert torch.equal(_valid_comps(res2, eng.cap), first)
    eng.close()


def test_pipeline_replication_at_1024_clips():
    import torch

    import cnn_oracle as cnn
    from cpx import synth
    from cpx.engine import TrackEngine
    from cpx.ml_tools import wrresnet as wr
    from cpx.pipeline import BatchPipeline

    K, T, R = 8, 270, 128
    frames, offs = synth.make_batch(K, T, seed=77)
    eng = TrackEngine(model="lepton3", max_frames=T)
    rng = np.random.default_rng(3)
    w = wr.random_weights(17, seed=2)
    w = cnn.calibrate_bn(w, rng.uniform(0, 255, size=(2, 160, 160, 2)).astype(np.float32))
    net = wr.WRResNetDevice(eng, w, 17)
    pipe = BatchPipeline(eng, net, n_labels=17, fp_index=4)
    meta1 = np.concatenate([eng.make_meta(T) for _ in range(K)])
    base = eng.upload_frames(frames)
    one = pipe.run(base, offs, meta1)
    assert one.n_tracks >= 8
    big = base.unsqueeze(0).expand(R, -1, -1, -1).reshape(R * K * T, eng.height, eng.width).contiguous()
    offs_big = (np.arange(R * K + 1, dtype=np.int64) * T).astype(np.int32)
    res = pipe.run(big, offs_big, np.tile(meta1, R))
    assert res.n_tracks == R * one.n_tracks and res.n_samples == R * one.n_samples
    assert res.n_tracks >= 1024
    assert np.array_equal(res.counts.reshape(R, -1), np.tile(one.counts.reshape(1, -1), (R, 1)))
    sc = res.scores.reshape(R, one.n_tracks, -1)
    assert torch.equal(sc, one.scores.unsqueeze(0).expand_as(sc))
    assert torch.equal(res.best.reshape(R, -1), one.best.unsqueeze(0).expand(R, -1))
    tc = res.track_clip.reshape(R, one.n_tracks, 2)
    assert torch.equal(tc[:, :, 1], one.track_clip[:, 1].unsqueeze(0).expand(R, -1))
    eng.close()
