"""BASELINE configs[3] as bench.py --config4 runs it: seeded clips of varying length,
LPT partition over the ranks, device batches, one all-gather of [clip_id, track_id, 17 x f32] records.  Here rank 0's
shard of a 2-rank partition runs on one GPU with the collective going through RCCL (world 1), and a sample of its clips
is checked against the oracle chain: tracks, the reference's segment plan under identity draws, network inputs through
the PyTorch-CPU forward, per-track class scores (1e-3)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config4_rank0_shard_matches_oracle_chain():
    import torch
    import torch.distributed as dist

    import classify_oracle as co
    import cnn_oracle as cnn
    import track_oracle as to
    from cpx.ml_tools import datasetstructures as ds
    from cpx.ml_tools import wrresnet as wr
    from cpx.sharding import partition_clips, unpack_records
    from helpers import IdentityDraws

    sys.path.insert(0, REPO)
    import bench

    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
    dist.init_process_group("nccl", device_id=device)   # RCCL, world 1: the collective path of bench.py
    try:
        rng = np.random.default_rng(5)
        w = wr.random_weights(17, seed=2)
        w = cnn.calibrate_bn(w, rng.uniform(0, 255, size=(2, 160, 160, 2)).astype(np.float32))
        n_clips = 28
        wl = bench.Config4Workload(torch, device, 0, 0, 2, n_clips, seed=77, lo=90, hi=200, sub_frames=700, cnn_chunk=8,
                                   weights=w)
        lengths = bench.config4_lengths(n_clips, 77, 90, 200)
        shards = partition_clips(lengths, 2)
        assert wl.mine == shards[0] and sorted(shards[0] + shards[1]) == list(range(n_clips))
        loads = [int(lengths[s].sum()) for s in shards]
        assert abs(loads[0] - loads[1]) <= int(lengths.max())
        assert len(wl.subs) >= 3 and sorted(i for sb in wl.subs for i in sb) == wl.mine   # several device batches
        for sb in wl.subs:
            assert sum(int(lengths[i]) for i in sb) <= 700 or len(sb) == 1
        got = wl.step(dist)
        assert got.world == 1 and got.capacity == wl.record_capacity and int(got.counts[0]) <= got.capacity
        rec, rec2 = got.records(), wl.step(dist).records()
        assert torch.equal(rec, rec2)                      # a step is a pure function of the resident clips
        clip_ids, track_ids, scores = (v.cpu().numpy() for v in unpack_records(rec))
        assert rec.shape[1] == 2 + 17 and rec.dtype == torch.int32
        assert list(clip_ids) == sorted(clip_ids) and set(clip_ids) <= set(wl.mine)
        assert (scores.sum(axis=1) <= 1.0 + 1e-5).all()      # 1, or the low-evidence cap of 0.5
        with_tracks = [i for i in wl.mine if (clip_ids == i).any()]
        assert len(with_tracks) >= 3
        checked = 0
        for i in with_tracks[:2] + with_tracks[-2:]:
            clip = wl.clip_frames(i).cpu().numpy().view(np.uint16)
            assert clip.shape[0] == lengths[i]
            n = clip.shape[0]
            out = to.track_clip(clip, [100000 + 114 * q for q in range(n)], [40000] * n, None, to.OracleConfig("lepton3"),
                                keep=True)
            rows = np.nonzero(clip_ids == i)[0]
            assert [int(t) for t in track_ids[rows]] == [t.id for t in out["tracks"]], i   # kept tracks, score order
            fr = out["frames"]
            for row, t in zip(rows, out["tracks"]):
                with IdentityDraws():
                    segs, _ = ds.get_segments(i, t.id, t.bounds[0].frame_number, regions=np.array(t.bounds, dtype=object),
                                              segment_width=25, segment_frame_spacing=9, ffc_frames=out["ffc_frames"],
                                              repeats=1, min_frames=0, segment_types=[ds.SegmentType.ALL_RANDOM_MASKED],
                                              max_segments=None, dont_filter=False, min_segments=1, seed=None)
                frames = [np.array([int(f) for f in s.frame_indices]) for s in segs]
                by_frame = {r.frame_number: r for r in t.bounds}
                x, _ = co.preprocess_segments(lambda q: clip[q], lambda q: fr[q]["filtered"].astype(np.float64), by_frame,
                                              t.bounds, frames, 32, (1, 1, 158, 118))
                _, probs = cnn.forward(w, x)
                want = co.classified_track(probs, prediction_frames=frames)
                assert np.abs(scores[row] - want).max() <= 1e-3, (i, t.id)
                checked += 1
        assert checked >= 3
        wl.close()
    finally:
        dist.destroy_process_group()
