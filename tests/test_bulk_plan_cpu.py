"""Batch planning of the file-fed drivers (cpx/track/bulk.py): decode launches bounded by recordings and compressed bytes,
tracking groups bounded by recordings and frames -- host logic, no GPU."""
import numpy as np

from cpx.track.bulk import DecodedGroup, plan_decode_batches


def test_decode_batches_cover_every_recording_within_the_budgets():
    rng = np.random.default_rng(0)
    for n, batch, budget in ((0, 1024, 100), (1, 1024, 100), (10, 4, 10 ** 9), (5000, 1024, 10 ** 12), (300, 64, 5000)):
        sizes = rng.integers(1, 200, n).tolist()
        runs = plan_decode_batches(sizes, batch, budget)
        assert [a for a, _ in runs] == [0] * (n > 0) + [b for _, b in runs[:-1]]          # contiguous, in order
        assert (runs[-1][1] if runs else 0) == n
        for a, b in runs:
            assert 1 <= b - a <= batch
            assert sum(sizes[a:b]) <= budget or b - a == 1                                  # a big one goes alone
    # several full launches: the first is half-sized
    runs = plan_decode_batches([1] * 8192, 2048, 1 << 40)
    assert runs[0] == (0, 1024) and all(b - a == 2048 for a, b in runs[1:-1])
    # a recording larger than the budget forms a launch of its own
    assert plan_decode_batches([10, 500, 10, 10], 1024, 35) == [(0, 1), (1, 2), (2, 4)]


def test_tracking_groups_respect_recording_and_frame_budgets():
    lens = np.array([5, 7, 300, 4, 4, 4, 9, 1], np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    total = int(offs[-1])
    frames = np.arange(total)                                        # stands for the device frames: split only slices
    g = DecodedGroup((160, 120, None), list(range(100, 108)), ["h%d" % i for i in range(8)], offs,
                     np.arange(total), frames)
    for max_clips, max_frames in ((100, None), (3, None), (100, 12), (2, 10), (100, 1)):
        parts = list(g.split(max_clips, max_frames))
        assert [f for p in parts for f in p.files] == g.files
        assert np.array_equal(np.concatenate([p.frames_dev for p in parts]), frames)
        for p in parts:
            n = len(p.files)
            assert 1 <= n <= max_clips and p.offs[0] == 0 and int(p.offs[-1]) == len(p.frames_dev) == len(p.slots)
            assert max_frames is None or int(p.offs[-1]) <= max_frames or n == 1
    assert len(list(g.split(100, None))) == 1
