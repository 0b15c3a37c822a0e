"""cpx_cptv_unpack (GPU CPTV payload decode) against the host reader: bit-exact on the
fixture clips, on synthetic files with odd delta widths, and on a ragged multi-clip batch."""
import os

import numpy as np
import pytest

from helpers import encode_cptv

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from cpx.engine import TrackEngine

    eng = TrackEngine(width=160, height=120, max_frames=1024)
    yield eng
    eng.close()


def host_frames(path):
    from cpx.cptv import CptvReader

    return np.stack([f.pix for f in CptvReader(path).read_all()])


@pytest.mark.parametrize("name", ["possum", "hedgehog"])
def test_fixture_decode(engine, golden_dir, name):
    from cpx.cptv import decode_clips_on_device

    path = os.path.join(golden_dir, name + ".cptv")
    headers, metas, frames_dev, offs = decode_clips_on_device(engine, [path])
    want = host_frames(path)
    got = frames_dev.cpu().numpy().view(np.uint16)
    assert list(offs) == [0, want.shape[0]] and headers[0].x_resolution == 160
    np.testing.assert_array_equal(got, want)


def synth_clip(rng, n, widths):
    H, W = 120, 160
    frames = np.zeros((n, H, W), np.uint16)
    cur = np.full((H, W), 3000, np.int64)
    for i in range(n):
        w = widths[i]
        if w >= 31:
            cur = rng.integers(5000, 60000, (H, W))
        elif w > 3 and i > 0:
            lim = min(1 << (w - 1), 4000) // 8 - 1
            cur = cur + rng.integers(-lim, lim + 1, (H, W))
        frames[i] = cur.astype(np.uint16)
    return frames


def test_odd_widths_and_ragged_batch(engine, tmp_path):
    from cpx.cptv import decode_clips_on_device

    rng = np.random.default_rng(5)
    specs = [(3, 5, 9, 12, 16, 17), (8,) * 11, (32, 31, 24, 8), (2, 7, 13)]
    paths, want = [], []
    for k, widths in enumerate(specs):
        frames = synth_clip(rng, len(widths), widths)
        p = tmp_path / ("c%d.cptv" % k)
        encode_cptv(p, frames, widths)
        np.testing.assert_array_equal(host_frames(p), frames)
        paths.append(p)
        want.append(frames)
    headers, metas, frames_dev, offs = decode_clips_on_device(engine, paths)
    got = frames_dev.cpu().numpy().view(np.uint16)
    assert list(offs) == list(np.cumsum([0] + [len(s) for s in specs]))
    np.testing.assert_array_equal(got, np.concatenate(want))


def test_wrong_resolution_is_refused(engine, tmp_path):
    from cpx.cptv import decode_clips_on_device

    p = tmp_path / "small.cptv"
    encode_cptv(p, np.full((2, 12, 10), 3000, np.uint16), (8, 8))
    with pytest.raises(ValueError):
        decode_clips_on_device(engine, [p])
