"""Host CPTV v2 reader: round trips through the test-side encoder at several delta widths, and the section index the
device decoder consumes."""
import os

import numpy as np
import pytest

from helpers import encode_cptv
from cpx.cptv import CptvReader


@pytest.mark.parametrize("widths", [(8, 8, 8, 8), (5, 9, 12, 16), (17, 31, 32, 16), (1, 3, 7, 11)])
def test_reader_round_trip(tmp_path, widths):
    rng = np.random.default_rng(sum(widths))
    H, W = 12, 10
    frames = np.zeros((len(widths), H, W), np.uint16)
    base = rng.integers(3000, 3500, (H, W))
    cur = base.copy()
    for i, w in enumerate(widths):
        # keep successive scan-order deltas inside the chosen width
        lim = max(0, min(1 << (w - 1), 4000) // 8 - 1) if w > 3 else 0
        step = rng.integers(-lim, lim + 1, (H, W)) if lim else np.zeros((H, W), np.int64)
        if i == 0 and w < 16:
            cur = np.full((H, W), 3000)  # a flat first frame fits any width
        elif w >= 31:
            cur = rng.integers(0, 65536, (H, W))
        else:
            cur = cur + step
        frames[i] = cur.astype(np.uint16)
    path = tmp_path / "t.cptv"
    try:
        encode_cptv(path, frames, widths, time_on=[1000 * i for i in range(len(widths))], last_ffc=[10] * len(widths))
    except AssertionError:
        pytest.skip("random data did not fit the width")
    r = CptvReader(path)
    assert (r.get_header().x_resolution, r.get_header().y_resolution) == (W, H)
    got = r.read_all()
    assert len(got) == len(widths)
    for i, f in enumerate(got):
        np.testing.assert_array_equal(f.pix, frames[i])
        assert f.time_on == 1000 * i and f.last_ffc_time == 10
    metas, offsets, ws = CptvReader(path).scan()
    assert list(ws) == list(widths) and len(metas) == len(widths) and metas[0].pix is None
    assert np.all(np.diff(offsets) > 0)


def test_reader_fixture_index(golden_dir):
    path = os.path.join(golden_dir, "possum.cptv")
    frames = CptvReader(path).read_all()
    metas, offsets, ws = CptvReader(path).scan()
    assert len(frames) == len(metas) == 161
    assert [f.time_on for f in frames] == [m.time_on for m in metas]
    assert [f.background_frame for f in frames] == [m.background_frame for m in metas]


def test_reader_rejects_truncated(tmp_path):
    frames = np.full((2, 6, 8), 3000, np.uint16)
    path = tmp_path / "t.cptv"
    encode_cptv(path, frames, (8, 8))
    import gzip
    raw = gzip.open(path, "rb").read()
    with gzip.open(path, "wb") as f:
        f.write(raw[:-5])
    r = CptvReader(path)
    assert r.next_frame() is not None
    with pytest.raises(ValueError):
        r.next_frame()


def test_encode_cptv_round_trip(tmp_path):
    """cpx.cptv.encode_cptv (the writer behind the synthetic recordings of bench.py from_files) against the reader:
    frames, times, background flag, model; 8-bit and 16-bit frames in one file."""
    from cpx import synth
    from cpx.cptv import CptvReader, encode_cptv

    rng = np.random.default_rng(4)
    clip = synth.make_clip(rng, 12, max_blobs=2)
    clip[5, :, ::2] += 3000           # stripes: this frame and the next need 16 bits per delta
    t_on = [1000 + 111 * i for i in range(12)]
    blob = encode_cptv(clip, t_on, [7] * 12, model=b"lepton3.5", background_first=True)
    p = tmp_path / "x.cptv"
    p.write_bytes(blob)
    r = CptvReader(str(p))
    h = r.get_header()
    assert (h.model, h.x_resolution, h.y_resolution, h.has_background_frame) == ("lepton3.5", 160, 120, True)
    _, _, widths = CptvReader(str(p)).scan()
    assert set(widths.tolist()) == {8, 16}
    fr = r.read_all()
    assert len(fr) == 12 and fr[0].background_frame and not fr[1].background_frame
    assert [f.time_on for f in fr] == t_on and all(f.last_ffc_time == 7 for f in fr)
    assert np.array_equal(np.stack([f.pix for f in fr]), clip)
