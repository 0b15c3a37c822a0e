"""The raw-gray containers of the IR file driver (cpx/track/grayvideo.py): .npy and YUV4MPEG2, luma only."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "classifier-pipeline_amd"))


def write_y4m(path, frames, chroma="420jpeg", fps=(10, 1)):
    T, H, W = frames.shape
    extra = {"mono": 0, "420jpeg": 2 * ((W + 1) // 2) * ((H + 1) // 2), "422": 2 * ((W + 1) // 2) * H, "444": 2 * W * H}[chroma]
    with open(path, "wb") as fh:
        fh.write(b"YUV4MPEG2 W%d H%d F%d:%d Ip A1:1 C%s\n" % (W, H, fps[0], fps[1], chroma.encode()))
        for f in frames:
            fh.write(b"FRAME\n" + f.tobytes() + bytes([128]) * extra)


@pytest.mark.parametrize("chroma", ["mono", "420jpeg", "422", "444"])
def test_y4m_luma(tmp_path, chroma):
    from cpx.track.grayvideo import read_gray_frames

    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (5, 48, 64), dtype=np.uint8)
    p = tmp_path / "v.y4m"
    write_y4m(p, frames, chroma)
    got, fps = read_gray_frames(p)
    assert np.array_equal(got, frames) and fps == 10.0


def test_npy_and_refusals(tmp_path):
    from cpx.track.grayvideo import read_gray_frames

    frames = np.arange(3 * 4 * 8, dtype=np.uint8).reshape(3, 4, 8)
    np.save(tmp_path / "v.npy", frames)
    got, fps = read_gray_frames(tmp_path / "v.npy")
    assert np.array_equal(np.asarray(got), frames) and fps is None
    np.save(tmp_path / "f.npy", frames.astype(np.float32))
    with pytest.raises(ValueError):
        read_gray_frames(tmp_path / "f.npy")
    (tmp_path / "t.y4m").write_bytes(b"YUV4MPEG2 W8 H4 F10:1 C420p10\nFRAME\n" + bytes(200))
    with pytest.raises(ValueError):
        read_gray_frames(tmp_path / "t.y4m")
    (tmp_path / "s.y4m").write_bytes(b"YUV4MPEG2 W8 H4 F10:1 Cmono\nFRAME\n" + bytes(10))
    with pytest.raises(ValueError):
        read_gray_frames(tmp_path / "s.y4m")
    with pytest.raises(ValueError):
        read_gray_frames(tmp_path / "x.avi")
