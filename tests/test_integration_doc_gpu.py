"""The ctypes stub INTEGRATION.md shows a maintainer of the reference (section 2) is executed as written -- only the
library path and the three names it takes from the reference's scope (`clip`, `config`, `cptv_frames`) are supplied --
and must give what the package's own binding gives for the same frames: the document cannot drift from the ABI."""
import os
import re
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_stub_runs_as_documented():
    from cpx import _lib
    from cpx.engine import TrackEngine
    from helpers import load_clip

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. Bind the C-ABI directly"):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert "cpx_track_batch" in code and "CpxConfig" in code
    code = code.replace('C.CDLL("libcpx_hip.so")', "C.CDLL(%r)" % _lib.LIB_PATH)

    frames, t_on, ffc, bgf, hdr = load_clip("possum")
    frames = frames[:40]
    eng = TrackEngine(model="lepton3", max_frames=4096)   # also gives the thresholds the stub reads from `clip`
    scope = {
        "clip": types.SimpleNamespace(background_thresh=eng.cfg.background_thresh),
        "config": types.SimpleNamespace(denoise=False),
        "cptv_frames": [types.SimpleNamespace(pix=f) for f in frames],
    }
    exec(compile(code, "INTEGRATION.md#2", "exec"), scope)
    assert scope["rc"] == 0
    got = scope["c"]                                      # COMP records [n, 64]
    info = scope["info"].cpu().numpy().reshape(len(frames), 20)

    res = eng.track_batch(eng.upload_frames(frames), np.array([0, len(frames)], np.int32), eng.make_meta(len(frames)))
    res.check()
    n_seen = 0
    for f in range(len(frames)):
        want = res.components(f)
        n = len(want)
        assert int(info[f, 1]) == n, f
        for k in ("x", "y", "width", "height", "area"):
            assert np.array_equal(got[f, :n][k], want[k]), (f, k)
        n_seen += n
    assert n_seen > 0
    eng.close()


def test_whole_file_stub_runs_as_documented():
    """The second stub of section 2 (whole .cptv files inflated and indexed on the device), executed after the first
    one in the same scope with `paths` = the two fixture recordings: every file decodes, frame counts and sizes equal
    the host reader's."""
    import zlib

    from cpx import _lib
    from cpx.engine import TrackEngine
    from helpers import GOLDEN, load_clip

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. Bind the C-ABI directly"):]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    assert len(blocks) >= 2 and "cpx_cptv_inflate" in blocks[1]
    first = blocks[0].replace('C.CDLL("libcpx_hip.so")', "C.CDLL(%r)" % _lib.LIB_PATH)
    frames = load_clip("possum")[0][:4]
    eng = TrackEngine(model="lepton3", max_frames=64)
    paths = [os.path.join(GOLDEN, n + ".cptv") for n in ("possum", "hedgehog")]
    scope = {
        "clip": types.SimpleNamespace(background_thresh=eng.cfg.background_thresh),
        "config": types.SimpleNamespace(denoise=False),
        "cptv_frames": [types.SimpleNamespace(pix=f) for f in frames],
        "paths": paths,
    }
    exec(compile(first, "INTEGRATION.md#2a", "exec"), scope)
    exec(compile(blocks[1], "INTEGRATION.md#2b", "exec"), scope)
    assert scope["rc"] == 0
    res = scope["res"]
    raw = scope["out_dev"].cpu().numpy()
    for k, p in enumerate(paths):
        want = zlib.decompress(open(p, "rb").read(), 47)
        assert int(res["status"][k]) == 0 and int(res["out_bytes"][k]) == len(want)
        o = int(scope["files"]["out_offset"][k])
        assert raw[o:o + len(want)].tobytes() == want
        assert int(res["n_frames"][k]) == len(load_clip(("possum", "hedgehog")[k])[0])
        assert (int(res["width"][k]), int(res["height"][k])) == (160, 120)
    eng.close()
