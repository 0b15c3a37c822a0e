"""Non-default thermal tracking configurations (every knob of config/trackingconfig.py:126-177 moved) against what the
REFERENCE did with them: the drop-in extract_file on the HIP kernels, configured from the SAME YAML text.  A knob that
were parsed but not plumbed to the kernels would show here.  Golden: tests/golden/make_golden_config_tracks.py."""
import os

import numpy as np
import pytest

from config_tracks_common import GOLDEN, load_config, load_golden, scene_frames
from helpers import encode_cptv

pytestmark = pytest.mark.gpu


def test_extract_file_follows_the_reference_under_non_default_configs(tmp_path):
    from cpx.track.trackextractor import extract_file

    rows, offsets, T, variants, cases = load_golden()
    n_kept = n_rej = 0
    for k, case in enumerate(cases):
        cfg = load_config(variants[case["variant"]], tmp_path, case["variant"])
        if case["scene"] == "possum":
            path = os.path.join(GOLDEN, "possum.cptv")
        else:
            frames, t_on, ffc, _, model = scene_frames(case, T)
            path = str(tmp_path / ("%s_%s.cptv" % (case["variant"], case["scene"])))
            encode_cptv(path, frames, [16] * T, time_on=t_on, last_ffc=ffc, model=model.encode())
        clip, _, meta = extract_file(path, cfg, False, save_meta=False)
        tag = (case["variant"], case["scene"])
        every = sorted(list(clip.tracks) + [t for _, t in clip.filtered_tracks], key=lambda t: t.get_id())
        got = [(t.get_id(), int(r.x), int(r.y), int(r.width), int(r.height), int(r.mass), int(r.frame_number),
                int(bool(r.blank))) for t in every for r in t.bounds_history]
        want = rows[offsets[k]:offsets[k + 1]]
        assert np.array_equal(np.asarray(got, np.int32).reshape(-1, 8), want), tag
        assert [t.get_id() for t in clip.tracks] == [c[0] for c in case["kept"]], tag
        for t, c in zip(clip.tracks, case["kept"]):
            assert float(t.stats.score) == pytest.approx(c[1], rel=1e-6), tag
        assert [[r, t.get_id()] for r, t in clip.filtered_tracks] == case["filtered"], tag
        assert len(meta["tracks"]) == len(case["kept"])
        n_kept += len(case["kept"])
        n_rej += len(case["filtered"])
    assert n_kept > 30 and n_rej > 20
