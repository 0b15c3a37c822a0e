"""Non-default thermal tracking configurations (every knob of config/trackingconfig.py:126-177 moved) against what the
REFERENCE did with them: the oracle restatement on CPU.  Golden: tests/golden/make_golden_config_tracks.py."""
import numpy as np
import pytest

from config_tracks_common import load_config, load_golden, oracle_config, scene_frames


def test_oracle_follows_the_reference_under_non_default_configs(tmp_path):
    import track_oracle as to

    rows, offsets, T, variants, cases = load_golden()
    reasons = set()
    for k, case in enumerate(cases):
        cfg = load_config(variants[case["variant"]], tmp_path, case["variant"])
        frames, t_on, ffc, bgf, model = scene_frames(case, T)
        out = to.track_clip(frames, t_on, ffc, bgf, oracle_config(cfg, model), keep=True)
        mine = sorted(list(out["tracks"]) + [t for _, t in out["filtered_tracks"]], key=lambda t: t.id)
        got = [(t.id, r.x, r.y, r.width, r.height, int(r.mass), r.frame_number, int(bool(r.blank)))
               for t in mine for r in t.bounds]
        want = rows[offsets[k]:offsets[k + 1]]
        tag = (case["variant"], case["scene"])
        assert np.array_equal(np.asarray(got, np.int32).reshape(-1, 8), want), tag
        assert [t.id for t in out["tracks"]] == [c[0] for c in case["kept"]], tag
        for t, c in zip(out["tracks"], case["kept"]):
            assert t.stats["score"] == pytest.approx(c[1], rel=1e-9), tag
        assert [[r, t.id] for r, t in out["filtered_tracks"]] == case["filtered"], tag
        reasons |= set(r for r, _ in case["filtered"])
    assert len(cases) >= 12 and len(reasons) >= 4
