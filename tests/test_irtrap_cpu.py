"""The IR tracker's host-side trap geometry and filters against what the REFERENCE's own functions
returned (tests/golden/irtrap_golden.json, make_golden_irtrap.py): Line / get_trap_lines, inside_trap_top,
inside_trap_bottom with Track.update_trapped_state, filter_track, rect_distance -- for both trap sizes."""
import json
import os
from types import SimpleNamespace

import pytest

from helpers import GOLDEN


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLDEN, "irtrap_golden.json")) as fh:
        return json.load(fh)


@pytest.mark.parametrize("size", ["L", "S"])
def test_trap_geometry_equals_reference(gold, size):
    from cpx.config import Config
    from cpx.track.irtrackextractor import IRTrackExtractor
    from cpx.track.region import Region
    from cpx.track.track import Track

    cfg = Config.get_defaults()
    ex = IRTrackExtractor(cfg.tracking, trap_size=size)
    g = gold[size]
    assert [[ex.left_bottom.m, ex.left_bottom.c], [ex.right_bottom.m, ex.right_bottom.c]] == g["lines"]
    trapped = 0
    for seq in g["sequences"]:
        for which in ("top", "bottom"):
            track = Track("c", id=1, tracking_config=cfg.tracking["IR"])
            for i, b in enumerate(seq["boxes"]):
                r = Region(b[0], b[1], b[2], b[3], centroid=[b[0] + b[2] // 2, b[1] + b[3] // 2], mass=b[4], frame_number=i)
                track.bounds_history.append(r)
                got = (ex.inside_trap_top if which == "top" else ex.inside_trap_bottom)(track)
                assert [bool(got), bool(track.in_trap), int(track.direction), bool(r.in_trap)] == seq[which][i], (seq["boxes"], which, i)
            trapped += bool(track.in_trap)
    assert trapped >= 100
    clip = SimpleNamespace(frames_per_second=10, filtered_tracks=[])
    for c in g["filter_track"]:
        track = Track("c", id=1, tracking_config=cfg.tracking["IR"])
        track.bounds_history = [None] * c["len"]
        stats = SimpleNamespace(max_offset=c["max_offset"], frames_moved=c["frames_moved"])
        assert bool(ex.filter_track(clip, track, stats)) == c["filtered"]
    assert [r for r, _ in clip.filtered_tracks] == g["reasons"]


def test_rect_distance_equals_reference(gold):
    from cpx.track.irtrackextractor import rect_distance

    for a, b, want in gold["rect_distance"]:
        assert rect_distance(a, b) == want
