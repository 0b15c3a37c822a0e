"""IR detection stage, oracle side: oracle/ir_oracle.py against what the reference's own
detect_objects_ir / IRTrackExtractor.merge_components returned for seeded 640x480 foreground masks
(tests/golden/ir_detect_golden.json, make_golden_ir.py)."""
import json
import os

import numpy as np

from helpers import GOLDEN, crc, ir_mask


def test_ir_oracle_matches_reference():
    import ir_oracle as iro

    with open(os.path.join(GOLDEN, "ir_detect_golden.json")) as fh:
        gold = json.load(fh)
    merged_total = 0
    for g in gold:
        img = ir_mask(g["case"])
        n, mask, stats = iro.detect_objects_ir(img, threshold=0)
        assert n - 1 == g["n"], g["case"]
        assert crc(mask.astype(np.int32)) == g["mask_crc"], g["case"]
        assert crc(stats[1:].astype(np.int32)) == g["stats_crc"], g["case"]
        if g["stats"] is not None:
            assert stats[1:].tolist() == g["stats"]
        merged = iro.merge_components(stats[1:].copy())
        assert [[int(v) for v in r] for r in merged] == g["merged"], g["case"]
        merged_total += len(merged)
    assert merged_total > 20


def test_host_merge_components_matches_reference():
    """The product's host-side merge (cpx/track/irdetect.py) on the reference's own component statistics."""
    from cpx.track import irdetect
    import ir_oracle as iro

    with open(os.path.join(GOLDEN, "ir_detect_golden.json")) as fh:
        gold = json.load(fh)
    n = 0
    for g in gold:
        _, _, stats = iro.detect_objects_ir(ir_mask(g["case"]), threshold=0)
        for scale in (None, 0.5, 2):
            want = iro.merge_components(stats[1:].copy(), scale)
            got = irdetect.merge_components(stats[1:].copy(), scale)
            assert [[int(v) for v in r] for r in got] == [[int(v) for v in r] for r in want], (g["case"], scale)
            if scale is None:
                assert [[int(v) for v in r] for r in got] == g["merged"], g["case"]
        n += len(g["merged"])
    rng = np.random.default_rng(3)
    for _ in range(300):
        a = rng.integers(0, 60, 5)
        b = rng.integers(0, 60, 5)
        assert irdetect.rect_distance(a, b) == iro.rect_distance(a, b)
    assert n > 20
