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
