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


def test_mog2_oracle_behaviour():
    """oracle/mog2_oracle.c (parity with cv2 unpinned): the properties the published algorithm guarantees -- the first
    frame is all foreground (no mode yet), a static noisy scene becomes background, a moving object is foreground,
    the background image tracks the scene, and learning rate 0 adds no mode."""
    import mog2_oracle as mo

    H, W = 60, 80
    rng = np.random.default_rng(4)
    scene = rng.integers(40, 200, size=(H, W)).astype(np.int32)
    m = mo.MOG2(W, H)
    noisy = lambda: np.clip(scene + rng.integers(-2, 3, size=(H, W)), 0, 255).astype(np.uint8)
    assert (m.apply(noisy()) == 255).all()
    for _ in range(20):
        mask = m.apply(noisy())
    assert (mask > 0).mean() < 0.01
    assert np.abs(m.getBackgroundImage().astype(int) - scene).max() <= 3
    f = noisy()
    f[10:30, 20:50] = 250
    mask = m.apply(f)
    assert (mask[10:30, 20:50] == 255).all() and (mask > 0).sum() <= 20 * 30 + 50
    before = [a.copy() for a in m.state()]
    m.apply(f, learning_rate=0.0)
    after = m.state()
    assert np.array_equal(before[3], after[3])  # learning rate 0: no new modes (and alpha = 0 moves nothing)
    assert np.allclose(before[2], after[2])
    m.close()
