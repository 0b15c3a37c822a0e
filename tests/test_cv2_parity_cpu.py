"""oracle/cv2_shim.py and oracle/mog2_oracle.c against OpenCV itself, from a fixture a maintainer makes where OpenCV
is installed (the build container has none):

    python tools/cv2_dump.py /tmp/cv2_fixture.npz
    CPX_CV2_FIXTURE=/tmp/cv2_fixture.npz python -m pytest tests/test_cv2_parity_cpu.py

Skipped while CPX_CV2_FIXTURE is unset.  What this pins beyond tests/clips/possum.txt: component NUMBERING with
several components (SURVEY a7'; with a canonical-relabel fallback that says which of the two failed), cv2.resize's
float32 rounding (A.8), the Kalman filter's last bits (A.5), the 1 x 2 MORPH_OPEN anchor and MOG2 (f4)."""
import os

import numpy as np
import pytest

FIXTURE = os.environ.get("CPX_CV2_FIXTURE")
pytestmark = pytest.mark.skipif(not FIXTURE, reason="needs a cv2 fixture: python tools/cv2_dump.py <out>, CPX_CV2_FIXTURE=<out>")


@pytest.fixture(scope="module")
def z():
    return np.load(FIXTURE)


def canonical(labels):
    """Labels renumbered by first appearance in raster order: equal partitions give equal arrays."""
    flat = labels.reshape(-1)
    _, first = np.unique(flat, return_index=True)
    order = np.argsort(first)
    vals = np.unique(flat)[order]
    lut = {int(v): (0 if v == 0 else None) for v in vals}
    k = 1
    for v in vals:
        if v != 0:
            lut[int(v)] = k
            k += 1
    return np.vectorize(lut.get)(labels)


def check_components(labels, stats, cent, want_labels, want_stats, want_cent, what):
    same_partition = np.array_equal(canonical(labels), canonical(want_labels))
    assert same_partition, "%s: the components themselves differ" % what
    assert np.array_equal(labels, want_labels), "%s: same components, different NUMBERING (SURVEY a7')" % what
    assert np.array_equal(stats, want_stats), what
    if want_cent is not None:
        assert np.allclose(cent, want_cent, rtol=0, atol=1e-9), what


def test_detect_objects_chain(z):
    import cv2_shim as cv2

    for i in range(len(z["detect_in"])):
        u8 = np.uint8(z["detect_in"][i])
        blur = cv2.GaussianBlur(u8, (5, 5), 0)
        assert np.array_equal(blur, z["detect_%d_blur" % i]), i
        _, th = cv2.threshold(blur, float(z["detect_thresh"][i]), 255, cv2.THRESH_BINARY)
        assert np.array_equal(th, z["detect_%d_thresh" % i]), i
        closed = cv2.morphologyEx(th, cv2.MORPH_CLOSE, (5, 5))
        assert np.array_equal(closed, z["detect_%d_close" % i]), i
        n, labels, stats, cent = cv2.connectedComponentsWithStats(closed)
        check_components(labels, stats, cent, z["detect_%d_labels" % i], z["detect_%d_stats" % i],
                         z["detect_%d_centroids" % i], "detect %d" % i)


def test_component_numbering_and_ir_chain(z):
    import cv2_shim as cv2

    for i in range(4):
        n, labels, stats, cent = cv2.connectedComponentsWithStats(z["cc_%d_in" % i])
        check_components(labels, stats, cent, z["cc_%d_labels" % i], z["cc_%d_stats" % i], z["cc_%d_centroids" % i], "cc %d" % i)
        opened = cv2.morphologyEx(z["ir_%d_in" % i], cv2.MORPH_OPEN, (15, 15))
        assert np.array_equal(opened, z["ir_%d_open" % i]), "MORPH_OPEN with a tuple kernel (anchor), mask %d" % i
        _, th = cv2.threshold(opened, 0, 255, cv2.THRESH_BINARY)
        n, labels, stats, _ = cv2.connectedComponentsWithStats(th)
        check_components(labels, stats, None, z["ir_%d_labels" % i], z["ir_%d_stats" % i], None, "ir %d" % i)


def test_nlm(z):
    import cv2_shim as cv2

    for a, want in zip(z["nlm_in"], z["nlm_out"]):
        assert np.array_equal(cv2.fastNlMeansDenoising(a, None), want)


def test_resize_area_uint8(z):
    """cv2.resize(uint8, INTER_AREA) at integer ratios (the IR tracker's `scale`) -> cv2_shim.resize, what
    cpx_ir_resize_area is checked against (tests/test_irtrack_gpu.py)."""
    import cv2_shim

    k = 0
    while "area_%d_in" % k in z.files:
        f = int(z["area_%d_factor" % k])
        for tag in ("", "gray_"):
            src = z["area_%d_%sin" % (k, tag)]
            got = cv2_shim.resize(src, (src.shape[1] // f, src.shape[0] // f), interpolation=cv2_shim.INTER_AREA)
            assert np.array_equal(got, z["area_%d_%sout" % (k, tag)]), (k, tag)
        k += 1
    if k == 0:
        pytest.skip("the fixture predates the INTER_AREA entries: re-run tools/cv2_dump.py")


def test_resize_float32(z):
    import cv2_shim as cv2

    k = 0
    while "resize_%d_in" % k in z.files:
        src, (dw, dh) = z["resize_%d_in" % k], z["resize_%d_size" % k]
        lin = cv2.resize(src, (int(dw), int(dh)), interpolation=cv2.INTER_LINEAR)
        near = cv2.resize(src, (int(dw), int(dh)), interpolation=cv2.INTER_NEAREST)
        assert np.array_equal(near, z["resize_%d_nearest" % k]), k
        assert np.array_equal(lin, z["resize_%d_linear" % k]), (k, float(np.abs(lin - z["resize_%d_linear" % k]).max()))
        k += 1
    assert k >= 4


def test_kalman(z):
    import cv2_shim as cv2

    for i in range(3):
        kf = cv2.KalmanFilter(4, 2)
        kf.measurementMatrix = np.eye(2, 4, dtype=np.float32)
        kf.transitionMatrix = np.array([[1, 0, 1, 0], [0, 1, 0, 1], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
        kf.processNoiseCov = np.eye(4, 4, dtype=np.float32) * 0.03
        got = []
        for p, b in zip(z["kalman_%d_pts" % i], z["kalman_%d_blank" % i]):
            if not b:
                kf.correct(p)
            got.append(np.asarray(kf.predict()).reshape(-1).copy())
        assert np.array_equal(np.stack(got), z["kalman_%d_pred" % i]), i


def test_contours(z):
    import cv2_shim as cv2

    for i in range(4):
        contours, _ = cv2.findContours(z["contour_%d_in" % i], cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_TC89_L1)
        assert [len(c) for c in contours] == list(z["contour_%d_lengths" % i]), i
        pts = np.concatenate([np.asarray(c).reshape(-1, 2) for c in contours]) if contours else np.zeros((0, 2), np.int32)
        assert np.array_equal(pts, z["contour_%d_points" % i]), i


def test_mog2_oracle(z):
    import mog2_oracle as mo

    frames = z["mog2_frames"]
    m = mo.MOG2(frames.shape[2], frames.shape[1])
    for f, want in zip(frames, z["mog2_masks"]):
        assert np.array_equal(m.apply(f), want)
    assert np.array_equal(m.getBackgroundImage(), z["mog2_background"])
