#!/usr/bin/env python3
"""Dump what OpenCV itself computes for every cv2 call on the hot path, on seeded inputs, into one .npz -- the fixture
against which oracle/cv2_shim.py (the NumPy restatement this build is checked with), oracle/mog2_oracle.c and the HIP
kernels can be pinned by anyone who has OpenCV (the build container does not: SURVEY F1).

    python tools/cv2_dump.py /tmp/cv2_fixture.npz
    CPX_CV2_FIXTURE=/tmp/cv2_fixture.npz python -m pytest tests/test_cv2_parity_cpu.py tests/test_cv2_parity_gpu.py

Imports only cv2 and numpy.  Calls and arguments are the reference's own (paths relative to /root/reference/src):
GaussianBlur / threshold / morphologyEx(MORPH_CLOSE, tuple kernel) / connectedComponentsWithStats
(ml_tools/imageprocessing.py:240-248), morphologyEx(MORPH_OPEN) + threshold + components (detect_objects_ir,
:185-199), fastNlMeansDenoising (track/cliptracker.py:116-117), resize float32 linear / nearest
(ml_tools/imageprocessing.py:77-82), resize uint8 INTER_AREA at integer ratios (track/irtrackextractor.py:445-451),
KalmanFilter(4, 2) (track/kalman.py:5-26), findContours(RETR_EXTERNAL,
CHAIN_APPROX_TC89_L1) (classify/thumbnail.py:91-96), createBackgroundSubtractorMOG2(history=1000,
detectShadows=False).apply (track/cliptracker.py:573-575)."""
import argparse

import numpy as np


def blobs(rng, h, w, n, lo=0, hi=255):
    """Smooth blobs + noise: images with structure at the scale of the 5x5 operators."""
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    img = rng.normal(20.0, 6.0, size=(h, w)).astype(np.float32)
    for _ in range(n):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        s = rng.uniform(1.5, 9.0)
        img += rng.uniform(30, 230) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
    return np.clip(img, lo, hi)


def masks(rng, h, w, kind):
    m = np.zeros((h, w), np.uint8)
    if kind == "speckle":
        m[rng.random((h, w)) < 0.12] = 255
    elif kind == "blocks":
        for _ in range(40):
            y, x = int(rng.integers(0, h - 4)), int(rng.integers(0, w - 4))
            m[y:y + int(rng.integers(1, 9)), x:x + int(rng.integers(1, 12))] = 255
    elif kind == "diagonals":   # components that touch only diagonally, born in different 2x2 blocks
        for k in range(0, min(h, w) - 1, 3):
            m[k, k] = m[k + 1, k + 1] = 255
            m[h - 1 - k, k] = 255
        m[rng.random((h, w)) < 0.02] = 255
    else:
        yy, xx = np.mgrid[:h, :w]
        for _ in range(12):
            cy, cx, ry, rx = rng.integers(0, h), rng.integers(0, w), rng.integers(2, 14), rng.integers(2, 20)
            m[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 255
    return m


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("out", help="the .npz to write")
    ap.add_argument("--seed", type=int, default=20251003)
    args = ap.parse_args()
    import cv2

    rng = np.random.default_rng(args.seed)
    out = {"cv2_version": np.array(cv2.__version__), "seed": np.array(args.seed)}
    H, W = 120, 160
    # ---- thermal detect_objects chain, stage by stage ----
    imgs = np.stack([blobs(rng, H, W, int(rng.integers(1, 9))) for _ in range(6)])
    thr = rng.uniform(8.0, 90.0, size=len(imgs))
    out["detect_in"], out["detect_thresh"] = imgs, thr
    for i, (im, t) in enumerate(zip(imgs, thr)):
        u8 = np.uint8(im)
        blur = cv2.GaussianBlur(u8, (5, 5), 0)
        _, th = cv2.threshold(blur, float(t), 255, cv2.THRESH_BINARY)
        closed = cv2.morphologyEx(th, cv2.MORPH_CLOSE, (5, 5))
        n, labels, stats, cent = cv2.connectedComponentsWithStats(closed)
        out["detect_%d_blur" % i], out["detect_%d_thresh" % i], out["detect_%d_close" % i] = blur, th, closed
        out["detect_%d_labels" % i], out["detect_%d_stats" % i], out["detect_%d_centroids" % i] = labels, stats, cent
    # ---- components on multi-component masks (numbering!) and the IR chain ----
    kinds = ["speckle", "blocks", "diagonals", "ellipses"]
    for i, kind in enumerate(kinds):
        m = masks(rng, H, W, kind)
        n, labels, stats, cent = cv2.connectedComponentsWithStats(m)
        out["cc_%d_in" % i], out["cc_%d_labels" % i], out["cc_%d_stats" % i], out["cc_%d_centroids" % i] = m, labels, stats, cent
        big = masks(rng, 480, 640, kind)
        opened = cv2.morphologyEx(big, cv2.MORPH_OPEN, (15, 15))
        _, th = cv2.threshold(opened, 0, 255, cv2.THRESH_BINARY)
        n, labels, stats, _ = cv2.connectedComponentsWithStats(th)
        out["ir_%d_in" % i], out["ir_%d_open" % i], out["ir_%d_labels" % i], out["ir_%d_stats" % i] = big, opened, labels, stats
    # ---- non-local means ----
    nl = np.stack([np.uint8(blobs(rng, 48, 64, 4)) for _ in range(3)] + [np.uint8(blobs(rng, H, W, 5))[30:78, 40:104]])
    out["nlm_in"] = nl
    out["nlm_out"] = np.stack([cv2.fastNlMeansDenoising(a, None) for a in nl])
    # ---- resize, float32 ----
    k = 0
    for (h, w), (dh, dw) in [((17, 23), (32, 32)), ((40, 9), (32, 7)), ((5, 5), (32, 32)), ((64, 50), (24, 19)),
                             ((31, 1), (32, 1)), ((28, 33), (27, 32))]:
        src = rng.normal(100, 40, size=(h, w)).astype(np.float32)
        out["resize_%d_in" % k] = src
        out["resize_%d_size" % k] = np.array([dw, dh])
        out["resize_%d_linear" % k] = cv2.resize(src, (dw, dh), interpolation=cv2.INTER_LINEAR)
        out["resize_%d_nearest" % k] = cv2.resize(src, (dw, dh), interpolation=cv2.INTER_NEAREST)
        k += 1
    # ---- resize, uint8 INTER_AREA at integer ratios (the IR tracker's `scale`, track/irtrackextractor.py:445-451) ----
    for k, f in enumerate((2, 4, 5)):
        src = masks(rng, 480, 640, "ellipses" if k else "blocks")
        src[rng.random(src.shape) < 0.05] = 255
        out["area_%d_in" % k] = src
        out["area_%d_factor" % k] = np.array(f)
        out["area_%d_out" % k] = cv2.resize(src, (640 // f, 480 // f), interpolation=cv2.INTER_AREA)
        gray = rng.integers(0, 256, size=(120, 160), dtype=np.uint8)
        out["area_%d_gray_in" % k] = gray
        out["area_%d_gray_out" % k] = cv2.resize(gray, (160 // f, 120 // f), interpolation=cv2.INTER_AREA)
    # ---- Kalman (track/kalman.py): correct then predict per seen frame, predict alone per blank ----
    for i in range(3):
        kf = cv2.KalmanFilter(4, 2)
        kf.measurementMatrix = np.eye(2, 4, dtype=np.float32)
        kf.transitionMatrix = np.array([[1, 0, 1, 0], [0, 1, 0, 1], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
        kf.processNoiseCov = np.eye(4, 4, dtype=np.float32) * 0.03
        pts = (np.cumsum(rng.normal(0, 2.5, size=(40, 2)), axis=0) + rng.uniform(20, 100, size=2)).astype(np.float32)
        blank = rng.random(40) < 0.2
        blank[0] = False
        res = []
        for p, b in zip(pts, blank):
            if not b:
                kf.correct(p)
            res.append(kf.predict().reshape(-1).copy())
        out["kalman_%d_pts" % i], out["kalman_%d_blank" % i], out["kalman_%d_pred" % i] = pts, blank, np.stack(res)
    # ---- contours ----
    for i, kind in enumerate(kinds):
        m = masks(rng, 40, 56, kind)
        contours, _ = cv2.findContours(m, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_TC89_L1)
        out["contour_%d_in" % i] = m
        out["contour_%d_lengths" % i] = np.array([len(c) for c in contours], np.int32)
        out["contour_%d_points" % i] = (np.concatenate([c.reshape(-1, 2) for c in contours]) if contours
                                        else np.zeros((0, 2), np.int32))
    # ---- MOG2 ----
    bg = cv2.createBackgroundSubtractorMOG2(history=1000, detectShadows=False)
    scene = np.uint8(blobs(rng, 96, 128, 6))
    frames, fg = [], []
    for t in range(24):
        f = np.clip(scene.astype(np.int16) + rng.integers(-3, 4, size=scene.shape), 0, 255).astype(np.uint8)
        f[30:50, (5 * t) % 100:(5 * t) % 100 + 18] = 240
        frames.append(f)
        fg.append(bg.apply(f))
    out["mog2_frames"], out["mog2_masks"], out["mog2_background"] = np.stack(frames), np.stack(fg), bg.getBackgroundImage()
    np.savez_compressed(args.out, **out)
    print("wrote %s: %d arrays (OpenCV %s)" % (args.out, len(out), cv2.__version__))


if __name__ == "__main__":
    main()
