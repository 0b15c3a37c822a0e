"""The HIP kernels that restate OpenCV operators on their own (cpx_ir_detect = MORPH_OPEN + threshold + components in
OpenCV's numbering; cpx_mog2_apply = BackgroundSubtractorMOG2) against OpenCV itself, from the fixture of
tools/cv2_dump.py.  Skipped while CPX_CV2_FIXTURE is unset (tests/test_cv2_parity_cpu.py pins the oracle the other
kernels -- blur / threshold / close / components inside cpx_frame_kernel, non-local means, resize, Kalman -- are
compared with bit for bit by the always-on GPU tests)."""
import os

import numpy as np
import pytest

FIXTURE = os.environ.get("CPX_CV2_FIXTURE")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not FIXTURE, reason="needs a cv2 fixture: python tools/cv2_dump.py <out>, CPX_CV2_FIXTURE=<out>")]


def test_ir_detect_kernel_matches_opencv():
    import torch

    from cpx._lib import COMPONENT_DTYPE
    from cpx.engine import TrackEngine

    z = np.load(FIXTURE)
    eng = TrackEngine(model="lepton3")
    imgs = np.stack([z["ir_%d_in" % i] for i in range(4)])
    counts, comps, labels = eng.ir_detect(torch.from_numpy(imgs).to(eng.device), threshold=0, max_components=8192,
                                          want_labels=True)
    labels = labels.cpu().numpy()
    for i in range(4):
        want_labels, want_stats = z["ir_%d_labels" % i], z["ir_%d_stats" % i]
        assert counts[i] == len(want_stats) - 1
        assert np.array_equal(labels[i], want_labels), "numbering / partition, mask %d" % i
        c = comps[i, : counts[i]]
        got = np.stack([c["x"], c["y"], c["width"], c["height"], c["area"]], axis=1)
        assert np.array_equal(got, want_stats[1:]), i
    eng.close()


def test_mog2_kernel_matches_opencv():
    import torch

    from cpx.engine import TrackEngine
    from cpx.track.irdetect import MOG2Background

    z = np.load(FIXTURE)
    frames = z["mog2_frames"]
    eng = TrackEngine(model="lepton3")
    bg = MOG2Background(eng, frames.shape[2], frames.shape[1], n_streams=1)
    for f, want in zip(frames, z["mog2_masks"]):
        bg.update_background(torch.from_numpy(np.ascontiguousarray(f)).to(eng.device))
        assert np.array_equal(bg._background.cpu().numpy(), want)
    bg.close()
    eng.close()
