"""The background update rule of WeightedBackground.process_frame (piclassifier/motiondetector.py:212-223),
    new = np.where(bg < f - w, bg, f);  w = np.where(bg < f - w, w + weight_add, 0)
evaluated by the kernel as an integer threshold test with a float64 fall-back (csrc/cpx_track.hip, streaming pass;
tables built in csrc/cpx_api.cpp).  Adversarial states: every pixel sits within one count of the point where the
float64 rounding of f - w decides (w = k-fold accumulation of weight_add, f - bg in {m-1, m, m+1} around m = rint(w)),
with k on both sides of the LDS tables' lengths.  The reference expression itself (NumPy float64) is the checker."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("weight_add", [0.1, 1.0, 0.3, 0.25, 0.7])
def test_background_keep_rule_at_the_rounding_edge(weight_add):
    from cpx import _lib
    from cpx.engine import TrackEngine

    H, W, e = 120, 160, 1
    max_frames = 2400
    eng = TrackEngine(model="lepton3", max_frames=max_frames, weight_add=weight_add)
    rng = np.random.default_rng(int(weight_add * 1000))
    # the k-fold accumulation, exactly as NumPy does it (repeated float64 addition)
    wt = np.zeros(max_frames + 2)
    for k in range(1, wt.size):
        wt[k] = wt[k - 1] + weight_add
    total_checked = ambiguous = 0
    for rep in range(3):
        k = rng.integers(0, max_frames, size=(H - 2 * e, W - 2 * e))
        # favour the counts where w is (nearly) an integer, and both sides of the 512 / 1024 table lengths
        near = np.nonzero(np.abs(wt[:max_frames] - np.rint(wt[:max_frames])) < 1e-6)[0]
        pick = rng.random(k.shape) < 0.6
        k[pick] = rng.choice(near, size=int(pick.sum()))
        w = wt[k]
        m = np.rint(w).astype(np.int64)
        d = m + rng.integers(-1, 2, size=m.shape)                      # f - bg in {m-1, m, m+1}
        bg_in = rng.integers(1000, 40000, size=(H, W)).astype(np.int64)
        f_in = np.clip(bg_in[e:-e, e:-e] + d, 0, 65535)
        frame = rng.integers(0, 65535, size=(H, W)).astype(np.int64)   # borders: anything
        frame[e:-e, e:-e] = f_in
        eng.set_background(0, bg_in.astype(np.float32), weights=w, average=float(np.average(bg_in[e:-e, e:-e])))
        dev = eng.upload_frames(frame.astype(np.uint16)[None])
        # (the noise frame has more components than the handle's capacity: that status is not what is tested here)
        eng.track_batch(dev, np.array([0, 1], np.int32), eng.make_meta(1), flags=_lib.TRACK_KEEP_BACKGROUND)
        eng.synchronize()
        got_bg, got_w, _ = eng.get_background(0)
        # the reference expression: one frame in the 45-frame window, so the mean fed to the model is the frame
        b = bg_in[e:-e, e:-e].astype(np.float64)
        f = f_in.astype(np.float64)
        keep = b < f - w
        want_bg = np.where(keep, b, f)
        want_w = np.where(keep, w + weight_add, 0.0)
        assert np.array_equal(got_bg[e:-e, e:-e].astype(np.float64), want_bg), rep
        assert np.array_equal(got_w, want_w), rep
        total_checked += keep.size
        ambiguous += int(((np.abs(w - m) < 1e-6) & (w != m) & (d == m)).sum())
    assert total_checked > 50000
    if weight_add in (0.1, 0.3, 0.7):
        assert ambiguous > 20         # the float64 fall-back decided some pixels
    eng.close()
