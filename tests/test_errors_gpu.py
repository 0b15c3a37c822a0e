"""Error behaviour of the C-ABI: capacities overflow into error codes (never truncated output), bad arguments and
unsupported geometries are refused with a message, handles survive errors."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_create_refuses_unsupported_geometry_and_bad_config():
    from cpx import _lib
    from cpx._lib import CpxError
    from cpx.engine import TrackEngine

    for kw in (dict(width=200, height=120), dict(width=162, height=120), dict(width=160, height=160),
               dict(width=160, height=120, edge_pixels=80), dict(width=160, height=120, max_components=0)):
        with pytest.raises(CpxError):
            TrackEngine(**kw)
    lib = _lib.load()
    assert lib.cpx_create(0, None, None) < 0
    assert b"null" in lib.cpx_last_error(None)


def test_null_and_out_of_sequence_arguments():
    from cpx._lib import CpxError
    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3", max_frames=32)
    lib, h = eng.lib, eng.h
    offs = np.array([0, 4], np.int32)
    meta = eng.make_meta(4)
    rc = lib.cpx_track_batch(h, None, offs.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(meta.ctypes.data), 1,
                             None, None, None, None, None)
    assert rc == -1 and "null" in eng._err()
    # an empty clip inside a batch
    frames = eng.upload_frames(np.full((4, 120, 160), 3000, np.uint16))
    with pytest.raises(CpxError):
        eng.track_batch(frames, np.array([0, 2, 2, 4], np.int32), meta)
    # a clip longer than the handle's max_frames
    long_frames = eng.upload_frames(np.full((40, 120, 160), 3000, np.uint16))
    with pytest.raises(CpxError):
        eng.track_batch(long_frames, np.array([0, 40], np.int32), eng.make_meta(40))
    # the handle is still usable
    eng.track_batch(frames, offs, meta).check()
    eng.close()


def test_track_capacity_overflow_is_reported():
    """More simultaneous objects than max_active_tracks: status CPX_ERR_OVERFLOW for that clip only."""
    from cpx._lib import CpxError
    from cpx.engine import TrackEngine
    from cpx.tracking import make_track_params

    T, H, W = 12, 120, 160
    rng = np.random.default_rng(4)
    base = rng.integers(2995, 3005, (T, H, W)).astype(np.uint16)
    busy = base.copy()
    for t in range(2, T):  # 12 warm squares appear and stay
        for k in range(12):
            y, x = 10 + 28 * (k // 4), 12 + 38 * (k % 4)
            busy[t, y:y + 8, x:x + 8] += 300
    quiet = base.copy()
    quiet[3:, 50:60, 70:80] += 300
    frames = np.concatenate([busy, quiet])
    eng = TrackEngine(model="lepton3", max_frames=T)
    offs = np.array([0, T, 2 * T], np.int32)
    meta = eng.make_meta(2 * T)
    res = eng.track_batch(eng.upload_frames(frames), offs, meta)
    res.check()
    small = make_track_params(max_active_tracks=4, max_tracks=8)
    assoc = eng.associate_batch(res, offs, meta, params=small)
    with pytest.raises(CpxError):
        assoc.check()
    status = assoc.status_dev.cpu().numpy()
    assert status[0] == -5 and status[1] == 0
    roomy = eng.associate_batch(res, offs, meta, params=make_track_params(max_active_tracks=16, max_tracks=64))
    roomy.check()
    assert len(roomy.clip_tracks(0)) >= 12 and len(roomy.clip_tracks(1)) >= 1
    eng.close()


def test_contour_chain_overflow_is_reported():
    """A border longer than the thumbnail kernel's chain capacity (8192 steps) is an error, not a short count."""
    import torch

    from cpx._lib import REGION_REF_DTYPE, CpxError
    from cpx.engine import TrackEngine

    H, W = 120, 160
    eng = TrackEngine(model="lepton3", max_frames=8)
    frames = np.full((1, H, W), 3000, np.uint16)
    dev = eng.upload_frames(frames)
    res = eng.track_batch(dev, np.array([0, 1], np.int32), eng.make_meta(1), want_labels=True)
    eng.synchronize()
    comb = np.zeros((H, W), np.int32)   # one connected comb: a spine plus a tooth in every other column
    comb[0, :] = 1
    comb[:, ::2] = 1
    res.labels_dev.copy_(torch.from_numpy(comb[None]).to(res.labels_dev.device))
    refs = np.zeros(1, REGION_REF_DTYPE)
    refs[0] = (0, 0, 0, W, H, 0)
    with pytest.raises(CpxError):
        eng.thumb_stats(dev, res, refs)
    # a modest shape on the same handle still works
    comb[:] = 0
    comb[10:30, 20:50] = 3
    res.labels_dev.copy_(torch.from_numpy(comb[None]).to(res.labels_dev.device))
    got = eng.thumb_stats(dev, res, refs)
    assert got[0]["contours"] == 4 and got[0]["status"] == 0
    eng.close()


def test_background_ownership_errors():
    """cpx_set_background / cpx_get_background / CPX_TRACK_KEEP_BACKGROUND refuse what they cannot honour: a
    non-integer background, a weight that is no accumulation of weight_add, a clip outside the batch, keeping a
    background that was never there -- and the handle stays usable."""
    from cpx import _lib
    from cpx._lib import CpxError
    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3", max_frames=16)
    frames = eng.upload_frames(np.full((6, 120, 160), 3000, np.uint16))
    offs = np.array([0, 6], np.int32)
    meta = eng.make_meta(6)
    # nothing to keep yet
    with pytest.raises(CpxError):
        eng.track_batch(frames, offs, meta, flags=_lib.TRACK_KEEP_BACKGROUND)
    with pytest.raises(CpxError):
        eng.get_background(0)
    bg = np.full((120, 160), 2990.5, np.float32)
    with pytest.raises(CpxError):
        eng.set_background(0, bg)                                  # not integer-valued
    bg[:] = 2990
    with pytest.raises(CpxError):
        eng.set_background(0, bg, weights=np.full((118, 158), 0.05))  # 0.05 is no multiple accumulation of 0.1
    eng.set_background(3, bg)                                      # staged for a clip the next batch does not have
    with pytest.raises(CpxError):
        eng.track_batch(frames, offs, meta, flags=_lib.TRACK_KEEP_BACKGROUND)
    # a proper hand-over: background 2990 under frames of 3000, weights = 3 accumulations
    w3 = np.full((118, 158), 0.1 + 0.1 + 0.1)
    eng.set_background(0, bg, weights=w3)
    res = eng.track_batch(frames, offs, meta, flags=_lib.TRACK_KEEP_BACKGROUND)
    res.check()
    got, w, avg = eng.get_background(0)
    assert got.shape == (120, 160) and w.shape == (118, 158) and np.isfinite(avg)
    assert (got[1:-1, 1:-1] >= 2990).all() and (got <= 3000).all()
    eng.close()
