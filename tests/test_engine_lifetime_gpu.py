"""Result lifetimes are enforced by the allocator, not by a warning.  TrackEngine allocates the outputs
of track_batch / associate_batch / ir_detect / ir_resize_area under its own HIP stream: a result dropped while its kernels
run gives its blocks back to THAT stream's pool, so an allocation on torch's default stream can never be handed memory a
kernel is still writing, and a later call on the engine is ordered behind the kernels.  A result also keeps its engine's
stream alive past TrackEngine.close()."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(eng, n_clips=192, frames=120, seed=9, base=12):
    """`n_clips` clips of `frames` frames on the device: `base` synthetic clips repeated (the generator is host code)."""
    from cpx import synth

    base = min(base, n_clips)
    fr, _ = synth.make_batch(base, frames, seed=seed)
    dev = eng.upload_frames(fr).repeat((n_clips + base - 1) // base, 1, 1)[: n_clips * frames].contiguous()
    offs = (np.arange(n_clips + 1, dtype=np.int64) * frames).astype(np.int32)
    return dev, offs, eng.make_meta(n_clips * frames)


def test_dropped_result_does_not_leak_into_other_streams_allocations():
    import torch

    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3", device=0, max_frames=256)
    frames, offs, meta = _batch(eng)
    total = int(offs[-1])
    held = eng.track_batch(frames, offs, meta, want_labels=True, want_filtered=True)
    want_f, want_l, want_info = held.filtered(), held.labels(), held.info.copy()   # (synchronises)
    shape_f = tuple(held.filtered_dev.shape)
    for trial in range(4):
        res = eng.track_batch(frames, offs, meta, want_labels=True, want_filtered=True)   # in flight: tens of milliseconds
        del res                                                                            # dropped at once
        gc.collect()
        # an allocation-heavy co-runner on the default stream: tensors of exactly the sizes just freed, filled with a
        # pattern.  Handed the dropped result's blocks, they would be written by the track kernels still running.
        junk_f = [torch.full(shape_f, 7.0, dtype=torch.float32, device=eng.device) for _ in range(3)]
        junk_l = [torch.full(shape_f, -3, dtype=torch.int32, device=eng.device) for _ in range(3)]
        junk_c = [torch.full((total * eng.cap * 8,), -5, dtype=torch.int32, device=eng.device) for _ in range(2)]
        again = eng.track_batch(frames, offs, meta, want_labels=True, want_filtered=True)
        torch.cuda.synchronize()
        eng.synchronize()
        assert all(bool((j == 7.0).all()) for j in junk_f), trial
        assert all(bool((j == -3).all()) for j in junk_l), trial
        assert all(bool((j == -5).all()) for j in junk_c), trial
        assert np.array_equal(again.filtered(), want_f) and np.array_equal(again.labels(), want_l), trial
        assert np.array_equal(again.info, want_info), trial
        del junk_f, junk_l, junk_c, again
    # the association's outputs follow the same rule
    a_held = eng.associate_batch(held, offs, meta)
    tracks_want = [a_held.clip_tracks(b) for b in range(4)]
    tmp = eng.associate_batch(held, offs, meta)
    del tmp
    junk = [torch.full((total * 16 * 14,), 11, dtype=torch.int32, device=eng.device) for _ in range(3)]
    a2 = eng.associate_batch(held, offs, meta)
    torch.cuda.synchronize()
    eng.synchronize()
    assert all(bool((j == 11).all()) for j in junk)
    for b in range(4):
        got = a2.clip_tracks(b)
        assert len(got) == len(tracks_want[b])
        for (r0, g0), (r1, g1) in zip(got, tracks_want[b]):
            assert r0.tobytes() == r1.tobytes() and g0.tobytes() == g1.tobytes()
    eng.close()


def test_outputs_belong_to_the_engines_stream_pool():
    """The blocks are the stream's: what a dropped result frees is what the engine's next call of the same shape gets, and
    never what an equal-sized allocation on the default stream gets."""
    import torch

    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3", device=0, max_frames=64)
    frames, offs, meta = _batch(eng, n_clips=8, frames=40)
    r = eng.track_batch(frames, offs, meta, want_filtered=True)
    ptr, shape = r.filtered_dev.data_ptr(), tuple(r.filtered_dev.shape)
    del r
    other = torch.empty(shape, dtype=torch.float32, device=eng.device)      # default stream
    assert other.data_ptr() != ptr
    r2 = eng.track_batch(frames, offs, meta, want_filtered=True)
    assert r2.filtered_dev.data_ptr() == ptr
    r2.check()
    eng.close()


def test_close_waits_for_live_results():
    """A grown / sibling engine closed while a result of it is referenced keeps its handle (and stream) until the result
    dies: the result stays readable, its memory goes back to a living stream, the handle is destroyed afterwards."""
    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3", device=0, max_frames=64)
    sib = eng.sibling(128)
    frames, offs, meta = _batch(sib, n_clips=4, frames=30)
    res = sib.track_batch(frames, offs, meta, want_filtered=True)
    ref = eng.track_batch(frames, offs, meta, want_filtered=True).filtered()
    sib.close()
    assert sib.h and sib._close_deferred          # deferred: a result is alive
    assert np.array_equal(res.filtered(), ref)
    del res
    gc.collect()
    assert not sib.h and not sib._close_deferred  # destroyed when the last result died
    sib.close()                                   # idempotent
    eng.close()
    assert not eng.h


def test_deferred_medians_are_complete_for_their_readers():
    """CPX_TRACK_DEFER_MEDIANS (include/cpx.h): the medians run on the handle's second stream behind the frame kernel.  They
    must be what the in-front form computes -- read after engine.synchronize(), by a stream-ordered reader after
    cpx_join_medians, and by the limits kernel (which subtracts them) -- and a following track call must not overtake them."""
    import ctypes as C

    import torch

    from cpx._lib import TRACK_DEFER_MEDIANS
    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3", device=0, max_frames=128)
    frames, offs, meta = _batch(eng, n_clips=256, frames=90)
    want = eng.track_batch(frames, offs, meta, want_filtered=True).info["thermal_median"].copy()
    assert np.all(want > 0)
    for trial in range(3):
        res = eng.track_batch(frames, offs, meta, want_filtered=True, flags=TRACK_DEFER_MEDIANS)
        assert np.array_equal(res.info["thermal_median"], want), trial          # (host accessor: engine.synchronize())
        # a stream-ordered device reader: join, then a copy on the engine's stream
        res = eng.track_batch(frames, offs, meta, want_filtered=True, flags=TRACK_DEFER_MEDIANS)
        assert eng.lib.cpx_join_medians(eng.h) == 0
        with torch.cuda.stream(eng.torch_stream()):
            med = res.info_dev.view(-1, 20)[:, 13].clone()
        eng.torch_stream().synchronize()
        assert np.array_equal(med.cpu().numpy().view(np.float32), want), trial
        # two deferred calls back to back into the same buffers: the second call's memset of the records waits for the first
        # call's medians, and its own medians are the ones read
        outputs = (res.comps_dev, res.info_dev, None, res.filtered_dev, None)
        eng.track_batch(frames, offs, meta, outputs=outputs, flags=TRACK_DEFER_MEDIANS)
        again = eng.track_batch(frames, offs, meta, outputs=outputs, flags=TRACK_DEFER_MEDIANS)
        assert np.array_equal(again.info["thermal_median"], want), trial
    eng.close()


def test_deferred_close_gives_the_handles_own_memory_back():
    """An engine closed while a result of it lives keeps its stream, not its memory: the workspaces (and the network's
    activation arena -- 41 GB at the bench's chunk) are released at close(); the bench's file-fed leg ran out of memory beside
    three such lingering engines before this was so."""
    import torch

    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3", device=0, max_frames=128)
    frames, offs, meta = _batch(eng, n_clips=512, frames=100)
    res = eng.track_batch(frames, offs, meta, want_filtered=True)
    want = res.filtered()[:3].copy()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    eng.close()
    assert eng.h and eng._close_deferred
    free1, _ = torch.cuda.mem_get_info()
    assert free1 - free0 >= 512 * 160 * 120 * 8, (free0, free1)   # at least the clips' background / window state
    assert np.array_equal(res.filtered()[:3], want)                 # the result is still readable
    del res
    gc.collect()
    assert not eng.h
