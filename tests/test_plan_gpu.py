"""cpx_plan_segments (the device planner bench.py times) against the REFERENCE's get_segments:
the planner is the member of the reference's random family in which every draw is the identity
(helpers.IdentityDraws), and tests/golden/segments_identity_golden.json holds what the reference itself returned under
those draws for 104 seeded tracks (lengths either side of the 40-usable-frame switch and of the half / quarter segment
cut-offs, blanks, FFC frames, zero masses, all-zero tracks).  Required: the same frame lists in the same (sorted,
padded) tile order, the same segments dropped by the mass test, and in_segment = the frames a kept segment uses."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def test_plan_segments_equals_reference_under_identity_draws():
    import torch

    from cpx._lib import CROP_REQ_DTYPE, REGION_REF_DTYPE, CpxError
    from cpx.engine import TrackEngine
    from cpx.tracking import REGION_DTYPE, TRACK_SUMMARY_DTYPE, make_filter_params

    with open(os.path.join(GOLDEN, "segments_identity_golden.json")) as fh:
        cases = json.load(fh)["cases"]
    B = len(cases)
    ma, mt, per = 4, 2, 25
    lens = [c["start"] + len(c["regions"]) for c in cases]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    total = int(offs[-1])
    eng = TrackEngine(model="lepton3", max_frames=max(lens))
    t_on = np.array([100000 + 111 * i for i in range(max(lens))], dtype=np.int64)
    meta = eng.make_meta(total)
    pool = np.zeros((total, ma), REGION_DTYPE)
    summ = np.zeros((B, mt), TRACK_SUMMARY_DTYPE)
    counts = np.zeros((B, 4), np.int32)
    slot = 2  # any slot of the pool row
    for b, c in enumerate(cases):
        n, start, f0 = len(c["regions"]), c["start"], int(offs[b])
        m = meta[f0:f0 + lens[b]]
        m["time_on_ms"], m["last_ffc_ms"], m["has_times"] = t_on[:lens[b]], 40000, 1
        for f in c["ffc"]:   # FFC-affected frames: integer milliseconds less than 9 apart
            m["last_ffc_ms"][f] = m["time_on_ms"][f] - 5
        rows = pool[f0 + start:f0 + start + n, slot]
        for i, (mass, blank, w, h) in enumerate(c["regions"]):
            rows[i]["x"], rows[i]["y"], rows[i]["width"], rows[i]["height"] = 5 + (i % 7), 6 + (i % 5), w, h
            rows[i]["mass"], rows[i]["frame_number"], rows[i]["flags"] = mass, start + i, 1 if blank else 0
        s = summ[b, 0]
        s["id"], s["slot"], s["start_frame"], s["n_frames"], s["reject"], s["rank"] = b + 1, slot, start, n, 0, 0
        summ[b, 1]["rank"] = 1
        summ[b, 1]["reject"] = 2  # a rejected second record: never planned
        nonblank = sum(1 for (_, blank, w, h) in c["regions"] if not blank and w > 0 and h > 0)
        counts[b] = (1, nonblank, len(c["segments"]), 0)
    prefix = (np.cumsum(counts, axis=0) - counts).astype(np.int32)
    n_tracks, n_refs, n_samples = (int(v) for v in counts.sum(axis=0)[:3])
    dev = eng.device
    t = torch
    pool_d, summ_d = eng._to_dev(pool), eng._to_dev(summ)
    ntr_d = t.full((B,), 2, dtype=t.int32, device=dev)
    prefix_d = t.from_numpy(prefix).to(dev)
    refs_d = t.full((n_refs * 6,), -1, dtype=t.int32, device=dev)
    toffs_d = t.full((n_tracks + 1,), -1, dtype=t.int32, device=dev)
    reqs_d = t.full((n_samples * per * 8,), -1, dtype=t.int32, device=dev)
    st_d = t.full((n_samples,), -1, dtype=t.int32, device=dev)
    tc_d = t.full((n_tracks, 2), -1, dtype=t.int32, device=dev)
    fp = make_filter_params(max_active_tracks=ma, max_tracks_per_clip=mt)
    t.cuda.synchronize()
    rc = eng.lib.cpx_plan_segments(
        eng.h, C.byref(fp), offs.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(meta.ctypes.data), B,
        C.c_void_p(pool_d.data_ptr()), C.c_void_p(summ_d.data_ptr()), C.c_void_p(ntr_d.data_ptr()),
        C.c_void_p(prefix_d.data_ptr()), 5, C.c_void_p(refs_d.data_ptr()), C.c_void_p(toffs_d.data_ptr()),
        C.c_void_p(reqs_d.data_ptr()), C.c_void_p(st_d.data_ptr()), C.c_void_p(tc_d.data_ptr()))
    if rc != 0:
        raise CpxError(rc, eng._err())
    eng.synchronize()
    # the offsets the pipeline feeds the planner: cpx_counts_prefix of the counts = NumPy's exclusive prefix sums + totals
    counts_d = t.from_numpy(np.ascontiguousarray(counts, np.int32)).to(dev)
    pre_d = t.full((B + 1, 4), -1, dtype=t.int32, device=dev)
    assert eng.lib.cpx_counts_prefix(eng.h, C.c_void_p(counts_d.data_ptr()), B, C.c_void_p(pre_d.data_ptr())) == 0
    eng.synchronize()
    pre = pre_d.cpu().numpy()
    assert np.array_equal(pre[:B], prefix) and np.array_equal(pre[B], counts.sum(axis=0))
    reqs = reqs_d.cpu().numpy().view(CROP_REQ_DTYPE).reshape(n_samples, per)
    refs = refs_d.cpu().numpy().view(REGION_REF_DTYPE)
    st, tc, toffs = st_d.cpu().numpy(), tc_d.cpu().numpy(), toffs_d.cpu().numpy()
    assert int(toffs[n_tracks]) == n_refs          # the closing offset is the planner's (it was a torch index write)
    assert (reqs["frame"] >= 0).all() and (st >= 0).all(), "the planner produced fewer segments than the reference"
    n_checked = 0
    for b, c in enumerate(cases):
        f0, start = int(offs[b]), c["start"]
        assert tuple(tc[b]) == (b, b + 1)
        s0 = int(prefix[b, 2])
        want = c["segments"]
        assert (st[s0:s0 + len(want)] == b).all(), b
        for k, frames in enumerate(want):
            q = reqs[s0 + k]
            assert [int(f) - f0 for f in q["frame"]] == frames, (b, k)    # frame lists, padding order = tile order
            assert list(q["tile"]) == list(range(per)) and (q["sample"] == s0 + k).all() and (q["track"] == b).all()
            for j, f in enumerate(frames):
                assert (q["width"][j], q["height"][j]) == tuple(c["regions"][f - start][2:4])
            n_checked += 1
        # the track's refs: every non-blank region, in_segment = used by a kept segment
        r0 = int(toffs[b])
        mine = refs[r0:r0 + int(counts[b, 1])]
        used = set(f for frames in want for f in frames)
        nb = [start + i for i, (_, blank, w, h) in enumerate(c["regions"]) if not blank and w > 0 and h > 0]
        assert [int(f) - f0 for f in mine["frame"]] == nb, b
        assert [int(v) for v in mine["in_segment"]] == [1 if f in used else 0 for f in nb], b
    assert n_checked == n_samples >= 200
    eng.close()
