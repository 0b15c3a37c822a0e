"""The association logic (classifier-pipeline_amd/csrc/cpx_assoc_core.h, the code the
GPU runs one lane per clip) compiled for the HOST and checked against the
oracle's tracker -- debugging aid that needs no GPU.  The product never loads
this build."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import load_clip

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("assoc") / "libassoc_host.so"
    subprocess.check_call([
        "g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
        "-I", os.path.join(REPO, "include"), "-I", os.path.join(REPO, "classifier-pipeline_amd", "csrc"),
        os.path.join(REPO, "tests", "native", "assoc_host.cpp"), "-o", str(out)])
    return C.CDLL(str(out))


def run_host(host_lib, out, cap=64, max_active=16, max_tracks=128):
    """Feed the oracle's per-frame components to the host build; returns tracks + region lists."""
    from cpx._lib import COMPONENT_DTYPE
    from cpx.tracking import REGION_DTYPE, TRACK_RECORD_DTYPE, make_track_params, track_regions
    import track_oracle as to

    fr = out["frames"]
    n = len(fr)
    comps = np.zeros((n, cap), COMPONENT_DTYPE)
    ncomp = np.zeros(n, np.int32)
    ffc = np.array([f["ffc"] for f in fr], np.int32)
    prev = None
    for t, f in enumerate(fr):
        k = f["n_components"]
        ncomp[t] = k
        st = f["stats"]
        comps["x"][t, :k], comps["y"][t, :k] = st[:, 0], st[:, 1]
        comps["width"][t, :k], comps["height"][t, :k], comps["area"][t, :k] = st[:, 2], st[:, 3], st[:, 4]
        if k:
            lab = f["mask"]
            ys, xs = np.nonzero(lab)
            comps["sum_x"][t, :k] = np.bincount(lab[ys, xs] - 1, weights=xs, minlength=k)
            comps["sum_y"][t, :k] = np.bincount(lab[ys, xs] - 1, weights=ys, minlength=k)
            if prev is not None:
                delta = to.delta_frame(f["filtered"].astype(np.float64), prev.astype(np.float64))
                for i in range(k):
                    x, y, w, h = st[i, :4]
                    comps["pixel_variance"][t, i] = np.var(delta[y:y + h, x:x + w])
        prev = f["filtered"]
    params = make_track_params(max_active_tracks=max_active, max_tracks=max_tracks)
    pool = np.zeros(n * max_active, REGION_DTYPE)
    tracks = np.zeros(max_tracks, TRACK_RECORD_DTYPE)
    ntr = C.c_int(0)
    regions = np.zeros((n, cap), REGION_DTYPE)
    rcount = np.zeros(n, np.int32)
    rc = host_lib.assoc_host_clip(C.byref(params), cap, n, ffc.ctypes.data_as(C.c_void_p),
                                  ncomp.ctypes.data_as(C.c_void_p), comps.ctypes.data_as(C.c_void_p),
                                  pool.ctypes.data_as(C.c_void_p), tracks.ctypes.data_as(C.c_void_p),
                                  C.byref(ntr), regions.ctypes.data_as(C.c_void_p),
                                  rcount.ctypes.data_as(C.c_void_p))
    assert rc == 0
    tr = [(tracks[i], track_regions(pool, tracks[i], max_active)) for i in range(ntr.value)]
    return tr, regions, rcount


def compare_tracks(tr, regions, rcount, out):
    from cpx.tracking import REGION_BLANK, REGION_BORDER, REGION_CROPPED

    # region lists per frame
    for t, regs in enumerate(out["region_history"]):
        assert rcount[t] == len(regs), t
        for i, r in enumerate(regs):
            g = regions[t, i]
            assert (g["x"], g["y"], g["width"], g["height"], g["mass"], g["id"]) == (
                r.x, r.y, r.width, r.height, int(r.mass), r.id), (t, i)
            assert bool(g["flags"] & REGION_CROPPED) == r.was_cropped
            assert bool(g["flags"] & REGION_BORDER) == r.is_along_border
            assert g["cx"] == float(r.centroid[0]) and g["cy"] == float(r.centroid[1])
    # every track ever created, untrimmed
    want = sorted(out["tracks"] + [t for _, t in out.get("filtered_tracks", [])], key=lambda t: t.id)
    assert len(tr) == len(want)
    for (rec, regs), w in zip(tr, want):
        assert (rec["id"], rec["start_frame"], rec["n_frames"]) == (w.id, w.start_frame, len(w.bounds)), w.id
        assert (rec["blank_frames"], rec["since_seen"], rec["rt_frames"]) == (w.blank_frames, w.since_seen, w.rt_frames)
        for g, r in zip(regs, w.bounds):
            assert (g["x"], g["y"], g["width"], g["height"], g["mass"], g["frame_number"]) == (
                r.x, r.y, r.width, r.height, int(r.mass), r.frame_number), (w.id, r.frame_number)
            assert bool(g["flags"] & REGION_BLANK) == r.blank
            assert g["cx"] == float(r.centroid[0]) and g["cy"] == float(r.centroid[1]), (w.id, r.frame_number)
            assert g["pixel_variance"] == np.float32(r.pixel_variance)


@pytest.mark.parametrize("name", ["possum", "hedgehog"])
def test_host_assoc_fixture(host_lib, name):
    import track_oracle as to

    frames, t_on, ffc, bgf, hdr = load_clip(name)
    out = to.track_clip(frames, t_on, ffc, bgf, to.OracleConfig(hdr.model), keep=True, apply_filter=False)
    tr, regions, rcount = run_host(host_lib, out)
    compare_tracks(tr, regions, rcount, out)


def test_host_assoc_synthetic(host_lib):
    import track_oracle as to
    from cpx import synth
    from cpx.tracking import REGION_BLANK, REGION_CENTROID_F32

    n_tracks = n_blank = n_kalman = 0
    for seed in range(1, 11):
        rng = np.random.default_rng(seed)
        clip = synth.make_clip(rng, 150, max_blobs=3)
        out = to.track_clip(clip, cfg=to.OracleConfig("lepton3"), keep=True, apply_filter=False)
        tr, regions, rcount = run_host(host_lib, out)
        compare_tracks(tr, regions, rcount, out)
        n_tracks += len(tr)
        for _, regs in tr:
            n_blank += int(np.sum((regs["flags"] & REGION_BLANK) != 0))
            n_kalman += int(np.sum((regs["flags"] & REGION_CENTROID_F32) != 0))
    # the comparison must have exercised tracks, blank frames and Kalman-predicted regions
    assert n_tracks > 10 and n_blank > 10 and n_kalman > 0, (n_tracks, n_blank, n_kalman)


# ---------------------------------------------------------------------------
# end-of-clip code (trim / statistics / rejects / segment plan), host build vs the oracle
# ---------------------------------------------------------------------------
def run_final(host_lib, tr_raw, pool, n_frames, ffc, max_active=16, max_tracks=128):
    from cpx._lib import CROP_REQ_DTYPE, REGION_REF_DTYPE
    from cpx.tracking import TRACK_RECORD_DTYPE, TRACK_SUMMARY_DTYPE, make_filter_params

    fp = make_filter_params(max_active_tracks=max_active, max_tracks_per_clip=max_tracks)
    recs = np.array([r for r, _ in tr_raw], dtype=TRACK_RECORD_DTYPE)
    n = len(recs)
    out = np.zeros(max(n, 1), TRACK_SUMMARY_DTYPE)
    counts = np.zeros(4, np.int32)
    refs = np.zeros(n_frames * max_active + 1, REGION_REF_DTYPE)
    toffs = np.zeros(max_tracks + 1, np.int32)
    reqs = np.zeros((n_frames + 25 * max_tracks) * 2, CROP_REQ_DTYPE)
    st = np.zeros(n_frames + max_tracks, np.int32)
    tc = np.zeros(2 * max_tracks, np.int32)
    pidx = np.arange(n_frames, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    host_lib.final_host_clip(C.byref(fp), p(pool), p(recs), n, p(np.asarray(ffc, np.int32)), p(pidx), n_frames,
                             p(out), p(counts), p(refs), p(toffs), p(reqs), p(st), p(tc))
    return out[:n], counts, refs, toffs, reqs, st, tc


def host_pool(host_lib, out, cap=64, max_active=16, max_tracks=128):
    """re-run the association keeping the raw pool (run_host slices it per track)."""
    from cpx.tracking import REGION_DTYPE, TRACK_RECORD_DTYPE

    tr, regions, rcount = run_host(host_lib, out, cap, max_active, max_tracks)
    n = len(out["frames"])
    pool = np.zeros(n * max_active, REGION_DTYPE)
    for rec, regs in tr:
        idx = (int(rec["start_frame"]) + np.arange(int(rec["n_frames"]))) * max_active + int(rec["slot"])
        pool[idx] = regs
    return tr, pool


REASON = {"Track filtered.  Too short": 1, "Track filtered.  Didn't move": 2, "Track filtered. Too Many Blanks": 3,
          "Track filtered.  Too Jittery": 4, "Track filtered.  Too static": 5, "Track filtered.  Too Dynamic": 6,
          "Track filtered.  Mass too small": 7}


def compare_final(summ, out_filtered):
    """summaries (creation order) vs the oracle after filter_tracks."""
    kept = {t.id: t for t in out_filtered["tracks"]}
    rej = {t.id: (reason, t) for reason, t in out_filtered["filtered_tracks"]}
    order = [t.id for t in sorted(list(kept.values()) + [t for _, t in rej.values()],
                                  key=lambda t: -t.stats["score"] if not np.isnan(t.stats["score"]) else 0)]
    for s in summ:
        tid = int(s["id"])
        t = kept.get(tid) or rej[tid][1]
        assert (s["start_frame"], s["n_frames"]) == (t.start_frame, len(t.bounds)), tid
        assert (s["blank_frames"], s["since_seen"]) == (t.blank_frames, t.since_seen), tid
        st = t.stats
        for k in ("frames_moved", "region_jitter", "jitter_bigger", "jitter_smaller", "blank_percent"):
            assert s[k] == st[k], (tid, k)
        for k in ("movement", "max_offset", "score", "average_mass", "median_mass", "delta_std", "mass_std",
                  "average_velocity"):
            a, b = float(s[k]), float(st[k])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-12 * max(1.0, abs(b)), (tid, k, a, b)
        assert s["reject"] == (0 if tid in kept else REASON[rej[tid][0]]), tid
    # kept tracks come out in the oracle's order
    kept_ranked = [int(s["id"]) for s in sorted(summ, key=lambda s: s["rank"]) if s["reject"] == 0]
    assert kept_ranked == [t.id for t in out_filtered["tracks"]]


@pytest.mark.parametrize("name", ["possum", "hedgehog"])
def test_host_final_fixture(host_lib, name):
    import track_oracle as to

    frames, t_on, ffc, bgf, hdr = load_clip(name)
    raw = to.track_clip(frames, t_on, ffc, bgf, to.OracleConfig(hdr.model), keep=True, apply_filter=False)
    tr, pool = host_pool(host_lib, raw)
    fin = to.track_clip(frames, t_on, ffc, bgf, to.OracleConfig(hdr.model), keep=True, apply_filter=True)
    n = len(raw["frames"])
    summ, counts, refs, toffs, reqs, st, tc = run_final(host_lib, tr, pool, n, [f["ffc"] for f in raw["frames"]])
    compare_final(summ, fin)
    assert counts[0] == len(fin["tracks"])
    # plan: every sample has 25 sorted tiles drawn from the track's usable frames; every usable frame
    # of the first covered*25 is used; refs cover all non-blank regions
    nref = 0
    for ti, t in enumerate(fin["tracks"]):
        nonblank = [r for r in t.bounds if not r.blank and r.width > 0 and r.height > 0]
        assert toffs[ti + 1] - toffs[ti] == len(nonblank)
        nref += len(nonblank)
        usable = [r.frame_number for r in nonblank if r.mass > 0]
        nseg = max(1, (len(usable) + 12) // 25) if usable else 0
        mine = [s for s in range(counts[2]) if st[s] == ti]
        assert len(mine) == nseg
        covered = usable[: 25 * nseg]
        used = []
        for s in mine:
            q = reqs[s * 25:(s + 1) * 25]
            assert list(q["tile"]) == list(range(25)) and np.all(q["sample"] == s) and np.all(q["track"] == ti)
            fr = list(q["frame"])
            assert fr == sorted(fr)
            used += fr
        assert sorted(set(used)) == covered
    assert counts[1] == nref


def test_host_final_synthetic(host_lib):
    import track_oracle as to
    from cpx import synth

    kept = rejected = 0
    for seed in range(1, 9):
        rng = np.random.default_rng(100 + seed)
        clip = synth.make_clip(rng, 150, max_blobs=3)
        raw = to.track_clip(clip, cfg=to.OracleConfig("lepton3"), keep=True, apply_filter=False)
        tr, pool = host_pool(host_lib, raw)
        fin = to.track_clip(clip, cfg=to.OracleConfig("lepton3"), keep=True, apply_filter=True)
        summ, counts, *_ = run_final(host_lib, tr, pool, 150, [0] * 150)
        compare_final(summ, fin)
        kept += len(fin["tracks"])
        rejected += len(fin["filtered_tracks"])
    assert kept > 5  # (rejects are exercised by the possum fixture: five "Didn't move" tracks)


def test_median_select_equals_numpy(host_lib):
    """np_median_select (quickselect, used by the finalize kernel) against np.median on adversarial inputs."""
    rng = np.random.default_rng(0)
    fn = host_lib.median_select_host
    fn.restype = C.c_double
    fn.argtypes = [C.POINTER(C.c_double), C.c_int]
    cases = [np.array([5.0]), np.array([2.0, 1.0]), np.array([3.0, 3.0, 3.0, 3.0]), np.arange(10, 0, -1.0),
             np.arange(9.0), np.array([1.0, 2.0, 2.0, 2.0, 9.0, 9.0])]
    for n in (3, 4, 7, 8, 31, 32, 100, 269, 270):
        cases.append(rng.integers(0, 5, n).astype(np.float64))        # many duplicates
        cases.append(rng.integers(0, 100000, n).astype(np.float64))
        cases.append(np.sort(rng.integers(0, 1000, n)).astype(np.float64))
        cases.append(np.sort(rng.integers(0, 1000, n))[::-1].astype(np.float64))
    for a in cases:
        buf = np.ascontiguousarray(a.copy())
        got = fn(buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size)
        assert got == float(np.median(a)), (a[:10], got)
        assert sorted(buf.tolist()) == sorted(a.tolist())  # a permutation of the input
