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
