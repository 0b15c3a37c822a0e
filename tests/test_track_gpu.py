"""GPU parity tests of the track stage: HIP kernels (through the C-ABI) against
(1) the vectors the reference itself produced for its fixture clips and
(2) the oracle restatement on seeded synthetic clips.  Integer / label / index
work is compared bit for bit; np.var (float32 in the reference) to 1e-4 rel."""
import numpy as np
import pytest

from helpers import crc, load_clip, load_golden

pytestmark = pytest.mark.gpu

VAR_RTOL, VAR_ATOL = 1e-4, 1e-3


@pytest.fixture(scope="module")
def engines():
    from cpx.engine import TrackEngine

    made = {}

    def get(model, **kw):
        key = (model, tuple(sorted(kw.items())))
        if key not in made:
            made[key] = TrackEngine(model=model or "lepton3", **kw)
        return made[key]

    yield get
    for e in made.values():
        e.close()


def _run_fixture(engine, name, lengths=None):
    frames, t_on, ffc, bgf, hdr = load_clip(name)
    n = frames.shape[0]
    lengths = lengths or [n]
    offs = [0]
    chunks, metas = [], []
    for L in lengths:
        chunks.append(frames[:L])
        metas.append(engine.make_meta(L, t_on[:L], ffc[:L], bgf[:L]))
        offs.append(offs[-1] + L)
    dev = engine.upload_frames(np.concatenate(chunks))
    res = engine.track_batch(dev, np.array(offs, np.int32), np.concatenate(metas), want_labels=True,
                             want_filtered=True, want_background=True)
    res.check()
    return res, offs, bgf


@pytest.mark.parametrize("name", ["possum", "hedgehog", "synth35"])
def test_fixture_clip_matches_reference_vectors(engines, name):
    z, _ = load_golden(name, 0)
    kept = [int(k) for k in z["kept"]]
    frames, t_on, ffc, bgf, hdr = load_clip(name)
    nb = int(np.sum(bgf))  # background frames precede the processed ones in the fixtures
    # full clip + truncated copies: final background of each copy == reference background after that frame
    lengths = [frames.shape[0]] + [k + 1 + nb for k in kept]
    eng = engines(hdr.model)
    res, offs, bgf = _run_fixture(eng, name, lengths)
    info, labels, filt, bgs = res.info, res.labels(), res.filtered(), res.background()
    proc = [i for i in range(lengths[0]) if not bgf[i]]
    assert len(proc) == int(z["n_frames"])
    for q, f in enumerate(proc):
        fi = info[f]
        assert fi["frame_number"] == q
        assert bool(fi["ffc_affected"]) == bool(z["ffc"][q])
        assert float(fi["threshold"]) == float(np.float32(z["threshold"][q])), q
        assert fi["background_average"] == z["bg_after_avg"][q], q
        assert crc(filt[f].astype(np.int32)) == z["crc_filtered"][q], q
        assert crc(labels[f]) == z["crc_mask"][q], q
        a, b = z["comp_offsets"][q], z["comp_offsets"][q + 1]
        assert fi["n_components"] == b - a, q
        c = res.components(f)
        got = np.stack([c["x"], c["y"], c["width"], c["height"], c["area"]], axis=1).reshape(-1, 5)
        assert np.array_equal(got, z["comp_stats"][a:b]), q
        cent = np.stack([c["sum_x"] / c["area"], c["sum_y"] / c["area"]], axis=1).reshape(-1, 2)
        assert np.array_equal(cent, z["comp_centroids"][a:b]), q
    for f in range(lengths[0]):
        if bgf[f]:
            assert info[f]["frame_number"] == -1
    # pixel_variance of the regions the reference kept (regions carry the component id)
    for q, f in enumerate(proc):
        a, b = z["region_offsets"][q], z["region_offsets"][q + 1]
        c = res.components(f)
        for r in z["regions"][a:b]:
            cid = int(r[5])
            assert (int(c["area"][cid]) == int(r[4]))
            np.testing.assert_allclose(c["pixel_variance"][cid], r[6], rtol=VAR_RTOL, atol=VAR_ATOL)
    # background after frame k (truncated copies) and after the last frame
    for j, k in enumerate(kept):
        assert np.array_equal(bgs[j + 1].astype(np.int32), z["kept_bg_after"][j]), k


def _oracle_clip(frames, model, t_on=None, ffc=None, bgf=None):
    import track_oracle as to

    cfg = to.OracleConfig(model)
    return to.track_clip(frames, t_on, ffc, bgf, cfg, keep=True, do_tracking=False), cfg


def _compare_with_oracle(res, f0, out, frames_of_clip, bgf=None):
    info, labels, filt = res.info, res.labels(), res.filtered()
    q = 0
    import track_oracle as to

    prev = None
    for i in range(frames_of_clip):
        f = f0 + i
        if bgf is not None and bgf[i]:
            assert info[f]["frame_number"] == -1
            continue
        o = out["frames"][q]
        fi = info[f]
        assert fi["frame_number"] == q
        assert fi["avg_change"] == o["avg_change"], (i, q)
        assert float(fi["threshold"]) == float(np.float32(o["threshold"])), (i, q)
        assert fi["norm_min"] == int(o["norm_min"]) and fi["norm_max"] == int(o["norm_max"])
        assert fi["background_average"] == o["bg_after_avg"], (i, q)
        assert np.array_equal(filt[f].astype(np.int32), o["filtered"]), (i, q)
        assert np.array_equal(labels[f], o["mask"]), (i, q)
        assert fi["n_components"] == o["n_components"]
        c = res.components(f)
        got = np.stack([c["x"], c["y"], c["width"], c["height"], c["area"]], axis=1).reshape(-1, 5)
        assert np.array_equal(got, o["stats"]), (i, q)
        if o["n_components"]:
            cent = np.stack([c["sum_x"] / c["area"], c["sum_y"] / c["area"]], axis=1)
            assert np.array_equal(cent, o["centroids"]), (i, q)
        # np.var of the delta frame over every component's bounding box
        if prev is not None and o["n_components"]:
            delta = to.delta_frame(o["filtered"].astype(np.float64), prev.astype(np.float64))
            for k in range(o["n_components"]):
                x, y, w, h, _ = o["stats"][k]
                want = np.var(delta[y:y + h, x:x + w])
                np.testing.assert_allclose(c["pixel_variance"][k], want, rtol=VAR_RTOL, atol=VAR_ATOL)
        elif o["n_components"]:
            assert np.all(c["pixel_variance"] == 0)
        prev = o["filtered"]
        # ClipStats inputs
        th = out["thermal"][q] if "thermal" in out else None
        q += 1
    return q


@pytest.mark.parametrize("model", ["lepton3", "lepton3.5"])
def test_synthetic_batch_matches_oracle(engines, model):
    from cpx import synth

    n_clips, T = 6, 64
    frames, offs = synth.make_batch(n_clips, T, seed=1234, model=model)
    eng = engines(model)
    t_on, ffc = synth.frame_times(T)
    meta = np.concatenate([eng.make_meta(T, t_on, ffc) for _ in range(n_clips)])
    res = eng.track_batch(eng.upload_frames(frames), offs, meta, want_labels=True, want_filtered=True,
                          want_background=True)
    res.check()
    bgs = res.background()
    ncomp = 0
    for b in range(n_clips):
        clip = frames[offs[b]:offs[b + 1]]
        out, _ = _oracle_clip(clip, model, t_on, ffc)
        _compare_with_oracle(res, int(offs[b]), out, T)
        assert np.array_equal(bgs[b].astype(np.int32), out["frames"][-1]["bg_after"])
        ncomp += sum(f["n_components"] for f in out["frames"])
        # thermal statistics (ClipStats.add_frame, clip.py:474-487)
        for i in range(T):
            fi = res.info[offs[b] + i]
            assert fi["thermal_min"] == clip[i].min() and fi["thermal_max"] == clip[i].max()
            assert fi["thermal_sum"] == int(clip[i].astype(np.int64).sum())
            assert float(fi["thermal_median"]) == float(np.median(clip[i]))
            assert fi["filtered_abs_sum"] == int(np.abs(out["frames"][i]["filtered"].astype(np.int64)).sum())
    assert ncomp > 0  # the comparison must have seen objects


def test_ragged_batch_background_frames_and_ffc(engines):
    """Clips of different lengths (1, 3, 50, 97 > window), background frames in
    the stream, FFC-flagged frames: same results as the oracle per clip."""
    from cpx import synth

    rng = np.random.default_rng(7)
    lens = [1, 3, 50, 97]
    eng = engines("lepton3")
    clips, metas, outs, offs = [], [], [], [0]
    for n in lens:
        clip = synth.make_clip(rng, n, model="lepton3")
        t_on, ffc = synth.frame_times(n)
        bgf = [False] * n
        if n >= 3:
            bgf[0] = True  # leading background frame (as in possum.cptv)
        if n >= 50:
            bgf[10] = True  # and one in the middle of the stream
            ffc = list(ffc)
            ffc[20] = t_on[20] - 5  # frames 20: (time_on - last_ffc) < 9 -> ffc affected
        clips.append(clip)
        metas.append(eng.make_meta(n, t_on, ffc, bgf))
        outs.append((_oracle_clip(clip, "lepton3", t_on, ffc, bgf)[0], bgf))
        offs.append(offs[-1] + n)
    res = eng.track_batch(eng.upload_frames(np.concatenate(clips)), np.array(offs, np.int32),
                          np.concatenate(metas), want_labels=True, want_filtered=True, want_background=True)
    res.check()
    for b, n in enumerate(lens):
        out, bgf = outs[b]
        nq = _compare_with_oracle(res, offs[b], out, n, bgf)
        assert nq == len(out["frames"])
        if nq:
            assert np.array_equal(res.background()[b].astype(np.int32), out["frames"][-1]["bg_after"])
        flags = [bool(res.info[offs[b] + i]["ffc_affected"]) for i in range(n) if not bgf[i]]
        assert flags == [f["ffc"] for f in out["frames"]]


def test_degenerate_frames(engines):
    """Constant frames (max == min paths of normalize), a saturated frame and a
    checkerboard that overflows the component capacity (error, never truncation)."""
    from cpx._lib import CpxError

    eng = engines("lepton3")
    H, W = 120, 160
    const = np.full((4, H, W), 3000, np.uint16)
    const[2] += 7  # uniform jump: x - avg_change == 0 everywhere
    const[3, 40:60, 50:90] = 65535  # saturated block
    out, _ = _oracle_clip(const, "lepton3")
    res = eng.track_batch(eng.upload_frames(const), np.array([0, 4], np.int32), eng.make_meta(4),
                          want_labels=True, want_filtered=True)
    res.check()
    _compare_with_oracle(res, 0, out, 4)
    for i in range(4):
        assert float(res.info[i]["thermal_median"]) == float(np.median(const[i]))
    # median with an odd split: half the pixels at 100, half at 201 -> 150.5; and 65535-saturated frames
    md = np.full((3, H, W), 100, np.uint16)
    md[0].reshape(-1)[H * W // 2:] = 201
    md[1][:] = 65535
    md[2].reshape(-1)[: H * W // 2 - 1] = 7
    md[2].reshape(-1)[H * W // 2 - 1:] = 65535
    r2 = eng.track_batch(eng.upload_frames(md), np.array([0, 3], np.int32), eng.make_meta(3))
    for i in range(3):
        assert float(r2.info[i]["thermal_median"]) == float(np.median(md[i])), i
    # 2x2-block checkerboard of hot pixels far above threshold -> thousands of components
    cb = np.full((2, H, W), 3000, np.uint16)
    cb[1, 4:116:6, 4:156:6] = 9000
    res = eng.track_batch(eng.upload_frames(cb), np.array([0, 2], np.int32), eng.make_meta(2), want_labels=True)
    n = int(res.info[1]["n_components"])
    out, _ = _oracle_clip(cb, "lepton3")
    assert n == out["frames"][1]["n_components"]
    if n > eng.cap:
        with pytest.raises(CpxError):
            res.check()


def test_no_optional_outputs_same_components(engines):
    """labels / filtered outputs off (state ping-pong path) gives identical components."""
    from cpx import synth

    frames, offs = synth.make_batch(3, 40, seed=99)
    eng = engines("lepton3")
    meta = eng.make_meta(frames.shape[0])
    dev = eng.upload_frames(frames)
    full = eng.track_batch(dev, offs, meta, want_labels=True, want_filtered=True)
    lean = eng.track_batch(dev, offs, meta)
    assert np.array_equal(full.info, lean.info)
    for f in range(frames.shape[0]):
        assert np.array_equal(full.components(f), lean.components(f))


# ---------------------------------------------------------------------------
# association stage (one GPU lane per clip) against the oracle's tracker
# ---------------------------------------------------------------------------
_PY_SIZED = [0]  # regions whose width / height the reference holds as Python ints, seen by _compare_assoc


def _compare_assoc(assoc, b, out, f0, proc_frames):
    from cpx.tracking import REGION_BLANK, REGION_BORDER, REGION_CROPPED

    for q, f in enumerate(proc_frames):
        regs = out["region_history"][q]
        got = assoc.frame_regions(f0 + f)
        assert len(got) == len(regs), (b, q)
        for g, r in zip(got, regs):
            assert (g["x"], g["y"], g["width"], g["height"], g["mass"], g["id"]) == (
                r.x, r.y, r.width, r.height, int(r.mass), r.id), (b, q)
            assert bool(g["flags"] & REGION_CROPPED) == r.was_cropped
            assert bool(g["flags"] & REGION_BORDER) == r.is_along_border
            assert g["cx"] == float(r.centroid[0]) and g["cy"] == float(r.centroid[1])
            np.testing.assert_allclose(g["pixel_variance"], float(r.pixel_variance), rtol=VAR_RTOL, atol=VAR_ATOL)
    want = sorted(out["tracks"] + [t for _, t in out.get("filtered_tracks", [])], key=lambda t: t.id)
    tr = assoc.clip_tracks(b)
    assert len(tr) == len(want), b
    nblank = 0
    for (rec, regs), w in zip(tr, want):
        assert (rec["id"], rec["start_frame"], rec["n_frames"]) == (w.id, w.start_frame, len(w.bounds)), (b, w.id)
        assert (rec["blank_frames"], rec["since_seen"], rec["rt_frames"]) == (w.blank_frames, w.since_seen, w.rt_frames)
        for g, r in zip(regs, w.bounds):
            assert (g["x"], g["y"], g["width"], g["height"], g["mass"], g["frame_number"]) == (
                r.x, r.y, r.width, r.height, int(r.mass), r.frame_number), (b, w.id, r.frame_number)
            assert bool(g["flags"] & REGION_BLANK) == r.blank
            # width / height typed as Python ints in the reference (decides the dtype of the next blank region)
            assert (bool(g["flags"] & 16), bool(g["flags"] & 32)) == (r.py[2], r.py[3]), (b, w.id, r.frame_number)
            assert g["cx"] == float(r.centroid[0]) and g["cy"] == float(r.centroid[1]), (b, w.id, r.frame_number)
            nblank += int(r.blank)
            _PY_SIZED[0] += int(r.py[2] or r.py[3])
    return len(tr), nblank


def test_association_matches_oracle_on_fixture_clips(engines):
    import track_oracle as to

    eng = engines("lepton3")
    clips, metas, offs, outs = [], [], [0], []
    for name in ("possum", "hedgehog"):
        frames, t_on, ffc, bgf, hdr = load_clip(name)
        clips.append(frames)
        metas.append(eng.make_meta(frames.shape[0], t_on, ffc, bgf))
        offs.append(offs[-1] + frames.shape[0])
        outs.append((to.track_clip(frames, t_on, ffc, bgf, to.OracleConfig(hdr.model), keep=True, apply_filter=False), bgf))
    meta = np.concatenate(metas)
    offs = np.array(offs, np.int32)
    res = eng.track_batch(eng.upload_frames(np.concatenate(clips)), offs, meta)
    res.check()
    assoc = eng.associate_batch(res, offs, meta)
    assoc.check()
    for b, (out, bgf) in enumerate(outs):
        proc = [i for i in range(len(bgf)) if not bgf[i]]
        n, _ = _compare_assoc(assoc, b, out, int(offs[b]), proc)
        assert n == (7 if b == 0 else len(out["tracks"]))


def test_association_matches_oracle_on_synthetic(engines):
    import track_oracle as to
    from cpx import synth

    eng = engines("lepton3")
    n_clips, T = 10, 150
    rng = np.random.default_rng(11)
    clips = [synth.make_clip(rng, T, max_blobs=3) for _ in range(n_clips)]
    offs = (np.arange(n_clips + 1) * T).astype(np.int32)
    meta = eng.make_meta(n_clips * T)
    res = eng.track_batch(eng.upload_frames(np.concatenate(clips)), offs, meta)
    res.check()
    assoc = eng.associate_batch(res, offs, meta)
    assoc.check()
    nt = nb = 0
    for b in range(n_clips):
        out = to.track_clip(clips[b], cfg=to.OracleConfig("lepton3"), keep=True, apply_filter=False)
        a, c = _compare_assoc(assoc, b, out, int(offs[b]), list(range(T)))
        nt += a
        nb += c
    assert nt > 10 and nb > 10


def test_denoise_matches_oracle_on_synthetic(engines):
    """tracking.denoise = True (cv2.fastNlMeansDenoising between normalise and blur): label images and
    components bit-identical to the oracle on synthetic clips (short: the NumPy NLM is slow)."""
    from cpx import synth
    import track_oracle as to

    rng = np.random.default_rng(77)
    clips = [synth.make_clip(rng, 14, max_blobs=3) for _ in range(3)]
    eng = engines("lepton3", denoise=True)
    offs = (np.arange(4) * 14).astype(np.int32)
    res = eng.track_batch(eng.upload_frames(np.concatenate(clips)), offs, eng.make_meta(42), want_labels=True,
                          want_filtered=True)
    res.check()
    labels = res.labels()
    ncomp = 0
    for b in range(3):
        cfg = to.OracleConfig("lepton3")
        cfg.denoise = True
        out = to.track_clip(clips[b], cfg=cfg, keep=True, do_tracking=False)
        for i, o in enumerate(out["frames"]):
            f = int(offs[b]) + i
            assert np.array_equal(labels[f], o["mask"]), (b, i)
            c = res.components(f)
            got = np.stack([c["x"], c["y"], c["width"], c["height"], c["area"]], axis=1).reshape(-1, 5)
            assert np.array_equal(got, o["stats"]), (b, i)
            ncomp += len(c)
    assert ncomp > 0


@pytest.mark.parametrize("W,H", [(32, 24), (64, 40), (80, 60), (128, 96), (184, 104), (160, 128)])
def test_other_resolutions_match_oracle(W, H):
    """The kernels take the resolution from the handle (the reference reads it from the CPTV header): smaller and
    non-4:3 frames inside the kernel envelope (W % 8 == 0, W < 192, W*H <= 20480) against the oracle, pixel stage and
    association.  The sizes also walk the streaming pass through its forms: no full round of pixel pairs (32 x 24), one
    (64 x 40), an even and an odd number of them, each with a ragged last round; 160 x 128 is the largest frame the
    kernel takes (20 pixels per lane, one workgroup per CU by its LDS footprint)."""
    import track_oracle as to
    from cpx import synth
    from cpx.engine import TrackEngine

    n_clips, T = 3, 70
    frames, offs = synth.make_batch(n_clips, T, seed=W * 1000 + H, h=H, w=W)
    eng = TrackEngine(width=W, height=H, model="lepton3", max_frames=T)
    meta = np.concatenate([eng.make_meta(T) for _ in range(n_clips)])
    res = eng.track_batch(eng.upload_frames(frames), offs, meta, want_labels=True, want_filtered=True,
                          want_background=True)
    res.check()
    assoc = eng.associate_batch(res, offs, meta)
    assoc.check()
    ncomp = ntracks = 0
    for b in range(n_clips):
        clip = frames[offs[b]:offs[b + 1]]
        out = to.track_clip(clip, None, None, None, to.OracleConfig("lepton3"), keep=True, apply_filter=False)
        _compare_with_oracle(res, int(offs[b]), out, T)
        assert np.array_equal(res.background()[b].astype(np.int32), out["frames"][-1]["bg_after"])
        ncomp += sum(f["n_components"] for f in out["frames"])
        got = assoc.clip_tracks(b)
        assert [int(r["id"]) for r, _ in got] == [t.id for t in out["tracks"]]
        for (rec, regs), t in zip(got, out["tracks"]):
            assert int(rec["start_frame"]) == t.start_frame and len(regs) == len(t.bounds)
            for r, o in zip(regs, t.bounds):
                assert (r["x"], r["y"], r["width"], r["height"], r["mass"], r["frame_number"]) == (
                    o.x, o.y, o.width, o.height, int(o.mass), o.frame_number)
                assert bool(r["flags"] & 1) == bool(o.blank)
        ntracks += len(got)
    assert ncomp > 0
    eng.close()


def test_pipelined_split_steps_equal_fused(monkeypatch):
    """CPX_TRACK_SPLIT_MIN_CLIPS (off by default): front halves of the frame step on one stream,
    back halves one step behind on a second one.  Same bits as the fused kernel."""
    from cpx import synth
    from cpx.engine import TrackEngine

    n_clips, T = 5, 60
    frames, offs = synth.make_batch(n_clips, T, seed=77)
    outs = []
    for split in ("0", "1"):
        monkeypatch.setenv("CPX_TRACK_SPLIT_MIN_CLIPS", split)
        eng = TrackEngine(model="lepton3", max_frames=T)
        meta = np.concatenate([eng.make_meta(T) for _ in range(n_clips)])
        res = eng.track_batch(eng.upload_frames(frames), offs, meta, want_labels=True, want_filtered=True,
                              want_background=True)
        res.check()
        outs.append((res.info.copy(), [res.components(f).copy() for f in range(n_clips * T)], res.labels(),
                     res.filtered(), res.background()))
        eng.close()
    a, b = outs
    assert np.array_equal(a[0], b[0])
    assert all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    assert sum(len(c) for c in a[1]) > 0


def test_connected_component_torture(engines):
    """Random blocky hot patterns: many components of irregular shape born in different 2x2 blocks, touching only
    diagonally, nested, wrapping around others -- label images, label numbering (OpenCV's block-raster order), boxes,
    areas and centroids against the oracle; frames whose component count exceeds the capacity must report it."""
    from cpx._lib import CpxError

    eng = engines("lepton3")
    H, W = 120, 160
    rng = np.random.default_rng(2024)
    clips, specs = [], []
    for k in range(36):
        cell = int(rng.choice([3, 4, 5, 6, 8, 12]))
        density = float(rng.uniform(0.12, 0.6))
        gh, gw = -(-H // cell), -(-W // cell)
        grid = rng.random((gh, gw)) < density
        if k % 5 == 0:   # diagonal-only contacts and thin bridges
            grid = np.zeros((gh, gw), bool)
            ii, jj = np.indices((gh, gw))
            grid[(ii + jj) % 2 == 0] = rng.random(((ii + jj) % 2 == 0).sum()) < density * 1.3
        mask = np.kron(grid, np.ones((cell, cell), bool))[:H, :W]
        f = np.full((2, H, W), 3000, np.uint16)
        f[1][mask] += np.uint16(rng.integers(200, 900))
        f[1] += rng.integers(0, 3, (H, W)).astype(np.uint16)
        clips.append(f)
        specs.append((cell, density))
    frames = np.concatenate(clips)
    offs = (np.arange(len(clips) + 1) * 2).astype(np.int32)
    res = eng.track_batch(eng.upload_frames(frames), offs, eng.make_meta(len(frames)), want_labels=True,
                          want_filtered=True)
    labels = res.labels()
    counts, overflowed = [], 0
    for k, f in enumerate(clips):
        out, _ = _oracle_clip(f, "lepton3")
        o = out["frames"][1]
        fi = res.info[2 * k + 1]
        n = o["n_components"]
        counts.append(n)
        assert fi["n_components"] == n, (k, specs[k])
        if n > eng.cap:
            overflowed += 1
            assert fi["status"] == -5
            continue
        assert fi["status"] == 0
        assert np.array_equal(labels[2 * k + 1], o["mask"]), (k, specs[k], n)
        c = res.components(2 * k + 1)
        got = np.stack([c["x"], c["y"], c["width"], c["height"], c["area"]], axis=1).reshape(-1, 5)
        assert np.array_equal(got, o["stats"]), (k, specs[k])
        if n:
            cent = np.stack([c["sum_x"] / c["area"], c["sum_y"] / c["area"]], axis=1)
            assert np.array_equal(cent, o["centroids"]), (k, specs[k])
    ok = [n for n in counts if n <= eng.cap]
    assert len(ok) >= 12 and max(ok) >= 20 and overflowed >= 1, (sorted(counts), overflowed)
    if overflowed:
        with pytest.raises(CpxError):
            res.check()


def test_association_busy_scenes(engines):
    """Up to 9 warm objects per clip crossing, overlapping and leaving: matching order, new-track creation under
    overlap, blank frames with Kalman predictions and track retirement against the oracle; lepton3.5 thresholds too."""
    import track_oracle as to
    from cpx import synth

    nt = nb = 0
    for model, seed in (("lepton3", 5), ("lepton3.5", 6)):
        eng = engines(model)
        n_clips, T = 6, 120
        rng = np.random.default_rng(seed)
        clips = [synth.make_clip(rng, T, model=model, max_blobs=9) for _ in range(n_clips)]
        offs = (np.arange(n_clips + 1) * T).astype(np.int32)
        meta = eng.make_meta(n_clips * T)
        res = eng.track_batch(eng.upload_frames(np.concatenate(clips)), offs, meta)
        res.check()
        assoc = eng.associate_batch(res, offs, meta)
        assoc.check()
        for b in range(n_clips):
            out = to.track_clip(clips[b], cfg=to.OracleConfig(model), keep=True, apply_filter=False)
            a, c = _compare_assoc(assoc, b, out, int(offs[b]), list(range(T)))
            nt += a
            nb += c
    assert nt > 40 and nb > 40
    assert _PY_SIZED[0] > 0  # the float32 blank-region case was exercised


@pytest.mark.parametrize("W,H", [(32, 24), (64, 48), (128, 96), (184, 104)])
def test_denoise_other_widths_match_oracle(W, H):
    """The NLM kernel has a compile-time-width form for 160-wide frames and a generic one: the generic form (other
    strides, other numbers of segments and sub-bands, a last band shorter than the others) against the oracle."""
    import track_oracle as to
    from cpx import synth
    from cpx.engine import TrackEngine

    n_clips, T = 2, 5
    frames, offs = synth.make_batch(n_clips, T, seed=W + H, h=H, w=W)
    eng = TrackEngine(width=W, height=H, model="lepton3", max_frames=T, denoise=True)
    meta = np.concatenate([eng.make_meta(T) for _ in range(n_clips)])
    res = eng.track_batch(eng.upload_frames(frames), offs, meta, want_labels=True)
    res.check()
    labels = res.labels()
    n = 0
    for b in range(n_clips):
        cfg = to.OracleConfig("lepton3")
        cfg.denoise = True
        out = to.track_clip(frames[offs[b]:offs[b + 1]], cfg=cfg, keep=True, do_tracking=False)
        for i, o in enumerate(out["frames"]):
            assert np.array_equal(labels[int(offs[b]) + i], o["mask"]), (b, i)
            n += int(o["mask"].max() > 0)
    assert n > 0
    eng.close()


def test_denoise_band_variants_by_batch_size(engines):
    """The NLM kernel is instantiated for 10 / 5 / 2 / 1 rows per pass-B thread (two columns each) and picked by batch
    size (whole frames per workgroup for big batches, bands of a frame for small ones): every variant must give the
    oracle's label images.  Two distinct short clips, replicated to the batch sizes that select each variant."""
    import torch
    import track_oracle as to
    from cpx import synth

    rng = np.random.default_rng(123)
    T = 6
    base = [synth.make_clip(rng, T, max_blobs=3) for _ in range(2)]
    want = []
    for clip in base:
        cfg = to.OracleConfig("lepton3")
        cfg.denoise = True
        out = to.track_clip(clip, cfg=cfg, keep=True, do_tracking=False)
        want.append(np.stack([o["mask"] for o in out["frames"]]))
    want = np.concatenate(want)                       # [2T, H, W]
    eng = engines("lepton3", denoise=True)
    dev_base = eng.upload_frames(np.concatenate(base))
    for B in (2, 32, 64, 128, 384):                   # -> 1, 1, 2, 5, 10 rows per thread at 160x120
        reps = B // 2
        frames = dev_base.unsqueeze(0).expand(reps, -1, -1, -1).reshape(B * T, 120, 160).contiguous()
        offs = (np.arange(B + 1) * T).astype(np.int32)
        res = eng.track_batch(frames, offs, eng.make_meta(B * T), want_labels=True)
        eng.synchronize()
        lab = res.labels_dev.reshape(reps, 2 * T, 120, 160)
        assert bool((lab == lab[0:1]).all().item()), B
        assert torch.equal(lab[0].cpu(), torch.from_numpy(want.astype(np.int32))), B
    assert want.max() > 0


def test_association_matches_reference_on_busy_scenes(engines, golden_dir):
    """The HIP association directly against what the REFERENCE produced on 11 seeded busy scenes
    (tests/golden/busy_tracks.npz, make_golden_busy.py): all tracks, all bounds, blank flags, Python-int typing."""
    import os

    from cpx import synth

    z = np.load(os.path.join(golden_dir, "busy_tracks.npz"))
    seeds, rows, goffs, T = z["seeds"], z["rows"], z["offsets"], int(z["frames"])
    eng = engines("lepton3")
    clips = [synth.make_clip(np.random.default_rng(1000 + int(s)), T, max_blobs=8) for s in seeds]
    offs = (np.arange(len(clips) + 1) * T).astype(np.int32)
    t_on, ffc = [100000 + 111 * i for i in range(T)], [40000] * T
    meta = np.concatenate([eng.make_meta(T, t_on, ffc) for _ in clips])
    res = eng.track_batch(eng.upload_frames(np.concatenate(clips)), offs, meta)
    res.check()
    assoc = eng.associate_batch(res, offs, meta)
    assoc.check()
    for k in range(len(clips)):
        got = []
        for rec, regs in assoc.clip_tracks(k):
            for g in regs:
                got.append((rec["id"], g["x"], g["y"], g["width"], g["height"], g["mass"], g["frame_number"],
                            int(bool(g["flags"] & 1)), int(bool(g["flags"] & 16)), int(bool(g["flags"] & 32))))
        # the device keeps the untrimmed tracks; the reference's golden holds them after trim()
        want = rows[goffs[k]:goffs[k + 1]]
        got = np.asarray(got, np.int32)
        idx = {tuple(r[[0, 6]]): i for i, r in enumerate(got)}
        sel = [idx[(int(w[0]), int(w[6]))] for w in want]
        assert np.array_equal(got[sel], want), int(seeds[k])


def test_median_of_adversarial_frames(engines):
    """cpx_median_kernel probes the mean first and gallops away from it before it bisects (round 6): frames whose median is as
    far from their mean as 16-bit values allow -- two levels split at every third, saturated neighbours, noise over the whole
    range -- must still select exactly, and terminate (a galloping step that kept growing through the bisection wrapped to
    zero on exactly such a frame)."""
    eng = engines("lepton3")
    H, W = 120, 160
    rng = np.random.default_rng(77)
    frames = []
    for a, b in ((0, 65535), (0, 1), (65534, 65535), (0, 2), (1, 65535), (7, 40000)):
        for frac in (0.3, 0.5, 0.7):
            x = np.full(H * W, a, np.uint16)
            x[int(H * W * frac):] = b
            frames.append(x.reshape(H, W))
            y = np.full(H * W, b, np.uint16)
            y[int(H * W * frac) - 1:] = a
            frames.append(rng.permutation(y).reshape(H, W))
    for _ in range(6):
        frames.append(rng.integers(0, 65536, size=(H, W)).astype(np.uint16))
        frames.append((rng.integers(0, 65533) + rng.integers(0, 3, size=(H, W))).astype(np.uint16))
    frames = np.stack(frames)
    n = frames.shape[0]
    res = eng.track_batch(eng.upload_frames(frames), np.array([0, n], np.int32), eng.make_meta(n))
    got = res.info["thermal_median"]
    for i in range(n):
        assert float(got[i]) == float(np.float32(np.median(frames[i]))), i
