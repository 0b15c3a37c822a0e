"""IR detection stage on the GPU: cpx_ir_detect against oracle/ir_oracle.py and against what the
reference's own detect_objects_ir returned (tests/golden/ir_detect_golden.json); host merge_components against both.
Integer work: bit-exact."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, IR_CASES, crc, ir_mask

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3")
    yield eng
    eng.close()


def _stats_of(comps, n):
    c = comps[:n]
    return np.stack([c["x"], c["y"], c["width"], c["height"], c["area"]], axis=1).astype(np.int32)


def test_ir_detect_matches_reference_golden(engine):
    import torch

    from cpx.engine import CpxError
    from cpx.track import irdetect

    with open(os.path.join(GOLDEN, "ir_detect_golden.json")) as fh:
        gold = json.load(fh)
    checked = 0
    for g in gold:
        img = ir_mask(g["case"])
        dev = torch.from_numpy(img).to(engine.device)
        if g["n"] > 1024:  # the caller's table too small: CPX_ERR_OVERFLOW with the true count, nothing truncated
            with pytest.raises(CpxError) as ei:
                engine.ir_detect(dev[None], 0, 1024)
            assert ei.value.code == -5 and "has %d components" % g["n"] in str(ei.value), ei.value
        n, labels, stats = irdetect.detect_objects_ir(engine, dev, threshold=0, max_components=20000,
                                                      want_labels=True)
        assert n - 1 == g["n"], g["case"]
        assert crc(labels.cpu().numpy().astype(np.int32)) == g["mask_crc"], g["case"]
        assert crc(stats[1:].astype(np.int32)) == g["stats_crc"], g["case"]
        merged = irdetect.merge_components(stats[1:].copy())
        assert [[int(v) for v in r] for r in merged] == g["merged"], g["case"]
        checked += 1
    assert checked == len(IR_CASES)


@pytest.mark.parametrize("shape", [(480, 640), (240, 320), (120, 192), (33, 64), (480, 64)])
def test_ir_detect_matches_oracle_random(engine, shape):
    import torch

    import ir_oracle as iro

    H, W = shape
    rng = np.random.default_rng(H * 1000 + W)
    frames = []
    for k in range(6):
        img = np.zeros((H, W), np.uint8)
        dens = [0.0, 0.01, 0.3, 0.6, 0.9, 1.0][k]
        img[rng.random((H, W)) < dens] = rng.integers(1, 256)
        if k == 2:  # values at and around a non-zero threshold
            img = rng.integers(0, 8, size=(H, W)).astype(np.uint8)
        frames.append(img)
    thr = 3
    dev = torch.from_numpy(np.stack(frames)).to(engine.device)
    for threshold in (0, thr):
        oracle = [iro.detect_objects_ir(f, threshold=threshold) for f in frames]
        cap = (H + 1) // 2 * ((W + 1) // 2)  # the most components an 8-connected labelling can have
        counts, comps, labels = engine.ir_detect(dev, threshold, cap, want_labels=True)  # LDS and scratch frames mixed
        for i in range(len(frames)):
            n, mask, stats = oracle[i]
            assert counts[i] == n - 1, (shape, i, threshold)
            assert np.array_equal(labels[i].cpu().numpy(), mask.astype(np.int32)), (shape, i, threshold)
            assert np.array_equal(_stats_of(comps[i], counts[i]), stats[1:, :5].astype(np.int32))
            if n > 1:  # centroids: OpenCV's are sum / area in float64
                c = comps[i][: counts[i]]
                lab = mask.astype(np.int64)
                yy, xx = np.mgrid[:H, :W]
                sx = np.bincount(lab.ravel(), weights=xx.ravel(), minlength=n)[1:]
                sy = np.bincount(lab.ravel(), weights=yy.ravel(), minlength=n)[1:]
                assert np.array_equal(c["sum_x"], sx.astype(np.int64)) and np.array_equal(c["sum_y"], sy.astype(np.int64))


def test_ir_detect_batch_and_arguments(engine):
    import torch

    from cpx.engine import CpxError

    imgs = np.stack([ir_mask(c) for c in IR_CASES if c["kind"] in ("blobs", "empty")])
    dev = torch.from_numpy(imgs).to(engine.device)
    counts, comps, _ = engine.ir_detect(dev, 0, 1024)
    for i in range(len(imgs)):
        c1, k1, _ = engine.ir_detect(dev[i:i + 1], 0, 1024)
        assert c1[0] == counts[i] and np.array_equal(k1[0][: c1[0]], comps[i][: counts[i]])
    # a caller's own limit below the frame's component count is an overflow, not a truncation
    busy = int(np.argmax(counts))
    with pytest.raises(CpxError) as ei:
        engine.ir_detect(dev[busy:busy + 1], 0, int(counts[busy]) - 1)
    assert ei.value.code == -5
    # every table past LDS at once: after the (downward-shifting) open two rows in three are set in every other
    # column -- 160 x 320 components of two pixels, 51200 runs per frame; three such frames share the scratch slots
    import ir_oracle as iro

    worst = np.zeros((480, 640), np.uint8)
    worst[0::3, 0::2] = 255
    worst[2::3, 0::2] = 255
    n, mask, stats = iro.detect_objects_ir(worst, threshold=0)
    assert n - 1 == 160 * 320
    many = torch.from_numpy(np.stack([worst] * 3 + [imgs[0]])).to(engine.device)
    counts4, comps4, labels4 = engine.ir_detect(many, 0, 160 * 320, want_labels=True)
    assert counts4.tolist() == [160 * 320] * 3 + [int(counts[0])]
    for i in range(3):
        assert np.array_equal(labels4[i].cpu().numpy(), mask.astype(np.int32))
        assert np.array_equal(_stats_of(comps4[i], counts4[i]), stats[1:, :5].astype(np.int32))
    assert np.array_equal(comps4[3][: counts[0]], comps[0][: counts[0]])
    with pytest.raises(CpxError):  # width not a multiple of 64
        engine.ir_detect(torch.zeros((1, 48, 100), dtype=torch.uint8, device=engine.device), 0, 16)
    with pytest.raises(ValueError):
        engine.ir_detect(torch.zeros((48, 128), dtype=torch.uint8, device=engine.device), 0, 16)


def _ir_video(rng, n, H, W):
    """uint8 frames: a static textured scene with sensor noise, a bright object crossing it, a lighting step."""
    scene = rng.integers(40, 200, size=(H, W)).astype(np.int32)
    frames = []
    for t in range(n):
        f = scene + rng.integers(-2, 3, size=(H, W))
        if t >= n // 2:
            f = f + 6  # global illumination change: modes drift / new modes appear
        if 5 <= t:
            x0 = (7 * t) % (W - 40)
            f[H // 3:H // 3 + 30, x0:x0 + 40] = 240 - (t % 3)
        frames.append(np.clip(f, 0, 255).astype(np.uint8))
    return np.stack(frames)


@pytest.mark.parametrize("shape,rates", [((120, 160), None), ((480, 640), None), ((64, 96), [1, -1, -1, 0.0, 0.01, -1, 0.5, -1])])
def test_mog2_matches_oracle(engine, shape, rates):
    """cpx_mog2_apply / cpx_mog2_background against oracle/mog2_oracle.c frame by frame: masks and background images
    bit for bit (float32 state, same operation order).  Parity with cv2 itself is unpinned."""
    import torch

    import mog2_oracle as mo
    from cpx.track.irdetect import MOG2Background

    H, W = shape
    rng = np.random.default_rng(H + W)
    n = 24 if H < 400 else 10
    video = _ir_video(rng, n, H, W)
    ora = mo.MOG2(W, H, history=1000, var_threshold=16.0)
    dev = MOG2Background(engine, W, H, n_streams=1, history=1000, var_threshold=16.0)
    fg_seen = 0
    for t in range(n):
        lr = -1 if rates is None else rates[t % len(rates)]
        want = ora.apply(video[t], lr)
        got = dev.update_background(torch.from_numpy(video[t]).to(engine.device), learning_rate=lr)
        assert np.array_equal(got.cpu().numpy(), want), (t, lr)
        assert np.array_equal(dev.background.cpu().numpy(), ora.getBackgroundImage()), t
        fg_seen += int((want > 0).sum()) if t > 5 else 0
    assert fg_seen > 300  # the moving object was detected
    dev.close()
    ora.close()


def test_mog2_streams_and_detection_chain(engine):
    """Several streams in lockstep equal the streams one by one, and mask -> cpx_ir_detect -> merge runs on the result."""
    import torch

    import ir_oracle as iro
    import mog2_oracle as mo
    from cpx.track import irdetect

    H, W, S, n = 240, 320, 3, 12
    rngs = [np.random.default_rng(50 + s) for s in range(S)]
    videos = [_ir_video(r, n, H, W) for r in rngs]
    oras = [mo.MOG2(W, H) for _ in range(S)]
    dev = irdetect.MOG2Background(engine, W, H, n_streams=S)
    for t in range(n):
        batch = torch.from_numpy(np.stack([v[t] for v in videos])).to(engine.device)
        masks = dev.update_background(batch)
        want = np.stack([o.apply(v[t]) for o, v in zip(oras, videos)])
        assert np.array_equal(masks.cpu().numpy(), want), t
    # the last masks through the detection stage
    res = irdetect.detect_objects_ir(engine, masks.contiguous(), threshold=0, max_components=8192)
    for s in range(S):
        n_o, _, stats_o = iro.detect_objects_ir(want[s], threshold=0)
        assert res[s][0] == n_o and np.array_equal(res[s][2][1:], stats_o[1:, :5].astype(np.int32))
        merged = irdetect.merge_components(res[s][2][1:].copy())
        assert [[int(v) for v in r] for r in merged] == [[int(v) for v in r] for r in iro.merge_components(stats_o[1:].copy())]
    dev.close()


@pytest.mark.parametrize("shape,with_mask", [((5, 480, 640), True), ((3, 37, 53), True), ((4, 120, 160), False), ((2, 1, 1), True)])
def test_ir_frame_statistics_match_numpy(engine, shape, with_mask):
    """cpx_ir_frame_statistics == np.min / np.max / np.median / np.nanmean of every frame and np.sum(|filtered|)
    (Clip.add_frame, track/clip.py:330-347): bit-exact, odd and even pixel counts, vector and byte paths."""
    import ctypes as C

    import torch

    from cpx._lib import IR_FRAME_STATS_DTYPE

    rng = np.random.default_rng(sum(shape))
    n = shape[0]
    frames = rng.integers(0, 256, shape, dtype=np.uint8)
    frames[0] = rng.integers(90, 93, shape[1:], dtype=np.uint8)        # a narrow histogram: the median sits between bins
    if n > 1:
        frames[1] = 255
    masks = (rng.random(shape) < 0.1).astype(np.uint8) * 255
    fd, md = torch.from_numpy(frames).to(engine.device), torch.from_numpy(masks).to(engine.device)
    hist = torch.empty((n, 256), dtype=torch.int32, device=engine.device)
    out = torch.full((n, 32), 7, dtype=torch.uint8, device=engine.device)
    torch.cuda.synchronize()
    rc = engine.lib.cpx_ir_frame_statistics(engine.h, C.c_void_p(fd.data_ptr()), C.c_void_p(md.data_ptr()) if with_mask else None,
                                            n, int(np.prod(shape[1:])), C.c_void_p(hist.data_ptr()), C.c_void_p(out.data_ptr()))
    assert rc == 0, engine._err()
    engine.synchronize()
    got = out.cpu().numpy().view(IR_FRAME_STATS_DTYPE).reshape(n)
    for i in range(n):
        assert got["min"][i] == frames[i].min() and got["max"][i] == frames[i].max()
        assert got["sum"][i] == int(frames[i].sum(dtype=np.int64))
        assert got["median_x2"][i] / 2.0 == float(np.median(frames[i]))
        assert got["sum"][i] / frames[i].size == np.nanmean(frames[i])
        assert got["filtered_sum"][i] == (int(masks[i].sum(dtype=np.int64)) if with_mask else 0)
    assert np.array_equal(hist.cpu().numpy(), np.stack([np.bincount(f.ravel(), minlength=256) for f in frames]))


def test_ir_merge_kernel_matches_oracle_on_fragment_soups(engine):
    """cpx_ir_merge through the C ABI against oracle/ir_oracle.merge_components (pinned by the reference-run golden
    above) + np.var of the uint8-wrapping frame difference: hundreds of fragments per frame -- chains of merges,
    restarts, rows sharing the anchor's x (never merged), equal areas (stable order), everything filtered, nothing
    filtered -- for several streams in one launch, with and without a previous frame."""
    import ctypes as C

    import torch

    import ir_oracle as iro
    from cpx._lib import COMPONENT_DTYPE, FRAME_INFO_DTYPE

    W, H, cap_in, cap_out, T = 640, 480, 512, 512, 5
    rng = np.random.default_rng(21)
    sets = []
    for kind in range(8):
        n = [0, 1, 40, 200, 400, 60, 120, 300][kind]
        rows = []
        for _ in range(n):
            w, h = int(rng.integers(1, 60)), int(rng.integers(1, 60))
            if kind == 5:
                w = h = int(rng.integers(1, 6))                       # nothing survives the small-fragment filter
            x, y = int(rng.integers(0, W - w)), int(rng.integers(0, H - h))
            if kind == 6:
                x = 8 * int(rng.integers(0, 10))                      # many rows share an x: the reference never merges those
            area = int(rng.integers(1, w * h + 1)) if kind != 7 else 50   # kind 7: equal areas (stable sort)
            rows.append([x, y, w, h, area])
        sets.append(rows)
    V = len(sets)
    comps = np.zeros((V, cap_in), COMPONENT_DTYPE)
    counts = np.zeros(V, np.int32)
    for v, rows in enumerate(sets):
        counts[v] = len(rows)
        for k, r in enumerate(rows):
            comps[v, k] = (r[0], r[1], r[2], r[3], r[4], 0, 0, 0.0)
    cur = rng.integers(0, 256, (V, H, W), dtype=np.uint8)
    prev = rng.integers(0, 256, (V, H, W), dtype=np.uint8)
    to = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(engine.device)
    comps_d, counts_d, cur_d, prev_d = to(comps), to(counts), to(cur), to(prev)
    p = lambda x: C.c_void_p(x.data_ptr())
    for use_prev, q in ((True, 3), (False, 0)):
        out_c = torch.zeros(V * T * cap_out * 32, dtype=torch.uint8, device=engine.device)
        out_i = torch.zeros(V * T * 80, dtype=torch.uint8, device=engine.device)
        status = torch.full((V,), 7, dtype=torch.int32, device=engine.device)
        torch.cuda.synchronize()
        rc = engine.lib.cpx_ir_merge(engine.h, p(comps_d), p(counts_d), V, cap_in, cap_out, p(cur_d),
                                     p(prev_d) if use_prev else None, W, H, q, T, p(out_c), p(out_i), p(status))
        assert rc == 0, engine._err()
        engine.synchronize()
        assert status.cpu().numpy().tolist() == [0] * V
        got_c = out_c.cpu().numpy().view(COMPONENT_DTYPE).reshape(V * T, cap_out)
        got_i = out_i.cpu().numpy().view(FRAME_INFO_DTYPE).reshape(V * T)
        for v, rows in enumerate(sets):
            want = iro.merge_components([np.array(r) for r in rows])
            row = v * T + q
            assert int(got_i["n_components"][row]) == len(want) and int(got_i["frame_number"][row]) == q, (v, len(want))
            g = got_c[row, :len(want)]
            for k, m in enumerate(want):
                x, y, w, h, area = (int(t) for t in m[:5])
                assert (int(g["x"][k]), int(g["y"][k]), int(g["width"][k]), int(g["height"][k]), int(g["area"][k])) == (x, y, w, h, area), (v, k)
                # the tracker's centroid: the truncated box centre (sum_x / area on the device record)
                assert int(g["sum_x"][k]) == int(x + w / 2) * area and int(g["sum_y"][k]) == int(y + h / 2) * area
                var = 0.0
                if use_prev:
                    x1, y1 = min(x + w, W), min(y + h, H)
                    d = (cur[v, y:y1, x:x1] - prev[v, y:y1, x:x1]).astype(np.float64)   # uint8 arithmetic wraps first
                    var = float(np.var(d)) if d.size else 0.0
                assert abs(float(g["pixel_variance"][k]) - var) <= 1e-5 * max(1.0, var), (v, k)
    # more rows surviving the small-fragment filter than the output capacity (the kernel's working set): reported for
    # that stream alone, never truncated
    survivors = [sum(1 for r in rows if r[4] > 40 or (r[2] > 16 and r[3] > 16)) for rows in sets]
    small_cap = 128
    assert any(n > small_cap for n in survivors) and any(0 < n <= small_cap for n in survivors)
    st = torch.zeros(V, dtype=torch.int32, device=engine.device)
    oc = torch.zeros(V * T * small_cap * 32, dtype=torch.uint8, device=engine.device)
    torch.cuda.synchronize()
    rc = engine.lib.cpx_ir_merge(engine.h, p(comps_d), p(counts_d), V, cap_in, small_cap, p(cur_d), None, W, H, 0, T,
                                 p(oc), None, p(st))
    assert rc == 0, engine._err()
    engine.synchronize()
    assert st.cpu().numpy().tolist() == [-5 if n > small_cap else 0 for n in survivors]
