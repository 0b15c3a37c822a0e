"""Synthetic 160x120 thermal clips for benchmarks and parity tests
(BASELINE.md section 3 / SURVEY.md section 8(d), config 2): smooth static background,
per-frame sensor noise, 0-3 warm Gaussian blobs on a random walk."""

import numpy as np


def smooth_noise(rng, h, w, cells=6):
    """Bilinear up-sampling of a coarse random grid -> values in roughly [-1, 1]."""
    g = rng.standard_normal((cells + 1, cells + 1))
    ys = np.linspace(0, cells, h)
    xs = np.linspace(0, cells, w)
    y0 = np.minimum(ys.astype(int), cells - 1)
    x0 = np.minimum(xs.astype(int), cells - 1)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_clip(rng, n_frames=270, h=120, w=160, model="lepton3", max_blobs=3):
    base = 28000.0 if model == "lepton3.5" else 2900.0
    scale = 4.0 if model == "lepton3.5" else 1.0
    bg = base + 40.0 * scale * smooth_noise(rng, h, w)
    frames = np.empty((n_frames, h, w), dtype=np.float32)
    frames[:] = bg[None]
    frames += rng.normal(0.0, 4.0 * scale, size=frames.shape).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for _ in range(int(rng.integers(0, max_blobs + 1))):
        sigma = rng.uniform(3.0, 8.0)
        amp = rng.uniform(60.0, 400.0) * scale
        start = int(rng.integers(0, max(1, n_frames // 2)))
        length = int(rng.integers(n_frames // 4 + 1, n_frames + 1))
        side = int(rng.integers(0, 4))
        if side == 0:
            px, py = -sigma, rng.uniform(0, h)
        elif side == 1:
            px, py = w + sigma, rng.uniform(0, h)
        elif side == 2:
            px, py = rng.uniform(0, w), -sigma
        else:
            px, py = rng.uniform(0, w), h + sigma
        vx = (w / 2 - px) / 40.0 + rng.uniform(-0.5, 0.5)
        vy = (h / 2 - py) / 40.0 + rng.uniform(-0.5, 0.5)
        for t in range(start, min(n_frames, start + length)):
            vx = float(np.clip(vx + rng.uniform(-0.4, 0.4), -3, 3))
            vy = float(np.clip(vy + rng.uniform(-0.4, 0.4), -3, 3))
            px += vx
            py += vy
            y0, y1 = int(max(0, py - 4 * sigma)), int(min(h, py + 4 * sigma + 1))
            x0, x1 = int(max(0, px - 4 * sigma)), int(min(w, px + 4 * sigma + 1))
            if y0 >= y1 or x0 >= x1:
                continue
            d2 = (yy[y0:y1, x0:x1] - py) ** 2 + (xx[y0:y1, x0:x1] - px) ** 2
            frames[t, y0:y1, x0:x1] += amp * np.exp(-d2 / (2 * sigma * sigma))
    return np.clip(np.rint(frames), 0, 65535).astype(np.uint16)


def make_batch(n_clips, n_frames=270, seed=1234, model="lepton3", h=120, w=160):
    """-> frames uint16 [n_clips*n_frames, h, w], clip_offsets int32 [n_clips+1]."""
    rng = np.random.default_rng(seed)
    clips = [make_clip(rng, n_frames, h, w, model) for _ in range(n_clips)]
    offs = np.arange(n_clips + 1, dtype=np.int32) * n_frames
    return np.concatenate(clips, axis=0), offs


def frame_times(n, step_ms=114, ffc_lead_ms=60000, start_ms=100000):
    t_on = [start_ms + i * step_ms for i in range(n)]
    ffc = [start_ms - ffc_lead_ms] * n
    return t_on, ffc
