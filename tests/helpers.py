"""Shared helpers for the parity tests."""
import json
import os
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


# Seeded synthetic recordings the reference was also run on (tests/golden/make_golden_synth.py): what the two fixture
# clips do not cover -- lepton3.5 thresholds (weight_add = 1), FFC-affected frames in the middle of a clip.
SYNTH_CLIPS = {
    "synth35": dict(seed=3501, model="lepton3.5", frames=80, blobs=4, ffc_at=(30, 31, 32, 33), background_first=True),
}


class _Header:
    def __init__(self, model):
        self.model = model
        self.x_resolution, self.y_resolution = 160, 120


def synth_clip(name):
    """frames, time_on, last_ffc, background flags of a SYNTH_CLIPS recipe (deterministic)."""
    from cpx import synth

    spec = SYNTH_CLIPS[name]
    n = spec["frames"]
    frames = synth.make_clip(np.random.default_rng(spec["seed"]), n, model=spec["model"], max_blobs=spec["blobs"])
    t_on = [100000 + 111 * i for i in range(n)]
    ffc = [40000] * n
    for i in spec["ffc_at"]:   # FFC-affected: integer milliseconds compared with FFC_PERIOD.seconds = 9
        ffc[i] = t_on[i] - 5
    bgf = [False] * n
    bgf[0] = bool(spec["background_first"])
    return frames, t_on, ffc, bgf, _Header(spec["model"])


def load_clip(name):
    """All frames of a fixture CPTV -> (frames u16 [N,H,W], time_on, last_ffc, background flags, header)."""
    from cpx.cptv import CptvReader

    if name in SYNTH_CLIPS:
        return synth_clip(name)

    r = CptvReader(os.path.join(GOLDEN, name + ".cptv"))
    fr = r.read_all()
    frames = np.stack([f.pix for f in fr])
    t_on = [f.time_on for f in fr]
    ffc = [f.last_ffc_time for f in fr]
    bgf = [bool(f.background_frame) for f in fr]
    return frames, t_on, ffc, bgf, r.get_header()


def load_golden(name, dn):
    z = np.load(os.path.join(GOLDEN, "%s_dn%d.npz" % (name, dn)))
    with open(os.path.join(GOLDEN, "%s_dn%d_tracks.json" % (name, dn))) as fh:
        tracks = json.load(fh)
    return z, tracks


def encode_cptv(path, frames, widths, time_on=None, last_ffc=None, model=b"lepton3", background_first=False):
    """Write a CPTV v2 file (test-side encoder, the inverse of cpx.cptv): `frames` uint16 [N,H,W],
    `widths[i]` bits per delta of frame i -- the caller picks widths wide enough for the data."""
    import gzip
    import struct

    frames = np.asarray(frames, np.uint16)
    N, H, W = frames.shape

    def field(code, data):
        return bytes([len(data)]) + code + data

    out = bytearray(b"CPTV\x02H")
    hdr = [field(b"T", struct.pack("<Q", 1600000000000000)), field(b"X", struct.pack("<I", W)),
           field(b"Y", struct.pack("<I", H)), field(b"C", b"\x01"), field(b"D", b"synthetic"),
           field(b"Z", b"\x09")]
    if model:
        hdr.append(field(b"E", model))
    if background_first:
        hdr.append(field(b"g", b"\x01"))
    out += bytes([len(hdr)]) + b"".join(hdr)
    snake = np.arange(W * H).reshape(H, W)
    snake[1::2] = snake[1::2, ::-1].copy()
    order = np.argsort(snake.reshape(-1))  # scan index -> pixel index
    prev = np.zeros(W * H, np.int64)
    for i in range(N):
        cur = frames[i].reshape(-1).astype(np.int64)
        diff = (cur - prev)[order]
        prev = cur
        deltas = np.diff(diff, prepend=0)
        w = int(widths[i])
        d = deltas[1:]
        assert d.min() >= -(1 << (w - 1)) and d.max() < (1 << (w - 1)), "width %d too narrow" % w
        u = (d & ((1 << w) - 1)).astype(np.uint64)
        bits = ((u[:, None] >> np.arange(w - 1, -1, -1, dtype=np.uint64)) & 1).astype(np.uint8)
        packed = np.packbits(bits.reshape(-1)).tobytes()
        payload = struct.pack("<i", int(deltas[0])) + packed
        fl = [field(b"w", bytes([w])), field(b"f", struct.pack("<I", len(payload)))]
        if time_on is not None:
            fl += [field(b"t", struct.pack("<I", int(time_on[i]))), field(b"c", struct.pack("<I", int(last_ffc[i])))]
        if background_first and i == 0:
            fl.append(field(b"g", b"\x01"))
        out += b"F" + bytes([len(fl)]) + b"".join(fl) + payload
    with gzip.open(str(path), "wb") as f:
        f.write(bytes(out))


# Seeded 640x480 foreground images for the IR detection stage (tests/golden/make_golden_ir.py)
IR_CASES = [dict(seed=1, kind="blobs", n=6), dict(seed=2, kind="blobs", n=25), dict(seed=3, kind="fragments", n=60),
            dict(seed=4, kind="noise", n=0), dict(seed=5, kind="empty", n=0), dict(seed=6, kind="fragments", n=200),
            dict(seed=7, kind="stripes", n=0), dict(seed=8, kind="blobs", n=3)]


def ir_mask(case, H=480, W=640):
    """uint8 [H, W]: 0 / 255 foreground as a background subtractor would hand to detect_objects_ir."""
    rng = np.random.default_rng(1000 + case["seed"])
    img = np.zeros((H, W), np.uint8)
    yy, xx = np.mgrid[:H, :W]
    kind = case["kind"]
    if kind == "blobs":
        for _ in range(case["n"]):
            cy, cx = rng.integers(0, H), rng.integers(0, W)
            ry, rx = rng.integers(3, 60), rng.integers(3, 80)
            img[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 255
        img[rng.random((H, W)) < 0.002] = 255          # speckle the open removes or keeps by its vertical rule
    elif kind == "fragments":                           # a fragmented object: many small pieces close together
        for _ in range(case["n"]):
            cy, cx = rng.integers(60, H - 60), rng.integers(60, W - 60)
            h, w = rng.integers(1, 14), rng.integers(1, 22)
            img[cy:cy + h, cx:cx + w] = 255
    elif kind == "noise":
        img[rng.random((H, W)) < 0.08] = 255
    elif kind == "stripes":
        img[:, ::3] = 255
        img[::5, :] = 0
        img[200:260, :] = 255
    return img


class _IdentityGenerator:
    """Stand-in for numpy.random.Generator whose draws are the identity (see IdentityDraws)."""

    def shuffle(self, x):
        return None

    def choice(self, a, size=None, replace=True):
        a = np.asarray(a)
        if size is None:
            return a[0]
        size = int(size)
        if not replace:
            if size > len(a):
                raise ValueError("Cannot take a larger sample than population when replace is False")
            return a[:size].copy()
        return a[np.arange(size) % len(a)]


class IdentityDraws:
    """Context manager that pins every random draw get_segments makes (the reference's and the host port's alike,
    datasetstructures.py:1045,1165,1197,1240,1278) to the identity: np.random.default_rng(...) returns a generator
    whose shuffle leaves the order, whose choice(replace=False) takes the first k and whose choice with replacement
    cycles through the array; the global np.random.shuffle is a no-op.  cpx_plan_segments plans exactly this member of
    the reference's random family (tests/golden/make_golden_segments_identity.py)."""

    def __enter__(self):
        self._rng, self._shuffle = np.random.default_rng, np.random.shuffle
        np.random.default_rng = lambda *a, **k: _IdentityGenerator()
        np.random.shuffle = lambda x: None
        return self

    def __exit__(self, *exc):
        np.random.default_rng, np.random.shuffle = self._rng, self._shuffle
        return False


def fake_predict(x):
    """The stand-in network of the classify goldens (tests/golden/make_golden_classify.py): a deterministic function
    of the input whose rows sum to ~1."""
    x = np.asarray(x, dtype=np.float64)
    feats = np.stack([x[:, i::17, :, :].mean(axis=(1, 2, 3)) for i in range(17)], axis=1)
    e = np.exp((feats - feats.max(axis=1, keepdims=True)) / 8.0)
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
