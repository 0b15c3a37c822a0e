"""Shared helpers for the parity tests."""
import json
import os
import zlib

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def load_clip(name):
    """All frames of a fixture CPTV -> (frames u16 [N,H,W], time_on, last_ffc, background flags, header)."""
    from cpx.cptv import CptvReader

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
