#!/usr/bin/env python3
"""Generates the per-stage golden vectors in tests/golden/ by RUNNING THE
REFERENCE ITSELF (imported from /root/reference/src under the harness in
oracle/refharness.py) on the reference's own fixture clips.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

Outputs, per clip and denoise setting, ``<clip>_dn<0|1>.npz`` holding
  * per-frame scalars: avg_change inputs/outputs, normalisation min/max, mapped
    threshold, background average, ffc flag, component count;
  * per-frame CRC32 of every full-frame intermediate (bit-exact check for all
    frames without storing them);
  * the full-frame intermediates for a subset of frames;
  * all component stats / centroids, all regions of interest;
and ``<clip>_dn<0|1>_tracks.json`` with the final tracks / filter reasons.
"""

import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import refharness as rh  # noqa: E402


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def run(clip_name, denoise, keep_frames, path=None):
    rh.install()
    cte = rh.ref("track.cliptrackextractor")
    rec = {"frames": []}
    cur = {}

    orig_detect = cte.detect_objects

    def detect(image, otsus=False, threshold=30, kernel=(15, 15)):
        out = orig_detect(image, otsus=otsus, threshold=threshold, kernel=kernel)
        cur["obj_filtered"] = np.array(image, copy=True)
        cur["threshold"] = float(threshold)
        cur["threshold_dtype"] = str(np.asarray(threshold).dtype)
        cur["n"] = int(out[0])
        cur["mask"] = out[1].copy()
        cur["stats"] = out[2].copy()
        cur["centroids"] = out[3].copy()
        return out

    cte.detect_objects = detect
    try:

        def frame_hook(clip, ex, idx):
            bgalg = ex.background_alg
            f = clip.frame_buffer.current_frame
            d = dict(cur)
            cur.clear()
            d["index"] = idx
            d["ffc"] = bool(clip.ffc_affected)
            d["bg_used_avg"] = float(bgalg.get_average())
            d["filtered"] = np.array(f.filtered, copy=True)
            d["thermal"] = np.array(f.thermal, copy=True)
            regs = clip.region_history[-1] if clip.region_history else []
            d["regions"] = [
                (
                    int(r.x), int(r.y), int(r.width), int(r.height), int(r.mass), int(r.id),
                    float(r.pixel_variance), bool(r.was_cropped), bool(r.is_along_border),
                    float(r.centroid[0]), float(r.centroid[1]),
                )
                for r in regs
            ]
            rec["frames"].append(d)

        # background state *after* each update is captured by wrapping process_frame of the bg alg
        clip, ex = None, None
        cfg = rh.default_config()
        cfg.tracking["thermal"].denoise = bool(denoise)
        clipmod = rh.ref("track.clip")
        ex = cte.ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False)
        clip = clipmod.Clip(cfg.tracking["thermal"], path or os.path.join(HERE, clip_name + ".cptv"))
        orig_pf = ex.process_frame

        def wrapped(c, fr):
            r = orig_pf(c, fr)
            frame_hook(c, ex, c.current_frame)
            return r

        ex.process_frame = wrapped
        ex.init_clip(clip)
        bgalg = ex.background_alg
        rec["init_bg"] = np.array(bgalg.background, copy=True)
        rec["init_avg"] = float(bgalg.get_average())
        orig_bg_pf = bgalg.process_frame
        bg_states = []

        def bg_pf(frame):
            orig_bg_pf(frame)
            bg_states.append(
                (
                    np.array(bgalg.background, copy=True),
                    np.array(bgalg.background_weight, copy=True),
                    float(bgalg.get_average()),
                )
            )

        bgalg.process_frame = bg_pf
        ex._track_clip(clip)
        clip.stats.completed()
    finally:
        cte.detect_objects = orig_detect

    frames = rec["frames"]
    n = len(frames)
    assert len(bg_states) == n
    out = {}
    out["init_bg"] = rec["init_bg"].astype(np.int32)
    out["init_avg"] = np.float64(rec["init_avg"])
    out["n_frames"] = np.int32(n)
    out["ffc"] = np.array([f["ffc"] for f in frames], dtype=np.uint8)
    out["bg_used_avg"] = np.array([f["bg_used_avg"] for f in frames], dtype=np.float64)
    out["bg_after_avg"] = np.array([s[2] for s in bg_states], dtype=np.float64)
    out["threshold"] = np.array([f["threshold"] for f in frames], dtype=np.float64)
    out["n_components"] = np.array([f["n"] - 1 for f in frames], dtype=np.int32)
    out["crc_filtered"] = np.array([crc(f["filtered"].astype(np.int32)) for f in frames], dtype=np.uint32)
    out["crc_obj_u8"] = np.array([crc(np.uint8(f["obj_filtered"])) for f in frames], dtype=np.uint32)
    out["crc_mask"] = np.array([crc(f["mask"].astype(np.int32)) for f in frames], dtype=np.uint32)
    out["crc_bg_after"] = np.array([crc(s[0].astype(np.int32)) for s in bg_states], dtype=np.uint32)
    out["crc_weight_after"] = np.array([crc(s[1].astype(np.float64)) for s in bg_states], dtype=np.uint32)
    # float image handed to detect_objects (pre-uint8) checksum only when denoise is off
    stats_off = [0]
    stats = []
    cents = []
    for f in frames:
        stats.append(f["stats"][1:])
        cents.append(f["centroids"][1:])
        stats_off.append(stats_off[-1] + f["n"] - 1)
    out["comp_offsets"] = np.array(stats_off, dtype=np.int32)
    out["comp_stats"] = np.concatenate(stats, axis=0).astype(np.int32) if stats else np.zeros((0, 5), np.int32)
    out["comp_centroids"] = np.concatenate(cents, axis=0).astype(np.float64) if cents else np.zeros((0, 2))
    reg_off = [0]
    regs = []
    for f in frames:
        regs.extend(f["regions"])
        reg_off.append(reg_off[-1] + len(f["regions"]))
    out["region_offsets"] = np.array(reg_off, dtype=np.int32)
    out["regions"] = np.array(regs, dtype=np.float64).reshape(-1, 11)
    keep = sorted(set(k for k in keep_frames if k < n))
    out["kept"] = np.array(keep, dtype=np.int32)
    out["kept_filtered"] = np.stack([frames[k]["filtered"].astype(np.int32) for k in keep])
    out["kept_obj_u8"] = np.stack([np.uint8(frames[k]["obj_filtered"]) for k in keep])
    out["kept_mask"] = np.stack([frames[k]["mask"].astype(np.int16) for k in keep])
    out["kept_bg_after"] = np.stack([bg_states[k][0].astype(np.int32) for k in keep])
    out["kept_weight_after"] = np.stack([bg_states[k][1].astype(np.float64) for k in keep])
    # clip stats (a8)
    st = clip.stats
    out["stats_median"] = np.array(st.frame_stats_median, dtype=np.float64)
    out["stats_min"] = np.array(st.frame_stats_min, dtype=np.float64)
    out["stats_max"] = np.array(st.frame_stats_max, dtype=np.float64)
    out["stats_mean"] = np.array(st.frame_stats_mean, dtype=np.float64)
    out["stats_filtered_sum"] = np.float64(st.filtered_sum)
    np.savez_compressed(os.path.join(HERE, "%s_dn%d.npz" % (clip_name, int(denoise))), **out)

    # tracks before filtering -> after filtering
    ex.apply_track_filtering(clip) if False else None  # (_track_clip already applied it)
    tracks = []
    for t in clip.tracks:
        tracks.append(
            {
                "id": t.get_id(),
                "start_frame": int(t.start_frame),
                "end_frame": int(t.end_frame),
                "score": float(t.stats.score),
                "stats": {k: float(v) for k, v in t.stats._asdict().items()},
                "positions": [
                    {
                        "x": int(r.x), "y": int(r.y), "width": int(r.width), "height": int(r.height),
                        "mass": int(r.mass), "frame_number": int(r.frame_number),
                        "pixel_variance": float(r.pixel_variance), "blank": bool(r.blank),
                        "centroid": [float(r.centroid[0]), float(r.centroid[1])],
                        "is_along_border": bool(r.is_along_border), "was_cropped": bool(r.was_cropped),
                    }
                    for r in t.bounds_history
                ],
                "vel_x": [float(v) for v in t.vel_x],
                "vel_y": [float(v) for v in t.vel_y],
            }
        )
    filt = [
        {"reason": reason, "id": t.get_id(), "start_frame": int(t.start_frame), "n": len(t)}
        for reason, t in clip.filtered_tracks
    ]
    with open(os.path.join(HERE, "%s_dn%d_tracks.json" % (clip_name, int(denoise))), "w") as fh:
        json.dump({"tracks": tracks, "filtered": filt, "ffc_frames": [int(x) for x in clip.ffc_frames]}, fh, indent=1)
    print(clip_name, "denoise", denoise, "frames", n, "tracks", [(t["id"], t["start_frame"], t["end_frame"]) for t in tracks])


if __name__ == "__main__":
    keep = list(range(0, 200, 16)) + list(range(38, 74, 3))
    for name in ("possum", "hedgehog"):
        for dn in (0, 1):
            run(name, dn, keep)
