"""Build-container experiment: the unmodified reference (under the harness) vs the oracle on busy synthetic clips,
looking for the Python-int-width / float32 blank-region case."""
import os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle")); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
import numpy as np
import refharness as rh
import track_oracle as to
from helpers import encode_cptv
from cpx import synth

rh.install()
cte = rh.ref("track.cliptrackextractor"); clipmod = rh.ref("track.clip")
tmp = tempfile.mkdtemp()
ndiff = nblank = npy = ntype = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    rng = np.random.default_rng(1000 + seed)
    T = 110
    clip = synth.make_clip(rng, T, max_blobs=8)
    p = os.path.join(tmp, "c%d.cptv" % seed)
    encode_cptv(p, clip, [16] * T, time_on=[100000 + 111 * i for i in range(T)], last_ffc=[40000] * T, model=b"lepton3")
    cfg = rh.default_config(); cfg.tracking["thermal"].denoise = False
    ex = cte.ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False)
    rc = clipmod.Clip(cfg.tracking["thermal"], p)
    ex.parse_clip(rc)
    ref_tracks = sorted(list(rc.tracks) + [t for _, t in rc.filtered_tracks], key=lambda t: t.get_id())
    out = to.track_clip(clip, [100000 + 111 * i for i in range(T)], [40000] * T, None, to.OracleConfig("lepton3"), keep=True)
    mine = sorted(list(out["tracks"]) + [t for _, t in out["filtered_tracks"]], key=lambda t: t.id)
    assert [t.get_id() for t in ref_tracks] == [t.id for t in mine], seed
    firsts_r = [(int(t.bounds_history[0].x), int(t.bounds_history[0].y), t.bounds_history[0].frame_number) for t in ref_tracks]
    firsts_o = [(t.bounds[0].x, t.bounds[0].y, t.bounds[0].frame_number) for t in mine]
    if firsts_r != firsts_o:
        print("seed", seed, "skipped: the reference created same-frame tracks in another order (set iteration, SURVEY F14)")
        continue
    for rt, mt in zip(ref_tracks, mine):
        assert len(rt.bounds_history) == len(mt.bounds), (seed, rt.get_id())
        for a, b in zip(rt.bounds_history, mt.bounds):
            nblank += bool(a.blank)
            npy += (type(a.width) is int) or (type(a.height) is int)
            if ((type(a.width) is int), (type(a.height) is int)) != (b.py[2], b.py[3]):
                ntype += 1
                print("TYPE seed", seed, "track", rt.get_id(), "frame", a.frame_number, type(a.width).__name__, type(a.height).__name__, b.py, (a.x, a.y, a.width, a.height), bool(a.blank))
            if (int(a.x), int(a.y), int(a.width), int(a.height), bool(a.blank)) != (b.x, b.y, b.width, b.height, bool(b.blank)):
                ndiff += 1
                print("DIFF seed", seed, "track", rt.get_id(), "frame", a.frame_number, (a.x, a.y, a.width, a.height, type(a.width).__name__, type(a.height).__name__), (b.x, b.y, b.width, b.height))
    print("seed", seed, "tracks", len(ref_tracks), "blank regions so far", nblank, "python-int sized regions", npy, "type mismatches", ntype, "diffs", ndiff)
