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
seed = int(sys.argv[1]); tmp = tempfile.mkdtemp()
rng = np.random.default_rng(1000 + seed); T = 110
clip = synth.make_clip(rng, T, max_blobs=8)
p = os.path.join(tmp, "c.cptv")
encode_cptv(p, clip, [16] * T, time_on=[100000 + 111 * i for i in range(T)], last_ffc=[40000] * T, model=b"lepton3")
cfg = rh.default_config(); cfg.tracking["thermal"].denoise = False
ex = cte.ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False)
rc = clipmod.Clip(cfg.tracking["thermal"], p)
ex.parse_clip(rc)
ref_tracks = sorted(list(rc.tracks) + [t for _, t in rc.filtered_tracks], key=lambda t: t.get_id())
out = to.track_clip(clip, [100000 + 111 * i for i in range(T)], [40000] * T, None, to.OracleConfig("lepton3"), keep=True, apply_filter=False)
mine = sorted(out["tracks"], key=lambda t: t.id)
print("ref ids", [(t.get_id(), t.start_frame, len(t.bounds_history)) for t in ref_tracks])
print("ora ids", [(t.id, t.start_frame, len(t.bounds)) for t in mine])
tid = int(sys.argv[2]) if len(sys.argv) > 2 else None
for rt in ref_tracks:
    if rt.get_id() == tid:
        for r in rt.bounds_history: print("ref", r.frame_number, r.x, r.y, r.width, r.height, r.mass, r.blank, type(r.width).__name__)
for mt in mine:
    if mt.id == tid:
        for r in mt.bounds: print("ora", r.frame_number, r.x, r.y, r.width, r.height, r.mass, r.blank)
print("first regions ref:", [(t.get_id(), (t.bounds_history[0].x, t.bounds_history[0].y, t.bounds_history[0].width, t.bounds_history[0].height, t.bounds_history[0].id)) for t in ref_tracks])
print("first regions ora:", [(t.id, (t.bounds[0].x, t.bounds[0].y, t.bounds[0].width, t.bounds[0].height, t.bounds[0].id)) for t in mine])
