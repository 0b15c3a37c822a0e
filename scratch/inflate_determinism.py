"""Is the inflate kernel's output the same for every copy of a file, launch after launch?  N copies of the bench's synthetic
recordings -> cpx_cptv_inflate -> every file's inflated bytes compared with zlib's (a weighted checksum on the device)."""
import json, os, sys, time, zlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
import bench
from cpx import synth
from cpx.cptv import encode_cptv
from cpx.engine import TrackEngine
from cpx.track.bulk import stage_blobs, decode_staged
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = TrackEngine(model="lepton3")
T = 270
ND = 8
host = bench.synth_on_device(torch, torch.device("cuda", 0), ND, T, seed=4321).cpu().numpy().view(np.uint16).reshape(ND, T, 120, 160)
t_on, ffc = synth.frame_times(T)
distinct = [encode_cptv(host[i], t_on, ffc, level=6) for i in range(ND)]
frames_want = [torch.from_numpy(host[i].view(np.int16)).cuda() for i in range(ND)]
blobs = [distinct[i % ND] for i in range(N)]
bad_total = 0
for rep in range(REPS):
    staged = stage_blobs(torch, blobs)
    d = decode_staged(eng, staged)
    nbad = len(d.errors)
    for g in d.groups:
        fr = g.frames_dev.view(len(g.files), T, 120, 160)
        for j in range(ND):
            sel = torch.tensor([k for k, i in enumerate(g.files) if i % ND == j], device="cuda")
            same = (fr[sel] == frames_want[j][None]).flatten(1).all(dim=1)
            nbad += int((~same).sum())
            if not bool(same.all()):
                k = int(sel[(~same).nonzero()[0, 0]])
                diff = (fr[k] != frames_want[j]).flatten(1).any(dim=1).nonzero().flatten()
                print("rep", rep, "file", g.files[k], "first bad frame", int(diff[0]), "bad frames", len(diff), flush=True)
    print(json.dumps({"rep": rep, "files": N, "bad": nbad, "errors": list(d.errors.items())[:3]}), flush=True)
    bad_total += nbad
    del d, staged
print("TOTAL BAD", bad_total)
