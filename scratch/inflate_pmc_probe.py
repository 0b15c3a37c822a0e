"""cpx_cptv_inflate_kernel for the PMC passes (scratch/pmc_inflate.sh): N copies of one synthetic recording of bench.py's
from_files workload (270 frames, sensor noise, gzip level 6), two launches."""
import json, os, sys, time, zlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
import bench
from cpx import synth
from cpx.cptv import encode_cptv
from cpx.engine import TrackEngine
from cpx.track.bulk import stage_blobs, inflate_launch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "synthetic"
eng = TrackEngine(model="lepton3")
if kind == "fixture":
    blobs = [open(os.path.join(REPO, "tests", "golden", f + ".cptv"), "rb").read() for f in ("possum", "hedgehog")]
else:
    T = 270
    host = bench.synth_on_device(torch, torch.device("cuda", 0), 1, T, seed=4321).cpu().numpy().view(np.uint16).reshape(1, T, 120, 160)
    t_on, ffc = synth.frame_times(T)
    blobs = [encode_cptv(host[0], t_on, ffc, level=6)]
inflated = [len(zlib.decompress(b, 47)) for b in blobs]
staged = stage_blobs(torch, [blobs[i % len(blobs)] for i in range(N)])
staged.paths = ["f%d" % i for i in range(N)]
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx = inflate_launch(eng, staged); ctx["done"].synchronize(); dt = time.perf_counter() - t0
out_bytes = sum(inflated[i % len(blobs)] for i in range(N)); in_bytes = sum(len(blobs[i % len(blobs)]) for i in range(N))
print(json.dumps({"files": N, "kind": kind, "compressed_bytes": in_bytes, "inflated_bytes": out_bytes, "launch_s": round(dt, 4),
                  "GBps_out": round(out_bytes / dt / 1e9, 2)}))
