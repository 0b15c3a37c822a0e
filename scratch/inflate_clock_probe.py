import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CPX_LIB"] = os.path.join(REPO, "scratch", "bin", "libcpx_clockprobe.so")
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
import numpy as np, torch, time
from cpx.engine import TrackEngine
from cpx.track.bulk import stage_blobs, decode_staged
eng = TrackEngine(model="lepton3")
raw = open(os.path.join(REPO, "tests", "golden", "possum.cptv"), "rb").read()
for n in (64, 1024, 4096):
    st = stage_blobs(torch, [raw] * n)
    d = decode_staged(eng, st); t0 = time.time(); d = decode_staged(eng, st); dt = time.time() - t0
    mhz = d.results["reserved"]
    print(n, "files:", round(dt, 3), "s; wave clock MHz min/median/max", int(mhz.min()), int(np.median(mhz)), int(mhz.max()))
