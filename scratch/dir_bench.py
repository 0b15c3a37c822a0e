"""Directory throughput of the drop-in drivers: N copies of the fixture clips through TrackExtractor.extract (device
batches of 64 files) vs extract_file one by one.  Prints one JSON line."""
import json, os, shutil, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
from cpx.config import Config
from cpx.track.trackextractor import TrackExtractor, extract_file

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out = {}
for dn in (False, True):
    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = dn
    tmp = tempfile.mkdtemp()
    frames = 0
    for i in range(N):
        for name, n in (("possum", 160), ("hedgehog", 119)):
            shutil.copy(os.path.join(REPO, "tests", "golden", name + ".cptv"), os.path.join(tmp, "%s_%03d.cptv" % (name, i)))
            frames += n
    files = sorted(f for f in os.listdir(tmp) if f.endswith(".cptv"))
    extract_file(os.path.join(tmp, files[0]), cfg, False, save_meta=False)  # warm-up
    t0 = time.time()
    for f in files[:16]:
        extract_file(os.path.join(tmp, f), cfg, False)
    one = (time.time() - t0) / 16
    ex = TrackExtractor(cfg)
    ex.extract(os.path.join(tmp))  # warm-up of the batch shapes
    t0 = time.time()
    ex.extract(os.path.join(tmp))
    dt = time.time() - t0
    out["denoise_%s" % ("on" if dn else "off")] = {
        "files": len(files), "frames": frames, "batched_s": round(dt, 3), "batched_files_per_s": round(len(files) / dt, 1),
        "batched_frames_per_s": round(frames / dt, 1), "one_by_one_s_per_file": round(one, 4),
        "one_by_one_files_per_s": round(1 / one, 1)}
    shutil.rmtree(tmp)
print(json.dumps({"config": "TrackExtractor.extract(directory) with metadata + thumbnails written, default config", "runs": out}))
