"""Directory throughput of TrackExtractor.extract (the batched file-fed path): N copies of the two fixture recordings,
metadata + thumbnails written.  Prints one JSON line per configuration."""
import json, os, shutil, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
from cpx.config import Config
from cpx.track.trackextractor import TrackExtractor

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
batch = (None if sys.argv[2] == "auto" else int(sys.argv[2])) if len(sys.argv) > 2 else 1024  # auto: bulk.auto_batch_files
for dn in (False, True)[: (2 if "--denoise" in sys.argv else 1)]:
    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = dn
    tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") and "--disk" not in sys.argv else None)
    for i in range(N):
        for name in ("possum", "hedgehog"):
            shutil.copy(os.path.join(REPO, "tests", "golden", name + ".cptv"), os.path.join(tmp, "%s_%05d.cptv" % (name, i)))
    ex = TrackExtractor(cfg)
    ex.batch_files = batch
    t0 = time.time(); ex.extract(tmp); warm = time.time() - t0
    for f in os.listdir(tmp):
        if f.endswith(".txt"): os.remove(os.path.join(tmp, f))
    t0 = time.time(); ex.extract(tmp); dt = time.time() - t0
    tm = ex.last_run
    print(json.dumps({"denoise": dn, "files": 2 * N, "batch_files": batch, "frames": tm["frames"], "seconds": round(dt, 3),
                      "first_run_seconds": round(warm, 3), "files_per_s": round(2 * N / dt, 1),
                      "frames_per_s": round(tm["frames"] / dt, 1),
                      "split_s": {k: round(v, 3) for k, v in tm.items() if k.endswith("_s")}}), flush=True)
    shutil.rmtree(tmp)
