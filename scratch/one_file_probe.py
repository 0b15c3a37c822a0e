"""Single recording through extract_file, denoise off / on: where the wall time goes (extractor timings)."""
import json, os, shutil, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
from cpx.config import Config
from cpx.track.trackextractor import extract_file
tmp = tempfile.mkdtemp()
dst = os.path.join(tmp, "possum.cptv")
shutil.copy(os.path.join(REPO, "tests", "golden", "possum.cptv"), dst)
for dn in (False, True):
    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = dn if hasattr(cfg.tracking["thermal"], "denoise") else dn
    for k in cfg.tracking.values():
        k.denoise = dn
    extract_file(dst, cfg, False, save_meta=False)
    best = None
    for _ in range(5):
        t0 = time.time()
        clip, ex, meta = extract_file(dst, cfg, False, save_meta=False)
        dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, dict(ex.timings))
    print("denoise", dn, "wall %.1f ms" % (best[0] * 1e3), {k: round(v * 1e3, 2) for k, v in best[1].items()})
