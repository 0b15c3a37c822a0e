import os, sys, time, shutil, tempfile, cProfile, pstats
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
from cpx.config import Config
from cpx.track.trackextractor import extract_file
tmp = tempfile.mkdtemp()
dst = os.path.join(tmp, "possum.cptv"); shutil.copy(os.path.join(REPO, "tests", "golden", "possum.cptv"), dst)
for dn in (False, True):
    cfg = Config.get_defaults(); cfg.tracking["thermal"].denoise = dn
    extract_file(dst, cfg, False, save_meta=False)
    t0 = time.time()
    for _ in range(3): clip, ex, meta = extract_file(dst, cfg, False, save_meta=False)
    print("denoise", dn, "wall/clip %.3f s" % ((time.time() - t0) / 3), ex.timings)
cfg = Config.get_defaults(); cfg.tracking["thermal"].denoise = False
pr = cProfile.Profile(); pr.enable()
extract_file(dst, cfg, False, save_meta=False)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
