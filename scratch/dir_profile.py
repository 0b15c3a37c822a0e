import os, shutil, sys, tempfile, cProfile, pstats
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
from cpx.config import Config
from cpx.track.trackextractor import TrackExtractor
cfg = Config.get_defaults(); cfg.tracking["thermal"].denoise = False
tmp = tempfile.mkdtemp()
for i in range(32):
    for name in ("possum", "hedgehog"):
        shutil.copy(os.path.join(REPO, "tests", "golden", name + ".cptv"), os.path.join(tmp, "%s_%03d.cptv" % (name, i)))
ex = TrackExtractor(cfg); ex.extract(tmp)
pr = cProfile.Profile(); pr.enable(); ex.extract(tmp); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
