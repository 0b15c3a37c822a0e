import os, sys, time, shutil, tempfile, cProfile, pstats
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
import numpy as np
from cpx.config import Config
from cpx.config.config import ModelConfig
from cpx.ml_tools import wrresnet as wr
from cpx.classify.clipclassifier import ClipClassifier
LABELS = ["bird", "cat", "deer", "dog", "false-positive", "hedgehog", "human", "kiwi", "leporidae", "mustelid",
          "penguin", "possum", "rodent", "sheep", "vehicle", "wallaby", "land-bird"]
tmp = tempfile.mkdtemp()
wr.save_model(os.path.join(tmp, "wr"), wr.random_weights(len(LABELS), seed=1), LABELS, hyperparams={"frame_size": 32})
cfg = Config.get_defaults()
cfg.classify.models = [ModelConfig.load({"id": 1, "name": "wr", "model_file": os.path.join(tmp, "wr.npz")})]
dst = os.path.join(tmp, "possum.cptv"); shutil.copy(os.path.join(REPO, "tests", "golden", "possum.cptv"), dst)
cc = ClipClassifier(cfg)
cc.process_file(dst, track=True, calculate_thumbnails=True)
t0 = time.time()
for _ in range(3): cc.process_file(dst, track=True, calculate_thumbnails=True)
print("process_file(track=True) default config: %.3f s per file" % ((time.time() - t0) / 3))
pr = cProfile.Profile(); pr.enable(); cc.process_file(dst, track=True, calculate_thumbnails=True); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
