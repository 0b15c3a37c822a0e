"""Which forwards of the fixture file-fed run leave fp16's range (CPX_CNN_DEBUG_OVF=1 prints the blocks)?"""
import os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
import bench
from cpx.classify.clipclassifier import ClipClassifier
from cpx.config import Config
from cpx.config.config import ModelConfig
from cpx.cptv import CptvReader
from cpx.engine import TrackEngine
from cpx.ml_tools import wrresnet as wr
from cpx.track.bulk import run_files_bulk
gold = os.path.join(REPO, "tests", "golden")
eng = TrackEngine(model="lepton3", max_frames=512)
fr, lens, metas = [], [], []
for name in ("possum", "hedgehog"):
    fs = CptvReader(os.path.join(gold, name + ".cptv")).read_all()
    fr.append(np.stack([f.pix for f in fs]).astype(np.uint16)); lens.append(len(fs))
    metas.append(eng.make_meta(len(fs), [f.time_on for f in fs], [f.last_ffc_time for f in fs], [bool(f.background_frame) for f in fs]))
offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
w = bench.synthetic_network_weights(torch, wr, eng, eng.upload_frames(np.concatenate(fr)), offs, np.concatenate(metas), None, 32, n_clips=2)
net = wr.WRResNetDevice(eng, w, 17)
print("bounds", [round(b, 1) for b in net.act_bounds])
eng.close()
tmp = tempfile.mkdtemp()
labels = ["l%d" % i for i in range(17)]; labels[4] = "false-positive"
wr.save_model(os.path.join(tmp, "wr"), w, labels, hyperparams={"frame_size": 32})
cfg = Config.get_defaults(); cfg.tracking["thermal"].denoise = False
cfg.classify.models = [ModelConfig.load({"id": 1, "name": "wr", "model_file": os.path.join(tmp, "wr.npz")})]
real = [open(os.path.join(gold, f + ".cptv"), "rb").read() for f in ("possum", "hedgehog")]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
out, tr = run_files_bulk(["f%05d.cptv" % i for i in range(N)], cfg, save_meta=False, want_text=True, batch_files=int(os.environ.get("BATCH", N)),
                         clip_classifier=ClipClassifier(cfg), blobs=[real[i % 2] for i in range(N)])
import time as _t
for rep in range(2):
    t0 = _t.time(); out, tr2 = run_files_bulk(["f%05d.cptv" % i for i in range(N)], cfg, save_meta=False, want_text=True, batch_files=int(os.environ.get("BATCH", N)), clip_classifier=ClipClassifier(cfg), blobs=[real[i % 2] for i in range(N)]); torch.cuda.synchronize(); print("pass", rep, round(_t.time() - t0, 3), {k: round(v, 3) for k, v in tr2.timings.items() if isinstance(v, float)})
