"""cProfile of the file-fed path (run_files_bulk with a classifier over in-memory synthetic recordings)."""
import cProfile, io, os, pstats, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
import numpy as np, torch
from cpx import synth
from cpx.classify.clipclassifier import ClipClassifier
from cpx.config import Config
from cpx.config.config import ModelConfig
from cpx.cptv import encode_cptv
from cpx.ml_tools import wrresnet as wr
from cpx.track.bulk import run_files_bulk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 270
classify = "--no-classify" not in sys.argv
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 2048
labels = ["bird", "cat", "deer", "dog", "false-positive", "hedgehog", "human", "kiwi", "leporidae", "mustelid", "penguin", "possum", "rodent", "sheep", "vehicle", "wallaby", "land-bird"]
tmp = tempfile.mkdtemp()
from cpx.engine import TrackEngine
_e = TrackEngine()
_w = wr.calibrate_bn_device(_e, wr.random_weights(17, seed=0), torch.rand((32, 160, 160, 2), device=_e.device) * 255)
_e.close()
wr.save_model(os.path.join(tmp, "wr"), _w, labels, hyperparams={"frame_size": 32})
cfg = Config.get_defaults(); cfg.tracking["thermal"].denoise = False
cfg.classify.models = [ModelConfig.load({"id": 1, "name": "wr-bench", "model_file": os.path.join(tmp, "wr.npz")})]
sys.path.insert(0, REPO)
import bench
t_on, ffc = synth.frame_times(T)
ND = 32
host = bench.synth_on_device(torch, torch.device("cuda", 0), ND, T, seed=4321).cpu().numpy().view(np.uint16).reshape(ND, T, 120, 160)
distinct = [encode_cptv(host[i], t_on, ffc, level=6) for i in range(ND)]
blobs = [distinct[i % ND] for i in range(N)]
if "--fixtures" in sys.argv:
    real = [open(os.path.join(REPO, "tests", "golden", f + ".cptv"), "rb").read() for f in ("possum", "hedgehog")]
    blobs = [real[i % 2] for i in range(N)]
names = ["s%05d.cptv" % i for i in range(N)]
cc = ClipClassifier(cfg) if classify else None
run_files_bulk(names, cfg, save_meta=False, want_text=True, batch_files=min(N, BATCH), clip_classifier=cc, blobs=blobs)
pr = cProfile.Profile()
torch.cuda.synchronize()
t0 = time.time()
if "--cprofile" in sys.argv:
    pr.enable()
out, tr = run_files_bulk(names, cfg, save_meta=False, want_text=True, batch_files=min(N, BATCH), clip_classifier=cc, blobs=blobs)
pr.disable()
from cpx.track import cliptrackextractor as _cte
print("fp16 overflow rerun in the last forward:", [e.cnn_last_overflow() for e in _cte._ENGINES.values()], [e.get_cnn_math() for e in _cte._ENGINES.values()])
print("seconds", round(time.time() - t0, 3), {k: round(v, 3) if isinstance(v, float) else v for k, v in tr.timings.items()})
if "--cprofile" in sys.argv:
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(28); print(st.getvalue()[:6000])
