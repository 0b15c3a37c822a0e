"""run_files_bulk over N copies of 32 synthetic recordings, REPS times: files skipped with an error, and metadata texts of
copies of one recording that differ from each other (they must be identical but for the file name and timings)."""
import json, os, re, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
import bench
from cpx import synth
from cpx.classify.clipclassifier import ClipClassifier
from cpx.config import Config
from cpx.config.config import ModelConfig
from cpx.cptv import encode_cptv
from cpx.ml_tools import wrresnet as wr
from cpx.track.bulk import run_files_bulk
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T = 270
labels = ["bird", "cat", "deer", "dog", "false-positive", "hedgehog", "human", "kiwi", "leporidae", "mustelid", "penguin", "possum", "rodent", "sheep", "vehicle", "wallaby", "land-bird"]
tmp = tempfile.mkdtemp()
wr.save_model(os.path.join(tmp, "wr"), wr.random_weights(17, seed=0), labels, hyperparams={"frame_size": 32})
cfg = Config.get_defaults(); cfg.tracking["thermal"].denoise = False
cfg.classify.models = [ModelConfig.load({"id": 1, "name": "wr-bench", "model_file": os.path.join(tmp, "wr.npz")})]
cfg.classify.meta_to_stdout = False
ND = 32
host = bench.synth_on_device(torch, torch.device("cuda", 0), ND, T, seed=4321).cpu().numpy().view(np.uint16).reshape(ND, T, 120, 160)
t_on, ffc = synth.frame_times(T)
distinct = [encode_cptv(host[i], t_on, ffc, level=6) for i in range(ND)]
if "--fixtures" in sys.argv:
    distinct = [open(os.path.join(REPO, "tests", "golden", f + ".cptv"), "rb").read() for f in ("possum", "hedgehog")]
    ND = 2
blobs = [distinct[i % ND] for i in range(N)]
names = ["s%05d.cptv" % i for i in range(N)]
cc = None if "--no-classify" in sys.argv else ClipClassifier(cfg)
def norm(text):
    text = re.sub(r'"(tracking_time|classify_time|predicted_time|source|file|original_tag_time)": [^,\n]*', '"x": 0', text)
    text = re.sub(r'^    "id": \d+,', '    "id": 0,', text, flags=re.M)
    return re.sub(r's\d{5}\.cptv', 'F', text)
for rep in range(REPS):
    if "--empty-cache" in sys.argv:
        torch.cuda.empty_cache()
    t0 = time.time()
    out, tr = run_files_bulk(names, cfg, save_meta=False, want_text=True, batch_files=2048, clip_classifier=cc, blobs=blobs)
    dt = time.time() - t0
    errs = {k: v[:80] for k, v in out.items() if v.startswith("error")}
    ref = {}
    differ = []
    for i, nme in enumerate(names):
        if nme in errs: continue
        t = norm(out[nme])
        if i % ND not in ref: ref[i % ND] = (t, nme)
        elif ref[i % ND][0] != t: differ.append(nme)
    if differ and rep == 0 and "--show-diff" in sys.argv:
        a, b = ref[names.index(differ[0]) % ND][0].splitlines(), norm(out[differ[0]]).splitlines()
        print([(x, y) for x, y in zip(a, b) if x != y][:6], len(a), len(b))
    print(json.dumps({"rep": rep, "seconds": round(dt, 2), "errors": errs, "n_differ": len(differ), "differ": differ[:5]}), flush=True)
