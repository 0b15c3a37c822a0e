"""Frame-kernel timing probe: B clips x T frames through cpx_track_batch, mean launch time by HIP events
(cpx_last_kernel_timing).  CPX_LIB selects an experiment build (scratch/build_variant.sh); no result checks --
the -DCPX_TIMING_STOP_AFTER builds return garbage on purpose."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "classifier-pipeline_amd"))
import numpy as np
import torch
from cpx import synth
from cpx.engine import TrackEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
labels = len(sys.argv) > 3 and sys.argv[3] == "labels"
frames, offs = synth.make_batch(64, T, seed=5)
eng = TrackEngine(model="lepton3", max_frames=T)
one = torch.from_numpy(frames.view(np.int16)).cuda()
dev = one.repeat(B // 64, 1, 1).contiguous()
offs = (np.arange(B + 1) * T).astype(np.int32)
meta = np.concatenate([eng.make_meta(T) for _ in range(B)])
best = None
for rep in range(3):
    res = eng.track_batch(dev, offs, meta, want_labels=labels, want_filtered=True)
    eng.synchronize()
    ms, n = eng.last_kernel_timing()
    best = ms / T if best is None else min(best, ms / T)
print(f"{os.path.basename(os.environ.get('CPX_LIB', 'shipped')):40s} per_step={os.environ.get('CPX_TRACK_PER_STEP', '0')} B={B} T={T} labels={labels}: {best * 1e3:8.1f} us per frame step ({n} launches)")
