"""NLM timing probe: B clips x T frames through cpx_track_batch with denoise on; wall time per frame (the NLM kernel
is ~97 % of it).  CPX_LIB selects an experiment build; no result checks."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "classifier-pipeline_amd"))
import numpy as np
import torch
from cpx import synth
from cpx.engine import TrackEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
frames, offs = synth.make_batch(64, T, seed=5)
eng = TrackEngine(model="lepton3", max_frames=T, denoise=True)
dev = torch.from_numpy(frames.view(np.int16)).cuda().repeat(B // 64, 1, 1).contiguous()
offs = (np.arange(B + 1) * T).astype(np.int32)
meta = np.concatenate([eng.make_meta(T) for _ in range(B)])
best = 1e9
for rep in range(2):
    t0 = time.perf_counter()
    eng.track_batch(dev, offs, meta, want_labels=False, want_filtered=False)
    eng.synchronize()
    best = min(best, time.perf_counter() - t0)
print(f"{os.path.basename(os.environ.get('CPX_LIB', 'shipped')):36s} B={B} T={T}: {best / (B * T) * 1e6:7.3f} us per frame")
