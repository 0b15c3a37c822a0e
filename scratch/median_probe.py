"""Median-kernel timing probe: B clips x T frames through cpx_track_batch; wall time of the whole call minus the frame
kernel's event time ~ medians + init.  CPX_LIB selects an experiment build."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "classifier-pipeline_amd"))
import numpy as np
import torch
from cpx import synth
from cpx.engine import TrackEngine

B, T = 4096, 120
frames, offs = synth.make_batch(64, T, seed=5)
eng = TrackEngine(model="lepton3", max_frames=T)
dev = torch.from_numpy(frames.view(np.int16)).cuda().repeat(B // 64, 1, 1).contiguous()
offs = (np.arange(B + 1) * T).astype(np.int32)
meta = np.concatenate([eng.make_meta(T) for _ in range(B)])
best = 1e9
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.track_batch(dev, offs, meta, want_labels=False, want_filtered=True)
    eng.synchronize()
    dt = time.perf_counter() - t0
    ms, n = eng.last_kernel_timing()
    best = min(best, dt * 1e3 - ms)
print(f"{os.path.basename(os.environ.get('CPX_LIB', 'shipped')):32s} medians + init + host: {best:7.2f} ms for {B * T} frames = {best * 1e6 / (B * T):6.2f} ns per frame")
