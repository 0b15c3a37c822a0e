"""Whole cpx_track_batch wall time (records' memset + medians + frame kernel), B clips x T frames; CPX_MEDIAN_BESIDE = 0
(medians in front of the frame kernel), 1 (on the second stream, enqueued behind it), 2 (enqueued first)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "classifier-pipeline_amd"))
import numpy as np
import torch
from cpx import synth
from cpx.engine import TrackEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 270
frames, offs = synth.make_batch(64, T, seed=5)
eng = TrackEngine(model="lepton3", max_frames=T)
one = torch.from_numpy(frames.view(np.int16)).cuda()
dev = one.repeat(B // 64, 1, 1).contiguous()
offs = (np.arange(B + 1) * T).astype(np.int32)
meta = np.concatenate([eng.make_meta(T) for _ in range(B)])
best = None
med = None
for rep in range(4):
    torch.cuda.synchronize(); eng.synchronize()
    t0 = time.perf_counter()
    res = eng.track_batch(dev, offs, meta)
    eng.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    ms, n = eng.last_kernel_timing()
    if best is None or dt < best[0]:
        best = (dt, ms)
    info = res.info
    m = np.asarray(info["thermal_median"]).copy()
    assert med is None or np.array_equal(m, med)
    med = m
print(f"CPX_MEDIAN_BESIDE={os.environ.get('CPX_MEDIAN_BESIDE', 'default')} B={B} T={T}: call {best[0]:7.2f} ms, frame kernel (events) {best[1]:7.2f} ms, median checksum {float(med.astype(np.float64).sum()):.1f}")
