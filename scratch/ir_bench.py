"""Timing of cpx_ir_detect on 640x480 foreground masks (detection stage only)."""
import json, sys, time
sys.path.insert(0, "classifier-pipeline_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np, torch
from cpx.engine import TrackEngine
from helpers import IR_CASES, ir_mask

eng = TrackEngine(model="lepton3")
res = {}
for kind, cases in (("blobs", [c for c in IR_CASES if c["kind"] == "blobs"]), ("fragments", [c for c in IR_CASES if c["kind"] == "fragments"]),
                    ("noise", [c for c in IR_CASES if c["kind"] == "noise"])):
    imgs = np.stack([ir_mask(c) for c in cases])
    n = 2048
    batch = torch.from_numpy(imgs[np.arange(n) % len(imgs)]).to(eng.device)
    cap = 20000 if kind == "noise" else 1024
    for labels in (False, True):
        eng.ir_detect(batch, 0, cap, want_labels=labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.ir_detect(batch, 0, cap, want_labels=labels)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        res["%s%s" % (kind, "+labels" if labels else "")] = {"frames_per_s": n / dt, "ms_per_batch": dt * 1e3, "batch": n,
                                                            "input_GBps": n * 640 * 480 / dt / 1e9}
# kernel only (no result download): the C-ABI call in a loop
import ctypes as C
from cpx._lib import COMPONENT_DTYPE
for kind in ("blobs", "noise"):
    cases = [c for c in IR_CASES if c["kind"] == kind]
    imgs = np.stack([ir_mask(c) for c in cases])
    n = 2048
    batch = torch.from_numpy(imgs[np.arange(n) % len(imgs)]).to(eng.device)
    cap = 20000 if kind == "noise" else 1024
    comps = torch.empty((n, cap, 8), dtype=torch.int32, device=eng.device)
    counts = torch.zeros(n, dtype=torch.int32, device=eng.device)
    status = torch.zeros(n, dtype=torch.int32, device=eng.device)
    labels = torch.empty((n, 480, 640), dtype=torch.int32, device=eng.device)
    for lab in (None, labels):
        def call():
            rc = eng.lib.cpx_ir_detect(eng.h, C.c_void_p(batch.data_ptr()), n, 640, 480, 0, cap, C.c_void_p(comps.data_ptr()),
                                       C.c_void_p(counts.data_ptr()), C.c_void_p(status.data_ptr()),
                                       C.c_void_p(lab.data_ptr()) if lab is not None else None)
            assert rc == 0
        torch.cuda.synchronize(); call(); eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            call()
        eng.synchronize()
        dt = (time.perf_counter() - t0) / 5
        res["kernel_%s%s" % (kind, "+labels" if lab is not None else "")] = {"frames_per_s": n / dt, "ms_per_batch": dt * 1e3,
                                                                           "input_GBps": n * 640 * 480 / dt / 1e9}
# MOG2 background model: 64 streams of 640x480 in lockstep, kernel only
from cpx.track.irdetect import MOG2Background
S = 64
bgm = MOG2Background(eng, 640, 480, n_streams=S)
vid = (torch.rand((8, S, 480, 640), device=eng.device) * 6 + 100).to(torch.uint8)
mask = torch.empty((S, 480, 640), dtype=torch.uint8, device=eng.device)
for t in range(8):
    bgm.update_background(vid[t])
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 40
for r in range(reps):
    rc = eng.lib.cpx_mog2_apply(bgm._m, C.c_void_p(vid[r % 8].data_ptr()), -1.0, C.c_void_p(mask.data_ptr()))
    assert rc == 0
eng.synchronize()
dt = (time.perf_counter() - t0) / reps
res["kernel_mog2_apply"] = {"frames_per_s": S / dt, "ms_per_batch": dt * 1e3, "streams": S,
                            "algorithmic_GBps": S * 640 * 480 * 124 / dt / 1e9}
bgm.close()
import mog2_oracle as mo
om = mo.MOG2(640, 480)
fr = vid[:, 0].cpu().numpy()
t0 = time.perf_counter()
for t in range(8):
    om.apply(fr[t])
res["mog2_oracle_cpu_frames_per_s"] = 8 / (time.perf_counter() - t0)
# CPU: the oracle (numpy restatement) on the same masks
import ir_oracle as iro
t0 = time.perf_counter()
for c in IR_CASES[:3]:
    iro.detect_objects_ir(ir_mask(c), 0)
res["oracle_cpu_frames_per_s"] = 3 / (time.perf_counter() - t0)
print(json.dumps(res, indent=1))
json.dump(res, open("gpurun_out/ir_bench.json", "w"), indent=1)
