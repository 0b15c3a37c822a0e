import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'classifier-pipeline_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from test_irtrack_gpu import ir_video
import cv2_shim, ir_oracle as iro, mog2_oracle as mo
from cpx.engine import TrackEngine
from cpx.track.irdetect import MOG2Background, detect_objects_ir, merge_components
frames = ir_video(5, n=50)
H, W = frames.shape[1:]
eng = TrackEngine(width=160, height=120, max_components=256, max_frames=1024)
bg = mo.MOG2(W, H, history=1000); bg.apply(frames[0], 1)
dbg = MOG2Background(eng, W, H, history=1000)
up = lambda f: torch.from_numpy(np.ascontiguousarray(f)).to(eng.device)
dbg.set_background(up(frames[0]).clone())
for q, frame in enumerate(frames):
    mask = bg.apply(frame, -1)
    fd = up(frame)
    dbg.update_background(fd, learning_rate=-1)
    dmask = dbg.compute_filtered(fd)
    eng.synchronize()
    same_mask = np.array_equal(dmask.cpu().numpy(), mask)
    small = cv2_shim.resize(mask, (160, 120), interpolation=cv2_shim.INTER_AREA)
    dsmall = eng.ir_resize_area(dmask, 4); eng.synchronize()
    same_small = np.array_equal(dsmall.cpu().numpy(), small)
    _, _, st = iro.detect_objects_ir(small, threshold=0)
    wide = torch.zeros((120, 192), dtype=torch.uint8, device=eng.device); wide[:, :160] = dsmall
    _, _, dst = detect_objects_ir(eng, wide, threshold=0, max_components=4096)
    a = sorted(tuple(int(v) for v in r) for r in st[1:]); b = sorted(tuple(int(v) for v in r) for r in dst[1:])
    m1 = iro.merge_components(list(st[1:]), 0.25); m2 = merge_components(list(dst[1:]), 0.25)
    if not (same_mask and same_small and a == b) or q == 21:
        print(q, same_mask, same_small, a == b, len(a), len(b))
        if a != b: print('  oracle', a[:6], '\n  device', b[:6])
        print('  merged oracle', [tuple(int(v) for v in m) for m in m1], ' device', [tuple(int(v) for v in m) for m in m2])
