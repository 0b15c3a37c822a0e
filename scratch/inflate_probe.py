"""GPU inflate throughput: N copies of the fixture recordings through cpx_cptv_inflate (kernel time by HIP events
around the call on the handle's stream), for several batch sizes.  Prints one JSON line per batch size."""
import ctypes as C, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
import numpy as np, torch
from cpx._lib import CPTV_FILE_DTYPE, CPTV_RESULT_DTYPE, CPTV_HEADER_BYTES
from cpx.engine import TrackEngine

eng = TrackEngine(model="lepton3")
dev = eng.device
blobs0 = [open(os.path.join(REPO, "tests", "golden", n + ".cptv"), "rb").read() for n in ("possum", "hedgehog")]
frames0 = [161, 120]
P = 160 * 120
for N in [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 4096]:
    blobs = [blobs0[i % 2] for i in range(N)]
    nfr = sum(frames0[i % 2] for i in range(N))
    sizes = np.array([len(b) for b in blobs], np.int64)
    in_off = np.zeros(N + 1, np.int64); np.cumsum((sizes + 15) & ~15, out=in_off[1:])
    isize = np.array([int.from_bytes(b[-4:], "little") for b in blobs], np.int64)
    out_off = np.zeros(N + 1, np.int64); np.cumsum(((isize + 15) & ~15) + 16, out=out_off[1:])
    slot_cap = isize // 2400 + 1
    slot_off = np.zeros(N + 1, np.int64); np.cumsum(slot_cap, out=slot_off[1:])
    files = np.zeros(N, CPTV_FILE_DTYPE)
    files["in_offset"], files["in_bytes"], files["out_offset"], files["out_capacity"] = in_off[:-1], sizes, out_off[:-1], isize
    files["slot_offset"], files["slot_capacity"] = slot_off[:-1], slot_cap
    stage = torch.empty(int(in_off[-1]) + 16, dtype=torch.uint8, pin_memory=True)
    sv = stage.numpy()
    for i, b in enumerate(blobs):
        sv[in_off[i]:in_off[i] + sizes[i]] = np.frombuffer(b, np.uint8)
    t0 = time.perf_counter()
    in_dev = stage.to(dev, non_blocking=True); torch.cuda.synchronize()
    h2d = time.perf_counter() - t0
    files_dev = eng._to_dev(files)
    out_dev = torch.empty(int(out_off[-1]) + 16, dtype=torch.uint8, device=dev)
    slots_dev = torch.empty(int(slot_off[-1]) * 8, dtype=torch.int32, device=dev)
    header_dev = torch.empty((N, CPTV_HEADER_BYTES), dtype=torch.uint8, device=dev)
    results_dev = torch.zeros(N * 10, dtype=torch.int32, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        rc = eng.lib.cpx_cptv_inflate(eng.h, p(in_dev), p(files_dev), N, p(out_dev), p(slots_dev), p(header_dev), p(results_dev))
        assert rc == 0
        eng.synchronize()
        best = min(best, time.perf_counter() - t0)
    res = results_dev.cpu().numpy().view(CPTV_RESULT_DTYPE)
    assert (res["status"] == 0).all(), res["status"][:8]
    nfr = int(res["n_frames"].sum())
    print(json.dumps({"files": N, "frames": nfr, "inflate_s": round(best, 4), "files_per_s": round(N / best, 1),
                      "frames_per_s": round(nfr / best, 1), "GBps_out": round(float(isize.sum()) / best / 1e9, 2),
                      "GBps_in": round(float(sizes.sum()) / best / 1e9, 2), "h2d_s": round(h2d, 4),
                      "h2d_GBps": round(float(sizes.sum()) / h2d / 1e9, 1)}), flush=True)
    del in_dev, out_dev, slots_dev, stage
eng.close()
