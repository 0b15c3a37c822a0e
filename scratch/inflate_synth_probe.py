"""The inflate kernel on the bench's synthetic recordings (sensor noise: near-incompressible bit-packed deltas):
sizes, and decode time for 64 / 2048 / 4096 copies; the same content as Huffman-only and level 1."""
import json, os, sys, time, zlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
import bench
from cpx import synth
from cpx.cptv import encode_cptv
from cpx.engine import TrackEngine
from cpx.track.bulk import stage_blobs, decode_staged
eng = TrackEngine(model="lepton3")
T = 270
host = bench.synth_on_device(torch, torch.device("cuda", 0), 2, T, seed=4321).cpu().numpy().view(np.uint16).reshape(2, T, 120, 160)
t_on, ffc = synth.frame_times(T)
raw = encode_cptv(host[0], t_on, ffc, level=6)
data = zlib.decompress(raw, 47)
def gz(level, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    return c.compress(data) + c.flush()
variants = {"level6": raw, "huffman_only": gz(6, zlib.Z_HUFFMAN_ONLY), "level1": gz(1), "stored": gz(0)}
for name, blob in variants.items():
    for n in (64, 2048, 4096):
        if n > 64 and name not in ("level6", "huffman_only"):
            continue
        staged = stage_blobs(torch, [blob] * n)
        best = 1e9
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter(); d = decode_staged(eng, staged); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        assert not d.errors, d.errors
        del d
        print(json.dumps({"variant": name, "files": n, "compressed_bytes": len(blob), "inflated_bytes": len(data), "decode_s": round(best, 4),
                          "ns_per_output_byte_per_wave": round(best / len(data) * 1e9, 1)}), flush=True)
