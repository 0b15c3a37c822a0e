"""Where does a wave's time go?  64 copies of the possum recording's content recompressed as (a) the original stream,
(b) Huffman only (every symbol a literal), (c) stored blocks (pure copy), (d) level 9, (e) RLE -- one wavefront each."""
import json, os, sys, time, zlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
import numpy as np, torch
from cpx.engine import TrackEngine
from cpx.track.bulk import stage_blobs, decode_staged
eng = TrackEngine(model="lepton3")
raw = open(os.path.join(REPO, "tests", "golden", "possum.cptv"), "rb").read()
data = zlib.decompress(raw, 47)
def gz(level, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    return c.compress(data) + c.flush()
variants = {"original": raw, "huffman_only": gz(6, zlib.Z_HUFFMAN_ONLY), "stored": gz(0), "level9": gz(9), "level1": gz(1), "rle": gz(6, zlib.Z_RLE)}
for name, blob in variants.items():
    staged = stage_blobs(torch, [blob] * 64)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); d = decode_staged(eng, staged); best = min(best, time.perf_counter() - t0)
    assert not d.errors, d.errors
    print(json.dumps({"variant": name, "compressed_bytes": len(blob), "inflated_bytes": len(data), "decode_s_64_files": round(best, 4),
                      "ns_per_output_byte": round(best / len(data) * 1e9, 1)}), flush=True)
