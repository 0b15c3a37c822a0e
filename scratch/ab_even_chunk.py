"""A/B of BatchPipeline.even_chunk (equal forwards) against the old chunking (full chunks + a remainder) on the default
bench step: python3 scratch/ab_even_chunk.py [old]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd"))
import bench
from cpx.pipeline import BatchPipeline
if len(sys.argv) > 1 and sys.argv[1] == "old":
    BatchPipeline.even_chunk = lambda self, n: max(1, min(self.cnn_chunk, n))
sys.argv = ["bench.py", "--cpu-clips", "0", "--no-extras", "--steps", "3"]
bench.main()
