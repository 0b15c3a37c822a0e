"""Bitwise repeatability of the network forward (every math mode): N forwards of the same batch, all equal to the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'classifier-pipeline_amd'))
import numpy as np, torch
from cpx.engine import TrackEngine
from cpx.ml_tools import wrresnet as wr
eng = TrackEngine()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
x = torch.rand((96, 160, 160, 2), device=eng.device) * 255
# (BatchNorm statistics fitted to the data: the fp16x2 forward then stays on its fp16 kernels, fused blocks included)
net = wr.WRResNetDevice(eng, wr.calibrate_bn_device(eng, wr.random_weights(17, seed=3), x[:32].contiguous()), 17)
for mode in ("fp16x2", "bf16x3", "bf16x2", "f32"):
    eng.set_cnn_math(mode)
    ref, _ = net.forward(x)
    ref = ref.clone()
    bad = 0
    for i in range(N):
        l, _ = net.forward(x)
        bad += 0 if torch.equal(l, ref) else 1
    print(mode, "forwards", N, "different from the first:", bad, "| fp16 overflow rerun:", eng.cnn_last_overflow() if mode == "fp16x2" else "-")
