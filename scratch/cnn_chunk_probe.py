"""Samples/s of the WR-ResNet forward as a function of the samples per call: small calls keep the activations in
the 256 MB memory-side cache (4 buffers x N x 6.5 MB live at stage 2)."""
import sys, time
sys.path.insert(0,'classifier-pipeline_amd'); sys.path.insert(0,'oracle')
import numpy as np, torch
from cpx.engine import TrackEngine
from cpx.ml_tools import wrresnet as wr
eng = TrackEngine()
if len(sys.argv) > 1: eng.set_cnn_math(sys.argv[1])
w = wr.random_weights(17, seed=3)
total = 2048
xall = torch.rand((total,160,160,2), device=eng.device)*255
w = wr.calibrate_bn_device(eng, w, xall[:32].contiguous())
net = wr.WRResNetDevice(eng, w, 17)
for N in (8, 16, 24, 32, 64, 128, 512, 2048):
    logits = torch.empty((total, 17), device=eng.device)
    torch.cuda.synchronize()
    net.forward_async(xall[:N], logits[:N])
    eng.synchronize(); torch.cuda.synchronize(); t=time.time()
    for rep in range(2):
        for i in range(0, total, N):
            net.forward_async(xall[i:i+N], logits[i:i+N])
    eng.synchronize(); torch.cuda.synchronize(); dt=(time.time()-t)/2
    print(eng.get_cnn_math(), 'overflow', eng.cnn_last_overflow(), 'N', N, 'samples/s', round(total/dt,1), 'TFLOP/s', round(total*12.62e9/dt/1e12,2), flush=True)
