import sys, time
sys.path.insert(0,'classifier-pipeline_amd'); sys.path.insert(0,'oracle')
import numpy as np, torch
from cpx.engine import TrackEngine
from cpx.ml_tools import wrresnet as wr
eng = TrackEngine()
w = wr.random_weights(17, seed=3)
net = wr.WRResNetDevice(eng, w, 17)
for N in (64, 256):
    x = torch.rand((N,160,160,2), device=eng.device)*255
    net.forward(x)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(3): net.forward(x)
    torch.cuda.synchronize(); dt=(time.time()-t)/3
    print(N, 'samples/s', N/dt, 'TFLOP/s', N*12.62e9/dt/1e12)
