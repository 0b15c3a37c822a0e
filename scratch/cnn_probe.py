import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'classifier-pipeline_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
from cpx.engine import TrackEngine
from cpx.ml_tools import wrresnet as wr
eng = TrackEngine()
w = wr.random_weights(17, seed=3)
if os.environ.get('PROBE_ZERO'):
    w = {k: (np.zeros_like(v) if hasattr(v, 'dtype') and v.dtype == np.float32 else v) for k, v in w.items()}
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.rand((N,160,160,2), device=eng.device)*255
if not os.environ.get('PROBE_ZERO') and not os.environ.get('PROBE_UNCALIBRATED'):
    w = wr.calibrate_bn_device(eng, w, x[:32].contiguous())  # BatchNorm statistics that fit the data, as a trained network's do
net = wr.WRResNetDevice(eng, w, 17)
if os.environ.get('PROBE_ZERO'):
    x.zero_()
net.forward(x)
eng.conv_timing(True)
torch.cuda.synchronize(); t=time.time()
for _ in range(3): net.forward(x)
torch.cuda.synchronize(); dt=(time.time()-t)/3
rep = eng.conv_timing()
print(N, 'samples/s', round(N/dt,1), 'TFLOP/s', round(N*12.62e9/dt/1e12,2), 'math', eng.get_cnn_math(), 'overflow rerun', eng.cnn_last_overflow())
tot = sum(v[1] for v in rep.values())
for k,(n,ms,fl) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print('cin_g %3d cout_g %3d stride %d%s: launches %2d  %7.2f ms (%4.1f%%)  %6.2f TFLOP/s' % (k//10000, (k%10000)//10, (k%10)%5, ' 1x1' if (k%10)>=5 else '    ', n, ms, 100*ms/tot, fl/(ms/1e3)/1e12))
