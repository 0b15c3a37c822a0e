"""What would a TWO-plane bf16 split (3 products: w0x0 + w0x1 + w1x0, or 4 with w1x1) of the stage-2 / stage-3 3x3
convolutions do to the logits?  CPU emulation on the network test's inputs (oracle/cnn_oracle.py forward with the
stride-1 3x3 convolutions of 32 / 64 channels per group replaced), against the float32 forward.
python3 scratch/bf16x2_error_probe.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "classifier-pipeline_amd")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np, torch
import torch.nn.functional as F
import cnn_oracle as co
from cpx.ml_tools import wrresnet as wr

def planes(t, n, rne):
    out, r = [], t.clone()
    for _ in range(n):
        if rne:
            p = r.to(torch.bfloat16).to(torch.float32)
        else:
            p = (r.view(torch.int32) & -65536).view(torch.float32)
        out.append(p); r = r - p
    return out

MODE = {}
orig_conv = co._conv
def conv_emul(x, w, name, stride, same):
    k = w[name + "/kernel"]
    cin_g, cout_g = k.shape[2], k.shape[3] // 2
    if not (MODE.get("on") and k.shape[0] == 3 and stride == 1 and cin_g in (32, 64) and cout_g in (32, 64)):
        return orig_conv(x, w, name, stride, same)
    kt = torch.from_numpy(k); b = torch.from_numpy(w[name + "/bias"])
    wt = kt.permute(3, 2, 0, 1).contiguous()
    H, W = x.shape[2], x.shape[3]
    xpad = F.pad(x, (1, 1, 1, 1))
    nx, nw, prods, rne = MODE["nx"], MODE["nw"], MODE["prods"], MODE["rne"]
    xs, ws = planes(xpad, nx, rne), planes(wt, nw, rne)
    acc = torch.zeros((x.shape[0], wt.shape[0], H, W), dtype=torch.float64)
    for (i, j) in prods:
        acc += F.conv2d(xs[j].double(), ws[i].double(), None, groups=2)
    return (acc + b.double()[None, :, None, None]).float()
co._conv = conv_emul

for fs, n in ((32, 3), (64, 1)):
    rng = np.random.default_rng(5 + fs)
    side = 5 * fs
    x = rng.uniform(0, 255, size=(n, side, side, 2)).astype(np.float32)
    x[:, ::7, :, 1] = 0.0
    w = co.calibrate_bn(wr.random_weights(17, seed=3), x)
    MODE["on"] = False
    want, _ = co.forward(w, x)
    print("fs", fs, "max |logit|", float(np.abs(want).max()))
    for label, nx, nw, prods, rne in (
        ("3 planes, 6 products (shipped)", 3, 3, [(1,1),(2,0),(0,2),(1,0),(0,1),(0,0)], False),
        ("2 planes x 2 planes, 3 products, truncating", 2, 2, [(1,0),(0,1),(0,0)], False),
        ("2 planes x 2 planes, 3 products, round-to-nearest", 2, 2, [(1,0),(0,1),(0,0)], True),
        ("2 x 2, 4 products, round-to-nearest", 2, 2, [(1,1),(1,0),(0,1),(0,0)], True),
        ("w 3 planes, x 2 planes, 5 products, rne", 2, 3, [(1,1),(2,0),(1,0),(0,1),(0,0)], True),
        ("1 plane (plain bf16), rne", 1, 1, [(0,0)], True)):
        MODE.update(on=True, nx=nx, nw=nw, prods=prods, rne=rne)
        got, _ = co.forward(w, x)
        print("  %-52s max |dlogit| %.3e" % (label, float(np.abs(got - want).max())))
