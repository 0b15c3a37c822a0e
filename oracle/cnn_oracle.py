"""TEST INFRASTRUCTURE ONLY -- plain PyTorch (CPU, float32) restatement of the reference's
WR-ResNet forward, the checker for the MFMA kernels.

PARITY UNPINNED against TensorFlow: neither TensorFlow nor the released model weights exist in
the build container, so what is pinned is the architecture as written in
src/ml_tools/resnet/wr_resnet.py:5-98 and src/ml_tools/kerasmodel.py:308-350: grouped (groups=2)
3x3 convolutions with bias and TensorFlow "SAME" padding (surplus at the bottom / right), strides
1 / 2 / 3, pre-activation basic blocks, 1x1 "valid" strided projection shortcuts, Keras
BatchNormalization (eps 1e-3, inference statistics), global average pooling, dense + sigmoid.
Weights use the Keras layouts of cpx.ml_tools.wrresnet (HWIO kernels).
"""

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
GROUPS = 2


def _conv(x, w, name, stride, same):
    k = torch.from_numpy(w[name + "/kernel"])  # [kh, kw, Cin/g, Cout]
    b = torch.from_numpy(w[name + "/bias"])
    wt = k.permute(3, 2, 0, 1).contiguous()    # [Cout, Cin/g, kh, kw]
    kh = k.shape[0]
    if same:
        H, W = x.shape[2], x.shape[3]
        Ho, Wo = -(-H // stride), -(-W // stride)
        ph = max((Ho - 1) * stride + kh - H, 0)
        pw = max((Wo - 1) * stride + kh - W, 0)
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    return F.conv2d(x, wt, b, stride=stride, padding=0, groups=GROUPS)


def _bn(x, w, name):
    g, b = torch.from_numpy(w[name + "/gamma"]), torch.from_numpy(w[name + "/beta"])
    m, v = torch.from_numpy(w[name + "/moving_mean"]), torch.from_numpy(w[name + "/moving_variance"])
    return (x - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + BN_EPS) * g[None, :, None, None] + \
        b[None, :, None, None]


def forward(w, x_nhwc, return_features=False):
    """x_nhwc: float32 [N,S,S,2] -> (logits, probs) numpy [N, n_labels]."""
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(x_nhwc, dtype=np.float32)).permute(0, 3, 1, 2)
        x = _conv(x, w, "conv1_1", 1, True)
        for stage in (2, 3, 4):
            for d in range(3):
                b = "%db%d" % (stage, d)
                s = (stage - 1) if d == 0 else 1
                shortcut = x
                y = F.relu(_bn(x, w, "bn%s_branch2a" % b))
                y = _conv(y, w, "res%s_branch2a" % b, s, True)
                y = F.relu(_bn(y, w, "bn%s_branch2b" % b))
                y = _conv(y, w, "res%s_branch2b" % b, 1, True)
                if d == 0:
                    shortcut = _conv(shortcut, w, "shortcut%d" % stage, s, False)
                x = F.relu(y + shortcut)
        x = F.relu(_bn(x, w, "final_bn"))
        feat = x.mean(dim=(2, 3))
        # head variants (kerasmodel.py:337-345): Dense(relu) layers of dense_sizes, then sigmoid or softmax
        h, k = feat, 0
        while "dense_%d/kernel" % k in w:
            h = F.relu(h @ torch.from_numpy(w["dense_%d/kernel" % k]) + torch.from_numpy(w["dense_%d/bias" % k]))
            k += 1
        logits = h @ torch.from_numpy(w["prediction/kernel"]) + torch.from_numpy(w["prediction/bias"])
        probs = torch.softmax(logits, dim=1) if str(w.get("prediction/activation", "sigmoid")) == "softmax" \
            else torch.sigmoid(logits)
        if return_features:
            return logits.numpy(), probs.numpy(), feat.numpy()
        return logits.numpy(), probs.numpy()


def calibrate_bn(w, x_nhwc):
    """Set every BatchNorm's moving statistics to the batch statistics of a calibration batch (what
    training would have produced), so that seeded random weights give O(1) activations / logits."""
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(x_nhwc, dtype=np.float32)).permute(0, 3, 1, 2)

        def fit(t, name):
            w[name + "/moving_mean"] = t.mean(dim=(0, 2, 3)).numpy().astype(np.float32)
            w[name + "/moving_variance"] = t.var(dim=(0, 2, 3), unbiased=False).numpy().astype(np.float32) + 1e-2

        x = _conv(x, w, "conv1_1", 1, True)
        for stage in (2, 3, 4):
            for d in range(3):
                b = "%db%d" % (stage, d)
                s = (stage - 1) if d == 0 else 1
                shortcut = x
                fit(x, "bn%s_branch2a" % b)
                y = F.relu(_bn(x, w, "bn%s_branch2a" % b))
                y = _conv(y, w, "res%s_branch2a" % b, s, True)
                fit(y, "bn%s_branch2b" % b)
                y = F.relu(_bn(y, w, "bn%s_branch2b" % b))
                y = _conv(y, w, "res%s_branch2b" % b, 1, True)
                if d == 0:
                    shortcut = _conv(shortcut, w, "shortcut%d" % stage, s, False)
                x = F.relu(y + shortcut)
        fit(x, "final_bn")
    return w
