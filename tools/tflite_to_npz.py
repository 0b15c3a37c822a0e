#!/usr/bin/env python3
"""A released TFLite WR-ResNet model (.tflite) -> the <out_base>.npz the cpx kernels read -- in pure Python: the
flatbuffer is parsed here, TensorFlow is not needed (only NumPy).

    python tools/tflite_to_npz.py model.tflite /tmp/wr [--sidecar model.json]

The reference loads this artefact with LiteInterpreter (/root/reference/src/ml_tools/interpreter.py:520-560); its CI
downloads it (.github/workflows/release.yml:46) and tests/clips/possum.txt's prediction came from one.  Anyone holding
the released file can therefore pin row a20 (CNN logits) of SURVEY section 8 without a TensorFlow installation:
convert, then run tests/test_tf_parity_gpu.py against outputs recorded with the TFLite runtime.

What is read: the float32 graph of WR-ResNet-22-4 (src/ml_tools/resnet/wr_resnet.py:5-98) as the TFLite converter
writes it -- CONV_2D (filter OHWI, bias, fused ReLU: a convolution with the BatchNorm that follows it folded in), MUL +
ADD by per-channel constants (a BatchNorm that follows a residual ADD cannot be folded: scale and shift), RELU, ADD of
two activations (the residual), MEAN (global average pooling), FULLY_CONNECTED, LOGISTIC / SOFTMAX.  The walk follows
the operators in order and fills the Keras-layout names of cpx/ml_tools/wrresnet.py; a folded or affine-only BatchNorm
becomes gamma = scale, beta = shift, moving_mean = 0, moving_variance = 1 - eps (so that the loader's
gamma / sqrt(var + eps) gives the scale back exactly).  Anything else (quantised tensors, another topology) is refused
with the operator that stopped the walk."""
import argparse
import os
import shutil
import sys

import numpy as np

# the reader lives in the package (cpx.ml_tools.tflite_reader: get_interpreter converts a .tflite on load with it)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "classifier-pipeline_amd"))
from cpx.ml_tools.tflite_reader import (BN_EPS, OPS, Graph, Table, bn_params, conv_kernel, convert,  # noqa: E402,F401
                                        identity_variance)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("model", help="the .tflite file")
    ap.add_argument("out_base", help="writes <out_base>.npz (and copies the sidecar to <out_base>.json)")
    ap.add_argument("--sidecar", default=None, help="the model's JSON sidecar (default: <model>.json next to it)")
    args = ap.parse_args()
    with open(args.model, "rb") as fh:
        g = Graph(fh.read())
    weights = convert(g)
    np.savez(args.out_base + ".npz", **weights)
    sidecar = args.sidecar or os.path.splitext(args.model)[0] + ".json"
    if os.path.exists(sidecar):
        shutil.copy(sidecar, args.out_base + ".json")
    else:
        print("no sidecar JSON found (%s): write %s.json with labels / hyperparams yourself" % (sidecar, args.out_base))
    print("wrote %s.npz: %d arrays, output %s" % (args.out_base, len(weights), weights["prediction/activation"]))


if __name__ == "__main__":
    main()
