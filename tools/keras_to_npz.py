#!/usr/bin/env python3
"""Keras model of the reference  ->  <out>.npz + <out>.json as the MI355X build loads them.

Runs wherever TensorFlow / Keras is installed; needs NOTHING from this repository (copy the file next to the model).

    python keras_to_npz.py <model.keras | saved_model_dir | model.h5> <out_base> [--sidecar <model>.json]
                           [--dump-io N]

The reference builds its classifier in ml_tools/kerasmodel.py:259-350: input -> the "WRResNet" sub-model
(ml_tools/resnet/wr_resnet.py:5-98) -> GlobalAveragePooling2D -> [Dense(size, relu) for size in dense_sizes] ->
Dropout -> Dense(n_labels, sigmoid | softmax, name="prediction").  Inside WRResNet every layer is named
(conv1_{s}, res{s}b{d}_branch2a|2b, bn{s}b{d}_branch2a|2b, final_bn) EXCEPT the three 1x1 projection shortcuts
(wr_resnet.py:88-93), which get Keras auto-names (conv2d, conv2d_1, ... -- the numbers depend on what else the
process created).  They are mapped by ORDER: the k-th 1x1 Conv2D is shortcut{k+2}.  Hidden Dense layers (auto-named
dense, dense_1, ...) become dense_0, dense_1, ... in order; the output layer keeps the name "prediction" and its
activation goes to 'prediction/activation'.

--dump-io N additionally writes <out>_io.npz with N seeded inputs in [0, 255], the model's outputs and its
pre-activation logits: the fixture tests/test_tf_parity_gpu.py needs to pin the HIP forward against TensorFlow
(CPX_TF_IO=<out>_io.npz CPX_TF_MODEL=<out> pytest tests/test_tf_parity_gpu.py -m gpu).
"""
import argparse
import json
import os
import shutil

import numpy as np


def convert_layers(layers):
    """layers: ordered list of dicts {"name", "class", "weights": {short name: array}, "config": {...}} (sub-models
    already flattened) -> dict of arrays in the cpx naming.  Pure NumPy: testable without TensorFlow."""
    out = {}
    n_short, n_hidden = 0, 0
    for lay in layers:
        cls, name, w, cfg = lay["class"], lay["name"], lay["weights"], lay.get("config", {})
        if cls == "Conv2D":
            k = np.asarray(w["kernel"], dtype=np.float32)
            if k.shape[0] == 1 and k.shape[1] == 1 and not name.startswith(("res", "conv1_")):
                name = "shortcut%d" % (n_short + 2)
                n_short += 1
            out[name + "/kernel"] = k
            out[name + "/bias"] = np.asarray(w.get("bias", np.zeros(k.shape[-1])), dtype=np.float32)
        elif cls == "BatchNormalization":
            c = len(np.asarray(w["moving_mean"]))
            eps = float(cfg.get("epsilon", 1e-3))
            if abs(eps - 1e-3) > 1e-12:  # the build folds BatchNorm with eps = 1e-3: carry another eps in the variance
                w = dict(w, moving_variance=np.asarray(w["moving_variance"], np.float64) + (eps - 1e-3))
            out[name + "/gamma"] = np.asarray(w.get("gamma", np.ones(c)), dtype=np.float32)
            out[name + "/beta"] = np.asarray(w.get("beta", np.zeros(c)), dtype=np.float32)
            out[name + "/moving_mean"] = np.asarray(w["moving_mean"], dtype=np.float32)
            out[name + "/moving_variance"] = np.asarray(w["moving_variance"], dtype=np.float32)
        elif cls == "Dense":
            act = cfg.get("activation", "linear")
            if name == "prediction" or lay is layers[-1] or act in ("sigmoid", "softmax"):
                if act not in ("sigmoid", "softmax"):
                    raise ValueError("output activation %r is not supported" % act)
                out["prediction/kernel"] = np.asarray(w["kernel"], dtype=np.float32)
                out["prediction/bias"] = np.asarray(w["bias"], dtype=np.float32)
                out["prediction/activation"] = np.array(act)
            else:
                if act != "relu":
                    raise ValueError("hidden dense activation %r is not supported" % act)
                out["dense_%d/kernel" % n_hidden] = np.asarray(w["kernel"], dtype=np.float32)
                out["dense_%d/bias" % n_hidden] = np.asarray(w["bias"], dtype=np.float32)
                n_hidden += 1
        elif w:
            raise ValueError("layer %s (%s) has weights this converter does not know" % (name, cls))
    if n_short != 3 or "prediction/kernel" not in out or "final_bn/gamma" not in out:
        raise ValueError("not a WR-ResNet classifier of the reference: %d projection shortcuts, prediction %s, final_bn %s"
                         % (n_short, "prediction/kernel" in out, "final_bn/gamma" in out))
    return out


def flatten_keras(model):
    """Ordered layer records of a tf.keras model, sub-models expanded in place."""
    recs = []
    for layer in model.layers:
        if hasattr(layer, "layers"):
            recs.extend(flatten_keras(layer))
            continue
        names = [v.name.split("/")[-1].split(":")[0] for v in layer.weights]
        vals = layer.get_weights()
        cfg = layer.get_config()
        recs.append({"name": layer.name, "class": layer.__class__.__name__, "weights": dict(zip(names, vals)),
                     "config": {k: cfg.get(k) for k in ("activation", "epsilon", "groups", "strides", "padding")}})
    return recs


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("model")
    ap.add_argument("out_base")
    ap.add_argument("--sidecar", help="the reference's <model>.json (labels, hyperparams, thresholds): copied to <out>.json")
    ap.add_argument("--dump-io", type=int, default=0)
    args = ap.parse_args()
    import tensorflow as tf

    model = tf.keras.models.load_model(args.model, compile=False)
    weights = convert_layers(flatten_keras(model))
    np.savez(args.out_base + ".npz", **weights)
    sidecar = args.sidecar or os.path.splitext(args.model)[0] + ".json"
    if os.path.exists(sidecar):
        shutil.copy(sidecar, args.out_base + ".json")
    else:
        print("no sidecar JSON found (%s): write %s.json with labels / hyperparams yourself" % (sidecar, args.out_base))
    print("wrote %s.npz: %d arrays, output %s, %d hidden dense layers" % (
        args.out_base, len(weights), weights["prediction/activation"], sum(k.startswith("dense_") for k in weights) // 2))
    if args.dump_io:
        rng = np.random.default_rng(0)
        shape = tuple(int(v) for v in model.inputs[0].shape[1:])
        x = rng.uniform(0, 255, size=(args.dump_io,) + shape).astype(np.float32)
        y = np.asarray(model(x, training=False), dtype=np.float32)
        pre = tf.keras.Model(model.inputs, model.layers[-1].input)
        feat = np.asarray(pre(x, training=False), dtype=np.float32)
        logits = feat @ weights["prediction/kernel"] + weights["prediction/bias"]
        np.savez(args.out_base + "_io.npz", x=x, probs=y, logits=logits.astype(np.float32))
        print("wrote %s_io.npz" % args.out_base)


if __name__ == "__main__":
    main()
