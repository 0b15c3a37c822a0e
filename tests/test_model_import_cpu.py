"""Model import (what makes the TF parity pinnable): tools/keras_to_npz.py maps a Keras
model of the reference -- WRResNet sub-model with AUTO-NAMED projection shortcuts (wr_resnet.py:88-93), hidden dense
layers, sigmoid or softmax output (kerasmodel.py:337-345) -- onto the build's weight names.  No TensorFlow here, so
the Keras side is a synthetic layer listing in Keras' order and naming; the converted archive must give the oracle
forward the very same logits as the weights it was made from."""
import importlib.util
import json
import os

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("keras_to_npz", os.path.join(REPO, "tools", "keras_to_npz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def keras_listing(w, first_auto=11):
    """The layer records tf.keras would list for the reference's model built from weights `w` (cpx names):
    shortcuts and hidden dense layers under Keras auto-names, parameter-free layers in between."""
    recs = [{"name": "input", "class": "InputLayer", "weights": {}, "config": {}}]
    auto = first_auto

    def conv(name, key):
        return {"name": name, "class": "Conv2D", "weights": {"kernel": w[key + "/kernel"], "bias": w[key + "/bias"]},
                "config": {"groups": 2}}

    def bn(name):
        return {"name": name, "class": "BatchNormalization", "config": {"epsilon": 1e-3},
                "weights": {k: w[name + "/" + k] for k in ("gamma", "beta", "moving_mean", "moving_variance")}}

    recs.append(conv("conv1_1", "conv1_1"))
    for stage in (2, 3, 4):
        for d in range(3):
            b = "%db%d" % (stage, d)
            recs += [bn("bn%s_branch2a" % b), {"name": "activation_%d" % auto, "class": "Activation", "weights": {}},
                     conv("res%s_branch2a" % b, "res%s_branch2a" % b),
                     {"name": "dropout_%d" % auto, "class": "Dropout", "weights": {}}, bn("bn%s_branch2b" % b),
                     conv("res%s_branch2b" % b, "res%s_branch2b" % b)]
            if d == 0:
                recs.append(conv("conv2d_%d" % auto, "shortcut%d" % stage))   # Keras auto-name
                auto += 1
            else:
                recs.append({"name": "identity_%d" % d, "class": "Identity", "weights": {}})
            recs.append({"name": "add_%d%d" % (stage, d), "class": "Add", "weights": {}})
    recs.append(bn("final_bn"))
    recs.append({"name": "global_average_pooling2d", "class": "GlobalAveragePooling2D", "weights": {}})
    k = 0
    while "dense_%d/kernel" % k in w:
        recs.append({"name": "dense_%d" % (k + 5), "class": "Dense", "config": {"activation": "relu"},
                     "weights": {"kernel": w["dense_%d/kernel" % k], "bias": w["dense_%d/bias" % k]}})
        k += 1
    recs.append({"name": "dropout", "class": "Dropout", "weights": {}})
    recs.append({"name": "prediction", "class": "Dense",
                 "config": {"activation": str(w.get("prediction/activation", "sigmoid"))},
                 "weights": {"kernel": w["prediction/kernel"], "bias": w["prediction/bias"]}})
    return recs


@pytest.mark.parametrize("dense_sizes,activation", [(None, "sigmoid"), ((48, 24), "softmax"), ((32,), "sigmoid")])
def test_converted_archive_gives_identical_oracle_logits(tmp_path, dense_sizes, activation):
    import cnn_oracle as co
    from cpx.ml_tools import wrresnet as wr

    tool = _tool()
    w = wr.random_weights(17, seed=4, dense_sizes=dense_sizes, activation=activation)
    conv = tool.convert_layers(keras_listing(w))
    np.savez(tmp_path / "m.npz", **conv)
    back = wr.load_weights(tmp_path / "m.npz")
    assert set(back) == set(w) | {"prediction/activation"}
    for k, v in w.items():
        if k != "prediction/activation":
            assert np.array_equal(back[k], v), k
    hidden, act = wr.head_of(back)
    assert len(hidden) == len(dense_sizes or ()) and act == activation
    x = np.random.default_rng(1).uniform(0, 255, size=(1, 32, 32, 2)).astype(np.float32)
    a, pa = co.forward(w, x)
    b, pb = co.forward(back, x)
    assert np.array_equal(a, b) and np.array_equal(pa, pb)
    if activation == "softmax":
        assert abs(float(pa.sum()) - 1.0) < 1e-5


def test_converter_rejects_other_architectures():
    from cpx.ml_tools import wrresnet as wr

    tool = _tool()
    w = wr.random_weights(17, seed=4)
    recs = [r for r in keras_listing(w) if not r["name"].startswith("conv2d_")]   # a model without projections
    with pytest.raises(ValueError):
        tool.convert_layers(recs)
    recs = keras_listing(w)
    recs[-1]["config"]["activation"] = "tanh"
    with pytest.raises(ValueError):
        tool.convert_layers(recs)


def test_interpreter_refuses_heads_it_does_not_run(tmp_path):
    """a sidecar the build cannot honour must fail at load, not classify with the wrong head."""
    from cpx.ml_tools import wrresnet as wr
    from cpx.ml_tools.interpreter import WRResNetInterpreter

    labels = ["l%d" % i for i in range(17)]

    def model(hp, explicit=None, **kw):
        base = tmp_path / ("m%d" % len(list(tmp_path.iterdir())))
        w = wr.random_weights(17, seed=1, **kw)
        if explicit:
            w["prediction/activation"] = explicit     # as tools/keras_to_npz.py always records it
        wr.save_model(base, w, labels, hyperparams=hp)
        return base.with_suffix(".npz")

    WRResNetInterpreter(model({}))                                             # the default head loads
    WRResNetInterpreter(model({"dense_sizes": [48, 24]}, dense_sizes=(48, 24)))
    WRResNetInterpreter(model({"multi_label": False}, activation="softmax"))
    for hp in ({"model_name": "efficientnetv2b3"}, {"mvm": True}, {"model_name": "inceptionv3"}):
        with pytest.raises(NotImplementedError):
            WRResNetInterpreter(model(hp))
    with pytest.raises(ValueError):
        WRResNetInterpreter(model({"dense_sizes": [64]}))                      # sidecar says hidden layer, weights have none
    with pytest.raises(ValueError):
        WRResNetInterpreter(model({"multi_label": False}, explicit="sigmoid"))
