"""tools/tflite_to_npz.py: a released .tflite WR-ResNet read in pure Python (flatbuffer parser + operator walk) and
mapped onto the build's weight names.  No TensorFlow here, so the .tflite side is a synthetic flatbuffer written by a
small builder below, laid out as the TFLite converter writes this network: CONV_2D with the following BatchNorm folded
in (ReLU fused), a BatchNorm after a residual ADD as MUL + ADD by constants, MEAN, FULLY_CONNECTED, LOGISTIC.  The
converted archive must give the oracle forward the logits of the weights the flatbuffer was made from."""
import importlib.util
import os
import struct

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("tflite_to_npz", os.path.join(REPO, "tools", "tflite_to_npz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ---- a minimal flatbuffer writer: parents first, children after them (unsigned forward offsets) -------------------
class FB:
    def __init__(self):
        self.b = bytearray(b"\0\0\0\0TFL3")

    def align(self, n=4):
        while len(self.b) % n:
            self.b.append(0)


def build_table(fb, fields):
    """fields: {id: ("I" | "i" | "b" | "B", value) scalar, ("ref", writer) a child written after the table}."""
    fb.align()
    n = (max(fields) + 1) if fields else 0
    body = bytearray()
    offs = [0] * n
    patches = []
    cur = 4
    for fid in sorted(fields):
        kind, val = fields[fid]
        size = {"I": 4, "i": 4, "b": 1, "B": 1, "ref": 4}[kind]
        while cur % size:
            body.append(0)
            cur += 1
        offs[fid] = cur
        if kind == "ref":
            patches.append((cur, val))
            body += b"\0\0\0\0"
        else:
            body += struct.pack("<" + kind, val)
        cur += size
    vt = struct.pack("<HH", 4 + 2 * n, 4 + len(body)) + b"".join(struct.pack("<H", o) for o in offs)
    vt_pos = len(fb.b)
    fb.b += vt
    fb.align()
    pos = len(fb.b)
    fb.b += struct.pack("<i", pos - vt_pos) + body
    for rel, writer in patches:
        fb.align()
        target = writer()
        struct.pack_into("<I", fb.b, pos + rel, target - (pos + rel))
    return pos


def vec_scalars(fb, fmt, values):
    def w():
        fb.align()
        pos = len(fb.b)
        fb.b += struct.pack("<I", len(values)) + struct.pack("<%d%s" % (len(values), fmt), *values)
        return pos
    return w


def vec_bytes(fb, data):
    def w():
        fb.align()
        pos = len(fb.b)
        fb.b += struct.pack("<I", len(data)) + bytes(data)
        return pos
    return w


def string(fb, s):
    def w():
        fb.align()
        pos = len(fb.b)
        raw = s.encode()
        fb.b += struct.pack("<I", len(raw)) + raw + b"\0"
        return pos
    return w


def vec_tables(fb, writers):
    def w():
        fb.align()
        pos = len(fb.b)
        fb.b += struct.pack("<I", len(writers)) + b"\0\0\0\0" * len(writers)
        for i, tw in enumerate(writers):
            fb.align()
            t = tw()
            loc = pos + 4 + 4 * i
            struct.pack_into("<I", fb.b, loc, t - loc)
        return pos
    return w


def tflite_of(w, hidden=()):
    """The flatbuffer the TFLite converter would write for the WR-ResNet with cpx weights `w` (folded BatchNorms)."""
    from cpx.ml_tools.wrresnet import bn_affine

    fb = FB()
    tensors, buffers, ops = [], [b""], []
    codes = {"ADD": 0, "CONV_2D": 3, "FULLY_CONNECTED": 9, "LOGISTIC": 14, "MUL": 18, "MEAN": 40, "SOFTMAX": 25}
    code_list = sorted(codes.values())

    def tensor(shape, data=None, name="t"):
        if data is not None:
            buffers.append(np.ascontiguousarray(data, "<f4" if data.dtype.kind == "f" else "<i4").tobytes())
            bi = len(buffers) - 1
        else:
            bi = 0
        tensors.append((list(shape), 0 if data is None or data.dtype.kind == "f" else 2, bi, name))
        return len(tensors) - 1

    def op(name, inputs, outputs, opt_type=0, opts=None):
        ops.append((code_list.index(codes[name]), inputs, outputs, opt_type, opts))

    def conv(x, kernel_hwio, bias, stride, same, relu):
        k = np.transpose(kernel_hwio, (3, 0, 1, 2))   # OHWI
        y = tensor([1, 0, 0, k.shape[0]])
        op("CONV_2D", [x, tensor(k.shape, k), tensor(bias.shape, bias)], [y], 1,
           {0: ("b", 0 if same else 1), 1: ("i", stride), 2: ("i", stride), 3: ("b", 1 if relu else 0)})
        return y

    def affine(x, scale, shift):
        m = tensor([1, 0, 0, len(scale)])
        op("MUL", [x, tensor(scale.shape, scale)], [m], 21, {0: ("b", 0)})
        a = tensor([1, 0, 0, len(scale)])
        op("ADD", [m, tensor(shift.shape, shift)], [a], 11, {0: ("b", 1)})
        return a

    x = tensor([1, 160, 160, 2], name="input")
    cur = conv(x, w["conv1_1/kernel"], w["conv1_1/bias"], 1, True, False)
    for stage in (2, 3, 4):
        for d in range(3):
            b = "%db%d" % (stage, d)
            s = (stage - 1) if d == 0 else 1
            sc, sh = bn_affine(w, "bn%s_branch2a" % b)
            act = affine(cur, sc, sh)
            # conv a with bn ..._branch2b folded in, ReLU fused
            sc2, sh2 = bn_affine(w, "bn%s_branch2b" % b)
            ka = w["res%s_branch2a/kernel" % b] * sc2[None, None, None, :]
            ba = w["res%s_branch2a/bias" % b] * sc2 + sh2
            short = cur
            if d == 0:
                short = conv(act, w["shortcut%d/kernel" % stage], w["shortcut%d/bias" % stage], s, False, False)
            mid = conv(act, ka.astype(np.float32), ba.astype(np.float32), s, True, True)
            out = conv(mid, w["res%s_branch2b/kernel" % b], w["res%s_branch2b/bias" % b], 1, True, False)
            y = tensor([1, 0, 0, out and 0])
            op("ADD", [out, short], [y], 11, {0: ("b", 0)})
            cur = y
    sc, sh = bn_affine(w, "final_bn")
    cur = affine(cur, sc, sh)
    gap = tensor([1, 256])
    op("MEAN", [cur, tensor([2], np.array([1, 2], np.int32))], [gap], 27, {0: ("b", 0)})
    cur = gap
    for i in range(len(hidden)):
        y = tensor([1, hidden[i]])
        op("FULLY_CONNECTED", [cur, tensor(w["dense_%d/kernel" % i].T.shape, np.ascontiguousarray(w["dense_%d/kernel" % i].T)),
                               tensor(w["dense_%d/bias" % i].shape, w["dense_%d/bias" % i])], [y], 8, {0: ("b", 1)})
        cur = y
    y = tensor([1, w["prediction/bias"].shape[0]])
    op("FULLY_CONNECTED", [cur, tensor(w["prediction/kernel"].T.shape, np.ascontiguousarray(w["prediction/kernel"].T)),
                           tensor(w["prediction/bias"].shape, w["prediction/bias"])], [y], 8, {0: ("b", 0)})
    z = tensor(tensors[y][0])
    op("SOFTMAX" if w.get("prediction/activation") == "softmax" else "LOGISTIC", [y], [z])

    def t_writer(t):
        shape, ttype, bi, name = t
        return lambda: build_table(fb, {0: ("ref", vec_scalars(fb, "i", shape)), 1: ("b", ttype), 2: ("I", bi),
                                        3: ("ref", string(fb, name))})

    def op_writer(o):
        ci, ins, outs, ot, opts = o
        f = {0: ("I", ci), 1: ("ref", vec_scalars(fb, "i", ins)), 2: ("ref", vec_scalars(fb, "i", outs))}
        if opts is not None:
            f[3] = ("B", ot)
            f[4] = ("ref", lambda: build_table(fb, opts))
        return lambda: build_table(fb, f)

    def sub_writer():
        return build_table(fb, {0: ("ref", vec_tables(fb, [t_writer(t) for t in tensors])),
                                1: ("ref", vec_scalars(fb, "i", [0])), 2: ("ref", vec_scalars(fb, "i", [len(tensors) - 1])),
                                3: ("ref", vec_tables(fb, [op_writer(o) for o in ops]))})

    def code_writer(c):
        return lambda: build_table(fb, {0: ("b", c if c < 127 else 127), 3: ("i", c)})

    def buf_writer(data):
        return lambda: build_table(fb, {0: ("ref", vec_bytes(fb, data))} if data else {})

    root = build_table(fb, {0: ("I", 3), 1: ("ref", vec_tables(fb, [code_writer(c) for c in code_list])),
                            2: ("ref", vec_tables(fb, [sub_writer])),
                            4: ("ref", vec_tables(fb, [buf_writer(d) for d in buffers]))})
    struct.pack_into("<I", fb.b, 0, root)
    return bytes(fb.b)


@pytest.mark.parametrize("hidden,activation", [((), "sigmoid"), ((48,), "softmax")])
def test_tflite_round_trip_gives_the_same_logits(tmp_path, hidden, activation):
    import cnn_oracle as co
    from cpx.ml_tools import wrresnet as wr

    tool = _tool()
    w = wr.random_weights(17, seed=5, dense_sizes=list(hidden) or None, activation=activation)
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 255, size=(2, 160, 160, 2)).astype(np.float32)
    w = co.calibrate_bn(w, x)
    blob = tflite_of(w, hidden)
    p = tmp_path / "m.tflite"
    p.write_bytes(blob)
    g = tool.Graph(blob)
    assert [o["name"] for o in g.ops].count("CONV_2D") == 22 and g.ops[-1]["name"] in ("LOGISTIC", "SOFTMAX")
    got = tool.convert(g)
    assert set(got) >= set(k for k in w if not k.endswith("/activation"))
    assert got["prediction/activation"] == activation
    want, want_p = co.forward(w, x)
    have, have_p = co.forward(got, x)
    assert float(np.abs(want - have).max()) <= 2e-4 and float(np.abs(want_p - have_p).max()) <= 1e-4
    # the command-line form writes the archive the interpreter loads
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "tflite_to_npz.py"), str(p), str(tmp_path / "wr")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    z = wr.load_weights(tmp_path / "wr.npz")
    assert np.array_equal(z["conv1_1/kernel"], got["conv1_1/kernel"]) and str(z["prediction/activation"]) == activation


def test_not_a_tflite_file_is_refused():
    tool = _tool()
    with pytest.raises(ValueError):
        tool.Graph(b"\0" * 64)


def test_get_interpreter_reads_a_tflite_model_on_load(tmp_path):
    """get_interpreter on a `.tflite` path (what the reference's LiteInterpreter takes, interpreter.py:520-560,597-628):
    the flatbuffer is converted on load by the package's own reader; Keras / RandomForest files name their converter."""
    import json

    import cnn_oracle as co
    from cpx.config.config import ModelConfig
    from cpx.ml_tools import wrresnet as wr
    from cpx.ml_tools.interpreter import get_interpreter

    w = wr.random_weights(17, seed=6)
    x = np.random.default_rng(4).uniform(0, 255, size=(2, 160, 160, 2)).astype(np.float32)
    w = co.calibrate_bn(w, x)
    (tmp_path / "model.tflite").write_bytes(tflite_of(w, ()))
    labels = ["l%d" % i for i in range(17)]
    with open(tmp_path / "model.json", "w") as fh:
        json.dump({"labels": labels, "hyperparams": {"frame_size": 32}, "type": "thermal"}, fh)
    interp = get_interpreter(ModelConfig.load({"id": 1, "name": "lite", "model_file": str(tmp_path / "model.tflite")}))
    assert interp.labels == labels and interp._weights is not None
    want, _ = co.forward(w, x)
    have, _ = co.forward(interp._weights, x)
    assert float(np.abs(want - have).max()) <= 2e-4
    for name in ("model.keras", "model.h5", "forest.sav"):
        with pytest.raises(NotImplementedError, match="keras_to_npz"):
            get_interpreter(ModelConfig.load({"id": 2, "name": "k", "model_file": str(tmp_path / name)}))
