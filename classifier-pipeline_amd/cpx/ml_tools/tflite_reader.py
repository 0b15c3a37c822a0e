"""Pure-Python reader of ONE kind of TFLite model: a float32 WR-ResNet graph (.tflite).  The flatbuffer is parsed here --
TensorFlow is not needed, only NumPy -- and the graph is walked into the Keras-layout weights of cpx/ml_tools/wrresnet.py.

The reference loads any `.tflite` with LiteInterpreter (src/ml_tools/interpreter.py:520-560), picked by the file's suffix
(interpreter.py:597-628).  This reader does NOT cover that: the artefact the reference's own CI classifies with
(.github/workflows/release.yml:46 downloads `inc3-tflite-15122023.tar`) is an Inception-v3, another topology, and is
REFUSED here with the operator that stopped the walk; get_interpreter then sends such a model to a model server, as it
does every family but WR-ResNet (cpx/ml_tools/interpreter.py).  Only a WR-ResNet-22-4 exported to float32 TFLite is read:
get_interpreter on such a path converts on load (load_tflite below); tools/tflite_to_npz.py writes the same arrays to an
.npz.

What is read: the float32 graph of WR-ResNet-22-4 (src/ml_tools/resnet/wr_resnet.py:5-98) as the TFLite converter
writes it -- CONV_2D (filter OHWI, bias, fused ReLU: a convolution with the BatchNorm that follows it folded in), MUL +
ADD by per-channel constants (a BatchNorm that follows a residual ADD cannot be folded: scale and shift), RELU, ADD of
two activations (the residual), MEAN (global average pooling), FULLY_CONNECTED, LOGISTIC / SOFTMAX.  The walk follows
the operators in order and fills the Keras-layout names; a folded or affine-only BatchNorm becomes gamma = scale,
beta = shift, moving_mean = 0, moving_variance = 1 - eps (so that the loader's gamma / sqrt(var + eps) gives the scale
back exactly).  Anything else (quantised tensors, another topology) is refused with the operator that stopped the walk."""
import struct

import numpy as np

BN_EPS = np.float32(1e-3)
OPS = {0: "ADD", 3: "CONV_2D", 9: "FULLY_CONNECTED", 14: "LOGISTIC", 18: "MUL", 19: "RELU", 25: "SOFTMAX", 40: "MEAN",
       22: "RESHAPE", 34: "PAD"}


# ---- flatbuffer access ---------------------------------------------------------------------------------------
class Table:
    def __init__(self, buf, pos):
        self.buf, self.pos = buf, pos
        self.vt = pos - struct.unpack_from("<i", buf, pos)[0]
        self.vt_size = struct.unpack_from("<H", buf, self.vt)[0]

    def _off(self, field):
        o = 4 + 2 * field
        if o >= self.vt_size:
            return 0
        return struct.unpack_from("<H", self.buf, self.vt + o)[0]

    def scalar(self, field, fmt, default=0):
        o = self._off(field)
        return struct.unpack_from("<" + fmt, self.buf, self.pos + o)[0] if o else default

    def _indirect(self, field):
        o = self._off(field)
        if not o:
            return None
        loc = self.pos + o
        return loc + struct.unpack_from("<I", self.buf, loc)[0]

    def table(self, field):
        p = self._indirect(field)
        return None if p is None else Table(self.buf, p)

    def string(self, field):
        p = self._indirect(field)
        if p is None:
            return None
        n = struct.unpack_from("<I", self.buf, p)[0]
        return bytes(self.buf[p + 4:p + 4 + n]).decode("utf-8", "replace")

    def vector(self, field, fmt=None):
        """fmt: struct code of scalar elements, or None for a vector of tables."""
        p = self._indirect(field)
        if p is None:
            return []
        n = struct.unpack_from("<I", self.buf, p)[0]
        if fmt is not None:
            return list(struct.unpack_from("<%d%s" % (n, fmt), self.buf, p + 4))
        out = []
        for i in range(n):
            loc = p + 4 + 4 * i
            out.append(Table(self.buf, loc + struct.unpack_from("<I", self.buf, loc)[0]))
        return out

    def bytes_vector(self, field):
        p = self._indirect(field)
        if p is None:
            return b""
        n = struct.unpack_from("<I", self.buf, p)[0]
        return bytes(self.buf[p + 4:p + 4 + n])


class Graph:
    """Tensors (shape, constant data), operators (name, inputs, outputs, options) of subgraph 0."""

    def __init__(self, data):
        buf = memoryview(data)
        if len(data) < 8 or bytes(buf[4:8]) != b"TFL3":
            raise ValueError("not a TFLite flatbuffer (file identifier TFL3 missing)")
        model = Table(buf, struct.unpack_from("<I", buf, 0)[0])
        codes = []
        for oc in model.vector(1):
            code = oc.scalar(3, "i", 0)
            if code == 0:
                code = oc.scalar(0, "b", 0)   # files written before builtin_code was widened
            codes.append(code)
        buffers = [b.bytes_vector(0) for b in model.vector(4)]
        sub = model.vector(2)[0]
        self.tensors = []
        for t in sub.vector(0):
            shape = t.vector(0, "i")
            ttype = t.scalar(1, "b", 0)
            raw = buffers[t.scalar(2, "I", 0)] if t.scalar(2, "I", 0) < len(buffers) else b""
            const = None
            if raw:
                if ttype == 0:
                    const = np.frombuffer(raw, "<f4").reshape(shape)
                elif ttype == 2:
                    const = np.frombuffer(raw, "<i4").reshape(shape)
                else:
                    raise NotImplementedError("tensor %r has type %d: only float32 models are read" % (t.string(3), ttype))
            self.tensors.append(dict(shape=shape, type=ttype, const=const, name=t.string(3)))
        self.inputs = sub.vector(1, "i")
        self.outputs = sub.vector(2, "i")
        self.ops = []
        for op in sub.vector(3):
            code = codes[op.scalar(0, "I", 0)]
            opts = op.table(4)
            o = dict(name=OPS.get(code, "OP_%d" % code), inputs=op.vector(1, "i"), outputs=op.vector(2, "i"), act=0)
            if opts is not None:
                if o["name"] == "CONV_2D":
                    o.update(padding=opts.scalar(0, "b", 0), stride_w=opts.scalar(1, "i", 1), stride_h=opts.scalar(2, "i", 1),
                             act=opts.scalar(3, "b", 0))
                elif o["name"] in ("ADD", "MUL", "FULLY_CONNECTED"):
                    o["act"] = opts.scalar(0, "b", 0)
            self.ops.append(o)

    def const(self, idx):
        return self.tensors[idx]["const"]


# ---- the walk: operators in order -> Keras-layout names -------------------------------------------------------
def identity_variance():
    """moving_variance v with float32(v + eps) == 1, so that gamma / sqrt(v + eps) == gamma."""
    v = np.float32(1) - BN_EPS
    for cand in (v, np.nextafter(v, np.float32(0)), np.nextafter(v, np.float32(2))):
        if np.float32(cand + BN_EPS) == np.float32(1):
            return np.float32(cand)
    raise AssertionError("no float32 variance gives 1 with eps")


def bn_params(scale, shift):
    c = scale.shape[0]
    return {"gamma": scale.astype(np.float32), "beta": shift.astype(np.float32),
            "moving_mean": np.zeros(c, np.float32), "moving_variance": np.full(c, identity_variance(), np.float32)}


def conv_kernel(g, op, groups=2):
    """TFLite OHWI [Cout, kh, kw, Cin/groups] -> Keras HWIO [kh, kw, Cin/groups, Cout]; bias."""
    w = g.const(op["inputs"][1])
    b = g.const(op["inputs"][2]) if len(op["inputs"]) > 2 and op["inputs"][2] >= 0 else None
    if w is None:
        raise ValueError("CONV_2D without a constant filter")
    k = np.ascontiguousarray(np.transpose(w, (1, 2, 3, 0))).astype(np.float32)
    return k, (np.zeros(w.shape[0], np.float32) if b is None else b.astype(np.float32))


def convert(g, blocks=3, filters=(16, 64, 128, 256)):
    ops = list(g.ops)
    pos = [0]

    def peek():
        return ops[pos[0]] if pos[0] < len(ops) else None

    def take(name):
        op = peek()
        if op is None or op["name"] != name:
            raise ValueError("operator %d: expected %s, found %s" % (pos[0], name, None if op is None else op["name"]))
        pos[0] += 1
        return op

    def affine(channels):
        """MUL const, ADD const (+ fused or separate RELU) -> (scale, shift)."""
        m = take("MUL")
        a = take("ADD")
        sc = next(g.const(i) for i in m["inputs"] if g.const(i) is not None).reshape(-1)
        sh = next(g.const(i) for i in a["inputs"] if g.const(i) is not None).reshape(-1)
        if sc.shape[0] != channels or sh.shape[0] != channels:
            raise ValueError("BatchNorm constants of %d / %d channels where %d were expected" % (sc.shape[0], sh.shape[0], channels))
        if a["act"] != 1:
            take("RELU")
        return sc, sh

    w = {}
    op = take("CONV_2D")
    w["conv1_1/kernel"], w["conv1_1/bias"] = conv_kernel(g, op)
    c_in = filters[0]
    for stage in (2, 3, 4):
        f = filters[stage - 1]
        for d in range(blocks):
            b = "%db%d" % (stage, d)
            sc, sh = affine(c_in)                       # pre-activation BatchNorm + ReLU of the block's input
            for k, v in bn_params(sc, sh).items():
                w["bn%s_branch2a/%s" % (b, k)] = v
            nxt = peek()
            shortcut = None
            if d == 0:
                # the 1x1 shortcut convolution reads the activated input; the converter may put it before or after
                # the main branch
                if nxt["name"] == "CONV_2D" and g.const(nxt["inputs"][1]).shape[1] == 1:
                    shortcut = take("CONV_2D")
            a = take("CONV_2D")                         # conv a with the following BatchNorm folded in, ReLU fused
            w["res%s_branch2a/kernel" % b], w["res%s_branch2a/bias" % b] = conv_kernel(g, a)
            if a["act"] != 1:
                take("RELU")
            for k, v in bn_params(np.ones(f, np.float32), np.zeros(f, np.float32)).items():
                w["bn%s_branch2b/%s" % (b, k)] = v
            cb = take("CONV_2D")
            w["res%s_branch2b/kernel" % b], w["res%s_branch2b/bias" % b] = conv_kernel(g, cb)
            if d == 0 and shortcut is None:
                shortcut = take("CONV_2D")
            if shortcut is not None:
                w["shortcut%d/kernel" % stage], w["shortcut%d/bias" % stage] = conv_kernel(g, shortcut)
            add = take("ADD")
            if add["act"] != 0:
                raise ValueError("operator %d: the residual ADD carries an activation" % (pos[0] - 1))
            c_in = f
    sc, sh = affine(c_in)
    for k, v in bn_params(sc, sh).items():
        w["final_bn/%s" % k] = v
    take("MEAN")
    n_hidden = 0
    while True:
        fc = take("FULLY_CONNECTED")
        wt = g.const(fc["inputs"][1])
        bias = g.const(fc["inputs"][2]) if len(fc["inputs"]) > 2 and fc["inputs"][2] >= 0 else np.zeros(wt.shape[0], np.float32)
        nxt = peek()
        last = nxt is None or nxt["name"] in ("LOGISTIC", "SOFTMAX")
        name = "prediction" if last else "dense_%d" % n_hidden
        w[name + "/kernel"] = np.ascontiguousarray(wt.T).astype(np.float32)   # TFLite [out, in] -> Keras [in, out]
        w[name + "/bias"] = bias.astype(np.float32)
        if last:
            w["prediction/activation"] = "softmax" if (nxt is not None and nxt["name"] == "SOFTMAX") else "sigmoid"
            break
        if fc["act"] != 1:
            take("RELU")
        n_hidden += 1
    return w


def load_tflite(path):
    """-> the weights dict of cpx.ml_tools.wrresnet (what load_weights returns for an .npz) of a .tflite model file."""
    with open(str(path), "rb") as fh:
        g = Graph(fh.read())
    return convert(g)
